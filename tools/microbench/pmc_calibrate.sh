#!/bin/bash
# GPU box: what FETCH_SIZE / WRITE_SIZE report for a copy whose bytes are known (tools/microbench/copy_probe, 8 GiB in, 8 GiB out)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/cal_$set -o run -- $R/tools/microbench/copy_probe 8192 > /dev/null 2>&1
  python3 - "$set" <<'PY'
import csv, glob, sys
c = sys.argv[1]
acc = {}
for f in glob.glob(f"/tmp/cal_{c}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-60:]
        acc.setdefault(k, []).append(float(row["Counter_Value"]))
for k, v in acc.items():
    print(f"{c} {k:62s} mean {sum(v)/len(v):.4g} KiB  x1024 = {sum(v)/len(v)*1024/2**30:.3f} GiB  (n={len(v)})")
PY
done
