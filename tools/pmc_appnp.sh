#!/bin/bash
# GPU-box script: LDS / wait / HBM counters of the column-sliced APPNP kernel, per launch shape (one --pmc pass per counter set,
# --kernel-trace only).  usage: tools/pmc_appnp.sh <tag> [extra bench.py args]
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o run -- python3 $R/bench.py --layer APPNP --steps 2 --warmup 1 --no-cpu-baseline --gpu-warm-seconds 0 "$@" > $OUT/p$i.json 2> $OUT/p$i.err || echo "set $i failed"
  python3 - "$OUT/p$i" <<'PY' >> $OUT/summary.txt
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "appnp_lds_kernel" in r["Kernel_Name"]:
            bwd = "<true>" in r["Kernel_Name"]
            acc[(r["Counter_Name"], "bwd" if bwd else "fwd", int(r["Grid_Size"]) // int(r["Workgroup_Size"]), int(r["Workgroup_Size"]))].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k[0]:24s} {k[1]} workgroups {k[2]:6d} x {k[3]:4d} threads: mean per dispatch {sum(v) / len(v):.4g} ({len(v)} dispatches)")
PY
  rm -rf $OUT/p$i
done
cat $OUT/summary.txt
