#!/bin/bash
# GPU-box script: instruction-mix / wait counters of one kernel of bench.py (one --pmc pass per counter set, --kernel-trace only).
# usage: tools/pmc_kernel.sh <tag> <kernel-name-substring> [extra bench.py args]
set -e
tag=$1; kern=$2; shift; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --gpu-warm-seconds 0 --no-bf16x3 --no-all-rows "$@" > $OUT/p$i.json 2> $OUT/p$i.err || echo "set $i failed"
  python3 - "$OUT/p$i" "$kern" <<'PY' >> $OUT/summary.txt
import csv, glob, sys, collections
d, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, "mean per dispatch", sum(v) / len(v), "dispatches", len(v))
PY
  rm -rf $OUT/p$i
done
cat $OUT/summary.txt
