#!/bin/bash
# GPU-box script: matrix-pipe and clock counters of the exact-fp32 GEMM kernel on one shape (one --pmc pass per counter set,
# --kernel-trace only).  usage: tools/pmc_gemm.sh <tag> nt|nn|tn I J K
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o run -- python3 $R/tools/gemm_one.py "$@" > $OUT/p$i.log 2> $OUT/p$i.err || echo "set $i failed"
  python3 - "$OUT/p$i" <<'PY' >> $OUT/summary.txt
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(list)
dur = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_f32_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_f32_kernel" in r["Kernel_Name"]:
            dur["dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, v in acc.items():
    print(k, "mean per dispatch", sum(v) / len(v), "dispatches", len(v))
for k, v in dur.items():
    print(k, "mean", sum(v) / len(v), "n", len(v))
PY
  rm -rf $OUT/p$i
done
cat $OUT/summary.txt
