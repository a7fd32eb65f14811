#!/bin/bash
# GPU-box script: the rocprofv3 evidence behind profiles/ for one round -- kernel-trace stats and PMC passes of bench.py
# itself (the default workload, S-products).  usage: tools/profile_round.sh <tag> [extra bench.py args]
# Counters are collected in passes of their own (--pmc with --kernel-trace only), one counter set per pass.
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --gpu-warm-seconds 0 --no-bf16x3 --no-all-rows "$@" > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
echo "stats done"
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $set | cut -c1-8 | tr ' ' '_')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$n -o run -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --gpu-warm-seconds 0 --no-bf16x3 --no-all-rows "$@" > $OUT/pmc_$n.json 2> $OUT/pmc_$n.err
  echo "pmc $n done"
done
python3 $R/tools/summarize_pmc.py $OUT > $OUT/pmc_summary.json
for d in $OUT/pmc_*/; do rm -rf "$d"; done   # per-dispatch CSVs are large; the summary travels back
rm -f $OUT/stats/run_kernel_trace.csv
cp $OUT/pmc_summary.json $OUT/pmc_bench_kernels.json; echo "summary done"
