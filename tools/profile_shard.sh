#!/bin/bash
# GPU-box script: rocprofv3 kernel-trace stats of ONE rank's step of an N-rank job (bench.py --shard heaviest/N): which kernels make
# up the part of the step that does not shrink with N.  usage: tools/profile_shard.sh <tag> <N> [extra bench.py args]
set -e
tag=$1; n=$2; shift; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $R/bench.py --shard heaviest/$n --steps 10 --warmup 3 --no-cpu-baseline --gpu-warm-seconds 0 --no-bf16x3 --no-all-rows --no-pruned "$@" > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
rm -f $OUT/stats/run_kernel_trace.csv
echo "stats done"
