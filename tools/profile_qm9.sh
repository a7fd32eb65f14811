#!/bin/bash
# GPU-box script: the S-qm9 bench line (config 5) + rocprofv3 kernel stats of a shorter run.  usage: tools/profile_qm9.sh <tag> [bench args]
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$tag
mkdir -p $OUT
python3 $R/bench.py --workload S-qm9 "$@" > $OUT/bench_S-qm9.json 2> $OUT/bench_S-qm9.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_S-qm9 -o run -- python3 $R/bench.py --workload S-qm9 --steps 2 --warmup 1 --no-cpu-baseline --gpu-warm-seconds 0 "$@" > $OUT/rocprof_S-qm9.json 2> $OUT/rocprof_S-qm9.err
rm -f $OUT/stats_S-qm9/run_kernel_trace.csv
echo "S-qm9 done"
