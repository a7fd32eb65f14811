#!/bin/bash
# GPU-box script: bench line + rocprofv3 kernel stats of a secondary operator line (bench.py --layer GATConv | APPNP).
# usage: tools/profile_layer.sh <tag> <layer> <workload> [extra bench.py args]
set -e
tag=$1; layer=$2; wl=$3; shift; shift; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$tag
mkdir -p $OUT
python3 $R/bench.py --layer $layer --workload $wl "$@" > $OUT/bench_${layer}_${wl}.json 2> $OUT/bench_${layer}_${wl}.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_${layer}_${wl} -o run -- python3 $R/bench.py --layer $layer --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --gpu-warm-seconds 0 "$@" > $OUT/rocprof_${layer}_${wl}.json 2> $OUT/rocprof_${layer}_${wl}.err
rm -f $OUT/stats_${layer}_${wl}/run_kernel_trace.csv
echo "$layer $wl done"
