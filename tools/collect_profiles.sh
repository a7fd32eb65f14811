#!/bin/bash
# GPU-box script: the artefacts under profiles/ (bench line, rocprofv3 kernel stats of the same command, PMC passes
# for the SpMM kernel).  usage: tools/collect_profiles.sh <tag>
set -e
tag=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
python3 $R/bench.py > $R/gpurun_out/$tag/bench.json 2> $R/gpurun_out/$tag/bench.err
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/stats -o run -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/$tag/bench_under_rocprof.json 2> $R/gpurun_out/$tag/rocprof.err
echo "stats done"
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $set | cut -c1-8 | tr ' ' '_')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/$tag/pmc_$n -o run -- python3 $R/tools/prof_spmm.py lds 16 10 0 > $R/gpurun_out/$tag/pmc_$n.log 2>&1
  echo "pmc $n done"
done
