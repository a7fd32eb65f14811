#!/bin/bash
# usage: tools/pmc_spmm.sh <tag> <workload>   (GPU box; one rocprofv3 --pmc pass per counter set over tools/prof_spmm.py)
set -e
tag=$1; wl=$2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}/stats -o run -- python3 $R/tools/prof_spmm.py $wl 5 > $R/gpurun_out/${tag}_stats.log 2>&1
rm -f $R/gpurun_out/${tag}/stats/run_kernel_trace.csv
for set in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/${tag}/pmc_$i -o run -- python3 $R/tools/prof_spmm.py $wl 5 > $R/gpurun_out/${tag}_pmc_$i.log 2>&1
  echo "pass $i done"
done
python3 $R/tools/summarize_pmc.py $R/gpurun_out/${tag} > $R/gpurun_out/${tag}/pmc_summary.json
rm -rf $R/gpurun_out/${tag}/pmc_[0-9]*   # the per-dispatch CSVs are large; the summary is what travels back
