#!/bin/bash
# usage: tools/pmc_passes.sh <tag> <python script + args...>   (run on the GPU box; one rocprofv3 --pmc pass per counter set)
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$i -o run -- python3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$i.log 2>&1
  echo "pass $i done"
done
