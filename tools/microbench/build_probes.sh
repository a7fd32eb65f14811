#!/bin/bash
# Builds the GPU-box microbenchmarks next to their sources (gfx950; hipcc cross-compiles without a GPU):
#   gemm_probe_<variant>  gemm_nt_kernel<4,false,true> at the bench's size with parts compiled out
#   mfma_peak             what v_mfma_f32_32x32x16_bf16 sustains, with and without LDS reads beside it
#   copy_probe            read-once / write-once streams by kernel shape
set -e
cd "$(dirname "$0")/../.."
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I include -I fit-gnn_amd/csrc"
$H tools/microbench/gemm_probe.hip -o tools/microbench/gemm_probe_full
$H -DPROBE_NO_MFMA tools/microbench/gemm_probe.hip -o tools/microbench/gemm_probe_no_mfma
$H -DPROBE_NO_ALOAD -DPROBE_NO_STORE tools/microbench/gemm_probe.hip -o tools/microbench/gemm_probe_no_hbm
$H -DPROBE_NO_LDSREAD -DPROBE_NO_ALOAD -DPROBE_NO_STORE tools/microbench/gemm_probe.hip -o tools/microbench/gemm_probe_no_hbm_no_ldsread
$H -DPROBE_NO_MFMA -DPROBE_NO_ALOAD -DPROBE_NO_STORE tools/microbench/gemm_probe.hip -o tools/microbench/gemm_probe_lds_valu_only
$H tools/microbench/mfma_peak.hip -o tools/microbench/mfma_peak
$H tools/microbench/copy_probe.hip -o tools/microbench/copy_probe
echo built
