"""GPU-box target for rocprofv3: 10 launches each of the hand-written GEMM kernels at the S-pubmed shapes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fit-gnn_amd"))
import torch
from fitgnn_amd import ops, _lib

torch.manual_seed(0)
a = torch.randn(90549, 512, device="cuda"); b = torch.randn(90549, 512, device="cuda")
w = torch.randn(512, 512, device="cuda"); out = torch.randn(90549, 512, device="cuda")
for _ in range(10):
    ops.gemm_atb(a, b)
    ops.gemm_nt(a, w)
    ops.gemm_nt_epilogue_bwd(a, w, out, _lib.EPI_ELU | _lib.EPI_DROPOUT, p=0.5, seed=1234)
torch.cuda.synchronize()
