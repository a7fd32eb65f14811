#!/usr/bin/env python3
"""GPU-box tool: one fitgnn_gemm_exact_f32 shape launched N times (for rocprofv3 --pmc passes: tools/pmc_gemm.sh).
usage: python tools/gemm_one.py nt|nn|tn I J K [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fit-gnn_amd"))
import torch
from fitgnn_amd import ops

form, I, J, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
if form == "nt": a, b = torch.randn(I, K, device="cuda"), torch.randn(J, K, device="cuda")
elif form == "nn": a, b = torch.randn(I, K, device="cuda"), torch.randn(K, J, device="cuda")
else: a, b = torch.randn(K, I, device="cuda"), torch.randn(K, J, device="cuda")
for _ in range(reps):
    ops.gemm_exact(a, b, form)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.gemm_exact(a, b, form)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print(f"{form} ({I}, {J}, {K}): {us:.1f} us, {2.0 * I * J * K / us / 1e6:.1f} TFLOP/s")
