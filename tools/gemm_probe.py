#!/usr/bin/env python3
"""GPU-box probe: time the dense GEMM shapes of the S-pubmed GD step under different BLAS backends."""
import os, sys, time
import torch

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

R, F, H, C = 90549, 500, 512, 3
dev = "cuda"
X = torch.randn(R, F, device=dev); W0 = torch.randn(H, F, device=dev); X1 = torch.randn(R, H, device=dev)
W1 = torch.randn(H, H, device=dev); Wl = torch.randn(C, H, device=dev); dH = torch.randn(R, H, device=dev)
shapes = {
    "fwd0  X[R,500] @ W0^T": lambda: torch.mm(X, W0.t()),
    "fwd1  X1[R,512] @ W1^T": lambda: torch.mm(X1, W1.t()),
    "dW0   dH^T[512,R] @ X[R,500]": lambda: torch.mm(dH.t(), X),
    "dW1   dH^T @ X1": lambda: torch.mm(dH.t(), X1),
    "dX1   dH[R,512] @ W1": lambda: torch.mm(dH, W1),
    "lt1   X1 @ Wl^T [R,3]": lambda: torch.mm(X1, Wl.t()),
}
flops = {"fwd0": 2*R*F*H, "fwd1": 2*R*H*H, "dW0": 2*R*F*H, "dW1": 2*R*H*H, "dX1": 2*R*H*H, "lt1": 2*R*H*C}
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
if mode == "rocblas":
    torch.backends.cuda.preferred_blas_library("cublas")
elif mode == "hipblaslt":
    torch.backends.cuda.preferred_blas_library("cublaslt")
elif mode == "high":
    torch.set_float32_matmul_precision("high")
elif mode == "tf32":
    torch.backends.cuda.matmul.allow_tf32 = True
print("mode", mode, "blas", torch.backends.cuda.preferred_blas_library())
ref = torch.mm(X1.double()[:2048], W1.double().t())
for k, fn in shapes.items():
    us = timeit(fn)
    print(f"{k:32s} {us:8.1f} us  {flops[k.split()[0]]/us/1e6:7.1f} TFLOP/s")
err = (torch.mm(X1[:2048], W1.t()).double() - ref).abs().max() / ref.abs().max()
print("rel err vs fp64:", float(err))
