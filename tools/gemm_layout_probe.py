#!/usr/bin/env python3
"""GPU-box probe: the three big products of the step under the 'high' policy, by operand layout."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import torch
from fitgnn_amd import ops
R, H = 90549, 512
dev = "cuda"
x = torch.randn(R, H, device=dev) * 0.1; W = torch.randn(H, H, device=dev) * 0.05
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
Wt = W.t().contiguous()
print("x @ W^T  (W row-major, as the forward)      ", min(timeit(lambda: ops.mm(x, W.t())) for _ in range(3)))
print("x @ W    (W row-major, as dX)               ", min(timeit(lambda: ops.mm(x, W)) for _ in range(3)))
print("x @ Wt^T (Wt = W^T materialised: same value)", min(timeit(lambda: ops.mm(x, Wt.t())) for _ in range(3)))
print("max diff", float((ops.mm(x, W) - ops.mm(x, Wt.t())).abs().max()))
for B in (32, 64, 128):
    Kc = R // B
    a = x[:B * Kc].view(B, Kc, H); b = x[:B * Kc].view(B, Kc, H)
    torch.set_float32_matmul_precision("high")
    print(f"bmm split-K B={B}", min(timeit(lambda: torch.bmm(a.transpose(1, 2), b)) for _ in range(3)))
    torch.set_float32_matmul_precision("highest")
print("-- split-K chunk sizes (weight gradient dH^T @ X, K = 90549 rows)")
torch.set_float32_matmul_precision("high")
y = torch.randn(R, H, device=dev) * 0.1
for Kc in (1024, 1408, 1414, 1536, 2048, 2816, 4096):
    B = R // Kc
    a = x[:B * Kc].view(B, Kc, H); b = y[:B * Kc].view(B, Kc, H)
    t1 = min(timeit(lambda: torch.bmm(a.transpose(1, 2), b)) for _ in range(3))
    print(f"Kc={Kc:5d} B={B:3d} tail={R - B * Kc:5d}: bmm {t1:7.1f} us")
