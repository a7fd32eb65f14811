#!/usr/bin/env python3
"""GPU-box tool: do an MFMA-bound and an HBM-bound kernel of the step overlap when issued on two HIP streams of one process?
x @ W^T (gemm_nt) on 4.1 M rows and the whole-subgraph SpMM on a 4.1 M-row star batch: each alone, back to back on one
stream, and concurrently on two streams."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from fitgnn_amd import ops
from fitgnn_amd.csr import CSRGraph

dev = torch.device("cuda")
S, R = 51, 4_100_000 // 51 * 51
nb = R // S
base = torch.arange(nb, device=dev).repeat_interleave(S - 1) * S
leaf = base + (torch.arange(nb * (S - 1), device=dev) % (S - 1)) + 1
ei = torch.stack([torch.cat([base, leaf]), torch.cat([leaf, base])])
g = CSRGraph(ei, R, mode="gcn", ptr=np.arange(0, R + 1, S), block_limit=1 << 20)
X = torch.randn(R, 512, device=dev); Y = torch.empty_like(X)
A = torch.randn(R, 512, device=dev); W = torch.randn(512, 512, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def gemm(): return ops.gemm_nt(A, W)
def spmm(): return ops.spmm_graph(g, X, out=Y)
def wall(fn, n=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def both_streams():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): gemm()
    with torch.cuda.stream(s2): spmm()
    cur.wait_stream(s1); cur.wait_stream(s2)
def two_gemms():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): gemm()
    with torch.cuda.stream(s2): gemm()
    cur.wait_stream(s1); cur.wait_stream(s2)
print(f"gemm alone {wall(gemm):.2f} ms | spmm alone {wall(spmm):.2f} ms | back to back {wall(lambda: (gemm(), spmm())):.2f} ms | "
      f"two streams {wall(both_streams):.2f} ms | two gemms on two streams {wall(two_gemms):.2f} ms (2 x alone = {2 * wall(gemm):.2f})", flush=True)
