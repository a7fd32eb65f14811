#!/usr/bin/env python3
"""GPU-box tool: the SpMM kernels on a synthetic batch of single-centre stars (blocks of `rows` rows: row 0 linked to all
the others), R rows in all: tiled (window-sized tiles) against the whole-subgraph kernel.  args: rows-per-star [total rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from fitgnn_amd import ops
from fitgnn_amd.csr import CSRGraph

S = int(sys.argv[1]) if len(sys.argv) > 1 else 51
R = (int(sys.argv[2]) if len(sys.argv) > 2 else 8_000_000) // S * S
dev = torch.device("cuda")
nb = R // S
base = torch.arange(nb, device=dev).repeat_interleave(S - 1) * S
leaf = base + (torch.arange(nb * (S - 1), device=dev) % (S - 1)) + 1
ei = torch.stack([torch.cat([base, leaf]), torch.cat([leaf, base])])
ptr = np.arange(0, R + 1, S)
g = CSRGraph(ei, R, mode="gcn", ptr=ptr, block_limit=1 << 20)
X = torch.randn(R, 512, device=dev); Y = torch.empty_like(X)
bytes_ = 8 * 512 * R + 8 * g.nnz + 4 * (R + 1)
def timeit(fn, n=8):
    fn(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, cfg in (("whole-subgraph", ops.OpConfig(split_large_blocks=True)), ("tiled", ops.OpConfig(split_large_blocks=False))):
    t = min(timeit(lambda: ops.spmm_graph(g, X, out=Y, cfg=cfg)) for _ in range(3))
    print(f"stars of {S} rows, R={R}: {name:15s} {t*1e3:9.1f} us  {bytes_/t/1e6:7.0f} GB/s ({bytes_/t/1e6/8000:.3f} of 8 TB/s)", flush=True)
