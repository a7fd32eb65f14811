#!/usr/bin/env python3
"""GPU-box tool: layer 0's table SpMM (row indirection) on the S-products union with and without the per-entry table index (xcol)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from fitgnn_amd import ops, workloads, data

name = sys.argv[1] if len(sys.argv) > 1 else "S-products"
dev = torch.device("cuda:0")
co = workloads.coarsen_workload(name, dev)
ei_d = torch.from_numpy(co["ei"]).to(dev)
sub, _ = workloads.assemble(name, ei_d, torch.from_numpy(co["assign"]).to(dev), co["n_clusters"])
b = workloads.batch_from_subgraphs(name, sub, dev)
g = b.graph
N = workloads.SHAPES[name][0]
table = torch.randn(N, 512, device=dev)
xrow = b.row_index.index
Y = torch.empty(g.n, 512, device=dev)
def run(with_xcol, reps=5):
    side = g.f
    if not with_xcol:
        orig = ops.spmm_blocks_raw
        def no_xcol(*a, **k):
            k["xcol"] = None
            return orig(*a, **k)
        ops.spmm_blocks_raw = no_xcol
    try:
        for _ in range(2): ops.spmm_graph(g, table, xrow=xrow, out=Y)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(reps): ops.spmm_graph(g, table, xrow=xrow, out=Y)
        torch.cuda.synchronize()
        return (time.time() - t) / reps * 1e3
    finally:
        if not with_xcol: ops.spmm_blocks_raw = orig
for rnd in range(3):
    print(f"round {rnd}: with xcol {run(True):.3f} ms, without {run(False):.3f} ms")
X = torch.randn(g.n, 512, device=dev)
for _ in range(2): ops.spmm_graph(g, X, out=Y)
torch.cuda.synchronize(); t = time.time()
for _ in range(5): ops.spmm_graph(g, X, out=Y)
torch.cuda.synchronize(); print(f"plain operand (no indirection): {(time.time()-t)/5*1e3:.3f} ms")
# one launch per 256-column slab: the table slab (169 MB at S-products) then fits the 256-MB memory-side cache
side = g.f
xcol = side.xcol[1]
def halves(reps=5):
    def once():
        for c0 in (0, 256):
            ops.spmm_blocks_raw(side.rowptr, side.col, side.val, side.blocks, side.long_rows, table[:, c0:c0 + 256], Y[:, c0:c0 + 256], xrow=xrow, xcol=xcol)
    for _ in range(2): once()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(reps): once()
    torch.cuda.synchronize()
    return (time.time() - t) / reps * 1e3
print(f"small tiles: {side.small_tiles.shape[0]} (not included below)")
for rnd in range(3):
    print(f"round {rnd}: two launches, one per column slab: {halves():.3f} ms")
