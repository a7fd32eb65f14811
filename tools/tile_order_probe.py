#!/usr/bin/env python3
"""GPU-box probe: does starting the miss-heavy tiles (hub subgraphs cut into pieces) first shorten the SpMM's tail?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import bench
from fitgnn_amd import _lib, ops
from fitgnn_amd.csr import CSRGraph

dev = torch.device("cuda")
batch, _, info = bench.build_workload("S-pubmed", 0, dev)
R = batch.n_rows
g = CSRGraph(batch.edge_index, R, mode="gcn", ptr=batch.ptr)
X = torch.randn(R, 512, device=dev)
def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
f = g.f
t = f.tiles.cpu().numpy().copy()            # [T, 8]: row_begin,row_end,win_begin,win_rows,nnz_begin,nnz_end,..
rowptr, col = f.rowptr.cpu().numpy(), f.col.cpu().numpy()
live = t[:, 1] > t[:, 0]
miss = np.zeros(len(t), dtype=np.int64)
for i in np.nonzero(live)[0]:
    c = col[t[i, 4]:t[i, 5]]
    miss[i] = int(((c < t[i, 2]) | (c >= t[i, 2] + min(t[i, 3], 16))).sum())
print("tiles", int(live.sum()), "with misses", int((miss > 0).sum()), "total misses", int(miss.sum()), "max per tile", int(miss.max()),
      "nnz", len(col))
base = min(timeit(lambda: ops.spmm_graph(g, X)) for _ in range(3))
# reorder inside every XCD's sequence (positions p % 8 == k): heaviest (misses, then nnz) first
out = t.copy()
for k in range(8):
    seg = t[k::8]
    key = miss[k::8] * 1000 + (seg[:, 5] - seg[:, 4])
    out[k::8] = seg[np.argsort(-key, kind="stable")]
f.tiles = torch.from_numpy(out).to(dev)
y_ref = None
new = min(timeit(lambda: ops.spmm_graph(g, X)) for _ in range(3))
print(f"default order {base:.1f} us   heavy-first {new:.1f} us")
