#!/usr/bin/env python3
"""GPU-box probe: folded backward with the gradient stream aliased to `out` (second stream cache-resident)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import torch
import bench
from fitgnn_amd import ops
from fitgnn_amd.csr import CSRGraph

dev = torch.device("cuda")
batch, _, info = bench.build_workload("S-pubmed", 0, dev)
R, H = batch.n_rows, 512
g = CSRGraph(batch.edge_index, R, mode="gcn", ptr=batch.ptr)
print("tiles", g.t.tiles.shape, "window", g.window_rows)
out = torch.randn(R, H, device=dev); dOut = torch.randn(R, H, device=dev)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
ops.FOLD_BACKWARD = True
for name, d in (("two streams", dOut), ("aliased", out)):
    print(name, min(timeit(lambda: ops.layer_backward(g, out, 0, 0.5, 7, None, False, dOut=d)) for _ in range(3)))
print("plain spmm T", min(timeit(lambda: ops.spmm_graph(g, dOut, transposed=True)) for _ in range(3)))
print("plain spmm F", min(timeit(lambda: ops.spmm_graph(g, dOut)) for _ in range(3)))
