#!/usr/bin/env python3
"""GPU-box probe: algorithmic GB/s of the SpMM on the S-pubmed union as a function of the operand width H (how much of the
gap to the copy ceiling comes from splitting 2-KiB rows into two 1-KiB slabs?), plus an identity pattern (pure tiled copy)."""
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
def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
g = CSRGraph(batch.edge_index, R, mode="gcn", ptr=batch.ptr)
empty = torch.zeros((2, 0), dtype=torch.long, device=dev)
gi = CSRGraph(empty, R, mode="gcn", ptr=batch.ptr)      # self loops only: Y = X, a tiled copy
for H in (128, 256, 512, 1024, 2048):
    X = torch.randn(R, H, device=dev)
    for name, gg in (("union", g), ("identity", gi)):
        nnz = int(gg.f.col.numel())
        us = min(timeit(lambda: ops.spmm_graph(gg, X)) for _ in range(3))
        byt = 8 * H * R + 8 * nnz + 4 * (R + 1)
        print(f"H={H:5d} {name:9s}: {us:7.1f} us  {byt / us / 1e3:7.1f} GB/s algorithmic", flush=True)
    src = torch.empty_like(X); 
    us = min(timeit(lambda: src.copy_(X)) for _ in range(3))
    print(f"H={H:5d} torch copy: {us:7.1f} us  {8 * H * R / us / 1e3:7.1f} GB/s", flush=True)
