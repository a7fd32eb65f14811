#!/usr/bin/env python3
"""GPU-box tool: time the folded backward (fitgnn_spmm_epilogue_bwd_f32) against the two-kernel path on the S-pubmed
union, variant by variant (epilogue flags, db on/off, head on/off)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import bench
from fitgnn_amd import ops
from fitgnn_amd.csr import CSRGraph
from fitgnn_amd._lib import EPI_DROPOUT, EPI_ELU

dev = torch.device("cuda")
batch, _, info = bench.build_workload("S-pubmed", 0, dev)
R, H, C = batch.n_rows, 512, 3
g = CSRGraph(batch.edge_index, R, mode="gcn", ptr=batch.ptr)
out = torch.nn.functional.elu(torch.randn(R, H, device=dev))
dOut, dy, Wl = torch.randn(R, H, device=dev), torch.randn(R, C, device=dev), torch.randn(C, H, device=dev)

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for head in (False, True):
    for epi, name in ((0, "none"), (EPI_ELU, "elu"), (EPI_ELU | EPI_DROPOUT, "elu+drop")):
        for want_db in (False, True):
            row = []
            for fold in (True, False):
                ops.FOLD_BACKWARD = fold
                fn = lambda: ops.layer_backward(g, out, epi, 0.5, 77, None, want_db, dOut=None if head else dOut,
                                                dy=dy if head else None, Wl=Wl if head else None, want_dWl=head and want_db)
                row.append(min(timeit(fn) for _ in range(3)))
            print(f"head={head} epi={name:8s} db={want_db}: fold {row[0]:.1f} us   two-kernel {row[1]:.1f} us", flush=True)
