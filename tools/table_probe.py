#!/usr/bin/env python3
"""GPU-box tool: what the layer-0 launch on the de-duplicated table pays for, on a workload's union (HIP events, interleaved rounds):
the launch as the step issues it / the same kernel reading a SEQUENTIAL table (operand row of union row r = row r of an [R, H] array:
no random gather) / the plain kernel with the same epilogue / both without the epilogue.   python tools/table_probe.py S-products"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

from fitgnn_amd import ops, workloads
from fitgnn_amd._lib import EPI_BIAS, EPI_DROPOUT, EPI_ELU


def timeit(fn, n=8):
    fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "S-products"
    dev = torch.device("cuda")
    w0 = workloads.coarsen_workload(wl, dev)
    sub, _ = workloads.assemble(wl, torch.from_numpy(w0["ei"]).to(dev), torch.from_numpy(w0["assign"]).to(dev), w0["n_clusters"])
    batch = workloads.batch_from_subgraphs(wl, sub, dev)
    g, H = batch.graph, 512
    R, n_table = g.n, int(batch.x_table.shape[0])
    xrow = batch.row_index.index
    T = torch.randn(n_table, H, device=dev)
    Tseq = torch.randn(R, H, device=dev)
    ar = torch.arange(R, dtype=torch.int32, device=dev)
    b = torch.randn(H, device=dev)
    epi = dict(bias=b, epilogue=EPI_BIAS | EPI_ELU | EPI_DROPOUT, p=0.5, seed=11)
    cases = {
        "table, epilogue (the step's launch)": lambda: ops.spmm_graph(g, T, xrow=xrow, **epi),
        "sequential table, epilogue": lambda: ops.spmm_graph(g, Tseq, xrow=ar, **epi),
        "plain operand, epilogue": lambda: ops.spmm_graph(g, Tseq, **epi),
        "table, no epilogue": lambda: ops.spmm_graph(g, T, xrow=xrow),
        "sequential table, no epilogue": lambda: ops.spmm_graph(g, Tseq, xrow=ar),
        "plain operand, no epilogue": lambda: ops.spmm_graph(g, Tseq),
    }
    print(f"{wl}: rows {R}, table rows {n_table}, nnz' {int(g.f.col.numel())}", flush=True)
    res = {k: [] for k in cases}
    for rnd in range(3):
        for k, fn in cases.items():
            res[k].append(timeit(fn))
    for k, v in res.items():
        print(f"  {k:40s} {min(v):9.1f} us (rounds: {', '.join('%.0f' % t for t in v)})")


if __name__ == "__main__":
    main()
