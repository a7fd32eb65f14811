#!/usr/bin/env python3
"""GPU-box tool: time the SpMM kernel on the S-pubmed union for several LDS window sizes (interleaved
rounds in one process, HIP events), next to a device copy of the same byte count."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

import bench
from fitgnn_amd import ops
from fitgnn_amd._lib import EPI_BIAS, EPI_DROPOUT, EPI_ELU
from fitgnn_amd.csr import CSRGraph


def timeit(fn, n=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "S-pubmed"
    windows = [int(w) for w in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 16, 24, 32, 48, 64, 96]
    dev = torch.device("cuda")
    if wl == "identity":  # structural ceiling: A = I (self loops only), every row read once and written once
        import types
        R0 = 90549
        batch = types.SimpleNamespace(n_rows=R0, nnz=R0, edge_index=torch.zeros((2, 0), dtype=torch.long, device=dev),
                                      ptr=np.arange(R0 + 1))
        info = {"workload": "identity"}
    else:
        batch, (F, C), info = bench.build_workload(wl, 0, dev)
    H, R = 512, batch.n_rows
    X = torch.randn(R, H, device=dev)
    Y = torch.empty_like(X)
    b = torch.randn(H, device=dev)
    bytes_spmm = 8 * H * R + 8 * batch.nnz + 4 * (R + 1)
    graphs = {w: CSRGraph(batch.edge_index, R, mode="gcn", ptr=batch.ptr, lds_rows=w, planned=True) for w in windows}
    graphs_c = {w: CSRGraph(batch.edge_index, R, mode="gcn", ptr=batch.ptr, lds_rows=w, planned=False) for w in windows}
    res = {w: {"plain": [], "epi": [], "T": [], "gplain": [], "gepi": []} for w in windows}
    GA = 0x100
    copy = []
    for rnd in range(5):
        for w, g in graphs.items():
            gc = graphs_c[w]
            res[w]["plain"].append(timeit(lambda: ops.spmm_graph(g, X, out=Y)))
            res[w]["epi"].append(timeit(lambda: ops.spmm_graph(g, X, out=Y, bias=b, epilogue=EPI_BIAS | EPI_ELU | EPI_DROPOUT, p=0.5, seed=7)))
            res[w]["gplain"].append(timeit(lambda: ops.spmm_graph(g, X, out=Y, epilogue=GA)))
            res[w]["gepi"].append(timeit(lambda: ops.spmm_graph(g, X, out=Y, bias=b, epilogue=GA | EPI_BIAS | EPI_ELU | EPI_DROPOUT, p=0.5, seed=7)))
            res[w]["T"].append(timeit(lambda: ops.spmm_graph(gc, X, out=Y)))
        copy.append(timeit(lambda: Y.copy_(X)))
    print(json.dumps(info))
    print(f"rows {R} nnz' {batch.nnz} bytes/launch {bytes_spmm/1e6:.1f} MB; copy of {8*H*R/1e6:.1f} MB: "
          f"{np.median(copy):.1f} us = {8*H*R/np.median(copy)/1e3:.0f} GB/s")
    for w in windows:
        p, e, t, gp, ge = (np.median(res[w][k]) for k in ("plain", "epi", "T", "gplain", "gepi"))
        miss = float((graphs[w].f.lcol < 0).float().mean())
        print(f"window {w:3d} tiles {graphs[w].n_tiles:5d} miss {miss:.3f}: plain {p:7.1f} us ({bytes_spmm/p/1e3:6.0f} GB/s)  "
              f"epilogue {e:7.1f} us ({bytes_spmm/e/1e3:6.0f} GB/s)  contiguous-window {t:7.1f} us | gather plain {gp:7.1f} us "
              f"({bytes_spmm/gp/1e3:6.0f} GB/s) epilogue {ge:7.1f} us ({bytes_spmm/ge/1e3:6.0f} GB/s)")


if __name__ == "__main__":
    main()
