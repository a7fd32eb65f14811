#!/usr/bin/env python3
"""GPU-box tool for rocprofv3 runs: a fixed, short sequence of SpMM launches on the S-pubmed union
(args: variant=gather|lds, window rows, launches, epilogue 0/1)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import torch

import bench
from fitgnn_amd import ops
from fitgnn_amd._lib import EPI_BIAS, EPI_DROPOUT, EPI_ELU, SPMM_GATHER
from fitgnn_amd.csr import CSRGraph

variant, window, n, epi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda")
batch, _, info = bench.build_workload("S-pubmed", 0, dev)
R = batch.n_rows
g = CSRGraph(batch.edge_index, R, mode="gcn", ptr=batch.ptr, lds_rows=window)
X = torch.randn(R, 512, device=dev)
Y = torch.empty_like(X)
b = torch.randn(512, device=dev)
flags = (SPMM_GATHER if variant == "gather" else 0) | ((EPI_BIAS | EPI_ELU | EPI_DROPOUT) if epi else 0)
for _ in range(n):
    ops.spmm_raw(g.rowptr, g.col, g.val, g.tiles, X, R, out=Y, window_rows=window, bias=b if epi else None, epilogue=flags,
                 p=0.5 if epi else 0.0, seed=3)
torch.cuda.synchronize()
print("rows", R, "nnz", batch.nnz, "bytes", 8 * 512 * R + 8 * batch.nnz + 4 * (R + 1))
