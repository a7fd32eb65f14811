#!/usr/bin/env python3
"""GPU-box tool for rocprofv3 PMC passes: a few launches of the folded backward and of the plain transposed SpMM on
the S-pubmed union (same graph, same operand size)."""
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
out = torch.randn(R, H, device=dev); dOut = torch.randn(R, H, device=dev)
ops.FOLD_BACKWARD = True
for _ in range(4):
    ops.layer_backward(g, out, 0, 0.5, 7, None, False, dOut=dOut)
    ops.spmm_graph(g, dOut, transposed=True)
torch.cuda.synchronize()
