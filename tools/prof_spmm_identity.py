#!/usr/bin/env python3
"""GPU-box tool: calibrate FETCH_SIZE on the SpMM kernels themselves -- a self-loops-only pattern over R rows in blocks of
100 rows (tile kernel: tiles; block kernel: whole blocks) reads X exactly once: 4 * 512 * R bytes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from fitgnn_amd import ops
from fitgnn_amd.csr import CSRGraph
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
dev = torch.device("cuda")
ptr = np.arange(0, R + 1, 100)
g = CSRGraph(torch.zeros((2, 0), dtype=torch.long, device=dev), R, mode="gcn", ptr=ptr, block_limit=4096)
X = torch.randn(R, 512, device=dev); Y = torch.empty_like(X)
for cfg in (ops.OpConfig(split_large_blocks=True), ops.OpConfig(split_large_blocks=False)):
    for _ in range(4):
        ops.spmm_graph(g, X, out=Y, cfg=cfg)
torch.cuda.synchronize()
print("X bytes", 4 * 512 * R, "KiB", 4 * 512 * R / 1024)
