#!/usr/bin/env python3
"""GPU-box tool for rocprofv3 --pmc passes: a few launches of the SpMM on a workload's union, whole-subgraph kernel and tiled
(args: workload, launches)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import torch

from fitgnn_amd import ops, workloads
from fitgnn_amd.csr import CSRGraph

wl, n = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda")
w0 = workloads.coarsen_workload(wl, dev)
sub, nnz_c = workloads.assemble(wl, torch.from_numpy(w0["ei"]).to(dev), torch.from_numpy(w0["assign"]).to(dev), w0["n_clusters"])
ptr = sub["ptr"].cpu().numpy()
R = int(ptr[-1])
if "seg_start" in sub:
    import numpy as np
    ptr = np.concatenate([np.nonzero(sub["seg_start"].cpu().numpy())[0], [R]]).astype(np.int64)
    sizes = np.diff(ptr)
    print("segments", len(sizes), "rows in segments <= 16:", int(sizes[sizes <= 16].sum()), "max", int(sizes.max()),
          "hist", np.histogram(sizes, bins=[0, 8, 16, 32, 48, 64, 96, 128, 256, 1024, 1 << 20])[0].tolist())
g = CSRGraph(sub["edge_index"], R, mode="gcn", ptr=ptr)
X = torch.randn(R, 512, device=dev)
Y = torch.empty_like(X)
for cfg in (ops.OpConfig(split_large_blocks=True), ops.OpConfig(split_large_blocks=False)):
    for _ in range(n):
        ops.spmm_graph(g, X, out=Y, cfg=cfg)
torch.cuda.synchronize()
print("rows", R, "nnz", int(nnz_c.sum()))
