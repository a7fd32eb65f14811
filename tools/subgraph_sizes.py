#!/usr/bin/env python3
"""GPU-box tool: size distribution of a workload's cluster subgraphs (rows per subgraph of the union).  python tools/subgraph_sizes.py [S-products]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from fitgnn_amd import workloads

wl = sys.argv[1] if len(sys.argv) > 1 else "S-products"
dev = torch.device("cuda")
w0 = workloads.coarsen_workload(wl, dev)
sub, nnz_c = workloads.assemble(wl, torch.from_numpy(w0["ei"]).to(dev), torch.from_numpy(w0["assign"]).to(dev), w0["n_clusters"])
size = np.diff(sub["ptr"].cpu().numpy())
R = size.sum()
print(f"{wl}: {len(size)} subgraphs, {R} rows, mean {size.mean():.1f}, max {size.max()}")
for cap in (64, 128, 192, 256, 384, 512, 1024):
    big = size > cap
    print(f"  subgraphs > {cap:4d} rows: {big.sum():6d} ({big.mean() * 100:5.2f} %), holding {size[big].sum() / R * 100:5.2f} % of the rows, {np.asarray(nnz_c)[big].sum() / np.asarray(nnz_c).sum() * 100:5.2f} % of nnz'")
print("  quantiles (rows):", {q: int(np.quantile(size, q)) for q in (0.5, 0.9, 0.99, 0.999)})
