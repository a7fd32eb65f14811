#!/usr/bin/env python3
"""GPU-box probe: how many of the SpMM's window misses would one (or two) pinned extra rows per tile capture?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch, collections
import bench
from fitgnn_amd.csr import CSRGraph
dev = torch.device("cuda")
batch, _, info = bench.build_workload("S-pubmed", 0, dev)
g = CSRGraph(batch.edge_index, batch.n_rows, mode="gcn", ptr=batch.ptr)
for name, side in (("forward", g.f), ("transposed", g.t)):
    t = side.tiles.cpu().numpy(); col = side.col.cpu().numpy()
    tot = cap1 = cap2 = 0
    for i in np.nonzero(t[:, 1] > t[:, 0])[0]:
        c = col[t[i, 4]:t[i, 5]]
        m = c[(c < t[i, 2]) | (c >= t[i, 2] + min(t[i, 3], 16))]
        if len(m):
            cnt = collections.Counter(m.tolist()).most_common(2)
            tot += len(m); cap1 += cnt[0][1]; cap2 += cnt[0][1] + (cnt[1][1] if len(cnt) > 1 else 0)
    print(f"{name}: misses {tot} of {len(col)} nnz; one pinned row captures {cap1} ({100*cap1/max(tot,1):.0f} %), two {cap2} ({100*cap2/max(tot,1):.0f} %)")
