#!/usr/bin/env python3
"""GPU-box tool: row-length distribution inside the subgraphs APPNP's plan keeps in LDS.  python tools/appnp_row_lengths.py [S-products]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from fitgnn_amd import ops, workloads

wl = sys.argv[1] if len(sys.argv) > 1 else "S-products"
dev = torch.device("cuda")
w0 = workloads.coarsen_workload(wl, dev)
sub, nnz_c = workloads.assemble(wl, torch.from_numpy(w0["ei"]).to(dev), torch.from_numpy(w0["assign"]).to(dev), w0["n_clusters"])
batch = workloads.batch_from_subgraphs(wl, sub, dev)
g = batch.graph
plan = ops.appnp_plan(g, 12)
rp = g.f.rowptr.cpu().numpy().astype(np.int64)
ln = np.diff(rp)
for name, rng in [("units", plan.units)] + [("lds blocks at slice %d, %d threads" % (la[0], la[5]), la[1]) for la in plan.lds_launches]:
    if rng is None or not len(rng):
        continue
    rr = rng.cpu().numpy()
    rows = (rr[:, 1] - rr[:, 0])
    ent = rp[rr[:, 1]] - rp[rr[:, 0]]
    print(f"{wl} {name}: {len(rr)} ranges, rows mean {rows.mean():.1f} max {rows.max()}, entries mean {ent.mean():.1f} max {ent.max()}")
    mask = np.zeros(g.n, dtype=bool)
    cover = np.zeros(g.n + 1, dtype=np.int64)
    np.add.at(cover, rr[:, 0], 1); np.add.at(cover, rr[:, 1], -1)
    mask = np.cumsum(cover[:-1]) > 0
    l = ln[mask]
    prev = 0
    for cap in (4, 8, 16, 32, 64, 128, 256, 1 << 30):
        sel = (l > prev) & (l <= cap)
        print(f"   rows with {prev + 1:4d}..{cap if cap < 1 << 30 else 'inf':>4} entries: {sel.sum() / len(rr):8.1f} per range ({sel.mean() * 100:5.1f} % of rows, {l[sel].sum() / l.sum() * 100:5.1f} % of entries)")
        prev = cap
print("unit slice/threads", plan.unit_slice, plan.unit_threads, "open rows", plan.n_open)
