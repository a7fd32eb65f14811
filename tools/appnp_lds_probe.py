#!/usr/bin/env python3
"""GPU-box tool: where the column-sliced APPNP kernel's time goes -- each launch of the plan timed at K = 0 (staging + one load / store
pass per slice), 1, 10, 20 (slope = one step), forward and backward.  python tools/appnp_lds_probe.py [S-products] [threads]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from fitgnn_amd import _lib, ops, workloads

wl = sys.argv[1] if len(sys.argv) > 1 else "S-products"
dev = torch.device("cuda")
w0 = workloads.coarsen_workload(wl, dev)
sub, nnz_c = workloads.assemble(wl, torch.from_numpy(w0["ei"]).to(dev), torch.from_numpy(w0["assign"]).to(dev), w0["n_clusters"])
batch = workloads.batch_from_subgraphs(wl, sub, dev)
g = batch.graph
h4 = 12
plan = ops.appnp_plan(g, h4)
L = _lib.lib()
x = torch.randn(g.n, 4 * h4, device=dev)
y = torch.empty_like(x)
st = _lib.stream_ptr(dev)
dp = _lib.dptr
launches = [("units", plan.unit_slice, plan.units, plan.n_units, plan.max_rows, plan.max_entries, plan.unit_threads)]
launches += [("blocks", *grp) for grp in plan.lds_launches]
dbg = os.environ.get("FITGNN_APPNP_LDS_DEBUG", "0")
only = sys.argv[2] if len(sys.argv) > 2 else ""
for name, sl, ranges, m, mr, me, threads in launches:
    if only and only != f"{name}{sl}x{threads}":
        continue
    rr = ranges.cpu().numpy()
    print(f"[dbg {dbg}] {name} slice {sl} threads {threads}: {m} ranges, rows mean {(rr[:, 1] - rr[:, 0]).mean():.0f} max {mr}, max entries {me}, "
          f"LDS {L.fitgnn_appnp_lds_bytes(mr, me, sl)} B")
    for bwd, side in ((0, g.f), (1, g.t)):
        out = []
        for K in (0, 1, 10, 20):
            ts = []
            for rep in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(L.fitgnn_appnp_lds_f32(dp(side.rowptr), dp(side.col), dp(side.val), dp(ranges), m, mr, me, dp(x), dp(y), h4, K, 0.1, bwd,
                                                  threads, sl, st), "lds")
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            out.append(min(ts[1:]))
        print(f"   {'bwd' if bwd else 'fwd'}: K=0 {out[0]:8.1f} us   K=1 {out[1]:8.1f}   K=10 {out[2]:8.1f}   K=20 {out[3]:8.1f}   per step {(out[3] - out[2]) / 10:7.1f} us")
