#!/usr/bin/env python3
"""GPU-box tool: the greedy selection (helper waves racing the selecting wave) 40 times per graph against the C oracle: the result
must not depend on the helpers' timing (120 runs, 0 mismatches: profiles/r03_stress_greedy.log)."""
import os, sys
ROOT = "/root/repo"
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, scipy.sparse as sp, torch
from fitgnn_amd import coarsening as co, data
from oracle import coarsen_oracle as orc
bad = 0
for (n, e, r) in [(19717, 44324, 0.5), (60000, 600000, 0.5), (5000, 30000, 0.7)]:
    ei = data.synthetic_graph(n, e, seed=11)
    W = sp.csr_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(n, n))
    Gr = co.Graph(W)
    A = np.random.default_rng(n + 1).standard_normal((n, 10))
    rowptr, col, ww = orc._csr32(W)
    off, mem = orc.closed_neighbourhoods(rowptr, col, n)
    dw = np.ascontiguousarray(Gr.dw)
    ref = orc.variation_costs(rowptr, col, ww, dw, A, off[:-1].copy(), np.diff(off).astype(np.int32), mem)
    so, sm, _ = orc.greedy_select(rowptr, col, ww, dw, A, off, mem, ref, int(np.floor(r * n)))
    for rep in range(int(os.environ.get("STRESS_REPS", "40"))):
        res = co.contract_level(Gr, A, r, keep_debug=True)
        if not (np.array_equal(res.sel_off, so) and np.array_equal(res.sel_mem, sm)):
            bad += 1
            print("MISMATCH", n, rep)
    print("graph", n, "done")
print("mismatches:", bad)
