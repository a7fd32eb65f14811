#!/usr/bin/env python3
"""GPU-box tool: where the single wave of the greedy selection spends its cycles (S-pubmed or S-products graph).
Needs the library built with the counters:  touch fit-gnn_amd/csrc/coarsen.hip && make -C fit-gnn_amd/csrc EXTRA=-DFITGNN_GREEDY_STAMPS
(rebuild without EXTRA afterwards: the counters cost time)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, scipy.sparse as sp, torch
from fitgnn_amd import _lib, coarsening, data, workloads

name = sys.argv[1] if len(sys.argv) > 1 else "S-pubmed"
N, E, F, C, r = workloads.SHAPES[name]
ei = data.synthetic_graph(N, E, seed=0)
W = sp.csr_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(N, N))
G = coarsening.Graph(W)
lk, Uk = coarsening.lanczos_smallest(G.L, 10, seed=0)
A = coarsening._spectral_level1(G, 10, Uk.copy(), lk.copy())
L = _lib.lib()
L.fitgnn_debug_greedy_counters.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
coarsening.contract_level(G, A, r); torch.cuda.synchronize()
L.fitgnn_debug_greedy_counters(buf, 1)
t0 = time.time()
coarsening.contract_level(G, A, r); torch.cuda.synchronize()
dt = time.time() - t0
L.fitgnn_debug_greedy_counters(buf, 1)
v = list(buf)
tot = sum(v[0:7])
print(f"{name}: contract_level {dt*1e3:.1f} ms; stamped cycles {tot}")
names = ["pop from list", "pop from heap", "mark check", "select", "prune", "re-cost", "heap push"]
cnts = [v[8], v[9], v[8] + v[9], v[10], v[11], v[12], v[12]]
for n, cyc, c in zip(names, v[0:7], cnts):
    print(f"  {n:14s} {100.0*cyc/max(tot,1):5.1f} %  n={c:7d}  cycles/op={cyc/max(c,1):8.1f}")
print(f"  max queue size {v[13]}, re-costs of list entries {v[14]} of {v[12]}, answered by a helper wave {v[15]}, from a stored match list {v[7]}")
