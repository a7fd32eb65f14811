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
if not hasattr(L, "fitgnn_debug_greedy_counters"):   # a library without the counters: the plain time only
    coarsening.contract_level(G, A, r); torch.cuda.synchronize()
    for _ in range(3):
        t0 = time.time()
        coarsening.contract_level(G, A, r); torch.cuda.synchronize()
        print(f"{name}: contract_level {(time.time() - t0)*1e3:.1f} ms (no counters in this build)")
    sys.exit(0)
L.fitgnn_debug_greedy_counters.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 24)()
coarsening.contract_level(G, A, r); torch.cuda.synchronize()
L.fitgnn_debug_greedy_counters(buf, 1)
L.fitgnn_debug_cost_counters.argtypes = [ctypes.c_void_p, ctypes.c_int]
cb = (ctypes.c_ulonglong * 12)()
L.fitgnn_debug_cost_counters(cb, 1)
t0 = time.time()
coarsening.contract_level(G, A, r); torch.cuda.synchronize()
dt = time.time() - t0
L.fitgnn_debug_greedy_counters(buf, 1)
v = list(buf)
tot = sum(v[0:6])
print(f"{name}: contract_level {dt*1e3:.1f} ms; stamped cycles {tot}")
names = ["pop from list", "pop from heap", "mark check", "select", "prune", "re-cost + push"]
cnts = [v[8], v[9], v[8] + v[9], v[10], v[11], v[12]]
for n, cyc, c in zip(names, v[0:6], cnts):
    print(f"  {n:14s} {100.0*cyc/max(tot,1):5.1f} %  n={c:7d}  cycles/op={cyc/max(c,1):8.1f}")
print(f"  max queue size {v[13]}, re-costs of list entries {v[14]} of {v[12]}, answered by a helper wave {v[15]}; queue re-costs answered by the helper {v[6]}; from a stored match list {v[7]}")
print(f"  a helper's list of a superset filtered for {v[16]} list entries; adjacency lists scanned by the selecting wave: list entries {v[18]}, queue sets {v[19]}")
L.fitgnn_debug_cost_counters(cb, 1)
c = list(cb)
ct = sum(c[0:9])
print(f"  re-costs on the selecting wave: {ct} cycles; members {c[10]}, matches {c[11]}, stored matches read {c[9]}")
for n, cyc in zip(["gather A rows", "column means", "centre", "stored list arrives", "filter stored list / scan adjacency", "export + fold rows (T, Y)", "M accumulate", "norm", "filter: first chunk"], c[0:9]):
    print(f"    {n:28s} {100.0*cyc/max(ct,1):5.1f} %")
