#!/usr/bin/env python3
"""GPU-box tool: where the single-wave greedy selection spends its cycles (needs lib/libfitgnn_dbg.so = coarsen.hip built
with -DFITGNN_GREEDY_STAMPS).  args: N E"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fit-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla, torch
from fitgnn_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "fit-gnn_amd", "lib", "libfitgnn_dbg.so")
from fitgnn_amd import coarsening, data

N, E = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (19717, 44324)
ei = data.synthetic_graph(N, E, seed=0)
W = sp.csr_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(N, N))
G = coarsening.Graph(W)
offset = 2 * max(G.dw)
T = offset * sp.eye(N, format="csc") - G.L
lk, Uk = spla.eigsh(T, k=10, which="LM", tol=1e-5, v0=np.random.default_rng(0).standard_normal(N))
lk, Uk = (offset - lk)[::-1], np.ascontiguousarray(Uk[:, ::-1])
A = coarsening._spectral_level1(G, 10, Uk.copy(), lk.copy())
L = _lib.lib()
L.fitgnn_debug_greedy_counters.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
coarsening.contract_level(G, A, 0.5); torch.cuda.synchronize()
L.fitgnn_debug_greedy_counters(buf, 1)
t0 = time.time(); res = coarsening.contract_level(G, A, 0.5); torch.cuda.synchronize(); dt = time.time() - t0
L.fitgnn_debug_greedy_counters(buf, 1)
v = list(buf)
names = ["pop", "mark check", "select+mark", "prune", "re-cost", "heap push"]
tot = sum(v[:6])
print(f"N={N} E={E}: contract_level {dt*1e3:.1f} ms, clusters {res.n}; pops {v[6]}, re-costs {v[7]}; stamped cycles {tot/1e6:.1f} M")
print("  cost fn phases (cycles per call incl. the initial cost kernel's calls):", {k: v[8 + i] for i, k in enumerate(["gather+mean", "W_S rows", "-", "norm", "total"])})
for n, c in zip(names, v[:6]):
    print(f"  {n:12s} {c/1e6:8.2f} Mcycles  {100*c/max(tot,1):5.1f} %  per event {c/max(v[7] if n in ('re-cost','heap push') else v[6],1):8.0f}")
