"""Where lanczos_smallest spends its time at a workload's size (round 3)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fit-gnn_amd"))
import numpy as np, scipy.sparse as sp, torch
from fitgnn_amd import coarsening, data, workloads

name = sys.argv[1] if len(sys.argv) > 1 else "S-products"
N, E, F, C, r = workloads.SHAPES[name]
ei = data.synthetic_graph(N, E, seed=0)
W = sp.csr_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(N, N))
G = coarsening.Graph(W)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    lk, Uk = coarsening.lanczos_smallest(G.L, 10, device="cuda", seed=0)
    torch.cuda.synchronize(); print(name, "lanczos_smallest", round(time.time() - t0, 3), "s", lk[:3])
R = G.L @ Uk - Uk * lk
print("residual", np.abs(R).max(), "orth", np.abs(Uk.T @ Uk - np.eye(10)).max())
