"""Seeded synthetic stand-ins of BASELINE.json's configurations (no dataset can be downloaded on the GPU
box), shared by bench.py and the full-size GPU tests so that what is tested is what is timed.

Shapes (dataset_info.csv:4-7, main.py:243,264): preferential-attachment graph with the dataset's (N, E) (seed 0),
features U[0,1) row-L1-normalised (--normalize_features, main.py:48; seed 1), uniform integer labels, every node a train
node.  The graph is coarsened ONCE by the HIP contraction step (variation_neighborhoods, Loukas r), every cluster gets its
1-hop "extra node" subgraph (utils.py:235-239), and all subgraphs form one block-diagonal union -- whole, or this
rank's shard of it (data.shard_clusters: LPT over nnz').
"""
import time

import numpy as np
import scipy.sparse as sp
import torch

from . import coarsening, data

SHAPES = {
    # name: (N, E, F, classes, Loukas r)
    "S-products": (165000, 4125000, 100, 47, 0.5),   # one ogbn-products community (<= 165 000 nodes, main.py:264), mean degree 50 assumed
    "S-pubmed": (19717, 44324, 500, 3, 0.5),
    "S-cora": (2708, 5278, 1433, 7, 0.5),
    "S-physics": (34493, 247962, 8415, 5, 0.7),      # CLI --coarsening_ratio 0.3 (main.py:278 passes 1 - ratio)
}


def coarsen_workload(name, device, spectral="device"):
    """Graph + partition of workload `name` (the part rank 0 does under data parallelism).
    Returns dict(ei int64 [2, 2E] numpy, W csr, Uk, lk, r, assign int64 [N], n_clusters, timings)."""
    N, E, F, C, r = SHAPES[name]
    t0 = time.time()
    ei = data.synthetic_graph(N, E, seed=0)
    W = sp.csr_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(N, N))
    G = coarsening.Graph(W)
    t1 = time.time()
    # spectral input of the contraction step (coarsening_utils.py:83-90): A = f(Uk, lk) is an INPUT of the accelerated
    # step; computed on the device by default (f4), by ARPACK with a fixed start vector otherwise
    if spectral == "device":
        lk, Uk = coarsening.lanczos_smallest(G.L, 10, device=device, seed=0)
    else:
        import scipy.sparse.linalg as spla
        offset = 2 * max(G.dw)
        T = offset * sp.eye(N, format="csc") - G.L
        lk, Uk = spla.eigsh(T, k=10, which="LM", tol=1e-5, v0=np.random.default_rng(0).standard_normal(N))
        lk, Uk = (offset - lk)[::-1], Uk[:, ::-1]
    Uk, lk = np.ascontiguousarray(Uk), np.ascontiguousarray(lk)
    torch.cuda.synchronize()
    t2 = time.time()
    Cmat, Gc, _ = coarsening.coarsen(G, r=r, method="variation_neighborhoods", Uk=Uk.copy(), lk=lk.copy(), device=device)
    torch.cuda.synchronize()
    t3 = time.time()
    return dict(ei=ei, W=W, Uk=Uk, lk=lk, r=r, assign=sp.csc_matrix(Cmat).indices.astype(np.int64), n_clusters=int(Cmat.shape[0]),
                C=Cmat, timings=dict(t_graph_s=round(t1 - t0, 2), t_spectral_s=round(t2 - t1, 2), spectral=spectral,
                                     t_coarsen_hip_s=round(t3 - t2, 3)))


def features_and_labels(name):
    N, E, F, C, r = SHAPES[name]
    rng = np.random.default_rng(1)
    X = rng.random((N, F), dtype=np.float32)
    X /= X.sum(1, keepdims=True)  # --normalize_features (main.py:48)
    y = rng.integers(0, C, size=N)
    return X, y


def assemble(name, ei_d, assign_d, n_clusters, layout="star", clusters=None):
    """All cluster subgraphs of the partition (device tensors) and their nnz'.  layout: the row order inside a subgraph
    (data.assemble_subgraphs_torch): star by star by default, which is what the train path uses.  clusters: only these
    clusters' subgraphs (a data-parallel rank's shard: see shard_before_assembly)."""
    N = SHAPES[name][0]
    sub = data.assemble_subgraphs_torch(ei_d, N, assign_d, n_clusters, extra_node=True, layout=layout, clusters=clusters)
    return sub, data.cluster_nnz(sub)


def shard_before_assembly(name, ei_d, assign_d, n_clusters, world, return_weights=False):
    """owner[c] = rank of cluster c (whole subgraphs, LPT), decided BEFORE any subgraph exists from
    data.cluster_weights_torch -- computed by every rank from the graph and the partition it already holds, so every rank gets
    the same answer and then assembles its own clusters only."""
    N = SHAPES[name][0]
    w = data.cluster_weights_torch(ei_d, N, assign_d, n_clusters, extra_node=True)
    owner = data.shard_clusters(None, w, world)
    return (owner, w) if return_weights else owner


def batch_from_subgraphs(name, sub, device, X=None, y=None):
    """SubgraphBatch (device resident, every cluster node a train node) of `sub` = the union or a select_clusters() part."""
    N = SHAPES[name][0]
    if X is None:
        X, y = features_and_labels(name)
    train_mask = np.ones(N, dtype=bool)  # every cluster node labelled: every subgraph takes part in the GD step
    return data.SubgraphBatch(sub, X, y, train_mask, device=device)
