"""GPU tier: the HIP contraction step vs the CPU oracle, bit-exact, on the golden inputs; and the
drop-in coarsen() vs the assignment the REAL reference produced (tests/golden)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from golden_util import Golden, cases, graph_names

pytestmark = pytest.mark.gpu

CASES = cases()
_cache = {}


def G(name):
    if name not in _cache:
        _cache[name] = Golden(name)
    return _cache[name]


@pytest.fixture(scope="module")
def mods():
    assert torch.cuda.is_available()
    from fitgnn_amd import _lib, coarsening
    from oracle import coarsen_oracle as orc

    return _lib, coarsening, orc


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("name", graph_names())
def test_family_and_costs_bit_exact(mods, name):
    _lib, co, orc = mods
    g = G(name)
    for r in (0.5, 0.9):
        for li in range(g.n_levels(r)):
            L = g.level(r, li)
            Gr = co.Graph(L["W"])
            res = co.contract_level(Gr, L["A"], L["r_cur"], keep_debug=True)
            rowptr, col, w = orc._csr32(L["W"])
            dw = np.ascontiguousarray(Gr.dw)
            A = np.ascontiguousarray(np.real(L["A"]), dtype=np.float64)
            off, mem = orc.closed_neighbourhoods(rowptr, col, Gr.N)
            ref = orc.variation_costs(rowptr, col, w, dw, A, off[:-1].copy(), np.diff(off).astype(np.int32), mem)
            got = res.cost0.cpu().numpy()
            assert np.array_equal(_bits(got), _bits(ref)), f"{name} r={r} level {li}: costs differ in bits"
            # selection, assignment: identical to the oracle run on the same inputs
            n_reduce = int(np.floor(L["r_cur"] * Gr.N))
            so, sm, _ = orc.greedy_select(rowptr, col, w, dw, A, off, mem, ref, n_reduce)
            assert np.array_equal(res.sel_off, so) and np.array_equal(res.sel_mem, sm)
            assign, cval, n = orc.build_assignment(Gr.N, so, sm)
            assert res.n == n
            assert np.array_equal(res.assign.cpu().numpy(), assign)
            assert np.array_equal(_bits(res.cval.cpu().numpy()), _bits(cval))
            # lift
            rp, cc, wc = orc.lift_adjacency(rowptr, col, w, assign, cval, n)
            Wc = co.lift_adjacency(res)
            assert np.array_equal(Wc.indptr, rp) and np.array_equal(Wc.indices, cc)
            assert np.array_equal(_bits(Wc.data), _bits(wc))


@pytest.mark.parametrize("name,r", CASES)
def test_coarsen_matches_reference_assignment(mods, name, r):
    """Drop-in coarsen() == the real reference's C / assignment / Gc.W / C.X on the golden inputs."""
    _lib, co, orc = mods
    g = G(name)
    fin = g.final(r)
    C, Gc, maps = co.coarsen(co.Graph(g.W), K=g.K, r=r, method="variation_neighborhoods",
                             Uk=None if g.Uk is None else g.Uk.copy(), lk=None if g.lk is None else g.lk.copy())
    Cc = sp.csc_matrix(C)
    assert Cc.shape == fin["C"].shape
    assert np.array_equal(Cc.indices, fin["C"].indices), "integer assignment differs from the reference"
    assert np.array_equal(Cc.data, fin["C"].data)
    assert len(maps) == fin["n_mapping_dicts"]
    # composed mapping dicts == assignment (utils.py:113-121)
    comp = np.array([_compose(maps, i) for i in range(g.N)])
    assert np.array_equal(comp, fin["assign"])
    assert Gc.N == fin["GcW"].shape[0]
    if Gc.N != g.N:
        assert np.array_equal(Gc.W.indptr, fin["GcW"].indptr) and np.array_equal(Gc.W.indices, fin["GcW"].indices)
        assert np.array_equal(Gc.W.data, fin["GcW"].data)
    # pooling through the reference's own spelling C.dot(X)
    CX = C.dot(g.X)
    assert np.array_equal(CX, fin["CX64"])
    Xc = C.pool(torch.from_numpy(g.X).cuda()).cpu().numpy()
    assert np.array_equal(Xc, fin["CX64"].astype(np.float32))


def _compose(maps, i):
    m = maps[0][i]
    for d in maps[1:]:
        m = d[m]
    return m


def test_large_sets_take_the_tiled_path(mods):
    """A hub with > 64 neighbours exercises the multi-tile branch of the cost kernel (nc > 64)."""
    _lib, co, orc = mods
    rng = np.random.default_rng(5)
    n = 400
    rows, cols = [], []
    for leaf in range(1, 200):  # hub 0 with 199 neighbours
        rows += [0, leaf]; cols += [leaf, 0]
    for i in range(1, n - 1):
        rows += [i, i + 1]; cols += [i + 1, i]
    extra = rng.integers(1, n, size=(2, 300))
    extra = extra[:, extra[0] != extra[1]]
    rows += extra[0].tolist() + extra[1].tolist(); cols += extra[1].tolist() + extra[0].tolist()
    W = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n)); W.data[:] = 1.0
    w = rng.uniform(0.5, 2.0, size=W.nnz)
    Wt = sp.triu(sp.csr_matrix((w, W.indices, W.indptr), shape=W.shape), 1)
    W = (Wt + Wt.T).tocsr()
    Gr = co.Graph(W)
    lam, U = np.linalg.eigh(Gr.L.toarray())
    A = U[:, 1:11] / np.sqrt(lam[1:11])
    res = co.contract_level(Gr, A, 0.6, keep_debug=True)
    rowptr, col, ww = orc._csr32(W)
    off, mem = orc.closed_neighbourhoods(rowptr, col, n)
    ref = orc.variation_costs(rowptr, col, ww, np.ascontiguousarray(Gr.dw), np.ascontiguousarray(A), off[:-1].copy(),
                              np.diff(off).astype(np.int32), mem)
    assert np.array_equal(_bits(res.cost0.cpu().numpy()), _bits(ref))
    so, sm, _ = orc.greedy_select(rowptr, col, ww, np.ascontiguousarray(Gr.dw), np.ascontiguousarray(A), off, mem, ref,
                                  int(np.floor(0.6 * n)))
    assert np.array_equal(res.sel_off, so) and np.array_equal(res.sel_mem, sm)


@pytest.mark.parametrize("n,e,r,what", [
    (19717, 44324, 0.5, "S-pubmed shape: ~10 000 list pops, ~4 800 re-costs, most of them answered by the helper waves"),
    (60000, 600000, 0.5, "mean degree 20: the re-insertion queue outgrows its 1 024 LDS slots (blocks in global memory)"),
    (230000, 260000, 0.4, "more nodes than the LDS bitmap holds: marks and block minima in global memory, no helper waves"),
], ids=["pubmed-shape", "queue-spills-lds", "global-state"])
def test_selection_at_size_is_the_oracles(mods, n, e, r, what):
    """The greedy selection at sizes where every branch of the single-graph kernel runs (tournament queue over many
    blocks, LDS / global slots, speculative re-costs and their validation, the global-state fallback): selected sets
    equal the C oracle's, element for element.  A is random -- the selection logic does not care what the columns mean."""
    _lib, co, orc = mods
    from fitgnn_amd import data

    ei = data.synthetic_graph(n, e, seed=3)
    W = sp.csr_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(n, n))
    Gr = co.Graph(W)
    A = np.random.default_rng(n).standard_normal((n, 10))
    res = co.contract_level(Gr, A, r, keep_debug=True)
    rowptr, col, ww = orc._csr32(W)
    off, mem = orc.closed_neighbourhoods(rowptr, col, n)
    dw = np.ascontiguousarray(Gr.dw)
    ref = orc.variation_costs(rowptr, col, ww, dw, A, off[:-1].copy(), np.diff(off).astype(np.int32), mem)
    assert np.array_equal(_bits(res.cost0.cpu().numpy()), _bits(ref))
    so, sm, n_recost = orc.greedy_select(rowptr, col, ww, dw, A, off, mem, ref, int(np.floor(r * n)))
    assert n_recost > 1000, what
    assert np.array_equal(res.sel_off, so) and np.array_equal(res.sel_mem, sm), what
    # and again: the helper waves' timing differs from run to run, the result must not
    res2 = co.contract_level(Gr, A, r, keep_debug=True)
    assert np.array_equal(res2.sel_off, so) and np.array_equal(res2.sel_mem, sm)


def test_pool_ragged_feature_width_and_large(mods):
    _lib, co, orc = mods
    rng = np.random.default_rng(3)
    for N, n, F in ((1000, 400, 7), (5000, 1200, 500), (64, 64, 1), (300, 1, 33)):
        assign = np.sort(rng.integers(0, n, size=N)).astype(np.int32)
        assign[:n] = np.arange(n)  # every cluster non-empty
        assign = rng.permutation(assign).astype(np.int32)
        cval = (1.0 / np.sqrt(np.bincount(assign, minlength=n)))[assign]
        X = rng.random((N, F), dtype=np.float32)
        x64, x32 = orc.pool_rows(assign, cval, n, X)
        got32, got64 = co.pool_rows(torch.from_numpy(assign).cuda(), torch.from_numpy(cval).cuda(), n,
                                    torch.from_numpy(X).cuda(), want_f64=True)
        assert np.array_equal(got32.cpu().numpy(), x32)
        assert np.array_equal(_bits(got64.cpu().numpy()), _bits(x64))


@pytest.mark.parametrize("r", [0.3, 0.5, 0.7, 0.9])
def test_batched_coarsening_equals_the_reference_per_component(mods, r):
    """coarsen_batch on the block-diagonal union of ALL golden graphs (one wavefront per component, components
    leaving the level loop at different levels) == the real reference's per-graph C, Gc.W and C.X."""
    _lib, co, orc = mods
    names = [n for n in graph_names() if (n, r) in CASES]
    gs = [G(n) for n in names]
    W = sp.block_diag([g.W for g in gs], format="csr")
    comp_off = np.concatenate([[0], np.cumsum([g.N for g in gs])])
    out = co.coarsen_batch(W, comp_off, r=r, K=10, A0=[g.A0() for g in gs])
    Cb = sp.csc_matrix(out.C())
    X = np.concatenate([g.X for g in gs])
    Xc = out.pool(torch.from_numpy(X).cuda()).cpu().numpy()
    for c, (g, name) in enumerate(zip(gs, names)):
        fin = g.final(r)
        b, e = int(comp_off[c]), int(comp_off[c + 1])
        cb, ce = int(out.cluster_off[c]), int(out.cluster_off[c + 1])
        assert ce - cb == fin["C"].shape[0], (name, ce - cb, fin["C"].shape)
        assert np.array_equal(out.assign[b:e] - cb, fin["assign"]), name
        assert np.array_equal(out.cval[b:e], fin["C"].data), name
        Wc = out.Wc[cb:ce, cb:ce].tocsr()
        if fin["GcW"].shape[0] != g.N:
            assert np.array_equal(Wc.indptr, fin["GcW"].indptr) and np.array_equal(Wc.indices, fin["GcW"].indices), name
            assert np.array_equal(Wc.data, fin["GcW"].data), name
        assert np.array_equal(Xc[cb:ce], fin["CX64"].astype(np.float32)), name
    assert out.Wc[:, :].nnz == sum(out.Wc[int(out.cluster_off[c]):int(out.cluster_off[c + 1]),
                                          int(out.cluster_off[c]):int(out.cluster_off[c + 1])].nnz for c in range(len(gs)))
    assert Cb.shape == (out.n_clusters, W.shape[0])


def test_device_eigensolver_feeds_the_contraction(mods):
    """lanczos_smallest (thick-restart Lanczos on the device) == ARPACK's eigenvalues to the tolerance the
    reference asks for; coarsen(spectral='device') returns a valid partition of the requested size."""
    import scipy.sparse.linalg as spla

    _lib, co, orc = mods
    g = G("cora_giant")
    Gr = co.Graph(g.W)
    lk, Uk = co.lanczos_smallest(Gr.L, 10)
    offset = 2 * max(Gr.dw)
    ref = np.sort(offset - spla.eigsh((offset * sp.eye(g.N, format="csc") - Gr.L), k=10, which="LM", tol=1e-5,
                                      v0=np.ones(g.N))[0])
    assert np.abs(lk - ref).max() < 1e-4 * offset
    R = Gr.L @ Uk - Uk * lk
    assert np.abs(R).max() < 1e-3 * offset and np.abs(Uk.T @ Uk - np.eye(10)).max() < 1e-8
    C, Gc, maps = co.coarsen(Gr, r=0.5, method="variation_neighborhoods", spectral="device")
    Cc = sp.csc_matrix(C)
    assert np.all(np.diff(Cc.indptr) == 1) and Cc.shape[0] == int(np.ceil(0.5 * g.N))
    assert np.allclose(np.asarray(Cc.power(2).sum(1)).ravel(), 1.0)
    # more than 16 eigenpairs: the rotation kernel takes 16 columns per launch, the solver rotates in groups (K > 16
    # used to fail in the final rotation, after the whole solve)
    lk20, Uk20 = co.lanczos_smallest(Gr.L, 20)
    ref20 = np.sort(np.linalg.eigvalsh(Gr.L.toarray()))[:20]
    assert lk20.shape == (20,) and Uk20.shape == (g.N, 20)
    assert np.abs(lk20 - ref20).max() < 1e-4 * offset
    assert np.abs(Gr.L @ Uk20 - Uk20 * lk20).max() < 1e-3 * offset and np.abs(Uk20.T @ Uk20 - np.eye(20)).max() < 1e-8


def test_pipeline_batched_components_equal_per_component_loop(mods):
    """pipeline.coarsening_classification on a dataset with many components (a block-diagonal union of golden graphs plus
    isolated nodes): the batched contraction gives the same clusters, C and Gc per component as the per-component loop."""
    import argparse

    from fitgnn_amd import pipeline

    gs = [G(n) for n in ("ring100", "cora26", "cora9", "star40", "cora2", "ring400", "ba600w")]
    W = sp.block_diag([g.W for g in gs] + [sp.csr_matrix((3, 3))], format="csr")   # + three isolated nodes
    W.data[:] = 1.0
    coo = W.tocoo()
    N = W.shape[0]
    data = pipeline.NodeData(torch.zeros(N, 4), np.stack([coo.row, coo.col]), torch.zeros(N, dtype=torch.long))
    args = argparse.Namespace()
    _lib, co, orc = mods

    def dense_prelude(Gr, K, Uk, lk):   # ARPACK starts from a random vector: give both paths the same exact eigenvectors
        return co._dense_prelude(Gr.W, 0, Gr.N, K)

    saved, co._spectral_level1 = co._spectral_level1, dense_prelude
    try:
        a = pipeline.coarsening_classification(args, data, 0.5, "variation_neighborhoods", batched=True)
        b = pipeline.coarsening_classification(args, data, 0.5, "variation_neighborhoods", batched=False)
    finally:
        co._spectral_level1 = saved
    assert a.n_clusters == b.n_clusters and np.array_equal(a.assign, b.assign)
    assert len(a.all_C) == len(b.all_C) == len(a.components)
    for Ca, Cb, Ga, Gb in zip(a.all_C, b.all_C, a.all_Gc, b.all_Gc):
        if Ca is None:
            assert Cb is None
            continue
        assert (sp.csc_matrix(Ca) != sp.csc_matrix(Cb)).nnz == 0 and (Ga.W != Gb.W).nnz == 0
