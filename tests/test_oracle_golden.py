"""Pin the CPU oracle (oracle/) against golden vectors produced by the REAL reference
(tests/golden/make_golden.py imports /root/reference/graph_coarsening unmodified).

Float costs: <= 1e-9 relative (the reference sums through BLAS in an unspecified order).
Integer logic (greedy selection, C pattern, assignment, lifted adjacency): exact, by replaying
the reference's own recorded cost stream through the oracle's selection loop.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from golden_util import Golden, cases, graph_names
from oracle import coarsen_oracle as orc

CASES = cases()
_cache = {}


def cost_atol(A, dw):
    """Absolute floor: the cost is a norm of B^T L B whose terms are O(|A|^2 dmax); sets with (almost)
    identical rows of A (twin leaves) have cost ~1e-28 that is pure cancellation residue."""
    return 1e-13 * float(np.abs(A).max(initial=0.0)) ** 2 * float(np.max(dw, initial=1.0))


def cost_rtol(A):
    """1e-9 relative, except on numerically singular levels: when the K x K eigenproblem of
    coarsening_utils.py:100 has rank < K (graphs with fewer than K nodes at level >= 2), its ~1e-16
    eigenvalues are inverted (:103) and A reaches 1e6..1e8; the centring B = A - mean then cancels
    8+ digits and the reference's own cost is rounding noise.  Such levels are compared at 20 %."""
    return 1e-9 if np.abs(A).max(initial=0.0) < 1e5 else 0.2


def G(name):
    if name not in _cache:
        _cache[name] = Golden(name)
    return _cache[name]


def _level_inputs(L):
    rowptr, col, w = orc._csr32(L["W"])
    dw = np.ascontiguousarray(L["dw"], dtype=np.float64)
    A = np.ascontiguousarray(L["A"], dtype=np.float64)
    N = L["W"].shape[0]
    off, mem = orc.closed_neighbourhoods(rowptr, col, N)
    return rowptr, col, w, dw, A, N, off, mem


@pytest.mark.parametrize("name,r", CASES)
def test_initial_costs_match_reference(name, r):
    g = G(name)
    for li in range(g.n_levels(r)):
        L = g.level(r, li)
        rowptr, col, w, dw, A, N, off, mem = _level_inputs(L)
        cost = orc.variation_costs(rowptr, col, w, dw, A, off[:-1].copy(), np.diff(off).astype(np.int32), mem)
        ref = L["cost0"]
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(cost), fin)
        assert np.all(np.abs(cost[fin] - ref[fin]) <= cost_rtol(A) * np.abs(ref[fin]) + cost_atol(A, dw))


@pytest.mark.parametrize("name,r", CASES)
def test_recost_trace_matches_reference(name, r):
    """Every re-cost the reference performed (coarsening_utils.py:646), recomputed on its recorded set."""
    g = G(name)
    for li in range(g.n_levels(r)):
        L = g.level(r, li)
        if len(L["trace_cost"]) == 0:
            continue
        rowptr, col, w, dw, A, N, _, _ = _level_inputs(L)
        toff = L["trace_off"].astype(np.int32)
        cost = orc.variation_costs(rowptr, col, w, dw, A, toff[:-1].copy(), np.diff(toff).astype(np.int32),
                                   L["trace_mem"].astype(np.int32))
        ref = L["trace_cost"]
        assert np.all(np.abs(cost - ref) <= cost_rtol(A) * np.abs(ref) + cost_atol(A, dw))


@pytest.mark.parametrize("name,r", CASES)
def test_selection_logic_exact_on_reference_cost_stream(name, r):
    """Replay the reference's recorded costs -> the selected sets, iC and survivors must be identical."""
    g = G(name)
    for li in range(g.n_levels(r)):
        L = g.level(r, li)
        rowptr, col, w, dw, A, N, off, mem = _level_inputs(L)
        n_reduce = int(np.floor(L["r_cur"] * N))
        sel_off, sel_mem, used = orc.greedy_select(rowptr, col, w, dw, A, off, mem, L["cost0"], n_reduce,
                                                   recost_stream=L["trace_cost"])
        assert used == len(L["trace_cost"])
        assert np.array_equal(sel_off, L["sel_off"])
        assert np.array_equal(sel_mem, L["sel_mem"])
        assign, cval, n = orc.build_assignment(N, sel_off, sel_mem)
        iC = orc.iC_from(assign, cval, n)
        ref = L["iC"]
        assert iC.shape == ref.shape
        assert np.array_equal(iC.indptr, ref.indptr) and np.array_equal(iC.indices, ref.indices)
        assert np.array_equal(iC.data, ref.data)  # 1/sqrt(nc): bit-exact


@pytest.mark.parametrize("name,r", CASES)
def test_lift_and_pool_exact_given_reference_assignment(name, r):
    g = G(name)
    fin = g.final(r)
    nl = g.n_levels(r)
    # last recorded level whose iC was actually applied (coarsening_utils.py:131-135 break rule)
    for li in range(nl):
        L = g.level(r, li)
        ref_iC = L["iC"]
        if ref_iC.shape[1] - ref_iC.shape[0] <= 2:
            continue
        rowptr, col, w = orc._csr32(L["W"])
        assign = np.asarray(ref_iC.indices, dtype=np.int32)  # one nnz per column: row index = cluster
        n = ref_iC.shape[0]
        cval = np.asarray(ref_iC.data, dtype=np.float64)
        rp, cc, wc = orc.lift_adjacency(rowptr, col, w, assign, cval, n)
        Wc = sp.csr_matrix((wc, cc, rp), shape=(n, n))
        if li + 1 < nl:
            ref_W = g.level(r, li + 1)["W"]
        else:
            ref_W = fin["GcW"]
        assert ref_W.shape == Wc.shape
        # SciPy's sparse products sum in a fixed order which the oracle restates: bit-exact, weighted too
        assert np.array_equal(Wc.indptr, ref_W.indptr) and np.array_equal(Wc.indices, ref_W.indices)
        assert np.array_equal(Wc.data, ref_W.data)
    C = fin["C"]
    assign = np.asarray(C.indices, dtype=np.int32)
    x64, x32 = orc.pool_rows(assign, np.asarray(C.data), C.shape[0], g.X)
    assert np.array_equal(x64, fin["CX64"])  # f64 accumulate in ascending member order: bit-exact
    assert np.array_equal(x32, fin["CX64"].astype(np.float32))
    assert np.array_equal(assign, fin["assign"])  # composed mapping dicts == row pattern of C


@pytest.mark.parametrize("name,r", CASES)
def test_end_to_end_driver(name, r):
    """Full multilevel driver on canonical costs.  Equality with the reference's partition is exact
    wherever the reference's own decision is not a floating-point near-tie; near-ties are reported
    (xfail-free): we require identical cluster COUNT always, identical assignment when no pair of
    competing costs is closer than 1e-10 relative."""
    g = G(name)
    fin = g.final(r)
    out = orc.coarsen_oracle(g.W, K=g.K, r=r, Uk=g.Uk, lk=g.lk)
    assert out["C"].shape == fin["C"].shape
    # invariants (SURVEY.md §8c known-answer properties)
    C = out["C"]
    assert np.all(np.diff(C.indptr) == 1)
    assert np.allclose(np.asarray(C.power(2).sum(axis=1)).ravel(), 1.0)
    same = np.array_equal(out["assign"], fin["assign"])
    if not same:
        # tolerated only if the reference's level costs contain a near-tie
        tie = False
        for li in range(g.n_levels(r)):
            c = np.sort(np.concatenate([g.level(r, li)["cost0"], g.level(r, li)["trace_cost"]]))
            c = c[np.isfinite(c)]
            gaps = np.diff(c) / np.maximum(c[1:], 1e-300)
            tie |= bool(np.any(gaps < 1e-10)) or cost_rtol(g.level(r, li)["A"]) > 1e-9
        assert tie, "partition differs from the reference without any near-tied cost"
    else:
        assert np.array_equal(out["C"].indices, fin["C"].indices)
        assert np.allclose(out["C"].data, fin["C"].data, rtol=0, atol=0)


def test_python_twin_equals_c_on_small_sets():
    g = G("cora26")
    L = g.level(0.5, 0)
    rowptr, col, w, dw, A, N, off, mem = _level_inputs(L)
    cost = orc.variation_costs(rowptr, col, w, dw, A, off[:-1].copy(), np.diff(off).astype(np.int32), mem)
    for i in range(N):
        S = mem[off[i]:off[i + 1]]
        assert orc.set_cost_py(rowptr, col, w, dw, A, S) == cost[i]
    g = G("ba600w")
    L = g.level(0.5, 0)
    rowptr, col, w, dw, A, N, off, mem = _level_inputs(L)
    cost = orc.variation_costs(rowptr, col, w, dw, A, off[:-1].copy(), np.diff(off).astype(np.int32), mem)
    for i in range(0, N, 7):
        S = mem[off[i]:off[i + 1]]
        assert orc.set_cost_py(rowptr, col, w, dw, A, S) == cost[i]


@pytest.mark.parametrize("name", graph_names())
def test_closed_neighbourhoods_match_scipy(name):
    g = G(name)
    rowptr, col, w = orc._csr32(g.W)
    off, mem = orc.closed_neighbourhoods(rowptr, col, g.N)
    Wb = ((g.W > 0) + sp.eye(g.N, dtype=bool, format="csr")).tocsr()
    Wb.sort_indices()
    assert np.array_equal(off, Wb.indptr) and np.array_equal(mem, Wb.indices)
