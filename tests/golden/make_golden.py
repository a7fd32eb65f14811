#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the *real* FIT-GNN reference.

Runs ONLY in the build container (needs /root/reference).  Nothing here travels into the
product; the GPU box never runs this file.  What it does:

  * puts a minimal stand-in for the third-party `pygsp.graphs.Graph` class into
    `sys.modules` (pygsp is a missing *dependency* of the reference, it is not vendored in
    /root/reference; semantics restated from pygsp 0.5.1's public behaviour, see SURVEY.md §8c:
    W -> csr f64 without explicit zeros, A = W > 0, dw = column sums, L = diag(dw) - W as csc);
  * imports the reference's `graph_coarsening.coarsening_utils` UNMODIFIED from /root/reference;
  * runs `coarsen(G, r=..., method="variation_neighborhoods", Uk=Uk, lk=lk)` on a set of small
    graphs with the spectral pair (Uk, lk) captured from one `eigsh` run and passed back in
    (the reference is not repeatable otherwise: ARPACK start vector is random);
  * records, per level, everything the contraction step consumes and produces by wrapping the
    module-level names `contract_variation_linear`, `get_coarsening_matrix` and `SortedList`
    (no reference source is edited or copied): spectral matrix A, W, dw, the initial cost of
    every candidate set, the ordered trace of every re-costed set, the selected contraction
    sets, iC; and the final C, assignment vector, Gc.W and C.X.

Output: one compressed .npz per (graph, r) + manifest.json.  Re-run: `python tests/golden/make_golden.py`.
"""
import json
import os
import pickle
import sys
import types

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True  # /root/reference is read-only

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


# --------------------------------------------------------------------------------------
# pygsp stand-in (third-party dependency of the reference; not part of /root/reference)
# --------------------------------------------------------------------------------------
class _Graph:
    def __init__(self, W, coords=None, **kw):
        W = sp.csr_matrix(W, dtype=np.float64)
        W.eliminate_zeros()
        W.sort_indices()
        self.W = W
        self.N = W.shape[0]
        self.A = (W > 0).tocsr()
        self.A.sort_indices()
        self.dw = np.ravel(W.sum(axis=0))
        self.L = (sp.diags(self.dw, 0) - W).tocsc()
        self.Ne = sp.tril(W).nnz
        if coords is not None:
            self.coords = coords

    def is_directed(self):
        return (abs(self.W - self.W.T) > 1e-12).nnz > 0

    def get_edge_list(self):
        t = sp.tril(self.W).tocoo()
        return t.row, t.col, t.data


def _install_pygsp_standin():
    pkg = types.ModuleType("pygsp")
    graphs = types.ModuleType("pygsp.graphs")
    graphs.Graph = _Graph
    pkg.graphs = graphs
    pkg.filters = types.ModuleType("pygsp.filters")
    pkg.reduction = types.ModuleType("pygsp.reduction")
    sys.modules["pygsp"] = pkg
    sys.modules["pygsp.graphs"] = graphs
    sys.modules["pygsp.filters"] = pkg.filters
    sys.modules["pygsp.reduction"] = pkg.reduction


# --------------------------------------------------------------------------------------
# input graphs
# --------------------------------------------------------------------------------------
def ring_with_chords(n):
    """The reference's own synthetic scale-test graph: i~i+1 and i~i+2 (save_graphs.py:98-113)."""
    i = np.arange(n)
    r = np.concatenate([i, (i + 1) % n, i, (i + 2) % n])
    c = np.concatenate([(i + 1) % n, i, (i + 2) % n, i])
    W = sp.csr_matrix((np.ones(r.size), (r, c)), shape=(n, n))
    W.data[:] = 1.0
    return W


def barabasi_albert(n, m, seed, weighted=False):
    import networkx as nx

    g = nx.barabasi_albert_graph(n, m, seed=seed)
    W = sp.csr_matrix(nx.to_scipy_sparse_array(g, dtype=np.float64))
    if weighted:
        rng = np.random.default_rng(seed + 1000)
        U = sp.triu(W, 1).tocoo()
        w = rng.uniform(0.25, 2.0, size=U.nnz)
        U = sp.csr_matrix((w, (U.row, U.col)), shape=W.shape)
        W = (U + U.T).tocsr()
    return W


def star_plus_path(n_leaves, n_path):
    """Hub with twin leaves (exactly tied costs in exact arithmetic) + a tail path."""
    n = 1 + n_leaves + n_path
    r, c = [], []
    for l in range(1, n_leaves + 1):
        r += [0, l]
        c += [l, 0]
    prev = 0
    for p in range(n_leaves + 1, n):
        r += [prev, p]
        c += [p, prev]
        prev = p
    return sp.csr_matrix((np.ones(len(r)), (r, c)), shape=(n, n))


def cora_components():
    """Real Cora from the raw Planetoid pickles the reference ships (Baselines/SGGC/APPNP/dataset)."""
    d = os.path.join(REF, "Baselines/SGGC/APPNP/dataset/cora/raw")
    with open(os.path.join(d, "ind.cora.graph"), "rb") as f:
        graph = pickle.load(f, encoding="latin1")
    n = 2708
    r, c = [], []
    for u, nbrs in graph.items():
        for v in nbrs:
            if u != v:
                r += [u, v]
                c += [v, u]
    W = sp.csr_matrix((np.ones(len(r)), (r, c)), shape=(n, n))
    W.data[:] = 1.0  # collapse duplicates
    ncomp, lab = sp.csgraph.connected_components(W, directed=False)
    comps = [np.sort(np.nonzero(lab == k)[0]) for k in range(ncomp)]
    comps.sort(key=lambda a: (-len(a), a[0]))
    return W, comps


# --------------------------------------------------------------------------------------
# recording wrappers around the reference's module-level names
# --------------------------------------------------------------------------------------
class Recorder:
    def __init__(self, cu):
        self.cu = cu
        self.levels = []
        self.cur = None
        self._orig_cvl = cu.contract_variation_linear
        self._orig_gcm = cu.get_coarsening_matrix
        self._orig_sl = cu.SortedList
        rec = self

        class RecSortedList(self._orig_sl):
            def __init__(self, iterable=None, key=None):
                fam = list(iterable)
                rec.cur["ids"] = {id(c): k for k, c in enumerate(fam)}
                rec.cur["cost0"] = np.array([c.cost for c in fam], dtype=np.float64)
                rec.cur["set0"] = [np.array(c.set, dtype=np.int64) for c in fam]
                super().__init__(fam, key=key)

            def add(self, value):
                rec.cur["trace_cand"].append(rec.cur["ids"][id(value)])
                rec.cur["trace_set"].append(np.array(value.set, dtype=np.int64))
                rec.cur["trace_cost"].append(float(value.cost))
                super().add(value)

        def cvl(G, A=None, K=10, r=0.5, mode="neighborhood"):
            rec.cur = dict(
                A=np.array(A, dtype=np.float64), W=G.W.copy().tocsr(), dw=np.array(G.dw, dtype=np.float64),
                r_cur=float(r), trace_cand=[], trace_set=[], trace_cost=[],
            )
            out = rec._orig_cvl(G, A=A, K=K, r=r, mode=mode)
            rec.cur["coarsening_list"] = [np.array(s, dtype=np.int64) for s in out]
            return out

        def gcm(G, partitioning):
            iC = rec._orig_gcm(G, partitioning)
            rec.cur["iC"] = iC.copy()
            rec.levels.append(rec.cur)
            rec.cur = None
            return iC

        cu.contract_variation_linear = cvl
        cu.get_coarsening_matrix = gcm
        cu.SortedList = RecSortedList

    def reset(self):
        self.levels = []
        self.cur = None


def ragged(list_of_arrays):
    off = np.zeros(len(list_of_arrays) + 1, dtype=np.int64)
    for k, a in enumerate(list_of_arrays):
        off[k + 1] = off[k] + len(a)
    mem = np.concatenate(list_of_arrays) if len(list_of_arrays) else np.zeros(0, dtype=np.int64)
    return off, mem.astype(np.int64)


def spectral_pair(W, K=10, seed=0):
    """One eigsh run as the reference does it (coarsening_utils.py:83-90), made repeatable with v0."""
    G = _Graph(W)
    N = G.N
    offset = 2 * max(G.dw)
    T = offset * sp.eye(N, format="csc") - G.L
    rng = np.random.default_rng(seed)
    if K >= N:
        # The reference takes its dense branch here (coarsening_utils.py:85-86): eigsh on a dense
        # array with k >= N falls through to scipy.linalg.eigh, which is deterministic, and yields
        # only N (< K) columns.  Injection needs len(lk) >= K, so such graphs run un-injected.
        return None, None
    lk, Uk = spla.eigsh(T, k=K, which="LM", tol=1e-5, v0=rng.standard_normal(N))
    lk = (offset - lk)[::-1]
    Uk = Uk[:, ::-1]
    return np.ascontiguousarray(Uk), np.ascontiguousarray(lk)


def main():
    _install_pygsp_standin()
    sys.path.insert(0, REF)
    from graph_coarsening import coarsening_utils as cu  # the unmodified reference module

    rec = Recorder(cu)

    Wc, comps = cora_components()
    sizes = [len(c) for c in comps]
    pick26 = next(c for c in comps if len(c) == 26)
    pick9 = next(c for c in comps if len(c) == 9)
    pick2 = next(c for c in comps if len(c) == 2)

    def sub(W, idx):
        return W[idx, :][:, idx].tocsr()

    graphs = {
        "ring100": ring_with_chords(100),
        "ring400": ring_with_chords(400),
        "ba3000": barabasi_albert(3000, 2, 1),
        "ba600w": barabasi_albert(600, 3, 7, weighted=True),
        "star40": star_plus_path(30, 9),
        "cora_giant": sub(Wc, comps[0]),
        "cora26": sub(Wc, pick26),
        "cora9": sub(Wc, pick9),
        "cora2": sub(Wc, pick2),
    }
    ratios = [0.3, 0.5, 0.7, 0.9]
    manifest = {"reference": "Roy-Shubhajit/FIT-GNN @ /root/reference (graph_coarsening/coarsening_utils.py)",
                "layout": "one npz per graph; shared inputs at top level, per-ratio outputs under r<pct>_*, "
                          "per-level records under r<pct>_L<level>_*; level 0 consumes W, dw(W), A(Uk,lk)",
                "cora_component_sizes_top": sizes[:5], "cases": []}
    K = 10

    def i32(a):
        return np.asarray(a, dtype=np.int32)

    for name, W in graphs.items():
        W = sp.csr_matrix(W, dtype=np.float64)
        W.sort_indices()
        N = W.shape[0]
        Uk, lk = spectral_pair(W, K=K, seed=0)
        rng = np.random.default_rng(1)
        X = rng.random((N, 8), dtype=np.float32)
        out = {
            "W_indptr": i32(W.indptr), "W_indices": i32(W.indices), "W_data": W.data,
            "Uk": np.zeros((0, 0)) if Uk is None else Uk, "lk": np.zeros(0) if lk is None else lk,
            "X": X, "K": np.int64(K), "ratios": np.array(ratios),
        }
        for r in ratios:
            rec.reset()
            G = _Graph(W)
            rp = f"r{int(round(r * 100)):02d}_"
            try:
                C, Gc, mdl = cu.coarsen(G, K=K, r=r, method="variation_neighborhoods",
                                        Uk=None if Uk is None else Uk.copy(),
                                        lk=None if lk is None else lk.copy())
            except Exception as e:  # record that the reference itself fails on this input
                manifest["cases"].append({"name": name, "r": r, "error": repr(e)})
                continue
            C = sp.csc_matrix(C)
            # composed assignment (utils.py:113-121 semantics) straight from the reference's dict list
            assign = np.zeros(N, dtype=np.int32)
            for i in range(N):
                m = mdl[0][i]
                for j in range(1, len(mdl)):
                    m = mdl[j][m]
                assign[i] = m
            out.update({
                rp + "n_levels_recorded": np.int64(len(rec.levels)), rp + "n_mapping_dicts": np.int64(len(mdl)),
                rp + "C_indptr": i32(C.indptr), rp + "C_indices": i32(C.indices), rp + "C_data": C.data,
                rp + "C_shape": np.array(C.shape, dtype=np.int64), rp + "assign": assign,
                rp + "CX64": C.dot(X),  # utils.py:161: csc(f64) . dense(f32) -> f64
            })
            GW = sp.csr_matrix(Gc.W)
            GW.sort_indices()
            out.update({rp + "GcW_indptr": i32(GW.indptr), rp + "GcW_indices": i32(GW.indices),
                        rp + "GcW_data": GW.data, rp + "Gc_N": np.int64(Gc.N)})
            for li, L in enumerate(rec.levels):
                p = rp + f"L{li}_"
                if li == 0:
                    if "L0_cost0" not in out:  # level 0 does not depend on r: stored once
                        out["L0_cost0"] = L["cost0"]
                        out["L0_A"] = L["A"] if Uk is None else np.zeros((0, 0))
                    else:
                        assert np.array_equal(out["L0_cost0"], L["cost0"], equal_nan=True)
                else:
                    Wl = L["W"]
                    Wl.sort_indices()
                    out[p + "A"] = L["A"]
                    out[p + "W_indptr"], out[p + "W_indices"] = i32(Wl.indptr), i32(Wl.indices)
                    out[p + "W_data"] = Wl.data
                    out[p + "dw"] = L["dw"]
                    out[p + "cost0"] = L["cost0"]
                out[p + "r_cur"] = np.float64(L["r_cur"])
                out[p + "trace_cand"] = i32(L["trace_cand"])
                out[p + "trace_cost"] = np.array(L["trace_cost"], dtype=np.float64)
                to, tm = ragged(L["trace_set"])
                out[p + "trace_off"], out[p + "trace_mem"] = i32(to), i32(tm)
                co, cm = ragged(L["coarsening_list"])
                out[p + "sel_off"], out[p + "sel_mem"] = i32(co), i32(cm)
                iC = sp.csc_matrix(L["iC"])
                out[p + "iC_indptr"], out[p + "iC_indices"] = i32(iC.indptr), i32(iC.indices)
                out[p + "iC_data"] = iC.data
                out[p + "iC_shape"] = np.array(iC.shape, dtype=np.int64)
            manifest["cases"].append({"name": name, "r": r, "file": f"coarsen_{name}.npz", "N": int(N),
                                      "n": int(C.shape[0]), "levels": len(rec.levels),
                                      "recosts": [len(L["trace_cost"]) for L in rec.levels]})
            print(name, r, "N", N, "n", C.shape[0], "levels", len(rec.levels),
                  "recosts", [len(L["trace_cost"]) for L in rec.levels])
        np.savez_compressed(os.path.join(HERE, f"coarsen_{name}.npz"), **out)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
