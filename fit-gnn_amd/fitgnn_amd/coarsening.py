"""`coarsen()` with the reference's signature and return values, contraction step on the MI355X.

Mirror of graph_coarsening/coarsening_utils.py:18-182 (`coarsen`) for the variation_neighborhoods
method, the only one BASELINE.json's north_star names.  Per level:

    host   spectral prelude  A = Uk diag(lk^-1/2)            (:75-96;  ARPACK / LAPACK)
           or               A = B diag(d^-1/2) V             (:99-105)
    device candidate family + local-variation costs          (:571-578, :555-561)   fitgnn_variation_costs_f64
    device greedy disjoint selection with re-costing         (:604-650)             fitgnn_greedy_select
    device assignment vector / C values                      (:212-254, :168-179)   fitgnn_build_assignment
    device adjacency lift Wc = zero_diag(P^T W P) symmetrised (:138-139, :201-205)   fitgnn_lift_adjacency

Returned objects expose what FIT-GNN's callers touch (utils.py:159-184, :723-752, main.py:144-151):
`C` is a scipy csc matrix (subclass whose `.dot(dense)` runs the pooling kernel), `Gc` a light graph with
`.N .W .A .dw .L`, `mapping_dict_list` the per-level dicts.  There is no CPU path for the contraction step.
"""
import numpy as np
import scipy.sparse as sp
import torch

from . import _lib


# ---------------------------------------------------------------------------------------------
# light graph object (what the variation path reads from pygsp.graphs.Graph)
# ---------------------------------------------------------------------------------------------
class Graph:
    def __init__(self, W, coords=None):
        W = sp.csr_matrix(W, dtype=np.float64)
        W.eliminate_zeros()
        W.sort_indices()
        self.W = W
        self.N = W.shape[0]
        self._A = self._dw = self._L = None
        if coords is not None:
            self.coords = coords
        self.info = {}

    @property
    def A(self):
        if self._A is None:
            self._A = (self.W > 0).tocsr()
        return self._A

    @property
    def dw(self):
        if self._dw is None:
            self._dw = np.ravel(self.W.sum(axis=0))
        return self._dw

    @property
    def L(self):
        if self._L is None:
            self._L = (sp.diags(self.dw, 0) - self.W).tocsc()
        return self._L

    @property
    def Ne(self):
        return sp.tril(self.W).nnz

    def is_directed(self):
        return (abs(self.W - self.W.T) > 1e-12).nnz > 0

    def subgraph(self, ind):
        return Graph(self.W[ind, :][:, ind])

    def extract_components(self):
        ncomp, lab = sp.csgraph.connected_components(self.W, directed=False)
        order = np.argsort(lab, kind="stable")
        bounds = np.concatenate([[0], np.cumsum(np.bincount(lab, minlength=ncomp))])
        firsts = [order[bounds[k]] for k in range(ncomp)]
        out = []
        for k in np.argsort(firsts, kind="stable"):  # pygsp discovers components by lowest unvisited node
            idx = np.sort(order[bounds[k]:bounds[k + 1]])
            g = self.subgraph(idx)
            g.info = {"orig_idx": idx.tolist()}
            out.append(g)
        return out


# ---------------------------------------------------------------------------------------------
# host prelude (not part of the accelerated step; same NumPy/SciPy calls as the reference)
# ---------------------------------------------------------------------------------------------
def _spectral_level1(G, K, Uk, lk):
    import scipy.sparse.linalg as spla

    if (Uk is not None) and (lk is not None) and (len(lk) >= K):
        mask = lk < 1e-10
        lk[mask] = 1
        lsinv = lk ** (-0.5)
        lsinv[mask] = 0
        return Uk[:, :K] @ np.diag(lsinv[:K])
    offset = 2 * max(G.dw)
    T = offset * sp.eye(G.N, format="csc") - G.L
    if K >= G.N:
        lk, Uk = spla.eigsh(T.toarray(), k=K, which="LM", tol=1e-5)
    else:
        lk, Uk = spla.eigsh(T, k=K, which="LM", tol=1e-5)
    lk = (offset - lk)[::-1]
    Uk = Uk[:, ::-1]
    mask = lk < 1e-10
    lk[mask] = 1
    lsinv = lk ** (-0.5)
    lsinv[mask] = 0
    return Uk @ np.diag(lsinv)


def lanczos_smallest(L, K, device="cuda", tol=1e-5, m=None, max_restarts=300, seed=0):
    """The K smallest eigenpairs of the graph Laplacian L on the device: thick-restart Lanczos with full (two-pass)
    reorthogonalisation in float64 on T = 2 max(diag L) I - L, the shifted operator the reference hands to ARPACK
    (coarsening_utils.py:83-90), same relative tolerance.  SURVEY §8 f4: 42 % of the reference's coarsening time is this
    solve.  One Lanczos step = five launches of csrc/lanczos.hip over a column-major basis (the CSR product, three projection
    passes, the normalisation; fixed-order sums: reproducible); the projected m x m eigenproblem and the basis rotation of a
    restart are small dense library calls.  Returns (lk ascending, Uk) as NumPy arrays, the (lk, Uk) coarsen() accepts."""
    Lh = _lib.lib()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.FitgnnError("lanczos_smallest runs on the MI355X (spectral='host' is the reference's ARPACK call)")
    # L is symmetric: a CSC matrix's arrays ARE the CSR arrays of the same matrix (Graph.L is CSC) -- no conversion, no T on the host:
    # the device product computes y = offset x - L x
    Lc = L if sp.isspmatrix_csc(L) or sp.isspmatrix_csr(L) else sp.csr_matrix(L)
    N = Lc.shape[0]
    offset = 2.0 * float(Lc.diagonal().max())
    rowptr = torch.from_numpy(np.ascontiguousarray(Lc.indptr, dtype=np.int32)).to(dev)
    col = torch.from_numpy(np.ascontiguousarray(Lc.indices, dtype=np.int32)).to(dev)
    val = torch.from_numpy(np.ascontiguousarray(Lc.data, dtype=np.float64)).to(dev)
    m = int(m or min(N - 1, max(4 * K + 20, 60)))
    if m + 1 > 128:
        raise ValueError("lanczos_smallest: at most 127 basis vectors (fitgnn_lanczos_project_f64)")
    K = min(K, m - 1)
    st = _lib.stream_ptr(dev)
    gen = torch.Generator(device="cpu").manual_seed(seed)
    V = torch.zeros(m + 1, N, dtype=torch.float64, device=dev)        # basis vector c = V[c] (contiguous)
    V2 = torch.zeros_like(V)                                           # the rotated basis of a restart
    v = torch.randn(N, generator=gen, dtype=torch.float64).to(dev)
    V[0] = v / v.norm()
    H = torch.zeros(m + 1, m, dtype=torch.float64, device=dev)
    w = torch.empty(N, dtype=torch.float64, device=dev)
    parts = int(Lh.fitgnn_lanczos_parts(N))
    part = torch.empty(parts * (m + 2), dtype=torch.float64, device=dev)
    ha, hb, hc = (torch.empty(m + 2, dtype=torch.float64, device=dev) for _ in range(3))

    def project(ncol, h_in, h_out):
        _lib.check(Lh.fitgnn_lanczos_project_f64(_lib.dptr(V), N, ncol, _lib.dptr(w), N, _lib.dptr(h_in), _lib.dptr(part), st), "lanczos_project")
        _lib.check(Lh.fitgnn_lanczos_reduce_f64(_lib.dptr(part), parts, ncol + 1, _lib.dptr(h_out), st), "lanczos_reduce")

    def rotate(src, Smat, dst):
        """dst[c] = sum_j Smat[j, c] src[j] (Smat: host array [m, nk]); the kernel rotates at most 16 columns per launch: groups."""
        for c0 in range(0, int(Smat.shape[1]), 16):
            blk = Smat[:, c0:c0 + 16]
            Sd = torch.from_numpy(np.ascontiguousarray(blk, dtype=np.float64)).to(dev)
            _lib.check(Lh.fitgnn_lanczos_rotate_f64(_lib.dptr(src), N, int(blk.shape[0]), _lib.dptr(Sd), int(blk.shape[1]), _lib.dptr(dst[c0:]), N, N,
                                                    st), "lanczos_rotate")

    j0 = 0
    for _ in range(max_restarts):
        for j in range(j0, m):
            _lib.check(Lh.fitgnn_lanczos_spmv_f64(_lib.dptr(rowptr), _lib.dptr(col), _lib.dptr(val), _lib.dptr(V[j]), _lib.dptr(w), N, -1.0, offset,
                                                  st), "fitgnn_lanczos_spmv_f64")
            # h = V^T w;  w -= V h, h2 = V^T w;  w -= V h2, |w|^2;  v_{j+1} = w / |w|, H[:, j] = h + h2
            project(j + 1, None, ha)
            project(j + 1, ha, hb)
            project(j + 1, hb, hc)
            _lib.check(Lh.fitgnn_lanczos_finish_f64(_lib.dptr(V), N, j, _lib.dptr(w), N, _lib.dptr(ha), _lib.dptr(hb), _lib.dptr(hc), _lib.dptr(H), m,
                                                    st), "fitgnn_lanczos_finish_f64")
        # the projected m x m eigenproblem on the host (60 x 60: LAPACK takes less than the launch of a device solver)
        Hh = H.cpu().numpy()
        Hm = (Hh[:m, :m] + Hh[:m, :m].T) / 2
        theta, S = np.linalg.eigh(Hm)
        order = np.argsort(-theta, kind="stable")
        idx = order[:K]
        resid = np.abs(Hh[m, m - 1] * S[m - 1, idx])
        if float(resid.max()) <= tol * float(np.abs(theta).max()):
            break
        keep = order[:min(K + 5, m - 2)]
        nk = int(keep.size)
        rotate(V, S[:, keep], V2)                 # V2[:nk] = the kept Ritz vectors
        V2[nk].copy_(V[m])
        Hn = np.zeros_like(Hh)
        Hn[:nk, :nk] = np.diag(theta[keep])
        Hn[nk, :nk] = Hh[m, m - 1] * S[m - 1, keep]
        V, V2 = V2, V
        H.copy_(torch.from_numpy(Hn))
        j0 = nk
    lk = offset - theta[idx]
    rotate(V, S[:, idx], V2)
    Uk = V2[:K].T.contiguous().cpu().numpy()
    o = np.argsort(lk)
    return lk[o], np.ascontiguousarray(Uk[:, o])


def _spectral_next(G, iC, B):
    B = iC.dot(B)
    d, V = np.linalg.eig(B.T @ (G.L).dot(B))
    mask = d == 0
    d[mask] = 1
    dinvsqrt = d ** (-1 / 2)
    dinvsqrt[mask] = 0
    return B, B @ np.diag(dinvsqrt) @ V


# ---------------------------------------------------------------------------------------------
# device pipeline for one level
# ---------------------------------------------------------------------------------------------
def _dev(a, dtype, device):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(device)


class LevelResult:
    __slots__ = ("N", "n", "assign", "cval", "cost0", "sel_off", "sel_mem", "rowptr", "col", "w", "device")


def contract_level(G, A, r_cur, device="cuda", keep_debug=False):
    """One contraction level on the GPU.  Returns LevelResult with device tensors assign (int32[N]),
    cval (float64[N]) and n (int)."""
    L = _lib.lib()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.FitgnnError("contract_level needs the MI355X (no CPU fallback)")
    st = _lib.stream_ptr(dev)
    N = G.N
    W = G.W
    A = np.ascontiguousarray(np.real(A), dtype=np.float64)
    K = A.shape[1]
    if not (1 <= K <= _lib.MAX_K):
        raise _lib.FitgnnError(f"K={K} outside [1,{_lib.MAX_K}]")
    rowptr = _dev(W.indptr, torch.int32, dev)
    col = _dev(W.indices, torch.int32, dev)
    w = _dev(W.data, torch.float64, dev)
    dw = _dev(G.dw, torch.float64, dev)
    Ad = _dev(A, torch.float64, dev)
    nnz = int(W.nnz)
    set_off = torch.empty(N + 1, dtype=torch.int32, device=dev)
    set_mem = torch.empty(nnz + N, dtype=torch.int32, device=dev)
    _lib.check(L.fitgnn_closed_neighbourhoods(_lib.dptr(rowptr), _lib.dptr(col), N, _lib.dptr(set_off), _lib.dptr(set_mem), st),
               "closed_neighbourhoods")
    set_len = (set_off[1:] - set_off[:-1]).contiguous()
    cost0 = torch.empty(N, dtype=torch.float64, device=dev)
    _lib.check(L.fitgnn_variation_costs_f64(_lib.dptr(rowptr), _lib.dptr(col), _lib.dptr(w), _lib.dptr(dw), _lib.dptr(Ad), K, K,
                                            _lib.dptr(set_off), _lib.dptr(set_len), _lib.dptr(set_mem), N, _lib.dptr(cost0), st),
               "variation_costs")
    n_reduce = int(np.floor(r_cur * N))  # coarsening_utils.py:612
    total = nnz + N
    wb = int(L.fitgnn_greedy_select_workspace_bytes(N, total))
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    sel_off = torch.empty(N + 1, dtype=torch.int32, device=dev)
    sel_mem = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
    sel_count = torch.zeros(2, dtype=torch.int32, device=dev)
    _lib.check(L.fitgnn_greedy_select(_lib.dptr(rowptr), _lib.dptr(col), _lib.dptr(w), _lib.dptr(dw), _lib.dptr(Ad), K, K, N,
                                      _lib.dptr(set_off), _lib.dptr(set_mem), _lib.dptr(cost0), n_reduce, _lib.dptr(sel_off),
                                      _lib.dptr(sel_mem), _lib.dptr(sel_count), _lib.dptr(work), wb, st), "greedy_select")
    assign = torch.empty(N, dtype=torch.int32, device=dev)
    cval = torch.empty(N, dtype=torch.float64, device=dev)
    n_out = torch.zeros(1, dtype=torch.int32, device=dev)
    wb2 = int(L.fitgnn_build_assignment_workspace_bytes(N))
    work2 = torch.empty(wb2, dtype=torch.uint8, device=dev)
    _lib.check(L.fitgnn_build_assignment(N, _lib.dptr(sel_off), _lib.dptr(sel_mem), _lib.dptr(sel_count), _lib.dptr(assign),
                                         _lib.dptr(cval), _lib.dptr(n_out), _lib.dptr(work2), wb2, st), "build_assignment")
    res = LevelResult()
    res.N, res.n, res.assign, res.cval, res.device = N, int(n_out.item()), assign, cval, dev
    res.rowptr, res.col, res.w = rowptr, col, w
    res.cost0 = cost0 if keep_debug else None
    if keep_debug:
        cnt = sel_count.cpu().numpy()
        res.sel_off = sel_off[: cnt[0] + 1].cpu().numpy()
        res.sel_mem = sel_mem[: cnt[1]].cpu().numpy()
    else:
        res.sel_off = res.sel_mem = None
    return res


def lift_adjacency(res):
    """Wc (scipy csr f64) = symmetrised zero-diagonal P^T W P of a LevelResult."""
    L = _lib.lib()
    dev = res.device
    st = _lib.stream_ptr(dev)
    nnz = int(res.col.numel())
    wb = int(L.fitgnn_lift_adjacency_workspace_bytes(res.N, nnz, res.n))
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    rp = torch.empty(res.n + 1, dtype=torch.int32, device=dev)
    cc = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
    wc = torch.empty(max(nnz, 1), dtype=torch.float64, device=dev)
    nz = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(L.fitgnn_lift_adjacency(res.N, _lib.dptr(res.rowptr), _lib.dptr(res.col), _lib.dptr(res.w), _lib.dptr(res.assign),
                                       _lib.dptr(res.cval), res.n, _lib.dptr(rp), _lib.dptr(cc), _lib.dptr(wc), _lib.dptr(nz),
                                       _lib.dptr(work), wb, st), "lift_adjacency")
    m = int(nz.item())
    return sp.csr_matrix((wc[:m].cpu().numpy(), cc[:m].cpu().numpy(), rp.cpu().numpy()), shape=(res.n, res.n))


def pool_rows(assign, cval, n, X, want_f64=False):
    """Xc = C . X on the device.  assign int32[N], cval float64[N] device tensors; X float32 [N,F] device."""
    L = _lib.lib()
    _lib.require_cuda(assign, cval, X)
    dev = X.device
    X = X.float().contiguous()
    N, F = X.shape
    Xc = torch.empty((n, F), dtype=torch.float32, device=dev)
    Xc64 = torch.empty((n, F), dtype=torch.float64, device=dev) if want_f64 else None
    wb = int(L.fitgnn_pool_rows_workspace_bytes(N, n))
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    _lib.check(L.fitgnn_pool_rows_f32(_lib.dptr(assign), _lib.dptr(cval), N, n, _lib.dptr(X), F, F, _lib.dptr(Xc), F,
                                      _lib.dptr(Xc64), _lib.dptr(work), wb, _lib.stream_ptr(dev)), "pool_rows")
    return (Xc, Xc64) if want_f64 else Xc


class CoarseningMatrix(sp.csc_matrix):
    """scipy csc matrix C [n x N] whose product with a dense [N x F] operand runs on the MI355X.

    FIT-GNN pools features and labels with `C.dot(X)` (utils.py:161,393,738,827); this subclass keeps that
    spelling working: `.dot(ndarray | torch.Tensor)` returns the float64 ndarray scipy would return, computed
    by fitgnn_pool_rows_f32 (f64 accumulation in ascending member order: bit-identical).  `.pool(X)` is the
    device-to-device form (float32 tensor in, float32 tensor out) the build's own pipeline uses.
    """

    def _vectors(self, device):
        key = str(device)
        cache = self.__dict__.setdefault("_fitgnn_dev", {})
        if key not in cache:
            csc = sp.csc_matrix(self)
            assert np.all(np.diff(csc.indptr) == 1), "C must have exactly one non-zero per column"
            cache[key] = (torch.as_tensor(csc.indices.astype(np.int32)).to(device),
                          torch.as_tensor(csc.data.astype(np.float64)).to(device))
        return cache[key]

    def pool(self, X):
        assign, cval = self._vectors(X.device)
        return pool_rows(assign, cval, self.shape[0], X)

    def dot(self, other):
        dense = torch.is_tensor(other) or (isinstance(other, np.ndarray) and other.ndim == 2)
        if not dense or not torch.cuda.is_available():
            if not dense:
                return sp.csc_matrix.dot(self, other)
            raise _lib.FitgnnError("C.dot(dense) runs on the MI355X only (no CPU fallback)")
        Xt = other if torch.is_tensor(other) else torch.from_numpy(np.ascontiguousarray(other))
        if Xt.dtype == torch.float64 and not torch.is_tensor(other):
            # label/mask pooling passes small f64 one-hot matrices (utils.py:726-742): values are exactly
            # representable in f32, so the f32 kernel input loses nothing
            pass
        dev = torch.device("cuda")
        assign, cval = self._vectors(dev)
        _, x64 = pool_rows(assign, cval, self.shape[0], Xt.to(dev).float(), want_f64=True)
        return x64.cpu().numpy()


# ---------------------------------------------------------------------------------------------
# the drop-in driver
# ---------------------------------------------------------------------------------------------
def coarsen(G, K=10, r=0.5, max_levels=10, method="variation_neighborhood", algorithm="greedy", Uk=None, lk=None,
            max_level_r=0.99, device="cuda", spectral="arpack"):
    """Same contract as graph_coarsening.coarsening_utils.coarsen (coarsening_utils.py:18-182) for
    method in {'variation_neighborhood', 'variation_neighborhoods'}: returns (C, Gc, mapping_dict_list)."""
    if "variation_neighborhood" not in method:
        raise NotImplementedError(f"method '{method}' is outside the accelerated hot path (variation_neighborhoods only)")
    if not hasattr(G, "W"):
        raise TypeError("G must expose .W (scipy sparse adjacency) and .N")
    if not isinstance(G, Graph):
        G = Graph(G.W, coords=getattr(G, "coords", None))
    r = np.clip(r, 0, 0.999)
    N = G.N
    n, n_target = N, np.ceil((1 - r) * N)
    dev = torch.device(device)
    L = _lib.lib()
    assign_tot = torch.arange(N, dtype=torch.int32, device=dev)
    cval_tot = torch.ones(N, dtype=torch.float64, device=dev)
    Gc = G
    mapping_dict_list = []
    B = iC = None
    for level in range(1, max_levels + 1):
        G = Gc
        r_cur = np.clip(1 - n_target / n, 0.0, max_level_r)
        if level == 1:
            if spectral == "device" and Uk is None and G.N > 4 * K:  # extension: the eigensolve on the MI355X
                lk, Uk = lanczos_smallest(G.L, K, device=dev)
            B = _spectral_level1(G, K, Uk, lk)
            A = B
        else:
            B, A = _spectral_next(G, iC, B)
        res = contract_level(G, A, r_cur, device=dev)
        assign_h = res.assign.cpu().numpy()
        iC = sp.csc_matrix((res.cval.cpu().numpy(), (assign_h, np.arange(G.N))), shape=(res.n, G.N))
        if iC.shape[1] - iC.shape[0] <= 2:  # :131-135 avoid too many levels for so few nodes
            mapping_dict_list.append({i: i for i in range(G.N)})
            break
        _lib.check(L.fitgnn_compose_levels(N, _lib.dptr(res.assign), _lib.dptr(res.cval), _lib.dptr(assign_tot),
                                           _lib.dptr(cval_tot), _lib.stream_ptr(dev)), "compose_levels")
        Wc = lift_adjacency(res)
        coords = None
        if hasattr(G, "coords"):
            coords = (iC.power(2)).dot(G.coords)  # coarsen_vector :190-191 (plot coordinates only)
        Gc = Graph(Wc, coords=coords)
        n = Gc.N
        # level mapping :168-179: keys 0..N-1 of the ORIGINAL graph, identity-padded past the level's size
        md = {i: int(assign_h[i]) for i in range(G.N)}
        for i in range(G.N, N):
            md[i] = res.n + (i - G.N)
        mapping_dict_list.append(md)
        if n <= n_target:
            break
    a = assign_tot.cpu().numpy()
    C = CoarseningMatrix(sp.csc_matrix((cval_tot.cpu().numpy(), (a, np.arange(N))), shape=(int(a.max()) + 1 if N else 0, N)))
    return C, Gc, mapping_dict_list


# ---------------------------------------------------------------------------------------------
# many graphs at once: the reference's per-component / per-dataset-graph Python loop as one batch
# ---------------------------------------------------------------------------------------------
class BatchCoarsening:
    """Result of coarsen_batch: `assign` int64[N] (global cluster id of every node; a component's clusters are a
    contiguous id range in ascending order of their minimum member, i.e. per-component id = assign - cluster_off[c]),
    `cval` float64[N] (the non-zero of C in that node's column), `comp_off` / `cluster_off` int64[n_comp+1],
    `Wc` scipy csr (block-diagonal coarse adjacency, global cluster ids), `levels` per-component level count."""
    __slots__ = ("assign", "cval", "comp_off", "cluster_off", "Wc", "levels", "n_clusters", "_dev")

    def C(self):
        N = len(self.assign)
        return CoarseningMatrix(sp.csc_matrix((self.cval, (self.assign, np.arange(N))), shape=(self.n_clusters, N)))

    def pool(self, X):
        """C . X for every component at once (device tensor in / out)."""
        a, c = self._dev
        return pool_rows(a.to(X.device), c.to(X.device), self.n_clusters, X)


def _dense_prelude(W, b, e, K):
    """Level-1 spectral input of one small component from a dense symmetric eigendecomposition (all eigenpairs,
    smallest K kept) -- same formula as coarsening_utils.py:89-96, exact eigenvectors instead of ARPACK's tol=1e-5."""
    Wd = W[b:e, b:e].toarray()
    Ld = np.diag(Wd.sum(0)) - Wd
    lk, Uk = np.linalg.eigh(Ld)
    k = min(K, e - b)
    lk, Uk = lk[:k].copy(), Uk[:, :k]
    mask = lk < 1e-10
    lk[mask] = 1
    lsinv = lk ** (-0.5)
    lsinv[mask] = 0
    return Uk @ np.diag(lsinv)


def _dense_prelude_batch(W, off, comps, K, A, Kc):
    """_dense_prelude for many components at once: components of equal size are stacked into one [G, n, n] array and
    handed to a single batched np.linalg.eigh (the same LAPACK routine per matrix as the one-at-a-time form).  Fills
    A[rows of c, :k] and Kc[c] in place."""
    coo = W.tocoo()
    comp_of = np.searchsorted(off, coo.row, side="right") - 1
    size = np.diff(off)
    todo = np.zeros(len(size), dtype=bool)
    todo[comps] = True
    for n in np.unique(size[comps]):
        ids = np.nonzero(todo & (size == n))[0]
        slot = np.full(len(size), -1, dtype=np.int64)
        slot[ids] = np.arange(len(ids))
        sel = slot[comp_of] >= 0
        g, r, c = slot[comp_of[sel]], coo.row[sel] - off[comp_of[sel]], coo.col[sel] - off[comp_of[sel]]
        Wd = np.zeros((len(ids), n, n))
        np.add.at(Wd, (g, r, c), coo.data[sel])
        Ld = -Wd
        Ld[:, np.arange(n), np.arange(n)] += Wd.sum(1)
        lk, Uk = np.linalg.eigh(Ld)
        k = int(min(K, n))
        lk, Uk = lk[:, :k].copy(), Uk[:, :, :k]
        mask = lk < 1e-10
        lk[mask] = 1
        lsinv = lk ** (-0.5)
        lsinv[mask] = 0
        Ag = Uk * lsinv[:, None, :]
        rows = (off[ids][:, None] + np.arange(n)[None, :]).ravel()
        A[rows, :k] = Ag.reshape(-1, k)
        Kc[ids] = k


def coarsen_batch(W, comp_off, r=0.5, K=10, max_levels=10, A0=None, max_level_r=0.99, device="cuda", spectral="arpack"):
    """coarsen() (coarsening_utils.py:18-182, method variation_neighborhoods) applied independently to every connected
    component of the block-diagonal adjacency W (scipy sparse [N x N]); component c = node range
    comp_off[c]:comp_off[c+1] and must be connected.  Per level ONE launch each of the family, cost, selection (one
    wavefront per component), assignment and lift kernels covers all components; a component leaves the loop exactly
    when the reference's driver would (target reached :180, or a level that removes <= 2 nodes :131-135, which is not
    applied).  The spectral prelude stays on the host per component (level 1: ARPACK as the reference, or
    spectral='dense'; later levels: the K x K eigenproblem of :98-104).  A0: optional list of per-component level-1
    matrices A (n_c x K_c), e.g. from injected (Uk, lk).  Returns BatchCoarsening."""
    L = _lib.lib()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.FitgnnError("coarsen_batch needs the MI355X (no CPU fallback)")
    st = _lib.stream_ptr(dev)
    W = sp.csr_matrix(W).astype(np.float64)
    W.sort_indices()
    comp_off = np.asarray(comp_off, dtype=np.int64)
    n_comp, N0 = len(comp_off) - 1, int(comp_off[-1])
    assert W.shape == (N0, N0)
    r = float(np.clip(r, 0, 0.999))
    size0 = np.diff(comp_off)
    n_cur = size0.copy()
    n_target = np.ceil((1 - r) * size0)
    active = size0 > 1
    levels = np.zeros(n_comp, dtype=np.int64)
    assign_tot = torch.arange(N0, dtype=torch.int32, device=dev)
    cval_tot = torch.ones(N0, dtype=torch.float64, device=dev)
    B = None            # pooled spectral basis [N_level x Kmax], rows of inactive components are not used
    node_K = None
    iC = None
    off = comp_off.copy()
    G = Graph(W)
    for level in range(1, max_levels + 1):
        if not active.any():
            break
        N = G.N
        r_cur = np.clip(1 - n_target / np.maximum(n_cur, 1), 0.0, max_level_r)
        n_reduce = np.where(active, np.floor(r_cur * n_cur), 0).astype(np.int64)  # :612 per component
        if level == 1:
            Kc = np.where(size0 <= K, size0, K).astype(np.int32)  # eigsh(dense, k=K >= N) returns N pairs (:85-86)
            Kmax = int(max(K, Kc.max()))
            A = np.zeros((N, Kmax))
            todo = [c for c in np.nonzero(active)[0] if A0 is None or A0[c] is None]
            if spectral == "dense" and todo:
                _dense_prelude_batch(W, off, np.asarray(todo), K, A, Kc)
                todo = []
            for c in np.nonzero(active)[0]:
                b, e = int(off[c]), int(off[c + 1])
                if A0 is not None and A0[c] is not None:
                    Ac = np.asarray(A0[c], dtype=np.float64)
                elif c in todo:
                    Ac = _spectral_level1(Graph(W[b:e, b:e]), K, None, None)
                else:
                    continue
                Kc[c] = Ac.shape[1]
                A[b:e, :Ac.shape[1]] = np.real(Ac)
            node_K_comp = Kc
            B = A
        else:
            B = iC.dot(B)                      # :97, all components at once (rows are independent)
            LB = G.L.dot(B)
            A = np.zeros_like(B)
            for c in np.nonzero(active)[0]:    # :98-104, K x K per component
                b, e, k = int(off[c]), int(off[c + 1]), int(node_K_comp[c])
                Bc = B[b:e, :k]
                d, V = np.linalg.eig(Bc.T @ LB[b:e, :k])
                mask = d == 0
                d[mask] = 1
                dinvsqrt = d ** (-1 / 2)
                dinvsqrt[mask] = 0
                A[b:e, :k] = np.real(Bc @ np.diag(dinvsqrt) @ V)
        node_K = np.repeat(node_K_comp, np.diff(off)).astype(np.int32)
        # ---- device: family, costs, per-component selection, assignment ----
        Wl = G.W
        rowptr, col = _dev(Wl.indptr, torch.int32, dev), _dev(Wl.indices, torch.int32, dev)
        w, dw = _dev(Wl.data, torch.float64, dev), _dev(G.dw, torch.float64, dev)
        Ad, nK = _dev(A, torch.float64, dev), _dev(node_K, torch.int32, dev)
        lda = int(A.shape[1])
        nnz = int(Wl.nnz)
        set_off = torch.empty(N + 1, dtype=torch.int32, device=dev)
        set_mem = torch.empty(nnz + N, dtype=torch.int32, device=dev)
        _lib.check(L.fitgnn_closed_neighbourhoods(_lib.dptr(rowptr), _lib.dptr(col), N, _lib.dptr(set_off), _lib.dptr(set_mem), st),
                   "closed_neighbourhoods")
        set_len = (set_off[1:] - set_off[:-1]).contiguous()
        cost0 = torch.empty(N, dtype=torch.float64, device=dev)
        _lib.check(L.fitgnn_variation_costs_batch_f64(_lib.dptr(rowptr), _lib.dptr(col), _lib.dptr(w), _lib.dptr(dw), _lib.dptr(Ad),
                                                      lda, lda, _lib.dptr(nK), _lib.dptr(set_off), _lib.dptr(set_len),
                                                      _lib.dptr(set_mem), N, _lib.dptr(cost0), st), "variation_costs_batch")
        wb = int(L.fitgnn_greedy_select_batch_workspace_bytes(N, nnz + N, n_comp))
        work = torch.empty(wb, dtype=torch.uint8, device=dev)
        sel_off = torch.empty(N + 1, dtype=torch.int32, device=dev)
        sel_mem = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
        sel_count = torch.zeros(2, dtype=torch.int32, device=dev)
        gain_d = torch.zeros(n_comp, dtype=torch.int64, device=dev)
        off_d, n_reduce_d = _dev(off, torch.int32, dev), _dev(n_reduce, torch.int64, dev)  # named: they must outlive the call
        _lib.check(L.fitgnn_greedy_select_batch(_lib.dptr(rowptr), _lib.dptr(col), _lib.dptr(w), _lib.dptr(dw), _lib.dptr(Ad), lda,
                                                lda, N, _lib.dptr(set_off), _lib.dptr(set_mem), _lib.dptr(cost0), n_comp,
                                                _lib.dptr(off_d), _lib.dptr(n_reduce_d),
                                                2, _lib.dptr(nK), _lib.dptr(sel_off), _lib.dptr(sel_mem), _lib.dptr(sel_count),
                                                _lib.dptr(gain_d), _lib.dptr(work), wb, st), "greedy_select_batch")
        assign = torch.empty(N, dtype=torch.int32, device=dev)
        cval = torch.empty(N, dtype=torch.float64, device=dev)
        n_out = torch.zeros(1, dtype=torch.int32, device=dev)
        wb2 = int(L.fitgnn_build_assignment_workspace_bytes(N))
        work2 = torch.empty(wb2, dtype=torch.uint8, device=dev)
        _lib.check(L.fitgnn_build_assignment(N, _lib.dptr(sel_off), _lib.dptr(sel_mem), _lib.dptr(sel_count), _lib.dptr(assign),
                                             _lib.dptr(cval), _lib.dptr(n_out), _lib.dptr(work2), wb2, st), "build_assignment")
        gain = gain_d.cpu().numpy()
        applied = active & (gain > 2)          # :131-135: a level removing <= 2 nodes is not applied and ends the loop
        levels[applied] += 1
        active = applied.copy()
        if not applied.any():
            break
        _lib.check(L.fitgnn_compose_levels(N0, _lib.dptr(assign), _lib.dptr(cval), _lib.dptr(assign_tot), _lib.dptr(cval_tot), st),
                   "compose_levels")
        res = LevelResult()
        res.N, res.n, res.assign, res.cval, res.device = N, int(n_out.item()), assign, cval, dev
        res.rowptr, res.col, res.w = rowptr, col, w
        Wc = lift_adjacency(res)
        assign_h = assign.cpu().numpy()
        iC = sp.csc_matrix((cval.cpu().numpy(), (assign_h, np.arange(N))), shape=(res.n, N))
        n_cur = np.where(applied, n_cur - gain, n_cur)
        new_off = np.zeros(n_comp + 1, dtype=np.int64)
        np.cumsum(n_cur, out=new_off[1:])
        assert int(new_off[-1]) == res.n, (level, int(new_off[-1]), res.n, gain.tolist(), n_reduce.tolist(), sel_count.cpu().tolist())
        off = new_off
        G = Graph(Wc)
        active &= n_cur > n_target             # :180
    out = BatchCoarsening()
    out.assign = assign_tot.cpu().numpy().astype(np.int64)
    out.cval = cval_tot.cpu().numpy()
    out.comp_off, out.cluster_off, out.Wc, out.levels = comp_off, off, G.W, levels
    out.n_clusters = int(off[-1])
    out._dev = (assign_tot, cval_tot)
    return out
