"""Static device-resident CSR batches for the SpMM kernels.

FIT-GNN feeds PyG layers an `edge_index` [2,E] int64 per call (network.py:31) and rebuilds the
block-diagonal batch on the CPU every epoch (run.py:336 G_DataLoader(shuffle=False)).
Here the conversion happens ONCE per static batch: int32 CSR by target row (+ its transpose for the
backward pass), GCN normalisation values, and the row tiles whose column windows the kernel stages in
LDS.  Index plumbing uses torch tensor ops and works on CPU tensors too (unit-tested without a GPU);
the arithmetic (normalisation, SpMM) runs only through libfitgnn_hip.so.
"""
import ctypes
import weakref

import numpy as np
import torch

from . import _lib

TILE_DTYPE = np.dtype([("row_begin", np.int32), ("row_end", np.int32), ("win_begin", np.int32), ("win_rows", np.int32)])


def block_boundaries(rowptr, col, n_rows):
    """Node offsets `ptr` of the diagonal blocks of a square CSR pattern (works for any matrix: a matrix
    that is not block diagonal yields one block).  i is a boundary iff every non-zero of rows < i has
    column < i and every non-zero of rows >= i has column >= i."""
    device = rowptr.device
    if n_rows == 0:
        return torch.zeros(1, dtype=torch.int64, device=device)
    counts = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(n_rows, device=device), counts)
    col64 = col.to(torch.int64)
    big = n_rows + 1
    cmax = torch.full((n_rows,), -1, dtype=torch.int64, device=device)
    cmin = torch.full((n_rows,), big, dtype=torch.int64, device=device)
    if col64.numel():
        cmax = cmax.scatter_reduce(0, rows, col64, reduce="amax", include_self=True)
        cmin = cmin.scatter_reduce(0, rows, col64, reduce="amin", include_self=True)
    idx = torch.arange(n_rows, device=device)
    cmax = torch.maximum(cmax, idx)  # a row without entries still occupies its own diagonal slot
    cmin = torch.minimum(cmin, idx)
    pre = torch.cummax(cmax, 0).values  # max column over rows <= i
    suf = torch.flip(torch.cummin(torch.flip(cmin, [0]), 0).values, [0])  # min column over rows >= i
    # boundary at i (1 <= i < n): pre[i-1] < i and suf[i] >= i
    ok = (pre[:-1] < idx[1:]) & (suf[1:] >= idx[1:])
    inner = idx[1:][ok]
    return torch.cat([torch.zeros(1, dtype=torch.int64, device=device), inner.to(torch.int64),
                      torch.tensor([n_rows], dtype=torch.int64, device=device)])


def make_tiles(ptr, max_rows):
    """Pack consecutive diagonal blocks into row tiles of at most `max_rows` rows (window == the tile's
    own row range, so every column of a block-diagonal batch is a window hit).  Blocks larger than
    `max_rows` are cut into `max_rows`-row pieces whose window is the piece itself (off-window columns
    are fetched from global memory by the kernel)."""
    ptr = np.asarray(ptr, dtype=np.int64)
    n = int(ptr[-1])
    tiles = []
    b, nb = 0, len(ptr) - 1
    while b < nb:
        start = int(ptr[b])
        size = int(ptr[b + 1]) - start
        if size > max_rows:
            for s in range(start, start + size, max_rows):
                e = min(s + max_rows, start + size)
                tiles.append((s, e, s, e - s))
            b += 1
            continue
        # furthest block end within start + max_rows
        e_idx = int(np.searchsorted(ptr, start + max_rows, side="right")) - 1
        e_idx = max(e_idx, b + 1)
        end = int(ptr[e_idx])
        tiles.append((start, end, start, end - start))
        b = e_idx
    out = np.zeros(len(tiles), dtype=TILE_DTYPE)
    if tiles:
        arr = np.asarray(tiles, dtype=np.int32)
        out["row_begin"], out["row_end"], out["win_begin"], out["win_rows"] = arr[:, 0], arr[:, 1], arr[:, 2], arr[:, 3]
    assert n == 0 or (out["row_begin"][0] == 0 and out["row_end"][-1] == n)
    return out


def _csr_from_coo(row, col, n_rows):
    """Sort COO by (row, col) -> (rowptr int32[n+1], col int32[nnz], perm)."""
    key = row.to(torch.int64) * max(n_rows, 1) + col.to(torch.int64)
    perm = torch.argsort(key, stable=True)
    counts = torch.bincount(row.to(torch.int64), minlength=n_rows)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=row.device)
    rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr.to(torch.int32), col[perm].to(torch.int32).contiguous(), perm


class CSRGraph:
    """Device CSR of one static batch: rows = target nodes, columns = source nodes.

    Attributes (torch tensors on `device`): rowptr, col (int32), val (float32, per-mode), and the
    transposed CSR rowptr_t, col_t, val_t used by the backward pass; tiles / tiles_t (int32 [T,4]).
    """

    def __init__(self, edge_index, num_nodes, mode="gcn", ptr=None, lds_rows=None):
        """mode: 'gcn'  -> add_remaining_self_loops + D^-1/2 (A+I) D^-1/2      (GCNConv, APPNP)
                 'sum'  -> plain adjacency, value 1 per edge                  (GINConv aggregation)
                 'mean' -> plain adjacency, value 1/in_degree(target)         (SAGEConv aggregation)
        """
        assert edge_index.dim() == 2 and edge_index.shape[0] == 2
        device = edge_index.device
        self.device, self.n, self.mode = device, int(num_nodes), mode
        src, dst = edge_index[0].to(torch.int64), edge_index[1].to(torch.int64)
        if mode == "gcn":
            keep = src != dst
            loop = torch.arange(self.n, device=device, dtype=torch.int64)
            src, dst = torch.cat([src[keep], loop]), torch.cat([dst[keep], loop])
        self.rowptr, self.col, _ = _csr_from_coo(dst, src, self.n)
        self.nnz = int(self.col.numel())
        # transpose: re-sort the forward entries by (col, row); perm_t carries values across
        counts = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64)
        rows_f = torch.repeat_interleave(torch.arange(self.n, device=device), counts)
        self.rowptr_t, self.col_t, self._perm_t = _csr_from_coo(self.col.to(torch.int64), rows_f, self.n)
        self.val = self.val_t = self.dinv = None
        self._ptr, self._lds_rows = ptr, lds_rows
        self.tiles = self.tiles_t = None
        self.n_tiles = 0
        if device.type == "cuda":
            self.finalize()

    # -- device-only part: values through the HIP library, tiles sized by the kernel's LDS window --
    def finalize(self, H_hint=512):
        L = _lib.lib()
        dev = self.device
        _lib.require_cuda(self.rowptr)
        st = _lib.stream_ptr(dev)
        if self.mode == "gcn":
            self.val = torch.empty(self.nnz, dtype=torch.float32, device=dev)
            self.dinv = torch.empty(self.n, dtype=torch.float32, device=dev)
            _lib.check(L.fitgnn_gcn_norm_csr_f32(_lib.dptr(self.rowptr), _lib.dptr(self.col), None, _lib.dptr(self.val),
                                                 _lib.dptr(self.dinv), self.n, st), "gcn_norm")
        elif self.mode == "sum":
            self.val = torch.ones(self.nnz, dtype=torch.float32, device=dev)
        elif self.mode == "mean":
            deg = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.float32).clamp(min=1.0)
            rows = torch.repeat_interleave(torch.arange(self.n, device=dev), (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64))
            self.val = (1.0 / deg)[rows].contiguous()
        else:
            raise ValueError(self.mode)
        self.val_t = self.val[self._perm_t].contiguous()
        cap = self._lds_rows or int(L.fitgnn_spmm_max_window_rows(H_hint))
        ptr = self._ptr if self._ptr is not None else block_boundaries(self.rowptr, self.col, self.n)
        ptr_np = ptr.detach().cpu().numpy() if torch.is_tensor(ptr) else np.asarray(ptr)
        tiles = make_tiles(ptr_np, cap)
        self.ptr = ptr_np
        self.tiles = torch.from_numpy(tiles.view(np.int32).reshape(-1, 4)).to(dev)
        self.tiles_t = self.tiles  # the transposed pattern has the same diagonal blocks
        self.n_tiles = int(self.tiles.shape[0])
        return self


# ---------------------------------------------------------------------------------------------
# cache: one CSRGraph per (edge_index storage, version, num_nodes, mode) -- static batches hit it
# ---------------------------------------------------------------------------------------------
_CACHE = {}


def csr_for(edge_index, num_nodes, mode="gcn"):
    key = (edge_index.data_ptr(), int(edge_index._version), tuple(edge_index.shape), int(num_nodes), mode,
           str(edge_index.device))
    hit = _CACHE.get(key)
    if hit is not None and hit[0]() is not None:
        return hit[1]
    g = CSRGraph(edge_index, num_nodes, mode=mode)
    if len(_CACHE) > 4096:
        _CACHE.clear()
    _CACHE[key] = (weakref.ref(edge_index), g)
    return g


def register(edge_index, graph, mode="gcn"):
    """Pre-seed the cache with a CSRGraph built with known block boundaries (SubgraphBatch does this)."""
    key = (edge_index.data_ptr(), int(edge_index._version), tuple(edge_index.shape), int(graph.n), mode,
           str(edge_index.device))
    _CACHE[key] = (weakref.ref(edge_index), graph)
    return graph
