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
TILE_INTS = 8  # sizeof(fitgnn_tile_t) / 4: the four fields above + nnz_begin, nnz_end, reserved[2]


def arrange_tiles_for_xcds(tiles8, n_xcd=8, work=None):
    """Lay the tile table out for the kernel's block -> tile mapping (array position p runs on XCD p % 8):
    every XCD gets a CONTIGUOUS range of the batch (neighbouring tiles share L2 lines) holding an equal share of
    the work (rows written + operand rows staged), interleaved as out[j*8 + k] = range_k[j]; shorter ranges are
    padded with empty tiles (row_begin == row_end), which the kernel skips.  Also used for fitgnn_block_t records (same
    size, rows in the first two fields; work = rows)."""
    t = np.ascontiguousarray(tiles8, dtype=np.int32).reshape(-1, TILE_INTS)
    T = t.shape[0]
    if T == 0:
        return t
    if work is None:
        work = (t[:, 1] - t[:, 0]).astype(np.int64) + t[:, 3].astype(np.int64)
    cum = np.concatenate([[0], np.cumsum(work)])
    bounds = np.searchsorted(cum, np.linspace(0, cum[-1], n_xcd + 1), side="left")
    bounds[0], bounds[-1] = 0, T
    bounds = np.maximum.accumulate(bounds)
    L = int(np.max(np.diff(bounds)))
    out = np.zeros((L * n_xcd, TILE_INTS), dtype=np.int32)
    for k in range(n_xcd):
        seg = t[bounds[k]:bounds[k + 1]]
        out[k:k + len(seg) * n_xcd:n_xcd] = seg
    return out


def tiles_to_device(tiles, rowptr, rowptr_host=None):
    """Host tile table (TILE_DTYPE) -> device int32 [T, 8] fitgnn_tile_t array with nnz_begin/nnz_end filled
    from the row pointers (rowptr_host: their host copy when the caller already holds one -- the table is then finished on the
    host and copied once)."""
    dev = rowptr.device
    t4 = np.ascontiguousarray(tiles).view(np.int32).reshape(-1, 4)
    rp = rowptr_host if rowptr_host is not None else rowptr.detach().cpu().numpy()
    out = np.zeros((t4.shape[0], TILE_INTS), dtype=np.int32)
    out[:, :4] = t4
    if t4.shape[0]:
        out[:, 4] = rp[t4[:, 0]]
        out[:, 5] = rp[t4[:, 1]]
    return torch.from_numpy(arrange_tiles_for_xcds(out)).to(dev)


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


def make_tiles(ptr, max_rows, whole=True):
    """Pack consecutive diagonal blocks into row tiles of at most `max_rows` rows (window == the tile's
    own row range, so every column of a block-diagonal batch is a window hit).  Blocks larger than
    `max_rows` are cut into `max_rows`-row pieces whose window is the piece itself (off-window columns
    are fetched from global memory by the kernel).  Runs in the library's host code (fitgnn_make_tiles_host: the S-products
    union has 165 000 stars and 515 000 tiles -- 0.1 s as a Python loop); make_tiles_py is the same packing in Python."""
    ptr = np.ascontiguousarray(ptr, dtype=np.int64)
    nb = len(ptr) - 1
    n = int(ptr[-1]) if nb >= 0 and len(ptr) else 0
    cap = nb + n // max(int(max_rows), 1) + 2
    buf = np.zeros((cap, 4), dtype=np.int32)
    nt = ctypes.c_int64(0)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    _lib.check(_lib.lib().fitgnn_make_tiles_host(vp(ptr), nb, int(max_rows), vp(buf), cap, ctypes.byref(nt)), "fitgnn_make_tiles_host")
    out = np.zeros(nt.value, dtype=TILE_DTYPE)
    if nt.value:
        arr = buf[: nt.value]
        out["row_begin"], out["row_end"], out["win_begin"], out["win_rows"] = arr[:, 0], arr[:, 1], arr[:, 2], arr[:, 3]
    assert (not whole) or n == 0 or (out["row_begin"][0] == 0 and out["row_end"][-1] == n)
    return out


def make_tiles_py(ptr, max_rows, whole=True):
    """make_tiles as a Python loop (the reference the host code is tested against)."""
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
    assert (not whole) or n == 0 or (out["row_begin"][0] == 0 and out["row_end"][-1] == n)
    return out


BLOCK_INTS = 8     # sizeof(fitgnn_block_t) / 4
LONG_ROW = 16      # rows with more non-zeros are a block's "long rows" (spmm.hip kLongRow)


def stream_ranges(ptr, n_rows, device, want=8192, min_rows=64):
    """(seg_ptr int32 [n_seg + 1], range_seg int32 [n_ranges + 1]) for fitgnn_spmm_csr_stream_f32: the segments `ptr` (row
    boundaries, ptr[0] = 0, ptr[-1] = n_rows) and, for each of ~`want` waves per column slab, its run of WHOLE segments with about
    the same number of rows (at least min_rows): a wave streams its rows in order, so equal rows = equal bytes."""
    ptr = np.asarray(ptr, dtype=np.int64)
    if len(ptr) < 2 or ptr[0] != 0 or ptr[-1] != n_rows:
        raise ValueError("segment boundaries must run from 0 to n_rows")
    ptr = np.unique(ptr)   # empty segments carry no rows
    n_seg = len(ptr) - 1
    n_ranges = int(max(1, min(want, n_rows // max(min_rows, 1), n_seg)))
    targets = (np.arange(n_ranges + 1, dtype=np.int64) * n_rows) // n_ranges
    rs = np.searchsorted(ptr, targets, side="left")
    rs[0], rs[-1] = 0, n_seg
    rs = np.maximum.accumulate(rs)
    return (torch.from_numpy(ptr.astype(np.int32)).to(device), torch.from_numpy(rs.astype(np.int32)).to(device))


def split_blocks(ptr, rowptr, cap, limit=None):
    """Diagonal blocks of cap < rows <= limit -> fitgnn_block_t records + their long rows (the whole-subgraph kernel: one
    workgroup per block and column slab walks it in cap-row pieces); everything else -> tiles (consecutive blocks packed,
    never across a whole-subgraph block; a block beyond `limit` is cut into cap-row tiles as before).  limit bounds the
    longest-running workgroup: by default rows / 2048 (a 256-CU chip then holds >= 8 such blocks per CU), at least 4 pieces.
    ptr: block row offsets (numpy); rowptr: device/host int tensor.
    Returns (tiles TILE_DTYPE array, blocks int32 [NB, 8], long_rows int32 [NL]) as numpy.
    Runs in the library's host code (fitgnn_split_blocks_host); split_blocks_py is the same in NumPy / Python."""
    ptr = np.ascontiguousarray(ptr, dtype=np.int64)
    rp = np.ascontiguousarray(rowptr.detach().cpu().numpy() if torch.is_tensor(rowptr) else rowptr, dtype=np.int32)   # (no copy for a host int32 array)
    nb = len(ptr) - 1
    n = int(ptr[-1])
    if limit is None:
        limit = max(4 * cap, n // 2048)
    t_cap = nb + n // max(int(cap), 1) + 2
    tiles4 = np.zeros((t_cap, 4), dtype=np.int32)
    blocks = np.zeros((max(nb, 1), BLOCK_INTS), dtype=np.int32)
    longs = np.zeros(max(n, 1), dtype=np.int32)
    nt, nl, nlong = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    _lib.check(_lib.lib().fitgnn_split_blocks_host(vp(ptr), nb, vp(rp), int(cap), int(limit), LONG_ROW, vp(tiles4), t_cap, ctypes.byref(nt),
                                                   vp(blocks), ctypes.byref(nl), vp(longs), len(longs), ctypes.byref(nlong)),
               "fitgnn_split_blocks_host")
    small = np.zeros(nt.value, dtype=TILE_DTYPE)
    if nt.value:
        arr = tiles4[: nt.value]
        small["row_begin"], small["row_end"], small["win_begin"], small["win_rows"] = arr[:, 0], arr[:, 1], arr[:, 2], arr[:, 3]
    blocks = blocks[: nl.value].copy()
    long_rows = longs[: nlong.value].copy()
    if nl.value:
        # row order, an equal share of the rows per XCD (position p runs on XCD p % 8): the stars of one subgraph run on one
        # XCD at about the same time, so what they still gather from each other (leaf -- leaf edges across stars: 12 % of the
        # entries of S-products) has a chance of being in that XCD's L2
        blocks = arrange_tiles_for_xcds(blocks, work=(blocks[:, 1] - blocks[:, 0]).astype(np.int64))
    return small, blocks, long_rows


def split_blocks_py(ptr, rowptr, cap, limit=None):
    """split_blocks in NumPy / Python (the reference the host code is tested against)."""
    ptr = np.asarray(ptr, dtype=np.int64)
    rp = rowptr.detach().cpu().numpy().astype(np.int64)
    size = np.diff(ptr)
    if limit is None:
        limit = max(4 * cap, int(ptr[-1]) // 2048)
    is_large = (size > cap) & (size <= limit)
    large = np.nonzero(is_large)[0]
    # tiles over the maximal runs of consecutive other blocks
    tiles = []
    nb = len(size)
    b = 0
    while b < nb:
        if is_large[b]:
            b += 1
            continue
        e = b
        while e < nb and not is_large[e]:
            e += 1
        tiles.append(make_tiles_py(ptr[b:e + 1], cap, whole=False) if e > b else None)
        b = e
    small = np.concatenate([t for t in tiles if t is not None and len(t)]) if any(t is not None and len(t) for t in tiles) else np.zeros(0, dtype=TILE_DTYPE)
    blocks = np.zeros((len(large), BLOCK_INTS), dtype=np.int32)
    long_rows = np.zeros(0, dtype=np.int32)
    if len(large):
        deg = np.diff(rp)
        r0, r1 = ptr[large], ptr[large + 1]
        blocks[:, 0], blocks[:, 1] = r0, r1
        blocks[:, 2], blocks[:, 3] = rp[r0], rp[r1]
        block_of_row = np.repeat(np.arange(len(large)), r1 - r0)
        rows = np.concatenate([np.arange(a, b_) for a, b_ in zip(r0, r1)]) if len(large) < 4096 else \
            (np.repeat(r0, r1 - r0) + (np.arange(int((r1 - r0).sum())) - np.repeat(np.cumsum(r1 - r0) - (r1 - r0), r1 - r0)))
        is_long = deg[rows] > LONG_ROW
        lr, lb = rows[is_long], block_of_row[is_long]
        cnt = np.bincount(lb, minlength=len(large))
        off = np.cumsum(cnt) - cnt
        blocks[:, 4], blocks[:, 5] = off, cnt
        long_rows = lr.astype(np.int32)
        # row order, an equal share of the rows per XCD (position p runs on XCD p % 8): the stars of one subgraph run on one
        # XCD at about the same time, so what they still gather from each other (leaf -- leaf edges across stars: 12 % of the
        # entries of S-products) has a chance of being in that XCD's L2
        blocks = arrange_tiles_for_xcds(blocks, work=(r1 - r0).astype(np.int64))
    return small, blocks, long_rows


def _csr_from_coo(row, col, n_rows):
    """Sort COO by (row, col) -> (rowptr int32[n+1], col int32[nnz], perm)."""
    key = row.to(torch.int64) * max(n_rows, 1) + col.to(torch.int64)
    perm = torch.argsort(key, stable=True)
    counts = torch.bincount(row.to(torch.int64), minlength=n_rows)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=row.device)
    rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr.to(torch.int32), col[perm].to(torch.int32).contiguous(), perm


class _Side:
    """One orientation of the pattern (forward or transposed) with its planned tiles.
    `tiles` covers every row.  `small_tiles` / `blocks` / `long_rows` (contiguous windows only) cover the rows once more, split
    by block size: tiles that pack the diagonal blocks of at most a window, and the larger blocks as fitgnn_block_t records
    for the whole-subgraph kernel (fitgnn_spmm_csr_blocks_f32), which reads each of their operand rows once."""
    __slots__ = ("rowptr", "col", "val", "tiles", "lcol", "win_cols", "n_tiles", "small_tiles", "blocks", "long_rows", "xcol")

    def __init__(self, rowptr, col):
        self.rowptr, self.col = rowptr, col
        self.xcol = None   # (index data_ptr, index[col]): the operand-table row of every entry, cached by ops.spmm_graph
        self.val = self.tiles = self.lcol = self.win_cols = None
        self.small_tiles = self.blocks = self.long_rows = None
        self.n_tiles = 0


def plan_tiles(rowptr, col, n_rows, max_rows, max_window, block_ptr=None):
    """fitgnn_plan_tiles_host on host copies of the pattern -> (tiles int32 [T,8], win_cols, lcol) numpy."""
    L = _lib.lib()
    rp = np.ascontiguousarray(rowptr.detach().cpu().numpy(), dtype=np.int32)
    cc = np.ascontiguousarray(col.detach().cpu().numpy(), dtype=np.int32)
    nnz = int(rp[-1]) if n_rows else 0
    tiles = np.zeros((max(n_rows, 1), TILE_INTS), dtype=np.int32)
    win = np.zeros(max(nnz, 1), dtype=np.int32)
    lcol = np.zeros(max(nnz, 1), dtype=np.int32)
    nt, nw = ctypes.c_int32(0), ctypes.c_int32(0)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    bp, nb = None, 0
    if block_ptr is not None:
        bp = np.ascontiguousarray(block_ptr.detach().cpu().numpy() if torch.is_tensor(block_ptr) else block_ptr, dtype=np.int64)
        nb = len(bp) - 1
    _lib.check(L.fitgnn_plan_tiles_host(vp(rp), vp(cc), n_rows, n_rows, None if bp is None else vp(bp), nb, int(max_rows),
                                        int(max_window), vp(tiles), ctypes.byref(nt), vp(win), ctypes.byref(nw), vp(lcol)),
               "plan_tiles")
    return tiles[: nt.value].copy(), win[: max(nw.value, 1)].copy(), lcol[: max(nnz, 1)].copy()


class CSRGraph:
    """Device CSR of one static batch: rows = target nodes, columns = source nodes.

    `f` / `t`: the forward and the transposed orientation (the backward pass multiplies by A^T), each with
    rowptr, col (int32), val (float32) and its planned tiles (tiles, lcol, win_cols).  Legacy aliases:
    rowptr/col/val/tiles and rowptr_t/col_t/val_t/tiles_t.
    """

    def __init__(self, edge_index, num_nodes, mode="gcn", ptr=None, lds_rows=None, planned=False, gather=False, split_large=True,
                 block_limit=None):
        """mode: 'gcn'  -> add_remaining_self_loops + D^-1/2 (A+I) D^-1/2      (GCNConv, APPNP)
                 'sum'  -> plain adjacency, value 1 per edge                  (GINConv aggregation)
                 'mean' -> plain adjacency, value 1/in_degree(target)         (SAGEConv aggregation)
                 'gat'  -> add_remaining_self_loops, values = attention        (GATConv)
        planned: tiles from fitgnn_plan_tiles_host (column-set windows for blocks larger than the window);
        False (default; measured faster on PubMed-like batches, profiles/): contiguous windows packed from the
        diagonal-block boundaries `ptr` (detected when not given).  gather: use the direct-gather kernel.
        """
        assert edge_index.dim() == 2 and edge_index.shape[0] == 2
        device = edge_index.device
        self.device, self.n, self.mode = device, int(num_nodes), mode
        self.planned, self.gather, self.split_large, self.block_limit = planned, gather, split_large, block_limit
        self.split_min_rows = 600_000   # x 2 KiB rows = 1.2 GB at hidden 512: far beyond the 256-MiB Infinity Cache (an eighth of the
                                        # S-products union, one rank of 8, is 1.03 M rows and must take the same kernels as the whole)
        # tiles from make_tiles: contiguous windows covering their own rows -> the folded backward kernel applies
        self.fold_ok = (not planned) and (not gather)
        src, dst = edge_index[0].to(torch.int64), edge_index[1].to(torch.int64)
        if mode in ("gcn", "gat"):  # add_remaining_self_loops: exactly one self loop per node
            keep = src != dst
            loop = torch.arange(self.n, device=device, dtype=torch.int64)
            src, dst = torch.cat([src[keep], loop]), torch.cat([dst[keep], loop])
        rowptr, col, _ = _csr_from_coo(dst, src, self.n)
        self.nnz = int(col.numel())
        # transpose: re-sort the forward entries by (col, row); perm_t carries values across
        counts = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
        rows_f = torch.repeat_interleave(torch.arange(self.n, device=device), counts)
        rowptr_t, col_t, self._perm_t = _csr_from_coo(col.to(torch.int64), rows_f, self.n)
        self.f, self.t = _Side(rowptr, col), _Side(rowptr_t, col_t)
        self.dinv = None
        self.seg = self.range_seg = None   # fitgnn_spmm_csr_stream_f32: segment starts, segments per wave (set with the blocks)
        self._ptr, self._lds_rows = ptr, lds_rows
        self.window_rows = 0
        if device.type == "cuda":
            self.finalize()

    # legacy aliases
    rowptr = property(lambda s: s.f.rowptr)
    col = property(lambda s: s.f.col)
    val = property(lambda s: s.f.val)
    tiles = property(lambda s: s.f.tiles)
    n_tiles = property(lambda s: s.f.n_tiles)
    rowptr_t = property(lambda s: s.t.rowptr)
    col_t = property(lambda s: s.t.col)
    val_t = property(lambda s: s.t.val)
    tiles_t = property(lambda s: s.t.tiles)

    # -- device-only part: values through the HIP library, tiles sized by the kernel's LDS window --
    def finalize(self):
        L = _lib.lib()
        dev = self.device
        f, t = self.f, self.t
        _lib.require_cuda(f.rowptr)
        st = _lib.stream_ptr(dev)
        if self.mode == "gcn":
            f.val = torch.empty(self.nnz, dtype=torch.float32, device=dev)
            self.dinv = torch.empty(self.n, dtype=torch.float32, device=dev)
            _lib.check(L.fitgnn_gcn_norm_csr_f32(_lib.dptr(f.rowptr), _lib.dptr(f.col), None, _lib.dptr(f.val),
                                                 _lib.dptr(self.dinv), self.n, st), "gcn_norm")
        elif self.mode in ("sum", "gat"):  # gat: values are replaced by attention weights per forward
            f.val = torch.ones(self.nnz, dtype=torch.float32, device=dev)
        elif self.mode == "mean":
            deg = (f.rowptr[1:] - f.rowptr[:-1]).to(torch.float32).clamp(min=1.0)
            rows = torch.repeat_interleave(torch.arange(self.n, device=dev), (f.rowptr[1:] - f.rowptr[:-1]).to(torch.int64))
            f.val = (1.0 / deg)[rows].contiguous()
        else:
            raise ValueError(self.mode)
        t.val = f.val[self._perm_t].contiguous()
        cap = self._lds_rows or int(L.fitgnn_spmm_default_window_rows())
        self.window_rows = cap
        ptr = self._ptr if self._ptr is not None else block_boundaries(f.rowptr, f.col, self.n)
        ptr_np = ptr.detach().cpu().numpy() if torch.is_tensor(ptr) else np.asarray(ptr)
        self.ptr = ptr_np
        if self.planned:
            for side in (f, t):  # the transposed pattern has the same diagonal blocks
                tiles, win, lcol = plan_tiles(side.rowptr, side.col, self.n, cap, cap, block_ptr=ptr_np if len(ptr_np) > 2 else None)
                side.tiles = torch.from_numpy(arrange_tiles_for_xcds(tiles)).to(dev)
                side.win_cols = torch.from_numpy(win).to(dev)
                side.lcol = torch.from_numpy(lcol).to(dev)
                side.n_tiles = int(tiles.shape[0])
        else:
            tiles = make_tiles(ptr_np, cap)
            has_large = bool(len(ptr_np) > 1 and np.max(np.diff(ptr_np)) > cap)
            for side in (f, t):  # same diagonal blocks, each side its own CSR offsets
                rp_host = np.ascontiguousarray(side.rowptr.detach().cpu().numpy(), dtype=np.int32)   # one copy per side for all tables below
                side.tiles = tiles_to_device(tiles, side.rowptr, rp_host)
                side.n_tiles = int(tiles.shape[0])
                if has_large and self.split_large and not self.gather and self.nnz > 0:
                    small, blocks, long_rows = split_blocks(ptr_np, rp_host, cap, self.block_limit)
                    # the whole-subgraph kernel saves re-reads from HBM: it pays when a good part of the batch sits in such runs
                    # and the operand is beyond the Infinity Cache (a launch on the 90 k-row PubMed union lasts 80 us: a second
                    # launch and a few long-running workgroups cost more than the re-reads, which stay on-die there)
                    in_blocks = int((blocks[:, 1] - blocks[:, 0]).sum()) if len(blocks) else 0
                    if self.block_limit is None and (in_blocks * 5 < self.n * 3 or self.n < self.split_min_rows):
                        continue   # measured: S-products (stars of ~50 rows) 8.27 -> 7.50 ms with the fused epilogue; S-physics
                                   # (stars of ~14 rows, 40 % of the rows in such runs) 399 -> 441 us: tiles stay the default there
                    side.small_tiles = tiles_to_device(small, side.rowptr, rp_host) if len(small) else torch.zeros((0, TILE_INTS), dtype=torch.int32, device=dev)
                    side.blocks = torch.from_numpy(blocks).to(dev)
                    side.long_rows = torch.from_numpy(long_rows if len(long_rows) else np.zeros(1, dtype=np.int32)).to(dev)
                    if self.seg is None:   # the segment-streaming kernel's view of the same runs (both sides share it)
                        self.seg, self.range_seg = stream_ranges(ptr_np, self.n, dev)
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


# ---------------------------------------------------------------------------------------------
# row subsets: the rows of A_hat whose outputs are consumed (the last layer only needs the clusters' own nodes)
# ---------------------------------------------------------------------------------------------
def make_tiles_pair(out_ptr, win_ptr, max_rows):
    """Tiles for a rectangular block-diagonal pattern: block b owns output rows out_ptr[b]:out_ptr[b+1] and reads operand
    rows win_ptr[b]:win_ptr[b+1].  Consecutive blocks are packed while both ranges stay within max_rows; a block that
    exceeds it is cut along its output rows, its window clamped by the kernel (rows beyond it are fetched directly)."""
    out_ptr, win_ptr = np.asarray(out_ptr, dtype=np.int64), np.asarray(win_ptr, dtype=np.int64)
    nb = len(out_ptr) - 1
    tiles, b = [], 0
    while b < nb:
        o0, w0 = int(out_ptr[b]), int(win_ptr[b])
        if out_ptr[b + 1] - o0 > max_rows or win_ptr[b + 1] - w0 > max_rows:
            for s in range(o0, int(out_ptr[b + 1]), max_rows):
                tiles.append((s, min(s + max_rows, int(out_ptr[b + 1])), w0, int(win_ptr[b + 1]) - w0))
            b += 1
            continue
        e = b + 1
        while e < nb and out_ptr[e + 1] - o0 <= max_rows and win_ptr[e + 1] - w0 <= max_rows:
            e += 1
        if out_ptr[e] > o0:
            tiles.append((o0, int(out_ptr[e]), w0, int(win_ptr[e]) - w0))
        b = e
    out = np.zeros(len(tiles), dtype=TILE_DTYPE)
    if tiles:
        arr = np.asarray(tiles, dtype=np.int32)
        out["row_begin"], out["row_end"], out["win_begin"], out["win_rows"] = arr[:, 0], arr[:, 1], arr[:, 2], arr[:, 3]
    return out


class RowSubset:
    """A_sub = A_hat[rows, :] of a CSRGraph (gcn mode) and its transpose, each with tiles, for a layer whose output is
    only consumed on `rows` (sorted union-row ids; e.g. the clusters' own nodes: the outputs of the last GCN layer on
    extra nodes never reach the loss, run.py:193-204 keeps out[mask] only).  `f`: [m x R] maps operand rows (all union
    rows) to the m kept rows; `t`: [R x m] is its adjoint for the backward pass."""

    def __init__(self, g, rows):
        dev = g.device
        rows = rows.to(dev).long()
        self.rows, self.m, self.n, self.window_rows = rows, int(rows.numel()), g.n, g.window_rows
        rp = g.f.rowptr.long()
        cnt = rp[rows + 1] - rp[rows]
        rowptr = torch.zeros(self.m + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(cnt, 0)
        src = torch.repeat_interleave(rp[rows], cnt) + (torch.arange(int(rowptr[-1]), device=dev) - torch.repeat_interleave(rowptr[:-1], cnt))
        col, val = g.f.col[src].contiguous(), g.f.val[src].contiguous()
        self.f = _Side(rowptr.to(torch.int32), col)
        self.f.val = val
        kept_row = torch.repeat_interleave(torch.arange(self.m, device=dev), cnt)        # compact row of every entry
        rowptr_t, col_t, perm = _csr_from_coo(col.long(), kept_row, g.n)
        self.t = _Side(rowptr_t, col_t)
        self.t.val = val[perm].contiguous()
        ptr = np.asarray(g.ptr, dtype=np.int64)                                          # union-row block boundaries
        kp = np.searchsorted(rows.cpu().numpy(), ptr, side="left")                       # kept-row block boundaries
        self.f.tiles = tiles_to_device(make_tiles_pair(kp, ptr, g.window_rows), self.f.rowptr)
        self.t.tiles = tiles_to_device(make_tiles_pair(ptr, kp, g.window_rows), self.t.rowptr)
        self.f.n_tiles, self.t.n_tiles = int(self.f.tiles.shape[0]), int(self.t.tiles.shape[0])
