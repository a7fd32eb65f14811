"""GPU tier: CSR SpMM / gcn_norm / fused epilogue kernels through the C ABI vs the CPU oracle.
Tolerance: 1e-4 relative fp32 (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RTOL = 1e-4


@pytest.fixture(scope="module")
def mods():
    assert torch.cuda.is_available()
    from fitgnn_amd import _lib, csr, ops
    from oracle import coarsen_oracle as orc
    from oracle import gnn_oracle as gorc

    return _lib, csr, ops, orc, gorc


def block_graph(sizes, seed, p):
    rng = np.random.default_rng(seed)
    src, dst, off = [], [], 0
    for s in sizes:
        if s > 1:
            m = max(1, int(p * s * (s - 1) / 2))
            a = rng.integers(0, s, size=m); b = rng.integers(0, s, size=m)
            k = a != b
            src += (off + a[k]).tolist() + (off + b[k]).tolist()
            dst += (off + b[k]).tolist() + (off + a[k]).tolist()
        off += s
    ei = np.unique(np.array([src, dst]), axis=1) if src else np.zeros((2, 0), dtype=np.int64)
    return torch.tensor(ei, dtype=torch.long), off


def rel_err(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-20))


# every width with the default window; the window sweep on two widths only
_HW = [(H, None) for H in (512, 256, 64, 100, 7, 1, 516)] + [(H, w) for H in (512, 7) for w in (8, 64, 96)]


@pytest.mark.parametrize("H,window", _HW)
@pytest.mark.parametrize("sizes", [[3, 9, 1, 30, 64, 2, 2, 5] * 6, [200, 3, 90], [1] * 70])
@pytest.mark.parametrize("planned", [True, False])
def test_spmm_matches_oracle(mods, H, sizes, window, planned):
    _lib, csr, ops, orc, gorc = mods
    ei, n = block_graph(sizes, seed=len(sizes) + H, p=0.3)
    g = csr.CSRGraph(ei.cuda(), n, mode="gcn", lds_rows=window, planned=planned)
    X = torch.randn(n, H)
    Y = ops.spmm_graph(g, X.cuda()).cpu()
    ref = torch.from_numpy(orc.spmm_csr_f32(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.val.cpu().numpy(), X.numpy()))
    assert rel_err(Y, ref) < RTOL
    # the direct-gather variant computes the same thing
    Yg = ops.spmm_graph(g, X.cuda(), epilogue=_lib.SPMM_GATHER).cpu()
    assert rel_err(Yg, ref) < RTOL
    # transposed CSR == transpose of the dense matrix
    Yt = ops.spmm_graph(g, X.cuda(), transposed=True).cpu()
    dense = torch.zeros(n, n)
    rows = torch.repeat_interleave(torch.arange(n), (g.rowptr[1:] - g.rowptr[:-1]).long().cpu())
    dense.index_put_((rows, g.col.long().cpu()), g.val.cpu(), accumulate=True)
    assert rel_err(Yt, dense.t() @ X) < RTOL


def test_gcn_norm_matches_pyg_formula(mods):
    _lib, csr, ops, orc, gorc = mods
    ei, n = block_graph([40, 7, 1, 100], seed=9, p=0.2)
    ei = torch.cat([ei, torch.tensor([[3, 3], [3, 3]])], 1)  # duplicated explicit self loop on node 3
    g = csr.CSRGraph(ei.cuda(), n, mode="gcn")
    row, col, w = gorc.gcn_norm(ei, n)
    dense_ref = torch.zeros(n, n).index_put_((col, row), w, accumulate=True)  # [target, source]
    rows = torch.repeat_interleave(torch.arange(n), (g.rowptr[1:] - g.rowptr[:-1]).long().cpu())
    dense = torch.zeros(n, n).index_put_((rows, g.col.long().cpu()), g.val.cpu(), accumulate=True)
    assert torch.allclose(dense, dense_ref, rtol=1e-6, atol=1e-7)


def test_window_misses_fall_back_to_global(mods):
    """Unstructured matrix (one block of 1000 rows): most columns are outside the 64-row LDS window."""
    _lib, csr, ops, orc, gorc = mods
    ei, n = block_graph([1000], seed=4, p=0.02)
    g = csr.CSRGraph(ei.cuda(), n, mode="gcn", planned=False)
    tl = g.tiles.cpu().numpy()
    assert int((tl[:, 1] > tl[:, 0]).sum()) == -(-1000 // g.window_rows)
    X = torch.randn(n, 512)
    Y = ops.spmm_graph(g, X.cuda()).cpu()
    ref = torch.from_numpy(orc.spmm_csr_f32(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.val.cpu().numpy(), X.numpy()))
    assert rel_err(Y, ref) < RTOL


def test_window_smaller_than_tile_is_clamped(mods):
    """Tiles built for 64-row windows but launched with a 16-row LDS window: misses go to global memory."""
    _lib, csr, ops, orc, gorc = mods
    ei, n = block_graph([40, 50, 64, 7] * 3, seed=8, p=0.3)
    g = csr.CSRGraph(ei.cuda(), n, mode="gcn", lds_rows=64, planned=False)
    X = torch.randn(n, 512)
    Y = ops.spmm_raw(g.rowptr, g.col, g.val, g.tiles, X.cuda(), n, window_rows=16).cpu()
    gp = csr.CSRGraph(ei.cuda(), n, mode="gcn", lds_rows=64, planned=True)   # planned for 64, launched with 16
    Yp = ops.spmm_raw(gp.rowptr, gp.col, gp.val, gp.tiles, X.cuda(), n, window_rows=16, lcol=gp.f.lcol,
                      win_cols=gp.f.win_cols).cpu()
    ref = torch.from_numpy(orc.spmm_csr_f32(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.val.cpu().numpy(), X.numpy()))
    assert rel_err(Y, ref) < RTOL
    assert rel_err(Yp, ref) < RTOL


def test_hub_subgraph_larger_than_window_is_planned_into_lds(mods):
    """Star subgraphs bigger than the window: the planner's column-set windows keep the hub row in LDS."""
    _lib, csr, ops, orc, gorc = mods
    src, dst, off = [], [], 0
    for leaves in (150, 40, 7, 90):
        for l in range(1, leaves + 1):
            src += [off, off + l]; dst += [off + l, off]
        off += leaves + 1
    ei = torch.tensor([src, dst], dtype=torch.long)
    g = csr.CSRGraph(ei.cuda(), off, mode="gcn", lds_rows=24, planned=True)
    lc = g.f.lcol.cpu().numpy()
    # every leaf row finds hub + itself in LDS; only the hub rows (wider than the window) go to global memory
    assert (lc < 0).mean() < 0.3
    X = torch.randn(off, 512)
    Y = ops.spmm_graph(g, X.cuda()).cpu()
    ref = torch.from_numpy(orc.spmm_csr_f32(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.val.cpu().numpy(), X.numpy()))
    assert rel_err(Y, ref) < RTOL
    Yt = ops.spmm_graph(g, X.cuda(), transposed=True).cpu()
    assert rel_err(Yt, ref) < RTOL  # symmetric matrix


def test_dense_tile_overflows_the_staged_csr_slice(mods):
    """A complete graph on 60 nodes has 3600 entries per tile > the staged CSR slice (32 per window row)."""
    _lib, csr, ops, orc, gorc = mods
    a, b = np.meshgrid(np.arange(60), np.arange(60))
    k = a != b
    ei = torch.tensor(np.stack([a[k], b[k]]), dtype=torch.long)
    ei = torch.cat([ei, ei + 60], 1)
    g = csr.CSRGraph(ei.cuda(), 120, mode="gcn", lds_rows=64)
    X = torch.randn(120, 256)
    Y = ops.spmm_graph(g, X.cuda()).cpu()
    ref = torch.from_numpy(orc.spmm_csr_f32(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.val.cpu().numpy(), X.numpy()))
    assert rel_err(Y, ref) < RTOL


def test_long_rows(mods):
    """A hub row with > 64 non-zeros exercises the chunked (64 pairs per pass) row loop."""
    _lib, csr, ops, orc, gorc = mods
    n = 500
    leaves = torch.arange(1, 400)
    ei = torch.cat([torch.stack([torch.zeros_like(leaves), leaves]), torch.stack([leaves, torch.zeros_like(leaves)])], 1)
    g = csr.CSRGraph(ei.cuda(), n, mode="gcn")
    X = torch.randn(n, 256)
    Y = ops.spmm_graph(g, X.cuda()).cpu()
    ref = torch.from_numpy(orc.spmm_csr_f32(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.val.cpu().numpy(), X.numpy()))
    assert rel_err(Y, ref) < RTOL


@pytest.mark.parametrize("H", [512, 33])
def test_fused_epilogue_and_its_backward(mods, H):
    _lib, csr, ops, orc, gorc = mods
    from fitgnn_amd._lib import EPI_BIAS, EPI_DROPOUT, EPI_ELU

    ei, n = block_graph([20, 5, 64, 3] * 4, seed=2, p=0.3)
    g = csr.CSRGraph(ei.cuda(), n, mode="gcn")
    torch.manual_seed(1)
    X, b = torch.randn(n, H), torch.randn(H)
    mask = (torch.rand(n, H) > 0.5).to(torch.uint8)
    base = torch.from_numpy(orc.spmm_csr_f32(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.val.cpu().numpy(), X.numpy()))
    z = base + b
    ref = torch.nn.functional.elu(z) * mask * 2.0
    out = ops.spmm_graph(g, X.cuda(), bias=b.cuda(), epilogue=EPI_BIAS | EPI_ELU | EPI_DROPOUT, p=0.5, mask=mask.cuda())
    assert rel_err(out.cpu(), ref) < RTOL
    outg = ops.spmm_graph(g, X.cuda(), bias=b.cuda(), epilogue=EPI_BIAS | EPI_ELU | EPI_DROPOUT | _lib.SPMM_GATHER, p=0.5,
                          mask=mask.cuda())
    assert rel_err(outg.cpu(), ref) < RTOL
    # backward of the epilogue vs autograd
    zz = z.clone().requires_grad_(True)
    o = torch.nn.functional.elu(zz) * mask * 2.0
    dOut = torch.randn(n, H)
    o.backward(dOut)
    dZ, db = ops.epilogue_bwd_raw(dOut.cuda(), out, EPI_ELU | EPI_DROPOUT, p=0.5, mask=mask.cuda())
    assert rel_err(dZ.cpu(), zz.grad) < RTOL
    assert rel_err(db.cpu(), zz.grad.sum(0)) < 1e-4
    # hash-based dropout: forward and backward regenerate the same pattern, keep rate ~ 1-p
    out2 = ops.spmm_graph(g, X.cuda(), bias=b.cuda(), epilogue=EPI_BIAS | EPI_ELU | EPI_DROPOUT, p=0.3, seed=1234)
    out2g = ops.spmm_graph(g, X.cuda(), bias=b.cuda(), epilogue=EPI_BIAS | EPI_ELU | EPI_DROPOUT | _lib.SPMM_GATHER, p=0.3, seed=1234)
    assert torch.equal(out2g != 0, out2 != 0)  # both variants draw the same dropout pattern
    kept = (out2 != 0).float().mean().item()
    assert abs(kept - 0.7) < 0.03
    dZ2, _ = ops.epilogue_bwd_raw(torch.ones(n, H).cuda(), out2, EPI_ELU | EPI_DROPOUT, p=0.3, seed=1234)
    assert torch.equal(dZ2 != 0, out2 != 0) or ((dZ2 != 0) ^ (out2 != 0)).float().mean().item() < 1e-3


def test_empty_and_degenerate(mods):
    _lib, csr, ops, orc, gorc = mods
    g = csr.CSRGraph(torch.zeros((2, 0), dtype=torch.long).cuda(), 5, mode="gcn")  # only self loops
    X = torch.randn(5, 8)
    Y = ops.spmm_graph(g, X.cuda()).cpu()
    assert torch.allclose(Y, X, rtol=1e-6, atol=1e-7)
    g0 = csr.CSRGraph(torch.zeros((2, 0), dtype=torch.long).cuda(), 3, mode="sum")  # no entries at all
    Y0 = ops.spmm_graph(g0, torch.randn(3, 4).cuda()).cpu()
    assert torch.count_nonzero(Y0) == 0


@pytest.mark.parametrize("H,C", [(512, 3), (512, 16), (100, 1), (33, 7)])
def test_head_backward_folded_into_epilogue_kernel(mods, H, C):
    """fitgnn_epilogue_bwd_head_f32 == epilogue backward applied to dOut = dy @ Wl."""
    _lib, csr, ops, orc, gorc = mods
    from fitgnn_amd._lib import EPI_DROPOUT, EPI_ELU

    torch.manual_seed(5)
    n = 777
    out = torch.nn.functional.elu(torch.randn(n, H))
    mask = (torch.rand(n, H) > 0.5).to(torch.uint8)
    out = out * mask * 2.0
    dy, Wl = torch.randn(n, C), torch.randn(C, H)
    dZ_ref, db_ref = ops.epilogue_bwd_raw((dy @ Wl).cuda(), out.cuda(), EPI_ELU | EPI_DROPOUT, p=0.5, mask=mask.cuda())
    dZ, db, dWl = ops.epilogue_bwd_head_raw(dy.cuda(), Wl.cuda(), out.cuda(), EPI_ELU | EPI_DROPOUT, p=0.5, mask=mask.cuda())
    assert rel_err(dZ.cpu(), dZ_ref.cpu()) < 1e-5
    assert rel_err(db.cpu(), db_ref.cpu()) < 1e-4
    assert rel_err(dWl.cpu(), (dy.double().t() @ out.double()).float()) < 1e-5


def star_blocks(sizes, centres, seed, extra=0.02):
    """Block-diagonal batch of star-shaped subgraphs: in every block the first `centres` rows link to all the others
    (--extra_node subgraphs, utils.py:235-239), plus a few random leaf -- leaf edges."""
    rng = np.random.default_rng(seed)
    src, dst, off = [], [], 0
    for s in sizes:
        c = min(centres, max(s - 1, 0))
        for h in range(c):
            leaves = np.arange(c, s)
            src += [off + h] * len(leaves) + (off + leaves).tolist()
            dst += (off + leaves).tolist() + [off + h] * len(leaves)
        m = int(extra * s * s)
        if m and s > 2:
            a, b = rng.integers(0, s, size=m), rng.integers(0, s, size=m)
            k = a != b
            src += (off + a[k]).tolist() + (off + b[k]).tolist()
            dst += (off + b[k]).tolist() + (off + a[k]).tolist()
        off += s
    return torch.tensor(np.unique(np.array([src, dst]), axis=1), dtype=torch.long), off


@pytest.mark.parametrize("H", [512, 256, 100])
@pytest.mark.parametrize("sizes,centres", [([100, 7, 17, 300, 3, 3, 64, 33, 2, 1000], 2), ([40] * 30, 1), ([500, 90], 12),
                                           ([18, 16, 17, 5, 5, 5, 2000], 3)])
def test_whole_subgraph_kernel_gives_the_tile_kernel_bits(mods, H, sizes, centres):
    """Diagonal blocks larger than the window run through fitgnn_spmm_csr_blocks_f32 (one workgroup walks the subgraph, the
    centre rows' accumulators carried across its 16-row pieces, their operand rows pinned in LDS): same bits as the tiled
    kernel, forward and transposed, with and without the fused epilogue -- blocks with 1, 2, 3 and more than 8 long rows,
    long rows beyond 64 entries, pieces with more than 256 non-zeros, small blocks packed between large ones."""
    _lib, csr, ops, orc, gorc = mods
    from fitgnn_amd._lib import EPI_BIAS, EPI_DROPOUT, EPI_ELU

    ei, n = star_blocks(sizes, centres, seed=H + len(sizes))
    ptr = np.concatenate([[0], np.cumsum(sizes)])
    g = csr.CSRGraph(ei.cuda(), n, mode="gcn", ptr=ptr)
    assert g.f.blocks is None, "a batch this small stays on tiles (the kernel pays beyond the Infinity Cache)"
    g64 = csr.CSRGraph(ei.cuda(), n, mode="gcn", ptr=ptr, block_limit=4096)   # an explicit limit forces the split
    nblk = lambda gg: int((gg.f.blocks[:, 1] > gg.f.blocks[:, 0]).sum())   # the record table is padded per XCD with empty records
    assert nblk(g64) == sum(1 for s_ in sizes if s_ > 16)
    g = csr.CSRGraph(ei.cuda(), n, mode="gcn", ptr=ptr, block_limit=64)       # blocks beyond 4 pieces are tiled again
    n_mid, in_mid = sum(1 for s_ in sizes if 16 < s_ <= 64), sum(s_ for s_ in sizes if 16 < s_ <= 64)
    assert (g.f.blocks is None and in_mid * 4 < n) or nblk(g) == n_mid
    torch.manual_seed(3)
    X = torch.randn(n, H).cuda()
    b = torch.randn(H).cuda()
    off = ops.OpConfig(split_large_blocks=False)
    # the two kernels that read every operand row once: the whole-subgraph kernel (one workgroup per run, LDS windows) and the
    # segment-streaming kernel (one wave per run of segments, no LDS; the default)
    blk, stream = ops.OpConfig(stream_kernel=False), ops.OpConfig(stream_kernel=True, profile=[])
    for transposed in (False, True):
        for kw in (dict(), dict(bias=b, epilogue=EPI_BIAS | EPI_ELU | EPI_DROPOUT, p=0.5, seed=123)):
            tiled = ops.spmm_graph(g, X, transposed=transposed, cfg=off, **kw)
            for gg in (g, g64):
                for cfg in (blk, stream):
                    assert torch.equal(ops.spmm_graph(gg, X, transposed=transposed, cfg=cfg, **kw), tiled)
    assert g64.seg is not None and len(stream.profile) >= 4   # one event pair per streamed launch
    ref = torch.from_numpy(orc.spmm_csr_f32(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.val.cpu().numpy(), X.cpu().numpy()))
    assert rel_err(ops.spmm_graph(g, X).cpu(), ref) < RTOL
    # row indirection into a de-duplicated operand table (layer 0): pattern row / column r reads Xt[xrow[r]]
    N0 = max(n // 3, 1)
    xrow = torch.randint(0, N0, (n,), dtype=torch.int32).cuda()
    Xt = torch.randn(N0, H).cuda()
    want = ops.spmm_graph(g64, Xt[xrow.long()].contiguous(), cfg=off)
    for gg in (g, g64):
        for cfg in (ops.DEFAULT, blk, stream, off):
            assert torch.equal(ops.spmm_graph(gg, Xt, xrow=xrow, cfg=cfg), want)
            assert torch.equal(ops.spmm_graph(gg, Xt, xrow=xrow, cfg=cfg, bias=b, epilogue=EPI_BIAS | EPI_ELU | EPI_DROPOUT, p=0.5, seed=5),
                               ops.spmm_graph(g64, Xt[xrow.long()].contiguous(), cfg=off, bias=b, epilogue=EPI_BIAS | EPI_ELU | EPI_DROPOUT, p=0.5, seed=5))


@pytest.mark.parametrize("H", [512, 64])
@pytest.mark.parametrize("use_mask", [False, True])
def test_backward_spmm_with_the_previous_layers_epilogue_in_its_store(mods, H, use_mask):
    """spmm_graph_dz (the derivative of dropout(ELU(.)) applied as the SpMM stores its rows, column sums per workgroup) ==
    the SpMM followed by the epilogue-backward kernel, bit for bit for dZ and to fp32 summation order for db; tiles only, the
    whole-subgraph kernel + tiles, with a compact operand behind a row indirection."""
    _lib, csr, ops, orc, gorc = mods
    from fitgnn_amd._lib import EPI_DROPOUT, EPI_ELU

    sizes = [100, 7, 17, 300, 3, 3, 64, 33, 2, 1000, 5, 5, 40]
    ei, n = star_blocks(sizes, 2, seed=H)
    ptr = np.concatenate([[0], np.cumsum(sizes)])
    torch.manual_seed(H)
    prev = torch.randn(n, H).cuda() * (torch.rand(n, H).cuda() > 0.3)        # a forward output: zeros where dropout hit
    mask = (torch.rand(n, H).cuda() > 0.5).to(torch.uint8) if use_mask else None
    flags, p, seed = EPI_ELU | EPI_DROPOUT, 0.5, 77
    for g in (csr.CSRGraph(ei.cuda(), n, mode="gcn", ptr=ptr), csr.CSRGraph(ei.cuda(), n, mode="gcn", ptr=ptr, block_limit=4096)):
        X = torch.randn(n, H).cuda()
        plain = ops.spmm_graph(g, X, transposed=True)
        want, want_db = ops.epilogue_bwd_raw(plain, prev, flags, p=p, seed=seed, mask=mask, want_db=True)
        for cfg in (ops.DEFAULT, ops.OpConfig(stream_kernel=True)):   # whole-subgraph kernel / segment-streaming kernel (where the graph is split)
            got, got_db = ops.spmm_graph_dz(g, X, prev, flags, p=p, seed=seed, mask=mask, want_db=True, cfg=cfg)
            assert torch.equal(got, want)
            assert rel_err(got_db.cpu(), want_db.cpu()) < 1e-5
        # compact operand: a third of the rows carry values, the others read a zero row
        rows = torch.randperm(n).cuda()[: n // 3].sort().values
        Xc = torch.cat([torch.randn(rows.numel(), H).cuda(), torch.zeros(ops.ZERO_ROWS, H).cuda()])
        pos = ops._compact_positions(g, rows)
        dense = torch.zeros(n, H).cuda(); dense[rows] = Xc[: rows.numel()]
        want2, _ = ops.epilogue_bwd_raw(ops.spmm_graph(g, dense, transposed=True), prev, flags, p=p, seed=seed, mask=mask, want_db=False)
        got2, none = ops.spmm_graph_dz(g, Xc, prev, flags, p=p, seed=seed, mask=mask, want_db=False, xrow=pos)
        assert none is None and torch.equal(got2, want2)


def test_many_narrow_partials_are_folded_in_two_stages(mods):
    """mm_at_b on a very tall pair with a two-column operand (GAT's h^T [da_src da_dst]): thousands of batched partial products of a
    few hundred floats, summed by fitgnn_sum_leading_f32 in two fixed-order stages (one workgroup walking all of them took 765 us
    at S-products) == the float64 product, the same bits twice, with and without a remainder of partials / rows."""
    _lib, csr, ops, orc, gorc = mods
    for R in (1408 * 640, 1408 * 700 + 77, 1408 * 513 + 1407):
        torch.manual_seed(R % 1000)
        a, b = torch.randn(R, 64).cuda(), torch.randn(R, 2).cuda()
        got = ops.mm_at_b(a, b)
        ref = (a.double().t() @ b.double())
        assert got.shape == (64, 2)
        assert float((got.double() - ref).abs().max() / ref.abs().max()) < 1e-5
        assert torch.equal(got, ops.mm_at_b(a, b))


def test_tiles_only_column_sum_partials_need_no_zero_fill(mods, monkeypatch):
    """spmm_graph_dz on a graph without whole-subgraph blocks hands its kernels an UNINITIALISED partial buffer (one launch less in a
    launch-bound batch step): with that buffer poisoned with NaN (FITGNN_POISON's switch) db is finite and the zero-filled path's, for
    H = 64 / 96 / 512 (one slab, a ragged slab, two slabs) -- the tile kernel writes every (tile, column)."""
    _lib, csr, ops, orc, gorc = mods
    from fitgnn_amd._lib import EPI_DROPOUT, EPI_ELU

    ei, n = block_graph([5, 3, 9, 2, 7, 30, 4, 17, 1, 64] * 6, seed=5, p=0.4)
    g = csr.CSRGraph(ei.cuda(), n, mode="gcn")
    assert g.t.blocks is None
    for H in (64, 96, 512):
        torch.manual_seed(H)
        X, prev = torch.randn(n, H).cuda(), torch.randn(n, H).cuda()
        monkeypatch.setattr(ops, "_POISON", False)
        want, want_db = ops.spmm_graph_dz(g, X, prev, EPI_ELU | EPI_DROPOUT, p=0.5, seed=9, want_db=True)
        monkeypatch.setattr(ops, "_POISON", True)
        got, got_db = ops.spmm_graph_dz(g, X, prev, EPI_ELU | EPI_DROPOUT, p=0.5, seed=9, want_db=True)
        assert torch.equal(got, want) and torch.isfinite(got_db).all() and torch.equal(got_db, want_db)


def test_backward_epilogue_with_a_window_smaller_than_its_column_sum_scratch(mods):
    """The tile kernel's backward epilogue folds its column sums through LDS (4 waves x 64 lanes x 4 floats = 4 KiB); a window
    of 2 or 3 rows allocates less than that for the window itself: the launcher keeps the allocation at the scratch's
    size, and dZ / db equal the default window's."""
    _lib, csr, ops, orc, gorc = mods
    from fitgnn_amd._lib import EPI_DROPOUT, EPI_ELU

    ei, n = block_graph([5, 3, 9, 2, 7, 30, 4] * 8, seed=3, p=0.4)
    torch.manual_seed(1)
    X, prev = torch.randn(n, 64).cuda(), torch.randn(n, 64).cuda()
    ref_g = csr.CSRGraph(ei.cuda(), n, mode="gcn")
    want, want_db = ops.spmm_graph_dz(ref_g, X, prev, EPI_ELU | EPI_DROPOUT, p=0.5, seed=9, want_db=True)
    for rows in (2, 3):
        g = csr.CSRGraph(ei.cuda(), n, mode="gcn", lds_rows=rows)
        got, got_db = ops.spmm_graph_dz(g, X, prev, EPI_ELU | EPI_DROPOUT, p=0.5, seed=9, want_db=True)
        assert torch.equal(got, want)
        assert rel_err(got_db.cpu(), want_db.cpu()) < 1e-5


@pytest.mark.parametrize("H", [512, 96, 260])
@pytest.mark.parametrize("sizes", [[100, 7, 17, 300, 3, 3, 64, 33, 2, 1000, 5, 5, 40], [3, 2], [1], [70] * 400])
def test_row_streaming_kernel_for_a_compact_operand(mods, H, sizes):
    """fitgnn_spmm_rows_compact[_dz]_f32 (every wave streams a range of rows; CSR entries through register tiles; zero rows from
    registers) == the tile / whole-subgraph kernels reading the same compact operand through a row indirection: the product bit for
    bit, with the previous layer's derivative in the store bit for bit (hashed dropout and an injected mask), column sums to fp32
    summation order; and both == the dense operand's product.  Row counts below one 64-row batch, entry counts across several
    64-entry tiles (the 1000-row star's centre), H that is not a multiple of the 256-column slab."""
    _lib, csr, ops, orc, gorc = mods
    from fitgnn_amd._lib import EPI_DROPOUT, EPI_ELU

    ei, n = star_blocks(sizes, 2, seed=H + len(sizes))
    ptr = np.concatenate([[0], np.cumsum(sizes)])
    g = csr.CSRGraph(ei.cuda(), n, mode="gcn", ptr=ptr)
    torch.manual_seed(H + n)
    rows = torch.randperm(n).cuda()[: max(n // 3, 1)].sort().values
    k = int(rows.numel())
    Xc = torch.cat([torch.randn(k, H).cuda(), torch.zeros(ops.ZERO_ROWS, H).cuda()])
    pos = ops._compact_positions(g, rows)
    dense = torch.zeros(n, H).cuda(); dense[rows] = Xc[:k]
    old, new = ops.OpConfig(compact_rows_kernel=False, profile=[]), ops.OpConfig(compact_rows_kernel=True, rows_kernel_min_rows=0, profile=[])
    for transposed in (True, False):
        want = ops.spmm_graph(g, dense, transposed=transposed)
        a = ops.spmm_graph(g, Xc, transposed=transposed, xrow=pos, zero_from=k, cfg=old)
        b = ops.spmm_graph(g, Xc, transposed=transposed, xrow=pos, zero_from=k, cfg=new)
        assert torch.equal(a, want) and torch.equal(b, want)
    assert new.profile[-1][2] == "compact" and len(new.profile) == 2
    prev = torch.randn(n, H).cuda() * (torch.rand(n, H).cuda() > 0.3)
    for mask in (None, (torch.rand(n, H).cuda() > 0.5).to(torch.uint8)):
        flags, p, seed = EPI_ELU | EPI_DROPOUT, 0.5, 77
        a, da = ops.spmm_graph_dz(g, Xc, prev, flags, p=p, seed=seed, mask=mask, want_db=True, xrow=pos, zero_from=k, cfg=old)
        b, db = ops.spmm_graph_dz(g, Xc, prev, flags, p=p, seed=seed, mask=mask, want_db=True, xrow=pos, zero_from=k, cfg=new)
        assert torch.equal(a, b)
        assert rel_err(db.cpu(), da.cpu()) < 1e-5 or float(da.abs().max()) == 0.0
        want, want_db = ops.epilogue_bwd_raw(ops.spmm_graph(g, dense, transposed=True), prev, flags, p=p, seed=seed, mask=mask, want_db=True)
        assert torch.equal(b, want)
        assert rel_err(db.cpu(), want_db.cpu()) < 1e-5 or float(want_db.abs().max()) == 0.0
        # ELU alone (eval mode: no dropout), no column sums
        b2, none = ops.spmm_graph_dz(g, Xc, prev, EPI_ELU, want_db=False, xrow=pos, zero_from=k, cfg=new)
        w2, _ = ops.epilogue_bwd_raw(ops.spmm_graph(g, dense, transposed=True), prev, EPI_ELU, want_db=False)
        assert none is None and torch.equal(b2, w2)
    assert new.profile[-1][2] == "compact_dz"
    # argument errors of the C entry points, before any GPU work
    L = _lib.lib()
    assert L.fitgnn_spmm_rows_compact_parts(0) == 0 and L.fitgnn_spmm_rows_compact_parts(8246057) <= 8192
    assert L.fitgnn_spmm_rows_compact_f32(None, None, None, 0, None, 512, 0, None, 512, 5, 510, None) == -1   # H % 4
    assert L.fitgnn_spmm_rows_compact_f32(None, None, None, 0, None, 512, 0, None, 512, 0, 512, None) == 0    # nothing to do


@pytest.mark.parametrize("H", [512, 96, 260])
@pytest.mark.parametrize("sizes,centres,loss", [([100, 7, 17, 300, 3, 3, 64, 33, 2, 1000, 5, 5, 40], 1, "centres"),
                                                ([60, 130, 2, 2, 700, 9], 3, "centres"), ([70] * 300, 2, "centres"),
                                                ([100, 7, 17, 300, 3, 3, 64, 33, 2, 1000, 5, 5, 40], 2, "random"), ([3, 2], 1, "random"),
                                                ([1], 1, "random")])
def test_two_hop_backward_gives_the_two_launches_bits(mods, H, sizes, centres, loss):
    """The two-hop backward (fitgnn_two_hop_rows_f32 + fitgnn_spmm_two_hop_blocks_f32, the rows outside the blocks on the tile kernel
    over the side table): G = A^T ((A^T dAH) . ELU'/dropout'(prev)) without dZ being stored as a whole == the compact dZ launch followed
    by the plain transposed SpMM, bit for bit (every entry enters both products in CSR order); column sums of dZ to fp32 summation
    order.  Blocks of every size (with block_limit 64 the larger ones and all small ones on tiles).  Loss rows = the centres of the
    stars (the production case: a leaf's columns are itself and loss rows, apart from the leaf -- leaf edges star_blocks adds) and a
    random third of the rows (rows with several loss columns, long rows that are not loss rows); centres with several hundred
    entries; hashed dropout and an injected mask; ELU alone."""
    _lib, csr, ops, orc, gorc = mods
    from fitgnn_amd._lib import EPI_DROPOUT, EPI_ELU

    ei, n = star_blocks(sizes, centres, seed=H + len(sizes))
    ptr = np.concatenate([[0], np.cumsum(sizes)])
    torch.manual_seed(H + n)
    if loss == "centres":
        rows = torch.cat([torch.arange(min(centres, max(s - 1, 1))) + o for s, o in zip(sizes, ptr[:-1])]).cuda().sort().values
    else:
        rows = torch.randperm(n).cuda()[: max(n // 3, 1)].sort().values
    k = int(rows.numel())
    Xc = torch.cat([torch.randn(k, H).cuda(), torch.zeros(ops.ZERO_ROWS, H).cuda()])
    prev = torch.randn(n, H).cuda() * (torch.rand(n, H).cuda() > 0.3)
    ran = 0
    for limit in (4096, 64):
        g = csr.CSRGraph(ei.cuda(), n, mode="gcn", ptr=ptr, block_limit=limit)
        if g.t.blocks is None:
            continue
        ran += 1
        cfg = ops.OpConfig(profile=[])
        pos = ops._compact_positions(g, rows)
        for flags, p, mask in ((EPI_ELU | EPI_DROPOUT, 0.5, None), (EPI_ELU | EPI_DROPOUT, 0.5, (torch.rand(n, H).cuda() > 0.5).to(torch.uint8)),
                               (EPI_ELU, 0.0, None)):
            link = ops.EpilogueLink()
            link.record(bool(flags & EPI_DROPOUT), p, 77, mask, True, g=g)
            assert ops.two_hop_supported(g, link, Xc, prev, cfg)
            G, db = ops.spmm_two_hop_blocks(g, Xc, prev, rows, pos, link, cfg=cfg)
            dZ, want_db = ops.spmm_graph_dz(g, Xc, prev, flags, p=p, seed=77, mask=mask, want_db=True, xrow=pos, zero_from=k)
            want = ops.spmm_graph(g, dZ, transposed=True)
            assert torch.equal(G, want)
            assert rel_err(db.cpu(), want_db.cpu()) < 1e-5 or float(want_db.abs().max()) == 0.0
        assert cfg.profile[-1][2] == "two_hop"
        ix = ops._two_hop_block_index(g, rows, pos)
        assert torch.equal(ix["zt_rows"][:k], rows.long()) and int(ix["zt_rows"].numel()) <= n
        # another graph's link: not taken
        assert not ops.two_hop_supported(csr.CSRGraph(ei.cuda(), n, mode="gcn", ptr=ptr, block_limit=limit), link, Xc, prev, cfg)
    assert ran >= 1 or max(sizes) <= 16
    if loss == "centres" and centres == 1 and len(sizes) > 3:   # pure stars (no leaf -- leaf edges): the table holds the loss rows and the
        ei0, _ = star_blocks(sizes, 1, seed=1, extra=0.0)       # rows outside the blocks, every other row is made in the window
        g0 = csr.CSRGraph(ei0.cuda(), n, mode="gcn", ptr=ptr, block_limit=4096)
        ix = ops._two_hop_block_index(g0, rows, ops._compact_positions(g0, rows))
        in_blocks = sum(s_ for s_ in sizes if s_ > 16)
        assert int(ix["zt_rows"].numel()) == k + (n - in_blocks) - sum(1 for s_ in sizes if s_ <= 16)
        assert int((ix["zrow"] < 0).sum()) == in_blocks - sum(1 for s_ in sizes if s_ > 16)
    # a batch that is not split into blocks: not taken
    g2 = csr.CSRGraph(ei.cuda(), n, mode="gcn", ptr=ptr)
    link = ops.EpilogueLink()
    link.record(True, 0.5, 77, None, True, g=g2)
    assert g2.t.blocks is not None or not ops.two_hop_supported(g2, link, Xc, prev, ops.DEFAULT)
    # argument errors of the C entry points, before any GPU work
    L = _lib.lib()
    z = [None] * 3
    assert L.fitgnn_two_hop_rows_f32(None, None, None, None, 512, 0, None, 0, None, 512, 0, 0.0, 0, None, None, 512, None) == 0
    assert L.fitgnn_two_hop_rows_f32(None, None, None, None, 512, 0, None, 3, None, 512, 0, 0.0, 0, None, None, 512, None) == -1
    assert L.fitgnn_spmm_two_hop_blocks_f32(*z, None, 512, None, 512, 0, 512, None, 0, None, None, None, None, None, 512, 0, None, None, 0, 0.0, 0, None,
                                            None, None) == 0
    assert L.fitgnn_spmm_two_hop_blocks_f32(*z, None, 512, None, 512, 9, 512, None, 2, None, None, None, None, None, 512, 0, None, None, 0, 0.0, 0, None,
                                            None, None) == -1


@pytest.mark.parametrize("H,C,with_dWl", [(512, 3, True), (512, 47, False), (64, 7, True)])
def test_head_backward_skips_rows_without_gradient_bit_for_bit(mods, H, C, with_dWl):
    """Rows whose head gradient dy is all zero (nodes outside the loss: most rows of an --extra_node subgraph) are written
    as zeros without reading `out` or forming dy @ Wl: identical bits to the kernel run on a dy whose zero rows were
    replaced by a denormal-free tiny value and compared where dy != 0, and exact +0 rows elsewhere -- also when those
    rows of `out` hold values that the long way round would have multiplied (only finiteness matters)."""
    _lib, csr, ops, orc, gorc = mods
    from fitgnn_amd._lib import EPI_DROPOUT, EPI_ELU

    torch.manual_seed(21)
    n = 5003
    out = torch.nn.functional.elu(torch.randn(n, H)).cuda()
    keep = torch.rand(n) < 0.1                       # 10 % of the rows are in the loss, in runs and singly
    keep[100:140] = True; keep[140:400] = False; keep[-1] = True
    dy = torch.randn(n, C) * keep[:, None]
    dy[7] = -0.0                                     # a row of negative zeros is a zero row too
    Wl = torch.randn(C, H).cuda()
    epi = EPI_ELU | EPI_DROPOUT
    dZ, db, dWl = ops.epilogue_bwd_head_raw(dy.cuda(), Wl, out, epi, p=0.5, seed=99, want_dWl=with_dWl)
    ref_dZ, ref_db = ops.epilogue_bwd_raw(dy.cuda() @ Wl, out, epi, p=0.5, seed=99)
    assert rel_err(dZ.cpu(), ref_dZ.cpu()) < 1e-5 and rel_err(db.cpu(), ref_db.cpu()) < 1e-4
    zero_rows = ~keep
    zero_rows[7] = True
    assert torch.count_nonzero(dZ[zero_rows.cuda()]) == 0
    assert not torch.signbit(dZ[zero_rows.cuda()]).any(), "+0, as 0 * dropout' * elu' gives"
    # the rows WITH gradient are computed exactly as when no row is skipped
    dense = dy.clone(); dense[zero_rows] = torch.randn(int(zero_rows.sum()), C)
    dZ2, _, _ = ops.epilogue_bwd_head_raw(dense.cuda(), Wl, out, epi, p=0.5, seed=99, want_dWl=with_dWl)
    live = (~zero_rows).cuda()
    assert torch.equal(dZ[live], dZ2[live])
    if with_dWl:
        assert rel_err(dWl.cpu(), (dy.double().t() @ out.cpu().double()).float()) < 1e-5


@pytest.mark.parametrize("head", [False, True])
@pytest.mark.parametrize("use_mask", [False, True])
def test_folded_backward_equals_two_kernels(mods, head, use_mask):
    """fitgnn_spmm_epilogue_bwd_f32 (dZ kept in LDS) == epilogue backward kernel followed by the transposed SpMM."""
    _lib, csr, ops, orc, gorc = mods
    from fitgnn_amd._lib import EPI_DROPOUT, EPI_ELU

    ei, n = block_graph([3, 9, 1, 30, 64, 2, 2, 5, 150, 7] * 5, seed=12, p=0.2)
    g = csr.CSRGraph(ei.cuda(), n, mode="gcn")
    assert g.fold_ok and g.window_rows <= 16
    H, C = 512, 3
    torch.manual_seed(9)
    out = torch.nn.functional.elu(torch.randn(n, H))
    mask = (torch.rand(n, H) > 0.5).to(torch.uint8).cuda() if use_mask else None
    if use_mask:
        out = out * mask.cpu() * 2.0
    else:  # production path: zeros where the hash dropped the element
        keep = ops.spmm_graph(g, torch.ones(n, H).cuda(), epilogue=EPI_DROPOUT, p=0.5, seed=77) != 0
        out = out * keep.cpu() * 2.0
    out = out.cuda()
    dOut, dy, Wl = torch.randn(n, H).cuda(), torch.randn(n, C).cuda(), torch.randn(C, H).cuda()
    epi = EPI_ELU | EPI_DROPOUT
    ref = ops.layer_backward(g, out, epi, 0.5, 77, mask, True, dOut=None if head else dOut, dy=dy if head else None,
                             Wl=Wl if head else None, want_dWl=head, cfg=ops.OpConfig(fold_backward=False))
    got = ops.layer_backward(g, out, epi, 0.5, 77, mask, True, dOut=None if head else dOut, dy=dy if head else None,
                             Wl=Wl if head else None, want_dWl=head, cfg=ops.OpConfig(fold_backward=True))
    assert rel_err(got[0].cpu(), ref[0].cpu()) < 1e-5
    assert rel_err(got[1].cpu(), ref[1].cpu()) < 1e-4
    if head:
        assert rel_err(got[2].cpu(), ref[2].cpu()) < 1e-4


def test_torch_ops_namespace(mods):
    """torch.ops.fitgnn.* (schema-registered custom ops over the C ABI): forward == the ctypes path, spmm_csr_pair's
    autograd == SpMM with the transposed pattern, opcheck-style fake kernels give the right shapes, CPU tensors are
    refused by the dispatcher (no CPU kernel is registered)."""
    _lib, csr, ops, orc, gorc = mods
    from fitgnn_amd import torch_ops

    ei, n = block_graph([3, 9, 1, 30, 64, 2, 2, 5] * 4, seed=4, p=0.3)
    g = csr.CSRGraph(ei.cuda(), n, mode="gcn")
    X = torch.randn(n, 96, device="cuda", requires_grad=True)
    Y = torch_ops.spmm(g, X)
    assert torch.equal(Y.detach(), ops.spmm_graph(g, X.detach()))
    dY = torch.randn_like(Y)
    Y.backward(dY)
    assert torch.equal(X.grad, ops.spmm_graph(g, dY, transposed=True))
    val, dinv = torch.ops.fitgnn.gcn_norm_csr(g.f.rowptr, g.f.col, None)
    assert torch.equal(val, g.f.val)
    Y2 = torch.ops.fitgnn.spmm_csr(g.f.rowptr, g.f.col, g.f.val, X.detach(), g.f.tiles, g.window_rows, None, 0, 0.0, 0, None)
    assert torch.equal(Y2, Y.detach())
    with torch.device("meta"):
        m = torch.ops.fitgnn.spmm_csr(torch.empty(n + 1, dtype=torch.int32), torch.empty(5, dtype=torch.int32), torch.empty(5),
                                      torch.empty(n, 96), torch.empty(3, 8, dtype=torch.int32), 16, None, 0, 0.0, 0, None)
    assert m.shape == (n, 96)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.fitgnn.spmm_csr(g.f.rowptr.cpu(), g.f.col.cpu(), g.f.val.cpu(), X.detach().cpu(), g.f.tiles.cpu(), 16, None, 0, 0.0, 0, None)
    assign = torch.tensor([0, 0, 1, 1, 1], dtype=torch.int32, device="cuda")
    cval = torch.tensor([2 ** -0.5] * 2 + [3 ** -0.5] * 3, dtype=torch.float64, device="cuda")
    Xs = torch.arange(10, dtype=torch.float32, device="cuda").view(5, 2)
    Xc = torch.ops.fitgnn.pool_rows(assign, cval, 2, Xs)
    ref = torch.stack([(Xs[:2].double() * 2 ** -0.5).sum(0), (Xs[2:].double() * 3 ** -0.5).sum(0)]).float()
    assert torch.allclose(Xc, ref, rtol=1e-6)
    # the GEMM kernels: a differentiable linear layer whose three products all run on them
    x = torch.randn(2048, 96, device="cuda", requires_grad=True)
    W = torch.randn(160, 96, device="cuda", requires_grad=True)
    y = torch.ops.fitgnn.linear(x, W)
    assert torch.equal(y, ops.gemm_nt(x.detach(), W.detach()))
    gy = torch.randn_like(y)
    y.backward(gy)
    assert torch.equal(W.grad, ops.gemm_atb(gy, x.detach()))
    assert float((x.grad - gy.double() @ W.detach().double()).abs().max()) < 2e-5 * float((gy.double() @ W.detach().double()).abs().max())
    with torch.device("meta"):
        assert torch.ops.fitgnn.gemm_atb(torch.empty(64, 8), torch.empty(64, 12)).shape == (8, 12)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.fitgnn.gemm_nt(x.detach().cpu(), W.detach().cpu())


# grad_W = grad_h^T @ x (csrc/gemm_atb.hip): row counts around the 32-row stage and the chunking, ragged column
# counts (500 = PubMed's feature width, not a multiple of the 256 tile), row strides wider than the operand
@pytest.mark.parametrize("R,M,N", [(90549, 512, 512), (19717, 512, 500), (4097, 260, 36), (1000, 64, 128),
                                   (256, 512, 4), (33, 8, 4), (31, 8, 8), (32, 4, 4), (1, 4, 4)])
def test_weight_gradient_gemm_matches_f64(mods, R, M, N):
    _lib, csr, ops, orc, gorc = mods
    g = torch.Generator().manual_seed(R + M + N)
    a = torch.randn(R, M, generator=g).cuda()
    b = torch.randn(R, N, generator=g).cuda()
    ref = (a.double().t() @ b.double())
    got = ops.gemm_atb(a, b)
    # three-product bf16 split, fp32 accumulate: ~5e-6 of the largest entry (the library's "high" path gives the same)
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-5
    assert torch.equal(got, ops.gemm_atb(a, b)), "fixed-order chunk sum: bit-reproducible"
    # strided operands: column windows of wider matrices
    wide_a = torch.randn(R, M + 8, generator=g).cuda()
    wide_b = torch.randn(R, N + 4, generator=g).cuda()
    va, vb = wide_a[:, 4:4 + M], wide_b[:, :N]
    got = ops.gemm_atb(va, vb)
    ref = va.double().t() @ vb.double()
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-5


def test_weight_gradient_gemm_is_the_split_policys_path(mods):
    _lib, csr, ops, orc, gorc = mods
    HIGH = ops.OpConfig(gemm_precision="high")   # the 3 x bf16 policy (the default is the exact fp32 kernel, tested below)
    a = torch.randn(3000, 64).cuda()
    b = torch.randn(3000, 96).cuda()   # operands narrower than 64 columns stay on the library path (mostly padding in a tile)
    assert HIGH.atb_kernel
    assert torch.equal(ops.mm_at_b(a, b, HIGH), ops.gemm_atb(a, b))
    ref = a.double().t() @ b.double()
    assert float((ops.mm_at_b(a, b, HIGH) - ref).abs().max() / ref.abs().max()) < 2e-5
    # exact zeros and signed values survive the hi/lo split
    z = torch.zeros(512, 8).cuda()
    assert float(ops.gemm_atb(z, b[:512]).abs().max()) == 0.0
    narrow = torch.randn(3000, 8).cuda()
    ref = narrow.double().t() @ b.double()
    assert float((ops.mm_at_b(narrow, b, HIGH) - ref).abs().max() / ref.abs().max()) < 2e-5


# The DEFAULT policy: the three products of a Linear in exact fp32 on v_mfma_f32_32x32x2_f32 (csrc/gemm_f32.hip).  Shapes: the
# configurations' own (hidden 512; feature widths 100 and 500; 47 / 48 class columns; the S-pubmed / S-products row counts cut
# down), rows around the 256 / 64-row tiles, k around the 32-wide stage and the chunking of the split-k form, strided operands.
EXACT_SHAPES = {
    "nt": [(90549, 512, 512), (20000, 512, 100), (4097, 260, 96), (1024, 512, 32), (257, 64, 64), (5, 4, 36), (19717, 500, 512),
           (300, 128, 500), (1, 4, 4), (33068, 512, 512), (34493, 512, 1024),   # the last two: a last round of 4 / 14 tiles, launched split over k
           (4861, 512, 512), (20625, 512, 512), (70000, 512, 512)],   # a 128-molecule QM9 batch (64 x 128 tiles), a rank's loss rows (128 x 128), 1.07 rounds
    "nn": [(90549, 512, 512), (4097, 96, 260), (257, 64, 64), (19717, 512, 500), (5, 36, 4), (1000, 100, 512), (33068, 512, 512),
           (165000, 512, 512), (4861, 512, 512), (20625, 512, 512)],
    "tn": [(90549, 512, 512), (19717, 512, 500), (165000, 48, 512), (40000, 512, 100), (4097, 260, 36), (1000, 64, 128), (256, 512, 4),
           (33, 8, 4), (31, 8, 8), (2048, 4, 4), (1, 4, 4), (2047, 64, 512), (4861, 512, 512)],
}


@pytest.mark.parametrize("form,dims", [(f, d) for f, ds in EXACT_SHAPES.items() for d in ds])
def test_exact_fp32_gemm_matches_f64(mods, form, dims):
    """fitgnn_gemm_exact_f32 against an fp64 product: the error is that of fp32 accumulation alone -- at most a few 1e-7 of the
    largest entry, no worse than the library's own fp32 GEMM on the same operands -- and the result is bit-reproducible."""
    _lib, csr, ops, orc, gorc = mods
    g = torch.Generator().manual_seed(sum(dims) + len(form))
    if form == "nt":      # (I, J, K): a [I, K] @ b [J, K]^T
        I, J, K = dims
        a, b = torch.randn(I, K, generator=g).cuda(), torch.randn(J, K, generator=g).cuda()
        ref = lambda x, y: x.double() @ y.double().t()      # noqa: E731
        lib = lambda x, y: x @ y.t()                        # noqa: E731
    elif form == "nn":    # (I, K, J): a [I, K] @ b [K, J]
        I, K, J = dims
        a, b = torch.randn(I, K, generator=g).cuda(), torch.randn(K, J, generator=g).cuda()
        ref = lambda x, y: x.double() @ y.double()          # noqa: E731
        lib = lambda x, y: x @ y                            # noqa: E731
    else:                 # (K, I, J): a [K, I]^T @ b [K, J]
        K, I, J = dims
        a, b = torch.randn(K, I, generator=g).cuda(), torch.randn(K, J, generator=g).cuda()
        ref = lambda x, y: x.double().t() @ y.double()      # noqa: E731
        lib = lambda x, y: x.t() @ y                        # noqa: E731
    want = ref(a, b)
    got = ops.gemm_exact(a, b, form)
    assert got.shape == want.shape
    err = float((got - want).abs().max() / want.abs().max())
    err_lib = float((lib(a, b) - want).abs().max() / want.abs().max())
    # fp32 accumulation over k: ~1e-6 of the largest entry at k = 512 (hipBLASLt's fp32 GEMM: 1.07e-6 on the same operands)
    assert err < 3e-6 and err <= 2 * err_lib + 1e-7, (form, dims, err, err_lib)
    assert torch.equal(got, ops.gemm_exact(a, b, form)), "fixed-order sums: bit-reproducible"
    # strided operands: column windows of wider matrices (row stride > extent)
    wa = torch.randn(a.shape[0], a.shape[1] + 8, generator=g).cuda()
    wb = torch.randn(b.shape[0], b.shape[1] + 4, generator=g).cuda()
    va, vb = wa[:, 4:4 + a.shape[1]], wb[:, : b.shape[1]]
    want = ref(va, vb)
    assert float((ops.gemm_exact(va, vb, form) - want).abs().max() / want.abs().max()) < 3e-6
    # exact zeros survive
    assert float(ops.gemm_exact(torch.zeros_like(a), b, form).abs().max()) == 0.0


@pytest.mark.parametrize("form,dims", [("nt", (4861, 512, 512)), ("nn", (19717, 512, 512)), ("nt", (3000, 260, 100)), ("nn", (700, 96, 132))])
def test_exact_fp32_gemm_tile_shapes_give_the_same_bits(mods, form, dims, monkeypatch):
    """Without a k split every output element is ONE MFMA chain over k in ascending order whatever tile it sits in: the 256 x 256,
    128 x 128 and 64 x 128 tile shapes (csrc/gemm_f32.hip: make_plan takes the smaller ones for grids under a round) give
    bit-identical products."""
    _lib, csr, ops, orc, gorc = mods
    g = torch.Generator().manual_seed(sum(dims))
    if form == "nt":
        I, J, K = dims
        a, b = torch.randn(I, K, generator=g).cuda(), torch.randn(J, K, generator=g).cuda()
    else:
        I, K, J = dims
        a, b = torch.randn(I, K, generator=g).cuda(), torch.randn(K, J, generator=g).cuda()
    outs = []
    for shape in ("0", "3", "4", "6"):   # S256, S128, S64x128, S64x64
        monkeypatch.setenv("FITGNN_GEMM_SHAPE", shape)
        monkeypatch.setenv("FITGNN_GEMM_NO_TAIL", "1")
        outs.append(ops.gemm_exact(a, b, form))
    monkeypatch.delenv("FITGNN_GEMM_SHAPE")
    monkeypatch.delenv("FITGNN_GEMM_NO_TAIL")
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    assert torch.equal(outs[0], ops.gemm_exact(a, b, form))   # (none of these shapes has a k split or a tail launch in its default plan)


def test_exact_fp32_gemm_is_the_default_policy_and_differentiates(mods):
    _lib, csr, ops, orc, gorc = mods
    assert ops.DEFAULT.gemm_precision == "exact"
    x = torch.randn(2048, 128, device="cuda", requires_grad=True)
    W = torch.randn(96, 128, device="cuda", requires_grad=True)
    y = ops.Linear.apply(x, W, ops.DEFAULT)
    assert torch.equal(y, ops.gemm_exact(x.detach(), W.detach(), "nt"))
    gy = torch.randn_like(y)
    y.backward(gy)
    assert torch.equal(x.grad, ops.gemm_exact(gy, W.detach(), "nn"))
    assert torch.equal(W.grad, ops.gemm_exact(gy, x.detach(), "tn"))
    xd, Wd = x.detach().double(), W.detach().double()
    assert float((x.grad - gy.double() @ Wd).abs().max()) < 3e-6 * float((gy.double() @ Wd).abs().max())
    assert float((W.grad - gy.double().t() @ xd).abs().max()) < 3e-6 * float((gy.double().t() @ xd).abs().max())
    # rows that cannot be read with 16-byte loads (3 class columns) run as fp32 library products
    n3, xs = torch.randn(2048, 3).cuda(), x.detach()
    ref = n3.double().t() @ xs.double()
    assert float((ops.mm_at_b(n3, xs) - ref).abs().max() / ref.abs().max()) < 3e-6
    # argument errors of the C entry point, before any GPU work
    L = _lib.lib()
    assert L.fitgnn_gemm_exact_f32(None, 4, 0, None, 4, 0, 8, 8, 8, None, 8, None, None) == -1          # NULL operands
    a = torch.randn(8, 6).cuda()
    assert L.fitgnn_gemm_exact_f32(_lib.dptr(a), 6, 0, _lib.dptr(a), 6, 0, 8, 8, 6, _lib.dptr(a), 8, None, None) == -1   # K % 4 != 0


# h = x W^T and dX = dH W (csrc/gemm_nt.hip): row counts around the 256-row tile, ragged N, K = 32..512, strided operands
@pytest.mark.parametrize("R,N,K", [(90549, 512, 512), (4097, 260, 96), (1024, 512, 32), (257, 64, 64), (5, 4, 32),
                                   (19717, 500, 512)])
def test_linear_gemm_matches_f64(mods, R, N, K):
    _lib, csr, ops, orc, gorc = mods
    g = torch.Generator().manual_seed(R + N + K)
    a = torch.randn(R, K, generator=g).cuda()
    b = torch.randn(N, K, generator=g).cuda()
    ref = a.double() @ b.double().t()
    got = ops.gemm_nt(a, b)
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-5
    assert torch.equal(got, ops.gemm_nt(a, b))
    wide_a = torch.randn(R, K + 8, generator=g).cuda()
    wide_b = torch.randn(N, K + 4, generator=g).cuda()
    va, vb = wide_a[:, 4:4 + K], wide_b[:, :K]
    ref = va.double() @ vb.double().t()
    assert float((ops.gemm_nt(va, vb) - ref).abs().max() / ref.abs().max()) < 2e-5


def test_linear_gemm_is_the_split_policys_path_and_differentiates(mods):
    _lib, csr, ops, orc, gorc = mods
    HIGH = ops.OpConfig(gemm_precision="high")
    assert HIGH.nt_kernel
    x = torch.randn(2048, 128, device="cuda", requires_grad=True)
    W = torch.randn(96, 128, device="cuda", requires_grad=True)
    y = ops.Linear.apply(x, W, HIGH)
    assert torch.equal(y, ops.gemm_nt(x.detach(), W.detach()))
    gy = torch.randn_like(y)
    y.backward(gy)
    xd, Wd = x.detach().double(), W.detach().double()
    assert float((x.grad - gy.double() @ Wd).abs().max()) < 2e-5 * float((gy.double() @ Wd).abs().max())
    assert float((W.grad - gy.double().t() @ xd).abs().max()) < 2e-5 * float((gy.double().t() @ xd).abs().max())


@pytest.mark.parametrize("n,C,wide", [(90549, 3, 0), (1000, 16, 4), (255, 1, 0), (257, 7, 1), (1, 5, 0), (70000, 47, 1), (3000, 64, 0),
                                      (5000, 80, 0)])
def test_narrow_column_sums(mods, n, C, wide):
    _lib, csr, ops, orc, gorc = mods
    x = torch.randn(n, C + wide).cuda()[:, :C]
    got = ops.colsum_narrow(x)
    ref = x.double().sum(0)
    assert float((got - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max())) * (n ** 0.5)
    assert torch.equal(got, ops.colsum_narrow(x))


def test_gemm_kernels_inside_a_captured_graph(mods):
    """MBTrainer / GraphTrainer capture whole steps in hipGraphs: the GEMM launchers (which raise the dynamic-LDS limit of
    their kernels on every call) must be capturable and replay to the same bits."""
    _lib, csr, ops, orc, gorc = mods
    a = torch.randn(4096, 128).cuda(); b = torch.randn(4096, 192).cuda(); w = torch.randn(192, 128).cuda()
    out = torch.randn(4096, 192).cuda()
    epi = _lib.EPI_ELU | _lib.EPI_DROPOUT
    eager = (ops.gemm_atb(a, b), ops.gemm_nt(a, w), *ops.gemm_nt_epilogue_bwd(a, w, out, epi, p=0.5, seed=99))
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):   # warm-up on the capture stream, as torch requires
        ops.gemm_atb(a, b); ops.gemm_nt(a, w); ops.gemm_nt_epilogue_bwd(a, w, out, epi, p=0.5, seed=99)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cap = (ops.gemm_atb(a, b), ops.gemm_nt(a, w), *ops.gemm_nt_epilogue_bwd(a, w, out, epi, p=0.5, seed=99))
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    for e, c in zip(eager, cap):
        assert torch.equal(e, c)


@pytest.mark.parametrize("H", [512, 96])
@pytest.mark.parametrize("sizes", [[3, 9, 1, 30, 64, 2, 2, 5] * 6, [200, 3, 90]])
def test_row_indirection_in_both_spmm_variants(mods, H, sizes):
    """Y = A_hat @ X_table[xrow] without materialising the gathered rows: the LDS-window kernel and the direct-gather
    kernel (used for layer 0 on a de-duplicated feature table) against the plain product on the gathered rows."""
    _lib, csr, ops, orc, gorc = mods
    ei, n = block_graph(sizes, seed=11, p=0.3)
    g = csr.CSRGraph(ei.cuda(), n, mode="gcn")
    n_table = max(4, n // 3)
    xrow = torch.randint(0, n_table, (n,), dtype=torch.int32).cuda()
    table = torch.randn(n_table, H).cuda()
    ref = ops.spmm_graph(g, table[xrow.long()].contiguous())
    for flag in (0, _lib.SPMM_GATHER):
        bias = torch.randn(H).cuda()
        got = ops.spmm_graph(g, table, xrow=xrow, epilogue=flag)
        assert torch.allclose(got, ref, rtol=1e-5, atol=1e-6), flag
        got_b = ops.spmm_graph(g, table, xrow=xrow, epilogue=flag | _lib.EPI_BIAS | _lib.EPI_ELU, bias=bias)
        ref_b = ops.spmm_graph(g, table[xrow.long()].contiguous(), epilogue=_lib.EPI_BIAS | _lib.EPI_ELU, bias=bias)
        assert torch.allclose(got_b, ref_b, rtol=1e-5, atol=1e-6), flag


@pytest.mark.parametrize("R,N,K", [(40000, 512, 64), (33000, 500, 96), (90549, 512, 512)])
def test_presplit_operand_staged_by_lds_dma_gives_the_same_bits(mods, R, N, K):
    """Full 256 x 256 grids take the small operand pre-split into the kernel's LDS image (one tiny kernel per call; b = W^T
    is read through its strides, never copied) and stage it by LDS-DMA: bit-identical to the register-staged path."""
    _lib, csr, ops, orc, gorc = mods
    g = torch.Generator().manual_seed(R + N + K)
    a = torch.randn(R, K, generator=g).cuda()
    W = torch.randn(K, N, generator=g).cuda()          # b = W^T as a strided view
    out = torch.randn(R, N, generator=g).cuda()
    epi = _lib.EPI_ELU | _lib.EPI_DROPOUT
    res = {}
    for pre in (False, True):
        cfg = ops.OpConfig(nt_presplit=pre)
        res[pre] = (ops.gemm_nt(a, W.t(), cfg), *ops.gemm_nt_epilogue_bwd(a, W.t(), out, epi, p=0.5, seed=7, cfg=cfg))
    for x, y in zip(res[False], res[True]):
        assert torch.equal(x, y)
    ref = a.double() @ W.double()
    assert float((res[True][0] - ref).abs().max() / ref.abs().max()) < 2e-5
