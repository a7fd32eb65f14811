"""torch-facing wrappers over the C ABI (include/fitgnn_hip.h) + the autograd Functions built on them.

torch supplies device memory, the current HIP stream and autograd bookkeeping; all arithmetic of the
message-passing path is done by libfitgnn_hip.so, the tall dense products of a layer included (MFMA kernels
csrc/gemm_nt.hip / gemm_atb.hip; BASELINE.json's north_star: MFMA for the dense weight GEMM).  Nothing here is
process-wide mutable state: switches travel in an OpConfig argument.
"""
import ctypes

import os

import torch

from . import _lib
from ._lib import EPI_BIAS, EPI_DROPOUT, EPI_ELU


class OpConfig:
    """Switches of the op layer, owned by whoever issues the ops (a trainer, a model, one call) -- there is no
    process-wide mutable state: every wrapper and autograd Function below takes the config it runs under as an
    argument (the Functions keep it on `ctx` for their backward, which autograd runs on another thread), so two
    trainers / streams / devices in one process never see each other's settings (re-entrant).

    gemm_precision   dense GEMM policy (BASELINE.json north_star: MFMA for the dense weight GEMMs).
                     "exact" (default): the reference's arithmetic -- fp32 operands, exact fp32 products, fp32 accumulation -- on
                     v_mfma_f32_32x32x2_f32 through the hand-written kernel csrc/gemm_f32.hip (x @ W^T, dH @ W and the split-k
                     dH^T @ x; ~1e-7 against fp64, the rounding of fp32 accumulation alone).  Operands whose rows cannot be read
                     with 16-byte loads (a column count that is not a multiple of 4: a 3-class head) run as fp32 library products.
                     "high": the tall fp32 products of a Linear as three bf16 products (hi.hi + hi.lo + lo.hi of the
                     two-term split of each operand) on the bf16 MFMA pipe with fp32 accumulation -- measured 4-5e-6
                     relative error vs fp64 -- through the hand-written kernels csrc/gemm_nt.hip (x @ W^T, dH @ W) and
                     csrc/gemm_atb.hip (dH^T @ x).  Shapes they do not take (operands narrower than 64 columns, fewer
                     than 1024 rows, a non-static operand whose K is not a multiple of 32) run as plain fp32 library
                     products.  "highest": the library's fp32 MFMA kernels everywhere.
    atb_kernel / nt_kernel / nt_presplit / fuse_dx_epilogue   switch the hand-written kernels off one by one (A/B).
    fold_backward    use fitgnn_spmm_epilogue_bwd_f32 (dZ kept in LDS) when the graph / shape allow it.  Off: measured
                     no faster than the two kernels (DESIGN.md "folded backward").
    dedup_gather     layer 0 on a de-duplicated table through the direct-gather SpMM variant.
    split_large_blocks  diagonal blocks larger than the SpMM window through the whole-subgraph kernel (every operand row read
                     once) instead of window-sized tiles (A/B switch; identical bits).
    last_layer_on_loss_rows  with a loss_rows promise, the last GCN layer runs aggregate-first -- A_hat h over EVERY row and edge, then
                     the dense part (x W^T, bias, ELU, dropout, the head, and in the backward both weight-side products) on the
                     loss rows alone: (A h) W^T = A (h W^T), and rows outside the loss feed nothing (FusedGCNLastLayerRows).
    compact_head_backward  with a loss_rows promise, the last layer's dZ (zero outside those rows) is produced compactly
                     [len(loss_rows) + 1 x H] (last row zero) and the backward SpMM reads it through a row indirection: the
                     [R x H] matrix that is 98 % zeros on an --extra_node union is neither written nor read.  Every edge is
                     still aggregated (most with the zero row, which stays in cache).
    compact_rows_kernel  a backward SpMM whose operand is compact (compact_head_backward / last_layer_on_loss_rows) runs on the
                     row-streaming kernel (fitgnn_spmm_rows_compact[_dz]_f32: no LDS windows, every wave streams a range of rows)
                     instead of the tile / whole-subgraph kernels with a row indirection (A/B switch; dZ bit-identical).
    rows_kernel_min_rows  ... for a backward product WITH the previous layer's derivative in its store (spmm_graph_dz) only from this
                     many rows on (32 768): the kernel gives every wave a contiguous range of >= 32 rows, so a 5 000-row batch of small
                     graphs is 150 ranges -- a few dozen workgroups walking their rows one after the other (38 us on a 128-molecule QM9
                     batch against 8 us for the tile kernel with the row indirection); it is the kernel for unions of 10^5 rows and more.
    stream_kernel    a batch whose runs go to the whole-subgraph kernel (split_large_blocks) runs on the segment-streaming kernel instead
                     (fitgnn_spmm_csr_stream[_dz]_f32: the same algorithm with one wave per run of segments, no LDS; one launch covers
                     every row; same bits).  Off: measured slower than the whole-subgraph kernel on S-products (7.6-7.9 vs 6.3-6.7 ms
                     per plain launch): hipcc's s_waitcnt vmcnt(0) in front of every first use keeps a wave to one memory
                     round trip per group of rows.
    two_hop_backward the layer below a last layer evaluated on the loss rows receives A_hat^T dZ straight from that layer's backward
                     (fitgnn_spmm_two_hop_blocks_f32: dZ = (A_hat^T dAH) . ELU'/dropout' is made in the whole-subgraph kernel's LDS
                     windows and never written as a whole) instead of dZ followed by its own plain SpMM: most of the 8 H R bytes of
                     writing and re-reading dZ are not moved (same bits for the gradient rows; the bias gradient's partial sums are
                     grouped differently).  Needs a batch that runs on the whole-subgraph kernel (split_large_blocks).
    narrow_input_first  a first GCN layer whose input has at most 32 columns and needs no gradient (QM9: 11 atom features into hidden
                     512) runs aggregate-first, (A_hat x) W^T, with A_hat x formed ONCE per (graph, input) and kept on the graph: per
                     step the layer is one pass over its output (fitgnn_dense_narrow_k_f32) and its backward one pass over the incoming
                     gradient (fitgnn_narrow_atb_f32: no SpMM, no dZ) -- FusedGCNLayerAggregatedInput.
    pooled_rows_last_layer  the last GCN layer of the *_graph_gs models aggregate-first on the rows their pool reads (x[mask]), output
                     compact (FusedGCNLayerRows): its dense products run on about half of the union's rows (A/B switch).
    appnp_in_lds     APPNP's K propagation steps for the subgraphs that fit a wavefront's LDS (<= 64 rows) in ONE launch with the signal
                     resident in LDS (fitgnn_appnp_units_f32), the per-step kernel only on the sub-matrix of the larger subgraphs (A/B).
    appnp_sliced     with appnp_in_lds: the K steps in LDS a slice of <= 4 float4 columns at a time (fitgnn_appnp_lds_f32): the units at
                     three times the wavefronts per CU, a long row summed by a whole wavefront, and every larger subgraph that fits LDS
                     one slice at a time (a 2 000-row subgraph of a 47-class signal does); off = the whole-signal units kernel (A/B).
    appnp_blocks     with appnp_in_lds: the larger subgraphs (<= 4 096 rows) one workgroup each, all K steps in ONE launch between two
                     scratch signals that stay in L2, CSR slice in LDS (fitgnn_appnp_blocks_f32); off = the per-step kernel on them (A/B).
    fused_pool_head  lt1(global_mean_pool(x[rows])) of the graph-level regression models as one launch each way (MeanPoolHead) instead
                     of pool, scale, library product and bias add (A/B switch).
    grad_sink        None, or an object with `.view(data_ptr)` -> the slice of a "fresh gradients" buffer that belongs to the parameter stored at
                     that address (train.FlatGrads with fresh=True).  Backward nodes that know it write a weight / bias gradient THERE
                     (the product's own output buffer) and return None for it, so autograd issues no `grad += new` launch per tensor; the
                     optimiser kernel folds the buffer into the accumulated gradients (fitgnn_adam_step_acc_f32).  A parameter must
                     feed ONE such node per backward (a second write would replace the first, not add to it).
    pad_table_min_k  static feature tables at least this wide whose width is not a multiple of 32 run layer 0's
                     products on a copy zero-padded once (real feature widths: 100, 500, 1 433, 8 415).
    profile / profile_gemm / profile_fused   None, or a list that collects HIP-event pairs around the SpMM / hand-written
                     GEMM / folded-backward launches (recorded on the stream the kernel is launched on).
    seed_bank        None, or a SeedBank supplying device-resident dropout seeds (steps captured in a hipGraph)."""
    __slots__ = ("gemm_precision", "atb_kernel", "nt_kernel", "nt_presplit", "fuse_dx_epilogue", "fold_backward",
                 "dedup_gather", "pad_table_min_k", "split_large_blocks", "compact_head_backward", "last_layer_on_loss_rows",
                 "compact_rows_kernel", "stream_kernel", "two_hop_backward", "narrow_input_first", "fused_pool_head", "pooled_rows_last_layer", "rows_kernel_min_rows", "appnp_in_lds", "appnp_blocks", "appnp_sliced", "grad_sink", "profile", "profile_gemm",
                 "profile_fused", "seed_bank")

    def __init__(self, gemm_precision="exact", atb_kernel=True, nt_kernel=True, nt_presplit=True, fuse_dx_epilogue=True,
                 fold_backward=False, dedup_gather=True, pad_table_min_k=0, split_large_blocks=True, compact_head_backward=True,
                 last_layer_on_loss_rows=True, compact_rows_kernel=True, stream_kernel=False, two_hop_backward=True, narrow_input_first=True, fused_pool_head=True, pooled_rows_last_layer=True, rows_kernel_min_rows=32768, appnp_in_lds=True, appnp_blocks=True, appnp_sliced=True, grad_sink=None,
                 profile=None, profile_gemm=None, profile_fused=None, seed_bank=None):
        if gemm_precision not in ("exact", "high", "highest"):
            raise ValueError(f"gemm_precision {gemm_precision!r}: 'exact', 'high' or 'highest'")
        self.gemm_precision, self.atb_kernel, self.nt_kernel, self.nt_presplit = gemm_precision, atb_kernel, nt_kernel, nt_presplit
        self.fuse_dx_epilogue, self.fold_backward, self.dedup_gather = fuse_dx_epilogue, fold_backward, dedup_gather
        self.pad_table_min_k, self.split_large_blocks = pad_table_min_k, split_large_blocks
        self.compact_head_backward, self.last_layer_on_loss_rows = compact_head_backward, last_layer_on_loss_rows
        self.compact_rows_kernel, self.stream_kernel, self.two_hop_backward = compact_rows_kernel, stream_kernel, two_hop_backward
        self.narrow_input_first, self.fused_pool_head, self.grad_sink = narrow_input_first, fused_pool_head, grad_sink
        self.pooled_rows_last_layer, self.rows_kernel_min_rows = pooled_rows_last_layer, int(rows_kernel_min_rows)
        self.appnp_in_lds, self.appnp_blocks, self.appnp_sliced = appnp_in_lds, appnp_blocks, appnp_sliced
        self.profile, self.profile_gemm, self.profile_fused, self.seed_bank = profile, profile_gemm, profile_fused, seed_bank

    def replace(self, **kw):
        """A copy with some fields changed."""
        cur = {k: getattr(self, k) for k in self.__slots__}
        cur.update(kw)
        return OpConfig(**cur)


DEFAULT = OpConfig()   # what ops run under when their caller names no config; never modified by this package


def _sink(cfg, t):
    """The fresh-gradient slice of parameter tensor `t` (or of the parameter stored at address `t`) under cfg.grad_sink, else None."""
    if cfg.grad_sink is None or t is None:
        return None
    return cfg.grad_sink.view(t if isinstance(t, int) else t.data_ptr())


def mm(a, b):
    """Library fp32 product (hipBLASLt fp32 MFMA kernels): the fallback for shapes the hand-written kernels do not take."""
    return torch.mm(a, b)


def _rows_16b(t):
    """A 2-D fp32 device matrix whose rows csrc/gemm_f32.hip can read with 16-byte loads."""
    return (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.shape[0] >= 1 and t.shape[1] >= 4 and t.shape[1] % 4 == 0
            and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.stride(0) >= t.shape[1] and t.data_ptr() % 16 == 0)


def _exact(cfg, a, b):
    return cfg.gemm_precision == "exact" and _rows_16b(a) and _rows_16b(b)


def gemm_exact(a, b, form, cfg=DEFAULT, out=None):
    """One product of a Linear in exact fp32 on the fp32 MFMA (fitgnn_gemm_exact_f32, csrc/gemm_f32.hip):
        form "nt": a [I, K] @ b [J, K]^T   (forward x W^T)
             "nn": a [I, K] @ b [K, J]     (grad_x = dH W: W is read as it lies, no transposed copy)
             "tn": a [K, I]^T @ b [K, J]   (grad_W = dH^T x: reduction over the rows, split over k with a fixed-order sum)."""
    L = _lib.lib()
    if form == "nt":
        (I, K), J, akm, bkm = a.shape, b.shape[0], 0, 0
        assert b.shape[1] == K
    elif form == "nn":
        (I, K), J, akm, bkm = a.shape, b.shape[1], 0, 1
        assert b.shape[0] == K
    elif form == "tn":
        (K, I), J, akm, bkm = a.shape, b.shape[1], 1, 1
        assert b.shape[0] == K
    else:
        raise ValueError(form)
    if out is None:
        out = torch.empty((I, J), dtype=torch.float32, device=a.device)
    else:   # the caller's [I, J] buffer (rows may be strided: e.g. the head of a taller matrix)
        assert out.shape == (I, J) and out.dtype == torch.float32 and out.stride(1) == 1 and out.is_cuda
    wb = int(L.fitgnn_gemm_exact_workspace_bytes(I, J, K, akm, bkm))
    ws = torch.empty(wb // 4, dtype=torch.float32, device=a.device) if wb else None
    ev = _gemm_events(cfg, "gemm_f32_kernel[%s]" % form, 2.0 * I * J * K)
    rc = L.fitgnn_gemm_exact_f32(_lib.dptr(a), a.stride(0), akm, _lib.dptr(b), b.stride(0), bkm, I, J, K, _lib.dptr(out), out.stride(0),
                                 _lib.dptr(ws), _lib.stream_ptr(a.device))
    _gemm_done(cfg, ev)
    _lib.check(rc, "fitgnn_gemm_exact_f32")
    return out


def padded_weight(W, Kp):
    """W [N, K] zero-padded to [N, Kp] columns (the partner of padded_table for a feature width that is not a multiple of 4)."""
    Wp = torch.zeros((W.shape[0], Kp), dtype=torch.float32, device=W.device)
    Wp[:, : W.shape[1]] = W
    return Wp


def _full_grid(R, N):
    return ((R + 255) // 256) * ((N + 255) // 256) >= 128


def _nt_ok(a, b, cfg):
    """a [R, K] @ b [N, K]^T can run on csrc/gemm_nt.hip.  b may be a strided view (e.g. W.t()) when the pre-split path
    applies: it reads b through its strides."""
    if not (cfg.nt_kernel and cfg.gemm_precision == "high" and a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32
            and a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and a.shape[1] % 32 == 0 and a.shape[0] >= 1024
            and b.shape[0] >= 64 and a.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0):
        return False
    if cfg.nt_presplit and _full_grid(a.shape[0], b.shape[0]):
        return True
    return b.stride(1) == 1 and b.stride(0) % 4 == 0 and b.data_ptr() % 16 == 0


def mm_xwt(x, W, cfg=DEFAULT):
    """x [R, K] @ W [N, K]^T (a Linear's forward) under the GEMM policy."""
    if _exact(cfg, x, W):
        return gemm_exact(x, W, "nt", cfg)
    if _nt_ok(x, W, cfg):
        return gemm_nt(x, W, cfg)
    return mm(x, W.t())


def _wt_operand(a, W, cfg):
    """W^T as the b operand of a @ W: the strided view when the pre-split path will read it through its strides (no copy),
    else a contiguous transpose."""
    if cfg.nt_presplit and cfg.nt_kernel and _full_grid(a.shape[0], W.shape[1]):
        return W.t()
    return W.t().contiguous()


def mm_by_transposed(a, W, cfg=DEFAULT, out=None):
    """a @ W for a square-ish weight W [out, in]: the library's kernel for a row-major right operand (NN) takes 201 us on
    the S-pubmed union, the one for a transposed right operand (the forward's x @ W^T form) 160 us -- materialise W^T
    (1 MB) and use the latter.  Bit-identical result.  out: write the product there (the exact kernel stores into it directly)."""
    if _exact(cfg, a, W):
        return gemm_exact(a, W, "nn", cfg, out=out)
    if out is not None:
        out.copy_(mm_by_transposed(a, W, cfg))
        return out
    if a.is_cuda and W.dim() == 2 and W.shape[0] >= 64 and W.shape[1] >= 64:
        Wt = _wt_operand(a, W, cfg)
        if _nt_ok(a, Wt, cfg):
            return gemm_nt(a, Wt, cfg)
        return mm(a, W.t().contiguous().t())
    return mm(a, W)


def _atb_ok(t):
    return (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0
            and t.shape[1] % 4 == 0 and t.shape[1] >= 4 and t.data_ptr() % 16 == 0)


class _timed:
    """`with _timed(cfg, kind):` -- when cfg.profile is a list, a HIP-event pair around the launches inside (recorded on torch's
    current stream = the one handed to the C ABI) is appended to it as (start, end, kind)."""
    __slots__ = ("cfg", "kind", "ev")

    def __init__(self, cfg, kind):
        self.cfg, self.kind, self.ev = cfg, kind, None

    def __enter__(self):
        if self.cfg is not None and self.cfg.profile is not None:
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record()
        return self

    def __exit__(self, *exc):
        if self.ev is not None:
            self.ev[1].record()
            self.cfg.profile.append((self.ev[0], self.ev[1], self.kind))
        return False


def _gemm_events(cfg, name, flops):
    """cfg.profile_gemm: list of (start event, end event, kernel name, bf16 flops) per hand-written GEMM launch."""
    if cfg.profile_gemm is None:
        return None
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), name, flops)
    ev[0].record()
    return ev


def _gemm_done(cfg, ev):
    if ev is not None:
        ev[1].record()
        cfg.profile_gemm.append(ev)


def gemm_atb(a, b, cfg=DEFAULT):
    """a^T @ b through the hand-written split-K MFMA kernel (csrc/gemm_atb.hip): 3 x bf16 products, fp32 accumulate,
    fixed-order sum of the row chunks."""
    L = _lib.lib()
    R, M, N = a.shape[0], a.shape[1], b.shape[1]
    ws = torch.empty(int(L.fitgnn_gemm_atb_workspace_bytes(R, M, N)) // 4, dtype=torch.float32, device=a.device)
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    ev = _gemm_events(cfg, "gemm_atb_kernel+atb_reduce_kernel", 6.0 * R * M * N)
    rc = L.fitgnn_gemm_atb_f32(_lib.dptr(a), a.stride(0), _lib.dptr(b), b.stride(0), R, M, N, _lib.dptr(out), _lib.dptr(ws),
                               _lib.stream_ptr(a.device))
    _gemm_done(cfg, ev)
    _lib.check(rc, "fitgnn_gemm_atb_f32")
    return out


def _presplit(b, K_pad=None):
    """The kernel's LDS image of b [N, K] (any strides: pass W.t() for the backward product) -- bf16 hi/lo fragments.
    K_pad (multiple of 32, >= K): the image's k extent, zero beyond K."""
    L = _lib.lib()
    N, K = b.shape
    Kp = K if K_pad is None else int(K_pad)
    img = torch.empty(int(L.fitgnn_gemm_nt_presplit_bytes(N, Kp)), dtype=torch.uint8, device=b.device)
    _lib.check(L.fitgnn_gemm_nt_presplit_f32(_lib.dptr(b), b.stride(0), b.stride(1), N, Kp, K, _lib.dptr(img),
                                             _lib.stream_ptr(b.device)), "fitgnn_gemm_nt_presplit_f32")
    return img


def padded_table(x):
    """A static operand table whose width is not a multiple of 32 (real feature widths: 100, 500, 1 433, 8 415), zero-padded
    once to the next multiple so that the hand-written GEMM kernels can take it; cached on the tensor object."""
    pad = getattr(x, "_fitgnn_pad", None)
    if pad is None or pad[0] != x._version:
        Kp = (x.shape[1] + 31) // 32 * 32
        xp = torch.zeros((x.shape[0], Kp), dtype=torch.float32, device=x.device)
        xp[:, : x.shape[1]] = x
        x._fitgnn_pad = pad = (x._version, xp)
    return pad[1]


def gemm_nt_padded_k(a_pad, b, cfg=DEFAULT):
    """a_pad [R, K'] (zero columns from K = b.shape[1] on) @ b [N, K]^T through the pre-split path."""
    L = _lib.lib()
    R, Kp, N = a_pad.shape[0], a_pad.shape[1], b.shape[0]
    out = torch.empty((R, N), dtype=torch.float32, device=a_pad.device)
    img = _presplit(b, K_pad=Kp)
    ev = _gemm_events(cfg, "gemm_nt_kernel<4,false,true>", 6.0 * R * N * Kp)
    rc = L.fitgnn_gemm_nt_pre_f32(_lib.dptr(a_pad), a_pad.stride(0), _lib.dptr(img), R, N, Kp, _lib.dptr(out), N,
                                  _lib.stream_ptr(a_pad.device))
    _gemm_done(cfg, ev)
    _lib.check(rc, "fitgnn_gemm_nt_pre_f32")
    return out


def _padded_table_path(Xt, W, cfg):
    """Layer 0 on a static feature table whose width is not a multiple of 32: run its two products on a zero-padded copy."""
    return (cfg.nt_kernel and cfg.atb_kernel and cfg.nt_presplit and cfg.gemm_precision == "high" and Xt.is_cuda
            and Xt.dtype == torch.float32 and Xt.shape[1] % 32 != 0 and Xt.shape[1] >= cfg.pad_table_min_k and not Xt.requires_grad
            and W.shape[0] % 4 == 0 and W.shape[0] >= 64 and Xt.shape[0] >= 1024)


def gemm_nt(a, b, cfg=DEFAULT):
    """a [R, K] @ b [N, K]^T through the hand-written MFMA kernel (csrc/gemm_nt.hip).  b may be any strided view."""
    L = _lib.lib()
    R, K, N = a.shape[0], a.shape[1], b.shape[0]
    out = torch.empty((R, N), dtype=torch.float32, device=a.device)
    if cfg.nt_presplit and _full_grid(R, N):
        img = _presplit(b)
        ev = _gemm_events(cfg, "gemm_nt_kernel<4,false,true>", 6.0 * R * N * K)
        rc = L.fitgnn_gemm_nt_pre_f32(_lib.dptr(a), a.stride(0), _lib.dptr(img), R, N, K, _lib.dptr(out), N, _lib.stream_ptr(a.device))
        _gemm_done(cfg, ev)
        _lib.check(rc, "fitgnn_gemm_nt_pre_f32")
        return out
    b = b.contiguous()
    _lib.check(L.fitgnn_gemm_nt_f32(_lib.dptr(a), a.stride(0), _lib.dptr(b), b.stride(0), R, N, K, _lib.dptr(out), N,
                                    _lib.stream_ptr(a.device)), "fitgnn_gemm_nt_f32")
    return out


def gemm_nt_epilogue_bwd(a, b, out, epilogue, p=0.0, seed=0, mask=None, want_db=True, cfg=DEFAULT):
    """(dZ, db) with dOut = a @ b^T formed inside the GEMM and transformed in its epilogue (csrc/gemm_nt.hip, EPI):
    what mm + epilogue_bwd_raw compute, without the [R x N] round trip of dOut."""
    _lib.require_cuda(a, b, out, mask)
    L = _lib.lib()
    R, K, N = a.shape[0], a.shape[1], b.shape[0]
    seed, epilogue = _seed_arg(seed, epilogue)
    dZ = torch.empty((R, N), dtype=torch.float32, device=a.device)
    db = torch.empty(N, dtype=torch.float32, device=a.device) if want_db else None
    wb = int(L.fitgnn_gemm_nt_epilogue_bwd_workspace_bytes(R, N))
    work = torch.empty(max(wb, 4), dtype=torch.uint8, device=a.device)
    if cfg.nt_presplit and _full_grid(R, N):
        b_arg, ldb = _presplit(b), 0
    else:
        b_arg = b.contiguous()
        ldb = b_arg.stride(0)
    ev = _gemm_events(cfg, "gemm_nt_kernel<4,true,%s>" % ("true" if ldb == 0 else "false"), 6.0 * R * N * K)
    rc = L.fitgnn_gemm_nt_epilogue_bwd_f32(_lib.dptr(a), a.stride(0), _lib.dptr(b_arg), ldb, R, N, K, _lib.dptr(out),
                                           _lib.dptr(dZ), epilogue, float(p), seed, _lib.dptr(mask), _lib.dptr(db),
                                           _lib.dptr(work), wb, _lib.stream_ptr(a.device))
    _gemm_done(cfg, ev)
    _lib.check(rc, "fitgnn_gemm_nt_epilogue_bwd_f32")
    return dZ, db


def head_weight_grad_rows(dy_c, out_c, cfg=DEFAULT):
    """dWl = dy_c^T @ out_c for a head too wide for the epilogue kernel's registers (ogbn-products: 47 classes): the class
    columns zero-padded to 64 so that the split-K kernel takes it (the library's batched product was the last library GEMM
    of the step)."""
    C = dy_c.shape[1]
    if cfg.gemm_precision == "exact" and dy_c.is_cuda and _rows_16b(out_c):
        Cp = (C + 3) // 4 * 4
        if Cp != C or not _rows_16b(dy_c):   # class columns zero-padded to a multiple of 4 (16-byte rows)
            pad = torch.zeros((dy_c.shape[0], Cp), dtype=torch.float32, device=dy_c.device)
            pad[:, :C] = dy_c
            dy_c = pad
        return gemm_exact(dy_c, out_c, "tn", cfg)[:C].contiguous()
    if dy_c.is_cuda and C < 64 and cfg.atb_kernel and cfg.gemm_precision == "high" and dy_c.shape[0] >= 256 and _atb_ok(out_c):
        pad = torch.zeros((dy_c.shape[0], 64), dtype=torch.float32, device=dy_c.device)
        pad[:, :C] = dy_c
        return gemm_atb(pad, out_c, cfg)[:C].contiguous()
    return mm_at_b(dy_c, out_c, cfg)


def mm_at_b(a, b, cfg=DEFAULT, out=None):
    """a^T @ b for tall operands a [R, M], b [R, N] (the weight-gradient product dH^T @ X, reduction over all R
    rows): the hand-written split-K kernel where it applies.  Otherwise the library: hipBLASLt serves this huge-K /
    small-MN shape poorly as one GEMM (541 us for R = 90k, M = N = 512); as a batched GEMM over ~1400-row slices plus
    a sum of the partial products it takes 394 us -- split-K by hand.  The partials are summed in a fixed order:
    reproducible."""
    R = a.shape[0]
    if out is not None:   # the caller's buffer (a gradient sink): the exact kernel stores there, other paths are copied
        if _exact(cfg, a, b):
            return gemm_exact(a, b, "tn", cfg, out=out)
        out.copy_(mm_at_b(a, b, cfg))
        return out
    if _exact(cfg, a, b):
        return gemm_exact(a, b, "tn", cfg)
    if (cfg.gemm_precision == "high" and cfg.atb_kernel and R >= 256 and min(a.shape[1], b.shape[1]) >= 64 and _atb_ok(a)
            and _atb_ok(b)):  # narrower operands would leave most of a 256 x 256 tile multiplying padding
        return gemm_atb(a, b, cfg)
    B = R // 1408
    if B < 4:
        return mm(a.t(), b)
    Kc = R // B
    main = B * Kc
    part = torch.bmm(a[:main].view(B, Kc, a.shape[1]).transpose(1, 2), b[:main].view(B, Kc, b.shape[1]))
    W = part.shape[1] * part.shape[2]
    if part.is_cuda and W % 4 == 0:
        L = _lib.lib()
        st = _lib.stream_ptr(part.device)
        # many partials of a narrow product (GAT's h^T [da_src da_dst] at S-products: 5 856 partials of 1 024 floats) would be ONE
        # workgroup walking them all (765 us): fold them in two fixed-order stages -- viewed as [B / G, G x W] the same kernel sums
        # every G-th partial in G x W / 1 024 workgroups, then the G sums (and the B % G left-over partials)
        G = 64
        if B >= 8 * G and W <= 16384:
            Bg = B // G
            flat = part.view(B, W)
            stage = torch.empty((G + B - Bg * G, W), dtype=torch.float32, device=part.device)
            _lib.check(L.fitgnn_sum_leading_f32(_lib.dptr(flat), Bg, G * W, _lib.dptr(stage), st), "fitgnn_sum_leading_f32")
            if B > Bg * G:
                stage[G:].copy_(flat[Bg * G:])
            part, B = stage, int(stage.shape[0])
            out = torch.empty((a.shape[1], b.shape[1]), dtype=torch.float32, device=part.device)
        else:
            out = torch.empty(part.shape[1:], dtype=torch.float32, device=part.device)
        _lib.check(L.fitgnn_sum_leading_f32(_lib.dptr(part), B, W, _lib.dptr(out), st), "fitgnn_sum_leading_f32")
    else:
        out = part.sum(0)
    if main < R:
        out = out + torch.mm(a[main:].t(), b[main:])
    return out


class Linear(torch.autograd.Function):
    """h = x W^T (GCNConv's bias-free Linear) under the GEMM policy above."""

    @staticmethod
    def forward(ctx, x, W, cfg):
        ctx.save_for_backward(x, W)
        ctx.cfg = cfg
        return mm_xwt(x, W, cfg)

    @staticmethod
    def backward(ctx, dh):
        x, W = ctx.saved_tensors
        dx = mm_by_transposed(dh, W, ctx.cfg) if ctx.needs_input_grad[0] else None
        dW = mm_at_b(dh, x, ctx.cfg) if ctx.needs_input_grad[1] else None
        return dx, dW, None


def colsum_narrow(x, out=None):
    """Column sums of a tall [rows, C <= 64] matrix (the head's bias gradient) through fitgnn_colsum_narrow_f32.  out: write them there
    (a gradient sink; None is returned when it was used)."""
    if not (x.is_cuda and x.dim() == 2 and x.shape[1] <= 64 and x.dtype == torch.float32 and x.stride(1) == 1):
        # torch's dim-0 reduction of a tall narrow matrix is slow (50 us for 90 k x 3): reduce the transposed copy instead
        return x.t().contiguous().sum(1) if (x.dim() == 2 and x.shape[0] > 4 * x.shape[1]) else x.sum(0)
    L = _lib.lib()
    n, C = x.shape
    if out is None:
        out = torch.empty(C, dtype=torch.float32, device=x.device)
    wb = int(L.fitgnn_colsum_narrow_workspace_bytes(n, C))
    work = torch.empty(max(wb, 4), dtype=torch.uint8, device=x.device)
    _lib.check(L.fitgnn_colsum_narrow_f32(_lib.dptr(x), x.stride(0), n, C, _lib.dptr(out), _lib.dptr(work), wb,
                                          _lib.stream_ptr(x.device)), "fitgnn_colsum_narrow_f32")
    return out


class BiasAdd(torch.autograd.Function):
    """y + b for a tall y with a narrow row (a Linear into the classes): the bias gradient through colsum_narrow instead of torch's
    dim-0 reduction of a [2.4 M x 47] matrix (870 us at S-products against ~100 for one pass over it)."""

    @staticmethod
    def forward(ctx, y, b):
        return y + b

    @staticmethod
    def backward(ctx, g):
        return g, (colsum_narrow(g if g.stride(1) == 1 else g.contiguous()) if ctx.needs_input_grad[1] else None)


class SmallLinear(torch.autograd.Function):
    """y = x W^T + b for a narrow output (lt1: hidden -> classes, network.py:34).  nn.Linear's addmm picks a 340 us
    kernel for [90k x 512] @ [512 x 3]; mm + a broadcast add is 10x faster.  Same arithmetic."""

    @staticmethod
    def forward(ctx, x, W, b, cfg):
        ctx.save_for_backward(x, W)
        ctx.cfg = cfg
        y = torch.mm(x, W.t())
        return y if b is None else y + b

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dx = torch.mm(dy, W) if ctx.needs_input_grad[0] else None
        # dy^T x is [classes x rows] @ [rows x hidden]: the library's kernel for that shape takes 340 us on a 90 k-row batch,
        # the split-K batched product 20
        dW = mm_at_b(_f32c(dy), _f32c(x), ctx.cfg) if ctx.needs_input_grad[1] else None
        db = colsum_narrow(dy) if ctx.needs_input_grad[2] else None
        return dx, dW, db, None


def softmax_nll_raw(z, idx, labels, scale):
    """(loss, dz): loss = scale * sum_t NLL(log_softmax(z[idx[t]]), labels[t]) as a one-element tensor and dz = d loss / d z, from ONE
    launch (fitgnn_softmax_nll_f32).  z may be a [rows x C] view with a row stride (the padded signal APPNPPropagate returns): it is
    read in place and dz comes back as the same view of an equally strided buffer.  A trainer that owns the step calls
    z.backward(dz) instead of loss.backward(): the same gradient without the multiplication by autograd's ones."""
    _lib.require_cuda(z, idx, labels)
    L = _lib.lib()
    if z.dtype != torch.float32 or z.stride(1) != 1 or z.stride(0) < z.shape[1]:
        z = _f32c(z)
    n_rows, C = z.shape
    ldz = int(z.stride(0))
    n = int(idx.numel())
    idx, labels = idx.long().contiguous(), labels.long().contiguous()
    loss = torch.empty(1, dtype=torch.float32, device=z.device)
    dz = torch.empty((n_rows, ldz), dtype=torch.float32, device=z.device)
    wb = int(L.fitgnn_softmax_nll_workspace_bytes(n))
    work = torch.empty(wb, dtype=torch.uint8, device=z.device)
    _lib.check(L.fitgnn_softmax_nll_f32(_lib.dptr(z), ldz, n_rows, C, _lib.dptr(idx), _lib.dptr(labels), n, float(scale),
                                        _lib.dptr(loss), _lib.dptr(dz), _lib.dptr(work), wb, _lib.stream_ptr(z.device)),
               "fitgnn_softmax_nll_f32")
    return loss, (dz if ldz == C else dz[:, :C])


class SoftmaxNLL(torch.autograd.Function):
    """scale * sum_t NLL(log_softmax(z[idx[t]]), labels[t]): Classify_node's log_softmax (network.py:35) + NLLLoss
    (run.py:341) over the train rows, with the gradient w.r.t. the logits produced by the same kernel
    (fitgnn_softmax_nll_f32) -- one pass instead of seven small kernels.  Returns a 0-dim loss."""

    @staticmethod
    def forward(ctx, z, idx, labels, scale):
        loss, dz = softmax_nll_raw(z, idx, labels, scale)
        ctx.save_for_backward(dz)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return dz * g, None, None, None


def l1_loss_raw(out, tgt, scale, loss_out=None):
    """(loss, grad): loss = scale * sum |out - tgt| as a one-element tensor (loss_out when given: the kernel writes it there) and
    grad = d loss / d out, from ONE launch (fitgnn_l1_loss_f32).  A trainer that owns the step calls out.backward(grad) instead of
    loss.backward(): the same gradient without autograd's ones-fill and the multiplication by it."""
    _lib.require_cuda(out, tgt)
    o, t = _f32c(out).reshape(-1), _f32c(tgt).reshape(-1)
    assert o.numel() == t.numel()
    loss = loss_out if loss_out is not None else torch.empty(1, dtype=torch.float32, device=o.device)
    grad = torch.empty_like(o)
    _lib.check(_lib.lib().fitgnn_l1_loss_f32(_lib.dptr(o), _lib.dptr(t), int(o.numel()), float(scale), _lib.dptr(loss), _lib.dptr(grad),
                                             _lib.stream_ptr(o.device)), "fitgnn_l1_loss_f32")
    return loss, grad.view(out.shape)


class L1Loss(torch.autograd.Function):
    """scale * sum |out - tgt| (torch.nn.L1Loss of run.py:518,716 on a regression head's few hundred outputs) with its gradient
    from the same launch (fitgnn_l1_loss_f32) instead of sub / abs / mean and their three backward kernels.  Returns a 0-dim loss."""

    @staticmethod
    def forward(ctx, out, tgt, scale):
        _lib.require_cuda(out, tgt)
        o, t = _f32c(out).reshape(-1), _f32c(tgt).reshape(-1)
        assert o.numel() == t.numel()
        loss = torch.empty(1, dtype=torch.float32, device=o.device)
        grad = torch.empty_like(o)
        _lib.check(_lib.lib().fitgnn_l1_loss_f32(_lib.dptr(o), _lib.dptr(t), int(o.numel()), float(scale), _lib.dptr(loss), _lib.dptr(grad),
                                                 _lib.stream_ptr(o.device)), "fitgnn_l1_loss_f32")
        ctx.save_for_backward(grad)
        ctx.shape = out.shape
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g).view(ctx.shape), None, None


_POISON = os.environ.get("FITGNN_POISON", "0") == "1"


def _scratch(shape, device):
    """A buffer its kernel writes completely before anything reads it: uninitialised -- or NaN under FITGNN_POISON=1 (tests)."""
    if _POISON:
        return torch.full(shape, float("nan"), dtype=torch.float32, device=device)
    return torch.empty(shape, dtype=torch.float32, device=device)


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def spmm_raw(rowptr, col, val, tiles, X, n_rows, bias=None, epilogue=0, p=0.0, seed=0, mask=None, out=None, window_rows=0,
             lcol=None, win_cols=None, xrow=None, cfg=DEFAULT, profile_kind=None, zero_from=-1):
    """Y = epilogue(A @ X) through fitgnn_spmm_csr_f32.  X: [n_cols_of_A, H] f32 contiguous."""
    _lib.require_cuda(rowptr, col, val, tiles, X, bias, mask)
    L = _lib.lib()
    X = _f32c(X)
    H = X.shape[1]
    seed, epilogue = _seed_arg(seed, epilogue)
    Y = out if out is not None else torch.empty((n_rows, H), dtype=torch.float32, device=X.device)
    ev = None
    if cfg.profile is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    rc = L.fitgnn_spmm_csr_f32(_lib.dptr(rowptr), _lib.dptr(col), _lib.dptr(val), _lib.dptr(X), X.stride(0) if X.numel() else H,
                               _lib.dptr(Y), Y.stride(0) if Y.numel() else H, n_rows, H, _lib.dptr(tiles), int(tiles.shape[0]),
                               _lib.dptr(lcol), _lib.dptr(win_cols), _lib.dptr(xrow), int(zero_from), int(window_rows), _lib.dptr(bias), epilogue, float(p),
                               seed, _lib.dptr(mask),
                               _lib.stream_ptr(X.device))
    if ev is not None:
        ev[1].record()
        cfg.profile.append((ev[0], ev[1], profile_kind or ("gather" if (epilogue & _lib.SPMM_GATHER) else "tile")))
    _lib.check(rc, "fitgnn_spmm_csr_f32")
    return Y


def epilogue_bwd_raw(dOut, out, epilogue, p=0.0, seed=0, mask=None, want_db=True, db_out=None):
    """dZ (and db = column sums of dZ) for out = dropout(ELU(z)) through fitgnn_epilogue_bwd_f32.  db_out: write db there."""
    _lib.require_cuda(dOut, out, mask)
    L = _lib.lib()
    dOut, out = _f32c(dOut), _f32c(out)
    n, H = dOut.shape
    seed, epilogue = _seed_arg(seed, epilogue)
    dZ = torch.empty_like(dOut)
    db = (db_out if db_out is not None else torch.empty(H, dtype=torch.float32, device=dOut.device)) if want_db else None
    wb = int(L.fitgnn_epilogue_bwd_workspace_bytes(n, H)) if want_db else 0
    work = torch.empty(max(wb, 4), dtype=torch.uint8, device=dOut.device)
    rc = L.fitgnn_epilogue_bwd_f32(_lib.dptr(dOut), _lib.dptr(out), _lib.dptr(dZ), n, H, epilogue, float(p),
                                   seed, _lib.dptr(mask), _lib.dptr(db), _lib.dptr(work), wb,
                                   _lib.stream_ptr(dOut.device))
    _lib.check(rc, "fitgnn_epilogue_bwd_f32")
    return dZ, db


def epilogue_bwd_rows_raw(dOutc, outc, rows, epilogue, p=0.0, seed=0, mask=None, want_db=True, db_out=None):
    """epilogue_bwd_raw on compact matrices: row i of dOutc / outc is ORIGINAL row rows[i] (its mask entry / dropout hash)
    (fitgnn_epilogue_bwd_rows_f32)."""
    _lib.require_cuda(dOutc, outc, rows, mask)
    L = _lib.lib()
    dOutc, outc = _f32c(dOutc), _f32c(outc)
    rows = (rows if rows.dtype == torch.int64 else rows.long()).contiguous()
    n, H = outc.shape
    seed, epilogue = _seed_arg(seed, epilogue)
    dZ = torch.empty_like(outc)
    db = (db_out if db_out is not None else torch.empty(H, dtype=torch.float32, device=outc.device)) if want_db else None
    wb = int(L.fitgnn_epilogue_bwd_workspace_bytes(n, H)) if want_db else 0
    work = torch.empty(max(wb, 4), dtype=torch.uint8, device=outc.device)
    rc = L.fitgnn_epilogue_bwd_rows_f32(_lib.dptr(dOutc), _lib.dptr(outc), _lib.dptr(rows), n, 1, _lib.dptr(dZ), H, epilogue, float(p), seed,
                                        _lib.dptr(mask), _lib.dptr(db), _lib.dptr(work), wb, _lib.stream_ptr(outc.device))
    _lib.check(rc, "fitgnn_epilogue_bwd_rows_f32")
    return dZ, db


def spmm_blocks_raw(rowptr, col, val, blocks, long_rows, X, Y, bias=None, epilogue=0, p=0.0, seed=0, mask=None, cfg=DEFAULT, xrow=None,
                    xcol=None, zero_from=-1):
    """The rows of the listed large diagonal blocks of Y = epilogue(A @ X) through fitgnn_spmm_csr_blocks_f32 (one workgroup
    walks a whole subgraph: every operand row read once)."""
    _lib.require_cuda(rowptr, col, val, blocks, long_rows, X, Y, bias, mask, xrow, xcol)
    L = _lib.lib()
    seed, epilogue = _seed_arg(seed, epilogue)
    H = X.shape[1]
    ev = None
    if cfg.profile is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    rc = L.fitgnn_spmm_csr_blocks_f32(_lib.dptr(rowptr), _lib.dptr(col), _lib.dptr(val), _lib.dptr(X), X.stride(0), _lib.dptr(Y), Y.stride(0),
                                      int(Y.shape[0]), H, _lib.dptr(blocks), int(blocks.shape[0]), _lib.dptr(long_rows), _lib.dptr(xrow),
                                      _lib.dptr(xcol), int(zero_from), _lib.dptr(bias),
                                      epilogue, float(p), seed, _lib.dptr(mask), _lib.stream_ptr(X.device))
    if ev is not None:
        ev[1].record()
        cfg.profile.append((ev[0], ev[1], "blocks"))
    _lib.check(rc, "fitgnn_spmm_csr_blocks_f32")
    return Y


def _same_index(entry, t):
    """A cache entry (tensor, version, ...) was built from index tensor `t` as it is now.  The entry HOLDS the tensor it was built
    from, so that storage cannot be freed and handed to another tensor while the entry lives (a raw address alone can be recycled
    by the caching allocator); an in-place edit of `t` (or of any alias) moves its version counter."""
    return (entry is not None and entry[0].data_ptr() == t.data_ptr() and entry[0].numel() == t.numel() and entry[0].dtype == t.dtype
            and entry[1] == t._version)


def _entry_rows(side, xrow):
    """int32 [nnz]: the operand-table row of every CSR entry of `side` under the row indirection `xrow` (xrow[col[e]]), listed once
    per (pattern side, index tensor): see fitgnn_spmm_csr_blocks_f32."""
    cached = getattr(side, "xcol", None)
    if not _same_index(cached, xrow):
        cached = (xrow, xrow._version, xrow.index_select(0, side.col.long()).contiguous())
        side.xcol = cached
    return cached[2]


def spmm_graph(g, X, transposed=False, **kw):
    """SpMM with a CSRGraph (forward or transposed pattern), using its planned tiles and kernel variant.  A batch with
    diagonal blocks larger than the window is covered by two launches on the same stream: the tile kernel over the small
    blocks' tiles, the whole-subgraph kernel over the large blocks (disjoint output rows)."""
    side = g.t if transposed else g.f
    epi = kw.pop("epilogue", 0) | (_lib.SPMM_GATHER if g.gather else 0)
    Xc = _f32c(X)
    val = kw.pop("val", None)   # per-call CSR values in the side's entry order (GATConv: the attention weights) instead of side.val
    if val is not None:
        if (kw.get("xrow") is not None and kw.get("zero_from", -1) >= 0 and kw.get("cfg", DEFAULT).compact_rows_kernel and Xc.shape[1] % 4 == 0
                and epi == 0 and kw.get("bias") is None and kw.get("out") is None):
            return _spmm_rows_compact(g, side, Xc, kw["xrow"], kw["zero_from"], kw.get("cfg", DEFAULT), kw.get("profile_kind"), val=val)
        split = (side.blocks is not None and Xc.shape[1] % 4 == 0 and Xc.data_ptr() % 16 == 0 and kw.get("cfg", DEFAULT).split_large_blocks)
        if not split:
            return spmm_raw(side.rowptr, side.col, val, side.tiles, Xc, g.n, epilogue=epi, window_rows=g.window_rows, lcol=side.lcol,
                            win_cols=side.win_cols, **kw)
        return _spmm_split(g, side, val, Xc, epi & ~_lib.SPMM_GATHER, kw)
    if (kw.get("xrow") is not None and kw.get("zero_from", -1) >= 0 and kw.get("cfg", DEFAULT).compact_rows_kernel and Xc.shape[1] % 4 == 0
            and (epi & ~_lib.SPMM_GATHER) == 0 and kw.get("bias") is None and kw.get("out") is None):
        return _spmm_rows_compact(g, side, Xc, kw["xrow"], kw["zero_from"], kw.get("cfg", DEFAULT), kw.get("profile_kind"))
    split = (side.blocks is not None and Xc.shape[1] % 4 == 0 and Xc.data_ptr() % 16 == 0 and kw.get("cfg", DEFAULT).split_large_blocks)
    if (split and kw.get("cfg", DEFAULT).stream_kernel and g.seg is not None and kw.get("zero_from", -1) < 0 and kw.get("out") is None
            and Xc.stride(0) % 4 == 0):
        cfg = kw.get("cfg", DEFAULT)
        return _spmm_stream(g, side, Xc, kw.get("xrow"), cfg, kw.get("profile_kind"), bias=kw.get("bias"), epilogue=epi & ~_lib.SPMM_GATHER,
                            p=kw.get("p", 0.0), seed=kw.get("seed", 0), mask=kw.get("mask"))
    if split:
        epi &= ~_lib.SPMM_GATHER   # the whole-subgraph kernel reads every operand row once: it supersedes the direct-gather variant
    if not split:
        return spmm_raw(side.rowptr, side.col, side.val, side.tiles, Xc, g.n, epilogue=epi, window_rows=g.window_rows,
                        lcol=side.lcol, win_cols=side.win_cols, **kw)
    return _spmm_split(g, side, side.val, Xc, epi, kw)


def _spmm_split(g, side, val, Xc, epi, kw):
    """The two launches of a batch with large diagonal blocks: tiles over the small blocks, the whole-subgraph kernel over the rest."""
    out = kw.pop("out", None)
    xrow = kw.pop("xrow", None)
    cfg = kw.pop("cfg", DEFAULT)
    kind = kw.pop("profile_kind", None)
    zero_from = kw.pop("zero_from", -1)
    Y = out if out is not None else torch.empty((g.n, Xc.shape[1]), dtype=torch.float32, device=Xc.device)
    ev = None
    if cfg.profile is not None:   # ONE event pair around both launches: together they are the SpMM
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    quiet = cfg if cfg.profile is None else cfg.replace(profile=None)
    if side.small_tiles.shape[0]:
        spmm_raw(side.rowptr, side.col, val, side.small_tiles, Xc, g.n, epilogue=epi, window_rows=g.window_rows, out=Y, cfg=quiet,
                 xrow=xrow, zero_from=zero_from, **kw)
    xcol = None
    if xrow is not None:   # the table row of every CSR entry, listed once per (pattern side, index): see fitgnn_spmm_csr_blocks_f32
        xcol = _entry_rows(side, xrow)
    spmm_blocks_raw(side.rowptr, side.col, val, side.blocks, side.long_rows, Xc, Y, epilogue=epi, cfg=quiet, xrow=xrow, xcol=xcol,
                    zero_from=zero_from, **kw)
    if ev is not None:
        ev[1].record()
        cfg.profile.append((ev[0], ev[1], kind or ("tile" if xrow is None else "table")))   # "table": layer 0 on the de-duplicated table
    return Y


def _spmm_stream(g, side, Xc, xrow, cfg, kind, bias=None, epilogue=0, p=0.0, seed=0, mask=None, dz=None):
    """epilogue(A @ X) over EVERY row of a segmented batch through the segment-streaming kernel.  dz = (prev, want_db): the
    epilogue flags / p / seed / mask are the previous layer's forward ones and the store applies its derivative -> (dZ, db)."""
    L = _lib.lib()
    _lib.require_cuda(Xc, xrow, bias, mask)
    H, dev = Xc.shape[1], Xc.device
    xcol = _entry_rows(side, xrow) if xrow is not None else None
    Y = torch.empty((g.n, H), dtype=torch.float32, device=dev)
    st = _lib.stream_ptr(dev)
    n_seg, n_ranges = int(g.seg.numel()) - 1, int(g.range_seg.numel()) - 1
    seed_v, epi_v = _seed_arg(seed, epilogue)
    ev = None
    if cfg.profile is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    db = None
    if dz is None:
        _lib.check(L.fitgnn_spmm_csr_stream_f32(_lib.dptr(side.rowptr), _lib.dptr(side.col), _lib.dptr(side.val), int(side.col.numel()),
                                                _lib.dptr(Xc), Xc.stride(0), _lib.dptr(Y), Y.stride(0), g.n, H, _lib.dptr(g.seg), n_seg,
                                                _lib.dptr(g.range_seg), n_ranges, _lib.dptr(xrow), _lib.dptr(xcol), _lib.dptr(bias), epi_v,
                                                float(p), seed_v, _lib.dptr(mask), st), "fitgnn_spmm_csr_stream_f32")
    else:
        prev, want_db = dz
        part = torch.empty((n_ranges, H), dtype=torch.float32, device=dev) if want_db else None
        _lib.check(L.fitgnn_spmm_csr_stream_dz_f32(_lib.dptr(side.rowptr), _lib.dptr(side.col), _lib.dptr(side.val), int(side.col.numel()),
                                                   _lib.dptr(Xc), Xc.stride(0), _lib.dptr(Y), Y.stride(0), g.n, H, _lib.dptr(g.seg), n_seg,
                                                   _lib.dptr(g.range_seg), n_ranges, _lib.dptr(xrow), _lib.dptr(xcol), _lib.dptr(prev), epi_v,
                                                   float(p), seed_v, _lib.dptr(mask), _lib.dptr(part), st), "fitgnn_spmm_csr_stream_dz_f32")
        if want_db:
            db = torch.empty(H, dtype=torch.float32, device=dev)
            _lib.check(L.fitgnn_colsum_partials_f32(_lib.dptr(part), n_ranges, H, _lib.dptr(db), st), "fitgnn_colsum_partials_f32")
    if ev is not None:
        ev[1].record()
        cfg.profile.append((ev[0], ev[1], kind or ("dz" if dz is not None else ("tile" if xrow is None else "table"))))
    return Y if dz is None else (Y, db)


def _spmm_rows_compact(g, side, Xc, xrow, zero_from, cfg, kind, dz=None, val=None):
    """A @ X for a compact operand through the row-streaming kernel.  dz = (prev, epilogue, p, seed, mask, want_db): with the
    previous layer's derivative in the store -> (dZ, db).  val: per-call CSR values (side's entry order) instead of side.val."""
    L = _lib.lib()
    _lib.require_cuda(Xc, xrow, val)
    val = side.val if val is None else val
    H, dev = Xc.shape[1], Xc.device
    xcol = _entry_rows(side, xrow)
    Y = torch.empty((g.n, H), dtype=torch.float32, device=dev)
    st = _lib.stream_ptr(dev)
    ev = None
    if cfg.profile is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    db = None
    if dz is None:
        _lib.check(L.fitgnn_spmm_rows_compact_f32(_lib.dptr(side.rowptr), _lib.dptr(xcol), _lib.dptr(val), int(side.col.numel()), _lib.dptr(Xc),
                                                  Xc.stride(0), int(zero_from), _lib.dptr(Y), Y.stride(0), g.n, H, st), "fitgnn_spmm_rows_compact_f32")
    else:
        prev, epilogue, p, seed, mask, want_db = dz
        seed_v, epi_v = _seed_arg(seed, epilogue & ~_lib.SPMM_GATHER)
        n_part = int(L.fitgnn_spmm_rows_compact_parts(g.n))
        part = torch.empty((n_part, H), dtype=torch.float32, device=dev) if want_db else None
        _lib.check(L.fitgnn_spmm_rows_compact_dz_f32(_lib.dptr(side.rowptr), _lib.dptr(xcol), _lib.dptr(val), int(side.col.numel()),
                                                     _lib.dptr(Xc), Xc.stride(0), int(zero_from), _lib.dptr(Y), Y.stride(0), g.n, H, _lib.dptr(prev),
                                                     epi_v, float(p), seed_v, _lib.dptr(mask), _lib.dptr(part), st),
                   "fitgnn_spmm_rows_compact_dz_f32")
        if want_db:
            db = torch.empty(H, dtype=torch.float32, device=dev)
            _lib.check(L.fitgnn_colsum_partials_f32(_lib.dptr(part), n_part, H, _lib.dptr(db), st), "fitgnn_colsum_partials_f32")
    if ev is not None:
        ev[1].record()
        cfg.profile.append((ev[0], ev[1], kind or ("compact" if dz is None else "compact_dz")))
    return Y if dz is None else (Y, db)


def two_hop_supported(g, link, Xc, prev, cfg):
    """The two-hop backward applies: the producer ran on this very graph, the batch runs on the whole-subgraph kernel, shapes are
    vectorisable."""
    return (cfg.two_hop_backward and link is not None and link.g is g and Xc.shape[1] % 4 == 0 and prev.is_contiguous()
            and prev.shape[1] == Xc.shape[1] and g.t.blocks is not None and cfg.split_large_blocks)


NO_ROW = 0x7fffffff


BLOCK_PIECE_ROWS = 16   # csrc/spmm.hip kBlkRows: rows of a whole-subgraph block staged in LDS together
BLOCK_CARRIED = 4       # kBlkLong: long rows of a block whose operand rows stay pinned in LDS


def _two_hop_block_index(g, rows, pos):
    """Index of fitgnn_spmm_two_hop_blocks_f32 on the transposed pattern, cached on the graph per index tensor.  The whole-subgraph
    kernel serves an entry (r, c) from LDS when r and c lie in the same 16-row piece of the same block, when c is one of the block's
    carried long rows, or when r is one (it takes its columns piece by piece); every other entry reads the column's dZ from the
    side table ZT.  In ZT: the loss rows (first, in compact order), every column read that way, the carried long rows (pinned from
    the table), the rows outside the blocks (the tile kernel reads them through a row indirection) and the rows with two or more
    loss columns (their dZ is not one product).  All remaining rows are "simple": dZ = (row_w * Xc[row_p]) . ELU'/dropout'(prev).
    Returns a dict: zcol int32 [nnz], zrow int32 [R] (-1: simple), zt_rows int64 [n_zt], row_p int32 [R], row_w f32 [R],
    tile_zt int64 (table rows of the rows outside the blocks)."""
    cache = getattr(g, "_two_hop_blk", None)
    if not _same_index(cache, rows):
        side, dev, R, n_sel = g.t, rows.device, g.n, int(rows.numel())
        col = side.col.long()
        counts = (side.rowptr[1:] - side.rowptr[:-1]).long()
        ar = torch.arange(R, device=dev)
        row_e = torch.repeat_interleave(ar, counts)
        blk = side.blocks.long()
        rb, cnt = blk[:, 0], blk[:, 1] - blk[:, 0]
        nb = int(blk.shape[0])
        in_rows = torch.repeat_interleave(rb, cnt) + (torch.arange(int(cnt.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt))
        block_of = torch.full((R,), -1, dtype=torch.int64, device=dev)
        block_of[in_rows] = torch.repeat_interleave(torch.arange(nb, device=dev), cnt)
        piece_of = torch.zeros(R, dtype=torch.int64, device=dev)
        piece_of[in_rows] = (in_rows - torch.repeat_interleave(rb, cnt)) // BLOCK_PIECE_ROWS
        carried = torch.zeros(R, dtype=torch.bool, device=dev)
        long_rows = side.long_rows.long()
        for k in range(BLOCK_CARRIED):
            sel = blk[:, 5] > k
            if bool(sel.any()):
                carried[long_rows[blk[sel, 4] + k]] = True
        pc = pos.long().index_select(0, col)
        loss_e = pc < n_sel
        b_r, b_c = block_of.index_select(0, row_e), block_of.index_select(0, col)
        served = (b_r >= 0) & (b_r == b_c) & ((piece_of.index_select(0, row_e) == piece_of.index_select(0, col)) | carried.index_select(0, col)
                                              | carried.index_select(0, row_e))
        n_loss_cols = torch.zeros(R, dtype=torch.int64, device=dev).index_add_(0, row_e, loss_e.long())
        in_zt = (block_of < 0) | carried | (n_loss_cols >= 2)
        in_zt[col[~served]] = True
        in_zt[rows.long()] = False                      # the loss rows take the first n_sel places
        other = torch.nonzero(in_zt).flatten()
        zrow = torch.full((R,), -1, dtype=torch.int64, device=dev)
        zrow[rows.long()] = torch.arange(n_sel, device=dev)
        zrow[other] = n_sel + torch.arange(other.numel(), device=dev)
        zc = zrow.index_select(0, col)
        zcol = torch.where(zc >= 0, zc, torch.full_like(zc, NO_ROW)).to(torch.int32).contiguous()
        assert bool((served | (zc >= 0)).all()), "an entry the kernel gathers has no table row"
        row_p = torch.full((R,), NO_ROW, dtype=torch.int32, device=dev)
        row_w = torch.zeros(R, dtype=torch.float32, device=dev)
        row_p[row_e[loss_e]] = pc[loss_e].to(torch.int32)
        row_w[row_e[loss_e]] = side.val[loss_e]
        tile_zt = zrow[block_of < 0]
        cache = (rows, rows._version, dict(zcol=zcol, zrow=zrow.to(torch.int32).contiguous(), zt_rows=torch.cat([rows.long(), other]).contiguous(),
                                           row_p=row_p, row_w=row_w, tile_zt=tile_zt.contiguous()))
        g._two_hop_blk = cache
    return cache[2]


def spmm_two_hop_blocks(g, Xc, prev, rows, pos, link, cfg=DEFAULT, profile_kind=None):
    """(A_hat^T dZ, db) with dZ = (A_hat^T Xc) . ELU'/dropout'(prev) made in the whole-subgraph kernel's LDS windows instead of being
    written and re-read (fitgnn_two_hop_rows_f32 for the side table, fitgnn_spmm_two_hop_blocks_f32; the rows outside the blocks
    through the tile kernel over the table).  Xc: the compact operand [len(rows) + ZERO_ROWS, H]; pos = _compact_positions(g, rows)."""
    L = _lib.lib()
    side = g.t
    Xc, prev = _f32c(Xc), _f32c(prev)
    _lib.require_cuda(Xc, prev, rows, pos, link.mask)
    H, dev = Xc.shape[1], Xc.device
    n_sel = int(rows.numel())
    ix = _two_hop_block_index(g, rows, pos)
    n_zt = int(ix["zt_rows"].numel())
    seed_v, epi_v = _seed_arg(link.seed, link.epi)
    st = _lib.stream_ptr(dev)
    ev = None
    if cfg.profile is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    ZT = torch.empty((max(n_zt, 1), H), dtype=torch.float32, device=dev)
    _lib.check(L.fitgnn_two_hop_rows_f32(_lib.dptr(side.rowptr), _lib.dptr(ix["zcol"]), _lib.dptr(side.val), _lib.dptr(Xc), Xc.stride(0), n_sel,
                                         _lib.dptr(ix["zt_rows"]), n_zt, _lib.dptr(prev), H, epi_v, float(link.p), seed_v, _lib.dptr(link.mask),
                                         _lib.dptr(ZT), ZT.stride(0), st), "fitgnn_two_hop_rows_f32")
    Y = torch.empty((g.n, H), dtype=torch.float32, device=dev)
    quiet = cfg if cfg.profile is None else cfg.replace(profile=None)
    if side.small_tiles.shape[0]:
        spmm_raw(side.rowptr, side.col, side.val, side.small_tiles, ZT, g.n, window_rows=g.window_rows, out=Y, cfg=quiet, xrow=ix["zrow"])
    n_blocks = int(side.blocks.shape[0])
    # one partial row of column sums per block + one for the rows outside the blocks (their dZ sits in the table)
    # (zeros, not uninitialised memory: the two-hop kernel leaves rows of `part` unwritten -- poisoned with NaN, conv.0.bias's gradient is NaN)
    part = torch.zeros((n_blocks + 1, H), dtype=torch.float32, device=dev) if link.want_db else None
    _lib.check(L.fitgnn_spmm_two_hop_blocks_f32(_lib.dptr(side.rowptr), _lib.dptr(side.col), _lib.dptr(side.val), _lib.dptr(ZT), ZT.stride(0),
                                                _lib.dptr(Y), Y.stride(0), g.n, H, _lib.dptr(side.blocks), n_blocks, _lib.dptr(side.long_rows),
                                                _lib.dptr(ix["zrow"]), _lib.dptr(ix["zcol"]), _lib.dptr(prev), _lib.dptr(Xc), Xc.stride(0), n_sel,
                                                _lib.dptr(ix["row_p"]), _lib.dptr(ix["row_w"]), epi_v, float(link.p), seed_v, _lib.dptr(link.mask),
                                                _lib.dptr(part), st), "fitgnn_spmm_two_hop_blocks_f32")
    db = None
    if link.want_db:
        if ix["tile_zt"].numel():
            torch.sum(ZT.index_select(0, ix["tile_zt"]), 0, out=part[n_blocks])
        db = _fold_partials(part, dev, st)
    if ev is not None:
        ev[1].record()
        cfg.profile.append((ev[0], ev[1], profile_kind or "two_hop"))
    return Y, db


def _fold_partials(part, dev, st):
    """Column sums of a [n x H] matrix of partial rows in a fixed order (fitgnn_colsum_partials_f32; many rows are folded in runs first)."""
    L = _lib.lib()
    n_part, H = int(part.shape[0]), int(part.shape[1])
    if n_part > 8192:
        G = 1024
        per = n_part // G
        head = part[: G * per].view(G, per, H).sum(1)
        part = torch.cat([head, part[G * per:]]) if n_part > G * per else head
        n_part = int(part.shape[0])
    db = torch.empty(H, dtype=torch.float32, device=dev)
    _lib.check(L.fitgnn_colsum_partials_f32(_lib.dptr(part.contiguous()), n_part, H, _lib.dptr(db), st), "fitgnn_colsum_partials_f32")
    return db


def spmm_graph_dz(g, X, prev, epilogue, p=0.0, seed=0, mask=None, want_db=True, transposed=True, xrow=None, cfg=DEFAULT, profile_kind=None,
                  zero_from=-1):
    """(dZ, db): dZ = (A @ X) * dropout' * ELU'(prev), the input gradient of the fused layer whose forward output is `prev`
    (fitgnn_spmm_csr_dz_f32 / fitgnn_spmm_csr_blocks_dz_f32: the derivative is applied as the rows are stored), db = column
    sums of dZ.  `epilogue`, p, seed, mask: the FORWARD's (ELU / dropout flags).  The direct-gather variant is not used here."""
    side = g.t if transposed else g.f
    Xc, prev = _f32c(X), _f32c(prev)
    _lib.require_cuda(Xc, prev, mask, xrow)
    L = _lib.lib()
    H = Xc.shape[1]
    if xrow is not None and zero_from >= 0 and cfg.compact_rows_kernel and H % 4 == 0 and g.n >= cfg.rows_kernel_min_rows:
        return _spmm_rows_compact(g, side, Xc, xrow, zero_from, cfg, profile_kind, dz=(prev, epilogue, p, seed, mask, want_db))
    if (side.blocks is not None and cfg.split_large_blocks and cfg.stream_kernel and g.seg is not None and zero_from < 0 and H % 4 == 0
            and Xc.stride(0) % 4 == 0):
        return _spmm_stream(g, side, Xc, xrow, cfg, profile_kind, epilogue=epilogue & ~_lib.SPMM_GATHER, p=p, seed=seed, mask=mask,
                            dz=(prev, want_db))
    dev = Xc.device
    seed_v, epi_v = _seed_arg(seed, epilogue & ~_lib.SPMM_GATHER)
    split = side.blocks is not None and cfg.split_large_blocks
    tiles = side.small_tiles if split else side.tiles
    n_tiles = int(tiles.shape[0])
    n_blocks = int(side.blocks.shape[0]) if split else 0
    Y = torch.empty((g.n, H), dtype=torch.float32, device=dev)
    # (zeros, not uninitialised memory: the whole-subgraph kernel writes partial rows for some of its segments only -- with the buffer
    # poisoned with NaN 17 of 24 block rows of a split test graph stay NaN and 19 GPU tests fail; the tile kernel writes every row.
    # The fill is a 4.6-us launch of a 180-us batch step at S-qm9)
    # Tiles only (a batch of small graphs: every S-qm9 step): no fill -- FITGNN_POISON=1 hands out NaN instead, which is how the tests
    # check that every row is written.
    part = (_scratch((n_tiles, H), dev) if n_blocks == 0 else torch.zeros((n_tiles + n_blocks, H), dtype=torch.float32, device=dev)) if want_db else None
    st = _lib.stream_ptr(dev)
    ev = None
    if cfg.profile is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if n_tiles:
        _lib.check(L.fitgnn_spmm_csr_dz_f32(_lib.dptr(side.rowptr), _lib.dptr(side.col), _lib.dptr(side.val), _lib.dptr(Xc), Xc.stride(0),
                                            _lib.dptr(Y), Y.stride(0), g.n, H, _lib.dptr(tiles), n_tiles,
                                            _lib.dptr(None if split else side.lcol), _lib.dptr(None if split else side.win_cols),
                                            _lib.dptr(xrow), int(zero_from), int(g.window_rows), _lib.dptr(prev), epi_v, float(p), seed_v,
                                            _lib.dptr(mask), _lib.dptr(part), st), "fitgnn_spmm_csr_dz_f32")
    if n_blocks:
        xcol = None
        if xrow is not None:
            xcol = _entry_rows(side, xrow)
        _lib.check(L.fitgnn_spmm_csr_blocks_dz_f32(_lib.dptr(side.rowptr), _lib.dptr(side.col), _lib.dptr(side.val), _lib.dptr(Xc), Xc.stride(0),
                                                   _lib.dptr(Y), Y.stride(0), g.n, H, _lib.dptr(side.blocks), n_blocks, _lib.dptr(side.long_rows),
                                                   _lib.dptr(xrow), _lib.dptr(xcol), int(zero_from), _lib.dptr(prev), epi_v, float(p), seed_v, _lib.dptr(mask),
                                                   _lib.dptr(None if part is None else part[n_tiles:]), st), "fitgnn_spmm_csr_blocks_dz_f32")
    if ev is not None:
        ev[1].record()
        cfg.profile.append((ev[0], ev[1], profile_kind or "dz"))
    db = None
    if want_db:
        n_part = n_tiles + n_blocks
        if n_part > 8192:
            # one partial row per star: 166 000 at S-products.  colsum_partials' grid is H / 16 workgroups (0.4 ms for them); fold
            # contiguous runs of rows first (torch's reduction: deterministic for a fixed shape), then the fixed-order kernel
            G = 1024
            per = n_part // G
            head = part[: G * per].view(G, per, H).sum(1)
            part2 = torch.cat([head, part[G * per:]]) if n_part > G * per else head
            db = torch.empty(H, dtype=torch.float32, device=dev)
            _lib.check(L.fitgnn_colsum_partials_f32(_lib.dptr(part2), int(part2.shape[0]), H, _lib.dptr(db), st), "fitgnn_colsum_partials_f32")
        else:
            db = torch.empty(H, dtype=torch.float32, device=dev)
            _lib.check(L.fitgnn_colsum_partials_f32(_lib.dptr(part), n_part, H, _lib.dptr(db), st), "fitgnn_colsum_partials_f32")
    return Y, db


def epilogue_bwd_head_raw(dy, Wl, out, epilogue, p=0.0, seed=0, mask=None, want_db=True, want_dWl=True):
    """dZ / db as epilogue_bwd_raw, with dOut = dy @ Wl formed inside the kernel (fitgnn_epilogue_bwd_head_f32);
    also returns dWl = dy^T @ out when asked."""
    _lib.require_cuda(dy, Wl, out, mask)
    L = _lib.lib()
    dy, Wl, out = _f32c(dy), _f32c(Wl), _f32c(out)
    n, H = out.shape
    C = Wl.shape[0]
    seed, epilogue = _seed_arg(seed, epilogue)
    dZ = torch.empty_like(out)
    db = torch.empty(H, dtype=torch.float32, device=out.device) if want_db else None
    dWl = torch.empty((C, H), dtype=torch.float32, device=out.device) if want_dWl else None
    wb = int(L.fitgnn_epilogue_bwd_head_workspace_bytes(n, H, C))
    work = torch.empty(max(wb, 4), dtype=torch.uint8, device=out.device)
    rc = L.fitgnn_epilogue_bwd_head_f32(_lib.dptr(dy), _lib.dptr(Wl), C, _lib.dptr(out), _lib.dptr(dZ), n, H, epilogue, float(p),
                                        seed, _lib.dptr(mask), _lib.dptr(db), _lib.dptr(dWl),
                                        _lib.dptr(work), wb, _lib.stream_ptr(out.device))
    _lib.check(rc, "fitgnn_epilogue_bwd_head_f32")
    return dZ, db, dWl


# zero rows appended to a compact operand: every row outside the selection reads one of them (r % ZERO_ROWS).  ONE zero row would be
# 16 cache lines requested by all 256 CUs at once -- the L2 channels that hold them serialise (measured: the compact launch no
# faster than the dense one); 256 rows (512 KB at H = 512) spread over every channel and still sit in L2.
ZERO_ROWS = 256


def epilogue_bwd_head_rows_raw(dy, Wl, out, rows, epilogue, p=0.0, seed=0, mask=None, want_db=True, want_dWl=True, inputs_compact=False,
                               zero_rows=ZERO_ROWS, db_out=None, dWl_out=None):
    """epilogue_bwd_head_raw over the rows `rows` only, compact: dZc [len(rows) + zero_rows, H] whose last rows are zero (the
    operand of every row outside `rows` in a backward SpMM), db, dWl (fitgnn_epilogue_bwd_head_rows_f32).  inputs_compact: dy and
    out hold those rows only (row i = original row rows[i])."""
    _lib.require_cuda(dy, Wl, out, mask, rows)
    L = _lib.lib()
    dy, Wl, out = _f32c(dy), _f32c(Wl), _f32c(out)
    rows = (rows if rows.dtype == torch.int64 else rows.long()).contiguous()
    n_sel, H = int(rows.numel()), out.shape[1]
    C = Wl.shape[0]
    seed, epilogue = _seed_arg(seed, epilogue)
    dZc = torch.empty((n_sel + zero_rows, H), dtype=torch.float32, device=out.device)
    if zero_rows:
        dZc[n_sel:].zero_()
    db = (db_out if db_out is not None else torch.empty(H, dtype=torch.float32, device=out.device)) if want_db else None
    dWl = (dWl_out if dWl_out is not None else torch.empty((C, H), dtype=torch.float32, device=out.device)) if want_dWl else None
    wb = int(L.fitgnn_epilogue_bwd_head_workspace_bytes(n_sel, H, C))
    work = torch.empty(max(wb, 4), dtype=torch.uint8, device=out.device)
    rc = L.fitgnn_epilogue_bwd_head_rows_f32(_lib.dptr(dy), _lib.dptr(Wl), C, _lib.dptr(out), _lib.dptr(rows), n_sel, 1 if inputs_compact else 0,
                                             _lib.dptr(dZc), H, epilogue, float(p), seed, _lib.dptr(mask), _lib.dptr(db), _lib.dptr(dWl),
                                             _lib.dptr(work), wb, _lib.stream_ptr(out.device))
    _lib.check(rc, "fitgnn_epilogue_bwd_head_rows_f32")
    return dZc, db, dWl


def epilogue_fwd_rows_(z, rows, bias, epilogue, p=0.0, seed=0, mask=None):
    """In place: z[i] = dropout(ELU(z[i] + bias)) with the dropout pattern of original row rows[i] (fitgnn_epilogue_fwd_rows_f32)."""
    _lib.require_cuda(z, rows, bias, mask)
    seed, epilogue = _seed_arg(seed, epilogue)
    rows = (rows if rows.dtype == torch.int64 else rows.long()).contiguous()
    _lib.check(_lib.lib().fitgnn_epilogue_fwd_rows_f32(_lib.dptr(z), z.stride(0), _lib.dptr(rows), z.shape[0], z.shape[1],
                                                       _lib.dptr(None if bias is None else _f32c(bias)), epilogue, float(p), seed,
                                                       _lib.dptr(mask), _lib.stream_ptr(z.device)), "fitgnn_epilogue_fwd_rows_f32")
    return z


def _compact_positions(g, rows):
    """int32 [g.n]: position of row r in `rows`, len(rows) + r % ZERO_ROWS for the others (a zero row of a compact operand); cached on the graph
    per index tensor (a trainer passes the same loss_rows every step)."""
    cache = getattr(g, "_compact_pos", None)
    if not _same_index(cache, rows):
        pos = int(rows.numel()) + torch.arange(g.n, dtype=torch.int32, device=rows.device) % ZERO_ROWS
        pos[rows.long()] = torch.arange(rows.numel(), dtype=torch.int32, device=rows.device)
        cache = (rows, rows._version, pos)
        g._compact_pos = cache
    return cache[2]


def head_rows_supported(out, Wl):
    H, C = out.shape[1], Wl.shape[0]
    return (out.is_cuda and out.dtype == torch.float32 and Wl.dtype == torch.float32 and H % 4 == 0 and out.stride(1) == 1
            and out.stride(0) % 4 == 0 and _lib.lib().fitgnn_head_rows_lds_bytes(H, C) <= 160 * 1024)


def head_rows(out, rows, Wl, bl, n_total=None):
    """y [R, C] with y[rows] = out[rows] @ Wl^T + bl and zeros elsewhere (fitgnn_head_rows_f32).  n_total: `out` holds the selected
    rows only (row i = original row rows[i]) and R = n_total."""
    _lib.require_cuda(out, rows, Wl, bl)
    compact = n_total is not None
    R, H = (int(n_total) if compact else out.shape[0]), out.shape[1]
    C = Wl.shape[0]
    Wl = _f32c(Wl)
    y = torch.zeros((R, C), dtype=torch.float32, device=out.device)
    rows = rows if rows.dtype == torch.int64 else rows.long()
    _lib.check(_lib.lib().fitgnn_head_rows_f32(_lib.dptr(out), out.stride(0), _lib.dptr(rows.contiguous()), rows.numel(), _lib.dptr(Wl),
                                               _lib.dptr(None if bl is None else _f32c(bl)), C, H, _lib.dptr(y), C, 1 if compact else 0,
                                               _lib.stream_ptr(out.device)), "fitgnn_head_rows_f32")
    return y


def segment_sum(seg_off, members, X, n_seg):
    """out[s] = sum of X[members[seg_off[s]:seg_off[s+1]]] (fitgnn_segment_sum_f32)."""
    _lib.require_cuda(seg_off, members, X)
    X = _f32c(X)
    F_ = X.shape[1]
    out = torch.empty((n_seg, F_), dtype=torch.float32, device=X.device)
    _lib.check(_lib.lib().fitgnn_segment_sum_f32(_lib.dptr(seg_off), _lib.dptr(members), n_seg, _lib.dptr(X), F_, F_, _lib.dptr(out),
                                                 F_, _lib.stream_ptr(X.device)), "fitgnn_segment_sum_f32")
    return out


class PoolIndex:
    """Sorted-segment view of a PyG `batch` vector for the graph-level pools (fitgnn_segment_sum / _max / _expand_f32): segment s =
    the rows of graph s.  rows (int64, optional): the pooled rows are x[rows] (the *_gs models' x[mask], network.py:129,200) and
    batch[i] is the graph of rows[i] -- the gather is folded into the pool."""

    def __init__(self, batch, size, rows=None, n_rows=None):
        dev = batch.device
        b = batch.to(torch.int64)
        self.n_seg = int(size)
        self.sorted = bool(b.numel() == 0 or bool((b[1:] >= b[:-1]).all()))
        cnt = torch.bincount(b, minlength=self.n_seg)
        off = torch.zeros(self.n_seg + 1, dtype=torch.int64, device=dev)
        off[1:] = torch.cumsum(cnt, 0)
        self.seg_off = off.to(torch.int32).contiguous()
        self.n_rows = int(n_rows if n_rows is not None else (b.numel() if rows is None else 0))
        if rows is None:
            self.members = torch.arange(b.numel(), dtype=torch.int32, device=dev)
            self.seg_of_row = b.to(torch.int32).contiguous()
        else:
            self.members = rows.to(torch.int32).contiguous()
            sr = torch.full((self.n_rows,), -1, dtype=torch.int32, device=dev)
            sr[rows.long()] = b.to(torch.int32)
            self.seg_of_row = sr
        self.inv_cnt = (1.0 / cnt.clamp(min=1).to(torch.float32)).contiguous()


def pool_index(batch, size, rows=None, n_rows=None):
    """PoolIndex of (batch, rows), cached on the batch tensor (a trainer passes the same static tensors every step; building it
    synchronises with the host once, before any capture)."""
    cache = getattr(batch, "_fitgnn_pool", None)
    if cache is None:
        cache = {}
        batch._fitgnn_pool = cache
    key = (batch._version, int(size), None if rows is None else (rows.data_ptr(), rows._version), n_rows)
    hit = cache.get(key)
    if hit is None:
        if len(cache) > 8:
            cache.clear()
        hit = (PoolIndex(batch, size, rows, n_rows), rows)   # (the entry holds `rows`: its address cannot be recycled meanwhile)
        cache[key] = hit
    return hit[0]


class SegmentMeanPool(torch.autograd.Function):
    """global_mean_pool over sorted segments: one gather-and-sum launch (fitgnn_segment_sum_f32) + the per-graph 1 / count; backward:
    one launch writing every row of dx (fitgnn_segment_expand_f32)."""

    @staticmethod
    def forward(ctx, x, pi):
        x = _f32c(x)
        ctx.pi, ctx.shape = pi, x.shape
        return segment_sum(pi.seg_off, pi.members, x, pi.n_seg) * pi.inv_cnt.unsqueeze(1)

    @staticmethod
    def backward(ctx, g):
        pi = ctx.pi
        n, F_ = ctx.shape
        g = _f32c(g)
        dx = torch.empty((n, F_), dtype=torch.float32, device=g.device)
        _lib.check(_lib.lib().fitgnn_segment_expand_f32(_lib.dptr(g), _lib.dptr(pi.seg_of_row), _lib.dptr(pi.inv_cnt), n, F_, _lib.dptr(dx),
                                                        _lib.stream_ptr(g.device)), "fitgnn_segment_expand_f32")
        return dx, None


class SegmentMaxPool(torch.autograd.Function):
    """global_max_pool over sorted segments (fitgnn_segment_max_f32) with the arg-max rows kept for the backward scatter (the first row
    on a tie takes the whole gradient; torch's scatter_reduce splits it evenly among tied rows)."""

    @staticmethod
    def forward(ctx, x, pi):
        x = _f32c(x)
        n, F_ = x.shape
        out = torch.empty((pi.n_seg, F_), dtype=torch.float32, device=x.device)
        arg = torch.empty((pi.n_seg, F_), dtype=torch.int32, device=x.device)
        _lib.check(_lib.lib().fitgnn_segment_max_f32(_lib.dptr(pi.seg_off), _lib.dptr(pi.members), pi.n_seg, _lib.dptr(x), x.stride(0), F_,
                                                     _lib.dptr(out), _lib.dptr(arg), _lib.stream_ptr(x.device)), "fitgnn_segment_max_f32")
        ctx.save_for_backward(arg)
        ctx.shape = x.shape
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        n, F_ = ctx.shape
        g = _f32c(g)
        dx = torch.zeros((n, F_), dtype=torch.float32, device=g.device)
        _lib.check(_lib.lib().fitgnn_segment_max_bwd_f32(_lib.dptr(g), _lib.dptr(arg), int(arg.shape[0]), F_, _lib.dptr(dx), F_,
                                                         _lib.stream_ptr(g.device)), "fitgnn_segment_max_bwd_f32")
        return dx, None


def pool_head_supported(x, Wl):
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] > 0 and Wl.dim() == 2 and Wl.shape[1] == x.shape[1]
            and bool(_lib.lib().fitgnn_pool_head_supported(int(x.shape[1]), int(Wl.shape[0]))))


class MeanPoolHead(torch.autograd.Function):
    """lt1(global_mean_pool(x[rows])) (Regress_graph_gs / _gc, network.py:164-166, :200-204) as one launch forward
    (fitgnn_pool_head_f32: gather-and-sum, 1 / count, the head's few dot products, its bias) and one backward
    (fitgnn_pool_head_bwd_f32: every row of dx, the head's weight and bias gradients) -- on a 128-graph batch the separate pool,
    scale, [128 x 512] @ [512 x 1] library product and bias add are four launches forward and seven backward."""

    @staticmethod
    def forward(ctx, x, pi, Wl, bl, cfg):
        L = _lib.lib()
        x, W = _f32c(x), _f32c(Wl)
        n, F_ = x.shape
        C = int(W.shape[0])
        pooled = torch.empty((pi.n_seg, F_), dtype=torch.float32, device=x.device)
        y = torch.empty((pi.n_seg, C), dtype=torch.float32, device=x.device)
        _lib.check(L.fitgnn_pool_head_f32(_lib.dptr(pi.seg_off), _lib.dptr(pi.members), pi.n_seg, _lib.dptr(x), x.stride(0), F_,
                                          _lib.dptr(pi.inv_cnt), _lib.dptr(W), _lib.dptr(bl), C, _lib.dptr(pooled), _lib.dptr(y),
                                          _lib.stream_ptr(x.device)), "fitgnn_pool_head_f32")
        ctx.save_for_backward(pooled, W)
        ctx.pi, ctx.shape, ctx.cfg = pi, (n, F_), cfg
        ctx.W_ptr, ctx.b_ptr = Wl.data_ptr(), (bl.data_ptr() if bl is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        pooled, W = ctx.saved_tensors
        L = _lib.lib()
        pi, (n, F_), cfg = ctx.pi, ctx.shape, ctx.cfg
        C = int(W.shape[0])
        dy = _f32c(dy)
        dev = dy.device
        dx = torch.empty((n, F_), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        sW = _sink(cfg, ctx.W_ptr) if ctx.needs_input_grad[2] else None
        sb = _sink(cfg, ctx.b_ptr) if (ctx.b_ptr is not None and ctx.needs_input_grad[3]) else None
        dW = (sW if sW is not None else torch.empty((C, F_), dtype=torch.float32, device=dev)) if ctx.needs_input_grad[2] else None
        db = (sb if sb is not None else torch.empty(C, dtype=torch.float32, device=dev)) if (ctx.b_ptr is not None and ctx.needs_input_grad[3]) else None
        _lib.check(L.fitgnn_pool_head_bwd_f32(_lib.dptr(dy), _lib.dptr(W), C, _lib.dptr(pooled), _lib.dptr(pi.seg_of_row), _lib.dptr(pi.inv_cnt),
                                              n if dx is not None else 0, pi.n_seg, F_, _lib.dptr(dx), _lib.dptr(dW), _lib.dptr(db),
                                              _lib.stream_ptr(dev)), "fitgnn_pool_head_bwd_f32")
        return dx, None, (None if sW is not None else dW), (None if sb is not None else db), None


def pool_supported(x):
    return x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[1] % 4 == 0 and x.shape[0] > 0


class RowIndex:
    """Row indirection of a de-duplicated operand table: union row r is a copy of table row `index[r]`.
    Holds the int32 index and its inverse (segments of union rows per table row) for the adjoint."""

    def __init__(self, index, n_table):
        idx = index.to(torch.int64)
        self.n_table = int(n_table)
        self.index = idx.to(torch.int32).contiguous()
        order = torch.argsort(idx, stable=True)
        self.members = order.to(torch.int32).contiguous()
        off = torch.zeros(self.n_table + 1, dtype=torch.int64, device=idx.device)
        off[1:] = torch.cumsum(torch.bincount(idx, minlength=self.n_table), 0)
        self.seg_off = off.to(torch.int32).contiguous()


def layer_backward(g, out, epi, p, seed, mask, want_db, dOut=None, dy=None, Wl=None, want_dWl=False, cfg=DEFAULT, loss_rows=None, db_out=None):
    """(dH, db, dWl) of one fused layer: dZ = epilogue'(dOut or dy @ Wl), db = colsum dZ, dH = A^T dZ.
    Epilogue-backward kernel + SpMM; with cfg.fold_backward one kernel (dZ stays in LDS) when the graph / shape allow it --
    measured on the S-pubmed union no faster than the two kernels (254 vs 223 us per hidden layer, 262 vs 263 us with the
    head; DESIGN.md "folded backward"): the extra operand stream and the transform sit on the tile's critical path
    (descriptor -> window loads -> barrier -> row loop) while the separate elementwise kernel streams at HBM rate."""
    L = _lib.lib()
    out = _f32c(out)
    n, H = out.shape
    head = dOut is None
    C = int(Wl.shape[0]) if head else 0
    if cfg.fold_backward and getattr(g, "fold_ok", False) and L.fitgnn_spmm_epilogue_bwd_supported(H, C, g.window_rows):
        side = g.t
        dev = out.device
        dH = torch.empty_like(out)
        db = torch.empty(H, dtype=torch.float32, device=dev) if want_db else None
        dWl = torch.empty((C, H), dtype=torch.float32, device=dev) if (head and want_dWl) else None
        nt = int(side.tiles.shape[0])
        wb = int(L.fitgnn_spmm_epilogue_bwd_workspace_bytes(nt, H, C if dWl is not None else 0)) if (want_db or dWl is not None) else 0
        work = torch.empty(max(wb, 4), dtype=torch.uint8, device=dev)
        if head:
            dy, Wl = _f32c(dy), _f32c(Wl)
        else:
            dOut = _f32c(dOut)
        ev = None
        if cfg.profile_fused is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        seed_v, epi_v = _seed_arg(seed, epi)
        rc = L.fitgnn_spmm_epilogue_bwd_f32(_lib.dptr(side.rowptr), _lib.dptr(side.col), _lib.dptr(side.val), _lib.dptr(side.tiles), nt,
                                            g.window_rows, _lib.dptr(dOut), _lib.dptr(dy), _lib.dptr(Wl), C, _lib.dptr(out),
                                            _lib.dptr(dH), n, H, epi_v, float(p), seed_v, _lib.dptr(mask),
                                            _lib.dptr(db), _lib.dptr(dWl), _lib.dptr(work), wb, _lib.stream_ptr(dev))
        if ev is not None:
            ev[1].record()
            cfg.profile_fused.append(ev)
        _lib.check(rc, "fitgnn_spmm_epilogue_bwd_f32")
        return dH, db, dWl
    if head and loss_rows is not None and cfg.compact_head_backward and H % 4 == 0 and loss_rows.numel() > 0:
        inside = want_dWl and bool(L.fitgnn_epilogue_bwd_head_supported(H, C, 1))
        dZc, db, dWl = epilogue_bwd_head_rows_raw(dy, Wl, out, loss_rows, epi, p=p, seed=seed, mask=mask, want_db=want_db, want_dWl=inside)
        if want_dWl and not inside:
            dWl = head_weight_grad_rows(_f32c(dy).index_select(0, loss_rows), out.index_select(0, loss_rows), cfg)
        return spmm_graph(g, dZc, transposed=True, cfg=cfg, xrow=_compact_positions(g, loss_rows), profile_kind="compact",
                          zero_from=int(loss_rows.numel())), db, dWl
    if head:
        inside = want_dWl and bool(L.fitgnn_epilogue_bwd_head_supported(H, C, 1))
        dZ, db, dWl = epilogue_bwd_head_raw(dy, Wl, out, epi, p=p, seed=seed, mask=mask, want_db=want_db, want_dWl=inside)
        if want_dWl and not inside:   # a wide head (ogbn-products: 47 classes): the kernel's registers hold 16 class rows
            if loss_rows is not None:   # dy is zero outside these rows (the caller's contract): dy^T out over them alone
                dWl = mm_at_b(_f32c(dy).index_select(0, loss_rows), out.index_select(0, loss_rows), cfg)
            else:
                dWl = mm_at_b(_f32c(dy), out, cfg)
    else:
        dZ, db = epilogue_bwd_raw(dOut, out, epi, p=p, seed=seed, mask=mask, want_db=want_db, db_out=db_out)   # (db is db_out when given)
        dWl = None
    return spmm_graph(g, dZ, transposed=True, cfg=cfg), db, dWl


class SpMM(torch.autograd.Function):
    """Y = A @ X (+ bias).  Backward: dX = A^T @ dY (same kernel on the transposed CSR), db = sum rows."""

    @staticmethod
    def forward(ctx, X, bias, g, cfg):
        ctx.g, ctx.cfg = g, cfg
        ctx.has_bias = bias is not None
        epi = EPI_BIAS if bias is not None else 0
        return spmm_graph(g, X, bias=bias, epilogue=epi, cfg=cfg)

    @staticmethod
    def backward(ctx, dY):
        g = ctx.g
        dY = _f32c(dY)
        dX = spmm_graph(g, dY, transposed=True, cfg=ctx.cfg) if ctx.needs_input_grad[0] else None
        db = dY.sum(0) if ctx.has_bias and ctx.needs_input_grad[1] else None
        return dX, db, None, None


class SpMMRows(torch.autograd.Function):
    """Y = A_hat[rows, :] X for a csr.RowSubset (only the rows whose outputs are consumed); backward = its adjoint."""

    @staticmethod
    def forward(ctx, X, sub, cfg):
        ctx.sub, ctx.cfg = sub, cfg
        f = sub.f
        return spmm_raw(f.rowptr, f.col, f.val, f.tiles, X, sub.m, window_rows=sub.window_rows, cfg=cfg)

    @staticmethod
    def backward(ctx, dY):
        t = ctx.sub.t
        return spmm_raw(t.rowptr, t.col, t.val, t.tiles, _f32c(dY), ctx.sub.n, window_rows=ctx.sub.window_rows, cfg=ctx.cfg), None, None


class EpilogueLink:
    """Connects a fused layer (producer of out = dropout(ELU(z))) with the ONE layer that consumes `out` as its input.
    The producer's forward records its epilogue here; the consumer's backward (which runs first) may then return
    dZ = epilogue'(dOut) in place of dOut, with the bias gradient, and marks the link; the producer's backward skips its
    own epilogue-backward kernel.  Only for strictly sequential stacks (network.py:29-33): `out` must have no other
    consumer, since what travels through autograd on this edge is no longer the plain gradient."""
    __slots__ = ("epi", "p", "seed", "mask", "want_db", "fused", "db", "g", "aggregated")

    def __init__(self):
        self.epi, self.p, self.seed, self.mask, self.want_db, self.fused, self.db = 0, 0.0, 0, None, False, False, None
        self.g, self.aggregated = None, False

    def record(self, drop, p, seed, mask, want_db, g=None):
        """g: the producer's graph.  A consumer on the SAME graph may go one step further than dZ and hand back A_hat^T dZ (the
        producer's own backward SpMM), marking `aggregated` (ops.FusedGCNLastLayerRows, OpConfig.two_hop_backward)."""
        self.epi = EPI_ELU | (EPI_DROPOUT if drop else 0)
        self.p, self.seed, self.mask, self.want_db = (p if drop else 0.0), seed, (mask if drop else None), bool(want_db)
        self.fused, self.db, self.g, self.aggregated = False, None, g, False


def _dx_through_link(cfg, link, dH, W, X):
    """dX = dH @ W for the consumer of a linked layer (tensor to return as the input gradient): with cfg.fuse_dx_epilogue
    the GEMM's epilogue applies the producing layer's ELU'/dropout' (csrc/gemm_nt.hip, EPI) instead of writing dOut and
    running the epilogue-backward kernel over it."""
    if (link is not None and cfg.fuse_dx_epilogue and cfg.gemm_precision == "high" and W.shape[1] % 4 == 0 and X.is_contiguous()
            and X.dtype == torch.float32):   # (only the 3 x bf16 kernel carries the epilogue: no W^T operand is formed otherwise)
        dH = _f32c(dH)
        Wt = _wt_operand(dH, W, cfg)
        if _nt_ok(dH, Wt, cfg):
            dZ, db = gemm_nt_epilogue_bwd(dH, Wt, X, link.epi, p=link.p, seed=link.seed, mask=link.mask, want_db=link.want_db,
                                          cfg=cfg)
            link.fused, link.db = True, db
            return dZ
    return mm_by_transposed(dH, W, cfg)


def _producer_backward(cfg, link, g, out, epi, p, seed, mask, has_bias, dOut, db_out=None):
    """(dH, db) of a fused layer: through the link when its consumer already applied the epilogue's derivative.  db_out: a buffer the
    bias gradient may be written to (it is then the returned db itself -- check identity)."""
    if link is not None and link.fused:
        db, aggregated = link.db, link.aggregated
        link.fused, link.db, link.aggregated = False, None, False
        if aggregated:   # the consumer's backward already ran this layer's SpMM over dZ (two-hop)
            return _f32c(dOut), db
        return spmm_graph(g, _f32c(dOut), transposed=True, cfg=cfg), db
    plain = not (cfg.fold_backward and getattr(g, "fold_ok", False))
    dH, db, _ = layer_backward(g, out, epi, p, seed, mask, has_bias, dOut=dOut, cfg=cfg, db_out=db_out if plain else None)
    return dH, db


class FusedGCNLayer(torch.autograd.Function):
    """out = dropout(ELU(A_hat (X W^T) + b)): GCNConv (network.py:31) + F.elu (:32) + F.dropout (:33) as
    one GEMM + one SpMM with fused epilogue.  `mask` (uint8 [N,H]) injects a dropout pattern for tests."""

    @staticmethod
    def forward(ctx, X, W, b, g, p, training, seed, mask, link_in, link_out, cfg):
        X = _f32c(X)
        Hm = mm_xwt(X, W, cfg)
        epi = EPI_ELU | (EPI_BIAS if b is not None else 0)
        drop = bool(training) and p > 0.0
        if drop:
            epi |= EPI_DROPOUT
        out = spmm_graph(g, Hm, bias=b, epilogue=epi, p=p if drop else 0.0, seed=seed, mask=mask if drop else None, cfg=cfg)
        ctx.save_for_backward(X, W, out, mask if drop else None)
        ctx.g, ctx.p, ctx.drop, ctx.seed, ctx.has_bias, ctx.cfg = g, p, drop, seed, b is not None, cfg
        ctx.link_in, ctx.link_out = link_in, link_out
        ctx.b_ptr = b.data_ptr() if b is not None else None
        if link_out is not None:
            link_out.record(drop, p, seed, mask, b is not None, g=g)
        return out

    @staticmethod
    def backward(ctx, dOut):
        X, W, out, mask = ctx.saved_tensors
        g = ctx.g
        epi = EPI_ELU | (EPI_DROPOUT if ctx.drop else 0)
        cfg = ctx.cfg
        sb = _sink(cfg, ctx.b_ptr) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        dH, db = _producer_backward(cfg, ctx.link_out, g, out, epi, ctx.p if ctx.drop else 0.0, ctx.seed, mask, ctx.has_bias, dOut, db_out=sb)
        if sb is not None and db is sb:
            db = None   # written to the gradient sink
        dW = None
        if ctx.needs_input_grad[1]:
            sW = _sink(cfg, W)
            if sW is not None:
                mm_at_b(dH, X, cfg, out=sW)
            else:
                dW = mm_at_b(dH, X, cfg)
        dX = _dx_through_link(cfg, ctx.link_in, dH, W, X) if ctx.needs_input_grad[0] else None
        return dX, dW, (db if ctx.has_bias else None), None, None, None, None, None, None, None, None


def narrow_input_supported(x, W, cfg=DEFAULT):
    """A first GCN layer can run on the aggregated input: a device input of at most 32 columns that needs no gradient, narrower than
    the layer, shapes the two narrow kernels take."""
    if not (cfg.narrow_input_first and x.is_cuda and x.dim() == 2 and not x.requires_grad and x.shape[0] > 0):
        return False
    K, H = int(x.shape[1]), int(W.shape[0])
    L = _lib.lib()
    return K < H and bool(L.fitgnn_dense_narrow_k_lds_bytes(K, H)) and bool(L.fitgnn_narrow_atb_workspace_bytes(int(x.shape[0]), K, H))


def aggregated_input(g, x, cfg=DEFAULT):
    """A_hat x for a static input x, formed once per (graph, input tensor) and kept on the graph.  The entry HOLDS x (its storage
    cannot be recycled for another tensor while the entry lives) and its version counter (an in-place edit re-forms the product)."""
    cache = getattr(g, "_agg_input", None)
    if cache is None or cache[0] is not x or cache[1] != x._version:
        with torch.no_grad():
            ax = spmm_graph(g, _f32c(x), cfg=cfg.replace(profile=None) if cfg.profile is not None else cfg)
        cache = (x, x._version, ax)
        g._agg_input = cache
    return cache[2]


class FusedGCNLayerAggregatedInput(torch.autograd.Function):
    """out = dropout(ELU((A_hat x) W^T + b)) for a first layer on a narrow static input (network.py:189-204 on QM9: GCNConv(11, 512)
    + F.elu + F.dropout): AX = A_hat x is an argument (ops.aggregated_input), the forward is one pass over the output, the backward
    one pass over the incoming gradient producing dW and db -- the input needs no gradient, so neither dZ nor A_hat^T dZ exist.
    Same values as FusedGCNLayer up to the order of the additions (A (x W^T) = (A x) W^T)."""

    @staticmethod
    def forward(ctx, AX, W, b, p, training, seed, mask, link_out, cfg):
        L = _lib.lib()
        n, K = AX.shape
        H = int(W.shape[0])
        Wc = _f32c(W)
        epi = EPI_ELU | (EPI_BIAS if b is not None else 0)
        drop = bool(training) and p > 0.0
        if drop:
            epi |= EPI_DROPOUT
        out = torch.empty((n, H), dtype=torch.float32, device=AX.device)
        seed_v, epi_v = _seed_arg(seed, epi)
        m = mask if drop else None
        rc = L.fitgnn_dense_narrow_k_f32(_lib.dptr(AX), AX.stride(0), _lib.dptr(Wc), Wc.stride(0), n, K, H, _lib.dptr(b), epi_v,
                                         float(p if drop else 0.0), seed_v, _lib.dptr(m), _lib.dptr(out), out.stride(0),
                                         _lib.stream_ptr(AX.device))
        _lib.check(rc, "fitgnn_dense_narrow_k_f32")
        ctx.save_for_backward(AX, out, m)
        ctx.p, ctx.drop, ctx.seed, ctx.has_bias, ctx.cfg, ctx.link_out, ctx.H = p, drop, seed, b is not None, cfg, link_out, H
        ctx.W_ptr, ctx.b_ptr = W.data_ptr(), (b.data_ptr() if b is not None else None)
        if link_out is not None:   # g=None: the consumer may hand back dZ, never A_hat^T dZ (this layer has no backward SpMM)
            link_out.record(drop, p, seed, mask, b is not None, g=None)
        return out

    @staticmethod
    def backward(ctx, dOut):
        AX, out, mask = ctx.saved_tensors
        L = _lib.lib()
        n, K = AX.shape
        H = ctx.H
        dOut = _f32c(dOut)
        link = ctx.link_out
        prev, epi = out, EPI_ELU | (EPI_DROPOUT if ctx.drop else 0)
        if link is not None and link.fused:   # the consumer already applied this layer's ELU' / dropout': dOut is dZ
            link.fused, link.db, link.aggregated = False, None, False
            prev, epi = None, 0
        seed_v, epi_v = _seed_arg(ctx.seed, epi)
        cfg = ctx.cfg
        sW = _sink(cfg, ctx.W_ptr) if ctx.needs_input_grad[1] else None
        sb = _sink(cfg, ctx.b_ptr) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        dW = sW if sW is not None else torch.empty((H, K), dtype=torch.float32, device=dOut.device)
        db = sb if sb is not None else torch.empty(H, dtype=torch.float32, device=dOut.device)
        wb = int(L.fitgnn_narrow_atb_workspace_bytes(n, K, H))
        work = torch.empty(wb // 4, dtype=torch.float32, device=dOut.device)
        rc = L.fitgnn_narrow_atb_f32(_lib.dptr(dOut), dOut.stride(0), _lib.dptr(prev), epi_v, float(ctx.p if ctx.drop else 0.0), seed_v,
                                     _lib.dptr(mask), _lib.dptr(AX), AX.stride(0), n, K, H, _lib.dptr(dW), _lib.dptr(db), _lib.dptr(work), wb,
                                     _lib.stream_ptr(dOut.device))
        _lib.check(rc, "fitgnn_narrow_atb_f32")
        dW = dW if (ctx.needs_input_grad[1] and sW is None) else None
        db = db if (ctx.has_bias and ctx.needs_input_grad[2] and sb is None) else None
        return None, dW, db, None, None, None, None, None, None


class FusedGCNLayerHead(torch.autograd.Function):
    """Last GCN layer + output head in one autograd node:
        out = dropout(ELU(A_hat (X W^T) + b))       (network.py:31-33)
        y   = out Wl^T + bl                          (network.py:34, lt1)
    so that the backward never materialises d(out) = dy @ Wl (a K = num_classes GEMM writing [rows x hidden]):
    the epilogue-backward kernel forms it on the fly.  Requires num_classes <= fitgnn_head_max_classes()."""

    @staticmethod
    def forward(ctx, X, W, b, Wl, bl, g, p, training, seed, mask, link_in, cfg, loss_rows=None):
        """loss_rows (int64 index tensor, optional): the only rows of y that reach the loss -- the gradient dy the backward
        receives is zero elsewhere, which lets the head's weight / bias gradients run over those rows alone, and the head's
        forward too: y is then zero on every other row."""
        X = _f32c(X)
        ctx.link_in, ctx.cfg, ctx.loss_rows = link_in, cfg, loss_rows
        Hm = mm_xwt(X, W, cfg)
        epi = EPI_ELU | (EPI_BIAS if b is not None else 0)
        drop = bool(training) and p > 0.0
        if drop:
            epi |= EPI_DROPOUT
        out = spmm_graph(g, Hm, bias=b, epilogue=epi, p=p if drop else 0.0, seed=seed, mask=mask if drop else None, cfg=cfg)
        if loss_rows is not None and head_rows_supported(out, Wl):
            # rows that never reach the loss are not evaluated (they read as zero): one pass over the kept rows
            y = head_rows(out, loss_rows, Wl, bl)
        else:
            y = torch.mm(out, Wl.t())
            if bl is not None:
                y = y + bl
        ctx.save_for_backward(X, W, Wl, out, mask if drop else None)
        ctx.g, ctx.p, ctx.drop, ctx.seed, ctx.has_bias, ctx.has_bl = g, p, drop, seed, b is not None, bl is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        X, W, Wl, out, mask = ctx.saved_tensors
        g = ctx.g
        dy = _f32c(dy)
        epi = EPI_ELU | (EPI_DROPOUT if ctx.drop else 0)
        cfg = ctx.cfg
        dH, db, dWl = layer_backward(g, out, epi, ctx.p if ctx.drop else 0.0, ctx.seed, mask, ctx.has_bias, dy=dy, Wl=Wl,
                                     want_dWl=ctx.needs_input_grad[3], cfg=cfg, loss_rows=ctx.loss_rows)
        # [R, C] column sums: torch's dim-0 reduction of a tall 3-column matrix takes 50 us, a transposed copy + dim-1
        # reduction 23, the two-pass kernel 9
        dy_l = dy if ctx.loss_rows is None else dy.index_select(0, ctx.loss_rows)
        dbl = colsum_narrow(dy_l) if ctx.has_bl and ctx.needs_input_grad[4] else None
        dW = mm_at_b(dH, X, cfg) if ctx.needs_input_grad[1] else None
        dX = _dx_through_link(cfg, ctx.link_in, dH, W, X) if ctx.needs_input_grad[0] else None
        return dX, dW, (db if ctx.has_bias else None), dWl, dbl, None, None, None, None, None, None, None, None


def _arange_rows(g, n, device):
    """int64 [n] = 0..n-1, cached on the graph (row ids of a compact matrix)."""
    cache = getattr(g, "_arange_rows", None)
    if cache is None or cache.numel() != n or cache.device != device:
        cache = torch.arange(n, dtype=torch.int64, device=device)
        g._arange_rows = cache
    return cache


class FusedGCNLastLayerRows(torch.autograd.Function):
    """FusedGCNLayerHead when only `rows` of the result reach the loss (run.py:193-204 keeps out[mask]; with --extra_node 2 % of a
    union's rows), evaluated aggregate-first:
        AH = A_hat X                              every row, every edge (the SpMM of the layer)
        out[rows] = dropout(ELU(AH[rows] W^T + b)) ; y[rows] = out[rows] Wl^T + bl ; y = 0 elsewhere
    (A X) W^T = A (X W^T) (network.py:31: GCNConv applies the Linear first; same values up to fp32 rounding), and a row outside
    `rows` feeds nothing downstream, so the dense part -- the layer's GEMM, its epilogue, the head and, in the backward,
        dW = dZ^T AH[rows],  dAH[rows] = dZ W   (dZ is zero outside `rows`: both products run over len(rows) rows)
    -- touches len(rows) rows instead of all of them; dX = A_hat^T dAH reads dAH in compact form through a row indirection
    (every edge aggregated, as in layer_backward's compact path)."""

    @staticmethod
    def forward(ctx, X, W, b, Wl, bl, g, p, training, seed, mask, rows, cfg, link_in=None, compact_out=False, fwd_sub=None):
        """compact_out: return the logits of the kept rows only, [len(rows), C] in the order of `rows`, instead of an [R, C] matrix
        that is zero elsewhere (the caller's loss then runs on them directly: no [R, C] logits, no [R, C] gradient).
        fwd_sub (csr.RowSubset over `rows`, optional): the forward aggregation itself on the kept rows only, A_hat[rows, :] X -- the
        pruned step (GDTrainer(prune_unused_rows=True)): rows nobody reads are neither aggregated nor stored; the backward is the
        same (its operand is zero outside `rows` either way)."""
        X = _f32c(X)
        rows = rows if rows.dtype == torch.int64 else rows.long()
        ctx.link_in, ctx.compact_out = link_in, bool(compact_out)
        if fwd_sub is not None:
            f = fwd_sub.f
            AHc = spmm_raw(f.rowptr, f.col, f.val, f.tiles, X, fwd_sub.m, window_rows=fwd_sub.window_rows, cfg=cfg, profile_kind="rows_fwd")
        else:
            AH = spmm_graph(g, X, cfg=cfg)                               # [R, K]
            AHc = AH.index_select(0, rows)                               # [n, K]
            del AH
        outc = mm_xwt(AHc, W, cfg)                                       # [n, H]
        if not outc.is_contiguous():
            outc = outc.contiguous()
        epi = EPI_ELU | (EPI_BIAS if b is not None else 0)
        drop = bool(training) and p > 0.0
        if drop:
            epi |= EPI_DROPOUT
        epilogue_fwd_rows_(outc, rows, b, epi, p=p if drop else 0.0, seed=seed, mask=mask if drop else None)
        if _exact(cfg, outc, Wl) and Wl.shape[0] >= 16:
            # (a head of a few classes stays on head_rows_kernel: a 128-column MFMA tile for 3 columns is slower than its LDS-resident weights)
            # the head on the kept rows as one more exact-fp32 product: [n, H] @ [C, H]^T on the fp32 MFMA (a 256 x 128 tile of which C
            # columns are stored: 0.2 ms at S-products against 0.68 ms for head_rows_kernel's LDS-resident weights), bias added after;
            # the [R, C] form is the same values scattered into zeros
            y = gemm_exact(outc, Wl, "nt", cfg)
            if bl is not None:
                y = y + bl
            if not compact_out:
                y = torch.zeros((g.n, y.shape[1]), dtype=torch.float32, device=y.device).index_copy_(0, rows, y)
        elif compact_out:
            y = head_rows(outc, _arange_rows(g, rows.numel(), rows.device), Wl, bl, n_total=rows.numel())
        else:
            y = head_rows(outc, rows, Wl, bl, n_total=g.n)
        # X (the previous layer's output) is kept only when that layer's epilogue backward is applied here, in the SpMM's store
        keep_x = link_in is not None and cfg.fuse_dx_epilogue and X.shape[1] % 4 == 0
        ctx.save_for_backward(W, Wl, AHc, outc, rows, mask if drop else None, X if keep_x else None)
        ctx.g, ctx.p, ctx.drop, ctx.seed, ctx.has_bias, ctx.has_bl, ctx.cfg = g, p, drop, seed, b is not None, bl is not None, cfg
        ctx.ptrs = (W.data_ptr(), b.data_ptr() if b is not None else None, Wl.data_ptr(), bl.data_ptr() if bl is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        W, Wl, AHc, outc, rows, mask, Xprev = ctx.saved_tensors
        g, cfg = ctx.g, ctx.cfg
        L = _lib.lib()
        H, C = outc.shape[1], Wl.shape[0]
        dy_c = _f32c(dy) if ctx.compact_out else _f32c(dy).index_select(0, rows)   # [n, C]
        epi = EPI_ELU | (EPI_DROPOUT if ctx.drop else 0)
        inside = ctx.needs_input_grad[3] and bool(L.fitgnn_epilogue_bwd_head_supported(H, C, 1))
        # weight / bias gradients go straight to the trainer's gradient sink where there is one (OpConfig.grad_sink): no `grad += new`
        pW, pb, pWl, pbl = ctx.ptrs
        sW = _sink(cfg, pW) if ctx.needs_input_grad[1] else None
        sb = _sink(cfg, pb) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        sWl = _sink(cfg, pWl) if inside else None
        sbl = _sink(cfg, pbl) if (ctx.has_bl and ctx.needs_input_grad[4]) else None
        dZc, db, dWl = epilogue_bwd_head_rows_raw(dy_c, Wl, outc, rows, epi, p=ctx.p if ctx.drop else 0.0, seed=ctx.seed, mask=mask,
                                                  want_db=ctx.has_bias, want_dWl=inside, inputs_compact=True, zero_rows=0, db_out=sb,
                                                  dWl_out=sWl)
        if sb is not None:
            db = None
        if sWl is not None:
            dWl = None
        if ctx.needs_input_grad[3] and not inside:
            dWl = head_weight_grad_rows(dy_c, outc, cfg)
        dbl = None
        if ctx.has_bl and ctx.needs_input_grad[4]:
            dbl = colsum_narrow(dy_c, out=sbl)
            if sbl is not None and dbl is sbl:
                dbl = None
        dW = None
        if ctx.needs_input_grad[1]:   # [H, K]
            if sW is not None:
                mm_at_b(dZc, AHc, cfg, out=sW)
            else:
                dW = mm_at_b(dZc, AHc, cfg)
        dX = None
        if ctx.needs_input_grad[0]:
            n, K = AHc.shape
            dAH = torch.empty((n + ZERO_ROWS, K), dtype=torch.float32, device=dZc.device)
            dAH[n:].zero_()
            mm_by_transposed(dZc, W, cfg, out=dAH[:n])                   # dZ @ W on the loss rows, stored in place
            link = ctx.link_in
            if Xprev is not None and two_hop_supported(g, link, dAH, Xprev, cfg):
                # ... and the producing layer's own backward SpMM over that dZ in the same pass: what travels back is A_hat^T dZ
                dX, db_prev = spmm_two_hop_blocks(g, dAH, Xprev, rows, _compact_positions(g, rows), link, cfg=cfg)
                link.fused, link.db, link.aggregated = True, db_prev, True
            elif Xprev is not None:
                # the producing layer's ELU' / dropout' applied as the rows are stored: what travels back on this edge is its dZ
                dX, db_prev = spmm_graph_dz(g, dAH, Xprev, link.epi, p=link.p, seed=link.seed, mask=link.mask, want_db=link.want_db,
                                            xrow=_compact_positions(g, rows), cfg=cfg, profile_kind="compact_dz", zero_from=n)
                link.fused, link.db = True, db_prev
            else:
                dX = spmm_graph(g, dAH, transposed=True, cfg=cfg, xrow=_compact_positions(g, rows), profile_kind="compact", zero_from=n)
        return dX, dW, (db if ctx.has_bias else None), dWl, dbl, None, None, None, None, None, None, None, None, None, None


class FusedGCNLayerRows(torch.autograd.Function):
    """A fused GCN layer of which only `rows` are read downstream, evaluated aggregate-first on those rows, output COMPACT:
        out_c[i] = dropout(ELU((A_hat X)[rows[i]] W^T + b))          [len(rows), H]
    -- the last conv layer of the *_graph_gs models, whose pool reads x[mask] (network.py:129-131, :200-202: with --extra_node about
    half of a subgraph union's rows).  (A X) W^T = A (X W^T) (network.py:31 applies the Linear first; same values up to fp32
    rounding): the layer's product, its epilogue and, in the backward, both weight-side products run over len(rows) rows instead of
    all of them; dX = A_hat^T dAH reads the compact gradient through a row indirection (every edge aggregated).  FusedGCNLastLayerRows
    is this node with the output head of the node-level models behind it.  link_in: X is the un-shared output of a fused layer whose
    ELU' / dropout' is then applied in the backward SpMM's store (what travels back on that edge is its dZ)."""

    @staticmethod
    def forward(ctx, X, W, b, g, p, training, seed, mask, rows, cfg, link_in=None):
        X = _f32c(X)
        rows = rows if rows.dtype == torch.int64 else rows.long()
        AHc = spmm_graph(g, X, cfg=cfg).index_select(0, rows)          # [n, K]
        outc = mm_xwt(AHc, W, cfg)
        if not outc.is_contiguous():
            outc = outc.contiguous()
        epi = EPI_ELU | (EPI_BIAS if b is not None else 0)
        drop = bool(training) and p > 0.0
        if drop:
            epi |= EPI_DROPOUT
        epilogue_fwd_rows_(outc, rows, b, epi, p=p if drop else 0.0, seed=seed, mask=mask if drop else None)
        keep_x = link_in is not None and cfg.fuse_dx_epilogue and X.shape[1] % 4 == 0
        ctx.save_for_backward(W, AHc, outc, rows, mask if drop else None, X if keep_x else None)
        ctx.g, ctx.p, ctx.drop, ctx.seed, ctx.has_bias, ctx.cfg, ctx.link_in = g, p, drop, seed, b is not None, cfg, link_in
        ctx.ptrs = (W.data_ptr(), b.data_ptr() if b is not None else None)
        return outc

    @staticmethod
    def backward(ctx, dOutc):
        W, AHc, outc, rows, mask, Xprev = ctx.saved_tensors
        g, cfg = ctx.g, ctx.cfg
        epi = EPI_ELU | (EPI_DROPOUT if ctx.drop else 0)
        pW, pb = ctx.ptrs
        sW = _sink(cfg, pW) if ctx.needs_input_grad[1] else None
        sb = _sink(cfg, pb) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        dZc, db = epilogue_bwd_rows_raw(dOutc, outc, rows, epi, p=ctx.p if ctx.drop else 0.0, seed=ctx.seed, mask=mask, want_db=ctx.has_bias,
                                        db_out=sb)
        if sb is not None:
            db = None
        dW = None
        if ctx.needs_input_grad[1]:
            if sW is not None:
                mm_at_b(dZc, AHc, cfg, out=sW)
            else:
                dW = mm_at_b(dZc, AHc, cfg)
        dX = None
        if ctx.needs_input_grad[0]:
            n, K = AHc.shape
            # rows [n, n + ZERO_ROWS) stand for the rows outside the selection: every kernel that takes zero_from treats them as zeros
            # WITHOUT loading them (staged as zeros / read from a row of zeros in LDS / multiplied from registers), so they are not filled
            dAH = torch.empty((n + ZERO_ROWS, K), dtype=torch.float32, device=dZc.device)
            mm_by_transposed(dZc, W, cfg, out=dAH[:n])
            link = ctx.link_in
            pos = _compact_positions(g, rows)
            if Xprev is not None:
                dX, db_prev = spmm_graph_dz(g, dAH, Xprev, link.epi, p=link.p, seed=link.seed, mask=link.mask, want_db=link.want_db, xrow=pos,
                                            cfg=cfg, profile_kind="compact_dz", zero_from=n)
                link.fused, link.db = True, db_prev
            else:
                dX = spmm_graph(g, dAH, transposed=True, cfg=cfg, xrow=pos, profile_kind="compact", zero_from=n)
        return dX, dW, (db if ctx.has_bias else None), None, None, None, None, None, None, None, None


class FusedGCNLayerDedup(torch.autograd.Function):
    """First GCN layer on a de-duplicated feature table: the union batch's rows are copies of N0 original nodes
    (X_union = Xt[index]), so  X_union W^T = (Xt W^T)[index]: one GEMM on N0 rows, the copies are resolved by the
    SpMM kernel's row indirection (never materialised); backward sums dH over each node's copies
    (fitgnn_segment_sum_f32) before the weight-gradient GEMM.  Same arithmetic as FusedGCNLayer on X_union."""

    @staticmethod
    def forward(ctx, Xt, W, b, g, ridx, p, training, seed, mask, link_out, cfg):
        Xt = _f32c(Xt)
        ctx.link_out, ctx.cfg = link_out, cfg
        ctx.wide = _padded_table_path(Xt, W, cfg)
        # exact fp32 on a feature width that is not a multiple of 4 (1 433, 8 415): rows of Xt / W are not 16-byte aligned --
        # both products run against the zero-padded copy of the static table (and a zero-padded W: a copy of the weight per call)
        ctx.exact_pad = (cfg.gemm_precision == "exact" and Xt.is_cuda and Xt.shape[1] % 4 != 0 and not Xt.requires_grad
                         and W.shape[0] % 4 == 0)
        if ctx.exact_pad:
            Xp = padded_table(Xt)
            Ht = gemm_exact(Xp, padded_weight(W, Xp.shape[1]), "nt", cfg)
        else:
            Ht = gemm_nt_padded_k(padded_table(Xt), W, cfg) if ctx.wide else mm_xwt(Xt, W, cfg)  # [N0, H]
        epi = EPI_ELU | (EPI_BIAS if b is not None else 0)
        drop = bool(training) and p > 0.0
        if drop:
            epi |= EPI_DROPOUT
        # the direct-gather SpMM variant: its operand table stays in L2 / MALL (5-12 us per S-pubmed step over the LDS-window
        # kernel, same bits)
        out = spmm_graph(g, Ht, bias=b, epilogue=epi | (_lib.SPMM_GATHER if cfg.dedup_gather else 0), p=p if drop else 0.0, seed=seed,
                         mask=mask if drop else None, xrow=ridx.index, cfg=cfg)
        ctx.save_for_backward(Xt, W, out, mask if drop else None)
        ctx.g, ctx.ridx, ctx.p, ctx.drop, ctx.seed, ctx.has_bias = g, ridx, p, drop, seed, b is not None
        if link_out is not None:
            link_out.record(drop, p, seed, mask, b is not None, g=g)
        return out

    @staticmethod
    def backward(ctx, dOut):
        Xt, W, out, mask = ctx.saved_tensors
        g, ridx = ctx.g, ctx.ridx
        epi = EPI_ELU | (EPI_DROPOUT if ctx.drop else 0)
        cfg = ctx.cfg
        dH, db = _producer_backward(cfg, ctx.link_out, g, out, epi, ctx.p if ctx.drop else 0.0, ctx.seed, mask, ctx.has_bias, dOut)  # [R, H]
        dHt = segment_sum(ridx.seg_off, ridx.members, dH, ridx.n_table)  # [N0, H] per original node
        if ctx.exact_pad and ctx.needs_input_grad[1]:
            dW = gemm_exact(_f32c(dHt), padded_table(Xt), "tn", cfg)[:, : Xt.shape[1]]
        elif ctx.wide and ctx.needs_input_grad[1]:
            dW = gemm_atb(_f32c(dHt), padded_table(Xt), cfg)[:, : Xt.shape[1]]   # [H, F'] on the padded table, F' - F zero columns dropped
        elif ctx.needs_input_grad[1]:
            sW = _sink(cfg, W)
            if sW is not None:
                mm_at_b(dHt, Xt, cfg, out=sW)
                dW = None
            else:
                dW = mm_at_b(dHt, Xt, cfg)
        else:
            dW = None
        dXt = mm(dHt, W) if ctx.needs_input_grad[0] else None
        return dXt, dW, (db if ctx.has_bias else None), None, None, None, None, None, None, None, None


class GATAggregate(torch.autograd.Function):
    """out = softmax-attention aggregation of h over the CSR graph g (GATConv.propagate, heads = 1) + bias.
    Inputs h [N,C], att_src [C], att_dst [C], bias [C] | None.  g must be a 'sum'-valued pattern with self loops
    (CSRGraph(mode='gat')); attention weights replace its values."""

    @staticmethod
    def forward(ctx, h, att_src, att_dst, bias, g, slope, act, p, training, seed, mask, cfg, ridx=None, link_out=None):
        """act=True fuses F.elu and F.dropout(p) (network.py:32-33) into the aggregation's epilogue, as for GCNConv.
        link_out (EpilogueLink, with act): the output feeds exactly ONE consumer, which may hand back the gradient of the
        pre-activation (its backward SpMM applies this layer's ELU' / dropout' in its store) and the bias gradient.
        ridx (RowIndex, optional): h is a de-duplicated TABLE [N0, C] and row r of the graph is a copy of table row ridx.index[r] (the
        first layer of a union batch: h = x W^T and the score dots run on the N0 original nodes; the aggregation, the SDDMM and the
        softmax read the table through the row indirection; the backward sums each node's copies before the weight-side products)."""
        L = _lib.lib()
        h = _f32c(h)
        nh, C = h.shape
        n = g.n
        dev = h.device
        st = _lib.stream_ptr(dev)
        a_src = torch.empty(nh, dtype=torch.float32, device=dev)
        a_dst = torch.empty(nh, dtype=torch.float32, device=dev)
        att_src, att_dst = _f32c(att_src.reshape(-1)), _f32c(att_dst.reshape(-1))
        with _timed(cfg, "gat_scores"):
            _lib.check(L.fitgnn_gat_scores_f32(_lib.dptr(h), C, nh, C, _lib.dptr(att_src), _lib.dptr(att_dst), _lib.dptr(a_src),
                                               _lib.dptr(a_dst), st), "gat_scores")
            if ridx is not None:   # per union row
                a_src, a_dst = a_src.index_select(0, ridx.index.long()), a_dst.index_select(0, ridx.index.long())
        alpha = torch.empty(g.nnz, dtype=torch.float32, device=dev)
        with _timed(cfg, "gat_edge_softmax"):
            _lib.check(L.fitgnn_gat_edge_softmax_f32(_lib.dptr(g.f.rowptr), _lib.dptr(g.f.col), _lib.dptr(a_src), _lib.dptr(a_dst),
                                                     float(slope), n, _lib.dptr(alpha), st), "gat_edge_softmax")
        epi = EPI_BIAS if bias is not None else 0
        drop = bool(act) and bool(training) and p > 0.0
        if act:
            epi |= EPI_ELU | (EPI_DROPOUT if drop else 0)
        # (a union whose runs go to the whole-subgraph kernel takes it here too: the attention weights are the launch's CSR values)
        out = spmm_graph(g, h, val=alpha, bias=bias, epilogue=epi, p=p if drop else 0.0, seed=seed, mask=mask if drop else None, cfg=cfg,
                         profile_kind="gat_aggregate", xrow=None if ridx is None else ridx.index)
        ctx.save_for_backward(h, att_src, att_dst, a_src, a_dst, alpha, out if act else None, mask if drop else None)
        ctx.g, ctx.slope, ctx.has_bias, ctx.cfg, ctx.ridx = g, slope, bias is not None, cfg, ridx
        ctx.act, ctx.drop, ctx.p, ctx.seed = bool(act), drop, p, seed
        ctx.link_out = link_out if act else None
        if ctx.link_out is not None:   # g=None: the consumer may hand back dZ, never an aggregated form of it
            link_out.record(drop, p, seed, mask, bias is not None, g=None)
        return out

    @staticmethod
    def backward(ctx, dOut):
        h, att_src, att_dst, a_src, a_dst, alpha, out, mask = ctx.saved_tensors
        g, L = ctx.g, _lib.lib()
        dOut = _f32c(dOut)
        db_fused = None
        cfg = ctx.cfg
        link = ctx.link_out
        if ctx.act and link is not None and link.fused:   # the consumer already applied this layer's ELU' / dropout': dOut is dZ
            db_fused = link.db
            link.fused, link.db, link.aggregated = False, None, False
        elif ctx.act:   # through ELU / dropout first: dOut becomes the gradient of the pre-activation, db its column sums
            with _timed(cfg, "gat_epilogue_bwd"):
                dOut, db_fused = epilogue_bwd_raw(dOut, out, EPI_ELU | (EPI_DROPOUT if ctx.drop else 0), p=ctx.p if ctx.drop else 0.0,
                                                  seed=ctx.seed, mask=mask, want_db=ctx.has_bias)
        C = h.shape[1]
        n = g.n
        ridx = ctx.ridx
        dev = h.device
        st = _lib.stream_ptr(dev)
        dalpha = torch.empty_like(alpha)
        # (with a table operand the entries' columns are its rows: index[col], listed once per graph and index)
        ecol = g.f.col if ridx is None else _entry_rows(g.f, ridx.index)
        with _timed(cfg, "gat_sddmm"):
            _lib.check(L.fitgnn_sddmm_csr_f32(_lib.dptr(g.f.rowptr), _lib.dptr(ecol), _lib.dptr(dOut), C, _lib.dptr(h), C, n, C,
                                              _lib.dptr(dalpha), st), "sddmm")
        ds = torch.empty_like(alpha)
        da_dst = torch.empty(n, dtype=torch.float32, device=dev)
        with _timed(cfg, "gat_softmax_bwd"):
            _lib.check(L.fitgnn_gat_softmax_bwd_f32(_lib.dptr(g.f.rowptr), _lib.dptr(g.f.col), _lib.dptr(a_src), _lib.dptr(a_dst),
                                                    _lib.dptr(alpha), _lib.dptr(dalpha), float(ctx.slope), n, _lib.dptr(ds),
                                                    _lib.dptr(da_dst), st), "gat_softmax_bwd")
        with _timed(cfg, "gat_transpose_edges"):
            ds_t = ds[g._perm_t].contiguous()
            da_src = torch.empty(n, dtype=torch.float32, device=dev)
            _lib.check(L.fitgnn_csr_row_sum_f32(_lib.dptr(g.t.rowptr), _lib.dptr(ds_t), n, _lib.dptr(da_src), st), "csr_row_sum")
            # dh = A_alpha^T dOut + da_src (x) att_src + da_dst (x) att_dst
            alpha_t = alpha[g._perm_t].contiguous()
        dh = spmm_graph(g, dOut, transposed=True, val=alpha_t, cfg=ctx.cfg, profile_kind="gat_aggregate_t")
        with _timed(cfg, "gat_rank1"):
            da = torch.stack([da_src, da_dst], dim=1)                              # [rows, 2]
            if ridx is not None:   # every node's copies summed: the table's gradient and its two score gradients
                dh = segment_sum(ridx.seg_off, ridx.members, dh, ridx.n_table)
                da = segment_sum(ridx.seg_off, ridx.members, da, ridx.n_table)   # (fixed order; index_add_'s atomics are not)
            # dh += da_src (x) att_src + da_dst (x) att_dst: one rank-2 pass over dh
            dh.addmm_(da, torch.stack([att_src, att_dst], dim=0))
        # d(att) = h^T da: both vectors in ONE tall-skinny product through the split-K path (two rocBLAS gemv calls on
        # [R x C]^T took 1.5 ms each on the S-pubmed union)
        datt = mm_at_b(da, h, ctx.cfg) if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) else None
        datt_src = datt[0] if ctx.needs_input_grad[1] else None
        datt_dst = datt[1] if ctx.needs_input_grad[2] else None
        if ctx.act:
            db = db_fused if ctx.has_bias else None
        else:
            db = dOut.sum(0) if ctx.has_bias and ctx.needs_input_grad[3] else None
        return dh, datt_src, datt_dst, db, None, None, None, None, None, None, None, None, None, None


def _gat_rank2_csr(g, rows, pos):
    """The transposed pattern of g with TWO extra entries per row, for the adjoint aggregation of FusedGATLastLayerRows with the
    scores' rank-2 term in the same launch: dX[r] = sum_e alpha_e dAX[.] + da_src[r] u_src + da_dst[r] u_dst -- the two vectors u ride
    as operand rows n, n + 1 of the compact operand (zero rows from n + 2 on) and the two extra entries of row r carry da_src[r],
    da_dst[r].  Static per (graph, loss rows): (rowptr_aug int32 [R + 1], xcol_aug int32 [nnz + 2 R], main int64 [nnz] = where the
    pattern's own entries sit, ext int64 [R] = where row r's first extra entry sits); cached on the graph."""
    cache = getattr(g, "_gat_rank2", None)
    if not _same_index(cache, rows):
        side, dev, R, n = g.t, rows.device, g.n, int(rows.numel())
        rp = side.rowptr.long()
        ar = torch.arange(R + 1, device=dev)
        rp_aug = rp + 2 * ar
        cnt = (rp[1:] - rp[:-1])
        row_e = torch.repeat_interleave(ar[:-1], cnt)
        main = torch.arange(int(rp[-1]), device=dev) + 2 * row_e                     # entry e of row r moves behind 2 r extras
        ext = rp_aug[1:] - 2                                                          # the last two slots of every row
        # operand rows: loss rows 0..n-1, then u_src, u_dst (n, n + 1), then the zero rows: a non-loss row r reads n + 2 + r % ZERO_ROWS
        posz = pos.long()
        posz = torch.where(posz >= n, posz + 2, posz)
        xcol_aug = torch.empty(int(rp_aug[-1]), dtype=torch.int32, device=dev)
        xcol_aug[main] = posz.index_select(0, side.col.long()).to(torch.int32)
        xcol_aug[ext] = n
        xcol_aug[ext + 1] = n + 1
        cache = (rows, rows._version, (rp_aug.to(torch.int32).contiguous(), xcol_aug, main, ext))
        g._gat_rank2 = cache
    return cache[2]


class FusedGATLastLayerRows(torch.autograd.Function):
    """The last GATConv layer + ELU + dropout + head when only `rows` of the result reach the loss (run.py:193-204 keeps out[mask]),
    evaluated aggregate-first like FusedGCNLastLayerRows: the attention scores need no transformed features,
        a_src = h . att_src = x . (W^T att_src),   a_dst likewise                (two dots per row on x, fitgnn_gat_scores_f32)
        out_i = sum_j alpha_ij (x_j W^T) + b = (sum_j alpha_ij x_j) W^T + b      (GATConv, heads = 1: network.py:13-17)
    so the aggregation runs on x over every row and edge, and the dense part -- x W^T, bias, ELU, dropout, the head and, in the
    backward, both weight-side products -- on the loss rows alone (the generic path runs three [R x K x H] products; with --extra_node
    unions 2 % of the rows reach the loss).  The backward's edge passes (SDDMM, softmax / LeakyReLU backward) touch the loss rows'
    entries only (fitgnn_sddmm_csr_rows_f32, fitgnn_gat_softmax_bwd_rows_f32) and the adjoint aggregation reads the compact gradient
    through the row-streaming kernel.  g: CSRGraph(mode='gat')."""

    @staticmethod
    def forward(ctx, X, W, att_src, att_dst, b, Wl, bl, g, slope, p, training, seed, mask, rows, cfg, compact_out=False, link_in=None):
        L = _lib.lib()
        X = _f32c(X)
        R, K = X.shape
        dev, st = X.device, _lib.stream_ptr(X.device)
        rows = rows if rows.dtype == torch.int64 else rows.long()
        att2 = torch.stack([_f32c(att_src.reshape(-1)), _f32c(att_dst.reshape(-1))], dim=0)     # [2, H]
        u = torch.mm(att2, W).contiguous()                                                        # [2, K]: W^T att
        a_src = torch.empty(R, dtype=torch.float32, device=dev)
        a_dst = torch.empty(R, dtype=torch.float32, device=dev)
        with _timed(cfg, "gat_scores"):
            _lib.check(L.fitgnn_gat_scores_f32(_lib.dptr(X), X.stride(0), R, K, _lib.dptr(u[0]), _lib.dptr(u[1]), _lib.dptr(a_src), _lib.dptr(a_dst),
                                               st), "gat_scores")
        alpha = torch.empty(g.nnz, dtype=torch.float32, device=dev)
        with _timed(cfg, "gat_edge_softmax"):
            _lib.check(L.fitgnn_gat_edge_softmax_f32(_lib.dptr(g.f.rowptr), _lib.dptr(g.f.col), _lib.dptr(a_src), _lib.dptr(a_dst), float(slope), R,
                                                     _lib.dptr(alpha), st), "gat_edge_softmax")
        AX = spmm_graph(g, X, val=alpha, cfg=cfg, profile_kind="gat_aggregate")                   # [R, K]: every row, every edge
        AXc = AX.index_select(0, rows)
        del AX
        outc = mm_xwt(AXc, W, cfg)
        if not outc.is_contiguous():
            outc = outc.contiguous()
        epi = EPI_ELU | (EPI_BIAS if b is not None else 0)
        drop = bool(training) and p > 0.0
        if drop:
            epi |= EPI_DROPOUT
        epilogue_fwd_rows_(outc, rows, b, epi, p=p if drop else 0.0, seed=seed, mask=mask if drop else None)
        if _exact(cfg, outc, Wl) and Wl.shape[0] >= 16:
            y = gemm_exact(outc, Wl, "nt", cfg)
            if bl is not None:
                y = y + bl
            if not compact_out:
                y = torch.zeros((R, y.shape[1]), dtype=torch.float32, device=dev).index_copy_(0, rows, y)
        elif compact_out:
            y = head_rows(outc, _arange_rows(g, rows.numel(), dev), Wl, bl, n_total=rows.numel())
        else:
            y = head_rows(outc, rows, Wl, bl, n_total=R)
        ctx.save_for_backward(X, W, att2, Wl, AXc, outc, rows, mask if drop else None, alpha, a_src, a_dst, u)
        ctx.g, ctx.slope, ctx.p, ctx.drop, ctx.seed, ctx.has_bias, ctx.has_bl, ctx.cfg, ctx.compact_out = (
            g, slope, p, drop, seed, b is not None, bl is not None, cfg, bool(compact_out))
        ctx.link_in = link_in   # X is the un-shared output of a fused layer: its ELU' / dropout' may go into the adjoint aggregation's store
        return y

    @staticmethod
    def backward(ctx, dy):
        X, W, att2, Wl, AXc, outc, rows, mask, alpha, a_src, a_dst, u = ctx.saved_tensors
        g, cfg, L = ctx.g, ctx.cfg, _lib.lib()
        dev, st = X.device, _lib.stream_ptr(X.device)
        R, K = X.shape
        H, C = outc.shape[1], Wl.shape[0]
        n = int(rows.numel())
        dy_c = _f32c(dy) if ctx.compact_out else _f32c(dy).index_select(0, rows)
        epi = EPI_ELU | (EPI_DROPOUT if ctx.drop else 0)
        inside = ctx.needs_input_grad[5] and bool(L.fitgnn_epilogue_bwd_head_supported(H, C, 1))
        dZc, db, dWl = epilogue_bwd_head_rows_raw(dy_c, Wl, outc, rows, epi, p=ctx.p if ctx.drop else 0.0, seed=ctx.seed, mask=mask,
                                                  want_db=ctx.has_bias, want_dWl=inside, inputs_compact=True, zero_rows=0)
        if ctx.needs_input_grad[5] and not inside:
            dWl = head_weight_grad_rows(dy_c, outc, cfg)
        dbl = colsum_narrow(dy_c) if ctx.has_bl and ctx.needs_input_grad[6] else None
        dW = mm_at_b(dZc, AXc, cfg)                                                               # [H, K]
        # the compact gradient of the aggregation's output: rows 0..n-1 = dZ W on the loss rows; with the rank-2 term folded into the
        # adjoint aggregation (below) rows n, n + 1 carry u = W^T att, and the zero rows follow
        fold = bool(ctx.needs_input_grad[0]) and cfg.compact_rows_kernel and K % 4 == 0
        lead = n + 2 if fold else n
        dAX = torch.empty((lead + ZERO_ROWS, K), dtype=torch.float32, device=dev)
        dAX[lead:].zero_()
        if fold:
            dAX[n:n + 2] = u
        mm_by_transposed(dZc, W, cfg, out=dAX[:n])                                                # dZ W on the loss rows
        # the aggregation's backward, on the loss rows' entries: d(alpha), then the softmax / LeakyReLU backward
        dalpha = torch.zeros_like(alpha)
        with _timed(cfg, "gat_sddmm"):
            _lib.check(L.fitgnn_sddmm_csr_rows_f32(_lib.dptr(g.f.rowptr), _lib.dptr(g.f.col), _lib.dptr(dAX), K, _lib.dptr(X), X.stride(0),
                                                   _lib.dptr(rows), n, K, _lib.dptr(dalpha), st), "sddmm_rows")
        ds = torch.zeros_like(alpha)
        da_dst = torch.zeros(R, dtype=torch.float32, device=dev)
        with _timed(cfg, "gat_softmax_bwd"):
            _lib.check(L.fitgnn_gat_softmax_bwd_rows_f32(_lib.dptr(g.f.rowptr), _lib.dptr(g.f.col), _lib.dptr(a_src), _lib.dptr(a_dst), _lib.dptr(alpha),
                                                         _lib.dptr(dalpha), float(ctx.slope), _lib.dptr(rows), n, _lib.dptr(ds), _lib.dptr(da_dst),
                                                         st), "gat_softmax_bwd_rows")
        with _timed(cfg, "gat_transpose_edges"):
            ds_t = ds[g._perm_t].contiguous()
            da_src = torch.empty(R, dtype=torch.float32, device=dev)
            _lib.check(L.fitgnn_csr_row_sum_f32(_lib.dptr(g.t.rowptr), _lib.dptr(ds_t), R, _lib.dptr(da_src), st), "csr_row_sum")
            alpha_t = alpha[g._perm_t].contiguous()
        da = torch.stack([da_src, da_dst], dim=1)                                                 # [R, 2]
        du = mm_at_b(da, X, cfg)                                                                  # [2, K] = da^T x
        dX = None
        if fold:
            # the adjoint aggregation AND the scores' rank-2 term  + da_src (x) W^T att_src + da_dst (x) W^T att_dst  in one launch of
            # the row-streaming kernel: u's two rows ride in the compact operand, da in two extra entries per row (_gat_rank2_csr)
            rp_aug, xcol_aug, main, ext = _gat_rank2_csr(g, rows, _compact_positions(g, rows))
            op = dAX
            with _timed(cfg, "gat_rank1"):
                val_aug = torch.empty(int(xcol_aug.numel()), dtype=torch.float32, device=dev)
                val_aug[main] = alpha_t
                val_aug[ext] = da_src
                val_aug[ext + 1] = da_dst
            dX = torch.empty((R, K), dtype=torch.float32, device=dev)
            link = ctx.link_in
            with _timed(cfg, "gat_aggregate_t"):
                if link is not None and X.is_contiguous():
                    # ... with the producing layer's ELU' / dropout' applied as the rows are stored (X is that layer's output): what travels
                    # back is its dZ and its bias gradient -- the 50-GB elementwise pass over [R x K] of its own backward is not run
                    n_part = int(L.fitgnn_spmm_rows_compact_parts(R))
                    part = torch.empty((n_part, K), dtype=torch.float32, device=dev) if link.want_db else None
                    seed_v, epi_v = _seed_arg(link.seed, link.epi)
                    _lib.check(L.fitgnn_spmm_rows_compact_dz_f32(_lib.dptr(rp_aug), _lib.dptr(xcol_aug), _lib.dptr(val_aug), int(xcol_aug.numel()),
                                                                 _lib.dptr(op), op.stride(0), n + 2, _lib.dptr(dX), dX.stride(0), R, K, _lib.dptr(X),
                                                                 epi_v, float(link.p), seed_v, _lib.dptr(link.mask), _lib.dptr(part), st),
                               "fitgnn_spmm_rows_compact_dz_f32")
                    db_prev = None
                    if link.want_db:
                        db_prev = torch.empty(K, dtype=torch.float32, device=dev)
                        _lib.check(L.fitgnn_colsum_partials_f32(_lib.dptr(part), n_part, K, _lib.dptr(db_prev), st), "fitgnn_colsum_partials_f32")
                    link.fused, link.db = True, db_prev
                else:
                    _lib.check(L.fitgnn_spmm_rows_compact_f32(_lib.dptr(rp_aug), _lib.dptr(xcol_aug), _lib.dptr(val_aug), int(xcol_aug.numel()),
                                                              _lib.dptr(op), op.stride(0), n + 2, _lib.dptr(dX), dX.stride(0), R, K, st),
                               "fitgnn_spmm_rows_compact_f32")
        elif ctx.needs_input_grad[0]:
            dX = spmm_graph(g, dAX, transposed=True, val=alpha_t, cfg=cfg, xrow=_compact_positions(g, rows), zero_from=n,
                            profile_kind="gat_aggregate_t")
            with _timed(cfg, "gat_rank1"):
                dX.addmm_(da, u)                                                                  # + da_src (x) W^T att_src + da_dst (x) W^T att_dst
        # u = att2 W: the scores' parameter gradients
        dW = dW + torch.mm(att2.t(), du)
        datt2 = torch.mm(du, W.t())                                                               # [2, H]
        return (dX, dW, datt2[0], datt2[1], (db if ctx.has_bias else None), dWl, dbl, None, None, None, None, None, None, None, None, None, None)


class AppnpPlan:
    """Which rows of a block-diagonal batch APPNP propagates inside LDS, and the sub-matrix of the others (built once per graph).

    lds_launches (sliced=True): the diagonal blocks beyond a unit whose CSR slice and two one-slice signal buffers fit LDS
    (fitgnn_appnp_lds_bytes), one launch per (slice width, workgroup size): each block at the widest slice it fits at, 4 / 8 / 16 wavefronts;
    blocks [n_blocks, 2] int32: the diagonal blocks beyond that but within fitgnn_appnp_block_rows() / _entries(), one workgroup each
    (when there are at least MIN_BLOCKS of them);
    units [n_units, 2] int32: runs of whole diagonal blocks (ops csr.block_boundaries: the pattern's own closed blocks = cluster
    subgraphs) of at most fitgnn_appnp_unit_rows(h4) rows and fitgnn_appnp_unit_entries() entries each, packed like SpMM tiles;
    open_rows (int64, ascending): the rows of every other block, with their own CSR in both orientations (rows and columns
    renumbered 0 .. n_open - 1: those blocks are closed too)."""

    MIN_BLOCKS = 128   # fewer subgraphs beyond a unit than this: they stay on the per-step launches

    def __init__(self, g, h4, blocks=True, sliced=True):
        from .csr import block_boundaries, make_tiles
        import numpy as np

        L = _lib.lib()
        cap_r, cap_e = int(L.fitgnn_appnp_unit_rows(int(h4))), int(L.fitgnn_appnp_unit_entries())
        self.cap_rows = cap_r
        dev = g.f.rowptr.device
        n = g.n
        ptr = block_boundaries(g.f.rowptr, g.f.col, n).cpu().numpy().astype(np.int64)
        rp = g.f.rowptr.cpu().numpy().astype(np.int64)
        # units: (i) the blocks of at most PACK rows, packed like SpMM tiles (parallelism: one wavefront per unit -- packing a 90 k-row
        # batch of 9-row subgraphs into 768-row units would leave 118 wavefronts); (ii) every block between PACK and the capacity alone
        PACK = min(64, cap_r)
        tiles = make_tiles(ptr, PACK)
        a, b = tiles["row_begin"].astype(np.int64), tiles["row_end"].astype(np.int64)
        is_b = np.zeros(n + 1, dtype=bool)
        is_b[ptr] = True
        closed = is_b[a] & is_b[b] & (rp[b] - rp[a] <= cap_e) & (b > a)
        size = np.diff(ptr)
        mid = (size > PACK) & (size <= cap_r) & (rp[ptr[1:]] - rp[ptr[:-1]] <= cap_e)
        a = np.concatenate([a[closed], ptr[:-1][mid]])
        b = np.concatenate([b[closed], ptr[1:][mid]])
        order = np.argsort(a, kind="stable")
        a, b = a[order], b[order]
        closed = np.ones(len(a), dtype=bool)
        self.units = torch.from_numpy(np.stack([a, b], 1).astype(np.int32)).to(dev).contiguous()
        self.n_units = int(len(a))
        self.max_rows = int((b - a).max()) if len(a) else 0
        self.max_entries = int((rp[b] - rp[a]).max()) if len(a) else 0
        # blocks: the subgraphs beyond a unit, one workgroup each (fitgnn_appnp_blocks_f32: K steps between two global scratch signals
        # that stay in L2, the block's CSR slice in LDS) -- when there are enough of them to fill the chip; a handful of large
        # blocks is better served by the per-step launches over all of their rows
        blk_r, blk_e = int(L.fitgnn_appnp_block_rows()), int(L.fitgnn_appnp_block_entries())
        ent = rp[ptr[1:]] - rp[ptr[:-1]]
        cover = np.zeros(n + 1, dtype=np.int64)
        np.add.at(cover, a, 1)
        np.add.at(cover, b, -1)
        in_unit = np.cumsum(cover)[np.minimum(ptr[:-1], n - 1)] > 0   # a block is inside a unit or outside all of them
        # sliced: the column-sliced LDS kernel (fitgnn_appnp_lds_f32) runs the units (a few wavefronts each) and, sixteen wavefronts
        # each, every larger subgraph whose CSR slice and two one-slice buffers fit LDS
        self.sliced = bool(sliced)
        pow2 = lambda v: 1 << (int(v).bit_length() - 1)   # noqa: E731
        keep, lds_max = int(L.fitgnn_appnp_lds_items_per_thread()), int(L.fitgnn_appnp_lds_max_bytes())
        self.unit_slice = pow2(min(4, h4))
        self.unit_threads = 64
        if self.n_units:
            self.unit_threads = max(64, -(-self.max_rows * self.unit_slice // (keep * 64)) * 64)
            if self.unit_threads > 1024 or L.fitgnn_appnp_lds_bytes(self.max_rows, self.max_entries, self.unit_slice) > lds_max:
                self.unit_slice, self.unit_threads = 1, max(64, -(-self.max_rows // (keep * 64)) * 64)
        # every larger subgraph that fits LDS a slice at a time goes to the launch of the WIDEST slice it fits at (fewer passes over its
        # K steps, wider global accesses); a launch sizes its LDS by its own largest range
        lds_fit = np.zeros(len(size), dtype=bool)
        self.lds_launches = []   # (slice, ranges [m, 2] int32, m, max_rows, max_entries, threads)
        if sliced:
            cand = ~in_unit & (size > 0) & (size <= keep * 1024)
            for sl in (4, 2, 1):
                if sl > pow2(h4):
                    continue
                todo = cand & ~lds_fit & (size * sl <= keep * 1024)
                need = np.array([L.fitgnn_appnp_lds_bytes(int(r), int(e), sl) if c else 0 for r, e, c in zip(size, ent, todo)], dtype=np.int64)
                fits = todo & (need <= lds_max)
                # by workgroup size too (a thread owns <= `keep` items): a 100-row block among 1 000-row ones would idle 900 threads and
                # take the largest block's LDS
                lo = 0
                for threads in (256, 512, 1024):
                    cls = fits & (size * sl > lo) & (size * sl <= keep * threads)
                    lo = keep * threads
                    # the launch is sized by (most rows, most entries) of its list -- two different blocks: drop the costliest until
                    # that fits (they get the next narrower slice)
                    for b_i in np.argsort(-np.where(cls, need, 0), kind="stable"):
                        if not cls.any() or L.fitgnn_appnp_lds_bytes(int(size[cls].max()), int(ent[cls].max()), sl) <= lds_max:
                            break
                        cls[b_i] = False
                    if cls.any():
                        la, lb = ptr[:-1][cls], ptr[1:][cls]
                        self.lds_launches.append((sl, torch.from_numpy(np.stack([la, lb], 1).astype(np.int32)).to(dev).contiguous(), int(len(la)),
                                                  int((lb - la).max()), int(ent[cls].max()), threads))
                        lds_fit |= cls
        la, lb = ptr[:-1][lds_fit], ptr[1:][lds_fit]
        self.n_lds_blocks = int(len(la))
        self.rows_in_lds_blocks = int((lb - la).sum())
        self.nnz_lds_blocks = int(ent[lds_fit].sum())
        big = ~in_unit & ~lds_fit & (size > 0) & (size <= blk_r) & (ent <= blk_e)
        if not blocks or int(big.sum()) < self.MIN_BLOCKS:
            big[:] = False
        ba, bb = ptr[:-1][big], ptr[1:][big]
        self.blocks = torch.from_numpy(np.stack([ba, bb], 1).astype(np.int32)).to(dev).contiguous() if len(ba) else None
        self.n_blocks = int(len(ba))
        self.block_max_rows = int((bb - ba).max()) if len(ba) else 0
        self.block_max_entries = int(ent[big].max()) if len(ba) else 0
        self.rows_in_blocks = int((bb - ba).sum())
        self.nnz_blocks = int(ent[big].sum())
        row_open = np.ones(n, dtype=bool)
        if self.n_units or self.n_blocks or self.n_lds_blocks:
            cover = np.zeros(n + 1, dtype=np.int64)
            np.add.at(cover, np.concatenate([a, ba, la]), 1)
            np.add.at(cover, np.concatenate([b, bb, lb]), -1)
            row_open = np.cumsum(cover[:-1]) == 0
        self.n_open = int(row_open.sum())
        self.rows_in_units = n - self.n_open - self.rows_in_blocks - self.rows_in_lds_blocks
        self.open_rows, self.sub = None, {}
        if self.n_open:
            open_t = torch.from_numpy(row_open).to(dev)
            self.open_rows = torch.nonzero(open_t).flatten()
            new_id = (torch.cumsum(open_t.to(torch.int64), 0) - 1).to(torch.int32)
            for name, side in (("f", g.f), ("t", g.t)):
                cnt = (side.rowptr[1:] - side.rowptr[:-1]).to(torch.int64)
                cnt_o = cnt[self.open_rows]
                rowptr = torch.zeros(self.n_open + 1, dtype=torch.int64, device=dev)
                rowptr[1:] = torch.cumsum(cnt_o, 0)
                ent = torch.repeat_interleave(open_t, cnt)                    # entries of open rows
                col = new_id[side.col[ent].long()].contiguous()
                self.sub[name] = (rowptr.to(torch.int32).contiguous(), col, None if side.val is None else side.val[ent].contiguous())
            self.nnz_open = int(self.sub["f"][1].numel())
        else:
            self.nnz_open = 0


def appnp_plan(g, h4, blocks=True, sliced=True):
    """The plan for a signal of h4 float4 columns (a unit holds 768 / h4 rows), cached on the graph per (h4, blocks, sliced)."""
    plans = getattr(g, "_appnp_plan", None)
    if plans is None:
        plans = {}
        g._appnp_plan = plans
    key = (int(h4), bool(blocks), bool(sliced))
    if key not in plans:
        plans[key] = AppnpPlan(g, h4, blocks, sliced)
    return plans[key]


class APPNPPropagate(torch.autograd.Function):
    """z_K of  z_{k+1} = (1 - alpha) A_hat z_k + alpha z0,  z_0 = z0  (APPNP's K propagation steps) on a class-wide signal,
    one narrow-SpMM launch per step with the teleport term in its epilogue; the backward pass propagates with A_hat^T and
    accumulates d z0 = alpha * sum_k dz_{k+1} + dz_0 in the same launches.  The K steps run on a copy of the signal whose rows are
    padded to whole float4s (fitgnn_spmm_narrow_padded_f32: a wave packs 64 / h4 rows, operand rows are contiguous 16-byte accesses)."""

    @staticmethod
    def _padded(z, h4, row_index=None):
        """[rows x 4 h4] contiguous: the rows z[row_index.index[r]] (all of z's rows, in order, without an index) padded with zeros --
        one pass (fitgnn_gather_rows_padded_f32); z itself when there is nothing to do."""
        H = z.shape[1]
        if row_index is None and H == 4 * h4 and z.is_contiguous():
            return z
        if z.stride(1) != 1:
            z = z.contiguous()
        idx = None if row_index is None else row_index.index
        n = int(z.shape[0] if idx is None else idx.numel())
        zp = torch.empty((n, 4 * h4), dtype=torch.float32, device=z.device)
        _lib.check(_lib.lib().fitgnn_gather_rows_padded_f32(_lib.dptr(z), z.stride(0), H, _lib.dptr(idx), n, _lib.dptr(zp), h4,
                                                            _lib.stream_ptr(z.device)), "fitgnn_gather_rows_padded_f32")
        return zp

    @staticmethod
    def _padded_grad(dz, h4):
        """The incoming gradient in the padded layout: as it is when it IS a [rows x H] view of a padded buffer (what SoftmaxNLL hands
        back for the view forward() returned: the pad columns never mix with the others, their content does not matter), else padded."""
        n, H = dz.shape
        if dz.dtype != torch.float32:
            dz = dz.float()
        if H == 4 * h4:
            return _f32c(dz)
        if (dz.stride(1) == 1 and dz.stride(0) == 4 * h4 and dz.storage_offset() % 4 == 0
                and dz.untyped_storage().nbytes() >= 4 * (dz.storage_offset() + n * 4 * h4)):
            return dz.as_strided((n, 4 * h4), (4 * h4, 1))
        return APPNPPropagate._padded(dz, h4)

    @staticmethod
    def _in_lds(plan, side, x, out, h4, K, alpha, backward, cfg, st, tag):
        """The launches whose K steps run in LDS: the units (the column-sliced kernel, or the whole-signal one under
        OpConfig(appnp_sliced=False)) and the larger subgraphs that fit LDS a slice at a time."""
        L = _lib.lib()
        if plan.n_units and plan.sliced:
            with _timed(cfg, "appnp_units" + tag):
                _lib.check(L.fitgnn_appnp_lds_f32(_lib.dptr(side.rowptr), _lib.dptr(side.col), _lib.dptr(side.val), _lib.dptr(plan.units),
                                                  plan.n_units, plan.max_rows, plan.max_entries, _lib.dptr(x), _lib.dptr(out), h4, K, float(alpha),
                                                  backward, plan.unit_threads, plan.unit_slice, st), "fitgnn_appnp_lds_f32")
        elif plan.n_units:
            with _timed(cfg, "appnp_units" + tag):
                _lib.check(L.fitgnn_appnp_units_f32(_lib.dptr(side.rowptr), _lib.dptr(side.col), _lib.dptr(side.val), _lib.dptr(plan.units),
                                                    plan.n_units, plan.max_rows, plan.max_entries, _lib.dptr(x), _lib.dptr(out), h4, K, float(alpha),
                                                    backward, st), "fitgnn_appnp_units_f32")
        for sl, ranges, m, max_rows, max_entries, threads in plan.lds_launches:
            with _timed(cfg, "appnp_lds_blocks" + tag):
                _lib.check(L.fitgnn_appnp_lds_f32(_lib.dptr(side.rowptr), _lib.dptr(side.col), _lib.dptr(side.val), _lib.dptr(ranges), m, max_rows,
                                                  max_entries, _lib.dptr(x), _lib.dptr(out), h4, K, float(alpha), backward, threads, sl, st),
                           "fitgnn_appnp_lds_f32")

    @staticmethod
    def forward(ctx, z0, g, K, alpha, cfg=None, row_index=None):
        """row_index (ops.RowIndex, optional): z0 is the class-wide output on a de-duplicated table and the signal's row r is
        z0[row_index.index[r]] -- gathered into the padded layout in one pass, its adjoint (fitgnn_segment_sum_f32) in backward.
        The result is the [rows x H] VIEW of the padded signal (no copy out of it)."""
        L = _lib.lib()
        if z0.dtype != torch.float32:
            z0 = z0.float()
        H = int(z0.shape[1])
        n = int(z0.shape[0] if row_index is None else row_index.index.numel())
        h4 = (H + 3) // 4
        st = _lib.stream_ptr(z0.device)
        f = g.f
        z0p = APPNPPropagate._padded(z0, h4, row_index)
        ctx.row_index = row_index
        plan = (appnp_plan(g, h4, cfg is None or cfg.appnp_blocks, cfg is None or cfg.appnp_sliced)
                if (cfg is None or cfg.appnp_in_lds) and K > 0 and h4 <= 16 else None)
        if plan is not None and plan.n_units == 0 and plan.n_blocks == 0 and plan.n_lds_blocks == 0:
            plan = None
        ctx.g, ctx.K, ctx.alpha, ctx.cfg, ctx.H, ctx.plan = g, K, alpha, cfg, H, plan
        if plan is not None:
            # the subgraphs that fit a wavefront's LDS: all K steps in one launch; the larger ones: one workgroup each, all K steps in
            # one launch between two scratch signals; rows of what is left: the per-step kernel on their own sub-matrix (gathered in,
            # scattered out)
            out = torch.empty_like(z0p)
            APPNPPropagate._in_lds(plan, f, z0p, out, h4, K, alpha, 0, cfg, st, "")
            if plan.n_blocks:
                t1, t2 = torch.empty_like(z0p), torch.empty_like(z0p)
                with _timed(cfg, "appnp_blocks"):
                    _lib.check(L.fitgnn_appnp_blocks_f32(_lib.dptr(f.rowptr), _lib.dptr(f.col), _lib.dptr(f.val), _lib.dptr(plan.blocks),
                                                         plan.n_blocks, plan.block_max_rows, plan.block_max_entries, _lib.dptr(z0p), _lib.dptr(out),
                                                         _lib.dptr(t1), _lib.dptr(t2), h4, K, float(alpha), 0, st), "fitgnn_appnp_blocks_f32")
            if plan.n_open:
                rp, cc, vv = plan.sub["f"]
                zb0 = z0p.index_select(0, plan.open_rows)
                z = zb0
                for _ in range(K):
                    nxt = torch.empty_like(zb0)
                    with _timed(cfg, "appnp_step"):
                        _lib.check(L.fitgnn_spmm_narrow_padded_f32(_lib.dptr(rp), _lib.dptr(cc), _lib.dptr(vv), _lib.dptr(z), _lib.dptr(nxt),
                                                                   plan.n_open, h4, 1.0 - alpha, _lib.dptr(zb0), float(alpha), None, 0.0, st),
                                   "spmm_narrow_padded")
                    z = nxt
                out.index_copy_(0, plan.open_rows, z)
            return out if H == 4 * h4 else out[:, :H]
        z = z0p
        for _ in range(K):
            nxt = torch.empty_like(z0p)
            with _timed(cfg, "appnp_step"):
                _lib.check(L.fitgnn_spmm_narrow_padded_f32(_lib.dptr(f.rowptr), _lib.dptr(f.col), _lib.dptr(f.val), _lib.dptr(z), _lib.dptr(nxt),
                                                           n, h4, 1.0 - alpha, _lib.dptr(z0p), float(alpha), None, 0.0, st), "spmm_narrow_padded")
            z = nxt
        if z is z0p and z0p is z0:
            z = z.clone()   # (K = 0: the input itself)
        return z if H == 4 * h4 else z[:, :H]

    @staticmethod
    def _grad_out(ctx, res, H, h4):
        """res [rows x 4 h4] -> the gradient w.r.t. z0: summed back per table row under a row index, else the [rows x H] view."""
        ri = ctx.row_index
        if ri is None:
            return res if H == 4 * h4 else res[:, :H]
        out = torch.empty((ri.n_table, H), dtype=torch.float32, device=res.device)
        _lib.check(_lib.lib().fitgnn_segment_sum_f32(_lib.dptr(ri.seg_off), _lib.dptr(ri.members), ri.n_table, _lib.dptr(res), 4 * h4, H,
                                                     _lib.dptr(out), H, _lib.stream_ptr(res.device)), "fitgnn_segment_sum_f32")
        return out

    @staticmethod
    def backward(ctx, dz):
        L = _lib.lib()
        n, H = dz.shape
        h4 = (H + 3) // 4
        st = _lib.stream_ptr(dz.device)
        t = ctx.g.t
        dz = APPNPPropagate._padded_grad(dz, h4)
        plan = ctx.plan
        if plan is not None:
            out = torch.empty_like(dz)
            APPNPPropagate._in_lds(plan, t, dz, out, h4, ctx.K, ctx.alpha, 1, ctx.cfg, st, "_t")
            if plan.n_blocks:
                t1, t2 = torch.empty_like(dz), torch.empty_like(dz)
                with _timed(ctx.cfg, "appnp_blocks_t"):
                    _lib.check(L.fitgnn_appnp_blocks_f32(_lib.dptr(t.rowptr), _lib.dptr(t.col), _lib.dptr(t.val), _lib.dptr(plan.blocks),
                                                         plan.n_blocks, plan.block_max_rows, plan.block_max_entries, _lib.dptr(dz), _lib.dptr(out),
                                                         _lib.dptr(t1), _lib.dptr(t2), h4, ctx.K, float(ctx.alpha), 1, st),
                               "fitgnn_appnp_blocks_f32")
            if plan.n_open:
                rp, cc, vv = plan.sub["t"]
                gb = dz.index_select(0, plan.open_rows)
                accb = torch.zeros_like(gb)
                for _ in range(ctx.K):
                    nxt = torch.empty_like(gb)
                    with _timed(ctx.cfg, "appnp_step_t"):
                        _lib.check(L.fitgnn_spmm_narrow_padded_f32(_lib.dptr(rp), _lib.dptr(cc), _lib.dptr(vv), _lib.dptr(gb), _lib.dptr(nxt),
                                                                   plan.n_open, h4, 1.0 - ctx.alpha, None, 0.0, _lib.dptr(accb), float(ctx.alpha), st),
                                   "spmm_narrow_padded")
                    gb = nxt
                out.index_copy_(0, plan.open_rows, accb + gb)
            return APPNPPropagate._grad_out(ctx, out, H, h4), None, None, None, None, None
        acc = torch.zeros_like(dz)
        for _ in range(ctx.K):   # dz_k = (1 - alpha) A^T dz_{k+1};  acc += alpha * dz_{k+1}
            nxt = torch.empty_like(dz)
            with _timed(ctx.cfg, "appnp_step_t"):
                _lib.check(L.fitgnn_spmm_narrow_padded_f32(_lib.dptr(t.rowptr), _lib.dptr(t.col), _lib.dptr(t.val), _lib.dptr(dz), _lib.dptr(nxt),
                                                           n, h4, 1.0 - ctx.alpha, None, 0.0, _lib.dptr(acc), float(ctx.alpha), st),
                           "spmm_narrow_padded")
            dz = nxt
        out = acc + dz
        return APPNPPropagate._grad_out(ctx, out, H, h4), None, None, None, None, None


_HEAD_MAX = None


def head_max_classes():
    """Widest head whose weight gradient the fused head-backward kernel accumulates itself."""
    global _HEAD_MAX
    if _HEAD_MAX is None:
        _HEAD_MAX = int(_lib.lib().fitgnn_head_max_classes())
    return _HEAD_MAX


def head_fusable(H, C):
    """The fused head-backward kernel takes a head of C classes on H hidden columns (its weight gradient inside the kernel up
    to head_max_classes(), as a separate product above)."""
    L = _lib.lib()
    return bool(L.fitgnn_epilogue_bwd_head_supported(int(H), int(C), 1) or L.fitgnn_epilogue_bwd_head_supported(int(H), int(C), 0))


def head_max_classes_wide():
    """Widest head the fused head-backward kernel takes at all (dOut = dy @ Wl formed on the fly; dWl separately)."""
    return int(_lib.lib().fitgnn_head_max_classes_wide())


class SeedBank:
    """Device-resident dropout seeds for steps captured in a hipGraph: kernel arguments are frozen at capture, so the
    kernels read their seed through a pointer (FITGNN_EPI_SEED_DEVICE) and `advance()` -- itself a kernel of the
    captured step -- moves every seed on, giving each replay fresh dropout patterns."""

    GOLD = -7046029254386353131  # 0x9E3779B97F4A7C15 as int64 (addition wraps modulo 2^64)

    def __init__(self, n, device):
        self.seeds = torch.randint(0, 2 ** 62, (n,), dtype=torch.int64).to(device)
        self.cursor = 0

    def advance(self):
        self.seeds.add_(self.GOLD)
        self.cursor = 0

    def take(self):
        s = self.seeds[self.cursor:self.cursor + 1]
        self.cursor += 1
        if self.cursor > self.seeds.numel():
            raise RuntimeError("SeedBank exhausted: size it for every dropout site of the step")
        return s


def next_seed(cfg=DEFAULT):
    """Per-call dropout seed: drawn from torch's generator (so torch.manual_seed controls it), or -- when the config carries
    a SeedBank (steps captured in a hipGraph) -- a one-element device tensor the kernels dereference."""
    if cfg.seed_bank is not None:
        return cfg.seed_bank.take()
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def _seed_arg(seed, epilogue):
    """(integer for the C ABI, epilogue flags): a tensor seed is passed by address with FITGNN_EPI_SEED_DEVICE."""
    if torch.is_tensor(seed):
        return int(seed.data_ptr()), int(epilogue) | _lib.EPI_SEED_DEVICE
    return int(seed) & 0xFFFFFFFFFFFFFFFF, int(epilogue)
