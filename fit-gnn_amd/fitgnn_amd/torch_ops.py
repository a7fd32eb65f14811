"""`torch.ops.fitgnn.*`: the C-ABI launchers as schema-registered PyTorch custom ops (north_star: "exposed
as PyTorch-ROCm custom ops").  Importing this module registers

    fitgnn::spmm_csr(rowptr, col, val, X, tiles, window_rows, bias?, epilogue, p, seed, mask?) -> Y
    fitgnn::spmm_csr_t(... of the TRANSPOSED pattern ...)          the adjoint, used as spmm_csr's backward
    fitgnn::gcn_norm_csr(rowptr, col, w?) -> (val, dinv)
    fitgnn::epilogue_bwd(dOut, out, epilogue, p, seed, mask?) -> (dZ, db)
    fitgnn::pool_rows(assign, cval, n, X) -> Xc
    fitgnn::variation_costs(rowptr, col, w?, dw, A, set_off, set_mem) -> cost
    fitgnn::lift_adjacency(rowptr, col, w, assign, cval, n) -> (rowptr_c, col_c, w_c)
    fitgnn::gemm_nt(a, b) -> a @ b^T          fitgnn::gemm_atb(a, b) -> a^T @ b      (3 x bf16 MFMA kernels, fp32 in/out)
    fitgnn::linear(x, W) -> x @ W^T           differentiable: dX = gemm_nt(dY, W^T), dW = gemm_atb(dY, x)

for the CUDA (HIP) dispatch key only -- there is no CPU kernel, a CPU tensor fails in the dispatcher -- with fake
(meta) kernels for shape inference and an autograd formula for spmm_csr over a pair of forward / transposed
patterns.  The higher-level autograd Functions of fitgnn_amd.ops (fused layers) call the same launchers directly.
"""
import torch

from . import _lib, coarsening, ops

_LIB = torch.library.Library("fitgnn", "DEF")
_LIB.define("spmm_csr(Tensor rowptr, Tensor col, Tensor val, Tensor X, Tensor tiles, int window_rows, Tensor? bias, "
            "int epilogue, float p, int seed, Tensor? mask) -> Tensor")
_LIB.define("spmm_csr_pair(Tensor rowptr, Tensor col, Tensor val, Tensor tiles, Tensor rowptr_t, Tensor col_t, Tensor val_t, "
            "Tensor tiles_t, Tensor X, int window_rows) -> Tensor")
_LIB.define("gcn_norm_csr(Tensor rowptr, Tensor col, Tensor? w) -> (Tensor, Tensor)")
_LIB.define("epilogue_bwd(Tensor dOut, Tensor out, int epilogue, float p, int seed, Tensor? mask) -> (Tensor, Tensor)")
_LIB.define("pool_rows(Tensor assign, Tensor cval, int n, Tensor X) -> Tensor")
_LIB.define("variation_costs(Tensor rowptr, Tensor col, Tensor? w, Tensor dw, Tensor A, Tensor set_off, Tensor set_mem) -> Tensor")
_LIB.define("lift_adjacency(Tensor rowptr, Tensor col, Tensor w, Tensor assign, Tensor cval, int n) -> (Tensor, Tensor, Tensor)")
_LIB.define("gemm_nt(Tensor a, Tensor b) -> Tensor")
_LIB.define("gemm_atb(Tensor a, Tensor b) -> Tensor")
_LIB.define("linear(Tensor x, Tensor W) -> Tensor")


def _spmm_csr(rowptr, col, val, X, tiles, window_rows, bias, epilogue, p, seed, mask):
    return ops.spmm_raw(rowptr, col, val, tiles, X, int(rowptr.numel()) - 1, bias=bias, epilogue=epilogue, p=p, seed=seed,
                        mask=mask, window_rows=window_rows)


def _spmm_csr_pair(rowptr, col, val, tiles, rowptr_t, col_t, val_t, tiles_t, X, window_rows):
    return ops.spmm_raw(rowptr, col, val, tiles, X, int(rowptr.numel()) - 1, window_rows=window_rows)


def _gcn_norm_csr(rowptr, col, w):
    _lib.require_cuda(rowptr, col, w)
    n = int(rowptr.numel()) - 1
    val = torch.empty(col.numel(), dtype=torch.float32, device=col.device)
    dinv = torch.empty(n, dtype=torch.float32, device=col.device)
    _lib.check(_lib.lib().fitgnn_gcn_norm_csr_f32(_lib.dptr(rowptr), _lib.dptr(col), _lib.dptr(w), _lib.dptr(val), _lib.dptr(dinv), n,
                                                  _lib.stream_ptr(col.device)), "fitgnn_gcn_norm_csr_f32")
    return val, dinv


def _epilogue_bwd(dOut, out, epilogue, p, seed, mask):
    return ops.epilogue_bwd_raw(dOut, out, epilogue, p=p, seed=seed, mask=mask, want_db=True)


def _pool_rows(assign, cval, n, X):
    return coarsening.pool_rows(assign, cval, n, X)


def _variation_costs(rowptr, col, w, dw, A, set_off, set_mem):
    _lib.require_cuda(rowptr, col, w, dw, A, set_off, set_mem)
    n_sets = int(set_off.numel()) - 1
    K = int(A.shape[1])
    set_len = (set_off[1:] - set_off[:-1]).contiguous()
    cost = torch.empty(n_sets, dtype=torch.float64, device=A.device)
    _lib.check(_lib.lib().fitgnn_variation_costs_f64(_lib.dptr(rowptr), _lib.dptr(col), _lib.dptr(w), _lib.dptr(dw), _lib.dptr(A), K,
                                                     int(A.stride(0)), _lib.dptr(set_off), _lib.dptr(set_len), _lib.dptr(set_mem), n_sets,
                                                     _lib.dptr(cost), _lib.stream_ptr(A.device)), "fitgnn_variation_costs_f64")
    return cost


def _lift_adjacency(rowptr, col, w, assign, cval, n):
    res = coarsening.LevelResult()
    res.N, res.n, res.assign, res.cval, res.device = int(rowptr.numel()) - 1, int(n), assign, cval, rowptr.device
    res.rowptr, res.col, res.w = rowptr, col, w
    Wc = coarsening.lift_adjacency(res)
    dev = rowptr.device
    return (torch.from_numpy(Wc.indptr.astype("int32")).to(dev), torch.from_numpy(Wc.indices.astype("int32")).to(dev),
            torch.from_numpy(Wc.data).to(dev))


def _gemm_nt(a, b):
    _lib.require_cuda(a, b)
    return ops.gemm_nt(a.contiguous(), b.contiguous())


def _gemm_atb(a, b):
    _lib.require_cuda(a, b)
    return ops.gemm_atb(a.contiguous(), b.contiguous())


for _name, _fn in (("gemm_nt", _gemm_nt), ("gemm_atb", _gemm_atb), ("linear", _gemm_nt), ("spmm_csr", _spmm_csr), ("spmm_csr_pair", _spmm_csr_pair), ("gcn_norm_csr", _gcn_norm_csr),
                   ("epilogue_bwd", _epilogue_bwd), ("pool_rows", _pool_rows), ("variation_costs", _variation_costs),
                   ("lift_adjacency", _lift_adjacency)):
    _LIB.impl(_name, _fn, "CUDA")


# ---- fake kernels: shapes / dtypes without touching data (torch.compile tracing, meta tensors) ----
@torch.library.register_fake("fitgnn::spmm_csr")
def _(rowptr, col, val, X, tiles, window_rows, bias, epilogue, p, seed, mask):
    return X.new_empty((rowptr.shape[0] - 1, X.shape[1]))


@torch.library.register_fake("fitgnn::spmm_csr_pair")
def _(rowptr, col, val, tiles, rowptr_t, col_t, val_t, tiles_t, X, window_rows):
    return X.new_empty((rowptr.shape[0] - 1, X.shape[1]))


@torch.library.register_fake("fitgnn::gcn_norm_csr")
def _(rowptr, col, w):
    return col.new_empty(col.shape, dtype=torch.float32), col.new_empty((rowptr.shape[0] - 1,), dtype=torch.float32)


@torch.library.register_fake("fitgnn::epilogue_bwd")
def _(dOut, out, epilogue, p, seed, mask):
    return torch.empty_like(dOut), dOut.new_empty((dOut.shape[1],))


@torch.library.register_fake("fitgnn::pool_rows")
def _(assign, cval, n, X):
    return X.new_empty((n, X.shape[1]), dtype=torch.float32)


@torch.library.register_fake("fitgnn::variation_costs")
def _(rowptr, col, w, dw, A, set_off, set_mem):
    return A.new_empty((set_off.shape[0] - 1,), dtype=torch.float64)


@torch.library.register_fake("fitgnn::gemm_nt")
def _(a, b):
    return a.new_empty((a.shape[0], b.shape[0]))


@torch.library.register_fake("fitgnn::gemm_atb")
def _(a, b):
    return a.new_empty((a.shape[1], b.shape[1]))


@torch.library.register_fake("fitgnn::linear")
def _(x, W):
    return x.new_empty((x.shape[0], W.shape[0]))


def _linear_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _linear_backward(ctx, dY):
    x, W = ctx.saved_tensors
    dY = dY.contiguous()
    dX = torch.ops.fitgnn.gemm_nt(dY, W.t().contiguous()) if ctx.needs_input_grad[0] else None
    dW = torch.ops.fitgnn.gemm_atb(dY, x) if ctx.needs_input_grad[1] else None
    return dX, dW


torch.library.register_autograd("fitgnn::linear", _linear_backward, setup_context=_linear_setup)


# ---- autograd: d/dX of Y = A X is A^T dY -- the same kernel on the transposed pattern ----
def _pair_setup(ctx, inputs, output):
    rowptr, col, val, tiles, rowptr_t, col_t, val_t, tiles_t, X, window_rows = inputs
    ctx.save_for_backward(rowptr, col, val, tiles, rowptr_t, col_t, val_t, tiles_t)
    ctx.window_rows = window_rows


def _pair_backward(ctx, dY):
    rowptr, col, val, tiles, rowptr_t, col_t, val_t, tiles_t = ctx.saved_tensors
    dX = torch.ops.fitgnn.spmm_csr_pair(rowptr_t, col_t, val_t, tiles_t, rowptr, col, val, tiles, dY.contiguous(), ctx.window_rows)
    return None, None, None, None, None, None, None, None, dX, None


torch.library.register_autograd("fitgnn::spmm_csr_pair", _pair_backward, setup_context=_pair_setup)


def spmm(g, X):
    """Differentiable Y = A_hat X through torch.ops.fitgnn.spmm_csr_pair for a fitgnn_amd.csr.CSRGraph."""
    f, t = g.f, g.t
    return torch.ops.fitgnn.spmm_csr_pair(f.rowptr, f.col, f.val, f.tiles, t.rowptr, t.col, t.val, t.tiles, X, g.window_rows)
