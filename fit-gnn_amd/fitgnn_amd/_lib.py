"""ctypes binding of libfitgnn_hip.so (the C ABI declared in include/fitgnn_hip.h).

The product has NO CPU fallback: if the shared library is missing or a call fails, an exception
is raised.  Build it with `python __graft_entry__.py` or `make -C fit-gnn_amd/csrc`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libfitgnn_hip.so")

c_i32, c_i64, c_u32, c_u64, c_f32, c_size = (ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint64,
                                              ctypes.c_float, ctypes.c_size_t)
ptr = ctypes.c_void_p

EPI_BIAS, EPI_ELU, EPI_DROPOUT, EPI_SEED_DEVICE = 1, 2, 4, 8
SPMM_GATHER = 0x100
MAX_K = 16

# name -> (restype, argtypes); mirrors include/fitgnn_hip.h one to one
SIGNATURES = {
    "fitgnn_abi_version": (ctypes.c_int, []),
    "fitgnn_error_string": (ctypes.c_char_p, [ctypes.c_int]),
    "fitgnn_stream_copy_f32": (ctypes.c_int, [ptr, ptr, c_i64, ptr]),
    "fitgnn_spmm_default_window_rows": (ctypes.c_int, []),
    "fitgnn_spmm_max_window_rows": (ctypes.c_int, [c_i32]),
    "fitgnn_gcn_norm_csr_f32": (ctypes.c_int, [ptr, ptr, ptr, ptr, ptr, c_i32, ptr]),
    "fitgnn_spmm_csr_f32": (ctypes.c_int, [ptr, ptr, ptr, ptr, c_i64, ptr, c_i64, c_i32, c_i32, ptr, c_i32, ptr, ptr,
                                           ptr, c_i32, c_i32, ptr, c_u32, c_f32, c_u64, ptr, ptr]),
    "fitgnn_spmm_csr_blocks_f32": (ctypes.c_int, [ptr, ptr, ptr, ptr, c_i64, ptr, c_i64, c_i32, c_i32, ptr, c_i32, ptr,
                                                  ptr, ptr, c_i32, ptr, c_u32, c_f32, c_u64, ptr, ptr]),
    "fitgnn_spmm_csr_dz_f32": (ctypes.c_int, [ptr, ptr, ptr, ptr, c_i64, ptr, c_i64, c_i32, c_i32, ptr, c_i32, ptr, ptr,
                                              ptr, c_i32, c_i32, ptr, c_u32, c_f32, c_u64, ptr, ptr, ptr]),
    "fitgnn_spmm_csr_blocks_dz_f32": (ctypes.c_int, [ptr, ptr, ptr, ptr, c_i64, ptr, c_i64, c_i32, c_i32, ptr, c_i32, ptr,
                                                     ptr, ptr, c_i32, ptr, c_u32, c_f32, c_u64, ptr, ptr, ptr]),
    "fitgnn_lanczos_parts": (c_i32, [c_i32]),
    "fitgnn_lanczos_spmv_f64": (ctypes.c_int, [ptr, ptr, ptr, ptr, ptr, c_i32, ctypes.c_double, ctypes.c_double, ptr]),
    "fitgnn_lanczos_project_f64": (ctypes.c_int, [ptr, c_i64, c_i32, ptr, c_i32, ptr, ptr, ptr]),
    "fitgnn_lanczos_reduce_f64": (ctypes.c_int, [ptr, c_i32, c_i32, ptr, ptr]),
    "fitgnn_lanczos_finish_f64": (ctypes.c_int, [ptr, c_i64, c_i32, ptr, c_i32, ptr, ptr, ptr, ptr, c_i32, ptr]),
    "fitgnn_lanczos_rotate_f64": (ctypes.c_int, [ptr, c_i64, c_i32, ptr, c_i32, ptr, c_i64, c_i32, ptr]),
    "fitgnn_spmm_csr_stream_f32": (ctypes.c_int, [ptr, ptr, ptr, c_i64, ptr, c_i64, ptr, c_i64, c_i32, c_i32, ptr, c_i32, ptr, c_i32, ptr, ptr,
                                                  ptr, c_u32, c_f32, c_u64, ptr, ptr]),
    "fitgnn_spmm_csr_stream_dz_f32": (ctypes.c_int, [ptr, ptr, ptr, c_i64, ptr, c_i64, ptr, c_i64, c_i32, c_i32, ptr, c_i32, ptr, c_i32, ptr, ptr,
                                                     ptr, c_u32, c_f32, c_u64, ptr, ptr, ptr]),
    "fitgnn_spmm_rows_compact_parts": (c_i32, [c_i32]),
    "fitgnn_spmm_rows_compact_f32": (ctypes.c_int, [ptr, ptr, ptr, c_i64, ptr, c_i64, c_i32, ptr, c_i64, c_i32, c_i32, ptr]),
    "fitgnn_spmm_rows_compact_dz_f32": (ctypes.c_int, [ptr, ptr, ptr, c_i64, ptr, c_i64, c_i32, ptr, c_i64, c_i32, c_i32, ptr, c_u32, c_f32,
                                                       c_u64, ptr, ptr, ptr]),
    "fitgnn_two_hop_rows_f32": (ctypes.c_int, [ptr, ptr, ptr, ptr, c_i64, c_i32, ptr, c_i32, ptr, c_i32, c_u32, c_f32, c_u64, ptr, ptr, c_i64, ptr]),
    "fitgnn_spmm_two_hop_blocks_f32": (ctypes.c_int, [ptr, ptr, ptr, ptr, c_i64, ptr, c_i64, c_i32, c_i32, ptr, c_i32, ptr, ptr, ptr, ptr, ptr, c_i64,
                                                      c_i32, ptr, ptr, c_u32, c_f32, c_u64, ptr, ptr, ptr]),
    "fitgnn_segment_sum_f32": (ctypes.c_int, [ptr, ptr, c_i32, ptr, c_i64, c_i32, ptr, c_i64, ptr]),
    "fitgnn_segment_max_f32": (ctypes.c_int, [ptr, ptr, c_i32, ptr, c_i64, c_i32, ptr, ptr, ptr]),
    "fitgnn_segment_max_bwd_f32": (ctypes.c_int, [ptr, ptr, c_i32, c_i32, ptr, c_i64, ptr]),
    "fitgnn_segment_expand_f32": (ctypes.c_int, [ptr, ptr, ptr, c_i64, c_i32, ptr, ptr]),
    "fitgnn_make_tiles_host": (ctypes.c_int, [ptr, c_i64, c_i32, ptr, c_i64, ptr]),
    "fitgnn_split_blocks_host": (ctypes.c_int, [ptr, c_i64, ptr, c_i32, c_i64, c_i32, ptr, c_i64, ptr, ptr, ptr, ptr, c_i64, ptr]),
    "fitgnn_plan_tiles_host": (ctypes.c_int, [ptr, ptr, c_i32, c_i32, ptr, c_i32, c_i32, c_i32, ptr, ptr, ptr, ptr, ptr]),
    "fitgnn_epilogue_bwd_workspace_bytes": (c_size, [c_i32, c_i32]),
    "fitgnn_epilogue_bwd_f32": (ctypes.c_int, [ptr, ptr, ptr, c_i32, c_i32, c_u32, c_f32, c_u64, ptr, ptr, ptr, c_size, ptr]),
    "fitgnn_head_max_classes": (ctypes.c_int, []),
    "fitgnn_head_max_classes_wide": (ctypes.c_int, []),
    "fitgnn_epilogue_bwd_head_supported": (ctypes.c_int, [c_i32, c_i32, c_i32]),
    "fitgnn_epilogue_bwd_head_workspace_bytes": (c_size, [c_i32, c_i32, c_i32]),
    "fitgnn_epilogue_bwd_head_f32": (ctypes.c_int, [ptr, ptr, c_i32, ptr, ptr, c_i32, c_i32, c_u32, c_f32, c_u64, ptr, ptr, ptr,
                                                   ptr, c_size, ptr]),
    "fitgnn_epilogue_bwd_head_rows_f32": (ctypes.c_int, [ptr, ptr, c_i32, ptr, ptr, c_i32, c_i32, ptr, c_i32, c_u32, c_f32, c_u64, ptr,
                                                        ptr, ptr, ptr, c_size, ptr]),
    "fitgnn_epilogue_bwd_rows_f32": (ctypes.c_int, [ptr, ptr, ptr, c_i32, c_i32, ptr, c_i32, c_u32, c_f32, c_u64, ptr, ptr, ptr, c_size, ptr]),
    "fitgnn_spmm_epilogue_bwd_supported": (ctypes.c_int, [c_i32, c_i32, c_i32]),
    "fitgnn_spmm_epilogue_bwd_workspace_bytes": (c_size, [c_i32, c_i32, c_i32]),
    "fitgnn_spmm_epilogue_bwd_f32": (ctypes.c_int, [ptr, ptr, ptr, ptr, c_i32, c_i32, ptr, ptr, ptr, c_i32, ptr, ptr, c_i32, c_i32,
                                                   c_u32, c_f32, c_u64, ptr, ptr, ptr, ptr, c_size, ptr]),
    "fitgnn_gat_scores_f32": (ctypes.c_int, [ptr, c_i64, c_i32, c_i32, ptr, ptr, ptr, ptr, ptr]),
    "fitgnn_gat_edge_softmax_f32": (ctypes.c_int, [ptr, ptr, ptr, ptr, c_f32, c_i32, ptr, ptr]),
    "fitgnn_sddmm_csr_f32": (ctypes.c_int, [ptr, ptr, ptr, c_i64, ptr, c_i64, c_i32, c_i32, ptr, ptr]),
    "fitgnn_sddmm_csr_rows_f32": (ctypes.c_int, [ptr, ptr, ptr, c_i64, ptr, c_i64, ptr, c_i32, c_i32, ptr, ptr]),
    "fitgnn_gat_softmax_bwd_rows_f32": (ctypes.c_int, [ptr, ptr, ptr, ptr, ptr, ptr, c_f32, ptr, c_i32, ptr, ptr, ptr]),
    "fitgnn_gat_softmax_bwd_f32": (ctypes.c_int, [ptr, ptr, ptr, ptr, ptr, ptr, c_f32, c_i32, ptr, ptr, ptr]),
    "fitgnn_l1_loss_f32": (ctypes.c_int, [ptr, ptr, c_i32, c_f32, ptr, ptr, ptr]),
    "fitgnn_softmax_nll_workspace_bytes": (c_size, [c_i32]),
    "fitgnn_softmax_nll_f32": (ctypes.c_int, [ptr, c_i64, c_i32, c_i32, ptr, ptr, c_i32, c_f32, ptr, ptr, ptr, c_size, ptr]),
    "fitgnn_adam_step_f32": (ctypes.c_int, [ptr, ptr, ptr, ptr, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, ptr, ptr]),
    "fitgnn_sum_leading_f32": (ctypes.c_int, [ptr, c_i32, c_i64, ptr, ptr]),
    "fitgnn_gemm_atb_workspace_bytes": (c_size, [c_i64, c_i32, c_i32]),
    "fitgnn_gemm_atb_f32": (ctypes.c_int, [ptr, c_i64, ptr, c_i64, c_i64, c_i32, c_i32, ptr, ptr, ptr]),
    "fitgnn_gemm_nt_f32": (ctypes.c_int, [ptr, c_i64, ptr, c_i64, c_i64, c_i32, c_i32, ptr, c_i64, ptr]),
    "fitgnn_gemm_exact_workspace_bytes": (c_size, [c_i64, c_i32, c_i64, c_i32, c_i32]),
    "fitgnn_gemm_exact_f32": (ctypes.c_int, [ptr, c_i64, c_i32, ptr, c_i64, c_i32, c_i64, c_i32, c_i64, ptr, c_i64, ptr, ptr]),
    "fitgnn_gemm_nt_presplit_bytes": (c_size, [c_i32, c_i32]),
    "fitgnn_gemm_nt_presplit_f32": (ctypes.c_int, [ptr, c_i64, c_i64, c_i32, c_i32, c_i32, ptr, ptr]),
    "fitgnn_gemm_nt_pre_f32": (ctypes.c_int, [ptr, c_i64, ptr, c_i64, c_i32, c_i32, ptr, c_i64, ptr]),
    "fitgnn_gemm_nt_epilogue_bwd_workspace_bytes": (c_size, [c_i64, c_i32]),
    "fitgnn_gemm_nt_epilogue_bwd_f32": (ctypes.c_int, [ptr, c_i64, ptr, c_i64, c_i64, c_i32, c_i32, ptr, ptr, ctypes.c_uint32, c_f32,
                                                       ctypes.c_uint64, ptr, ptr, ptr, c_size, ptr]),
    "fitgnn_colsum_partials_f32": (ctypes.c_int, [ptr, c_i32, c_i32, ptr, ptr]),
    "fitgnn_head_rows_lds_bytes": (c_size, [c_i32, c_i32]),
    "fitgnn_head_rows_f32": (ctypes.c_int, [ptr, c_i64, ptr, c_i32, ptr, ptr, c_i32, c_i32, ptr, c_i64, c_i32, ptr]),
    "fitgnn_epilogue_fwd_rows_f32": (ctypes.c_int, [ptr, c_i64, ptr, c_i32, c_i32, ptr, c_u32, c_f32, c_u64, ptr, ptr]),
    "fitgnn_dense_narrow_k_lds_bytes": (c_size, [c_i32, c_i32]),
    "fitgnn_dense_narrow_k_f32": (ctypes.c_int, [ptr, c_i64, ptr, c_i64, c_i32, c_i32, c_i32, ptr, c_u32, c_f32, c_u64, ptr, ptr, c_i64, ptr]),
    "fitgnn_narrow_atb_workspace_bytes": (c_size, [c_i32, c_i32, c_i32]),
    "fitgnn_narrow_atb_f32": (ctypes.c_int, [ptr, c_i64, ptr, c_u32, c_f32, c_u64, ptr, ptr, c_i64, c_i32, c_i32, c_i32, ptr, ptr, ptr, c_size,
                                             ptr]),
    "fitgnn_adam_step_acc_f32": (ctypes.c_int, [ptr, ptr, ptr, ptr, ptr, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, ptr, ptr, c_i32, c_u64, ptr]),
    "fitgnn_pool_head_supported": (ctypes.c_int, [c_i32, c_i32]),
    "fitgnn_pool_head_f32": (ctypes.c_int, [ptr, ptr, c_i32, ptr, c_i64, c_i32, ptr, ptr, ptr, c_i32, ptr, ptr, ptr]),
    "fitgnn_pool_head_bwd_f32": (ctypes.c_int, [ptr, ptr, c_i32, ptr, ptr, ptr, c_i64, c_i32, c_i32, ptr, ptr, ptr, ptr]),
    "fitgnn_colsum_narrow_workspace_bytes": (c_size, [c_i32, c_i32]),
    "fitgnn_colsum_narrow_f32": (ctypes.c_int, [ptr, c_i64, c_i32, c_i32, ptr, ptr, c_size, ptr]),
    "fitgnn_spmm_narrow_f32": (ctypes.c_int, [ptr, ptr, ptr, ptr, ptr, c_i32, c_i32, c_f32, ptr, c_f32, ptr, c_f32, ptr]),
    "fitgnn_spmm_narrow_padded_f32": (ctypes.c_int, [ptr, ptr, ptr, ptr, ptr, c_i32, c_i32, c_f32, ptr, c_f32, ptr, c_f32, ptr]),
    "fitgnn_appnp_unit_rows": (ctypes.c_int, [c_i32]),
    "fitgnn_appnp_unit_entries": (ctypes.c_int, []),
    "fitgnn_appnp_block_rows": (ctypes.c_int, []),
    "fitgnn_appnp_block_entries": (ctypes.c_int, []),
    "fitgnn_appnp_blocks_f32": (ctypes.c_int, [ptr, ptr, ptr, ptr, c_i32, c_i32, c_i32, ptr, ptr, ptr, ptr, c_i32, c_i32, c_f32, c_i32, ptr]),
    "fitgnn_appnp_lds_bytes": (ctypes.c_int64, [c_i32, c_i32, c_i32]),
    "fitgnn_appnp_lds_max_bytes": (ctypes.c_int, []),
    "fitgnn_appnp_lds_items_per_thread": (ctypes.c_int, []),
    "fitgnn_appnp_lds_f32": (ctypes.c_int, [ptr, ptr, ptr, ptr, c_i32, c_i32, c_i32, ptr, ptr, c_i32, c_i32, c_f32, c_i32, c_i32, c_i32, ptr]),
    "fitgnn_gather_rows_padded_f32": (ctypes.c_int, [ptr, c_i64, c_i32, ptr, c_i64, ptr, c_i32, ptr]),
    "fitgnn_appnp_units_f32": (ctypes.c_int, [ptr, ptr, ptr, ptr, c_i32, c_i32, c_i32, ptr, ptr, c_i32, c_i32, c_f32, c_i32, ptr]),
    "fitgnn_csr_row_sum_f32": (ctypes.c_int, [ptr, ptr, c_i32, ptr, ptr]),
    "fitgnn_induced_edges_count": (ctypes.c_int, [ptr, ptr, ptr, ptr, ptr, ptr, c_i64, ptr, ptr]),
    "fitgnn_induced_edges_fill": (ctypes.c_int, [ptr, ptr, ptr, ptr, ptr, ptr, ptr, c_i64, ptr, ptr, ptr, ptr]),
    "fitgnn_batch_offsets": (ctypes.c_int, [ptr, ptr, c_i32, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr]),
    "fitgnn_batch_gather": (ctypes.c_int, [c_i32, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, c_i32, ptr, c_i32, c_i32,
                                           c_i32, c_i32, c_i32, c_i32, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, c_i32, ptr, ptr, ptr, ptr, ptr,
                                           c_i32, ptr]),
    "fitgnn_closed_neighbourhoods": (ctypes.c_int, [ptr, ptr, c_i32, ptr, ptr, ptr]),
    "fitgnn_variation_costs_f64": (ctypes.c_int, [ptr, ptr, ptr, ptr, ptr, c_i32, c_i64, ptr, ptr, ptr, c_i32, ptr, ptr]),
    "fitgnn_variation_costs_batch_f64": (ctypes.c_int, [ptr, ptr, ptr, ptr, ptr, c_i32, c_i64, ptr, ptr, ptr, ptr, c_i32, ptr, ptr]),
    "fitgnn_greedy_select_workspace_bytes": (c_size, [c_i32, c_i64]),
    "fitgnn_greedy_select": (ctypes.c_int, [ptr, ptr, ptr, ptr, ptr, c_i32, c_i64, c_i32, ptr, ptr, ptr, c_i64, ptr, ptr,
                                            ptr, ptr, c_size, ptr]),
    "fitgnn_greedy_select_batch_workspace_bytes": (c_size, [c_i32, c_i64, c_i32]),
    "fitgnn_greedy_select_batch": (ctypes.c_int, [ptr, ptr, ptr, ptr, ptr, c_i32, c_i64, c_i32, ptr, ptr, ptr, c_i32, ptr, ptr,
                                                  c_i64, ptr, ptr, ptr, ptr, ptr, ptr, c_size, ptr]),
    "fitgnn_build_assignment_workspace_bytes": (c_size, [c_i32]),
    "fitgnn_build_assignment": (ctypes.c_int, [c_i32, ptr, ptr, ptr, ptr, ptr, ptr, ptr, c_size, ptr]),
    "fitgnn_compose_levels": (ctypes.c_int, [c_i32, ptr, ptr, ptr, ptr, ptr]),
    "fitgnn_lift_adjacency_workspace_bytes": (c_size, [c_i32, c_i64, c_i32]),
    "fitgnn_lift_adjacency": (ctypes.c_int, [c_i32, ptr, ptr, ptr, ptr, ptr, c_i32, ptr, ptr, ptr, ptr, ptr, c_size, ptr]),
    "fitgnn_pool_rows_workspace_bytes": (c_size, [c_i32, c_i32]),
    "fitgnn_pool_rows_f32": (ctypes.c_int, [ptr, ptr, c_i32, c_i32, ptr, c_i64, c_i32, ptr, c_i64, ptr, ptr, c_size, ptr]),
}

_lib = None


class FitgnnError(RuntimeError):
    pass


def lib():
    """The loaded library; raises (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FitgnnError(f"{LIB_PATH} not found: build the HIP extension first "
                              "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        # torch first: its wheel carries its own HIP runtime, and the process must hold ONE -- the library's libamdhip64
        # dependency then resolves to the copy torch has loaded.  Loaded the other way round (library first, e.g. build()
        # followed by smoke() in one process) the two runtimes coexist and this library's sees no device (hipErrorNoDevice).
        import torch  # noqa: F401

        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.fitgnn_abi_version() != 1:
            raise FitgnnError("libfitgnn_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().fitgnn_error_string(int(rc)).decode()
        raise FitgnnError(f"{what}: {msg} (code {rc})")


def dptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise FitgnnError("fitgnn ops run on the MI355X only: got a CPU tensor (no CPU fallback exists)")
