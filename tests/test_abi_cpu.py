"""CPU tier: the C-ABI library loads and exports every symbol include/fitgnn_hip.h declares; host-only
size queries work; product code refuses CPU tensors (no fallback)."""
import os
import re

import pytest
import torch

from fitgnn_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "fitgnn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fitgnn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/fitgnn_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes signature table out of sync with the header"
    assert L.fitgnn_abi_version() == 1


def test_error_strings_and_size_queries():
    L = _lib.lib()
    assert b"workspace" in L.fitgnn_error_string(-2)
    assert b"bad argument" in L.fitgnn_error_string(-1)
    assert 8 <= L.fitgnn_spmm_default_window_rows() <= L.fitgnn_spmm_max_window_rows(512)
    assert L.fitgnn_epilogue_bwd_workspace_bytes(130, 512) == 33 * 512 * 4  # 4-row chunks, 33 of them
    assert L.fitgnn_epilogue_bwd_workspace_bytes(100000, 512) <= 1024 * 512 * 4  # at most 1024 row chunks
    assert L.fitgnn_greedy_select_workspace_bytes(1000, 6000) > 1000 * 4
    assert L.fitgnn_lift_adjacency_workspace_bytes(10, 50, 5) > 0
    assert L.fitgnn_pool_rows_workspace_bytes(100, 50) > 0
    assert L.fitgnn_build_assignment_workspace_bytes(100) >= 2 * 101 * 4
    # split-K weight-gradient GEMM: a multiple of 8 row chunks of one [M x N] partial each, ~256 workgroups
    assert L.fitgnn_gemm_atb_workspace_bytes(90549, 512, 512) == 64 * 512 * 512 * 4
    assert L.fitgnn_gemm_atb_workspace_bytes(33, 8, 4) == 8 * 8 * 4 * 4
    assert L.fitgnn_gemm_atb_workspace_bytes(0, 8, 4) == 0


def test_bad_arguments_are_rejected_without_touching_the_gpu():
    L = _lib.lib()
    assert L.fitgnn_spmm_csr_f32(None, None, None, None, 4, None, 4, -1, 4, None, 0, None, None, None, -1, 0, None, 0, 0.0, 0, None, None) == -1
    assert L.fitgnn_spmm_csr_f32(None, None, None, None, 4, None, 4, 0, 4, None, 0, None, None, None, -1, 0, None, 0, 0.0, 0, None, None) == 0
    # the backward-epilogue flag belongs to the *_dz_* entry points only
    assert L.fitgnn_spmm_csr_f32(None, None, None, None, 4, None, 4, 8, 4, None, 1, None, None, None, -1, 0, None, 0x10, 0.0, 0, None, None) == -1
    # the loss-row entry points: argument errors before any GPU work
    assert L.fitgnn_head_rows_f32(None, 512, None, -1, None, None, 3, 512, None, 3, 0, None) == -1       # negative row count
    assert L.fitgnn_head_rows_f32(None, 510, None, 4, None, None, 3, 510, None, 3, 0, None) == -1        # H % 4 != 0
    assert L.fitgnn_head_rows_f32(None, 512, None, 0, None, None, 3, 512, None, 3, 0, None) == 0         # nothing to do
    assert L.fitgnn_head_rows_lds_bytes(512, 47) == (47 * 516 + 4 * 4 * 512) * 4
    assert L.fitgnn_epilogue_fwd_rows_f32(None, 512, None, 5, 510, None, 0, 0.0, 0, None, None) == -1   # H % 4 != 0
    assert L.fitgnn_epilogue_fwd_rows_f32(None, 512, None, 0, 512, None, 0, 0.0, 0, None, None) == 0
    assert L.fitgnn_epilogue_bwd_head_rows_f32(None, None, 3, None, None, 4, 0, None, 512, 0, 0.0, 0, None, None, None, None, 0, None) == -1
    assert L.fitgnn_spmm_csr_dz_f32(None, None, None, None, 4, None, 4, -1, 4, None, 0, None, None, None, -1, 0, None, 0, 0.0, 0, None, None, None) == -1
    assert L.fitgnn_spmm_csr_blocks_dz_f32(None, None, None, None, 4, None, 4, 8, 4, None, 1, None, None, None, -1, None, 0, 0.0, 0, None, None, None) == -1
    assert L.fitgnn_variation_costs_f64(None, None, None, None, None, 17, 17, None, None, None, 1, None, None) == -1
    assert L.fitgnn_variation_costs_f64(None, None, None, None, None, 10, 10, None, None, None, 0, None, None) == 0
    assert L.fitgnn_pool_rows_f32(None, None, 5, 0, None, 4, 4, None, 4, None, None, 0, None) == 0
    assert L.fitgnn_gemm_atb_f32(None, 8, None, 8, 100, 8, 8, None, None, None) == -1
    assert L.fitgnn_gemm_atb_f32(None, 6, None, 8, 100, 6, 8, None, None, None) == -1  # M % 4 != 0
    assert L.fitgnn_gemm_nt_f32(None, 48, None, 48, 100, 8, 48, None, 8, None) == -1   # K % 32 != 0
    assert L.fitgnn_gemm_nt_f32(None, 64, None, 64, 0, 8, 64, None, 8, None) == 0       # no rows: nothing to do


def test_no_cpu_fallback():
    from fitgnn_amd import nn as fnn

    conv = fnn.GCNConv(4, 8)
    x = torch.randn(5, 4)
    ei = torch.tensor([[0, 1, 2, 3], [1, 0, 3, 2]])
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            conv(x, ei)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "fit-gnn_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(d, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), f"{f} mentions the oracle"


def test_fused_gemm_epilogue_entry_points():
    L = _lib.lib()
    assert L.fitgnn_gemm_nt_epilogue_bwd_workspace_bytes(90549, 512) == 354 * 512 * 4   # one partial row per 256-row tile
    assert L.fitgnn_gemm_nt_epilogue_bwd_workspace_bytes(0, 512) == 0
    # too small a workspace and N % 4 != 0 are refused before any launch
    assert L.fitgnn_gemm_nt_epilogue_bwd_f32(None, 64, None, 64, 100, 8, 64, None, None, 0, 0.0, 0, None, None, None, 0, None) == -1
    assert L.fitgnn_colsum_partials_f32(None, 1, 8, None, None) == -1
    assert L.fitgnn_head_max_classes() == 16 and L.fitgnn_head_max_classes_wide() == 48
    S = L.fitgnn_epilogue_bwd_head_supported
    assert S(512, 16, 1) and S(512, 47, 0) and not S(512, 47, 1) and not S(512, 49, 0)
    assert S(16, 4, 1) and not S(16, 7, 1)        # hidden 16: 4 lanes own columns -> at most 4 classes (Cora has 7)
    assert S(320, 16, 1) and not S(320, 17, 0)    # last slab 64 columns = 16 lanes
    assert S(33, 7, 1) and not S(3, 4, 1)
    assert L.fitgnn_gemm_nt_presplit_bytes(512, 512) == 2 * 16 * 32768   # 2 column tiles x 16 stages x 32-KB LDS image
    assert L.fitgnn_gemm_nt_presplit_bytes(500, 64) == 2 * 2 * 32768
    assert L.fitgnn_gemm_nt_presplit_bytes(512, 48) == 0                  # K % 32 != 0: not supported
    assert L.fitgnn_gemm_nt_presplit_f32(None, 1, 1, 8, 64, 64, None, None) == -1
    assert L.fitgnn_gemm_nt_presplit_f32(None, 1, 1, 8, 64, 65, None, None) == -1   # K_valid > K


def test_round3_entry_points_check_their_arguments_without_a_gpu():
    """fitgnn_gemm_exact_f32, the row- / segment-streaming SpMMs and the Lanczos step: argument errors are reported before any GPU
    work, size queries are host-only."""
    L = _lib.lib()
    # exact fp32 GEMM: k split only where it shortens the launch, in multiples of 8 chunks
    # 1 290 tiles of 256 x 256 = 5 full rounds of 256 workgroups + 10 tiles: no k split of the grid, but the last round's rows
    # (165 000 - 5 * 128 * 256 = 1 160) are a launch of their own, split eight ways over k (gemm_f32.hip: make_plan)
    assert L.fitgnn_gemm_exact_workspace_bytes(165000, 512, 512, 0, 0) == 8 * (165000 - 163840) * 512 * 4
    assert L.fitgnn_gemm_exact_workspace_bytes(5 * 128 * 256, 512, 512, 0, 0) == 0                 # whole rounds only: one launch
    assert L.fitgnn_gemm_exact_workspace_bytes(165000, 512, 512, 1, 1) != 8 * 1160 * 512 * 4       # the tail launch needs a k-minor a
    wb = L.fitgnn_gemm_exact_workspace_bytes(512, 512, 165000, 1, 1)                               # a few tiles of a long k: split,
    assert wb >= 16 * 512 * 512 * 4 and wb % (8 * 512 * 512 * 4) == 0                              # in multiples of 8 chunks
    wb = L.fitgnn_gemm_exact_workspace_bytes(34493, 512, 8448, 0, 0)                               # 270 tiles of a long k: split
    assert wb > 0 and (wb // (34493 * 512 * 4)) % 8 == 0
    assert L.fitgnn_gemm_exact_workspace_bytes(0, 8, 8, 0, 0) == 0
    assert L.fitgnn_gemm_exact_f32(None, 8, 0, None, 8, 0, 8, 8, 8, None, 8, None, None) == -1      # NULL operands
    assert L.fitgnn_gemm_exact_f32(None, 8, 1, None, 8, 0, 8, 8, 8, None, 8, None, None) == -1      # (k-major, k-minor): not a form
    assert L.fitgnn_gemm_exact_f32(None, 8, 0, None, 8, 0, 0, 8, 8, None, 8, None, None) == -1      # I = 0
    # compact-operand row streaming
    assert L.fitgnn_spmm_rows_compact_parts(1) == 1 and L.fitgnn_spmm_rows_compact_parts(64 * 8192 + 1) <= 8192
    assert L.fitgnn_spmm_rows_compact_dz_f32(None, None, None, 0, None, 512, 0, None, 512, 4, 512, None, 0, 0.0, 0, None, None, None) == -1
    # the two-hop backward: nothing to do / null arrays / a bias flag among the forward's epilogue flags
    assert L.fitgnn_two_hop_rows_f32(None, None, None, None, 512, 0, None, 0, None, 512, 0, 0.0, 0, None, None, 512, None) == 0
    assert L.fitgnn_two_hop_rows_f32(None, None, None, None, 512, 0, None, 3, None, 512, 0, 0.0, 0, None, None, 512, None) == -1
    z = [None] * 3
    assert L.fitgnn_spmm_two_hop_blocks_f32(*z, None, 512, None, 512, 0, 512, None, 0, None, None, None, None, None, 512, 0, None, None, 0, 0.0, 0,
                                            None, None, None) == 0
    assert L.fitgnn_spmm_two_hop_blocks_f32(*z, None, 512, None, 512, 9, 512, None, 2, None, None, None, None, None, 512, 0, None, None, 0, 0.0, 0,
                                            None, None, None) == -1
    assert L.fitgnn_spmm_rows_compact_f32(None, None, None, 0, None, 512, -1, None, 512, 4, 512, None) == -1   # zero_from < 0
    # segment streaming: xrow and xcol come together
    assert L.fitgnn_spmm_csr_stream_f32(None, None, None, 0, None, 512, None, 512, 0, 512, None, 0, None, 0, None, None, None, 0, 0.0, 0,
                                        None, None) == 0                                            # nothing to do
    assert L.fitgnn_spmm_csr_stream_f32(None, None, None, 5, None, 512, None, 512, 4, 512, None, 1, None, 1, None, None, None, 0, 0.0, 0,
                                        None, None) == -1                                           # NULL arrays
    assert L.fitgnn_spmm_csr_stream_dz_f32(None, None, None, 5, None, 512, None, 512, 4, 510, None, 1, None, 1, None, None, None, 0, 0.0, 0,
                                           None, None, None) == -1
    # Lanczos step
    assert L.fitgnn_lanczos_parts(0) == 0 and L.fitgnn_lanczos_parts(165000) == (165000 + 511) // 512
    assert L.fitgnn_lanczos_spmv_f64(None, None, None, None, None, 0, -1.0, 2.0, None) == 0
    assert L.fitgnn_lanczos_spmv_f64(None, None, None, None, None, 5, -1.0, 2.0, None) == -1
    assert L.fitgnn_lanczos_project_f64(None, 10, 200, None, 10, None, None, None) == -1            # more than 128 basis vectors
    assert L.fitgnn_lanczos_reduce_f64(None, 4, 200, None, None) == -1
    assert L.fitgnn_lanczos_rotate_f64(None, 10, 60, None, 17, None, 10, 10, None) == -1            # more than 16 rotated columns
    assert L.fitgnn_lanczos_finish_f64(None, 4, 0, None, 10, None, None, None, None, 1, None) == -1  # ldv < n


def test_round4_graph_step_entry_points_check_their_arguments_without_a_gpu():
    """The narrow first layer, the pool + head pair, the accumulating Adam and the device batch assembly: shape support queries are
    host-only, argument errors are reported before any launch."""
    L = _lib.lib()
    lds = L.fitgnn_dense_narrow_k_lds_bytes
    assert lds(11, 512) == (11 * 512 + 16 * 11) * 4 and lds(32, 128) > 0
    assert lds(33, 512) == 0 and lds(0, 512) == 0 and lds(11, 510) == 0 and lds(32, 1024) == 0   # K > 32, H % 4, W^T beyond 64 KiB
    assert L.fitgnn_dense_narrow_k_f32(None, 11, None, 11, 5, 11, 512, None, 0, 0.0, 0, None, None, 512, None) == -1       # NULL operands
    assert L.fitgnn_dense_narrow_k_f32(None, 11, None, 11, 0, 11, 512, None, 0, 0.0, 0, None, None, 512, None) == 0        # nothing to do
    assert L.fitgnn_dense_narrow_k_f32(None, 11, None, 11, 5, 40, 512, None, 0, 0.0, 0, None, None, 512, None) == -1       # K not supported
    wb = L.fitgnn_narrow_atb_workspace_bytes
    assert wb(4861, 11, 512) == ((4861 + 15) // 16) * (512 * 11 + 512) * 4
    assert wb(100, 11, 96) == 0 and wb(100, 33, 512) == 0 and wb(0, 11, 512) == 0                 # H / 4 must divide 256; K <= 32
    assert L.fitgnn_narrow_atb_f32(None, 512, None, 0, 0.0, 0, None, None, 11, 100, 11, 512, None, None, None, 0, None) == -1
    S = L.fitgnn_pool_head_supported
    assert S(512, 1) and S(64, 8) and S(1024, 3) and not S(96, 1) and not S(512, 9) and not S(510, 1) and not S(2048, 1)
    assert L.fitgnn_pool_head_f32(None, None, 0, None, 512, 512, None, None, None, 1, None, None, None) == 0
    assert L.fitgnn_pool_head_f32(None, None, 4, None, 512, 512, None, None, None, 1, None, None, None) == -1
    assert L.fitgnn_pool_head_bwd_f32(None, None, 1, None, None, None, 10, 4, 512, None, None, None, None) == -1
    assert L.fitgnn_adam_step_acc_f32(None, None, None, None, None, 6, 0.01, 0.9, 0.999, 1e-8, 0.0, None, None, 0, 0, None) == -1   # n % 4
    assert L.fitgnn_adam_step_acc_f32(None, None, None, None, None, 0, 0.01, 0.9, 0.999, 1e-8, 0.0, None, None, 0, 0, None) == 0
    assert L.fitgnn_adam_step_acc_f32(None, None, None, None, None, 8, 0.01, 0.9, 0.999, 1e-8, 0.0, None, None, 0, 0, None) == -1   # NULL buffers
    assert L.fitgnn_batch_offsets(None, None, 2000, None, None, None, None, None, None, None, None, None) == -1                    # B > 1024
    assert L.fitgnn_batch_offsets(None, None, 128, None, None, None, None, None, None, None, None, None) == -1
    assert L.fitgnn_appnp_unit_rows(12) == 64 and L.fitgnn_appnp_unit_rows(1) == 768 and L.fitgnn_appnp_unit_rows(17) == 0
    assert L.fitgnn_appnp_unit_entries() == 2048
    assert L.fitgnn_appnp_units_f32(None, None, None, None, 0, 0, 0, None, None, 12, 10, 0.1, 0, None) == 0           # nothing to do
    assert L.fitgnn_appnp_units_f32(None, None, None, None, 5, 64, 100, None, None, 12, 10, 0.1, 0, None) == -1       # NULL arrays
    assert L.fitgnn_appnp_units_f32(None, None, None, None, 5, 65, 100, None, None, 12, 10, 0.1, 0, None) == -1       # a unit beyond the capacity
    assert L.fitgnn_gather_rows_padded_f32(None, 47, 47, None, 0, None, 12, None) == 0            # nothing to do
    assert L.fitgnn_gather_rows_padded_f32(None, 47, 47, None, 10, None, 12, None) == -1          # NULL arrays
    assert L.fitgnn_gather_rows_padded_f32(None, 47, 47, None, 10, None, 11, None) == -1          # 4 h4 < H
    assert L.fitgnn_appnp_lds_items_per_thread() == 4 and L.fitgnn_appnp_lds_max_bytes() == 160 * 1024
    assert 0 < L.fitgnn_appnp_lds_bytes(64, 300, 4) < 16 * 1024 and L.fitgnn_appnp_lds_bytes(2000, 7000, 2) < 160 * 1024 < L.fitgnn_appnp_lds_bytes(2000, 7000, 4)
    assert L.fitgnn_appnp_lds_bytes(64, 300, 3) == -1 and L.fitgnn_appnp_lds_bytes(70000, 300, 1) == -1
    assert L.fitgnn_appnp_lds_f32(None, None, None, None, 0, 0, 0, None, None, 12, 10, 0.1, 0, 64, 4, None) == 0            # nothing to do
    assert L.fitgnn_appnp_lds_f32(None, None, None, None, 5, 64, 100, None, None, 12, 10, 0.1, 0, 64, 4, None) == -1        # NULL arrays
    assert L.fitgnn_appnp_lds_f32(None, None, None, None, 5, 65, 100, None, None, 12, 10, 0.1, 0, 64, 4, None) == -1        # 65 x 4 items > 4 x 64
    assert L.fitgnn_appnp_lds_f32(None, None, None, None, 5, 4096, 16000, None, None, 12, 10, 0.1, 0, 1024, 1, None) == -1  # beyond LDS
    assert L.fitgnn_appnp_block_rows() == 4096 and L.fitgnn_appnp_block_entries() == 16384
    assert L.fitgnn_appnp_blocks_f32(None, None, None, None, 0, 0, 0, None, None, None, None, 12, 10, 0.1, 0, None) == 0   # nothing to do
    assert L.fitgnn_appnp_blocks_f32(None, None, None, None, 5, 64, 100, None, None, None, None, 12, 10, 0.1, 0, None) == -1   # NULL arrays
    assert L.fitgnn_appnp_blocks_f32(None, None, None, None, 5, 4097, 100, None, None, None, None, 12, 10, 0.1, 0, None) == -1   # beyond the capacity
    assert L.fitgnn_appnp_blocks_f32(None, None, None, None, 5, 64, 100, None, None, None, None, 12, 0, 0.1, 0, None) == -1   # K >= 1
    n13 = [None] * 13
    n9 = [None] * 9
    assert L.fitgnn_batch_gather(128, *n13, 11, None, 1, 11, 64, 64, 64, 64, *n9, 11, None, None, None, None, None, 0, None) == -1   # NULL arrays
    assert L.fitgnn_batch_gather(128, *n13, 8, None, 1, 11, 64, 64, 64, 64, *n9, 11, None, None, None, None, None, 0, None) == -1    # ld_ax_g < K
