// spmm.hip -- CSR SpMM with LDS-staged row windows for block-diagonal subgraph batches (gfx950).
//
// Replaces the propagate step of torch_geometric's GCNConv/SAGEConv/GINConv/APPNP as called from
// FIT-GNN network.py:31,60,90,126,161,197 (gather x[row] -> multiply -> scatter-add into col), and its
// autograd backward (the same product with the transposed CSR).
//
// Mapping (MI355X): one 256-thread workgroup (4 waves) per (row tile, 64*VEC-column slab).
//   1. the tile's column window -- rows [win_begin, win_begin+win_rows) of X, slab columns only -- is
//      streamed HBM -> registers -> LDS with 16-byte coalesced loads (1 KiB per wave instruction);
//   2. each wave owns rows of the tile round-robin; the row's (col,val) pairs are read 64 at a time,
//      one pair per lane, and broadcast with v_readlane; the lane accumulates its VEC columns from
//      LDS (window hit, wave-uniform test) or straight from global/L2 (miss);
//   3. bias / ELU / dropout are applied in registers and the row is stored with one coalesced write.
// HBM traffic per tile ~= window rows read once + tile rows written once: the algorithmic minimum
// 4*H*(N+N) + 8*nnz for a block-diagonal batch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "fitgnn_hip.h"

namespace {

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;
constexpr int kLdsBytes = 64 * 1024;

template <int VEC> struct Pack;
template <> struct Pack<4> {
    using T = float4;
    static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ void fma(T &a, float v, const T &x) {
        a.x = fmaf(v, x.x, a.x); a.y = fmaf(v, x.y, a.y); a.z = fmaf(v, x.z, a.z); a.w = fmaf(v, x.w, a.w);
    }
    static __device__ __forceinline__ float get(const T &a, int i) { return i == 0 ? a.x : i == 1 ? a.y : i == 2 ? a.z : a.w; }
    static __device__ __forceinline__ void set(T &a, int i, float f) { if (i == 0) a.x = f; else if (i == 1) a.y = f; else if (i == 2) a.z = f; else a.w = f; }
};
template <> struct Pack<1> {
    using T = float;
    static __device__ __forceinline__ T zero() { return 0.f; }
    static __device__ __forceinline__ void fma(T &a, float v, const T &x) { a = fmaf(v, x, a); }
    static __device__ __forceinline__ float get(const T &a, int) { return a; }
    static __device__ __forceinline__ void set(T &a, int, float f) { a = f; }
};

template <int VEC>
__global__ __launch_bounds__(kThreads) void spmm_tile_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const float *__restrict__ X, int64_t ldx, float *__restrict__ Y, int64_t ldy, int32_t H,
    const fitgnn_tile_t *__restrict__ tiles, int32_t n_tiles, int32_t tiles_per_xcd, int32_t lds_rows,
    const float *__restrict__ bias, uint32_t epi, float p_drop, uint64_t seed, const uint8_t *__restrict__ mask) {
    using P = Pack<VEC>;
    using T = typename P::T;
    constexpr int SLAB = 64 * VEC;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T *lds = reinterpret_cast<T *>(lds_raw);

    // XCD-aware tile mapping: consecutive block ids are dealt round-robin over the 8 XCDs, so give
    // each XCD a contiguous range of tiles (neighbouring tiles share L2 lines on the miss path).
    const int bid = blockIdx.x;
    const int t = (bid & 7) * tiles_per_xcd + (bid >> 3);
    if (t >= n_tiles) return;
    const fitgnn_tile_t tile = tiles[t];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col0 = blockIdx.y * SLAB + lane * VEC;
    const bool live = col0 + VEC <= H;
    const int win_begin = tile.win_begin;
    const int win_rows = min(tile.win_rows, lds_rows);

    // ---- stage the window: HBM -> LDS, 4 independent 16-byte loads in flight per lane ----
    {
        int r = wave;
        for (; r + 3 * kWaves < win_rows; r += 4 * kWaves) {
            T v0 = P::zero(), v1 = P::zero(), v2 = P::zero(), v3 = P::zero();
            if (live) {
                v0 = *reinterpret_cast<const T *>(X + (int64_t)(win_begin + r) * ldx + col0);
                v1 = *reinterpret_cast<const T *>(X + (int64_t)(win_begin + r + kWaves) * ldx + col0);
                v2 = *reinterpret_cast<const T *>(X + (int64_t)(win_begin + r + 2 * kWaves) * ldx + col0);
                v3 = *reinterpret_cast<const T *>(X + (int64_t)(win_begin + r + 3 * kWaves) * ldx + col0);
            }
            lds[(r)*64 + lane] = v0;
            lds[(r + kWaves) * 64 + lane] = v1;
            lds[(r + 2 * kWaves) * 64 + lane] = v2;
            lds[(r + 3 * kWaves) * 64 + lane] = v3;
        }
        for (; r < win_rows; r += kWaves) {
            T v0 = P::zero();
            if (live) v0 = *reinterpret_cast<const T *>(X + (int64_t)(win_begin + r) * ldx + col0);
            lds[r * 64 + lane] = v0;
        }
    }
    __syncthreads();

    const float keep_scale = (epi & FITGNN_EPI_DROPOUT) ? 1.0f / (1.0f - p_drop) : 1.0f;
    for (int row = tile.row_begin + wave; row < tile.row_end; row += kWaves) {
        const int e0 = __builtin_amdgcn_readfirstlane(rowptr[row]);
        const int e1 = __builtin_amdgcn_readfirstlane(rowptr[row + 1]);
        T acc = P::zero();
        for (int base = e0; base < e1; base += 64) {
            const int cnt = min(64, e1 - base);
            int my_c = 0;
            float my_v = 0.f;
            if (lane < cnt) {
                my_c = col[base + lane];
                my_v = val[base + lane];
            }
            int k = 0;
            for (; k + 4 <= cnt; k += 4) {
                const int c0 = __builtin_amdgcn_readlane(my_c, k) - win_begin;
                const int c1 = __builtin_amdgcn_readlane(my_c, k + 1) - win_begin;
                const int c2 = __builtin_amdgcn_readlane(my_c, k + 2) - win_begin;
                const int c3 = __builtin_amdgcn_readlane(my_c, k + 3) - win_begin;
                const float w0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), k));
                const float w1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), k + 1));
                const float w2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), k + 2));
                const float w3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), k + 3));
                const bool in0 = (unsigned)c0 < (unsigned)win_rows, in1 = (unsigned)c1 < (unsigned)win_rows;
                const bool in2 = (unsigned)c2 < (unsigned)win_rows, in3 = (unsigned)c3 < (unsigned)win_rows;
                T x0, x1, x2, x3;
                if (in0 && in1 && in2 && in3) {  // wave-uniform: four LDS reads in flight
                    x0 = lds[c0 * 64 + lane]; x1 = lds[c1 * 64 + lane];
                    x2 = lds[c2 * 64 + lane]; x3 = lds[c3 * 64 + lane];
                } else {
                    x0 = x1 = x2 = x3 = P::zero();
                    if (in0) x0 = lds[c0 * 64 + lane]; else if (live) x0 = *reinterpret_cast<const T *>(X + (int64_t)(c0 + win_begin) * ldx + col0);
                    if (in1) x1 = lds[c1 * 64 + lane]; else if (live) x1 = *reinterpret_cast<const T *>(X + (int64_t)(c1 + win_begin) * ldx + col0);
                    if (in2) x2 = lds[c2 * 64 + lane]; else if (live) x2 = *reinterpret_cast<const T *>(X + (int64_t)(c2 + win_begin) * ldx + col0);
                    if (in3) x3 = lds[c3 * 64 + lane]; else if (live) x3 = *reinterpret_cast<const T *>(X + (int64_t)(c3 + win_begin) * ldx + col0);
                }
                P::fma(acc, w0, x0); P::fma(acc, w1, x1); P::fma(acc, w2, x2); P::fma(acc, w3, x3);
            }
            for (; k < cnt; ++k) {
                const int c = __builtin_amdgcn_readlane(my_c, k) - win_begin;
                const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), k));
                T x = P::zero();
                if ((unsigned)c < (unsigned)win_rows) x = lds[c * 64 + lane];
                else if (live) x = *reinterpret_cast<const T *>(X + (int64_t)(c + win_begin) * ldx + col0);
                P::fma(acc, w, x);
            }
        }
        if (!live) continue;
        // ---- fused epilogue (GCNConv bias, network.py:32 F.elu, :33 F.dropout) ----
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float z = P::get(acc, i);
            if (epi & FITGNN_EPI_BIAS) z += bias[col0 + i];
            if (epi & FITGNN_EPI_ELU) z = z > 0.f ? z : expm1f(z);
            if (epi & FITGNN_EPI_DROPOUT) {
                const uint64_t idx = (uint64_t)row * (uint64_t)H + (uint64_t)(col0 + i);
                const bool keep = mask ? (mask[idx] != 0) : fitgnn::dropout_keep(seed, idx, p_drop);
                z = keep ? z * keep_scale : 0.f;
            }
            P::set(acc, i, z);
        }
        *reinterpret_cast<T *>(Y + (int64_t)row * ldy + col0) = acc;
    }
}

template <int VEC>
int launch(const int32_t *rowptr, const int32_t *col, const float *val, const float *X, int64_t ldx, float *Y,
           int64_t ldy, int32_t H, const fitgnn_tile_t *tiles, int32_t n_tiles, const float *bias, uint32_t epi,
           float p_drop, uint64_t seed, const uint8_t *mask, hipStream_t s) {
    constexpr int SLAB = 64 * VEC;
    const int n_slabs = (H + SLAB - 1) / SLAB;
    const int tiles_per_xcd = (n_tiles + 7) / 8;
    const int lds_rows = kLdsBytes / (SLAB * 4);
    dim3 grid(tiles_per_xcd * 8, n_slabs);
    hipLaunchKernelGGL(spmm_tile_kernel<VEC>, grid, dim3(kThreads), kLdsBytes, s, rowptr, col, val, X, ldx, Y, ldy, H,
                       tiles, n_tiles, tiles_per_xcd, lds_rows, bias, epi, p_drop, seed, mask);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int fitgnn_spmm_max_window_rows(int32_t H) {
    const bool vec = (H % 4) == 0;
    return kLdsBytes / ((vec ? 256 : 64) * 4);
}

extern "C" int fitgnn_spmm_csr_f32(const int32_t *rowptr, const int32_t *col, const float *val, const float *X,
                                   int64_t ldx, float *Y, int64_t ldy, int32_t n_rows, int32_t H,
                                   const fitgnn_tile_t *tiles, int32_t n_tiles, const float *bias, uint32_t epilogue,
                                   float p_drop, uint64_t seed, const uint8_t *mask, void *stream) {
    if (n_rows < 0 || H < 0 || n_tiles < 0) return FITGNN_E_BADARG;
    if (n_rows == 0 || H == 0 || n_tiles == 0) return 0;
    // col/val may be NULL only for a matrix without non-zeros (they are then never dereferenced)
    if (!rowptr || !X || !Y || !tiles) return FITGNN_E_BADARG;
    if ((epilogue & FITGNN_EPI_BIAS) && !bias) return FITGNN_E_BADARG;
    if ((epilogue & FITGNN_EPI_DROPOUT) && !(p_drop >= 0.f && p_drop < 1.f)) return FITGNN_E_BADARG;
    if (ldx < H || ldy < H) return FITGNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (H % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && (((uintptr_t)X | (uintptr_t)Y) % 16 == 0);
    if (vec) return launch<4>(rowptr, col, val, X, ldx, Y, ldy, H, tiles, n_tiles, bias, epilogue, p_drop, seed, mask, s);
    return launch<1>(rowptr, col, val, X, ldx, Y, ldy, H, tiles, n_tiles, bias, epilogue, p_drop, seed, mask, s);
}
