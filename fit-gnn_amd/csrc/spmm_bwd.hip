// spmm_bwd.hip -- backward SpMM with the epilogue's backward folded into its operand staging.
//
// For out = dropout(ELU(A_hat H + b)) (GCNConv + F.elu + F.dropout, network.py:31-33) the backward needs
//     dZ = dOut . dropout' . ELU'(out)          (elementwise)
//     db = column sums of dZ                      (GCNConv bias)
//     dH = A_hat^T dZ                             (propagate, transposed CSR)
// fitgnn_epilogue_bwd_f32 + fitgnn_spmm_csr_f32 do this with dZ written to and re-read from HBM (2 x 4H bytes per
// row).  Here dZ exists only in LDS: while a tile's window rows are staged, each row of dOut / out is read once,
// transformed in registers and stored to LDS; the bias gradient is summed over the rows a tile OWNS (every row is
// owned by exactly one tile: tiles partition the rows and a tile's contiguous window covers its own rows) and
// reduced over tiles in a fixed order.  HEAD mode also folds the output head (network.py:34): dOut = dy @ Wl is
// formed on the fly and dWl = dy^T @ out accumulated, as in fitgnn_epilogue_bwd_head_f32.
// Requirements (checked by the launcher): H % 4 == 0, contiguous windows (no lcol / win_cols), window_rows <= 16.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "common.h"
#include "fitgnn_hip.h"

namespace {

constexpr int kWaves = 4;
constexpr int kThreads = 256;
constexpr int kRows = 16;    // LDS window rows (the high-occupancy SpMM configuration)
constexpr int kMPR = 16;     // staged CSR entries per window row
constexpr int kHeadC = 3;    // widest head folded here (wider heads use the unfused kernels)

struct XformArgs {
    const float *dOut;  // [rows x H] (HEAD: unused)
    const float *out;   // forward output of the layer (post-dropout), [rows x H]
    const float *dy;    // HEAD: [rows x C]
    const float *Wl;    // HEAD: [C x H]
    int32_t C;
    int32_t n_rows;
    uint32_t epi;
    float p_drop;
    uint64_t seed;
    const uint8_t *mask;
    float *partial;      // [n_tiles x Wp] per-tile gradient partials (db at [0,H), dWl row c at [H + c*H, ...)) or null
    int32_t Wp;          // partial row width: H (db only) or H * (1 + C)
    int32_t want_dWl;
};

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void f4fma(float4 &a, float w, const float4 &x) {
    a.x = fmaf(w, x.x, a.x); a.y = fmaf(w, x.y, a.y); a.z = fmaf(w, x.z, a.z); a.w = fmaf(w, x.w, a.w);
}
// sum over the four 16-lane row groups of a wave (lanes l, l^16, l^32, l^48), fixed order
__device__ __forceinline__ float4 rowgroup_sum(float4 v) {
    v.x += __shfl_xor(v.x, 16); v.y += __shfl_xor(v.y, 16); v.z += __shfl_xor(v.z, 16); v.w += __shfl_xor(v.w, 16);
    v.x += __shfl_xor(v.x, 32); v.y += __shfl_xor(v.y, 32); v.z += __shfl_xor(v.z, 32); v.w += __shfl_xor(v.w, 32);
    return v;
}

// Staging layout: wave w owns the 64-column quarter [w*64, w*64+64) of the workgroup's 256-column slab for ALL the
// window rows -- lane = (row group rg = lane/16, column quad q = lane%16), so one load instruction covers four
// rows x 256 B.  The gradient partials over a tile's rows are then complete inside one wave (two cross-lane adds
// over the row groups): no cross-wave reduction, no extra LDS, no extra barrier.  The LDS window keeps the
// [row][64 x float4] layout the row loop reads (slot w*16+q == the row loop's lane).
template <bool HEAD, int EPI>
__global__ __launch_bounds__(kThreads, (EPI >= 0 ? (HEAD ? 5 : 7) : (HEAD ? 6 : 8))) void spmm_bwd_fused_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    float *__restrict__ Y, int64_t ldy, int32_t H, const fitgnn_tile_t *__restrict__ tiles, int32_t n_tiles, int32_t n_slabs,
    XformArgs xa) {
    constexpr int SLAB = 256;
    __shared__ float4 lds[kRows * 64];
    __shared__ int32_t s_rp[kRows + 4];
    __shared__ int32_t s_col[kRows * kMPR];
    __shared__ float s_val[kRows * kMPR];
    constexpr int meta_cap = kRows * kMPR;

    const int bid = blockIdx.x;
    const int seq = bid >> 3;
    const int slab = seq % n_slabs;
    const int t = (seq / n_slabs) * 8 + (bid & 7);
    if (t >= n_tiles) return;
    const fitgnn_tile_t tile = tiles[t];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col0 = slab * SLAB + lane * 4;  // row-loop columns of this lane
    const bool live = col0 + 4 <= H;
    const int colL = live ? col0 : max(H - 4, 0);
    const int win_begin = tile.win_begin;
    const int win_rows = min(tile.win_rows, kRows);
    const int tile_rows = tile.row_end - tile.row_begin;
    const int rp_rows = min(tile_rows, kRows);
    const int E0 = tile.nnz_begin;
    const int n_meta = min(tile.nnz_end - E0, meta_cap);

    const uint32_t epi = EPI >= 0 ? (uint32_t)EPI : xa.epi;
    const uint8_t *const mask = EPI >= 0 ? nullptr : xa.mask;
    const float scale = (epi & FITGNN_EPI_DROPOUT) ? 1.0f / (1.0f - xa.p_drop) : 1.0f;
    const float unscale = (epi & FITGNN_EPI_DROPOUT) ? (1.0f - xa.p_drop) : 1.0f;
    const uint32_t thresh = fitgnn::dropout_threshold(xa.p_drop);
    const uint64_t seed = fitgnn::resolve_seed(xa.seed, xa.epi);
    // dZ of operand row `grow` at columns [cbase, cbase+4), from g (dOut row; HEAD: dy row @ Wl) and o (out row)
    auto transform = [&](int64_t grow, int cbase, float4 g, float4 o) -> float4 {
        float gv[4] = {g.x, g.y, g.z, g.w};
        const float ov[4] = {o.x, o.y, o.z, o.w};
        const uint64_t idx0 = (uint64_t)grow * (uint64_t)H + (uint64_t)cbase;
        uint64_t bits = 0;
        if ((epi & FITGNN_EPI_DROPOUT) && !mask) bits = fitgnn::dropout_bits(seed, idx0 >> 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float d = gv[i];
            if (epi & FITGNN_EPI_DROPOUT) {
                const bool keep = mask ? (mask[idx0 + i] != 0) : fitgnn::dropout_keep(bits, (int)((idx0 + i) & 3), thresh);
                d = keep ? d * scale : 0.f;
            }
            if (epi & FITGNN_EPI_ELU) {
                const float e = ov[i] * unscale;
                d = e > 0.f ? d : d * (e + 1.0f);
            }
            gv[i] = d;
        }
        return make_float4(gv[0], gv[1], gv[2], gv[3]);
    };

    // ---- stage the window: read dOut / out (or dy / out), transform, keep dZ in LDS only ----
    {
        const int q = lane & 15, rg = lane >> 4;
        const int scol0 = slab * SLAB + wave * 64 + q * 4;
        const bool slive = scol0 + 4 <= H;
        const int scolL = slive ? scol0 : max(H - 4, 0);
        float4 wl[kHeadC], wacc[kHeadC];
        float4 dbacc = f4zero();
        if (HEAD) {
#pragma unroll
            for (int c = 0; c < kHeadC; ++c) {
                const float4 w = *reinterpret_cast<const float4 *>(xa.Wl + (int64_t)min(c, xa.C - 1) * H + scolL);
                wl[c] = c < xa.C ? w : f4zero();
                wacc[c] = f4zero();
            }
        }
        float4 g[4], o[4];
        float dyl[4][kHeadC];
        int mc = 0, rpv = 0;
        float mv = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // always four loads per stream (static vmcnt); out-of-window rows are clamped
            const int r = j * 4 + rg;
            const int64_t grow = min(win_begin + min(r, max(win_rows - 1, 0)), xa.n_rows - 1);
            o[j] = *reinterpret_cast<const float4 *>(xa.out + grow * H + scolL);
            if (HEAD) {
#pragma unroll
                for (int c = 0; c < kHeadC; ++c) dyl[j][c] = xa.dy[grow * xa.C + min(c, xa.C - 1)];
            } else {
                g[j] = *reinterpret_cast<const float4 *>(xa.dOut + grow * H + scolL);
            }
        }
        if ((int)threadIdx.x <= rp_rows) rpv = rowptr[tile.row_begin + threadIdx.x];
        if ((int)threadIdx.x < n_meta) { mc = col[E0 + threadIdx.x]; mv = val[E0 + threadIdx.x]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = j * 4 + rg;
            const bool valid = r < win_rows;
            const int64_t grow = min(win_begin + min(r, max(win_rows - 1, 0)), xa.n_rows - 1);
            const bool own = valid && grow >= tile.row_begin && grow < tile.row_end;  // rows this tile owns feed the sums
            float4 gg = HEAD ? f4zero() : g[j];
            if (HEAD) {
#pragma unroll
                for (int c = 0; c < kHeadC; ++c) f4fma(gg, dyl[j][c], wl[c]);
            }
            const float4 dz = transform(grow, scolL, gg, o[j]);
            if (valid) lds[r * 64 + wave * 16 + q] = dz;
            if (own) { dbacc.x += dz.x; dbacc.y += dz.y; dbacc.z += dz.z; dbacc.w += dz.w; }
            if (HEAD) {
#pragma unroll
                for (int c = 0; c < kHeadC; ++c) f4fma(wacc[c], own ? dyl[j][c] : 0.f, o[j]);
            }
        }
        if ((int)threadIdx.x <= rp_rows) s_rp[threadIdx.x] = rpv;
        if ((int)threadIdx.x < n_meta) { s_col[threadIdx.x] = mc; s_val[threadIdx.x] = mv; }
        for (int i = threadIdx.x + kThreads; i < n_meta; i += kThreads) { s_col[i] = col[E0 + i]; s_val[i] = val[E0 + i]; }
        // ---- gradient partials of this tile: complete inside the wave ----
        if (xa.partial) {
            float *prow = xa.partial + (int64_t)t * xa.Wp;
            dbacc = rowgroup_sum(dbacc);
            if (rg == 0 && slive) *reinterpret_cast<float4 *>(prow + scol0) = dbacc;
            if (HEAD && xa.want_dWl) {
#pragma unroll
                for (int c = 0; c < kHeadC; ++c) {
                    if (c < xa.C) {
                        const float4 wsum = rowgroup_sum(wacc[c]);
                        if (rg == 0 && slive) *reinterpret_cast<float4 *>(prow + (int64_t)(1 + c) * H + scol0) = wsum;
                    }
                }
            }
        }
    }
    __syncthreads();

    // operand row fetched outside the window (hub subgraphs cut into pieces): transform it on the fly
    auto miss_row = [&](int grow) -> float4 {
        const float4 o = *reinterpret_cast<const float4 *>(xa.out + (int64_t)grow * H + colL);
        float4 g = f4zero();
        if (HEAD) {
            for (int c = 0; c < xa.C; ++c)
                f4fma(g, xa.dy[(int64_t)grow * xa.C + c], *reinterpret_cast<const float4 *>(xa.Wl + (int64_t)c * H + colL));
        } else {
            g = *reinterpret_cast<const float4 *>(xa.dOut + (int64_t)grow * H + colL);
        }
        return transform(grow, colL, g, o);
    };
    for (int row = tile.row_begin + wave; row < tile.row_end; row += kWaves) {
        const int lr = row - tile.row_begin;
        int e0, e1;
        if (lr < rp_rows) { e0 = s_rp[lr]; e1 = s_rp[lr + 1]; } else { e0 = rowptr[row]; e1 = rowptr[row + 1]; }
        e0 = __builtin_amdgcn_readfirstlane(e0);
        e1 = __builtin_amdgcn_readfirstlane(e1);
        float4 acc = f4zero();
        for (int base = e0; base < e1; base += 64) {
            const int cnt = min(64, e1 - base);
            int my_c = 0;
            float my_v = 0.f;
            if (lane < cnt) {
                const int i = base + lane - E0;
                if (i < n_meta) { my_c = s_col[i]; my_v = s_val[i]; } else { my_c = col[base + lane]; my_v = val[base + lane]; }
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
                float4 x0, x1, x2, x3;
                if (in0 && in1 && in2 && in3) {  // wave-uniform: four LDS reads in flight
                    x0 = lds[c0 * 64 + lane]; x1 = lds[c1 * 64 + lane];
                    x2 = lds[c2 * 64 + lane]; x3 = lds[c3 * 64 + lane];
                } else {
                    if (in0) x0 = lds[c0 * 64 + lane]; else x0 = miss_row(c0 + win_begin);
                    if (in1) x1 = lds[c1 * 64 + lane]; else x1 = miss_row(c1 + win_begin);
                    if (in2) x2 = lds[c2 * 64 + lane]; else x2 = miss_row(c2 + win_begin);
                    if (in3) x3 = lds[c3 * 64 + lane]; else x3 = miss_row(c3 + win_begin);
                }
                f4fma(acc, w0, x0); f4fma(acc, w1, x1); f4fma(acc, w2, x2); f4fma(acc, w3, x3);
            }
            for (; k < cnt; ++k) {
                const int c = __builtin_amdgcn_readlane(my_c, k) - win_begin;
                const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), k));
                float4 x;
                if ((unsigned)c < (unsigned)win_rows) x = lds[c * 64 + lane]; else x = miss_row(c + win_begin);
                f4fma(acc, w, x);
            }
        }
        if (live) *reinterpret_cast<float4 *>(Y + (int64_t)row * ldy + col0) = acc;
    }
}

// src [n x W] -> dst [G x W]: dst row g = src rows g, g+G, g+2G, ... summed in that order.  Applied three times
// (n_tiles -> 256 -> 16 -> 1) it is a fixed-order column sum whose every stage has plenty of independent loads.
// Columns >= split go to dst2 (the last stage writes db and dWl straight to their destinations).
__global__ __launch_bounds__(256) void colsum_stage_kernel(const float *__restrict__ src, int32_t n, int32_t W, int32_t G,
                                                           float *__restrict__ dst, float *__restrict__ dst2, int32_t split) {
    const int h = blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y;
    if (h >= W) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = g;
    for (; r + 3 * G < n; r += 4 * G) {
        s0 += src[(int64_t)r * W + h];
        s1 += src[(int64_t)(r + G) * W + h];
        s2 += src[(int64_t)(r + 2 * G) * W + h];
        s3 += src[(int64_t)(r + 3 * G) * W + h];
    }
    for (; r < n; r += G) s0 += src[(int64_t)r * W + h];
    const float v = (s0 + s1) + (s2 + s3);
    if (dst2 && h >= split) dst2[(int64_t)g * (W - split) + (h - split)] = v;
    else if (dst) dst[(int64_t)g * W + h] = v;
}
constexpr int kG1 = 256, kG2 = 16;

inline size_t align_up(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace

extern "C" size_t fitgnn_spmm_epilogue_bwd_workspace_bytes(int32_t n_tiles, int32_t H, int32_t C) {
    if (n_tiles <= 0 || H <= 0 || C < 0) return 0;
    const size_t W = (size_t)H * (size_t)(1 + C);
    return align_up((size_t)n_tiles * W * 4) + align_up((size_t)kG1 * W * 4) + align_up((size_t)kG2 * W * 4);
}

extern "C" int fitgnn_spmm_epilogue_bwd_supported(int32_t H, int32_t C, int32_t window_rows) {
    // (a head's dy values are fetched one class per lane by the lanes that own columns of the slab: not more classes than
    // the last 256-column slab has such lanes -- the condition of fitgnn_epilogue_bwd_head_supported)
    return (H > 0 && H % 4 == 0 && C >= 0 && C <= kHeadC && C <= ((H - 1) % 256 + 1) / 4 && window_rows > 0 &&
            window_rows <= kRows) ? 1 : 0;
}

extern "C" int fitgnn_spmm_epilogue_bwd_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                                            const fitgnn_tile_t *tiles, int32_t n_tiles, int32_t window_rows,
                                            const float *dOut, const float *dy, const float *Wl, int32_t C, const float *out,
                                            float *dH, int32_t n_rows, int32_t H, uint32_t epilogue, float p_drop, uint64_t seed,
                                            const uint8_t *mask, float *db, float *dWl, void *work, size_t work_bytes,
                                            void *stream) {
    if (n_rows < 0 || H < 0 || n_tiles < 0) return FITGNN_E_BADARG;
    if (n_rows == 0 || H == 0 || n_tiles == 0) return 0;
    const bool head = dOut == nullptr;
    if (!fitgnn_spmm_epilogue_bwd_supported(H, head ? C : 0, window_rows)) return FITGNN_E_BADARG;
    if (!rowptr || !tiles || !out || !dH) return FITGNN_E_BADARG;
    if (head && (!dy || !Wl || C < 1)) return FITGNN_E_BADARG;
    if (!head && dWl) return FITGNN_E_BADARG;
    if ((epilogue & FITGNN_EPI_DROPOUT) && !(p_drop >= 0.f && p_drop < 1.f)) return FITGNN_E_BADARG;
    if ((((uintptr_t)dOut | (uintptr_t)out | (uintptr_t)dH | (uintptr_t)Wl) % 16) != 0) return FITGNN_E_ALIGN;
    const int Cw = dWl ? C : 0;
    if ((db || dWl) && (!work || work_bytes < fitgnn_spmm_epilogue_bwd_workspace_bytes(n_tiles, H, Cw))) return FITGNN_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int Wp = H * (1 + Cw);
    float *partial = (db || dWl) ? (float *)work : nullptr;
    XformArgs xa{dOut, out, dy, Wl, C, n_rows, epilogue, p_drop, seed, mask, partial, Wp, dWl ? 1 : 0};
    const int n_slabs = (H + 255) / 256;
    const int tiles_per_xcd = (n_tiles + 7) / 8;
    dim3 grid(tiles_per_xcd * 8 * n_slabs);
    // the training configuration (ELU + hash dropout, network.py:32-33) has a straight-line instantiation
    constexpr int kTrainEpi = FITGNN_EPI_ELU | FITGNN_EPI_DROPOUT;
    const bool train_epi = (epilogue & (FITGNN_EPI_ELU | FITGNN_EPI_DROPOUT)) == (uint32_t)kTrainEpi && !mask;
#define FITGNN_LAUNCH_FUSED(HEAD_, EPI_)                                                                                        \
    hipLaunchKernelGGL((spmm_bwd_fused_kernel<HEAD_, EPI_>), grid, dim3(kThreads), 0, s, rowptr, col, val, dH, (int64_t)H, H, tiles, \
                       n_tiles, n_slabs, xa)
    if (head) { if (train_epi) FITGNN_LAUNCH_FUSED(true, kTrainEpi); else FITGNN_LAUNCH_FUSED(true, -1); }
    else { if (train_epi) FITGNN_LAUNCH_FUSED(false, kTrainEpi); else FITGNN_LAUNCH_FUSED(false, -1); }
#undef FITGNN_LAUNCH_FUSED
    if (partial) {  // [n_tiles x Wp] -> [kG1 x Wp] -> [kG2 x Wp] -> db | dWl
        float *l1 = (float *)((char *)work + align_up((size_t)n_tiles * Wp * 4));
        float *l2 = (float *)((char *)l1 + align_up((size_t)kG1 * Wp * 4));
        const int gx = (Wp + 255) / 256;
        hipLaunchKernelGGL(colsum_stage_kernel, dim3(gx, kG1), dim3(256), 0, s, partial, n_tiles, Wp, kG1, l1, nullptr, 0);
        hipLaunchKernelGGL(colsum_stage_kernel, dim3(gx, kG2), dim3(256), 0, s, l1, kG1, Wp, kG2, l2, nullptr, 0);
        hipLaunchKernelGGL(colsum_stage_kernel, dim3(gx, 1), dim3(256), 0, s, l2, kG2, Wp, 1, db, dWl, H);
    }
    return (int)hipGetLastError();
}
