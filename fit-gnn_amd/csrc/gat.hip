// gat.hip -- graph-attention kernels on the CSR batch (PyG GATConv with FIT-GNN's defaults: heads = 1,
// negative_slope = 0.2, add_self_loops, no attention dropout; network.py:13 with --layer_name GATConv).
//
//   h = x W^T;  s_ij = a_src.h_j + a_dst.h_i;  e_ij = LeakyReLU(s_ij);  alpha_ij = softmax_j(e_ij) over the
//   incoming edges of i (CSR row i);  out_i = sum_j alpha_ij h_j + bias.
// The aggregation itself is fitgnn_spmm_csr_f32 with val = alpha.  This file adds the per-node score dots, the
// per-row edge softmax, the per-edge gradient d(alpha_ij) = dOut_i . h_j (SDDMM) and the softmax/LeakyReLU
// backward.  Rows of h / dOut are read with 16-byte lane loads; per-row reductions are 64-lane butterflies.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "fitgnn_hip.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// a_src[i] = h_i . att_src,  a_dst[i] = h_i . att_dst          (one wave per row)
__global__ __launch_bounds__(256) void gat_scores_kernel(const float *__restrict__ h, int64_t ldh, int32_t n, int32_t C,
                                                         const float *__restrict__ att_src, const float *__restrict__ att_dst,
                                                         float *__restrict__ a_src, float *__restrict__ a_dst) {
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const float *hr = h + (int64_t)row * ldh;
    float s = 0.f, d = 0.f;
    if ((C & 3) == 0 && (ldh & 3) == 0) {
        for (int c = lane * 4; c < C; c += 256) {
            const float4 v = *reinterpret_cast<const float4 *>(hr + c);
            const float4 as = *reinterpret_cast<const float4 *>(att_src + c);
            const float4 ad = *reinterpret_cast<const float4 *>(att_dst + c);
            s += v.x * as.x + v.y * as.y + v.z * as.z + v.w * as.w;
            d += v.x * ad.x + v.y * ad.y + v.z * ad.z + v.w * ad.w;
        }
    } else {
        for (int c = lane; c < C; c += 64) { s += hr[c] * att_src[c]; d += hr[c] * att_dst[c]; }
    }
    s = wave_sum(s);
    d = wave_sum(d);
    if (lane == 0) { a_src[row] = s; a_dst[row] = d; }
}

// alpha over each CSR row: softmax_j LeakyReLU(a_src[col] + a_dst[row]).
// Rows of a coarsened-subgraph union are short (a leaf of a star has 2-4 entries, only the centres are long): one THREAD per row
// for rows of at most kShortRow entries (its scores stay in registers between the three passes), and the long rows of a wave's 64
// rows one after the other by the whole wave (entries on lanes, 64-lane butterflies).  One wave per row left 60 of 64 lanes idle
// on three dependent passes: 3.4 ms per launch on the S-products union.
constexpr int kShortRow = 8;

__global__ __launch_bounds__(256) void gat_edge_softmax_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                               const float *__restrict__ a_src, const float *__restrict__ a_dst,
                                                               float slope, int32_t n, float *__restrict__ alpha) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool on = row < n;
    int e0 = 0, len = 0;
    float ad = 0.f;
    if (on) {
        e0 = rowptr[row];
        len = rowptr[row + 1] - e0;
        ad = a_dst[row];
    }
    if (on && len <= kShortRow) {
        float sc[kShortRow];
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < kShortRow; ++k) {
            float v = -INFINITY;
            if (k < len) {
                v = a_src[col[e0 + k]] + ad;
                v = v > 0.f ? v : slope * v;
            }
            sc[k] = v;
            m = fmaxf(m, v);
        }
        float z = 0.f;
#pragma unroll
        for (int k = 0; k < kShortRow; ++k) {
            const float p = k < len ? __expf(sc[k] - m) : 0.f;
            sc[k] = p;
            z += p;
        }
        const float inv = z > 0.f ? 1.0f / z : 0.f;
#pragma unroll
        for (int k = 0; k < kShortRow; ++k)
            if (k < len) alpha[e0 + k] = sc[k] * inv;
    }
    unsigned long long longs = __ballot(on && len > kShortRow);
    while (longs) {   // wave-uniform
        const int src = __ffsll((long long)longs) - 1;
        longs &= longs - 1;
        const int b0 = __shfl(e0, src, 64), b1 = b0 + __shfl(len, src, 64);
        const float adr = __shfl(ad, src, 64);
        float m = -INFINITY;
        for (int e = b0 + lane; e < b1; e += 64) {
            float v = a_src[col[e]] + adr;
            v = v > 0.f ? v : slope * v;
            m = fmaxf(m, v);
        }
        m = wave_max(m);
        float z = 0.f;
        for (int e = b0 + lane; e < b1; e += 64) {
            float v = a_src[col[e]] + adr;
            v = v > 0.f ? v : slope * v;
            const float p = __expf(v - m);
            alpha[e] = p;
            z += p;
        }
        z = wave_sum(z);
        const float inv = z > 0.f ? 1.0f / z : 0.f;
        for (int e = b0 + lane; e < b1; e += 64) alpha[e] *= inv;
    }
}

// d_alpha[e] = dOut[row] . h[col[e]]        (SDDMM; one wave per row keeps dOut[row] in registers)
// sel (may be NULL): the launch covers the rows sel[0 .. n) only and dOut is COMPACT (row i of dOut belongs to row sel[i]); entries of
// other rows are not written (the caller zeroes dalpha)
template <int MAXV>  // MAXV float4 per lane: C <= 256 * MAXV
__global__ __launch_bounds__(256) void gat_sddmm_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                        const float *__restrict__ dOut, int64_t ldo,
                                                        const float *__restrict__ h, int64_t ldh, int32_t n, int32_t C,
                                                        float *__restrict__ dalpha, const int64_t *__restrict__ sel) {
    const int wi = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wi >= n) return;
    const int row = sel ? (int)sel[wi] : wi;
    float4 g[MAXV];
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
        const int c = (v * 64 + lane) * 4;
        g[v] = c < C ? *reinterpret_cast<const float4 *>(dOut + (int64_t)wi * ldo + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int e0 = rowptr[row], e1 = rowptr[row + 1];
    // the row's column ids sit on the lanes; four edges per step: their operand rows are all in flight before the first
    // dot product (one edge at a time is a chain col -> row fetch -> reduction per edge)
    for (int base = e0; base < e1; base += 64) {
        const int cnt = min(64, e1 - base);
        const int my_c = lane < cnt ? col[base + lane] : 0;
        for (int k = 0; k < cnt; k += 4) {
            const int n4 = min(4, cnt - k);  // wave-uniform
            float4 x[4][MAXV];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cj = __builtin_amdgcn_readlane(my_c, k + min(j, n4 - 1));
                const float *hr = h + (int64_t)cj * ldh;
#pragma unroll
                for (int v = 0; v < MAXV; ++v) {
                    const int c = (v * 64 + lane) * 4;
                    x[j][v] = c < C ? *reinterpret_cast<const float4 *>(hr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            float sj[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s = 0.f;
#pragma unroll
                for (int v = 0; v < MAXV; ++v)
                    s += g[v].x * x[j][v].x + g[v].y * x[j][v].y + g[v].z * x[j][v].z + g[v].w * x[j][v].w;
                sj[j] = s;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) sj[j] = wave_sum(sj[j]);
            if (lane < n4) dalpha[base + k + lane] = lane == 0 ? sj[0] : lane == 1 ? sj[1] : lane == 2 ? sj[2] : sj[3];
        }
    }
}
__global__ __launch_bounds__(256) void gat_sddmm_scalar_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                               const float *__restrict__ dOut, int64_t ldo,
                                                               const float *__restrict__ h, int64_t ldh, int32_t n, int32_t C,
                                                               float *__restrict__ dalpha, const int64_t *__restrict__ sel) {
    const int wi = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wi >= n) return;
    const int row = sel ? (int)sel[wi] : wi;
    const int e0 = rowptr[row], e1 = rowptr[row + 1];
    for (int e = e0; e < e1; ++e) {
        const float *hr = h + (int64_t)col[e] * ldh;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += dOut[(int64_t)wi * ldo + c] * hr[c];
        s = wave_sum(s);
        if (lane == 0) dalpha[e] = s;
    }
}

// softmax + LeakyReLU backward per row:  ds_e = alpha_e (dalpha_e - sum_k alpha_k dalpha_k) * lrelu'(s_e);
// da_dst[row] = sum_e ds_e.  ds is left per edge for the column-side sum (done on the transposed order).  Threads / waves as in
// the forward kernel.  sel (may be NULL): the launch covers the rows sel[0 .. n) only (the caller zeroes ds / da_dst).
__global__ __launch_bounds__(256) void gat_softmax_bwd_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                              const float *__restrict__ a_src, const float *__restrict__ a_dst,
                                                              const float *__restrict__ alpha, const float *__restrict__ dalpha,
                                                              float slope, int32_t n, float *__restrict__ ds,
                                                              float *__restrict__ da_dst, const int64_t *__restrict__ sel) {
    const int wi = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool on = wi < n;
    int row = 0, e0 = 0, len = 0;
    float ad = 0.f;
    if (on) {
        row = sel ? (int)sel[wi] : wi;
        e0 = rowptr[row];
        len = rowptr[row + 1] - e0;
        ad = a_dst[row];
    }
    if (on && len <= kShortRow) {
        float al[kShortRow], da[kShortRow];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < kShortRow; ++k) {
            al[k] = k < len ? alpha[e0 + k] : 0.f;
            da[k] = k < len ? dalpha[e0 + k] : 0.f;
            dot += al[k] * da[k];
        }
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < kShortRow; ++k)
            if (k < len) {
                const float sv = a_src[col[e0 + k]] + ad;
                const float d = al[k] * (da[k] - dot) * (sv > 0.f ? 1.0f : slope);
                ds[e0 + k] = d;
                acc += d;
            }
        da_dst[row] = acc;
    }
    unsigned long long longs = __ballot(on && len > kShortRow);
    while (longs) {   // wave-uniform
        const int src = __ffsll((long long)longs) - 1;
        longs &= longs - 1;
        const int b0 = __shfl(e0, src, 64), b1 = b0 + __shfl(len, src, 64);
        const float adr = __shfl(ad, src, 64);
        const int r_l = __shfl(row, src, 64);
        float dot = 0.f;
        for (int e = b0 + lane; e < b1; e += 64) dot += alpha[e] * dalpha[e];
        dot = wave_sum(dot);
        float acc = 0.f;
        for (int e = b0 + lane; e < b1; e += 64) {
            const float sv = a_src[col[e]] + adr;
            const float d = alpha[e] * (dalpha[e] - dot) * (sv > 0.f ? 1.0f : slope);
            ds[e] = d;
            acc += d;
        }
        acc = wave_sum(acc);
        if (lane == 0) da_dst[r_l] = acc;
    }
}

// y[row] = sum of v over the CSR row (used for da_src on the transposed order).  Eight lanes per row, eight rows per wavefront: a lane
// adds entries start + sub, start + sub + 8, ... in order (two in flight), the eight partial sums fold by xor shuffles in a fixed
// tree.  (One WAVEFRONT per row -- 8.2 M wavefronts of 3.3 entries and a six-stage reduction each at S-products -- took 1.43 ms for
// 108 MB of values.)
__global__ __launch_bounds__(256) void csr_row_sum_kernel(const int32_t *__restrict__ rowptr, const float *__restrict__ v, int32_t n,
                                                          float *__restrict__ y) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = t >> 3;
    const int sub = threadIdx.x & 7;
    float acc = 0.f;
    if (row < n) {
        const int e1 = rowptr[row + 1];
        int e = rowptr[row] + sub;
        for (; e + 8 < e1; e += 16) acc = (acc + v[e]) + v[e + 8];
        if (e < e1) acc += v[e];
    }
    acc += __shfl_xor(acc, 4, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 1, 64);
    if (row < n && sub == 0) y[row] = acc;
}

inline dim3 wave_grid(int32_t n) { return dim3((unsigned)(((int64_t)n * 64 + 255) / 256)); }
inline dim3 thread_grid(int32_t n) { return dim3((unsigned)(((int64_t)n + 255) / 256)); }

}  // namespace

extern "C" int fitgnn_gat_scores_f32(const float *h, int64_t ldh, int32_t n, int32_t C, const float *att_src,
                                     const float *att_dst, float *a_src, float *a_dst, void *stream) {
    if (n < 0 || C < 0 || ldh < C) return FITGNN_E_BADARG;
    if (n == 0) return 0;
    if (!h || !att_src || !att_dst || !a_src || !a_dst) return FITGNN_E_BADARG;
    hipLaunchKernelGGL(gat_scores_kernel, wave_grid(n), dim3(256), 0, (hipStream_t)stream, h, ldh, n, C, att_src, att_dst, a_src,
                       a_dst);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_gat_edge_softmax_f32(const int32_t *rowptr, const int32_t *col, const float *a_src, const float *a_dst,
                                           float negative_slope, int32_t n, float *alpha, void *stream) {
    if (n < 0) return FITGNN_E_BADARG;
    if (n == 0) return 0;
    if (!rowptr || !a_src || !a_dst) return FITGNN_E_BADARG;
    hipLaunchKernelGGL(gat_edge_softmax_kernel, thread_grid(n), dim3(256), 0, (hipStream_t)stream, rowptr, col, a_src, a_dst,
                       negative_slope, n, alpha);
    return (int)hipGetLastError();
}

static int sddmm_launch(const int32_t *rowptr, const int32_t *col, const float *dOut, int64_t ldo, const float *h, int64_t ldh, int32_t n,
                        int32_t C, float *dalpha, const int64_t *sel, hipStream_t s) {
    const bool vec = (C % 4 == 0) && (ldo % 4 == 0) && (ldh % 4 == 0) && ((((uintptr_t)dOut | (uintptr_t)h) % 16) == 0);
    if (vec && C <= 256)
        hipLaunchKernelGGL(gat_sddmm_kernel<1>, wave_grid(n), dim3(256), 0, s, rowptr, col, dOut, ldo, h, ldh, n, C, dalpha, sel);
    else if (vec && C <= 512)
        hipLaunchKernelGGL(gat_sddmm_kernel<2>, wave_grid(n), dim3(256), 0, s, rowptr, col, dOut, ldo, h, ldh, n, C, dalpha, sel);
    else if (vec && C <= 1024)
        hipLaunchKernelGGL(gat_sddmm_kernel<4>, wave_grid(n), dim3(256), 0, s, rowptr, col, dOut, ldo, h, ldh, n, C, dalpha, sel);
    else
        hipLaunchKernelGGL(gat_sddmm_scalar_kernel, wave_grid(n), dim3(256), 0, s, rowptr, col, dOut, ldo, h, ldh, n, C, dalpha, sel);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_sddmm_csr_f32(const int32_t *rowptr, const int32_t *col, const float *dOut, int64_t ldo, const float *h,
                                    int64_t ldh, int32_t n, int32_t C, float *dalpha, void *stream) {
    if (n < 0 || C < 0 || ldo < C || ldh < C) return FITGNN_E_BADARG;
    if (n == 0) return 0;
    if (!rowptr || !dOut || !h) return FITGNN_E_BADARG;
    return sddmm_launch(rowptr, col, dOut, ldo, h, ldh, n, C, dalpha, nullptr, (hipStream_t)stream);
}

extern "C" int fitgnn_sddmm_csr_rows_f32(const int32_t *rowptr, const int32_t *col, const float *dOut_c, int64_t ldo, const float *h,
                                         int64_t ldh, const int64_t *sel, int32_t n_sel, int32_t C, float *dalpha, void *stream) {
    if (n_sel < 0 || C < 0 || ldo < C || ldh < C) return FITGNN_E_BADARG;
    if (n_sel == 0) return 0;
    if (!rowptr || !dOut_c || !h || !sel || !dalpha) return FITGNN_E_BADARG;
    return sddmm_launch(rowptr, col, dOut_c, ldo, h, ldh, n_sel, C, dalpha, sel, (hipStream_t)stream);
}

extern "C" int fitgnn_gat_softmax_bwd_f32(const int32_t *rowptr, const int32_t *col, const float *a_src, const float *a_dst,
                                          const float *alpha, const float *dalpha, float negative_slope, int32_t n, float *ds,
                                          float *da_dst, void *stream) {
    if (n < 0) return FITGNN_E_BADARG;
    if (n == 0) return 0;
    if (!rowptr || !a_src || !a_dst || !da_dst) return FITGNN_E_BADARG;
    hipLaunchKernelGGL(gat_softmax_bwd_kernel, thread_grid(n), dim3(256), 0, (hipStream_t)stream, rowptr, col, a_src, a_dst, alpha,
                       dalpha, negative_slope, n, ds, da_dst, (const int64_t *)nullptr);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_gat_softmax_bwd_rows_f32(const int32_t *rowptr, const int32_t *col, const float *a_src, const float *a_dst,
                                               const float *alpha, const float *dalpha, float negative_slope, const int64_t *sel,
                                               int32_t n_sel, float *ds, float *da_dst, void *stream) {
    if (n_sel < 0) return FITGNN_E_BADARG;
    if (n_sel == 0) return 0;
    if (!rowptr || !a_src || !a_dst || !da_dst || !sel) return FITGNN_E_BADARG;
    hipLaunchKernelGGL(gat_softmax_bwd_kernel, thread_grid(n_sel), dim3(256), 0, (hipStream_t)stream, rowptr, col, a_src, a_dst, alpha,
                       dalpha, negative_slope, n_sel, ds, da_dst, sel);
    return (int)hipGetLastError();
}

// Narrow SpMM for propagation on class-wide signals (APPNP: K = 10 steps on [rows x num_classes]):
//   Y[r, c] = beta * sum_e val[e] X[col[e], c] + gamma * Z0[r, c];   optionally ACC[r, c] += delta * X[r, c].
// One thread per (row, column): a row's H threads read the same (col, val) stream (served by the cache) and
// neighbouring columns of X.  The tiled kernel gives H < 64 columns a 64-lane slab each, i.e. 3 live lanes for H = 3.
__global__ __launch_bounds__(256) void spmm_narrow_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                          const float *__restrict__ val, const float *__restrict__ X,
                                                          float *__restrict__ Y, int32_t n, int32_t H, float beta,
                                                          const float *__restrict__ Z0, float gamma, float *__restrict__ ACC,
                                                          float delta) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)n * H) return;
    const int r = (int)(t / H), c = (int)(t - (int64_t)r * H);
    float acc = 0.f;
    const int e1 = rowptr[r + 1];
    for (int e = rowptr[r]; e < e1; ++e) acc = fmaf(val[e], X[(int64_t)col[e] * H + c], acc);
    float y = beta * acc;
    if (Z0) y = fmaf(gamma, Z0[t], y);
    Y[t] = y;
    if (ACC) ACC[t] = fmaf(delta, X[t], ACC[t]);
}

extern "C" int fitgnn_spmm_narrow_f32(const int32_t *rowptr, const int32_t *col, const float *val, const float *X, float *Y,
                                      int32_t n_rows, int32_t H, float beta, const float *Z0, float gamma, float *ACC,
                                      float delta, void *stream) {
    if (n_rows < 0 || H < 0) return FITGNN_E_BADARG;
    if (n_rows == 0 || H == 0) return 0;
    if (!rowptr || !X || !Y) return FITGNN_E_BADARG;
    const int64_t threads = (int64_t)n_rows * H;
    hipLaunchKernelGGL(spmm_narrow_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rowptr, col, val,
                       X, Y, n_rows, H, beta, Z0, gamma, ACC, delta);
    return (int)hipGetLastError();
}

// The same product for a signal whose rows are PADDED to whole float4s (ld = 4 * h4 floats, pad columns zero): the layout APPNP's K
// steps run in.  A wave packs G = 64 / h4 consecutive rows: lane = (row slot, float4 column), so an operand row is one contiguous
// 16 * h4-byte access of h4 lanes and a wave instruction gathers G rows; the first kLongFrom entries of every row are taken in CSR
// order by the row's own lanes (a leaf of a star has 2-4), the rest of a long row (a centre) is split over all G slots and folded
// across them -- fixed order: the result does not depend on the launch geometry.  No LDS; consecutive rows of a subgraph sit in one
// wave, so the gathers of a block-diagonal batch hit L1 / L2.
constexpr int kLongFrom = 8;

__global__ __launch_bounds__(256) void spmm_narrow_packed_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                                 const float *__restrict__ val, const float4 *__restrict__ X,
                                                                 float4 *__restrict__ Y, int32_t n, int32_t h4, float beta,
                                                                 const float4 *__restrict__ Z0, float gamma, float4 *__restrict__ ACC,
                                                                 float delta, int32_t groups_per_wave) {
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6);
    const int G = 64 / h4;
    const int slot = lane / h4, q = lane - slot * h4;
    const bool lane_on = slot < G;
    const long n_groups = ((long)n + G - 1) / G;
    const long g_begin = (long)wave * groups_per_wave;
    long g_end = g_begin + groups_per_wave;
    if (g_end > n_groups) g_end = n_groups;
    for (long g = g_begin; g < g_end; ++g) {
        const long r = g * G + slot;
        const bool on = lane_on && r < n;
        int e = 0, len = 0;
        if (on) {
            e = rowptr[r];
            len = rowptr[r + 1] - e;
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int head = len < kLongFrom ? len : kLongFrom;
        int most = head;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) most = max(most, __shfl_xor(most, off, 64));
        for (int k = 0; k < most; ++k) {
            if (k < head) {
                const int c = col[e + k];
                const float v = val[e + k];
                const float4 x = X[(long)c * h4 + q];
                acc.x = fmaf(v, x.x, acc.x); acc.y = fmaf(v, x.y, acc.y); acc.z = fmaf(v, x.z, acc.z); acc.w = fmaf(v, x.w, acc.w);
            }
        }
        // long rows of the group, one after the other (wave-uniform): every slot takes every G-th remaining entry
        unsigned long long longs = __ballot(on && q == 0 && len > kLongFrom);
        while (longs) {
            const int src = __ffsll((long long)longs) - 1;   // lane (slot s0, q = 0)
            longs &= longs - 1;
            const int e0 = __shfl(e, src, 64) + kLongFrom;
            const int e1 = __shfl(e, src, 64) + __shfl(len, src, 64);
            float4 part = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane_on) {
                for (int k = e0 + slot; k < e1; k += G) {
                    const int c = col[k];
                    const float v = val[k];
                    const float4 x = X[(long)c * h4 + q];
                    part.x = fmaf(v, x.x, part.x); part.y = fmaf(v, x.y, part.y); part.z = fmaf(v, x.z, part.z); part.w = fmaf(v, x.w, part.w);
                }
            }
            float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int s2 = 0; s2 < G; ++s2) {   // fold the slots in slot order
                const int from = s2 * h4 + q;
                tot.x += __shfl(part.x, from, 64); tot.y += __shfl(part.y, from, 64);
                tot.z += __shfl(part.z, from, 64); tot.w += __shfl(part.w, from, 64);
            }
            if (lane_on && slot == src / h4) { acc.x += tot.x; acc.y += tot.y; acc.z += tot.z; acc.w += tot.w; }
        }
        if (on) {
            const long o = r * h4 + q;
            float4 y = make_float4(beta * acc.x, beta * acc.y, beta * acc.z, beta * acc.w);
            if (Z0) {
                const float4 z = Z0[o];
                y.x = fmaf(gamma, z.x, y.x); y.y = fmaf(gamma, z.y, y.y); y.z = fmaf(gamma, z.z, y.z); y.w = fmaf(gamma, z.w, y.w);
            }
            Y[o] = y;
            if (ACC) {
                const float4 x = X[o];
                float4 a = ACC[o];
                a.x = fmaf(delta, x.x, a.x); a.y = fmaf(delta, x.y, a.y); a.z = fmaf(delta, x.z, a.z); a.w = fmaf(delta, x.w, a.w);
                ACC[o] = a;
            }
        }
    }
}

extern "C" int fitgnn_spmm_narrow_padded_f32(const int32_t *rowptr, const int32_t *col, const float *val, const float *X, float *Y,
                                             int32_t n_rows, int32_t h4, float beta, const float *Z0, float gamma, float *ACC,
                                             float delta, void *stream) {
    if (n_rows < 0 || h4 < 1 || h4 > 16) return FITGNN_E_BADARG;
    if (n_rows == 0) return 0;
    if (!rowptr || !col || !val || !X || !Y) return FITGNN_E_BADARG;
    if ((((uintptr_t)X | (uintptr_t)Y | (uintptr_t)Z0 | (uintptr_t)ACC) % 16) != 0) return FITGNN_E_ALIGN;
    const int G = 64 / h4;
    const long n_groups = ((long)n_rows + G - 1) / G;
    // ~16 k waves (8 per SIMD and a few rounds), each a contiguous run of groups
    long waves = n_groups < 16384 ? n_groups : 16384;
    const int per = (int)((n_groups + waves - 1) / waves);
    waves = (n_groups + per - 1) / per;
    hipLaunchKernelGGL(spmm_narrow_packed_kernel, dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rowptr,
                       col, val, (const float4 *)X, (float4 *)Y, n_rows, h4, beta, (const float4 *)Z0, gamma, (float4 *)ACC, delta, per);
    return (int)hipGetLastError();
}

// The padded layout the propagation runs in, made in ONE pass from the model's class-wide output: dst[r] = src[index ? index[r] : r]
// (H floats, row stride lds) followed by zeros up to 4 h4 floats.  Replaces index_select (the de-duplicated table's rows -> union
// rows) + zero fill + strided copy: three passes over the [R x 47] signal, 1.9 ms of a 20-ms step at S-products.
__global__ __launch_bounds__(256) void gather_rows_padded_kernel(const float *__restrict__ src, int64_t lds, int32_t H,
                                                                 const int32_t *__restrict__ index, int64_t n, float4 *__restrict__ dst,
                                                                 int32_t h4) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * h4) return;
    const int64_t r = t / h4;
    const int c = 4 * (int)(t - r * h4);
    const float *row = src + (index ? (int64_t)index[r] : r) * lds + c;
    float4 v;
    v.x = c < H ? row[0] : 0.f;
    v.y = c + 1 < H ? row[1] : 0.f;
    v.z = c + 2 < H ? row[2] : 0.f;
    v.w = c + 3 < H ? row[3] : 0.f;
    dst[t] = v;
}

extern "C" int fitgnn_gather_rows_padded_f32(const float *src, int64_t lds, int32_t H, const int32_t *index, int64_t n_rows, float *dst,
                                             int32_t h4, void *stream) {
    if (n_rows < 0 || H < 1 || h4 < 1 || 4 * h4 < H || lds < H) return FITGNN_E_BADARG;
    if (n_rows == 0) return 0;
    if (!src || !dst) return FITGNN_E_BADARG;
    if (((uintptr_t)dst % 16) != 0) return FITGNN_E_ALIGN;
    const int64_t total = n_rows * h4;
    hipLaunchKernelGGL(gather_rows_padded_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, lds, H, index,
                       n_rows, (float4 *)dst, h4);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// APPNP's K propagation steps with the signal resident in LDS (fitgnn_appnp_units_f32).
//
// The batches this library serves are block diagonal: a cluster subgraph shares no edge with another (utils.py:248), so
// z_{k+1} = (1 - alpha) A_hat z_k + alpha z_0 never mixes two subgraphs -- and a subgraph of a few dozen rows x 47 classes is a few
// kilobytes.  One launch per propagation step (spmm_narrow_packed_kernel) reads and writes the whole signal K times: 20 passes per
// training step for K = 10.  Here a "unit" -- a run of consecutive rows closed under the pattern (whole subgraphs; at most 768 / h4
// rows and 2 048 CSR entries; the caller lists them) -- is loaded ONCE by one wavefront, stepped K times between two LDS buffers (its CSR
// slice staged beside them, columns re-based to the unit), and stored once.  A lane owns the items (row, float4 column) lane,
// lane + 64, ...: the 12 column groups of a long row (a star's centre) sit on 12 lanes, so a step costs the longest row once, not 12
// times.  Forward: the teleport operand z_0 stays in the lane's registers.  Backward (BWD; the pattern handed in is the transposed
// one): g_{k+1} = (1 - alpha) A^T g_k with acc += alpha g_k in registers, result acc + g_K -- APPNPPropagate.backward's recurrence.
// Rows outside the units (subgraphs larger than a unit) are the caller's: it runs the per-step kernel on their sub-matrix.
constexpr int kAppnpItems = 12;      // float4 items per lane: a unit holds at most 64 * 12 = 768 (row, float4 column) items,
constexpr int kAppnpRowsMax = 768;   // i.e. 768 / h4 rows (768 rows of a 3-class signal, 64 of a 47-class one)
constexpr int kAppnpEntries = 2048;  // CSR entries of a unit staged in LDS (44 KiB per wavefront in all: three units per CU)
__host__ __device__ inline int appnp_rows(int h4) { return (64 * kAppnpItems) / h4 < kAppnpRowsMax ? (64 * kAppnpItems) / h4 : kAppnpRowsMax; }

template <bool BWD>
__global__ __launch_bounds__(64) void appnp_units_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                         const float *__restrict__ val, const int32_t *__restrict__ units, int32_t n_units,
                                                         const float4 *__restrict__ X, float4 *__restrict__ Y, int32_t h4, int32_t K,
                                                         float alpha, int32_t cap_rows, int32_t cap_entries) {
    // LDS: two signal buffers of cap_rows x h4 float4, row pointers, the CSR slice -- sized by the launch for ITS largest unit
    extern __shared__ __attribute__((aligned(16))) unsigned char au_lds[];
    float4 *buf0 = reinterpret_cast<float4 *>(au_lds);
    float4 *buf1 = buf0 + cap_rows * h4;
    int32_t *s_rp = reinterpret_cast<int32_t *>(buf1 + cap_rows * h4);
    int32_t *s_col = s_rp + cap_rows + 4;
    float *s_val = reinterpret_cast<float *>(s_col + cap_entries);
    const int u = blockIdx.x;
    if (u >= n_units) return;
    const int lane = threadIdx.x;
    const int r0 = units[2 * u], r1 = units[2 * u + 1];
    const int n = r1 - r0;
    const int E0 = rowptr[r0];
    const int nE = rowptr[r1] - E0;
    for (int i = lane; i <= n; i += 64) s_rp[i] = rowptr[r0 + i] - E0;
    for (int e = lane; e < nE; e += 64) {
        s_col[e] = col[E0 + e] - r0;
        s_val[e] = val[E0 + e];
    }
    const int total = n * h4;
    const int64_t base = (int64_t)r0 * h4;   // the unit's rows are consecutive: item i is X[base + i]
    float4 keep[kAppnpItems];
#pragma unroll
    for (int j = 0; j < kAppnpItems; ++j) {
        const int i = lane + 64 * j;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < total) {
            x = X[base + i];
            buf0[i] = x;
        }
        keep[j] = BWD ? make_float4(0.f, 0.f, 0.f, 0.f) : x;
    }
    __syncthreads();
    const float beta = 1.0f - alpha;
    for (int k = 0; k < K; ++k) {
        const float4 *cur = (k & 1) ? buf1 : buf0;
        float4 *nxt = (k & 1) ? buf0 : buf1;
#pragma unroll
        for (int j = 0; j < kAppnpItems; ++j) {
            const int i = lane + 64 * j;
            if (i >= total) break;
            const int row = i / h4, q = i - row * h4;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const int e1 = s_rp[row + 1];
            // four entries at a time: their (column, value) reads, then their four operand reads, are issued together -- one at a time an
            // entry is two dependent LDS round trips, and a star's centre (50-300 entries) made the whole unit wait for its chain (eight
            // at a time pads the leaves' 3-entry rows to 8 reads: slower at S-products); a last group is padded with its last entry at
            // weight 0; the additions stay in CSR order
            for (int e = s_rp[row]; e < e1; e += 4) {
                int c[4];
                float v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int et = min(e + t, e1 - 1);
                    c[t] = s_col[et];
                    v[t] = e + t < e1 ? s_val[et] : 0.f;
                }
                float4 x[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) x[t] = cur[c[t] * h4 + q];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc.x = fmaf(v[t], x[t].x, acc.x); acc.y = fmaf(v[t], x[t].y, acc.y);
                    acc.z = fmaf(v[t], x[t].z, acc.z); acc.w = fmaf(v[t], x[t].w, acc.w);
                }
            }
            float4 y = make_float4(beta * acc.x, beta * acc.y, beta * acc.z, beta * acc.w);
            if (BWD) {
                const float4 x = cur[i];
                keep[j].x = fmaf(alpha, x.x, keep[j].x); keep[j].y = fmaf(alpha, x.y, keep[j].y);
                keep[j].z = fmaf(alpha, x.z, keep[j].z); keep[j].w = fmaf(alpha, x.w, keep[j].w);
            } else {
                y.x = fmaf(alpha, keep[j].x, y.x); y.y = fmaf(alpha, keep[j].y, y.y);
                y.z = fmaf(alpha, keep[j].z, y.z); y.w = fmaf(alpha, keep[j].w, y.w);
            }
            nxt[i] = y;
        }
        __syncthreads();
    }
    const float4 *fin = (K & 1) ? buf1 : buf0;
#pragma unroll
    for (int j = 0; j < kAppnpItems; ++j) {
        const int i = lane + 64 * j;
        if (i >= total) break;
        float4 y = fin[i];
        if (BWD) { y.x += keep[j].x; y.y += keep[j].y; y.z += keep[j].z; y.w += keep[j].w; }
        Y[base + i] = y;
    }
}

extern "C" int fitgnn_appnp_unit_rows(int32_t h4) { return (h4 >= 1 && h4 <= 16) ? appnp_rows(h4) : 0; }
extern "C" int fitgnn_appnp_unit_entries(void) { return kAppnpEntries; }

extern "C" int fitgnn_appnp_units_f32(const int32_t *rowptr, const int32_t *col, const float *val, const int32_t *units, int32_t n_units,
                                      int32_t max_rows, int32_t max_entries, const float *X, float *Y, int32_t h4, int32_t K, float alpha,
                                      int32_t backward, void *stream) {
    if (n_units < 0 || h4 < 1 || h4 > 16 || K < 0) return FITGNN_E_BADARG;
    if (n_units > 0 && (max_rows < 1 || max_rows > appnp_rows(h4) || max_entries < 1 || max_entries > kAppnpEntries)) return FITGNN_E_BADARG;
    if (n_units == 0) return 0;
    if (!rowptr || !col || !val || !units || !X || !Y) return FITGNN_E_BADARG;
    if ((((uintptr_t)X | (uintptr_t)Y) % 16) != 0) return FITGNN_E_ALIGN;
    const int cap_rows = (max_rows + 3) / 4 * 4, cap_entries = (max_entries + 3) / 4 * 4;
    const size_t lds = (size_t)2 * cap_rows * h4 * sizeof(float4) + (size_t)(cap_rows + 4) * sizeof(int32_t) +
                       (size_t)cap_entries * (sizeof(int32_t) + sizeof(float));   // <= 24 KiB + 3 KiB + 16 KiB
    if (backward)
        hipLaunchKernelGGL(appnp_units_kernel<true>, dim3((unsigned)n_units), dim3(64), lds, (hipStream_t)stream, rowptr, col, val, units,
                           n_units, (const float4 *)X, (float4 *)Y, h4, K, alpha, cap_rows, cap_entries);
    else
        hipLaunchKernelGGL(appnp_units_kernel<false>, dim3((unsigned)n_units), dim3(64), lds, (hipStream_t)stream, rowptr, col, val, units,
                           n_units, (const float4 *)X, (float4 *)Y, h4, K, alpha, cap_rows, cap_entries);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The subgraphs too large for a unit (fitgnn_appnp_blocks_f32): one WORKGROUP per closed diagonal block, all K steps in one launch.
//
// A block of ~1 600 rows x 47 classes is 300 KB per buffer: not LDS, but a few hundred KB that one workgroup reads and writes K times
// stay in its XCD's L2 -- so the steps ping-pong between two global scratch signals T1 / T2 (only the block's own rows are touched),
// separated by workgroup barriers instead of launches (the block is closed: no other workgroup reads or writes its rows).  The
// block's CSR slice (row pointers, columns, values: read K times) is staged in LDS once, which leaves one global round trip per
// group of rows (the operand gather) where the per-step kernel has three (row pointer -> column -> operand).  The arithmetic of a
// row is spmm_narrow_packed_kernel's, entry for entry (a wave packs G = 64 / h4 rows; the first kLongFrom entries by the row's own
// lanes, the rest of a long row split over the G slots and folded in slot order): the two paths give the same bits.
// Forward: z_{k+1} = (1 - alpha) A z_k + alpha X, result in Y.  Backward (transposed pattern handed in): g_{k+1} = (1 - alpha) A^T g_k,
// Y accumulates alpha g_k (read-modify-write by the item's own lane) and ends as that sum + g_K.
constexpr int kAppnpBlockThreads = 1024;
constexpr int kAppnpBlockRows = 4096;      // rows of a block: its row pointers in LDS (16 KiB)
constexpr int kAppnpBlockEntries = 16384;  // CSR entries of a block in LDS (128 KiB); a launch sizes its LDS by ITS largest block

template <bool BWD>
__global__ __launch_bounds__(kAppnpBlockThreads) void appnp_blocks_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                                          const float *__restrict__ val, const int32_t *__restrict__ blocks,
                                                                          int32_t n_blocks, const float4 *X, float4 *Y, float4 *T1, float4 *T2,
                                                                          int32_t h4, int32_t K, float alpha, int32_t cap_rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ab_lds[];
    int32_t *s_rp = reinterpret_cast<int32_t *>(ab_lds);
    const int b = blockIdx.x;
    if (b >= n_blocks) return;
    const int r0 = blocks[2 * b], r1 = blocks[2 * b + 1];
    const int n = r1 - r0;
    const int E0 = rowptr[r0];
    const int nE = rowptr[r1] - E0;
    int32_t *s_col = s_rp + cap_rows + 4;
    float *s_val = reinterpret_cast<float *>(s_col + ((nE + 3) & ~3));
    const int n_threads = blockDim.x;
    for (int i = threadIdx.x; i <= n; i += n_threads) s_rp[i] = rowptr[r0 + i] - E0;
    for (int e = threadIdx.x; e < nE; e += n_threads) {
        s_col[e] = col[E0 + e];
        s_val[e] = val[E0 + e];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_waves = n_threads >> 6;
    const int G = 64 / h4;
    const int slot = lane / h4, q = lane - slot * h4;
    const bool lane_on = slot < G;
    const int n_groups = (n + G - 1) / G;
    const float beta = 1.0f - alpha;
    for (int k = 0; k < K; ++k) {
        const float4 *cur = k == 0 ? X : ((k & 1) ? T1 : T2);
        float4 *nxt = (k & 1) ? T2 : T1;
        const bool last = k == K - 1;
        for (int g = wave; g < n_groups; g += n_waves) {
            const int lr = g * G + slot;   // row within the block
            const bool on = lane_on && lr < n;
            int e = 0, len = 0;
            if (on) {
                e = s_rp[lr];
                len = s_rp[lr + 1] - e;
            }
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const int head = len < kLongFrom ? len : kLongFrom;
            int most = head;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) most = max(most, __shfl_xor(most, off, 64));
            // the head entries: (column, value) of all of them first (LDS), then the operand rows (L2), then the sums in CSR order
            {
                int c[kLongFrom];
                float v[kLongFrom];
#pragma unroll
                for (int t = 0; t < kLongFrom; ++t) {
                    const bool has = t < head;
                    c[t] = has ? s_col[e + t] : 0;
                    v[t] = has ? s_val[e + t] : 0.f;
                }
                float4 x[kLongFrom];
#pragma unroll
                for (int t = 0; t < kLongFrom; ++t)
                    if (t < most) x[t] = t < head ? cur[(int64_t)c[t] * h4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t < kLongFrom; ++t)
                    if (t < head) {
                        acc.x = fmaf(v[t], x[t].x, acc.x); acc.y = fmaf(v[t], x[t].y, acc.y);
                        acc.z = fmaf(v[t], x[t].z, acc.z); acc.w = fmaf(v[t], x[t].w, acc.w);
                    }
            }
            unsigned long long longs = __ballot(on && q == 0 && len > kLongFrom);
            while (longs) {
                const int src = __ffsll((long long)longs) - 1;
                longs &= longs - 1;
                const int e0 = __shfl(e, src, 64) + kLongFrom;
                const int e1 = __shfl(e, src, 64) + __shfl(len, src, 64);
                float4 part = make_float4(0.f, 0.f, 0.f, 0.f);
                if (lane_on) {
                    for (int kk = e0 + slot; kk < e1; kk += G) {
                        const int c = s_col[kk];
                        const float v = s_val[kk];
                        const float4 x = cur[(int64_t)c * h4 + q];
                        part.x = fmaf(v, x.x, part.x); part.y = fmaf(v, x.y, part.y); part.z = fmaf(v, x.z, part.z); part.w = fmaf(v, x.w, part.w);
                    }
                }
                float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int s2 = 0; s2 < G; ++s2) {
                    const int from = s2 * h4 + q;
                    tot.x += __shfl(part.x, from, 64); tot.y += __shfl(part.y, from, 64);
                    tot.z += __shfl(part.z, from, 64); tot.w += __shfl(part.w, from, 64);
                }
                if (lane_on && slot == src / h4) { acc.x += tot.x; acc.y += tot.y; acc.z += tot.z; acc.w += tot.w; }
            }
            if (on) {
                const int64_t o = (int64_t)(r0 + lr) * h4 + q;
                float4 y = make_float4(beta * acc.x, beta * acc.y, beta * acc.z, beta * acc.w);
                if (!BWD) {
                    const float4 z = X[o];
                    y.x = fmaf(alpha, z.x, y.x); y.y = fmaf(alpha, z.y, y.y); y.z = fmaf(alpha, z.z, y.z); y.w = fmaf(alpha, z.w, y.w);
                    (last ? Y : nxt)[o] = y;
                } else {
                    const float4 x = cur[o];
                    float4 a = k == 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : Y[o];
                    a.x = fmaf(alpha, x.x, a.x); a.y = fmaf(alpha, x.y, a.y); a.z = fmaf(alpha, x.z, a.z); a.w = fmaf(alpha, x.w, a.w);
                    if (last) {
                        a.x += y.x; a.y += y.y; a.z += y.z; a.w += y.w;
                    } else {
                        nxt[o] = y;
                    }
                    Y[o] = a;
                }
            }
        }
        __syncthreads();   // the step's rows (global, this workgroup's own) before the next step gathers them
    }
}

extern "C" int fitgnn_appnp_block_rows(void) { return kAppnpBlockRows; }
extern "C" int fitgnn_appnp_block_entries(void) { return kAppnpBlockEntries; }

extern "C" int fitgnn_appnp_blocks_f32(const int32_t *rowptr, const int32_t *col, const float *val, const int32_t *blocks, int32_t n_blocks,
                                       int32_t max_rows, int32_t max_entries, const float *X, float *Y, float *T1, float *T2, int32_t h4,
                                       int32_t K, float alpha, int32_t backward, void *stream) {
    if (n_blocks < 0 || h4 < 1 || h4 > 16 || K < 1) return FITGNN_E_BADARG;
    if (n_blocks > 0 && (max_rows < 1 || max_rows > kAppnpBlockRows || max_entries < 0 || max_entries > kAppnpBlockEntries)) return FITGNN_E_BADARG;
    if (n_blocks == 0) return 0;
    if (!rowptr || !col || !val || !blocks || !X || !Y || !T1 || !T2) return FITGNN_E_BADARG;
    if (X == Y || T1 == T2 || X == T1 || X == T2 || Y == T1 || Y == T2) return FITGNN_E_BADARG;
    if ((((uintptr_t)X | (uintptr_t)Y | (uintptr_t)T1 | (uintptr_t)T2) % 16) != 0) return FITGNN_E_ALIGN;
    const int cap_rows = (max_rows + 3) / 4 * 4, cap_entries = (max_entries + 3) / 4 * 4 + 4;
    const size_t lds = (size_t)(cap_rows + 4) * sizeof(int32_t) + (size_t)cap_entries * (sizeof(int32_t) + sizeof(float));
    const void *fn = backward ? (const void *)appnp_blocks_kernel<true> : (const void *)appnp_blocks_kernel<false>;
    int threads = kAppnpBlockThreads;
    if (const char *ev = getenv("FITGNN_APPNP_BLOCK_THREADS")) {   // tuning knob: 64 .. 1024, whole wavefronts
        const int t = atoi(ev);
        if (t >= 64 && t <= kAppnpBlockThreads && t % 64 == 0) threads = t;
    }
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    if (backward)
        hipLaunchKernelGGL(appnp_blocks_kernel<true>, dim3((unsigned)n_blocks), dim3(threads), lds, (hipStream_t)stream, rowptr, col,
                           val, blocks, n_blocks, (const float4 *)X, (float4 *)Y, (float4 *)T1, (float4 *)T2, h4, K, alpha, cap_rows);
    else
        hipLaunchKernelGGL(appnp_blocks_kernel<false>, dim3((unsigned)n_blocks), dim3(threads), lds, (hipStream_t)stream, rowptr, col,
                           val, blocks, n_blocks, (const float4 *)X, (float4 *)Y, (float4 *)T1, (float4 *)T2, h4, K, alpha, cap_rows);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The K steps in LDS for subgraphs of any size that fits, by COLUMN SLICES (fitgnn_appnp_lds_f32).
//
// The recurrence never mixes two columns of the signal, so a workgroup need not hold a subgraph's whole [rows x 4 h4] signal: it
// stages the CSR slice of its range once (columns re-based to the range as 16-bit, values, row pointers) and then, for one slice of
// w <= `slice` float4 columns after the other, loads the slice of its rows, steps it K times between two LDS buffers and stores it.
// A 1 600-row subgraph of a 47-class signal is 300 KB whole and 25 KB per buffer as a one-float4 slice; a 50-row star whose
// wavefront held 24 KB of buffers (one wavefront per SIMD) holds 6 KB at w = 4.  Work of a step: a thread owns the items (row,
// slice column) tid, tid + T, ... (at most kLdsKeep: the teleport operand z_0, or the backward's running alpha-sum, stays in its
// registers) and sums a SHORT row's entries in CSR order, four at a time; a row of more than kLdsShort entries (a star's centre) is
// taken by a whole wavefront -- lane = (entry slot, slice column), 64 / w entries per round, slots folded by xor shuffles in a fixed
// tree -- so a step costs a centre a few LDS round trips instead of a chain as long as its row.  Long rows are listed at staging
// (LDS counter: the list's order varies, no result depends on it); their z_0 / alpha-sum lives in LDS beside the buffers.
constexpr int kLdsKeep = 4;     // items per thread: rows of a range x w <= kLdsKeep x threads
constexpr int kLdsFly = 2;      // items of a thread whose operand reads are in flight together (2: the step's results, held in
                                // registers until the barrier, spill at 1 024 threads x 128 registers)
constexpr int kLdsShort = 16;   // rows of more entries than this are summed by a wavefront
constexpr int kLdsMaxBytes = 160 * 1024;

struct LdsPlan { int cap_rows, cap_entries, cap_long; size_t bytes; };
__host__ __device__ inline LdsPlan appnp_lds_plan(int max_rows, int max_entries, int slice) {
    LdsPlan p;
    p.cap_rows = (max_rows + 3) / 4 * 4;
    p.cap_entries = (max_entries + 7) / 8 * 8 + 8;
    p.cap_long = (max_entries / (kLdsShort + 1) + 8) / 8 * 8;
    p.bytes = (size_t)p.cap_rows * slice * 16 + (size_t)2 * p.cap_long * slice * 16 + (size_t)p.cap_entries * 4 + (size_t)(p.cap_rows + 4) * 4 +
              (size_t)p.cap_entries * 2 + (size_t)p.cap_long * 2 + 16;
    return p;
}

template <bool BWD>
__global__ __launch_bounds__(1024) void appnp_lds_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                         const float *__restrict__ val, const int32_t *__restrict__ ranges, int32_t n_ranges,
                                                         const float4 *__restrict__ X, float4 *__restrict__ Y, int32_t h4, int32_t K, float alpha,
                                                         int32_t max_rows, int32_t max_entries, int32_t slice, int32_t dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char al_lds[];
    const LdsPlan P = appnp_lds_plan(max_rows, max_entries, slice);
    float4 *buf = reinterpret_cast<float4 *>(al_lds);          // the slice of the signal: ONE buffer (see the step loop)
    float4 *s_keep = buf + P.cap_rows * slice;                  // long rows: z_0 / alpha-sum
    float4 *s_ylong = s_keep + P.cap_long * slice;              // long rows: the step's results until the barrier
    float *s_val = reinterpret_cast<float *>(s_ylong + P.cap_long * slice);
    int32_t *s_rp = reinterpret_cast<int32_t *>(s_val + P.cap_entries);
    uint16_t *s_col = reinterpret_cast<uint16_t *>(s_rp + P.cap_rows + 4);
    uint16_t *s_long = s_col + P.cap_entries;
    int32_t *s_cnt = reinterpret_cast<int32_t *>(s_long + P.cap_long);
    const int u = blockIdx.x;
    if (u >= n_ranges) return;
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, W = T >> 6;
    const int r0 = ranges[2 * u], r1 = ranges[2 * u + 1];
    const int n = r1 - r0;
    const int E0 = rowptr[r0];
    const int nE = rowptr[r1] - E0;
    if (n > max_rows || nE > max_entries || (n << 0) * slice > kLdsKeep * T) return;   // not the list the launch was sized for (uniform)
    if (tid == 0) *s_cnt = 0;
    for (int i = tid; i <= n; i += T) s_rp[i] = rowptr[r0 + i] - E0;
    for (int e = tid; e < nE; e += T) {
        s_col[e] = (uint16_t)(col[E0 + e] - r0);
        s_val[e] = val[E0 + e];
    }
    __syncthreads();
    for (int i = tid; i < n; i += T)
        if (s_rp[i + 1] - s_rp[i] > kLdsShort) s_long[atomicAdd(s_cnt, 1)] = (uint16_t)i;
    __syncthreads();
    const int n_long = *s_cnt;
    const float beta = 1.0f - alpha;
    for (int c0 = 0; c0 < h4;) {
        // this slice: w = the largest power of two <= min(slice, h4 - c0) columns
        int lw = 0;
        while ((2 << lw) <= slice && (2 << lw) <= h4 - c0) ++lw;
        const int w = 1 << lw;
        const int total = n << lw;
        float4 keep[kLdsKeep];
#pragma unroll
        for (int j = 0; j < kLdsKeep; ++j) {
            const int i = tid + j * T;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < total) {
                x = X[(int64_t)(r0 + (i >> lw)) * h4 + c0 + (i & (w - 1))];
                buf[i] = x;
            }
            keep[j] = BWD ? make_float4(0.f, 0.f, 0.f, 0.f) : x;
        }
        __syncthreads();
        for (int i = tid; i < (n_long << lw); i += T)   // the long rows' z_0 / alpha-sum
            s_keep[i] = BWD ? make_float4(0.f, 0.f, 0.f, 0.f) : buf[((int)s_long[i >> lw] << lw) + (i & (w - 1))];
        __syncthreads();
        // a thread's items keep their rows across the K steps, and the CSR does not change: row bounds and the first four entries'
        // operand positions -- all there is for 98 % of the rows of a coarsened batch -- are read ONCE per slice, so a step is one LDS
        // round trip for such an item (its four operand reads and their values), two of the thread's items in flight together
        // (packed: entry offset << 5 | count; operand item indices (< 4 096) two per register -- 1 024 threads leave 128 registers each)
        uint32_t it_en[kLdsKeep], it_c[kLdsKeep][2];
#pragma unroll
        for (int j = 0; j < kLdsKeep; ++j) {
            const int i = tid + j * T;
            int e = 0, cnt = 0;   // cnt 0 = not this thread's (beyond the range, or a wavefront's long row)
            if (i < total) {
                const int row = i >> lw;
                e = s_rp[row];
                const int len = s_rp[row + 1] - e;
                cnt = len <= kLdsShort ? len : 0;
            }
            it_en[j] = ((uint32_t)e << 5) | (uint32_t)cnt;
            uint32_t c4[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) c4[t] = s_col[e + (t < cnt ? t : 0)];   // (unconditional reads, all in flight: entry 0 exists)
#pragma unroll
            for (int t = 0; t < 4; ++t) c4[t] = t < cnt ? (uint32_t)(((int)c4[t] << lw) + (i & (w - 1))) : 0u;
            it_c[j][0] = c4[0] | (c4[1] << 16);
            it_c[j][1] = c4[2] | (c4[3] << 16);
        }
        // ONE signal buffer: a step's results wait in registers (a thread's own items) and in s_ylong (the long rows) until every
        // wavefront has gathered, then overwrite the buffer -- two barriers per step instead of one (a barrier costs 0.05 us here)
        // for half the LDS per row, i.e. twice the slice width for a large subgraph: half the passes over its K steps
        for (int k = 0; k < K; ++k) {
            const float4 *cur = buf;
            float4 yv[kLdsKeep];
            if (!(dbg & 2))
#pragma unroll
            for (int jh = 0; jh < kLdsKeep; jh += kLdsFly) {   // kLdsFly items' eight operand reads in flight (all four items': spills at 1 024 threads)
            if (BWD) {   // the running alpha-sum first (its operand is not live beside the eight gathers below)
#pragma unroll
                for (int j = jh; j < jh + kLdsFly; ++j)
                    if ((it_en[j] & 31u) > 0) {
                        const float4 xs = cur[tid + j * T];
                        keep[j].x = fmaf(alpha, xs.x, keep[j].x); keep[j].y = fmaf(alpha, xs.y, keep[j].y);
                        keep[j].z = fmaf(alpha, xs.z, keep[j].z); keep[j].w = fmaf(alpha, xs.w, keep[j].w);
                    }
            }
            float4 x[kLdsKeep][4];
            float it_v[kLdsKeep][4];   // (the values ride the same round trip as the operands: keeping them too spills)
#pragma unroll
            for (int j = jh; j < jh + kLdsFly; ++j)   // (the sixteen operand ADDRESSES, hoisted out of the K loop, would undo the packing)
                asm volatile("" : "+v"(it_c[j][0]), "+v"(it_c[j][1]), "+v"(it_en[j]));
#pragma unroll
            for (int j = jh; j < jh + kLdsFly; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    x[j][t] = cur[(it_c[j][t >> 1] >> (16 * (t & 1))) & 0xffffu];   // (an absent entry reads slot 0 at weight 0)
                    it_v[j][t] = t < (int)(it_en[j] & 31u) ? s_val[(it_en[j] >> 5) + t] : 0.f;
                }
#pragma unroll
            for (int j = jh; j < jh + kLdsFly; ++j) {
                if ((it_en[j] & 31u) > 0) {
                    const int i = tid + j * T;
                    const int q = i & (w - 1);
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc.x = fmaf(it_v[j][t], x[j][t].x, acc.x); acc.y = fmaf(it_v[j][t], x[j][t].y, acc.y);
                        acc.z = fmaf(it_v[j][t], x[j][t].z, acc.z); acc.w = fmaf(it_v[j][t], x[j][t].w, acc.w);
                    }
                    const int e1 = (int)(it_en[j] >> 5) + (int)(it_en[j] & 31u);
                    for (int e = (int)(it_en[j] >> 5) + 4; e < e1; e += 2) {   // entries 5 .. 16 (2 % of the rows), two at a time; sums in CSR order
                        const int eb = min(e + 1, e1 - 1);
                        const int ca = s_col[e], cb = s_col[eb];
                        const float va = s_val[e], vb = e + 1 < e1 ? s_val[eb] : 0.f;
                        const float4 xa = cur[(ca << lw) + q], xb = cur[(cb << lw) + q];
                        acc.x = fmaf(va, xa.x, acc.x); acc.y = fmaf(va, xa.y, acc.y); acc.z = fmaf(va, xa.z, acc.z); acc.w = fmaf(va, xa.w, acc.w);
                        acc.x = fmaf(vb, xb.x, acc.x); acc.y = fmaf(vb, xb.y, acc.y); acc.z = fmaf(vb, xb.z, acc.z); acc.w = fmaf(vb, xb.w, acc.w);
                    }
                    float4 y = make_float4(beta * acc.x, beta * acc.y, beta * acc.z, beta * acc.w);
                    if (!BWD) {
                        y.x = fmaf(alpha, keep[j].x, y.x); y.y = fmaf(alpha, keep[j].y, y.y);
                        y.z = fmaf(alpha, keep[j].z, y.z); y.w = fmaf(alpha, keep[j].w, y.w);
                    }
                    yv[j] = y;
                }
            }
            }
            if (!(dbg & 1)) {
                // the long rows, 8 / w at a time per wavefront: lane = (group, entry slot 0..7, slice column); a lane sums entries
                // e0 + slot, e0 + slot + 8, ... in order (four in flight), the eight slots fold by xor shuffles in a fixed tree
                const int q_l = lane & (w - 1), slot = (lane >> lw) & 7, grp = lane >> (lw + 3);
                const int rpw = 8 >> lw;
                for (int jb = 0; jb < n_long; jb += W * rpw) {
                    const int j = jb + wave * rpw + grp;
                    const bool on = j < n_long;
                    int row = 0, e = 0, e1 = 0;
                    if (on) {
                        row = s_long[j];
                        e = s_rp[row] + slot;
                        e1 = s_rp[row + 1];
                    }
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (; e < e1; e += 32) {
                        int c[4];
                        float v[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const bool has = e + 8 * t < e1;
                            c[t] = s_col[has ? e + 8 * t : e];
                            v[t] = has ? s_val[e + 8 * t] : 0.f;
                        }
                        float4 x[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) x[t] = cur[(c[t] << lw) + q_l];
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            acc.x = fmaf(v[t], x[t].x, acc.x); acc.y = fmaf(v[t], x[t].y, acc.y);
                            acc.z = fmaf(v[t], x[t].z, acc.z); acc.w = fmaf(v[t], x[t].w, acc.w);
                        }
                    }
#pragma unroll
                    for (int st = 4; st >= 1; st >>= 1) {
                        const int off = st << lw;
                        acc.x += __shfl_xor(acc.x, off, 64); acc.y += __shfl_xor(acc.y, off, 64);
                        acc.z += __shfl_xor(acc.z, off, 64); acc.w += __shfl_xor(acc.w, off, 64);
                    }
                    if (on && slot == 0) {
                        const int i = (row << lw) + q_l;
                        float4 y = make_float4(beta * acc.x, beta * acc.y, beta * acc.z, beta * acc.w);
                        float4 kp = s_keep[(j << lw) + q_l];
                        if (BWD) {
                            const float4 x = cur[i];
                            kp.x = fmaf(alpha, x.x, kp.x); kp.y = fmaf(alpha, x.y, kp.y); kp.z = fmaf(alpha, x.z, kp.z); kp.w = fmaf(alpha, x.w, kp.w);
                            s_keep[(j << lw) + q_l] = kp;
                        } else {
                            y.x = fmaf(alpha, kp.x, y.x); y.y = fmaf(alpha, kp.y, y.y); y.z = fmaf(alpha, kp.z, y.z); y.w = fmaf(alpha, kp.w, y.w);
                        }
                        s_ylong[(j << lw) + q_l] = y;
                    }
                }
            }
            __syncthreads();   // every gather of the step done
#pragma unroll
            for (int j = 0; j < kLdsKeep; ++j)
                if ((it_en[j] & 31u) > 0) buf[tid + j * T] = yv[j];
            if (!(dbg & 1))
                for (int i = tid; i < (n_long << lw); i += T) buf[((int)s_long[i >> lw] << lw) + (i & (w - 1))] = s_ylong[i];
            __syncthreads();
        }
        float4 *fin = buf;
        if (BWD) {   // the long rows' sums: folded into the final buffer first (the lanes that hold them are not the items' owners)
            for (int i = tid; i < (n_long << lw); i += T) {
                const int at = ((int)s_long[i >> lw] << lw) + (i & (w - 1));
                float4 y = fin[at];
                const float4 kp = s_keep[i];
                y.x += kp.x; y.y += kp.y; y.z += kp.z; y.w += kp.w;
                fin[at] = y;
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < kLdsKeep; ++j) {
            int i = tid + j * T;
            asm volatile("" : "+v"(i));   // re-derive the store addresses here: held across the K steps they cost eight registers (spills)
            if (i < total) {
                float4 y = fin[i];
                if (BWD) { y.x += keep[j].x; y.y += keep[j].y; y.z += keep[j].z; y.w += keep[j].w; }   // (a long row's keep[j] stayed 0)
                Y[(int64_t)(r0 + (i >> lw)) * h4 + c0 + (i & (w - 1))] = y;
            }
        }
        __syncthreads();   // the buffers are the next slice's
        c0 += w;
    }
}

extern "C" int64_t fitgnn_appnp_lds_bytes(int32_t max_rows, int32_t max_entries, int32_t slice) {
    if (max_rows < 1 || max_rows > 65535 || max_entries < 0 || (slice != 1 && slice != 2 && slice != 4)) return -1;
    return (int64_t)appnp_lds_plan(max_rows, max_entries, slice).bytes;
}
extern "C" int fitgnn_appnp_lds_max_bytes(void) { return kLdsMaxBytes; }
extern "C" int fitgnn_appnp_lds_items_per_thread(void) { return kLdsKeep; }

extern "C" int fitgnn_appnp_lds_f32(const int32_t *rowptr, const int32_t *col, const float *val, const int32_t *ranges, int32_t n_ranges,
                                    int32_t max_rows, int32_t max_entries, const float *X, float *Y, int32_t h4, int32_t K, float alpha,
                                    int32_t backward, int32_t threads, int32_t slice, void *stream) {
    if (n_ranges < 0 || h4 < 1 || h4 > 16 || K < 0) return FITGNN_E_BADARG;
    if (n_ranges == 0) return 0;
    if (threads < 64 || threads > 1024 || threads % 64 != 0 || (slice != 1 && slice != 2 && slice != 4)) return FITGNN_E_BADARG;
    if (max_rows < 1 || max_rows > 65535 || max_entries < 0 || (int64_t)max_rows * slice > (int64_t)kLdsKeep * threads) return FITGNN_E_BADARG;
    const LdsPlan P = appnp_lds_plan(max_rows, max_entries, slice);
    if (P.bytes > (size_t)kLdsMaxBytes) return FITGNN_E_BADARG;
    if (!rowptr || !col || !val || !ranges || !X || !Y) return FITGNN_E_BADARG;
    if ((((uintptr_t)X | (uintptr_t)Y) % 16) != 0) return FITGNN_E_ALIGN;
    const void *fn = backward ? (const void *)appnp_lds_kernel<true> : (const void *)appnp_lds_kernel<false>;
    const char *dv = getenv("FITGNN_APPNP_LDS_DEBUG");
    const int dbg = dv ? atoi(dv) : 0;
    if (P.bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.bytes);
        if (e != hipSuccess) return (int)e;
    }
    if (backward)
        hipLaunchKernelGGL(appnp_lds_kernel<true>, dim3((unsigned)n_ranges), dim3(threads), P.bytes, (hipStream_t)stream, rowptr, col, val, ranges,
                           n_ranges, (const float4 *)X, (float4 *)Y, h4, K, alpha, max_rows, max_entries, slice, dbg);
    else
        hipLaunchKernelGGL(appnp_lds_kernel<false>, dim3((unsigned)n_ranges), dim3(threads), P.bytes, (hipStream_t)stream, rowptr, col, val, ranges,
                           n_ranges, (const float4 *)X, (float4 *)Y, h4, K, alpha, max_rows, max_entries, slice, dbg);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_csr_row_sum_f32(const int32_t *rowptr, const float *v, int32_t n, float *y, void *stream) {
    if (n < 0) return FITGNN_E_BADARG;
    if (n == 0) return 0;
    if (!rowptr || !y) return FITGNN_E_BADARG;
    hipLaunchKernelGGL(csr_row_sum_kernel, dim3((unsigned)(((int64_t)n * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rowptr, v, n, y);
    return (int)hipGetLastError();
}
