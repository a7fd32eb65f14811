// gcn_ops.hip -- symmetric GCN normalisation and the backward of the fused bias/ELU/dropout epilogue.
//
// Replaces torch_geometric.nn.conv.gcn_conv.gcn_norm (called by GCNConv.forward from FIT-GNN
// network.py:31) and the autograd backward of network.py:32-33 (F.elu, F.dropout) + GCNConv's bias.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "common.h"
#include "fitgnn_hip.h"

namespace {

// deg[i] = sum of incoming edge weights (self loop included by the caller); dinv = deg^-1/2, inf -> 0
__global__ void gcn_deg_kernel(const int32_t *__restrict__ rowptr, const float *__restrict__ w, float *__restrict__ dinv,
                               int32_t n_rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    float d = 0.f;
    if (w) {
        for (int e = e0; e < e1; ++e) d += w[e];
    } else {
        d = (float)(e1 - e0);
    }
    dinv[i] = d > 0.f ? 1.0f / sqrtf(d) : 0.f;
}

__global__ void gcn_val_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                               const float *__restrict__ w, const float *__restrict__ dinv, float *__restrict__ val,
                               int32_t n_rows) {
    // one wave per row: coalesced over the row's non-zeros
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    const int e0 = rowptr[row], e1 = rowptr[row + 1];
    const float di = dinv[row];
    for (int e = e0 + lane; e < e1; e += 64) {
        const float we = w ? w[e] : 1.0f;
        val[e] = di * we * dinv[col[e]];
    }
}

constexpr int kMaxChunks = 1024;  // row chunks of the bias-gradient reduction (one partial row of H floats each)

inline int chunk_rows_for(int n_rows) {
    const int per = (n_rows + kMaxChunks - 1) / kMaxChunks;
    return std::max(4, (per + 3) / 4 * 4);
}

// dZ = dOut * dropout' * elu'  and per-(row chunk) column partial sums for the bias gradient.
constexpr int kMaxHeadC = 16;     // widest head (classes) whose weight gradient is accumulated in this kernel's registers
constexpr int kMaxHeadWide = 48;  // widest head whose dOut = dy @ Wl is formed here (CW == 0: dWl computed elsewhere)

// dOut is either read (HEAD == false) or formed on the fly as dy @ Wl (HEAD == true: the backward of the output
// head lt1, network.py:34, whose K = num_classes GEMM would otherwise write and re-read a full [rows x H] matrix).
// CW > 0 (HEAD only): also accumulate the head's weight gradient dWl[c][h] = sum_rows dy[row][c] * out[row][h] for
// c < C <= CW in registers (out is being read anyway), reduced through partialW like the bias gradient.
template <int VEC, bool HEAD, int CW>
__global__ __launch_bounds__(256) void epilogue_bwd_kernel(const float *__restrict__ dOut, const float *__restrict__ out,
                                                           float *__restrict__ dZ, int32_t n_rows, int32_t H,
                                                           int32_t chunk_rows, uint32_t epi, float p_drop, uint64_t seed_arg,
                                                           const uint8_t *__restrict__ mask, float *__restrict__ partial,
                                                           const float *__restrict__ dy, const float *__restrict__ Wl,
                                                           int32_t C, float *__restrict__ partialW,
                                                           const int64_t *__restrict__ sel, int32_t compact_in) {
    // sel (may be NULL): row i of dZ / of the sums is ORIGINAL row sel[i] -- the compact form over the rows that reach the loss
    // (fitgnn_epilogue_bwd_head_rows_f32).  The mask / dropout hash is always that row's; dy and out are indexed by it too
    // unless compact_in (they are then compact themselves: row i)
    const uint64_t seed = fitgnn::resolve_seed(seed_arg, epi);
    constexpr int SLAB = 64 * VEC;
    __shared__ float red[4][SLAB];
    __shared__ float s_w[HEAD ? (CW == 0 ? kMaxHeadWide : kMaxHeadC) * SLAB : 1];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int col0 = blockIdx.x * SLAB + lane * VEC;
    const bool live = col0 + VEC <= H;
    const int r0 = blockIdx.y * chunk_rows;
    const int r1 = min(r0 + chunk_rows, n_rows);
    const float scale = (epi & FITGNN_EPI_DROPOUT) ? 1.0f / (1.0f - p_drop) : 1.0f;
    const float unscale = (epi & FITGNN_EPI_DROPOUT) ? (1.0f - p_drop) : 1.0f;
    const uint32_t thresh = fitgnn::dropout_threshold(p_drop);
    float sum[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) sum[i] = 0.f;
    float wacc[CW > 0 ? CW : 1][VEC];
#pragma unroll
    for (int c = 0; c < (CW > 0 ? CW : 1); ++c)
#pragma unroll
        for (int i = 0; i < VEC; ++i) wacc[c][i] = 0.f;
    if (HEAD) {  // this slab's columns of the head weight, [C x SLAB]
        for (int i = threadIdx.x; i < C * SLAB; i += 256) {
            const int c = i / SLAB, j = i - c * SLAB;
            const int h = blockIdx.x * SLAB + j;
            s_w[i] = h < H ? Wl[(int64_t)c * H + h] : 0.f;
        }
        __syncthreads();
    }
    if (live) {
        constexpr int U = 4;  // rows per wave in flight: all loads of a step are issued before the first use
        for (int row0 = r0 + wave; row0 < r1; row0 += 4 * U) {
            float g[U][VEC], o[U][VEC], dyl[U];
            bool nz[U];
            if (HEAD) {
                // The head's gradient rows first: a row of zeros (a node that is not in the loss -- with --extra_node most
                // rows of a subgraph, utils.py:695-698) makes dOut, hence dZ, a row of zeros whatever `out` holds, so its
                // `out` row is not read and its C x VEC products are not formed.  Same bits as the long way round.
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int row = min(row0 + 4 * u, r1 - 1);
                    const int64_t srow = (sel && !compact_in) ? sel[row] : (int64_t)row;
                    dyl[u] = lane < C ? dy[srow * C + lane] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) nz[u] = __ballot((__float_as_uint(dyl[u]) & 0x7fffffffu) != 0u) != 0ull;  // wave-uniform
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = min(row0 + 4 * u, r1 - 1);  // clamped: rows past the chunk are loaded again, never used
                const int64_t base = ((sel && !compact_in) ? sel[row] : (int64_t)row) * H + col0;
                if (HEAD && !nz[u]) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) o[u][i] = 0.f;
                    continue;
                }
                if (VEC == 4) {
                    const float4 ov = *reinterpret_cast<const float4 *>(out + base);
                    o[u][0] = ov.x; o[u][1 % VEC] = ov.y; o[u][2 % VEC] = ov.z; o[u][3 % VEC] = ov.w;
                    if (!HEAD) {
                        const float4 gv = *reinterpret_cast<const float4 *>(dOut + base);
                        g[u][0] = gv.x; g[u][1 % VEC] = gv.y; g[u][2 % VEC] = gv.z; g[u][3 % VEC] = gv.w;
                    }
                } else {
                    o[u][0] = out[base];
                    if (!HEAD) g[u][0] = dOut[base];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = row0 + 4 * u;
                if (row >= r1) break;  // wave-uniform
                const int64_t zbase = (int64_t)row * H + col0;                       // where the row of dZ goes
                const int64_t base = (sel ? sel[row] : (int64_t)row) * H + col0;      // the row's mask / dropout index
                if (HEAD && !nz[u]) {  // dZ row = +0 (what 0 * dropout' * elu' gives); nothing to add to the sums
                    if (VEC == 4) {
                        *reinterpret_cast<float4 *>(dZ + zbase) = make_float4(0.f, 0.f, 0.f, 0.f);
                    } else {
                        dZ[zbase] = 0.f;
                    }
                    continue;
                }
                if (HEAD) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) g[u][i] = 0.f;
                    for (int c = 0; c < C; ++c) {
                        const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dyl[u]), c));
#pragma unroll
                        for (int i = 0; i < VEC; ++i) g[u][i] = fmaf(d, s_w[c * SLAB + lane * VEC + i], g[u][i]);
                    }
                    if (CW > 0) {
#pragma unroll
                        for (int c = 0; c < (CW > 0 ? CW : 1); ++c) {
                            const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dyl[u]), c));  // 0 for c >= C
#pragma unroll
                            for (int i = 0; i < VEC; ++i) wacc[c][i] = fmaf(d, o[u][i], wacc[c][i]);
                        }
                    }
                }
                uint64_t bits = 0;
                if ((epi & FITGNN_EPI_DROPOUT) && !mask) bits = fitgnn::dropout_bits(seed, (uint64_t)base >> 2);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float d = g[u][i];
                    if (epi & FITGNN_EPI_DROPOUT) {
                        const uint64_t idx = (uint64_t)base + i;
                        const bool keep = mask ? (mask[idx] != 0) : fitgnn::dropout_keep(bits, (int)(idx & 3), thresh);
                        d = keep ? d * scale : 0.f;
                    }
                    if (epi & FITGNN_EPI_ELU) {
                        const float e = o[u][i] * unscale;  // pre-dropout ELU output; exp(z) = e + 1 for z <= 0
                        d = e > 0.f ? d : d * (e + 1.0f);
                    }
                    g[u][i] = d;
                    sum[i] += d;
                }
                if (VEC == 4) {
                    *reinterpret_cast<float4 *>(dZ + zbase) = make_float4(g[u][0], g[u][1 % VEC], g[u][2 % VEC], g[u][3 % VEC]);
                } else {
                    dZ[zbase] = g[u][0];
                }
            }
        }
    }
    if (partial) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) red[wave][lane * VEC + i] = sum[i];
        __syncthreads();
        if (wave == 0 && live) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const int c = lane * VEC + i;
                partial[(int64_t)blockIdx.y * H + col0 + i] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
            }
        }
    }
    if (CW > 0 && partialW) {
        for (int cc = 0; cc < C; ++cc) {
            __syncthreads();
#pragma unroll
            for (int c = 0; c < (CW > 0 ? CW : 1); ++c)
                if (c == cc) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) red[wave][lane * VEC + i] = wacc[c][i];
                }
            __syncthreads();
            if (wave == 0 && live) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const int j = lane * VEC + i;
                    partialW[((int64_t)blockIdx.y * C + cc) * H + col0 + i] = ((red[0][j] + red[1][j]) + red[2][j]) + red[3][j];
                }
            }
        }
    }
}

// db[h] = sum over chunks (fixed order: reproducible).  16 columns x 64 chunk-phases per block: every thread sums
// n_chunks / 64 partials (a thread's loads are a serial chain of L2 round trips: keep it short), then a fixed
// binary tree over the 64 phases in LDS.
constexpr int kColsumPhases = 64, kColsumCols = 16;
// db2 / split (optional): columns >= split go to db2[h - split] (two destinations for one reduction: fitgnn_narrow_atb_f32's dW | db);
// tr_rows / tr_cols (optional): the first tr_rows x tr_cols columns are a row-major matrix that is stored transposed
__global__ __launch_bounds__(kColsumPhases * kColsumCols) void colsum_partials_kernel(const float *__restrict__ partial,
                                                                                      int32_t n_chunks, int32_t H,
                                                                                      float *__restrict__ db, float *__restrict__ db2 = nullptr,
                                                                                      int32_t split = 0, int32_t tr_rows = 0, int32_t tr_cols = 0) {
    __shared__ float red[kColsumPhases][kColsumCols];
    const int cl = threadIdx.x % kColsumCols, ph = threadIdx.x / kColsumCols;
    const int h = blockIdx.x * kColsumCols + cl;
    float s = 0.f;
    if (h < H) {
#pragma unroll 8   // eight loads in flight, added in the same order
        for (int c = ph; c < n_chunks; c += kColsumPhases) s += partial[(int64_t)c * H + h];
    }
    red[ph][cl] = s;
    __syncthreads();
    for (int w = kColsumPhases / 2; w >= 1; w >>= 1) {
        if (ph < w) red[ph][cl] = red[ph][cl] + red[ph + w][cl];
        __syncthreads();
    }
    if (ph == 0 && h < H) {
        if (db2 && h >= split) db2[h - split] = red[0][cl];
        else if (tr_rows > 0 && h < tr_rows * tr_cols) db[(h % tr_cols) * tr_rows + h / tr_cols] = red[0][cl];   // [tr_rows x tr_cols] -> its transpose
        else db[h] = red[0][cl];
    }
}

// out[w] = sum_b part[b][w], partials combined in a fixed tree: four interleaved running sums, then (s0+s1)+(s2+s3)
__global__ __launch_bounds__(256) void sum_leading_kernel(const float4 *__restrict__ part, int32_t B, int64_t W4,
                                                          float4 *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= W4) return;
    float4 s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    int b = 0;
    for (; b + 4 <= B; b += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 v = part[(int64_t)(b + k) * W4 + i];
            s[k].x += v.x; s[k].y += v.y; s[k].z += v.z; s[k].w += v.w;
        }
    }
    for (; b < B; ++b) {
        const float4 v = part[(int64_t)b * W4 + i];
        s[0].x += v.x; s[0].y += v.y; s[0].z += v.z; s[0].w += v.w;
    }
    out[i] = make_float4((s[0].x + s[1].x) + (s[2].x + s[3].x), (s[0].y + s[1].y) + (s[2].y + s[3].y),
                         (s[0].z + s[1].z) + (s[2].z + s[3].z), (s[0].w + s[1].w) + (s[2].w + s[3].w));
}

// Column sums of a tall matrix with a few columns (the head's bias gradient, sum over rows of dy [rows x classes]):
// block b sums rows [b * rows_per_block, ...) -- thread t takes rows t, t + 256, ... of the range, then a fixed tree
// over the 256 threads -- into partial[b][C]; colsum_partials_kernel adds the blocks.  C <= kNarrowMaxC (ogbn-products: 47).
constexpr int kNarrowMaxC = 64;
__global__ __launch_bounds__(256) void narrow_colsum_kernel(const float *__restrict__ x, int64_t ldx, int32_t n_rows, int32_t C,
                                                            int32_t rows_per_block, float *__restrict__ partial) {
    __shared__ float red[256];
    float acc[kNarrowMaxC];
#pragma unroll
    for (int c = 0; c < kNarrowMaxC; ++c) acc[c] = 0.f;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(r0 + rows_per_block, n_rows);
    for (int r = r0 + (int)threadIdx.x; r < r1; r += 256) {
        const float *xr = x + (int64_t)r * ldx;
#pragma unroll
        for (int c = 0; c < kNarrowMaxC; ++c)
            if (c < C) acc[c] += xr[c];
    }
    for (int c = 0; c < C; ++c) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < kNarrowMaxC; ++k)
            if (k == c) v = acc[k];
        red[threadIdx.x] = v;
        __syncthreads();
        for (int w = 128; w >= 1; w >>= 1) {
            if ((int)threadIdx.x < w) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * C + c] = red[0];
        __syncthreads();
    }
}

// torch.optim.Adam (amsgrad=False, maximize=False) over one flat parameter buffer: g += wd * p (L2, as torch's
// weight_decay), m = b1 m + (1-b1) g, v = b2 v + (1-b2) g^2, p -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps).
// `step` is a device counter (float, as torch keeps it) advanced by thread 0: the update can sit in a captured graph.
__global__ __launch_bounds__(256) void adam_flat_kernel(float4 *__restrict__ p, const float4 *__restrict__ g, float4 *__restrict__ m,
                                                        float4 *__restrict__ v, int64_t n4, float lr, float b1, float b2, float eps,
                                                        float wd, float *__restrict__ step) {
    const float t = *step + 1.0f;
    const float bc1 = 1.0f - powf(b1, t), bc2s = sqrtf(1.0f - powf(b2, t));
    const float step_size = lr / bc1;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
        float *pa = &pp.x, *ga = &gg.x, *ma = &mm.x, *va = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = ga[k] + wd * pa[k];
            ma[k] = b1 * ma[k] + (1.0f - b1) * gk;
            va[k] = b2 * va[k] + (1.0f - b2) * gk * gk;
            pa[k] -= step_size * ma[k] / (sqrtf(va[k]) / bc2s + eps);
        }
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}
__global__ void adam_step_advance_kernel(float *step) { *step += 1.0f; }

// The same update for a step whose backward wrote its weight gradients into a buffer of their own (g_new, laid out like the accumulated
// buffer g_acc: ops.GradSink) instead of adding them to g_acc tensor by tensor: the gradient used is g_acc + g_new and it is stored
// back to g_acc (run.py:254-304 never clears the gradients inside an epoch, so they accumulate over the batch steps), g_new is cleared -- the six to
// eight `grad += new` launches of a step fold into this one.  The LAST workgroup to finish (a ticket counter next to the step
// count: state[1], left at zero) advances the step count and, when given, the dropout seeds of the next captured step (every
// workgroup has read the count by the time it takes its ticket): no separate one-thread launches.
__global__ __launch_bounds__(256) void adam_flat_acc_kernel(float4 *__restrict__ p, float4 *__restrict__ g_acc, float4 *__restrict__ g_new,
                                                            float4 *__restrict__ m, float4 *__restrict__ v, int64_t n4, float lr, float b1,
                                                            float b2, float eps, float wd, float *__restrict__ state,
                                                            unsigned long long *__restrict__ seeds, int32_t n_seeds,
                                                            unsigned long long seed_stride) {
    const float t = state[0] + 1.0f;
    const float bc1 = 1.0f - powf(b1, t), bc2s = sqrtf(1.0f - powf(b2, t));
    const float step_size = lr / bc1;
    // four float4 per thread: a quarter of the workgroups, i.e. of the tickets taken below (the atomics of ~260 workgroups were most
    // of the launch on a 270 k-float model)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = ((int64_t)blockIdx.x * 4 + u) * 256 + threadIdx.x;
        if (i >= n4) continue;
        float4 pp = p[i], gg = g_acc[i], mm = m[i], vv = v[i];
        if (g_new) {
            const float4 gn = g_new[i];
            gg.x += gn.x; gg.y += gn.y; gg.z += gn.z; gg.w += gn.w;
            g_acc[i] = gg;
            g_new[i] = make_float4(0.f, 0.f, 0.f, 0.f);   // consumed: a step whose backward writes only some slices adds nothing stale
        }
        float *pa = &pp.x, *ga = &gg.x, *ma = &mm.x, *va = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = ga[k] + wd * pa[k];
            ma[k] = b1 * ma[k] + (1.0f - b1) * gk;
            va[k] = b2 * va[k] + (1.0f - b2) * gk * gk;
            pa[k] -= step_size * ma[k] / (sqrtf(va[k]) / bc2s + eps);
        }
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
    __syncthreads();   // every thread of the workgroup has read state[0]
    if (threadIdx.x == 0) {
        unsigned *ticket = reinterpret_cast<unsigned *>(state + 1);
        __threadfence();
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
            state[0] = t;
            *ticket = 0u;
            for (int q = 0; q < n_seeds; ++q) seeds[q] += seed_stride;
        }
    }
}

// NLLLoss(log_softmax(z)[idx], labels) (network.py:35 + run.py:341) and its gradient in one pass over the selected rows:
//   part[block] = sum over the block's rows of (logsumexp(z_r) - z_r[label]) * scale,  dz[r] = (softmax(z_r) - onehot) * scale
// dz is zero elsewhere (cleared by the launcher).  One thread per selected row; fixed-order block and grid reductions.
__global__ __launch_bounds__(256) void softmax_nll_kernel(const float *__restrict__ z, int64_t ldz, int32_t C,
                                                          const int64_t *__restrict__ idx, const int64_t *__restrict__ labels,
                                                          int32_t n, float scale, float *__restrict__ dz,
                                                          float *__restrict__ part) {
    __shared__ float red[256];
    const int t = blockIdx.x * 256 + threadIdx.x;
    float loss = 0.f;
    if (t < n) {
        const int64_t r = idx[t];
        const float *zr = z + r * ldz;
        float m = zr[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, zr[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(zr[c] - m);
        const float lse = m + logf(se);
        const int lab = (int)labels[t];
        loss = (lse - zr[lab]) * scale;
        float *dr = dz + r * ldz;
        for (int c = 0; c < C; ++c) dr[c] = (expf(zr[c] - lse) - (c == lab ? 1.f : 0.f)) * scale;
    }
    red[threadIdx.x] = loss;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void sum_partials_kernel(const float *__restrict__ part, int32_t n, float *__restrict__ out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// ------------------------------------------------------------------------------------------------
// forward epilogue on compact rows: z[i] <- dropout(ELU(z[i] + b)) with the dropout pattern of ORIGINAL row rows[i]
// ------------------------------------------------------------------------------------------------
// The arithmetic of the SpMM kernels' store epilogue (spmm.hip finish_row), for a last layer evaluated on the loss rows only
// (aggregate first, then the dense part on the rows that are kept): row i of the compact matrix is row rows[i] of the
// union, and its dropout hash / mask entry is that row's.
__global__ __launch_bounds__(256) void epilogue_fwd_rows_kernel(float *__restrict__ z, int64_t ldz, const int64_t *__restrict__ rows,
                                                                int32_t n, int32_t H, const float *__restrict__ bias, uint32_t epi,
                                                                float p_drop, uint64_t seed_arg, const uint8_t *__restrict__ mask) {
    const uint64_t seed = fitgnn::resolve_seed(seed_arg, epi);
    const float keep_scale = (epi & FITGNN_EPI_DROPOUT) ? 1.0f / (1.0f - p_drop) : 1.0f;
    const uint32_t thresh = fitgnn::dropout_threshold(p_drop);
    const int H4 = H >> 2;
    const int64_t total = (int64_t)n * H4;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t i = t / H4;
        const int col0 = (int)(t - i * H4) * 4;
        const int64_t orow = rows ? rows[i] : i;
        float4 v = *reinterpret_cast<float4 *>(z + i * ldz + col0);
        float x[4] = {v.x, v.y, v.z, v.w};
        const uint64_t idx0 = (uint64_t)orow * (uint64_t)H + (uint64_t)col0;
        uint64_t bits = 0;
        if ((epi & FITGNN_EPI_DROPOUT) && !mask) bits = fitgnn::dropout_bits(seed, idx0 >> 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float y = x[e] + ((epi & FITGNN_EPI_BIAS) ? bias[col0 + e] : 0.f);
            if (epi & FITGNN_EPI_ELU) y = y > 0.f ? y : __expf(y) - 1.0f;
            if (epi & FITGNN_EPI_DROPOUT) {
                const bool keep = mask ? (mask[idx0 + e] != 0) : fitgnn::dropout_keep(bits, (int)((idx0 + e) & 3), thresh);
                y = keep ? y * keep_scale : 0.f;
            }
            x[e] = y;
        }
        *reinterpret_cast<float4 *>(z + i * ldz + col0) = make_float4(x[0], x[1], x[2], x[3]);
    }
}

// ------------------------------------------------------------------------------------------------
// a layer's dense part on a FEW input columns: out = dropout(ELU(a W^T + b)),  a [n x K], K <= 32
// ------------------------------------------------------------------------------------------------
// GCNConv on an input with fewer columns than the layer is wide (QM9: 11 atom features -> hidden 512, network.py:189-204)
// is evaluated aggregate-first, (A_hat x) W^T: A_hat x has K columns, and since neither the graph nor the input features
// change between steps it is formed ONCE per batch (ops.FusedGCNLayerAggregatedInput) -- what is left of the layer per step is
// this pass: a K-term dot product per output element and the SpMM kernels' store epilogue (spmm.hip finish_row: same flags,
// same arithmetic, the dropout hash / mask entry of (row, column)).  A [n x K] @ [K x H] product with K = 11 is a pass over the
// OUTPUT, not a GEMM: W^T sits in LDS as [K][H] for the block's lifetime, a block takes kNarrowKRows rows of `a` at a time,
// a thread owns four consecutive output columns of one row (ds_read_b128 of W^T, the row's value a broadcast) and adds the
// K products in ascending k.
constexpr int kNarrowKMax = 32;
constexpr int kNarrowKRows = 16;
__global__ __launch_bounds__(256) void dense_narrow_k_kernel(const float *__restrict__ a, int64_t lda, const float *__restrict__ W,
                                                             int64_t ldw, int32_t n, int32_t K, int32_t H,
                                                             const float *__restrict__ bias, uint32_t epi, float p_drop,
                                                             uint64_t seed_arg, const uint8_t *__restrict__ mask,
                                                             float *__restrict__ out, int64_t ldo, int32_t rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) float nk_lds[];
    float *s_w = nk_lds;                    // [K][H]: W^T
    float *s_a = nk_lds + (size_t)K * H;    // [kNarrowKRows][K]
    const uint64_t seed = fitgnn::resolve_seed(seed_arg, epi);
    const float keep_scale = (epi & FITGNN_EPI_DROPOUT) ? 1.0f / (1.0f - p_drop) : 1.0f;
    const uint32_t thresh = fitgnn::dropout_threshold(p_drop);
    const int H4 = H >> 2;
    // W^T into LDS: the global reads walk W as it lies (coalesced, eight in flight per thread); the transposing stores collide on
    // banks (stride H), which costs cycles, not memory round trips
    {
        const int total = K * H;
        for (int i0 = threadIdx.x; i0 < total; i0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                const int j = i / K, k = i - j * K;
                v[u] = i < total ? W[(int64_t)j * ldw + k] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                const int j = i / K, k = i - j * K;
                if (i < total) s_w[k * H + j] = v[u];
            }
        }
    }
    const int rb = blockIdx.x * rows_per_block;
    const int re = min(n, rb + rows_per_block);
    for (int r0 = rb; r0 < re; r0 += kNarrowKRows) {
        const int rows = min(kNarrowKRows, re - r0);
        __syncthreads();   // W^T staged (first pass) / the previous chunk's rows consumed
        for (int i = threadIdx.x; i < rows * K; i += 256) {
            const int r = i / K, k = i - r * K;
            s_a[i] = a[(int64_t)(r0 + r) * lda + k];
        }
        __syncthreads();
        // a thread owns ONE column group and takes the chunk's rows four at a time (PH = 256 / H4 row phases when H4 divides 256,
        // else a plain item loop): W^T's float4 of a k is read once per four rows and the four dot products advance together --
        // taken one (row, column group) at a time, every k was an LDS round trip with nothing else in flight
        if (256 % H4 == 0) {
            const int PH = 256 / H4, cg = threadIdx.x % H4, ph = threadIdx.x / H4;
            const int col0 = 4 * cg;
            for (int rg = ph; rg < rows; rg += 4 * PH) {
                float x[4][4];
                int rr[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    rr[u] = min(rg + u * PH, rows - 1);   // clamped: computed again, not stored
                    x[u][0] = x[u][1] = x[u][2] = x[u][3] = 0.f;
                }
#pragma unroll 4
                for (int k = 0; k < K; ++k) {
                    const float4 w = *reinterpret_cast<const float4 *>(s_w + k * H + col0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float av = s_a[rr[u] * K + k];
                        x[u][0] = fmaf(av, w.x, x[u][0]); x[u][1] = fmaf(av, w.y, x[u][1]);
                        x[u][2] = fmaf(av, w.z, x[u][2]); x[u][3] = fmaf(av, w.w, x[u][3]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (rg + u * PH >= rows) break;
                    const int64_t orow = r0 + rr[u];
                    const uint64_t idx0 = (uint64_t)orow * (uint64_t)H + (uint64_t)col0;
                    uint64_t bits = 0;
                    if ((epi & FITGNN_EPI_DROPOUT) && !mask) bits = fitgnn::dropout_bits(seed, idx0 >> 2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float y = x[u][e] + ((epi & FITGNN_EPI_BIAS) ? bias[col0 + e] : 0.f);
                        if (epi & FITGNN_EPI_ELU) y = y > 0.f ? y : __expf(y) - 1.0f;
                        if (epi & FITGNN_EPI_DROPOUT) {
                            const bool keep = mask ? (mask[idx0 + e] != 0) : fitgnn::dropout_keep(bits, (int)((idx0 + e) & 3), thresh);
                            y = keep ? y * keep_scale : 0.f;
                        }
                        x[u][e] = y;
                    }
                    *reinterpret_cast<float4 *>(out + orow * ldo + col0) = make_float4(x[u][0], x[u][1], x[u][2], x[u][3]);
                }
            }
            continue;
        }
        for (int it = threadIdx.x; it < rows * H4; it += 256) {
            const int r = it / H4;
            const int col0 = (it - r * H4) * 4;
            float x[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < K; ++k) {
                const float av = s_a[r * K + k];
                const float4 w = *reinterpret_cast<const float4 *>(s_w + k * H + col0);
                x[0] = fmaf(av, w.x, x[0]); x[1] = fmaf(av, w.y, x[1]); x[2] = fmaf(av, w.z, x[2]); x[3] = fmaf(av, w.w, x[3]);
            }
            const int64_t orow = r0 + r;
            const uint64_t idx0 = (uint64_t)orow * (uint64_t)H + (uint64_t)col0;
            uint64_t bits = 0;
            if ((epi & FITGNN_EPI_DROPOUT) && !mask) bits = fitgnn::dropout_bits(seed, idx0 >> 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float y = x[e] + ((epi & FITGNN_EPI_BIAS) ? bias[col0 + e] : 0.f);
                if (epi & FITGNN_EPI_ELU) y = y > 0.f ? y : __expf(y) - 1.0f;
                if (epi & FITGNN_EPI_DROPOUT) {
                    const bool keep = mask ? (mask[idx0 + e] != 0) : fitgnn::dropout_keep(bits, (int)((idx0 + e) & 3), thresh);
                    y = keep ? y * keep_scale : 0.f;
                }
                x[e] = y;
            }
            *reinterpret_cast<float4 *>(out + orow * ldo + col0) = make_float4(x[0], x[1], x[2], x[3]);
        }
    }
}

// ... and its backward: dW[j][k] = sum_r dZ[r][j] a[r][k]  ([H x K]) and db[j] = sum_r dZ[r][j] in ONE pass over the incoming
// gradient.  With `prev` (the layer's output o = dropout(ELU(z))) the input is the gradient w.r.t. o and dZ is formed in registers
// (epilogue_bwd_kernel's arithmetic, the forward's flags / seed / mask) -- it is never written: the layer's input needs no
// gradient.  A block takes a range of rows; a thread owns four columns j of every PH-th row of the range (PH = 256 / (H / 4) row
// phases) and keeps its 4 x K + 4 sums in registers; the rows' a values come from LDS as broadcasts.  Per (block, phase) one
// partial row [H x K | H], reduced in a fixed order by colsum_partials_kernel.
constexpr int kNarrowAtbRows = 16;   // rows per block
template <int KT>
__global__ __launch_bounds__(256) void narrow_atb_kernel(const float *__restrict__ dZ, int64_t ldz, const float *__restrict__ prev,
                                                         uint32_t epi, float p_drop, uint64_t seed_arg, const uint8_t *__restrict__ mask,
                                                         const float *__restrict__ a, int64_t lda, int32_t n, int32_t K, int32_t H,
                                                         float *__restrict__ partial) {
    __shared__ float s_a[kNarrowAtbRows * KT];
    __shared__ float4 s_red[256];
    const uint64_t seed = fitgnn::resolve_seed(seed_arg, epi);
    const float keep_scale = (epi & FITGNN_EPI_DROPOUT) ? 1.0f / (1.0f - p_drop) : 1.0f;
    const float unscale = (epi & FITGNN_EPI_DROPOUT) ? 1.0f - p_drop : 1.0f;
    const uint32_t thresh = fitgnn::dropout_threshold(p_drop);
    const int H4 = H >> 2, PH = 256 / H4;
    const int cg = threadIdx.x % H4, ph = threadIdx.x / H4;
    const int r0 = blockIdx.x * kNarrowAtbRows;
    const int rows = min(kNarrowAtbRows, n - r0);
    for (int i = threadIdx.x; i < kNarrowAtbRows * KT; i += 256) {
        const int r = i / KT, k = i - r * KT;
        s_a[i] = (r < rows && k < K) ? a[(int64_t)(r0 + r) * lda + k] : 0.f;
    }
    __syncthreads();
    float acc[4][KT], bsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < KT; ++k) acc[e][k] = 0.f;
    constexpr int U = KT <= 16 ? 8 : 4;   // rows in flight per thread: every load of a group is issued before the first use
    for (int rg = ph; rg < rows; rg += PH * U) {
        float4 dv[U], ov[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = r0 + min(rg + u * PH, rows - 1);   // clamped: loaded again, never used
            dv[u] = *reinterpret_cast<const float4 *>(dZ + row * ldz + 4 * cg);
            ov[u] = prev ? *reinterpret_cast<const float4 *>(prev + row * (int64_t)H + 4 * cg) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = rg + u * PH;
            if (r >= rows) break;
            const int64_t row = r0 + r;
            float d[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
            if (prev) {
                const float o[4] = {ov[u].x, ov[u].y, ov[u].z, ov[u].w};
                const uint64_t idx0 = (uint64_t)row * (uint64_t)H + (uint64_t)(4 * cg);
                uint64_t bits = 0;
                if ((epi & FITGNN_EPI_DROPOUT) && !mask) bits = fitgnn::dropout_bits(seed, idx0 >> 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float g = d[e];
                    if (epi & FITGNN_EPI_DROPOUT) {
                        const bool keep = mask ? (mask[idx0 + e] != 0) : fitgnn::dropout_keep(bits, (int)((idx0 + e) & 3), thresh);
                        g = keep ? g * keep_scale : 0.f;
                    }
                    if (epi & FITGNN_EPI_ELU) {
                        const float ev = o[e] * unscale;
                        g = ev > 0.f ? g : g * (ev + 1.0f);
                    }
                    d[e] = g;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) bsum[e] += d[e];
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const float av = s_a[r * KT + k];
                acc[0][k] = fmaf(d[0], av, acc[0][k]); acc[1][k] = fmaf(d[1], av, acc[1][k]);
                acc[2][k] = fmaf(d[2], av, acc[2][k]); acc[3][k] = fmaf(d[3], av, acc[3][k]);
            }
        }
    }
    // the row phases of the block are added in ascending order through LDS, one k at a time, and the block's partial row is stored
    // k-major, [K][H] then the H bias sums: float4 stores, coalesced over the column groups (stored [H][K] every lane wrote 4-byte
    // pieces 44 bytes apart); the final reduction transposes
    float *dst = partial + (int64_t)blockIdx.x * ((int64_t)H * K + H);
#pragma unroll
    for (int k = 0; k <= KT; ++k) {
        if (k < KT && k >= K) continue;
        const float4 mine = k < KT ? make_float4(acc[0][k < KT ? k : 0], acc[1][k < KT ? k : 0], acc[2][k < KT ? k : 0], acc[3][k < KT ? k : 0])
                                   : make_float4(bsum[0], bsum[1], bsum[2], bsum[3]);
        __syncthreads();
        s_red[threadIdx.x] = mine;
        __syncthreads();
        if (ph == 0) {
            float4 t = s_red[cg];
            for (int q = 1; q < PH; ++q) {
                const float4 o = s_red[q * H4 + cg];
                t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
            }
            *reinterpret_cast<float4 *>(dst + (int64_t)(k < KT ? k : K) * H + 4 * cg) = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// output head on selected rows: y[rows[i]] = out[rows[i]] Wl^T + bl
// ------------------------------------------------------------------------------------------------
// Only the rows that reach the loss (run.py:193-204 keeps out[mask]: a cluster's own train nodes, 2 % of an --extra_node union)
// need the head's output, and a [rows x H] @ [H x C] product with C = 3..47 is a pass over `out`, not a GEMM.  Wl (padded to
// H + 4 floats per class) stays in LDS for the block's lifetime; a wave stages kHeadRows gathered rows in LDS (coalesced) and
// its lanes are (class, part): P = 64 / pow2(C) lanes share a class and take every P-th 16-byte chunk of h (a 16-lane read
// group then covers consecutive chunks: no bank conflict; the row values are broadcasts), so a 3-class head walks 8 chunks
// per lane, not 128.  Per (row, class): chunks in ascending order within a part, then the parts by a fixed xor tree.
constexpr int kHeadRows = 4;
__global__ __launch_bounds__(256) void head_rows_kernel(const float *__restrict__ out, long ldo, const int64_t *__restrict__ rows,
                                                        int n_rows, const float *__restrict__ Wl, const float *__restrict__ bl, int C,
                                                        int H, int P, float *__restrict__ y, long ldy, int out_compact) {
    extern __shared__ __attribute__((aligned(16))) float head_lds[];
    const int HP = H + 4;
    float *w_lds = head_lds;                                              // [C][HP]
    float *x_lds = head_lds + (size_t)C * HP + (threadIdx.x >> 6) * kHeadRows * H;  // this wave's [kHeadRows][H]
    const int H4 = H >> 2;
    for (int i = threadIdx.x; i < C * H4; i += blockDim.x) {
        const int c = i / H4, h4 = i - c * H4;
        *reinterpret_cast<float4 *>(w_lds + (size_t)c * HP + 4 * h4) = *reinterpret_cast<const float4 *>(Wl + (size_t)c * H + 4 * h4);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_groups = (n_rows + kHeadRows - 1) / kHeadRows;
    const int per_pass = 64 / P;  // classes per pass over the lanes
    for (int g = blockIdx.x * 4 + wave; g < n_groups; g += gridDim.x * 4) {
        const int r0 = g * kHeadRows;
        for (int i = lane; i < kHeadRows * H4; i += 64) {
            const int r = i / H4, h4 = i - r * H4;
            const int ri = min(r0 + r, n_rows - 1);
            const long src = out_compact ? (long)ri : (long)rows[ri];   // out_compact: row i of `out` IS the i-th selected row
            *reinterpret_cast<float4 *>(x_lds + r * H + 4 * h4) = *reinterpret_cast<const float4 *>(out + src * ldo + 4 * h4);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int c0 = 0; c0 < C; c0 += per_pass) {
            const int c = c0 + lane / P, part = lane % P;
            const bool live = c < C;
            float acc[kHeadRows];
#pragma unroll
            for (int r = 0; r < kHeadRows; ++r) acc[r] = 0.f;
            const float *wr = w_lds + (size_t)(live ? c : 0) * HP;
#pragma unroll 4   // the loop is a chain of LDS round trips otherwise: four iterations' reads in flight
            for (int h4 = part; h4 < H4; h4 += P) {
                const float4 w = *reinterpret_cast<const float4 *>(wr + 4 * h4);
#pragma unroll
                for (int r = 0; r < kHeadRows; ++r) {
                    const float4 x = *reinterpret_cast<const float4 *>(x_lds + r * H + 4 * h4);
                    acc[r] = fmaf(x.x, w.x, acc[r]);
                    acc[r] = fmaf(x.y, w.y, acc[r]);
                    acc[r] = fmaf(x.z, w.z, acc[r]);
                    acc[r] = fmaf(x.w, w.w, acc[r]);
                }
            }
            for (int off = 1; off < P; off <<= 1) {  // lanes of one class are contiguous: xor stays inside the class
#pragma unroll
                for (int r = 0; r < kHeadRows; ++r) acc[r] += __shfl_xor(acc[r], off, 64);
            }
            if (live && part == 0) {
                const float b = bl ? bl[c] : 0.f;
#pragma unroll
                for (int r = 0; r < kHeadRows; ++r)
                    if (r0 + r < n_rows) y[rows[r0 + r] * ldy + c] = acc[r] + b;
            }
        }
        __builtin_amdgcn_wave_barrier();  // the staged rows are rewritten by the next group
    }
}

}  // namespace

// loss[0] = scale * sum_i |out[i] - tgt[i]|, grad[i] = scale * sign(out[i] - tgt[i])  (L1Loss, run.py:518,716 on the graph-level /
// node-regression outputs: a few hundred values): ONE workgroup, a fixed-order tree -- instead of sub, abs, mean and their backward
__global__ __launch_bounds__(256) void l1_loss_kernel(const float *__restrict__ out, const float *__restrict__ tgt, int32_t n, float scale,
                                                      float *__restrict__ loss, float *__restrict__ grad) {
    __shared__ float part[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float d = out[i] - tgt[i];
        acc += fabsf(d);
        grad[i] = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = scale * part[0];
}

extern "C" int fitgnn_l1_loss_f32(const float *out, const float *tgt, int32_t n, float scale, float *loss, float *grad, void *stream) {
    if (n < 0) return FITGNN_E_BADARG;
    if (!loss) return FITGNN_E_BADARG;
    if (n > 0 && (!out || !tgt || !grad)) return FITGNN_E_BADARG;
    hipLaunchKernelGGL(l1_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, out, tgt, n, scale, loss, grad);
    return (int)hipGetLastError();
}

extern "C" size_t fitgnn_softmax_nll_workspace_bytes(int32_t n) { return (size_t)((n > 0 ? n : 0) + 255) / 256 * sizeof(float) + 16; }

extern "C" int fitgnn_softmax_nll_f32(const float *z, int64_t ldz, int32_t n_rows, int32_t C, const int64_t *idx,
                                      const int64_t *labels, int32_t n, float scale, float *loss, float *dz, void *work,
                                      size_t work_bytes, void *stream) {
    if (n_rows < 0 || C < 1 || n < 0 || ldz < C) return FITGNN_E_BADARG;
    if (!loss) return FITGNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (dz && n_rows > 0) FITGNN_RETURN_IF_HIP(hipMemsetAsync(dz, 0, (size_t)n_rows * (size_t)ldz * sizeof(float), s));
    if (n == 0) return (int)hipMemsetAsync(loss, 0, sizeof(float), s);
    if (!z || !idx || !labels || !dz || !work) return FITGNN_E_BADARG;
    if (work_bytes < fitgnn_softmax_nll_workspace_bytes(n)) return FITGNN_E_WORKSPACE;
    const int blocks = (n + 255) / 256;
    hipLaunchKernelGGL(softmax_nll_kernel, dim3(blocks), dim3(256), 0, s, z, ldz, C, idx, labels, n, scale, dz, (float *)work);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, (const float *)work, blocks, loss);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                                    float beta1, float beta2, float eps, float weight_decay, float *step, void *stream) {
    if (n < 0 || (n % 4) != 0) return FITGNN_E_BADARG;
    if (n == 0) return 0;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step) return FITGNN_E_BADARG;
    if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16) != 0) return FITGNN_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (float4 *)param, (const float4 *)grad,
                       (float4 *)exp_avg, (float4 *)exp_avg_sq, n4, lr, beta1, beta2, eps, weight_decay, step);
    hipLaunchKernelGGL(adam_step_advance_kernel, dim3(1), dim3(1), 0, s, step);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_adam_step_acc_f32(float *param, float *grad_acc, float *grad_new, float *exp_avg, float *exp_avg_sq, int64_t n,
                                        float lr, float beta1, float beta2, float eps, float weight_decay, float *state, uint64_t *seeds,
                                        int32_t n_seeds, uint64_t seed_stride, void *stream) {
    if (n < 0 || (n % 4) != 0 || n_seeds < 0) return FITGNN_E_BADARG;
    if (n == 0) return 0;
    if (!param || !grad_acc || !exp_avg || !exp_avg_sq || !state || (n_seeds > 0 && !seeds)) return FITGNN_E_BADARG;
    if ((((uintptr_t)param | (uintptr_t)grad_acc | (uintptr_t)grad_new | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16) != 0 ||
        ((uintptr_t)state % 4) != 0)
        return FITGNN_E_ALIGN;
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(adam_flat_acc_kernel, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, (float4 *)param,
                       (float4 *)grad_acc, (float4 *)grad_new, (float4 *)exp_avg, (float4 *)exp_avg_sq, n4, lr, beta1, beta2, eps,
                       weight_decay, state, (unsigned long long *)seeds, n_seeds, (unsigned long long)seed_stride);
    return (int)hipGetLastError();
}

extern "C" size_t fitgnn_head_rows_lds_bytes(int32_t H, int32_t C) {
    if (H <= 0 || C <= 0) return 0;
    return ((size_t)C * (H + 4) + (size_t)4 * kHeadRows * H) * sizeof(float);
}

extern "C" int fitgnn_head_rows_f32(const float *out, int64_t ldo, const int64_t *rows, int32_t n_rows, const float *Wl,
                                    const float *bl, int32_t C, int32_t H, float *y, int64_t ldy, int32_t out_compact, void *stream) {
    if (n_rows < 0 || C < 1 || H < 4 || (H % 4) != 0 || ldo < H || (ldo % 4) != 0 || ldy < C) return FITGNN_E_BADARG;
    if (n_rows == 0) return 0;
    if (!out || !rows || !Wl || !y) return FITGNN_E_BADARG;
    if ((((uintptr_t)out | (uintptr_t)Wl) % 16) != 0) return FITGNN_E_ALIGN;
    const size_t lds = fitgnn_head_rows_lds_bytes(H, C);
    if (lds > 160 * 1024) return FITGNN_E_BADARG;  // the head's weights do not fit LDS: the caller uses a GEMM
    static std::atomic<uint64_t> lds_done{0};
    if (const int rc = fitgnn_lds_limit_once((const void *)head_rows_kernel, 160 * 1024, lds_done)) return rc;
    const int groups = (n_rows + kHeadRows - 1) / kHeadRows;
    const int blocks = std::min((groups + 3) / 4, 256);
    int P = 1;  // lanes per class: 64 / pow2(C), at most one per 16-byte chunk of h
    while (P * 2 * C <= 64 && P * 2 <= H / 4) P *= 2;
    hipLaunchKernelGGL(head_rows_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, out, (long)ldo, rows, n_rows, Wl, bl, C, H, P, y,
                       (long)ldy, (int)out_compact);
    return (int)hipGetLastError();
}

extern "C" size_t fitgnn_colsum_narrow_workspace_bytes(int32_t n_rows, int32_t C) {
    if (n_rows <= 0 || C <= 0) return 0;
    const int blocks = std::min(256, (n_rows + 255) / 256);
    return (size_t)blocks * C * sizeof(float);
}

extern "C" int fitgnn_colsum_narrow_f32(const float *x, int64_t ldx, int32_t n_rows, int32_t C, float *out, void *work,
                                        size_t work_bytes, void *stream) {
    if (n_rows < 0 || C < 1 || C > kNarrowMaxC || ldx < C || !out) return FITGNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (n_rows == 0) return (int)hipMemsetAsync(out, 0, (size_t)C * sizeof(float), s);
    if (!x || !work) return FITGNN_E_BADARG;
    if (work_bytes < fitgnn_colsum_narrow_workspace_bytes(n_rows, C)) return FITGNN_E_WORKSPACE;
    const int blocks = std::min(256, (n_rows + 255) / 256);
    const int rows_per_block = (n_rows + blocks - 1) / blocks;
    hipLaunchKernelGGL(narrow_colsum_kernel, dim3(blocks), dim3(256), 0, s, x, ldx, n_rows, C, rows_per_block, (float *)work);
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((C + kColsumCols - 1) / kColsumCols), dim3(kColsumPhases * kColsumCols), 0, s,
                       (const float *)work, blocks, C, out);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_colsum_partials_f32(const float *partial, int32_t n_chunks, int32_t H, float *out, void *stream) {
    if (n_chunks < 1 || H < 1 || !partial || !out) return FITGNN_E_BADARG;
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((H + kColsumCols - 1) / kColsumCols), dim3(kColsumPhases * kColsumCols), 0,
                       (hipStream_t)stream, partial, n_chunks, H, out);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_sum_leading_f32(const float *part, int32_t B, int64_t W, float *out, void *stream) {
    if (B < 1 || W < 0 || (W % 4) != 0) return FITGNN_E_BADARG;
    if (W == 0) return 0;
    if (!part || !out) return FITGNN_E_BADARG;
    if ((((uintptr_t)part | (uintptr_t)out) % 16) != 0) return FITGNN_E_ALIGN;
    const int64_t W4 = W / 4;
    hipLaunchKernelGGL(sum_leading_kernel, dim3((unsigned)((W4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)part, B, W4, (float4 *)out);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_gcn_norm_csr_f32(const int32_t *rowptr, const int32_t *col, const float *w, float *val,
                                       float *dinv, int32_t n_rows, void *stream) {
    if (n_rows < 0) return FITGNN_E_BADARG;
    if (n_rows == 0) return 0;
    if (!rowptr || !col || !val || !dinv) return FITGNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gcn_deg_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s, rowptr, w, dinv, n_rows);
    const int64_t threads = (int64_t)n_rows * 64;
    hipLaunchKernelGGL(gcn_val_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, rowptr, col, w, dinv,
                       val, n_rows);
    return (int)hipGetLastError();
}

extern "C" size_t fitgnn_epilogue_bwd_workspace_bytes(int32_t n_rows, int32_t H) {
    if (n_rows <= 0 || H <= 0) return 0;
    const int cr = chunk_rows_for(n_rows);
    const size_t chunks = ((size_t)n_rows + cr - 1) / cr;
    return chunks * (size_t)H * sizeof(float);
}

namespace {
// Class c's dy value is fetched by lane c of every wave, and only the lanes that own columns of the wave's slab are active:
// the head must not have more classes than the narrowest (= last) slab has active lanes.
bool head_supported(int32_t H, int32_t C, bool with_dWl) {
    if (H < 1 || C < 1 || C > (with_dWl ? kMaxHeadC : kMaxHeadWide)) return false;
    const bool vec = (H % 4) == 0;
    const int slab = vec ? 256 : 64, per_lane = vec ? 4 : 1;
    const int last_cols = (H - 1) % slab + 1;
    return C <= last_cols / per_lane;
}

int epilogue_bwd_launch(const float *dOut, const float *dy, const float *Wl, int32_t C, const float *out, float *dZ,
                        int32_t n_rows, int32_t H, uint32_t epilogue, float p_drop, uint64_t seed, const uint8_t *mask,
                        float *db, float *dWl, void *work, size_t work_bytes, void *stream, const int64_t *sel = nullptr,
                        int32_t compact_in = 0) {
    if (n_rows < 0 || H < 0) return FITGNN_E_BADARG;
    if (n_rows == 0 || H == 0) return 0;
    const bool head = dOut == nullptr;
    if (!out || !dZ) return FITGNN_E_BADARG;
    if (head && (!dy || !Wl || !head_supported(H, C, dWl != nullptr))) return FITGNN_E_BADARG;
    if (!head && dWl) return FITGNN_E_BADARG;
    if ((epilogue & FITGNN_EPI_DROPOUT) && !(p_drop >= 0.f && p_drop < 1.f)) return FITGNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int cr = chunk_rows_for(n_rows);
    const int chunks = (n_rows + cr - 1) / cr;
    const size_t need = ((db ? 1 : 0) + (dWl ? (size_t)C : 0)) * (size_t)chunks * (size_t)H * sizeof(float);
    if (need && (!work || work_bytes < need)) return FITGNN_E_WORKSPACE;
    float *partial = db ? (float *)work : nullptr;
    float *partialW = dWl ? (float *)work + (db ? (size_t)chunks * H : 0) : nullptr;
    const bool vec = (H % 4 == 0) && ((((uintptr_t)dOut | (uintptr_t)out | (uintptr_t)dZ) % 16) == 0);
    // x (fastest) = column slab: both halves of every row are in flight together; y = row chunk
    const dim3 grid(vec ? (H + 255) / 256 : (H + 63) / 64, chunks);
#define FITGNN_LAUNCH_EB(V, HD, CWV)                                                                                     \
    hipLaunchKernelGGL((epilogue_bwd_kernel<V, HD, CWV>), grid, dim3(256), 0, s, dOut, out, dZ, n_rows, H, cr, epilogue,   \
                       p_drop, seed, mask, partial, dy, Wl, C, partialW, sel, compact_in)
    if (!head) {
        if (vec) FITGNN_LAUNCH_EB(4, false, 0); else FITGNN_LAUNCH_EB(1, false, 0);
    } else if (!dWl) {
        if (vec) FITGNN_LAUNCH_EB(4, true, 0); else FITGNN_LAUNCH_EB(1, true, 0);
    } else if (C <= 4) {
        if (vec) FITGNN_LAUNCH_EB(4, true, 4); else FITGNN_LAUNCH_EB(1, true, 4);
    } else {
        if (vec) FITGNN_LAUNCH_EB(4, true, kMaxHeadC); else FITGNN_LAUNCH_EB(1, true, kMaxHeadC);
    }
#undef FITGNN_LAUNCH_EB
    if (db) hipLaunchKernelGGL(colsum_partials_kernel, dim3((H + kColsumCols - 1) / kColsumCols), dim3(kColsumPhases * kColsumCols), 0, s, partial, chunks, H, db);
    if (dWl)  // partialW rows are [C x H] per chunk: the same reduction over C*H "columns"
        hipLaunchKernelGGL(colsum_partials_kernel, dim3((C * H + kColsumCols - 1) / kColsumCols), dim3(kColsumPhases * kColsumCols), 0, s, partialW, chunks, C * H, dWl);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int fitgnn_epilogue_bwd_f32(const float *dOut, const float *out, float *dZ, int32_t n_rows, int32_t H,
                                       uint32_t epilogue, float p_drop, uint64_t seed, const uint8_t *mask, float *db,
                                       void *work, size_t work_bytes, void *stream) {
    if (n_rows > 0 && H > 0 && !dOut) return FITGNN_E_BADARG;
    return epilogue_bwd_launch(dOut, nullptr, nullptr, 0, out, dZ, n_rows, H, epilogue, p_drop, seed, mask, db, nullptr, work,
                               work_bytes, stream);
}

extern "C" int fitgnn_head_max_classes(void) { return kMaxHeadC; }
extern "C" int fitgnn_head_max_classes_wide(void) { return kMaxHeadWide; }
extern "C" int fitgnn_epilogue_bwd_head_supported(int32_t H, int32_t C, int32_t with_dWl) { return head_supported(H, C, with_dWl != 0) ? 1 : 0; }

extern "C" size_t fitgnn_epilogue_bwd_head_workspace_bytes(int32_t n_rows, int32_t H, int32_t C) {
    if (n_rows <= 0 || H <= 0 || C < 0) return 0;
    return fitgnn_epilogue_bwd_workspace_bytes(n_rows, H) * (size_t)(1 + C);
}

extern "C" int fitgnn_epilogue_bwd_head_rows_f32(const float *dy, const float *Wl, int32_t C, const float *out, const int64_t *rows,
                                                 int32_t n_sel, int32_t inputs_compact, float *dZc, int32_t H, uint32_t epilogue,
                                                 float p_drop, uint64_t seed, const uint8_t *mask, float *db, float *dWl, void *work,
                                                 size_t work_bytes, void *stream) {
    if (n_sel > 0 && !rows) return FITGNN_E_BADARG;
    return epilogue_bwd_launch(nullptr, dy, Wl, C, out, dZc, n_sel, H, epilogue, p_drop, seed, mask, db, dWl, work, work_bytes, stream,
                               rows, inputs_compact ? 1 : 0);
}

extern "C" int fitgnn_epilogue_bwd_rows_f32(const float *dOut, const float *out, const int64_t *rows, int32_t n_sel, int32_t inputs_compact,
                                            float *dZc, int32_t H, uint32_t epilogue, float p_drop, uint64_t seed, const uint8_t *mask,
                                            float *db, void *work, size_t work_bytes, void *stream) {
    if (n_sel > 0 && (!rows || !dOut)) return FITGNN_E_BADARG;
    return epilogue_bwd_launch(dOut, nullptr, nullptr, 0, out, dZc, n_sel, H, epilogue, p_drop, seed, mask, db, nullptr, work, work_bytes,
                               stream, rows, inputs_compact ? 1 : 0);
}

extern "C" int fitgnn_epilogue_fwd_rows_f32(float *z, int64_t ldz, const int64_t *rows, int32_t n, int32_t H, const float *bias,
                                            uint32_t epilogue, float p_drop, uint64_t seed, const uint8_t *mask, void *stream) {
    if (n < 0 || H < 4 || (H % 4) != 0 || ldz < H || (ldz % 4) != 0) return FITGNN_E_BADARG;
    if (n == 0) return 0;
    if (!z) return FITGNN_E_BADARG;
    if ((epilogue & FITGNN_EPI_BIAS) && !bias) return FITGNN_E_BADARG;
    if ((epilogue & FITGNN_EPI_DROPOUT) && !(p_drop >= 0.f && p_drop < 1.f)) return FITGNN_E_BADARG;
    if (((uintptr_t)z % 16) != 0) return FITGNN_E_ALIGN;
    const int64_t total = (int64_t)n * (H / 4);
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(epilogue_fwd_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, z, ldz, rows, n, H, bias, epilogue, p_drop,
                       seed, mask);
    return (int)hipGetLastError();
}

extern "C" size_t fitgnn_dense_narrow_k_lds_bytes(int32_t K, int32_t H) {
    if (K < 1 || K > kNarrowKMax || H < 4 || (H % 4) != 0) return 0;
    const size_t b = ((size_t)K * (size_t)H + (size_t)kNarrowKRows * (size_t)K) * sizeof(float);
    return b <= 64 * 1024 ? b : 0;
}

extern "C" int fitgnn_dense_narrow_k_f32(const float *a, int64_t lda, const float *W, int64_t ldw, int32_t n, int32_t K, int32_t H,
                                         const float *bias, uint32_t epilogue, float p_drop, uint64_t seed, const uint8_t *mask,
                                         float *out, int64_t ldo, void *stream) {
    if (n < 0 || lda < K || ldw < K || ldo < H || (ldo % 4) != 0) return FITGNN_E_BADARG;
    const size_t lds = fitgnn_dense_narrow_k_lds_bytes(K, H);
    if (lds == 0) return FITGNN_E_BADARG;
    if (n == 0) return 0;
    if (!a || !W || !out) return FITGNN_E_BADARG;
    if ((epilogue & FITGNN_EPI_BIAS) && !bias) return FITGNN_E_BADARG;
    if ((epilogue & FITGNN_EPI_DROPOUT) && !(p_drop >= 0.f && p_drop < 1.f)) return FITGNN_E_BADARG;
    if (((uintptr_t)out % 16) != 0) return FITGNN_E_ALIGN;
    // one range of rows per block: W^T is staged once per block
    int rows_per_block = std::max((n + 511) / 512, 8);   // two workgroups per CU: one's LDS round trips under the other's
    const int blocks = (n + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(dense_narrow_k_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, a, lda, W, ldw, n, K, H, bias, epilogue,
                       p_drop, seed, mask, out, ldo, rows_per_block);
    return (int)hipGetLastError();
}

// H / 4 must divide 256 (a block's threads = whole row phases): hidden widths 16 ... 1024 in powers of two
extern "C" size_t fitgnn_narrow_atb_workspace_bytes(int32_t n, int32_t K, int32_t H) {
    if (n <= 0 || K < 1 || K > kNarrowKMax || H < 4 || (H % 4) != 0 || (H / 4) > 256 || (256 % (H / 4)) != 0) return 0;
    const size_t blocks = ((size_t)n + kNarrowAtbRows - 1) / kNarrowAtbRows;
    return blocks * ((size_t)H * (size_t)K + (size_t)H) * sizeof(float);
}

extern "C" int fitgnn_narrow_atb_f32(const float *d, int64_t ldd, const float *prev, uint32_t epilogue, float p_drop, uint64_t seed,
                                     const uint8_t *mask, const float *a, int64_t lda, int32_t n, int32_t K, int32_t H, float *dW, float *db,
                                     void *work, size_t work_bytes, void *stream) {
    if (n <= 0 || ldd < H || (ldd % 4) != 0 || lda < K) return FITGNN_E_BADARG;
    const size_t need = fitgnn_narrow_atb_workspace_bytes(n, K, H);
    if (need == 0) return FITGNN_E_BADARG;
    if (!d || !a || !dW || !db) return FITGNN_E_BADARG;
    if ((epilogue & FITGNN_EPI_DROPOUT) && !(p_drop >= 0.f && p_drop < 1.f)) return FITGNN_E_BADARG;
    if (!work || work_bytes < need) return FITGNN_E_WORKSPACE;
    if ((((uintptr_t)d | (uintptr_t)prev) % 16) != 0) return FITGNN_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (n + kNarrowAtbRows - 1) / kNarrowAtbRows;
    const int parts = blocks;
    const int width = H * K + H;
    float *partial = (float *)work;
    if (K <= 8) hipLaunchKernelGGL(narrow_atb_kernel<8>, dim3(blocks), dim3(256), 0, s, d, ldd, prev, epilogue, p_drop, seed, mask, a, lda, n, K, H, partial);
    else if (K <= 16) hipLaunchKernelGGL(narrow_atb_kernel<16>, dim3(blocks), dim3(256), 0, s, d, ldd, prev, epilogue, p_drop, seed, mask, a, lda, n, K, H, partial);
    else hipLaunchKernelGGL(narrow_atb_kernel<32>, dim3(blocks), dim3(256), 0, s, d, ldd, prev, epilogue, p_drop, seed, mask, a, lda, n, K, H, partial);
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((width + kColsumCols - 1) / kColsumCols), dim3(kColsumPhases * kColsumCols), 0, s, partial,
                       parts, width, dW, db, H * K, K, H);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_epilogue_bwd_head_f32(const float *dy, const float *Wl, int32_t C, const float *out, float *dZ,
                                            int32_t n_rows, int32_t H, uint32_t epilogue, float p_drop, uint64_t seed,
                                            const uint8_t *mask, float *db, float *dWl, void *work, size_t work_bytes,
                                            void *stream) {
    return epilogue_bwd_launch(nullptr, dy, Wl, C, out, dZ, n_rows, H, epilogue, p_drop, seed, mask, db, dWl, work, work_bytes,
                               stream);
}
