// lift_pool.hip -- adjacency lift Wc = Pinv^T W Pinv and feature pooling Xc = C X on gfx950.
//
// Replaces (FIT-GNN): coarsen_matrix graph_coarsening/coarsening_utils.py:201-205, zero_diag
// graph_coarsening/graph_utils.py:82-90, the symmetrisation coarsening_utils.py:139, and C.dot(X) at
// utils.py:161,393,738,827.  Summation orders are SciPy's, so results are bit-identical to the
// reference; compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"
#include "fitgnn_hip.h"

namespace {

constexpr size_t kAlign = 256;
inline size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }
inline dim3 blocks_for(int64_t threads, int block = 256) { return dim3((unsigned)((threads + block - 1) / block)); }
constexpr uint64_t kInvalid = ~0ull;

__device__ __forceinline__ int row_of_edge(const int32_t *rowptr, int N, int e) {
    // largest u with rowptr[u] <= e
    int lo = 0, hi = N;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (rowptr[mid] <= e) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// stage 1 input: one entry per directed edge (u,v): key = u*n + assign[v], value = w_uv * p_v
__global__ void lift_stage1_kernel(int32_t N, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                   const double *__restrict__ w, const int32_t *__restrict__ assign,
                                   const double *__restrict__ cval, int32_t n, int32_t nnz, uint64_t *__restrict__ key,
                                   double *__restrict__ val) {
#pragma clang fp contract(off)
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const int u = row_of_edge(rowptr, N, e);
    const int v = col[e];
    const int b = assign[v];
    if (assign[u] == b) { key[e] = kInvalid; val[e] = 0.0; return; }  // feeds only the diagonal (zero_diag)
    const double pv = cval[v] * (1.0 / cval[v]);
    key[e] = (uint64_t)u * (uint64_t)n + (uint64_t)b;
    val[e] = (w ? w[e] : 1.0) * pv;
}

__global__ void run_head_kernel(const uint64_t *__restrict__ key, int32_t m, int32_t *__restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint64_t k = key[i];
    flag[i] = (k != kInvalid && (i == 0 || key[i - 1] != k)) ? 1 : 0;
}

// one thread per run head: sequential sum of the run (ascending original order: the sort is stable).
// MODE 1: emit stage-2 entries  key2 = assign[u]*n + b, value = y * p_u
// MODE 2: emit the unique (a,b) keys and their sums
template <int MODE>
__global__ void run_sum_kernel(const uint64_t *__restrict__ key, const double *__restrict__ val, int32_t m,
                               const int32_t *__restrict__ run_idx, const int32_t *__restrict__ assign,
                               const double *__restrict__ cval, int32_t n, uint64_t *__restrict__ okey,
                               double *__restrict__ oval) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint64_t k = key[i];
    const bool head = k != kInvalid && (i == 0 || key[i - 1] != k);
    if (!head) return;
    double s = val[i];
    for (int j = i + 1; j < m && key[j] == k; ++j) s = s + val[j];
    const int r = run_idx[i];
    if (MODE == 1) {
        const int u = (int)(k / (uint64_t)n), b = (int)(k % (uint64_t)n);
        const double pu = cval[u] * (1.0 / cval[u]);
        okey[r] = (uint64_t)assign[u] * (uint64_t)n + (uint64_t)b;
        oval[r] = s * pu;
    } else {
        okey[r] = k;
        oval[r] = s;
    }
}

__global__ void fill_invalid_kernel(uint64_t *__restrict__ key, double *__restrict__ val, const int32_t *__restrict__ n_valid,
                                    int32_t m) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m || i < n_valid[0]) return;
    key[i] = kInvalid;
    val[i] = 0.0;
}

__device__ __forceinline__ int lower_bound_u64(const uint64_t *a, int n, uint64_t v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// out = (s_ab + s_ba)/2, flag entries that survive (non-zero)
__global__ void symmetrise_kernel(const uint64_t *__restrict__ ukey, const double *__restrict__ usum,
                                  const int32_t *__restrict__ n_unique, int32_t n, int32_t cap, double *__restrict__ sym,
                                  int32_t *__restrict__ keep) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const int cnt = n_unique[0];
    if (i >= cnt) { keep[i] = 0; return; }
    const uint64_t k = ukey[i];
    const uint64_t a = k / (uint64_t)n, b = k % (uint64_t)n;
    const uint64_t tk = b * (uint64_t)n + a;
    const int p = lower_bound_u64(ukey, cnt, tk);
    const double t = (p < cnt && ukey[p] == tk) ? usum[p] : 0.0;
    const double v = (usum[i] + t) / 2.0;
    sym[i] = v;
    keep[i] = v != 0.0 ? 1 : 0;
}

__global__ void lift_emit_kernel(const uint64_t *__restrict__ ukey, const double *__restrict__ sym,
                                 const int32_t *__restrict__ pos, const int32_t *__restrict__ n_unique, int32_t n,
                                 int32_t cap, int32_t *__restrict__ col_c, double *__restrict__ w_c,
                                 int32_t *__restrict__ nnz_c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) nnz_c[0] = pos[cap];
    if (i >= cap || i >= n_unique[0]) return;
    if (pos[i + 1] == pos[i]) return;  // dropped
    col_c[pos[i]] = (int32_t)(ukey[i] % (uint64_t)n);
    w_c[pos[i]] = sym[i];
}

__global__ void lift_rowptr_kernel(const uint64_t *__restrict__ ukey, const int32_t *__restrict__ pos,
                                   const int32_t *__restrict__ n_unique, int32_t n, int32_t *__restrict__ rowptr_c) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a > n) return;
    const int cnt = n_unique[0];
    const int p = lower_bound_u64(ukey, cnt, (uint64_t)a * (uint64_t)n);
    rowptr_c[a] = pos[p];
}

struct LiftLayout {
    size_t k0, k1, v0, v1, flag, counts, sort_tmp, sort_tmp_bytes, scan_tmp, scan_tmp_bytes, total;
};
LiftLayout lift_layout(int64_t nnz) {
    LiftLayout L{};
    const size_t m = (size_t)(nnz > 0 ? nnz : 1);
    size_t o = 0;
    L.k0 = o; o += align_up(m * 8);
    L.k1 = o; o += align_up(m * 8);
    L.v0 = o; o += align_up(m * 8);
    L.v1 = o; o += align_up(m * 8);
    L.flag = o; o += align_up((m + 1) * 4);
    L.counts = o; o += align_up(16);
    size_t t = 0;
    (void)rocprim::radix_sort_pairs(nullptr, t, (uint64_t *)nullptr, (uint64_t *)nullptr, (double *)nullptr,
                                    (double *)nullptr, m, 0, 64, (hipStream_t)0);
    L.sort_tmp_bytes = t;
    L.sort_tmp = o; o += align_up(t);
    size_t t2 = 0;
    (void)rocprim::exclusive_scan(nullptr, t2, (int32_t *)nullptr, (int32_t *)nullptr, 0, m + 1, rocprim::plus<int32_t>(),
                                  (hipStream_t)0);
    L.scan_tmp_bytes = t2;
    L.scan_tmp = o; o += align_up(t2);
    L.total = o;
    return L;
}

// ------------------------------------------------------------------------------------------------
// pooling
// ------------------------------------------------------------------------------------------------
__global__ void iota_kernel(int32_t N, const int32_t *__restrict__ assign, uint32_t *__restrict__ key,
                            int32_t *__restrict__ id) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) { key[i] = (uint32_t)assign[i]; id[i] = i; }
}
__global__ void cluster_offsets_kernel(const uint32_t *__restrict__ skey, int32_t N, int32_t n, int32_t *__restrict__ off) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n) return;
    int lo = 0, hi = N;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (skey[mid] < (uint32_t)c) lo = mid + 1; else hi = mid;
    }
    off[c] = lo;
}

// one wave per (cluster, 64*VEC-column slab): f64 accumulation over members in ascending node order
template <int VEC>
__global__ __launch_bounds__(256) void pool_rows_kernel(const int32_t *__restrict__ off, const int32_t *__restrict__ members,
                                                        const double *__restrict__ cval, int32_t n,
                                                        const float *__restrict__ X, int64_t ldx, int32_t F,
                                                        float *__restrict__ Xc, int64_t ldxc, double *__restrict__ Xc64) {
#pragma clang fp contract(off)
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= n) return;
    const int f0 = (blockIdx.y * 64 + lane) * VEC;
    if (f0 >= F) return;
    const int m0 = off[c], m1 = off[c + 1];
    double acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.0;
    for (int m = m0; m < m1; ++m) {
        const int node = members[m];
        const double v = cval[node];
        const float *src = X + (int64_t)node * ldx + f0;
        float x[VEC];
        if (VEC == 4) {
            const float4 q = *reinterpret_cast<const float4 *>(src);
            x[0] = q.x; x[1 % VEC] = q.y; x[2 % VEC] = q.z; x[3 % VEC] = q.w;
        } else {
            x[0] = src[0];
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) { const double prod = v * (double)x[i]; acc[i] = acc[i] + prod; }
    }
    float *dst = Xc + (int64_t)c * ldxc + f0;
    if (VEC == 4) {
        *reinterpret_cast<float4 *>(dst) = make_float4((float)acc[0], (float)acc[1 % VEC], (float)acc[2 % VEC], (float)acc[3 % VEC]);
    } else {
        dst[0] = (float)acc[0];
    }
    if (Xc64) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) Xc64[(int64_t)c * F + f0 + i] = acc[i];
    }
}

// f32 segment sum (one wave per (segment, 64*VEC-column slab)), members in the given order.  The member ids of
// a segment are fetched with one lane-parallel load and broadcast by v_readlane; rows are loaded four at a time
// (always four loads: missing ones re-read the first row with weight 0) so that the loads overlap instead of
// forming a dependent id -> row chain per member.
template <int VEC>
__global__ __launch_bounds__(256) void segment_sum_kernel(const int32_t *__restrict__ off, const int32_t *__restrict__ members,
                                                          int32_t n_seg, const float *__restrict__ X, int64_t ldx, int32_t F,
                                                          float *__restrict__ out, int64_t ldo, int32_t n_slabs) {
    // 1-D grid, column slab fastest: both 1-KiB halves of every 2-KiB row are in flight together (a slab-major order
    // streams the first half of every row, then the second: half the channels idle, as measured on the SpMM)
    const int slab = blockIdx.x % n_slabs;
    const int sgm = (blockIdx.x / n_slabs) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (sgm >= n_seg) return;
    const int f0 = (slab * 64 + lane) * VEC;
    const bool live = f0 + VEC <= F;
    const float *Xs = X + (live ? f0 : (F >= VEC ? F - VEC : 0));
    const int m0 = __builtin_amdgcn_readfirstlane(off[sgm]), m1 = __builtin_amdgcn_readfirstlane(off[sgm + 1]);
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    for (int base = m0; base < m1; base += 64) {
        const int cnt = min(64, m1 - base);
        const int my = lane < cnt ? members[base + lane] : 0;
        for (int k = 0; k < cnt; k += 4) {
            float v[4][VEC];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool has = k + j < cnt;
                const int node = __builtin_amdgcn_readlane(my, has ? k + j : k);
                const float *src = Xs + (int64_t)node * ldx;
                if (VEC == 4) {
                    const float4 q = *reinterpret_cast<const float4 *>(src);
                    v[j][0] = q.x; v[j][1 % VEC] = q.y; v[j][2 % VEC] = q.z; v[j][3 % VEC] = q.w;
                } else {
                    v[j][0] = src[0];
                }
                if (!has) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) v[j][i] = 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] += v[j][i];
        }
    }
    if (!live) return;
    float *dst = out + (int64_t)sgm * ldo + f0;
    if (VEC == 4) *reinterpret_cast<float4 *>(dst) = make_float4(acc[0], acc[1 % VEC], acc[2 % VEC], acc[3 % VEC]);
    else dst[0] = acc[0];
}

struct PoolLayout {
    size_t key_in, key_out, id_in, members, off, sort_tmp, sort_tmp_bytes, total;
};
PoolLayout pool_layout(int32_t N, int32_t n) {
    PoolLayout L{};
    const size_t m = (size_t)(N > 0 ? N : 1);
    size_t o = 0;
    L.key_in = o; o += align_up(m * 4);
    L.key_out = o; o += align_up(m * 4);
    L.id_in = o; o += align_up(m * 4);
    L.members = o; o += align_up(m * 4);
    L.off = o; o += align_up(((size_t)(n > 0 ? n : 0) + 1) * 4);
    size_t t = 0;
    (void)rocprim::radix_sort_pairs(nullptr, t, (uint32_t *)nullptr, (uint32_t *)nullptr, (int32_t *)nullptr,
                                    (int32_t *)nullptr, m, 0, 32, (hipStream_t)0);
    L.sort_tmp_bytes = t;
    L.sort_tmp = o; o += align_up(t);
    L.total = o;
    return L;
}

}  // namespace

extern "C" size_t fitgnn_lift_adjacency_workspace_bytes(int32_t N, int64_t nnz, int32_t n) {
    (void)N; (void)n;
    if (nnz < 0) return 0;
    return lift_layout(nnz).total;
}

extern "C" int fitgnn_lift_adjacency(int32_t N, const int32_t *rowptr, const int32_t *col, const double *w,
                                     const int32_t *assign, const double *cval, int32_t n, int32_t *rowptr_c,
                                     int32_t *col_c, double *w_c, int32_t *nnz_c, void *work, size_t work_bytes,
                                     void *stream) {
    if (N < 0 || n < 0 || !rowptr_c || !nnz_c) return FITGNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (N == 0 || n == 0) {
        FITGNN_RETURN_IF_HIP(hipMemsetAsync(rowptr_c, 0, ((size_t)n + 1) * sizeof(int32_t), s));
        return (int)hipMemsetAsync(nnz_c, 0, sizeof(int32_t), s);
    }
    if (!rowptr || !col || !assign || !cval || !col_c || !w_c || !work) return FITGNN_E_BADARG;
    int32_t nnz = 0;  // 4-byte read-back: sizes every launch below
    FITGNN_RETURN_IF_HIP(hipMemcpyAsync(&nnz, rowptr + N, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    FITGNN_RETURN_IF_HIP(hipStreamSynchronize(s));
    if (nnz == 0) {
        FITGNN_RETURN_IF_HIP(hipMemsetAsync(rowptr_c, 0, ((size_t)n + 1) * sizeof(int32_t), s));
        return (int)hipMemsetAsync(nnz_c, 0, sizeof(int32_t), s);
    }
    const LiftLayout L = lift_layout(nnz);
    if (work_bytes < L.total) return FITGNN_E_WORKSPACE;
    char *base = (char *)work;
    uint64_t *k0 = (uint64_t *)(base + L.k0), *k1 = (uint64_t *)(base + L.k1);
    double *v0 = (double *)(base + L.v0), *v1 = (double *)(base + L.v1);
    int32_t *flag = (int32_t *)(base + L.flag);
    void *sort_tmp = base + L.sort_tmp, *scan_tmp = base + L.scan_tmp;
    size_t st = L.sort_tmp_bytes, ct = L.scan_tmp_bytes;
    const dim3 g = blocks_for(nnz), b(256);

    // ---- stage 1: y[u][b] = sum_v w_uv p_v  (W . Pinv) ----
    hipLaunchKernelGGL(lift_stage1_kernel, g, b, 0, s, N, rowptr, col, w, assign, cval, n, nnz, k0, v0);
    FITGNN_RETURN_IF_HIP(rocprim::radix_sort_pairs(sort_tmp, st, k0, k1, v0, v1, (size_t)nnz, 0, 64, s));
    hipLaunchKernelGGL(run_head_kernel, g, b, 0, s, k1, nnz, flag);
    FITGNN_RETURN_IF_HIP(hipMemsetAsync(flag + nnz, 0, sizeof(int32_t), s));
    FITGNN_RETURN_IF_HIP(rocprim::exclusive_scan(scan_tmp, ct, flag, flag, 0, (size_t)nnz + 1, rocprim::plus<int32_t>(), s));
    hipLaunchKernelGGL(run_sum_kernel<1>, g, b, 0, s, k1, v1, nnz, flag, assign, cval, n, k0, v0);
    hipLaunchKernelGGL(fill_invalid_kernel, g, b, 0, s, k0, v0, flag + nnz, nnz);
    // ---- stage 2: s[a][b] = sum_u y[u][b] p_u  (Pinv^T . y) ----
    st = L.sort_tmp_bytes;
    FITGNN_RETURN_IF_HIP(rocprim::radix_sort_pairs(sort_tmp, st, k0, k1, v0, v1, (size_t)nnz, 0, 64, s));
    hipLaunchKernelGGL(run_head_kernel, g, b, 0, s, k1, nnz, flag);
    FITGNN_RETURN_IF_HIP(hipMemsetAsync(flag + nnz, 0, sizeof(int32_t), s));
    ct = L.scan_tmp_bytes;
    FITGNN_RETURN_IF_HIP(rocprim::exclusive_scan(scan_tmp, ct, flag, flag, 0, (size_t)nnz + 1, rocprim::plus<int32_t>(), s));
    hipLaunchKernelGGL(run_sum_kernel<2>, g, b, 0, s, k1, v1, nnz, flag, assign, cval, n, k0, v0);
    // unique count -> counts[0]
    int32_t *counts = (int32_t *)(base + L.counts);
    FITGNN_RETURN_IF_HIP(hipMemcpyAsync(counts, flag + nnz, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    // ---- symmetrise, drop zeros, emit CSR ----
    hipLaunchKernelGGL(symmetrise_kernel, g, b, 0, s, k0, v0, counts, n, nnz, v1, flag);
    FITGNN_RETURN_IF_HIP(hipMemsetAsync(flag + nnz, 0, sizeof(int32_t), s));
    ct = L.scan_tmp_bytes;
    FITGNN_RETURN_IF_HIP(rocprim::exclusive_scan(scan_tmp, ct, flag, flag, 0, (size_t)nnz + 1, rocprim::plus<int32_t>(), s));
    hipLaunchKernelGGL(lift_emit_kernel, g, b, 0, s, k0, v1, flag, counts, n, nnz, col_c, w_c, nnz_c);
    hipLaunchKernelGGL(lift_rowptr_kernel, blocks_for((int64_t)n + 1), b, 0, s, k0, flag, counts, n, rowptr_c);
    return (int)hipGetLastError();
}

extern "C" size_t fitgnn_pool_rows_workspace_bytes(int32_t N, int32_t n) {
    if (N < 0 || n < 0) return 0;
    return pool_layout(N, n).total;
}

extern "C" int fitgnn_pool_rows_f32(const int32_t *assign, const double *cval, int32_t N, int32_t n, const float *X,
                                    int64_t ldx, int32_t F, float *Xc, int64_t ldxc, double *Xc64, void *work,
                                    size_t work_bytes, void *stream) {
    if (N < 0 || n < 0 || F < 0) return FITGNN_E_BADARG;
    if (n == 0 || F == 0) return 0;
    if (!assign || !cval || !X || !Xc || !work || ldx < F || ldxc < F) return FITGNN_E_BADARG;
    const PoolLayout L = pool_layout(N, n);
    if (work_bytes < L.total) return FITGNN_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char *base = (char *)work;
    uint32_t *key_in = (uint32_t *)(base + L.key_in), *key_out = (uint32_t *)(base + L.key_out);
    int32_t *id_in = (int32_t *)(base + L.id_in), *members = (int32_t *)(base + L.members);
    int32_t *off = (int32_t *)(base + L.off);
    if (N > 0) {
        hipLaunchKernelGGL(iota_kernel, blocks_for(N), dim3(256), 0, s, N, assign, key_in, id_in);
        size_t st = L.sort_tmp_bytes;
        // stable: members of a cluster stay in ascending node order (the order scipy's csc product visits them)
        FITGNN_RETURN_IF_HIP(rocprim::radix_sort_pairs((void *)(base + L.sort_tmp), st, key_in, key_out, id_in, members,
                                                       (size_t)N, 0, 32, s));
    }
    hipLaunchKernelGGL(cluster_offsets_kernel, blocks_for((int64_t)n + 1), dim3(256), 0, s, key_out, N, n, off);
    const bool vec = (F % 4 == 0) && (ldx % 4 == 0) && (ldxc % 4 == 0) && ((((uintptr_t)X | (uintptr_t)Xc) % 16) == 0);
    if (vec) {
        dim3 grid((n + 3) / 4, (F + 255) / 256);
        hipLaunchKernelGGL(pool_rows_kernel<4>, grid, dim3(256), 0, s, off, members, cval, n, X, ldx, F, Xc, ldxc, Xc64);
    } else {
        dim3 grid((n + 3) / 4, (F + 63) / 64);
        hipLaunchKernelGGL(pool_rows_kernel<1>, grid, dim3(256), 0, s, off, members, cval, n, X, ldx, F, Xc, ldxc, Xc64);
    }
    return (int)hipGetLastError();
}

// ---- graph-level pooling on sorted segments (global_mean_pool / global_max_pool of network.py:93,131,164,202) ----------------------
// out[s][c] = max over the segment's member rows of X[member][c], arg[s][c] = the member row that holds it (the first one on a tie;
// an empty segment gives -inf / -1).  One wave per (segment, 256-column slab), 16-byte row accesses, members broadcast from registers.
__global__ __launch_bounds__(256) void segment_max_kernel(const int32_t *__restrict__ off, const int32_t *__restrict__ members, int32_t n_seg,
                                                          const float *__restrict__ X, int64_t ldx, int32_t F, float *__restrict__ out,
                                                          int32_t *__restrict__ arg, int32_t n_slabs) {
    const int slab = blockIdx.x % n_slabs;
    const int sgm = (blockIdx.x / n_slabs) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (sgm >= n_seg) return;
    const int f0 = (slab * 64 + lane) * 4;
    if (f0 >= F) return;
    const int m0 = off[sgm], m1 = off[sgm + 1];
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int who[4] = {-1, -1, -1, -1};
    for (int m = m0; m < m1; ++m) {
        const int node = members ? members[m] : m;
        const float *src = X + (int64_t)node * ldx + f0;
        float v[4];
        if (f0 + 4 <= F) { const float4 q = *reinterpret_cast<const float4 *>(src); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
        else { for (int i = 0; i < 4; ++i) v[i] = f0 + i < F ? src[i] : -INFINITY; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (v[i] > best[i] || who[i] < 0) { best[i] = v[i]; who[i] = node; }
    }
    for (int i = 0; i < 4 && f0 + i < F; ++i) {
        out[(int64_t)sgm * F + f0 + i] = best[i];
        arg[(int64_t)sgm * F + f0 + i] = who[i];
    }
}

// dst[r] = scale[seg] * src[seg] for seg = seg_of_row[r] >= 0, a row of zeros otherwise: the backward of a mean / sum pool over row
// subsets (every row of dst is written: no separate zero fill)
__global__ __launch_bounds__(256) void segment_expand_kernel(const float *__restrict__ src, const int32_t *__restrict__ seg_of_row,
                                                             const float *__restrict__ scale, int64_t n_rows, int32_t F4,
                                                             float4 *__restrict__ dst) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_rows * F4) return;
    const int64_t r = t / F4;
    const int c = (int)(t - r * F4);
    const int sgm = seg_of_row[r];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sgm >= 0) {
        v = reinterpret_cast<const float4 *>(src)[(int64_t)sgm * F4 + c];
        if (scale) { const float w = scale[sgm]; v.x *= w; v.y *= w; v.z *= w; v.w *= w; }
    }
    dst[t] = v;
}

// Mean pool over row subsets + a narrow head in ONE launch (Regress_graph_gs / _gc: lt1(global_mean_pool(x[mask])), network.py:164-166,
// :200-204 -- on a 128-graph batch four launches of a few microseconds each: gather-and-sum, 1 / count, a [128 x 512] @ [512 x 1]
// library product, the bias).  One workgroup per graph: its 256 threads are PH = 256 / (F / 4) row phases x F / 4 column groups; a
// phase sums every PH-th member row (four rows in flight), the phases are added in ascending order through LDS, phase 0 scales by
// 1 / count, stores the pooled row (the backward's operand) and reduces the C dot products with the head's rows by a fixed tree.
constexpr int kPoolHeadMaxC = 8;
__global__ __launch_bounds__(256) void pool_head_kernel(const int32_t *__restrict__ off, const int32_t *__restrict__ members,
                                                        const float *__restrict__ X, int64_t ldx, int32_t F,
                                                        const float *__restrict__ inv_cnt, const float *__restrict__ W,
                                                        const float *__restrict__ b, int32_t C, float *__restrict__ pooled,
                                                        float *__restrict__ y) {
    __shared__ float4 s_red[256];
    __shared__ float s_dot[256];
    const int sgm = blockIdx.x;
    const int F4 = F >> 2, PH = 256 / F4;
    const int cg = threadIdx.x % F4, ph = threadIdx.x / F4;
    const int m0 = off[sgm], m1 = off[sgm + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = m0 + ph; base < m1; base += 4 * PH) {
        int node[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) node[u] = members[min(base + u * PH, m1 - 1)];
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4 *>(X + (int64_t)node[u] * ldx + 4 * cg);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (base + u * PH < m1) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    s_red[threadIdx.x] = acc;
    __syncthreads();
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ph == 0) {
        for (int q = 0; q < PH; ++q) {
            const float4 t = s_red[q * F4 + cg];
            tot.x += t.x; tot.y += t.y; tot.z += t.z; tot.w += t.w;
        }
        const float w = inv_cnt[sgm];
        tot.x *= w; tot.y *= w; tot.z *= w; tot.w *= w;
        *reinterpret_cast<float4 *>(pooled + (int64_t)sgm * F + 4 * cg) = tot;
    }
    for (int c = 0; c < C; ++c) {
        __syncthreads();
        if (ph == 0) {
            const float4 wv = *reinterpret_cast<const float4 *>(W + (int64_t)c * F + 4 * cg);
            s_dot[cg] = fmaf(tot.w, wv.w, fmaf(tot.z, wv.z, fmaf(tot.y, wv.y, tot.x * wv.x)));
        }
        __syncthreads();
        for (int w2 = F4 >> 1; w2 >= 1; w2 >>= 1) {
            if ((int)threadIdx.x < w2) s_dot[threadIdx.x] += s_dot[threadIdx.x + w2];
            __syncthreads();
        }
        if (threadIdx.x == 0) y[(int64_t)sgm * C + c] = s_dot[0] + (b ? b[c] : 0.f);
    }
}

// ... and its backward in one launch: blocks [0, nb_dx) write every row of dx (dx[r] = (1 / count) sum_c dy[s][c] W[c] for a pooled row
// of graph s, zeros for the others); the blocks after them form dW[c][f] = sum_s dy[s][c] pooled[s][f] and db[c] = sum_s dy[s][c]
// (ascending s: reproducible).
__global__ __launch_bounds__(256) void pool_head_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ W, int32_t C,
                                                            const float *__restrict__ pooled, const int32_t *__restrict__ seg_of_row,
                                                            const float *__restrict__ inv_cnt, int64_t n_rows, int32_t n_seg, int32_t F,
                                                            int32_t nb_dx, float4 *__restrict__ dx, float *__restrict__ dW,
                                                            float *__restrict__ db) {
    const int F4 = F >> 2;
    if ((int)blockIdx.x < nb_dx) {
        const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (t >= n_rows * F4) return;
        const int64_t r = t / F4;
        const int c4 = (int)(t - r * F4);
        const int sgm = seg_of_row[r];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sgm >= 0) {
            for (int c = 0; c < C; ++c) {
                const float d = dy[(int64_t)sgm * C + c];
                const float4 wv = *reinterpret_cast<const float4 *>(W + (int64_t)c * F + 4 * c4);
                v.x = fmaf(d, wv.x, v.x); v.y = fmaf(d, wv.y, v.y); v.z = fmaf(d, wv.z, v.z); v.w = fmaf(d, wv.w, v.w);
            }
            const float w = inv_cnt[sgm];
            v.x *= w; v.y *= w; v.z *= w; v.w *= w;
        }
        dx[t] = v;
        return;
    }
    const int q = ((int)blockIdx.x - nb_dx) * 256 + threadIdx.x;
    if (q < C * F) {
        if (!dW) return;
        const int c = q / F, f = q - c * F;
        float acc = 0.f;
#pragma unroll 8
        for (int sg = 0; sg < n_seg; ++sg) acc = fmaf(dy[(int64_t)sg * C + c], pooled[(int64_t)sg * F + f], acc);
        dW[q] = acc;
    } else if (q < C * F + C) {
        if (!db) return;
        const int c = q - C * F;
        float acc = 0.f;
        for (int sg = 0; sg < n_seg; ++sg) acc += dy[(int64_t)sg * C + c];
        db[c] = acc;
    }
}

// dst[arg[s][c]][c] += g[s][c] (dst zeroed by the caller; every (row, column) is the maximum of at most one segment when the segments
// are disjoint, so plain read-modify-write stores do not collide)
__global__ __launch_bounds__(256) void segment_max_bwd_kernel(const float *__restrict__ g, const int32_t *__restrict__ arg, int64_t n,
                                                              int32_t F, float *__restrict__ dst, int64_t ldd) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int a = arg[t];
    if (a >= 0) dst[(int64_t)a * ldd + (t % F)] = g[t];
}

extern "C" int fitgnn_segment_max_f32(const int32_t *seg_off, const int32_t *members, int32_t n_seg, const float *X, int64_t ldx,
                                      int32_t F, float *out, int32_t *arg, void *stream) {
    if (n_seg < 0 || F < 0 || ldx < F) return FITGNN_E_BADARG;
    if (n_seg == 0 || F == 0) return 0;
    if (!seg_off || !X || !out || !arg) return FITGNN_E_BADARG;
    if ((ldx % 4) != 0 || ((uintptr_t)X % 16) != 0) return FITGNN_E_ALIGN;
    const int n_slabs = (F + 255) / 256;
    hipLaunchKernelGGL(segment_max_kernel, dim3((unsigned)((n_seg + 3) / 4) * n_slabs), dim3(256), 0, (hipStream_t)stream, seg_off, members,
                       n_seg, X, ldx, F, out, arg, n_slabs);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_segment_max_bwd_f32(const float *g, const int32_t *arg, int32_t n_seg, int32_t F, float *dst, int64_t ldd,
                                          void *stream) {
    if (n_seg < 0 || F < 0 || ldd < F) return FITGNN_E_BADARG;
    if (n_seg == 0 || F == 0) return 0;
    if (!g || !arg || !dst) return FITGNN_E_BADARG;
    const int64_t n = (int64_t)n_seg * F;
    hipLaunchKernelGGL(segment_max_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, arg, n, F, dst, ldd);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_segment_expand_f32(const float *src, const int32_t *seg_of_row, const float *scale, int64_t n_rows, int32_t F,
                                         float *dst, void *stream) {
    if (n_rows < 0 || F < 0 || (F % 4) != 0) return FITGNN_E_BADARG;
    if (n_rows == 0 || F == 0) return 0;
    if (!src || !seg_of_row || !dst) return FITGNN_E_BADARG;
    if ((((uintptr_t)src | (uintptr_t)dst) % 16) != 0) return FITGNN_E_ALIGN;
    const int64_t n = n_rows * (F / 4);
    hipLaunchKernelGGL(segment_expand_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, seg_of_row, scale, n_rows,
                       F / 4, (float4 *)dst);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_pool_head_supported(int32_t F, int32_t C) {
    return (F >= 4 && (F % 4) == 0 && (F / 4) <= 256 && (256 % (F / 4)) == 0 && C >= 1 && C <= kPoolHeadMaxC) ? 1 : 0;
}

extern "C" int fitgnn_pool_head_f32(const int32_t *seg_off, const int32_t *members, int32_t n_seg, const float *X, int64_t ldx, int32_t F,
                                    const float *inv_cnt, const float *W, const float *b, int32_t C, float *pooled, float *y,
                                    void *stream) {
    if (n_seg < 0 || !fitgnn_pool_head_supported(F, C) || ldx < F || (ldx % 4) != 0) return FITGNN_E_BADARG;
    if (n_seg == 0) return 0;
    if (!seg_off || !members || !X || !inv_cnt || !W || !pooled || !y) return FITGNN_E_BADARG;
    if ((((uintptr_t)X | (uintptr_t)W | (uintptr_t)pooled) % 16) != 0) return FITGNN_E_ALIGN;
    hipLaunchKernelGGL(pool_head_kernel, dim3((unsigned)n_seg), dim3(256), 0, (hipStream_t)stream, seg_off, members, X, ldx, F, inv_cnt, W, b,
                       C, pooled, y);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_pool_head_bwd_f32(const float *dy, const float *W, int32_t C, const float *pooled, const int32_t *seg_of_row,
                                        const float *inv_cnt, int64_t n_rows, int32_t n_seg, int32_t F, float *dx, float *dW, float *db,
                                        void *stream) {
    if (n_rows < 0 || n_seg < 0 || !fitgnn_pool_head_supported(F, C)) return FITGNN_E_BADARG;
    if (n_rows == 0 && n_seg == 0) return 0;
    if (!dy || !W || !pooled || !seg_of_row || !inv_cnt || (n_rows > 0 && !dx)) return FITGNN_E_BADARG;
    if ((((uintptr_t)W | (uintptr_t)dx) % 16) != 0) return FITGNN_E_ALIGN;
    const int64_t n = n_rows * (F / 4);
    const int nb_dx = (int)((n + 255) / 256);
    const int nb_w = (dW || db) ? (C * F + C + 255) / 256 : 0;
    hipLaunchKernelGGL(pool_head_bwd_kernel, dim3((unsigned)(nb_dx + nb_w)), dim3(256), 0, (hipStream_t)stream, dy, W, C, pooled, seg_of_row,
                       inv_cnt, n_rows, n_seg, F, nb_dx, (float4 *)dx, dW, db);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_segment_sum_f32(const int32_t *seg_off, const int32_t *members, int32_t n_seg, const float *X,
                                      int64_t ldx, int32_t F, float *out, int64_t ldo, void *stream) {
    if (n_seg < 0 || F < 0 || ldx < F || ldo < F) return FITGNN_E_BADARG;
    if (n_seg == 0 || F == 0) return 0;
    if (!seg_off || !X || !out) return FITGNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (F % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) && ((((uintptr_t)X | (uintptr_t)out) % 16) == 0);
    if (vec) {
        const int n_slabs = (F + 255) / 256;
        dim3 grid((unsigned)((n_seg + 3) / 4) * n_slabs);
        hipLaunchKernelGGL(segment_sum_kernel<4>, grid, dim3(256), 0, s, seg_off, members, n_seg, X, ldx, F, out, ldo, n_slabs);
    } else {
        const int n_slabs = (F + 63) / 64;
        dim3 grid((unsigned)((n_seg + 3) / 4) * n_slabs);
        hipLaunchKernelGGL(segment_sum_kernel<1>, grid, dim3(256), 0, s, seg_off, members, n_seg, X, ldx, F, out, ldo, n_slabs);
    }
    return (int)hipGetLastError();
}
