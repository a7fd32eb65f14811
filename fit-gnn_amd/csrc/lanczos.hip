// lanczos.hip -- the inner step of the spectral prelude on the device, float64, gfx950 only.
//
// Reference: graph_coarsening/coarsening_utils.py:83-90 hands T = 2 max(dw) I - L to ARPACK's eigsh(k = 10, which = 'LM',
// tol = 1e-5): an implicitly restarted Lanczos iteration whose cost is the matvec plus the (re)orthogonalisation of the new
// vector against the basis (42 % of the reference's coarsening time).  fitgnn_amd.coarsening.lanczos_smallest runs
// thick-restart Lanczos with full two-pass reorthogonalisation; round 2 issued one step as one sparse product plus four f64
// `gemv`s of the library over a row-major basis (2.58 ms each at N = 165 000: 62 % of all GPU time of a bench run).  Here the
// step is five launches over a COLUMN-major basis V [m + 1][N] (every basis vector contiguous):
//     w = T v_j                                         fitgnn_lanczos_spmv_f64      (CSR, 8 lanes per row)
//     h  = V^T w                                        fitgnn_lanczos_project_f64 (no subtraction) + fitgnn_lanczos_reduce_f64
//     w -= V h ;  h2 = V^T w                            project + reduce
//     w -= V h2 ; |w|^2                                 project + reduce
//     beta = |w| ; v_{j+1} = w / beta ; H[:, j] = h + h2, H[j+1, j] = beta       fitgnn_lanczos_finish_f64
// One projection pass reads V once from HBM (rows on lanes: coalesced for every column); the dot products leave a workgroup as one
// partial row per workgroup and are folded in a fixed order by a one-workgroup reduce launch -- no atomics: the result is
// reproducible.  (The first version let every workgroup of the NEXT pass fold the previous pass's partial rows itself: 323
// dependent loads in front of each workgroup, 87 us per pass for 80 MB.)  A restart's basis rotation is fitgnn_lanczos_rotate_f64;
// the m x m projected eigenproblem is solved on the host (60 x 60).
#include "common.h"
#include "fitgnn_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxCols = 128;   // basis vectors a projection pass takes (the solver uses m <= 100)
constexpr int kRowsPerThread = 2;   // 512 rows per workgroup: 323 workgroups at N = 165 000

__device__ __forceinline__ double wave_sum(double v) {   // fixed butterfly: every lane ends with the same sum
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// y = alpha (A x) + beta x for a CSR matrix in f64 (alpha = -1, beta = 2 max(dw), A = L: the shifted operator T = beta I - L the
// reference hands to ARPACK, coarsening_utils.py:83-88, without building it); 8 lanes share a row (a Laplacian row of the S-products
// graph holds ~51 entries, PubMed's ~5), their partial sums are added in a fixed order.
__global__ __launch_bounds__(kThreads) void lanczos_spmv_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                               const double *__restrict__ val, const double *__restrict__ x,
                                                               double *__restrict__ y, int n, double alpha, double beta) {
    const int g = (blockIdx.x * kThreads + threadIdx.x) >> 3, l = threadIdx.x & 7;
    double s = 0.0;
    if (g < n) {
        const int e0 = rowptr[g], e1 = rowptr[g + 1];
        for (int e = e0 + l; e < e1; e += 8) s += val[e] * x[col[e]];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (g < n && l == 0) y[g] = alpha * s + beta * x[g];
}

// One projection pass over the rows [b * rows_per_block, ...) of workgroup b:
//   w[i] -= sum_c V[c][i] h_in[c]         (c ascending; h_in NULL: no subtraction)
//   part_out[b][c] = sum_i V[c][i] w[i]   (c < ncol)   and   part_out[b][ncol] = sum_i w[i]^2     (row stride ncol + 1)
__global__ __launch_bounds__(kThreads) void lanczos_project_kernel(const double *__restrict__ V, int64_t ldv, int ncol, double *__restrict__ w,
                                                                  int n, int rows_per_block, const double *__restrict__ h_in,
                                                                  double *__restrict__ part_out) {
    __shared__ double s_h[kMaxCols];
    __shared__ double s_red[4][kMaxCols + 1];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (h_in) {
        if (t < ncol) s_h[t] = h_in[t];
        __syncthreads();
    }
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(n, r0 + rows_per_block);
    double wi[kRowsPerThread];
    double nrm = 0.0;
#pragma unroll
    for (int k = 0; k < kRowsPerThread; ++k) {
        const int i = r0 + t + k * kThreads;
        wi[k] = 0.0;
        if (i < r1) {
            double v = w[i];
            if (h_in) {
                double s = 0.0;
                for (int c = 0; c < ncol; ++c) s += V[(int64_t)c * ldv + i] * s_h[c];
                v -= s;
                w[i] = v;
            }
            wi[k] = v;
            nrm += v * v;
        }
    }
    for (int c = 0; c < ncol; ++c) {
        double p = 0.0;
#pragma unroll
        for (int k = 0; k < kRowsPerThread; ++k) {
            const int i = r0 + t + k * kThreads;
            if (i < r1) p += V[(int64_t)c * ldv + i] * wi[k];
        }
        p = wave_sum(p);
        if (lane == 0) s_red[wave][c] = p;
    }
    nrm = wave_sum(nrm);
    if (lane == 0) s_red[wave][ncol] = nrm;
    __syncthreads();
    if (t <= ncol) part_out[(int64_t)blockIdx.x * (ncol + 1) + t] = ((s_red[0][t] + s_red[1][t]) + s_red[2][t]) + s_red[3][t];
}

// out[c] = sum over the n_part partial rows of part[.][c], c <= ncol (the last column is |w|^2): four interleaved running sums per
// column, combined in a fixed order.  One workgroup: the partial rows are a few hundred.
__global__ __launch_bounds__(kThreads) void lanczos_reduce_kernel(const double *__restrict__ part, int n_part, int ncol1, double *__restrict__ out) {
    __shared__ double s_q[4][64];
    const int t = threadIdx.x, c64 = t & 63, r = t >> 6;
    for (int c0 = 0; c0 < ncol1; c0 += 64) {
        const int c = c0 + c64;
        double s = 0.0;
        if (c < ncol1)
            for (int b = r; b < n_part; b += 4) s += part[(int64_t)b * ncol1 + c];
        s_q[r][c64] = s;
        __syncthreads();
        if (r == 0 && c < ncol1) out[c] = ((s_q[0][c64] + s_q[1][c64]) + s_q[2][c64]) + s_q[3][c64];
        __syncthreads();
    }
}

// beta = sqrt(hc[ncol]) (the squared norm left by the last pass); V[j + 1] = w / max(beta, tiny); workgroup 0 also writes column j
// of the projected matrix: H[c][j] = ha[c] + hb[c] (the two passes' coefficients), H[j + 1][j] = beta.
__global__ __launch_bounds__(kThreads) void lanczos_finish_kernel(double *__restrict__ V, int64_t ldv, int j, const double *__restrict__ w, int n,
                                                                 const double *__restrict__ ha, const double *__restrict__ hb,
                                                                 const double *__restrict__ hc, double *__restrict__ Hm, int ldh) {
    const int t = threadIdx.x;
    const int ncol = j + 1;
    const double beta = sqrt(hc[ncol]);
    const double inv = 1.0 / fmax(beta, 1e-300);
    if (blockIdx.x == 0) {
        if (t < ncol) Hm[(int64_t)t * ldh + j] = ha[t] + hb[t];
        if (t == 0) Hm[(int64_t)(j + 1) * ldh + j] = beta;
    }
    for (int64_t i = (int64_t)blockIdx.x * kThreads + t; i < n; i += (int64_t)gridDim.x * kThreads) V[(int64_t)(j + 1) * ldv + i] = w[i] * inv;
}

// Basis rotation of a restart / of the final Ritz vectors: out[c][i] = sum_{j < m} S[j][c] V[j][i], c < nk <= 16 (j ascending).
constexpr int kMaxRot = 16;
__global__ __launch_bounds__(kThreads) void lanczos_rotate_kernel(const double *__restrict__ V, int64_t ldv, int m, const double *__restrict__ S,
                                                                 int nk, double *__restrict__ out, int64_t ldo, int n) {
    __shared__ double s_S[kMaxCols * kMaxRot];
    for (int q = threadIdx.x; q < m * nk; q += kThreads) s_S[q] = S[q];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    double acc[kMaxRot];
#pragma unroll
    for (int c = 0; c < kMaxRot; ++c) acc[c] = 0.0;
    for (int jj = 0; jj < m; ++jj) {
        const double v = V[(int64_t)jj * ldv + i];
#pragma unroll
        for (int c = 0; c < kMaxRot; ++c)
            if (c < nk) acc[c] += s_S[jj * nk + c] * v;
    }
#pragma unroll
    for (int c = 0; c < kMaxRot; ++c)
        if (c < nk) out[(int64_t)c * ldo + i] = acc[c];
}

inline int project_blocks(int n) {
    const int per = kThreads * kRowsPerThread;
    return (n + per - 1) / per;
}

}  // namespace

extern "C" int32_t fitgnn_lanczos_parts(int32_t n) { return n <= 0 ? 0 : project_blocks(n); }

extern "C" int fitgnn_lanczos_spmv_f64(const int32_t *rowptr, const int32_t *col, const double *val, const double *x, double *y, int32_t n,
                                       double alpha, double beta, void *stream) {
    if (n < 0) return FITGNN_E_BADARG;
    if (n == 0) return 0;
    if (!rowptr || !x || !y) return FITGNN_E_BADARG;
    const int64_t threads = (int64_t)n * 8;
    hipLaunchKernelGGL(lanczos_spmv_kernel, dim3((unsigned)((threads + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream, rowptr,
                       col, val, x, y, n, alpha, beta);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_lanczos_project_f64(const double *V, int64_t ldv, int32_t ncol, double *w, int32_t n, const double *h_in,
                                          double *part_out, void *stream) {
    if (n < 0 || ncol < 1 || ncol > kMaxCols || ldv < n) return FITGNN_E_BADARG;
    if (n == 0) return 0;
    if (!V || !w || !part_out) return FITGNN_E_BADARG;
    const int nb = project_blocks(n);
    hipLaunchKernelGGL(lanczos_project_kernel, dim3((unsigned)nb), dim3(kThreads), 0, (hipStream_t)stream, V, ldv, ncol, w, n,
                       kThreads * kRowsPerThread, h_in, part_out);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_lanczos_reduce_f64(const double *part, int32_t n_part, int32_t ncol1, double *out, void *stream) {
    if (n_part < 0 || ncol1 < 1 || ncol1 > kMaxCols + 1) return FITGNN_E_BADARG;
    if (!part || !out) return FITGNN_E_BADARG;
    hipLaunchKernelGGL(lanczos_reduce_kernel, dim3(1), dim3(kThreads), 0, (hipStream_t)stream, part, n_part, ncol1, out);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_lanczos_finish_f64(double *V, int64_t ldv, int32_t j, const double *w, int32_t n, const double *ha, const double *hb,
                                         const double *hc, double *H, int32_t ldh, void *stream) {
    if (n < 0 || j < 0 || j + 1 > kMaxCols || ldv < n || ldh < j + 1) return FITGNN_E_BADARG;
    if (n == 0) return 0;
    if (!V || !w || !ha || !hb || !hc || !H) return FITGNN_E_BADARG;
    const int grid = (int)fmin(1024.0, (double)((n + kThreads - 1) / kThreads));
    hipLaunchKernelGGL(lanczos_finish_kernel, dim3((unsigned)(grid < 1 ? 1 : grid)), dim3(kThreads), 0, (hipStream_t)stream, V, ldv, j, w, n,
                       ha, hb, hc, H, ldh);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_lanczos_rotate_f64(const double *V, int64_t ldv, int32_t m, const double *S, int32_t nk, double *out, int64_t ldo,
                                         int32_t n, void *stream) {
    if (n < 0 || m < 1 || m > kMaxCols || nk < 1 || nk > kMaxRot || ldv < n || ldo < n) return FITGNN_E_BADARG;
    if (n == 0) return 0;
    if (!V || !S || !out) return FITGNN_E_BADARG;
    hipLaunchKernelGGL(lanczos_rotate_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream, V, ldv, m, S,
                       nk, out, ldo, n);
    return (int)hipGetLastError();
}
