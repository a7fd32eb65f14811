// lanczos.hip -- the inner step of the spectral prelude on the device, float64, gfx950 only.
//
// Reference: graph_coarsening/coarsening_utils.py:83-90 hands T = 2 max(dw) I - L to ARPACK's eigsh(k = 10, which = 'LM',
// tol = 1e-5): an implicitly restarted Lanczos iteration whose cost is the matvec plus the (re)orthogonalisation of the new
// vector against the basis (42 % of the reference's coarsening time).  fitgnn_amd.coarsening.lanczos_smallest runs
// thick-restart Lanczos with full two-pass reorthogonalisation; round 2 issued one step as one sparse product plus four f64
// `gemv`s of the library over a row-major basis (2.58 ms each at N = 165 000: 62 % of all GPU time of a bench run).  Here the
// step is five launches over a COLUMN-major basis V [m + 1][N] (every basis vector contiguous):
//     w = T v_j                                         fitgnn_lanczos_spmv_f64      (CSR, 8 lanes per row)
//     h  = V^T w                                        fitgnn_lanczos_project_f64   (no subtraction)
//     w -= V h ;  h2 = V^T w                            fitgnn_lanczos_project_f64
//     w -= V h2 ; |w|^2                                 fitgnn_lanczos_project_f64
//     beta = |w| ; v_{j+1} = w / beta ; H[:, j] = h + h2, H[j+1, j] = beta       fitgnn_lanczos_finish_f64
// One projection pass reads V once from HBM (rows on lanes: coalesced for every column); the dot products leave a workgroup as one
// partial row per workgroup and are folded in a fixed order by the next launch -- no atomics: the result is reproducible.
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
//   h = sum of the n_prev partial rows of the PREVIOUS pass (part_in, may be NULL: no subtraction)
//   w[i] -= sum_c V[c][i] h[c]            (c ascending)
//   part_out[b][c] = sum_i V[c][i] w[i]   (c < ncol)   and   norm_out[b] = sum_i w[i]^2
__global__ __launch_bounds__(kThreads) void lanczos_project_kernel(const double *__restrict__ V, int64_t ldv, int ncol, double *__restrict__ w,
                                                                  int n, int rows_per_block, const double *__restrict__ part_in, int n_prev,
                                                                  double *__restrict__ part_out, double *__restrict__ norm_out) {
    __shared__ double s_h[kMaxCols];
    __shared__ double s_red[4][kMaxCols + 1];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (part_in) {
        if (t < ncol) {
            double s = 0.0;
            for (int b = 0; b < n_prev; ++b) s += part_in[(int64_t)b * ncol + t];
            s_h[t] = s;
        }
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
            if (part_in) {
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
    if (lane == 0) s_red[wave][kMaxCols] = nrm;
    __syncthreads();
    if (t < ncol) part_out[(int64_t)blockIdx.x * ncol + t] = ((s_red[0][t] + s_red[1][t]) + s_red[2][t]) + s_red[3][t];
    if (t == 0 && norm_out) norm_out[blockIdx.x] = ((s_red[0][kMaxCols] + s_red[1][kMaxCols]) + s_red[2][kMaxCols]) + s_red[3][kMaxCols];
}

// beta = sqrt(sum of the norm partials); V[j + 1] = w / max(beta, tiny); workgroup 0 also writes column j of the projected matrix:
// H[c][j] = h[c] + h2[c] (the two passes' coefficient sums), H[j + 1][j] = beta.
__global__ __launch_bounds__(kThreads) void lanczos_finish_kernel(double *__restrict__ V, int64_t ldv, int j, const double *__restrict__ w, int n,
                                                                 const double *__restrict__ norm_part, const double *__restrict__ part_a,
                                                                 const double *__restrict__ part_b, int n_part, double *__restrict__ Hm, int ldh) {
    __shared__ double s_beta;
    const int t = threadIdx.x;
    if (t == 0) {
        double s = 0.0;
        for (int b = 0; b < n_part; ++b) s += norm_part[b];
        s_beta = sqrt(s);
    }
    __syncthreads();
    const double beta = s_beta;
    const double inv = 1.0 / fmax(beta, 1e-300);
    const int ncol = j + 1;
    if (blockIdx.x == 0) {
        if (t < ncol) {
            double a = 0.0, b2 = 0.0;
            for (int b = 0; b < n_part; ++b) {
                a += part_a[(int64_t)b * ncol + t];
                b2 += part_b[(int64_t)b * ncol + t];
            }
            Hm[(int64_t)t * ldh + j] = a + b2;
        }
        if (t == 0) Hm[(int64_t)(j + 1) * ldh + j] = beta;
    }
    for (int64_t i = (int64_t)blockIdx.x * kThreads + t; i < n; i += (int64_t)gridDim.x * kThreads) V[(int64_t)(j + 1) * ldv + i] = w[i] * inv;
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

extern "C" int fitgnn_lanczos_project_f64(const double *V, int64_t ldv, int32_t ncol, double *w, int32_t n, const double *part_in,
                                          double *part_out, double *norm_out, void *stream) {
    if (n < 0 || ncol < 1 || ncol > kMaxCols || ldv < n) return FITGNN_E_BADARG;
    if (n == 0) return 0;
    if (!V || !w || !part_out) return FITGNN_E_BADARG;
    const int nb = project_blocks(n);
    hipLaunchKernelGGL(lanczos_project_kernel, dim3((unsigned)nb), dim3(kThreads), 0, (hipStream_t)stream, V, ldv, ncol, w, n,
                       kThreads * kRowsPerThread, part_in, nb, part_out, norm_out);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_lanczos_finish_f64(double *V, int64_t ldv, int32_t j, const double *w, int32_t n, const double *norm_part,
                                         const double *part_a, const double *part_b, double *H, int32_t ldh, void *stream) {
    if (n < 0 || j < 0 || j + 1 > kMaxCols || ldv < n || ldh < j + 1) return FITGNN_E_BADARG;
    if (n == 0) return 0;
    if (!V || !w || !norm_part || !part_a || !part_b || !H) return FITGNN_E_BADARG;
    const int nb = project_blocks(n);
    const int grid = (int)fmin(1024.0, (double)((n + kThreads - 1) / kThreads));
    hipLaunchKernelGGL(lanczos_finish_kernel, dim3((unsigned)(grid < 1 ? 1 : grid)), dim3(kThreads), 0, (hipStream_t)stream, V, ldv, j, w, n,
                       norm_part, part_a, part_b, nb, H, ldh);
    return (int)hipGetLastError();
}
