// gemm_atb.hip -- out[M x N] = a^T @ b for two tall row-major fp32 operands a [R x M], b [R x N] (gfx950 only).
//
// This is the weight-gradient product of every layer on the train path (grad_W = dH^T @ X, SURVEY.md 8 a11: "Backward
// = same SpMM on A^T + grad_W = x^T . grad_h GEMM"; reference call sites run.py:207 / :246 `loss.backward()` through
// torch_geometric's Linear): a reduction over ALL union rows (R ~ 1e5) into a small square.  The library serves the
// shape poorly (one call: 541 us at R = 90 549, M = N = 512; batched split-K through bmm: 201 us); its floor is the
// one pass over both operands (371 MB, ~75 us).
//
// Arithmetic: each fp32 operand x is split into hi = its upper 16 bits as a bf16 (truncation, so x - hi is exact) and
// lo = bf16_rne(x - hi); the product is hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the
// same three-product scheme as the library's "high" fp32 GEMM; measured ~5e-6 relative error against fp64).
//
// Decomposition: split-K.  Workgroup (chunk, tile) reduces rows [chunk*chunk_rows, +chunk_rows) into one 256 x 256
// output tile and writes it to partial[chunk]; fitgnn_sum_leading_f32 then adds the chunks in a fixed order, so the
// result is reproducible run to run (no atomics).  8 waves per workgroup, each owning a 64 x 128 block (2 x 4 MFMA
// tiles, 128 accumulator registers).  Per 32-row stage, waves 0-3 fetch the a-side 32 x 256 slab and waves 4-7 the
// b-side one: a lane reads 8 rows x 4 consecutive columns (8 x global_load_dwordx4, 1 KB contiguous per wave and row),
// which is exactly the 8 consecutive k an MFMA operand lane holds for one column -- the row-major-to-operand
// "transpose" costs no data movement, only the choice of which registers are packed together.  Fragments are stored in
// LDS in operand order ([k16 step][hi|lo][32-column tile][k-half][column] x 16 B), so every MFMA operand is one
// conflict-free ds_read_b128.  LDS is double-buffered (2 x 64 KB): one barrier per stage; the next stage's global loads
// are issued before the MFMA block of the current one.
// The 4 (or ntile) workgroups that share a row chunk get consecutive slots on the same XCD (blockIdx % 8), so the
// second read of an operand slab comes from that XCD's L2.
#include "common.h"
#include "fitgnn_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTile = 256;               // output tile edge
constexpr int kStage = 32;               // rows per pipeline stage (two k16 MFMA steps)
constexpr int kThreads = 512;
constexpr int kBlk = 64 * 16;            // [k-half 2][column 32] x 16 B: the operand of one MFMA, 1 KB
constexpr int kPart = 8 * kBlk;          // 8 column tiles
constexpr int kStep = 2 * kPart;         // hi, lo
constexpr int kOperand = 2 * kStep;      // two k16 steps
constexpr int kBuf = 2 * kOperand;       // a side, b side: 64 KB
constexpr int kLdsBytes = 2 * kBuf;      // double buffer

__device__ __forceinline__ uint32_t pack_bf16_rne(float x0, float x1) {
    f32x2 v = {x0, x1};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(uint32_t, r);
}

__global__ __launch_bounds__(kThreads, 1) void gemm_atb_kernel(const float *__restrict__ a, long lda,
                                                               const float *__restrict__ b, long ldb, long R, int M,
                                                               int N, long R_main, int chunk_rows, int tiles_n, int ntile,
                                                               float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = slot % ntile, chunk = xcd + 8 * (slot / ntile);
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const long k_begin = (long)chunk * chunk_rows;
    const long k_end = k_begin + chunk_rows < R_main ? k_begin + chunk_rows : R_main;

    // staging role of this wave: side (a or b) and which 8 of the stage's 32 rows
    const int side = wave >> 2, kgroup = wave & 3;
    const float *src = side ? b : a;
    const long ld = side ? ldb : lda;
    const int ncols = side ? N : M;
    int col = (side ? tn : tm) * kTile + 4 * lane;
    // a column past the operand only feeds output rows/columns that are never written: any in-bounds address will do
    col = col + 4 <= ncols ? col : ncols - 4;
    // Fragment slot of column r (5 bits r4..r0) inside its [tile][k-half] group of 32: bits (r0, r1, r4^r0, r3, r2)
    // from the top.  LDS services a ds_write_b128 in groups of 8 consecutive lanes over 32 banks and a ds_read_b128 in
    // the 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} over 64 banks.  A writer lane holds columns 4j..4j+3, so
    // with slots in column order each write instruction put its 8 lanes on 2 slots modulo 8 (4-way conflict: measured,
    // 55 % of all LDS-array cycles were conflict cycles); this permutation makes every write group hit 8 consecutive
    // slots and keeps every read group on 16 distinct slots modulo 16.
    const int wr_base = side * kOperand + (kgroup >> 1) * kStep + (lane >> 3) * kBlk + (kgroup & 1) * 512;
    const int wr_j = (lane & 7) * 16, wr_j4 = ((lane & 7) ^ 4) * 16;
    const int r5 = lane & 31;
    const int rd_lane = (lane >> 5) * 512 +
                        ((((r5 >> 2) & 1) | (((r5 >> 3) & 1) << 1) | ((((r5 >> 4) ^ r5) & 1) << 2) | (((r5 >> 1) & 1) << 3) |
                          ((r5 & 1) << 4)) * 16);

    f32x4 g[8];
    // The in-loop global loads and their waits are written by hand: left to the compiler, the refill of g is scheduled
    // across the last uses of the previous contents, which costs register copies at the loop edge and with them a wait
    // for the loads right after their issue (no prefetch left).  `after` is a fake input that pins the load behind the
    // conversion of the rows it overwrites.
    const unsigned col_bytes = (unsigned)col * 4u;
    auto load_row = [&](long k0, int i, uint32_t after) {
        long row = k0 + 8 * kgroup + i;
        row = row < R ? row : R - 1;
        const float *rowp = src + row * ld;  // wave-uniform base + one 32-bit lane offset
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(g[i]) : "v"(col_bytes), "s"(rowp), "v"(after));
    };
    // convert the stage held in g into `buf` and, row pair by row pair, refill g with the stage at k_next: the loads of
    // stage s+2 are in flight for a whole iteration before their first use.  8 loads are outstanding on entry; after
    // each pair two new ones are queued behind the old, so "at most 6 outstanding" always retires the next old pair.
    auto convert_and_reload = [&](unsigned char *buf, long k_next) {
        uint32_t hi[4][4], lo[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            asm volatile("s_waitcnt vmcnt(6)" : "+v"(g[2 * p]), "+v"(g[2 * p + 1]));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float x0 = g[2 * p][c], x1 = g[2 * p + 1][c];
                const uint32_t h = pack_bf16_rne(x0, x1);
                hi[c][p] = h;
                const float l0 = x0 - __uint_as_float(h << 16);
                const float l1 = x1 - __uint_as_float(h & 0xffff0000u);
                lo[c][p] = pack_bf16_rne(l0, l1);
            }
            const uint32_t after = lo[0][p] ^ lo[1][p] ^ lo[2][p] ^ lo[3][p];
            load_row(k_next, 2 * p, after);
            load_row(k_next, 2 * p + 1, after);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int off = wr_base + ((c & 1) ? wr_j4 : wr_j) + 128 * (c >> 1) + 256 * (c & 1);
            *reinterpret_cast<uint4 *>(buf + off) = make_uint4(hi[c][0], hi[c][1], hi[c][2], hi[c][3]);
            *reinterpret_cast<uint4 *>(buf + off + kPart) = make_uint4(lo[c][0], lo[c][1], lo[c][2], lo[c][3]);
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](const unsigned char *buf, int ks) {
        const unsigned char *pa = buf + ks * kStep + (2 * wm) * kBlk + rd_lane;
        const unsigned char *pb = buf + kOperand + ks * kStep + (4 * wn) * kBlk + rd_lane;
        // operand order keeps 24 fragment registers live: (lo_a, hi_b) -> (hi_a, hi_b) -> (hi_a, lo_b); small terms
        // first, 8 independent accumulators between two uses of the same one
        bf16x8 fa[2], fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8 *>(pb + j * kBlk);
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8 *>(pa + kPart + i * kBlk);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8 *>(pa + i * kBlk);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8 *>(pb + kPart + j * kBlk);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };

    // Stage s+1 is converted and written to the other LDS buffer while stage s is multiplied.
    const int nstage = k_end > k_begin ? (int)((k_end - k_begin) / kStage) : 0;
    if (nstage > 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) load_row(k_begin, i, 0u);
        convert_and_reload(lds, k_begin + kStage);
    }
    for (int s = 0; s < nstage; ++s) {
        const unsigned char *cur = lds + (s & 1) * kBuf;
        unsigned char *nxt = lds + ((s + 1) & 1) * kBuf;
        __syncthreads();
        compute(cur, 0);
        convert_and_reload(nxt, k_begin + (long)kStage * (s + 2));
        compute(cur, 1);
    }
    // the prefetch past the last stage is still in flight: g must stay allocated until it has landed, or the loads
    // would return into registers the epilogue has already reused
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7])
                 :
                 : "memory");

    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float *out = partial + (long)chunk * M * N;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = tn * kTile + (4 * wn + j) * 32 + (lane & 31);
            const int m0 = tm * kTile + (2 * wm + i) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (r & 3) + 8 * (r >> 2);
                if (m < M && n < N) out[(long)m * N + n] = acc[i][j][r];
            }
        }
    }
}

// out = sum over chunks of partial (fixed order) + the product over the last R % 32 rows, in plain fp32.
__global__ __launch_bounds__(256) void atb_reduce_kernel(const float4 *__restrict__ partial, int nchunks, const float *__restrict__ a,
                                                         long lda, const float *__restrict__ b, long ldb, long r_begin,
                                                         long R, int M, int N, float4 *__restrict__ out) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;  // 4 consecutive outputs of one row m
    const long MN4 = (long)M * N / 4;
    if (q >= MN4) return;
    // nchunks is a multiple of 8: eight independent running sums (eight loads in flight per lane), combined in a
    // fixed order
    float4 part[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) part[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < nchunks; c += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 v = partial[(long)(c + u) * MN4 + q];
            part[u].x += v.x, part[u].y += v.y, part[u].z += v.z, part[u].w += v.w;
        }
    }
    float4 acc = part[0];
#pragma unroll
    for (int u = 1; u < 8; ++u) acc.x += part[u].x, acc.y += part[u].y, acc.z += part[u].z, acc.w += part[u].w;
    const int m = (int)(q / (N / 4)), n = (int)(q % (N / 4)) * 4;
    for (long k = r_begin; k < R; ++k) {
        const float av = a[k * lda + m];
        const float4 bv = *reinterpret_cast<const float4 *>(b + k * ldb + n);
        acc.x += av * bv.x, acc.y += av * bv.y, acc.z += av * bv.z, acc.w += av * bv.w;
    }
    out[q] = acc;
}

struct Plan {
    int tiles_m, tiles_n, ntile, nchunks, chunk_rows;
    int64_t r_main;
};

Plan make_plan(int64_t R, int M, int N) {
    Plan p;
    p.tiles_m = (M + kTile - 1) / kTile;
    p.tiles_n = (N + kTile - 1) / kTile;
    p.ntile = p.tiles_m * p.tiles_n;
    int want = (256 + p.ntile - 1) / p.ntile;                    // one workgroup per CU
    p.r_main = R - R % kStage;
    const int64_t most = p.r_main / kStage > 0 ? p.r_main / kStage : 1;  // at least one stage per chunk
    if (want > most) want = (int)most;
    p.nchunks = ((want + 7) / 8) * 8;
    const int64_t rows = (p.r_main + p.nchunks - 1) / p.nchunks;
    p.chunk_rows = (int)(((rows + kStage - 1) / kStage) * kStage);
    return p;
}

}  // namespace

extern "C" size_t fitgnn_gemm_atb_workspace_bytes(int64_t R, int32_t M, int32_t N) {
    if (R <= 0 || M <= 0 || N <= 0) return 0;
    const Plan p = make_plan(R, M, N);
    return (size_t)p.nchunks * M * N * sizeof(float);
}

extern "C" int fitgnn_gemm_atb_f32(const float *a, int64_t lda, const float *b, int64_t ldb, int64_t R, int32_t M,
                                   int32_t N, float *out, void *workspace, void *stream) {
    if (R <= 0 || M < 4 || N < 4 || (M % 4) != 0 || (N % 4) != 0 || lda < M || ldb < N || (lda % 4) != 0 ||
        (ldb % 4) != 0)
        return FITGNN_E_BADARG;
    if (!a || !b || !out || !workspace) return FITGNN_E_BADARG;
    if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out | (uintptr_t)workspace) % 16) != 0) return FITGNN_E_ALIGN;
    const Plan p = make_plan(R, M, N);
    static std::atomic<uint64_t> lds_done{0};
    if (const int rc = fitgnn_lds_limit_once((const void *)gemm_atb_kernel, kLdsBytes, lds_done)) return rc;
    hipLaunchKernelGGL(gemm_atb_kernel, dim3((unsigned)(p.ntile * p.nchunks)), dim3(kThreads), kLdsBytes,
                       (hipStream_t)stream, a, (long)lda, b, (long)ldb, (long)R, M, N, (long)p.r_main, p.chunk_rows, p.tiles_n, p.ntile,
                       (float *)workspace);
    FITGNN_RETURN_IF_HIP(hipGetLastError());
    const int64_t MN4 = (int64_t)M * N / 4;
    hipLaunchKernelGGL(atb_reduce_kernel, dim3((unsigned)((MN4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)workspace, p.nchunks, a, (long)lda, b, (long)ldb, (long)p.r_main, (long)R, M, N,
                       (float4 *)out);
    return (int)hipGetLastError();
}
