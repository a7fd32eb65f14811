// gemm_nt.hip -- c[R x N] = a[R x K] @ b[N x K]^T for row-major fp32 operands (gfx950 only).
//
// The two remaining tall GEMMs of a hidden layer on the train path: the forward h = x W^T of GCNConv's bias-free Linear
// (network.py:31 through torch_geometric's GCNConv) and, with b = W^T materialised (1 MB), the input
// gradient dX = dH @ W of its backward (run.py:207 / :246 `loss.backward()`).  R ~ 1e5 union rows, N = K = 512.
//
// Same arithmetic and machinery as gemm_atb.hip (read its header first): every fp32 operand is split into bf16 hi
// (round to nearest) and lo = bf16(x - hi); hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate.  What
// differs is the operand geometry: both operands are K-contiguous here, so the 8 consecutive k of an MFMA operand lane
// are 32 contiguous bytes of one row.  Eight lanes read one row's 128-byte stage slice (8 rows per
// global_load_dwordx4), a lane converts its 4 k to half a fragment and stores it with ds_write_b64.
// Workgroup = 256 x 256 output tile, full K (no split), 8 waves of 64 x 128; LDS double-buffered per 32-wide k stage.
// LDS image: [side][k16 step][hi|lo][32-row tile][k-half][slot] x 16 B with
//   slot(row r4..r0, step ks, half h) = (r0, r1, r4^r0, r3, r2) ^ (ks << 1 | h)
// -- the permutation of gemm_atb.hip (16-lane ds_read_b128 groups stay on distinct slots modulo 16) plus a per-(ks, h)
// XOR that spreads the 16 lanes of a ds_write_b64 group (2 rows x 8 quarter-slices) over all 32 banks.
#include "common.h"
#include "fitgnn_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kStage = 32;
constexpr int kBlk = 64 * 16;        // [k-half 2][row 32] x 16 B: the operand of one MFMA
// WMN = waves along a tile edge / 64-row blocks per side: 4 -> 256 x 256 tile, 8 waves of 64 x 128, 128 KB LDS (one workgroup
// per CU); 2 -> 128 x 128 tile, 4 waves of 64 x 64, 64 KB LDS (two workgroups per CU, whose barriers are independent)
template <int WMN>
struct Geo {
    static constexpr int kTile = 64 * WMN;
    static constexpr int kThreads = 128 * WMN;
    static constexpr int kPart = 2 * WMN * kBlk;   // row tiles of one side
    static constexpr int kStep = 2 * kPart;        // hi, lo
    static constexpr int kOperand = 2 * kStep;     // two k16 steps
    static constexpr int kBuf = 2 * kOperand;      // a side, b side
    static constexpr int kLdsBytes = 2 * kBuf;
};

__device__ __forceinline__ uint32_t pack_bf16_rne(float x0, float x1) {
    f32x2 v = {x0, x1};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(uint32_t, r);
}

// EPI: the product is the gradient dOut of a fused layer output  out = dropout(ELU(z))  and what is stored is
//   dZ = keep ? dOut / (1 - p) * (o > 0 ? 1 : o + 1) : 0,   o = out * (1 - p)
// (the arithmetic of epilogue_bwd_kernel, gcn_ops.hip), with the per-tile column sums of dZ written to `col_part`
// [tiles_m x N] for the bias gradient -- the [R x N] gradient is never written and re-read un-transformed.
template <int WMN, bool EPI>
__global__ __launch_bounds__(128 * WMN, WMN == 2 ? 2 : 1) void gemm_nt_kernel(const float *__restrict__ a, long lda,
                                                              const float *__restrict__ b, long ldb, long R, int N, int K,
                                                              int tiles_m, int tiles_n, float *__restrict__ c, long ldc,
                                                              const float *__restrict__ out, float *__restrict__ col_part,
                                                              uint32_t epi, float p_drop, uint64_t seed_arg,
                                                              const uint8_t *__restrict__ mask) {
    static_assert(!EPI || WMN == 4, "the fused epilogue is written for the 256 x 256 tile");
    constexpr int kTile = Geo<WMN>::kTile, kPart = Geo<WMN>::kPart, kStep = Geo<WMN>::kStep, kOperand = Geo<WMN>::kOperand,
                  kBuf = Geo<WMN>::kBuf, JT = WMN;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the column tiles of one row tile sit on consecutive slots of one XCD: the a slab's second reader hits that L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tn = slot % tiles_n, tm = xcd + 8 * (slot / tiles_n);
    if (tm >= tiles_m) return;

    // staging role: waves 0-3 the a side, 4-7 the b side; a wave owns 64 rows of the tile, 8 per load
    const int side = wave / WMN, ws = wave % WMN;
    const float *src = side ? b : a;
    const long ld = side ? ldb : lda;
    const long nrows = side ? (long)N : R;
    const long tile_base = (long)(side ? tn : tm) * kTile;  // < nrows
    const int f = lane & 7;                       // which float4 of the row's 32-k slice
    const int ks_w = f >> 2, h_w = (f >> 1) & 1;  // its k16 step, k-half; (f & 1): which half of the fragment
    const int j = lane >> 3;                      // row within the group of 8: r0..r2
    const int slot_j = ((j >> 2) & 1) | ((j & 1) << 2) | (((j >> 1) & 1) << 3) | ((j & 1) << 4);
    const int wr_lane = side * kOperand + ks_w * kStep + (2 * ws) * kBlk + h_w * 512 + 8 * (f & 1) +
                        ((slot_j ^ (ks_w << 1 | h_w)) * 16);
    // addresses: wave-uniform base (tile's first row, stage's k) + a 32-bit lane offset per load (row within the tile,
    // clamped into the operand: rows past its end only feed outputs that are never stored)
    const float *tile_src = src + tile_base * ld;
    unsigned voff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        long row = tile_base + 64 * ws + 8 * i + (lane >> 3);
        row = row < nrows ? row : nrows - 1;
        voff[i] = (unsigned)((row - tile_base) * ld * 4 + f * 16);
    }

    f32x4 g[8];
    // hand-placed loads and counted waits, as in gemm_atb.hip (`after`: fake dependence on the conversion of the
    // registers being refilled)
    auto load_row = [&](int k0, int i, uint32_t after) {
        const float *base = tile_src + k0;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(g[i]) : "v"(voff[i]), "s"(base), "v"(after));
    };
    auto convert_and_reload = [&](unsigned char *buf, int k_next) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("s_waitcnt vmcnt(7)" : "+v"(g[i]));
            const uint32_t h0 = pack_bf16_rne(g[i][0], g[i][1]), h1 = pack_bf16_rne(g[i][2], g[i][3]);
            const uint32_t l0 = pack_bf16_rne(g[i][0] - __uint_as_float(h0 << 16), g[i][1] - __uint_as_float(h0 & 0xffff0000u));
            const uint32_t l1 = pack_bf16_rne(g[i][2] - __uint_as_float(h1 << 16), g[i][3] - __uint_as_float(h1 & 0xffff0000u));
            // row 8 i + j: r3 = i & 1, r4 = (i >> 1) & 1 flip slot bits 1 and 2; i >> 2 selects the 32-row tile
            const int off = (wr_lane ^ (((i & 1) << 5) | (((i >> 1) & 1) << 6))) + (i >> 2) * kBlk;
            *reinterpret_cast<uint2 *>(buf + off) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(buf + off + kPart) = make_uint2(l0, l1);
            load_row(k_next, i, l0 ^ l1);
        }
    };

    const int r5 = lane & 31, hh = lane >> 5;
    const int slot_r = ((r5 >> 2) & 1) | (((r5 >> 3) & 1) << 1) | ((((r5 >> 4) ^ r5) & 1) << 2) | (((r5 >> 1) & 1) << 3) |
                       ((r5 & 1) << 4);
    int rd_lane[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) rd_lane[ks] = ks * kStep + hh * 512 + ((slot_r ^ (ks << 1 | hh)) * 16);

    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][JT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < JT; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    auto compute = [&](const unsigned char *buf, int ks) {
        const unsigned char *pa = buf + (2 * wm) * kBlk + rd_lane[ks];
        const unsigned char *pb = buf + kOperand + (JT * wn) * kBlk + rd_lane[ks];
        bf16x8 fa[2], fb[JT];
#pragma unroll
        for (int jj = 0; jj < JT; ++jj) fb[jj] = *reinterpret_cast<const bf16x8 *>(pb + jj * kBlk);
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8 *>(pa + kPart + i * kBlk);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < JT; ++jj)
                acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[jj], acc[i][jj], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8 *>(pa + i * kBlk);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < JT; ++jj)
                acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[jj], acc[i][jj], 0, 0, 0);
#pragma unroll
        for (int jj = 0; jj < JT; ++jj) fb[jj] = *reinterpret_cast<const bf16x8 *>(pb + kPart + jj * kBlk);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < JT; ++jj)
                acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[jj], acc[i][jj], 0, 0, 0);
    };

    const int nstage = K / kStage;  // K % 32 == 0 (launcher)
    // prefetches past the last stage re-read the last one (never converted into a buffer that is multiplied)
    auto k_of = [&](int s) { return (s < nstage ? s : nstage - 1) * kStage; };
#pragma unroll
    for (int i = 0; i < 8; ++i) load_row(0, i, 0u);
    convert_and_reload(lds, k_of(1));
    for (int s = 0; s < nstage; ++s) {
        const unsigned char *cur = lds + (s & 1) * kBuf;
        unsigned char *nxt = lds + ((s + 1) & 1) * kBuf;
        __syncthreads();
        compute(cur, 0);
        convert_and_reload(nxt, k_of(s + 2));
        compute(cur, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7])
                 :
                 : "memory");

    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    if (!EPI) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int jj = 0; jj < JT; ++jj) {
                const int n = tn * kTile + (JT * wn + jj) * 32 + (lane & 31);
                const long m0 = (long)tm * kTile + (2 * wm + i) * 32 + 4 * (lane >> 5);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long m = m0 + (r & 3) + 8 * (r >> 2);
                    if (m < R && n < N) c[m * ldc + n] = acc[i][jj][r];
                }
            }
        }
        return;
    }
    if constexpr (EPI) {
    const uint64_t seed = fitgnn::resolve_seed(seed_arg, epi);
    const bool drop = (epi & FITGNN_EPI_DROPOUT) != 0, elu = (epi & FITGNN_EPI_ELU) != 0;
    const float scale = drop ? 1.0f / (1.0f - p_drop) : 1.0f;
    const float unscale = drop ? (1.0f - p_drop) : 1.0f;
    const uint32_t thresh = fitgnn::dropout_threshold(p_drop);
    // The accumulators go through LDS once (each wave its own 32 x 128 block per pass) so that the epilogue runs on
    // rows: a lane then owns 4 consecutive columns -- exactly one dropout group (one hash per float4, as in
    // epilogue_bwd_kernel) -- and `out` / dZ move as 16-byte accesses, 512 contiguous bytes per half wave.
    float colsum[4] = {0.f, 0.f, 0.f, 0.f};
    float *blk = reinterpret_cast<float *>(lds) + wave * (32 * 128);
    const int n = tn * kTile + wn * 128 + 4 * (lane & 31);
    const int nc = n < N ? n : N - 4;  // N % 4 == 0
    const uint64_t n4 = (uint64_t)(N >> 2);
    __syncthreads();  // every wave has left the last LDS stage
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                blk[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 128 + jj * 32 + (lane & 31)] = acc[i][jj][r];
        const long mrow = (long)tm * kTile + (2 * wm + i) * 32 + (lane >> 5);
        float4 o[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {  // the 16 loads of this pass before the first use
            long m = mrow + 2 * k;
            m = m < R ? m : R - 1;
            o[k] = *reinterpret_cast<const float4 *>(out + m * (long)N + nc);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const long m = mrow + 2 * k;
            const float4 v = *reinterpret_cast<const float4 *>(blk + (2 * k + (lane >> 5)) * 128 + 4 * (lane & 31));
            float d[4] = {v.x, v.y, v.z, v.w};
            const float ov[4] = {o[k].x, o[k].y, o[k].z, o[k].w};
            const bool live = m < R && n < N;
            const uint64_t idx = (uint64_t)m * (uint64_t)N + (uint64_t)n;
            uint64_t bits = 0;
            if (drop && !mask) bits = fitgnn::dropout_bits(seed, (uint64_t)m * n4 + (uint64_t)(n >> 2));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (drop) {
                    const bool keep = mask ? (live && mask[idx + e] != 0) : fitgnn::dropout_keep(bits, e, thresh);
                    d[e] = keep ? d[e] * scale : 0.f;
                }
                if (elu) {
                    const float ev = ov[e] * unscale;
                    d[e] = ev > 0.f ? d[e] : d[e] * (ev + 1.0f);
                }
            }
            if (live) {
                *reinterpret_cast<float4 *>(c + m * ldc + n) = make_float4(d[0], d[1], d[2], d[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) colsum[e] += d[e];
            }
        }
        __builtin_amdgcn_wave_barrier();  // the block is rewritten by the next pass
    }
    // bias gradient: this workgroup's 256 rows per column, in a fixed order: lane halves, then the 4 row-waves
    __syncthreads();
    float *red = reinterpret_cast<float *>(lds);  // [4 wm][256 columns]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float other = __shfl_xor(colsum[e], 32);
        if (lane < 32) red[wm * kTile + wn * 128 + 4 * lane + e] = colsum[e] + other;
    }
    __syncthreads();
    if (tid < kTile) {
        const int nn = tn * kTile + tid;
        if (nn < N) col_part[(long)tm * N + nn] = (red[tid] + red[kTile + tid]) + (red[2 * kTile + tid] + red[3 * kTile + tid]);
    }
    }
}

}  // namespace

extern "C" int fitgnn_colsum_partials_f32(const float *partial, int32_t n_chunks, int32_t H, float *out, void *stream);

namespace {
int launch_nt(bool epi_on, const float *a, int64_t lda, const float *b, int64_t ldb, int64_t R, int32_t N, int32_t K, float *c,
              int64_t ldc, const float *out, float *col_part, uint32_t epi, float p_drop, uint64_t seed, const uint8_t *mask,
              void *stream) {
    if (R < 0 || N <= 0 || K < kStage || (K % kStage) != 0 || lda < K || ldb < K || ldc < N || (lda % 4) != 0 || (ldb % 4) != 0)
        return FITGNN_E_BADARG;
    if (R == 0) return 0;
    if (!a || !b || !c) return FITGNN_E_BADARG;
    if ((((uintptr_t)a | (uintptr_t)b) % 16) != 0) return FITGNN_E_ALIGN;
    // 128 x 128 tiles (two independent workgroups per CU) lose to 256 x 256 on a full grid (208 vs 175 us at R = 90 549:
    // twice the L2 -> LDS traffic per flop) and win when the large tiles would leave most CUs idle (18 vs 36 us at R = 300)
    const int64_t big_grid = ((R + Geo<4>::kTile - 1) / Geo<4>::kTile) * ((N + Geo<4>::kTile - 1) / Geo<4>::kTile);
    const bool small = !epi_on && big_grid < 128;
    const int T = small ? Geo<2>::kTile : Geo<4>::kTile;
    const int lds_bytes = small ? Geo<2>::kLdsBytes : Geo<4>::kLdsBytes;
    const int tiles_m = (int)((R + T - 1) / T), tiles_n = (N + T - 1) / T;
    const int groups = (tiles_m + 7) / 8;
    auto kern = epi_on ? gemm_nt_kernel<4, true> : small ? gemm_nt_kernel<2, false> : gemm_nt_kernel<4, false>;
    FITGNN_RETURN_IF_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)(groups * 8 * tiles_n)), dim3(small ? Geo<2>::kThreads : Geo<4>::kThreads), lds_bytes, (hipStream_t)stream, a,
                       (long)lda, b, (long)ldb, (long)R, N, K, tiles_m, tiles_n, c, (long)ldc, out, col_part, epi, p_drop, seed,
                       mask);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int fitgnn_gemm_nt_f32(const float *a, int64_t lda, const float *b, int64_t ldb, int64_t R, int32_t N, int32_t K,
                                  float *c, int64_t ldc, void *stream) {
    return launch_nt(false, a, lda, b, ldb, R, N, K, c, ldc, nullptr, nullptr, 0u, 0.f, 0ull, nullptr, stream);
}

extern "C" size_t fitgnn_gemm_nt_epilogue_bwd_workspace_bytes(int64_t R, int32_t N) {
    if (R <= 0 || N <= 0) return 0;
    return (size_t)((R + Geo<4>::kTile - 1) / Geo<4>::kTile) * (size_t)N * sizeof(float);
}

extern "C" int fitgnn_gemm_nt_epilogue_bwd_f32(const float *a, int64_t lda, const float *b, int64_t ldb, int64_t R, int32_t N,
                                               int32_t K, const float *out, float *dZ, uint32_t epilogue, float p_drop,
                                               uint64_t seed, const uint8_t *mask, float *db, void *work, size_t work_bytes,
                                               void *stream) {
    if (R > 0 && (!out || !dZ || !work)) return FITGNN_E_BADARG;
    if ((epilogue & FITGNN_EPI_DROPOUT) && !(p_drop >= 0.f && p_drop < 1.f)) return FITGNN_E_BADARG;
    if ((N % 4) != 0) return FITGNN_E_BADARG;  // the dropout groups of 4 columns must not straddle rows
    if ((((uintptr_t)out | (uintptr_t)dZ) % 16) != 0) return FITGNN_E_ALIGN;
    if (work_bytes < fitgnn_gemm_nt_epilogue_bwd_workspace_bytes(R, N)) return FITGNN_E_WORKSPACE;
    const int rc = launch_nt(true, a, lda, b, ldb, R, N, K, dZ, N, out, (float *)work, epilogue, p_drop, seed, mask, stream);
    if (rc != 0 || R == 0 || !db) return rc;
    return fitgnn_colsum_partials_f32((const float *)work, (int32_t)((R + Geo<4>::kTile - 1) / Geo<4>::kTile), N, db, stream);
}
