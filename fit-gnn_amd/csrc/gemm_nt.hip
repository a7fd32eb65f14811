// gemm_nt.hip -- c[R x N] = a[R x K] @ b[N x K]^T for row-major fp32 operands (gfx950 only).
//
// The two remaining tall GEMMs of a hidden layer on the train path: the forward h = x W^T of GCNConv's bias-free Linear
// (network.py:31 through torch_geometric's GCNConv) and, with b = W^T, the input gradient
// dX = dH @ W of its backward (run.py:207 / :246 `loss.backward()`).  R ~ 1e5 union rows, N = K = 512.
// Variants (template arguments): tile edge 256 or 128; EPI = the previous fused layer's epilogue backward applied to the
// accumulators (dOut never written); PRE = b given as a pre-split LDS image (nt_presplit_kernel reads it through its
// strides: W^T is never materialised) and staged by LDS-DMA.
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

// tools/microbench/gemm_probe.hip compiles this file with parts of the kernel removed (which of its streams bounds it)
#ifdef PROBE_NO_MFMA
#define PROBE_MFMA(x) asm volatile("" : "+v"(acc[i][jj]) : "v"(fa[i]), "v"(fb[jj]))
#else
#define PROBE_MFMA(x) x
#endif
#ifdef PROBE_NO_LDSREAD
#define PROBE_FRAG(dst, src) asm volatile("" : "=v"(dst) : "v"(src))
#else
#define PROBE_FRAG(dst, src) dst = *reinterpret_cast<const bf16x8 *>(src)
#endif

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
// PRE: the b operand arrives pre-split (nt_presplit_kernel below): for every (column tile, 32-wide k stage) the exact 32-KB
// LDS image of the b side, so its staging is 4 LDS-DMA instructions per wave and stage (global_load_lds_dwordx4: no
// registers, no conversion, no ds_write) and all 8 waves share the a side (4 loads each instead of 8).
template <int WMN, bool EPI, bool PRE>
__global__ __launch_bounds__(128 * WMN, WMN == 2 ? 2 : 1) void gemm_nt_kernel(const float *__restrict__ a, long lda,
                                                              const float *__restrict__ b, long ldb, long R, int N, int K,
                                                              int tiles_m, int tiles_n, float *__restrict__ c, long ldc,
                                                              const float *__restrict__ out, float *__restrict__ col_part,
                                                              uint32_t epi, float p_drop, uint64_t seed_arg,
                                                              const uint8_t *__restrict__ mask) {
    static_assert(!EPI || WMN == 4, "the fused epilogue is written for the 256 x 256 tile");
    static_assert(!PRE || WMN == 4, "the pre-split b image is laid out for the 256 x 256 tile");
    constexpr int NG = PRE ? 4 : 8;  // float4 loads per lane and stage
    constexpr int kTile = Geo<WMN>::kTile, kPart = Geo<WMN>::kPart, kStep = Geo<WMN>::kStep, kOperand = Geo<WMN>::kOperand,
                  kBuf = Geo<WMN>::kBuf, JT = WMN;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the column tiles of one row tile sit on consecutive slots of one XCD: the a slab's second reader hits that L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tn = slot % tiles_n, tm = xcd + 8 * (slot / tiles_n);
    if (tm >= tiles_m) return;

    // staging role: waves 0-3 the a side, 4-7 the b side, a wave owns 64 rows of the tile, 8 per load; with PRE every wave
    // stages 32 rows of the a side (4 loads) and an eighth of the b side's stage image by DMA
    const int side = PRE ? 0 : wave / WMN, ws = wave % WMN;
    const float *src = side ? b : a;
    const long ld = side ? ldb : lda;
    const long nrows = side ? (long)N : R;
    const long tile_base = (long)(side ? tn : tm) * kTile;  // < nrows
    const int f = lane & 7;                       // which float4 of the row's 32-k slice
    const int ks_w = f >> 2, h_w = (f >> 1) & 1;  // its k16 step, k-half; (f & 1): which half of the fragment
    const int j = lane >> 3;                      // row within the group of 8: r0..r2
    const int slot_j = ((j >> 2) & 1) | ((j & 1) << 2) | (((j >> 1) & 1) << 3) | ((j & 1) << 4);
    const int row_tile0 = PRE ? wave : 2 * ws;    // first 32-row tile this wave stages
    const int wr_lane = side * kOperand + ks_w * kStep + row_tile0 * kBlk + h_w * 512 + 8 * (f & 1) +
                        ((slot_j ^ (ks_w << 1 | h_w)) * 16);
    // addresses: wave-uniform base (tile's first row, stage's k) + a 32-bit lane offset per load (row within the tile,
    // clamped into the operand: rows past its end only feed outputs that are never stored)
    const float *tile_src = src + tile_base * ld;
    unsigned voff[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        long row = tile_base + 32 * row_tile0 + 8 * i + (lane >> 3);
        row = row < nrows ? row : nrows - 1;
        voff[i] = (unsigned)((row - tile_base) * ld * 4 + f * 16);
    }

    f32x4 g[NG];
    // hand-placed loads and counted waits, as in gemm_atb.hip (`after`: fake dependence on the conversion of the
    // registers being refilled)
    auto load_row = [&](int k0, int i, uint32_t after) {
#ifdef PROBE_NO_ALOAD  // every stage reads the same few rows: L2 hits instead of the HBM stream
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(g[i]) : "v"(voff[i] & 0xffffu), "s"(a + k0), "v"(after));
#else
        const float *base = tile_src + k0;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(g[i]) : "v"(voff[i]), "s"(base), "v"(after));
#endif
    };
    // Outstanding vector-memory operations when g[i] is converted, oldest first: the rest of the stage held in g, (PRE) the 4
    // DMA instructions of this iteration, the i loads already refilled -- NG - 1 (+ 4) in every case.
    auto convert_and_reload = [&](unsigned char *buf, int k_next) {
#pragma unroll
        for (int i = 0; i < NG; ++i) {
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
    // PRE: this wave's 4 KB of the b side's stage image, global -> LDS (destination = M0 + 16 * lane: the image is stored
    // in LDS order, so consecutive lanes copy consecutive 16 bytes)
    const unsigned char *bimg = reinterpret_cast<const unsigned char *>(b) + (long)tn * (K / kStage) * kOperand +
                                wave * 4096 + lane * 16;
    auto dma_stage = [&](unsigned char *buf, int stage) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned char *gsrc = bimg + (long)stage * kOperand + q * 1024;
            const unsigned lds_dst = (unsigned)(uintptr_t)(buf + kOperand + wave * 4096 + q * 1024);  // LDS byte address
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(gsrc), "s"(lds_dst)
                         : "memory");
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
        for (int jj = 0; jj < JT; ++jj) PROBE_FRAG(fb[jj], pb + jj * kBlk);
#pragma unroll
        for (int i = 0; i < 2; ++i) PROBE_FRAG(fa[i], pa + kPart + i * kBlk);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < JT; ++jj)
                PROBE_MFMA(acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[jj], acc[i][jj], 0, 0, 0));
#pragma unroll
        for (int i = 0; i < 2; ++i) PROBE_FRAG(fa[i], pa + i * kBlk);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < JT; ++jj)
                PROBE_MFMA(acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[jj], acc[i][jj], 0, 0, 0));
#pragma unroll
        for (int jj = 0; jj < JT; ++jj) PROBE_FRAG(fb[jj], pb + kPart + jj * kBlk);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < JT; ++jj)
                PROBE_MFMA(acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[jj], acc[i][jj], 0, 0, 0));
    };

    const int nstage = K / kStage;  // K % 32 == 0 (launcher)
    // prefetches past the last stage re-read the last one (never converted into a buffer that is multiplied)
    auto k_of = [&](int s) { return (s < nstage ? s : nstage - 1) * kStage; };
#pragma unroll
    for (int i = 0; i < NG; ++i) load_row(0, i, 0u);
    if (PRE) dma_stage(lds, 0);
    convert_and_reload(lds, k_of(1));
    for (int s = 0; s < nstage; ++s) {
        const unsigned char *cur = lds + (s & 1) * kBuf;
        unsigned char *nxt = lds + ((s + 1) & 1) * kBuf;
        if (PRE) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // the DMA of stage s has landed (4 younger loads may fly)
        __syncthreads();
        if (PRE) dma_stage(nxt, s + 1 < nstage ? s + 1 : nstage - 1);
        compute(cur, 0);
        convert_and_reload(nxt, k_of(s + 2));
        compute(cur, 1);
    }
    // prefetches (and the last DMA) are still in flight: g stays allocated until they have landed
    if constexpr (PRE) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]) : : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[NG - 4]), "+v"(g[NG - 3]), "+v"(g[NG - 2]), "+v"(g[NG - 1])
                     :
                     : "memory");
    }

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
#ifdef PROBE_NO_STORE
                    if (m < R && n < N && acc[i][jj][r] == 12345.678f) c[m * ldc + n] = acc[i][jj][r];
#else
                    if (m < R && n < N) c[m * ldc + n] = acc[i][jj][r];
#endif
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

// b [N x K] (element (n, k) at b[n * sn + k * sk]: also serves b = W^T without materialising it) -> for every column tile
// and 32-wide k stage the 32-KB b-side LDS image of gemm_nt_kernel<4, *, true>; rows past N are zero.  One thread per
// (row, 4 consecutive k).
__global__ __launch_bounds__(256) void nt_presplit_kernel(const float *__restrict__ b, long sn, long sk, int N, int K,
                                                          int K_valid, unsigned char *__restrict__ img) {
    constexpr int kPart = Geo<4>::kPart, kStep = Geo<4>::kStep, kOperand = Geo<4>::kOperand, kTile = Geo<4>::kTile;
    const int k4 = K / 4;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int tiles_n = (N + kTile - 1) / kTile;
    if (t >= (long)tiles_n * kTile * k4) return;
    const int n = (int)(t / k4), k = (int)(t % k4) * 4;
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k + e < K_valid) x[e] = b[n * sn + (k + e) * sk];  // k >= K_valid: zero padding up to the stage width
    }
    const uint32_t h0 = pack_bf16_rne(x[0], x[1]), h1 = pack_bf16_rne(x[2], x[3]);
    const uint32_t l0 = pack_bf16_rne(x[0] - __uint_as_float(h0 << 16), x[1] - __uint_as_float(h0 & 0xffff0000u));
    const uint32_t l1 = pack_bf16_rne(x[2] - __uint_as_float(h1 << 16), x[3] - __uint_as_float(h1 & 0xffff0000u));
    const int tn = n / kTile, row = n % kTile, r = row & 31;
    const int stage = k / kStage, kk = k % kStage, ks = kk >> 4, h = (kk >> 3) & 1, q = (kk >> 2) & 1;
    const int slot = ((r >> 2) & 1) | (((r >> 3) & 1) << 1) | ((((r >> 4) ^ r) & 1) << 2) | (((r >> 1) & 1) << 3) | ((r & 1) << 4);
    unsigned char *dst = img + ((long)tn * (K / kStage) + stage) * kOperand + ks * kStep + (row >> 5) * kBlk + h * 512 +
                         ((slot ^ (ks << 1 | h)) * 16) + 8 * q;
    *reinterpret_cast<uint2 *>(dst) = make_uint2(h0, h1);
    *reinterpret_cast<uint2 *>(dst + kPart) = make_uint2(l0, l1);
}

}  // namespace

extern "C" int fitgnn_colsum_partials_f32(const float *partial, int32_t n_chunks, int32_t H, float *out, void *stream);

namespace {
// b_is_image: b points at the pre-split image of nt_presplit_kernel (256 x 256 tiles only)
int launch_nt(bool epi_on, bool b_is_image, const float *a, int64_t lda, const float *b, int64_t ldb, int64_t R, int32_t N, int32_t K,
              float *c, int64_t ldc, const float *out, float *col_part, uint32_t epi, float p_drop, uint64_t seed,
              const uint8_t *mask, void *stream) {
    if (R < 0 || N <= 0 || K < kStage || (K % kStage) != 0 || lda < K || ldc < N || (lda % 4) != 0) return FITGNN_E_BADARG;
    if (!b_is_image && (ldb < K || (ldb % 4) != 0)) return FITGNN_E_BADARG;
    if (R == 0) return 0;
    if (!a || !b || !c) return FITGNN_E_BADARG;
    if ((((uintptr_t)a | (uintptr_t)b) % 16) != 0) return FITGNN_E_ALIGN;
    // 128 x 128 tiles (two independent workgroups per CU) lose to 256 x 256 on a full grid (208 vs 175 us at R = 90 549:
    // twice the L2 -> LDS traffic per flop) and win when the large tiles would leave most CUs idle (18 vs 36 us at R = 300)
    const int64_t big_grid = ((R + Geo<4>::kTile - 1) / Geo<4>::kTile) * ((N + Geo<4>::kTile - 1) / Geo<4>::kTile);
    const bool small = !epi_on && !b_is_image && big_grid < 128;
    const int T = small ? Geo<2>::kTile : Geo<4>::kTile;
    const int lds_bytes = small ? Geo<2>::kLdsBytes : Geo<4>::kLdsBytes;
    const int tiles_m = (int)((R + T - 1) / T), tiles_n = (N + T - 1) / T;
    const int groups = (tiles_m + 7) / 8;
    auto kern = b_is_image ? (epi_on ? gemm_nt_kernel<4, true, true> : gemm_nt_kernel<4, false, true>)
                           : (epi_on ? gemm_nt_kernel<4, true, false> : small ? gemm_nt_kernel<2, false, false> : gemm_nt_kernel<4, false, false>);
    static std::atomic<uint64_t> lds_done[5];  // one per kernel variant
    const int variant = b_is_image ? (epi_on ? 0 : 1) : (epi_on ? 2 : small ? 3 : 4);
    if (const int rc = fitgnn_lds_limit_once((const void *)kern, lds_bytes, lds_done[variant])) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)(groups * 8 * tiles_n)), dim3(small ? Geo<2>::kThreads : Geo<4>::kThreads), lds_bytes, (hipStream_t)stream, a,
                       (long)lda, b, (long)ldb, (long)R, N, K, tiles_m, tiles_n, c, (long)ldc, out, col_part, epi, p_drop, seed,
                       mask);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int fitgnn_gemm_nt_f32(const float *a, int64_t lda, const float *b, int64_t ldb, int64_t R, int32_t N, int32_t K,
                                  float *c, int64_t ldc, void *stream) {
    return launch_nt(false, false, a, lda, b, ldb, R, N, K, c, ldc, nullptr, nullptr, 0u, 0.f, 0ull, nullptr, stream);
}

extern "C" size_t fitgnn_gemm_nt_presplit_bytes(int32_t N, int32_t K) {
    if (N <= 0 || K < kStage || (K % kStage) != 0) return 0;
    return (size_t)((N + Geo<4>::kTile - 1) / Geo<4>::kTile) * (size_t)(K / kStage) * Geo<4>::kOperand;
}

extern "C" int fitgnn_gemm_nt_presplit_f32(const float *b, int64_t stride_n, int64_t stride_k, int32_t N, int32_t K,
                                           int32_t K_valid, void *image, void *stream) {
    if (N <= 0 || K < kStage || (K % kStage) != 0 || K_valid < 1 || K_valid > K || !b || !image) return FITGNN_E_BADARG;
    if (((uintptr_t)image % 16) != 0) return FITGNN_E_ALIGN;
    const long threads = (long)((N + Geo<4>::kTile - 1) / Geo<4>::kTile) * Geo<4>::kTile * (K / 4);
    hipLaunchKernelGGL(nt_presplit_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, b,
                       (long)stride_n, (long)stride_k, N, K, K_valid, (unsigned char *)image);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_gemm_nt_pre_f32(const float *a, int64_t lda, const void *b_image, int64_t R, int32_t N, int32_t K, float *c,
                                      int64_t ldc, void *stream) {
    return launch_nt(false, true, a, lda, (const float *)b_image, 0, R, N, K, c, ldc, nullptr, nullptr, 0u, 0.f, 0ull, nullptr,
                     stream);
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
    const bool image = ldb == 0;  // ldb == 0: b is a pre-split image (fitgnn_gemm_nt_presplit_f32)
    const int rc = launch_nt(true, image, a, lda, b, ldb, R, N, K, dZ, N, out, (float *)work, epilogue, p_drop, seed, mask, stream);
    if (rc != 0 || R == 0 || !db) return rc;
    return fitgnn_colsum_partials_f32((const float *)work, (int32_t)((R + Geo<4>::kTile - 1) / Geo<4>::kTile), N, db, stream);
}
