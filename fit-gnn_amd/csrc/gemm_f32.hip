// gemm_f32.hip -- the dense products of a layer in the REFERENCE's arithmetic: fp32 operands, fp32 products, fp32 accumulation
// (gfx950 only).
//
// GCNConv's Linear (network.py:13-31 through torch_geometric: h = x W^T, no bias) and its backward (grad_x = grad_h W,
// grad_W = grad_h^T x; run.py:207 `loss.backward()`) are fp32 GEMMs in the reference.  gemm_nt.hip / gemm_atb.hip compute them as
// three bf16 products of a two-term split (4-5e-6 relative error against fp64: inside north_star's 1e-4 on logits, but narrower than
// the reference's own arithmetic).  This file is the fp32-faithful form on the matrix cores: v_mfma_f32_32x32x2_f32 multiplies fp32
// operands exactly and accumulates in fp32 (MI355X_MICROARCH.md: "exact f32, == fmaf chain"; 64 FLOP/clk/SIMD, 157 TFLOP/s dense
// peak = 1/16 of the bf16 rate), so the result differs from an fp64 product only by fp32 accumulation rounding (measured ~1e-7).
//
// ONE kernel serves the three products of a Linear; an operand is either "k-minor" (k contiguous in memory: x [rows x K], W [N x K])
// or "k-major" (k is the row index: the tall operands of grad_W = dH^T x, and W read as W^T in grad_x = dH W):
//     forward      c[i][j] = sum_k x[i][k] W[j][k]        A k-minor, B k-minor
//     grad_x       c[i][j] = sum_k dH[i][k] W[k][j]       A k-minor, B k-major (no transposed copy of W)
//     grad_W       c[i][j] = sum_r dH[r][i] x[r][j]       A k-major, B k-major, split over the rows r (fixed-order sum: reproducible)
// Mapping: a workgroup of WM x WN waves owns a (64 WM) x (128 WN) output tile, each wave 2 x 4 MFMA blocks of 32 x 32 (128
// accumulator registers).  K runs in stages of 32: both operand slabs are staged HBM -> registers -> LDS with 16-byte accesses,
// double-buffered, one barrier per stage, the next stage's global loads in flight under the current stage's 256 MFMAs per wave.
// LDS images: k-minor [row][32 k] with a row pitch of 36 dwords (a lane's ds_read_b128 brings the 4 k of four consecutive MFMA steps;
// pitch 36 keeps the 16-lane read groups on distinct banks), k-major [k][row] as in memory (one ds_read_b32 per MFMA step, lanes on
// consecutive banks).  The kernel is bound by the fp32 matrix pipe (64 cycles per MFMA and SIMD against 6 LDS reads per 32 MFMAs);
// operand traffic is (TI + TJ) x 4 bytes per 2 TI TJ flops = 64 flop/byte at 256 x 256, 2.4 TB/s at the fp32 MFMA peak.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.h"
#include "fitgnn_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBK = 32;     // k per stage
constexpr int kPitch = 36;  // dwords per row of a k-minor LDS image

// A workgroup of WM x WN waves, each wave MI x NJ MFMA blocks of 32 x 32: tile (32 MI WM) x (32 NJ WN).
template <int WM, int WN, int MI, int NJ, bool AKM, bool BKM>
struct Geo {
    static constexpr int TI = 32 * MI * WM, TJ = 32 * NJ * WN, THREADS = 64 * WM * WN;
    static constexpr int A_DW = AKM ? kBK * TI : TI * kPitch;
    static constexpr int B_DW = BKM ? kBK * TJ : TJ * kPitch;
    static constexpr int STAGE_DW = A_DW + B_DW;
    static constexpr int LDS_BYTES = 2 * STAGE_DW * 4;
    static constexpr int NA = TI * 8 / THREADS, NB = TJ * 8 / THREADS;   // float4 per thread and stage
    static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit the CU's LDS");
    static_assert(TI * 8 % THREADS == 0 && TJ * 8 % THREADS == 0, "stage not divisible over the threads");
};

// One operand slab of a stage, global -> registers.  T rows (k-minor) or T columns (k-major) starting at t0; rows / columns past
// `lim` are clamped to the last valid ones (they only feed outputs that are never stored), k past `klim` reads as zero.
template <bool KM, int T, int NV, int THREADS>
__device__ __forceinline__ void load_slab(float4 (&r)[NV], const float *__restrict__ P, long ld, long t0, long lim, long k0, long klim) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
        const int v = tid + p * THREADS;
        if (!KM) {
            const int row = v >> 3, k4 = v & 7;
            long i = t0 + row;
            i = i < lim ? i : lim - 1;
            const long k = k0 + 4 * k4;
            r[p] = k < klim ? *reinterpret_cast<const float4 *>(P + i * ld + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            constexpr int Q = T / 4;
            const int kr = v / Q, c4 = v % Q;
            long i = t0 + 4 * c4;
            i = i + 4 <= lim ? i : lim - 4;
            const long k = k0 + kr;
            r[p] = k < klim ? *reinterpret_cast<const float4 *>(P + k * ld + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

template <bool KM, int T, int NV, int THREADS>
__device__ __forceinline__ void store_slab(const float4 (&r)[NV], float *lds) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
        const int v = tid + p * THREADS;
        if (!KM) {
            const int row = v >> 3, k4 = v & 7;
            *reinterpret_cast<float4 *>(lds + row * kPitch + 4 * k4) = r[p];
        } else {
            constexpr int Q = T / 4;
            const int kr = v / Q, c4 = v % Q;
            *reinterpret_cast<float4 *>(lds + kr * T + 4 * c4) = r[p];
        }
    }
}

// The operand values of block row/column `o` (32 wide) for the four MFMA steps of k-octet kk: lane l holds index o + l % 32 and
// k = 8 kk + 4 (l / 32) + step.
template <bool KM, int T>
__device__ __forceinline__ void frag(float (&f)[4], const float *lds, int o, int kk, int lane) {
    if (!KM) {
        const float4 v = *reinterpret_cast<const float4 *>(lds + (o + (lane & 31)) * kPitch + 8 * kk + 4 * (lane >> 5));
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
        const float *p = lds + (8 * kk + 4 * (lane >> 5)) * T + o + (lane & 31);
#pragma unroll
        for (int s = 0; s < 4; ++s) f[s] = p[s * T];
    }
}

// nchunks > 1: split over k.  Workgroup (chunk, tile) reduces k in [chunk * chunk_k, +chunk_k) and stores its tile into
// partial[chunk] (an [I x J] matrix each); sum_chunks_kernel adds them in a fixed order.
template <int WM, int WN, int MI, int NJ, bool AKM, bool BKM, bool PIPE = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_f32_kernel(const float *__restrict__ A, long lda, const float *__restrict__ B, long ldb,
                                                                   long I, int J, long K, float *__restrict__ C, long ldc, int tiles_i,
                                                                   int tiles_j, int nchunks, long chunk_k) {
    using G = Geo<WM, WN, MI, NJ, AKM, BKM>;
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // block -> (chunk, tile): blocks are dealt round-robin over the 8 XCDs, so give the workgroups that share an operand slab
    // consecutive slots of ONE XCD (its L2 serves the second read): without a k split the J tiles of a row tile, with one the
    // tiles of a chunk
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int ti, tj, chunk = 0;
    if (nchunks > 1) {
        const int ntile = tiles_i * tiles_j;
        const int tile = slot % ntile;
        chunk = xcd + 8 * (slot / ntile);
        ti = tile / tiles_j, tj = tile % tiles_j;
        if (chunk >= nchunks) return;
    } else {
        tj = slot % tiles_j;
        ti = xcd + 8 * (slot / tiles_j);
        if (ti >= tiles_i) return;
    }
    const long i0 = (long)ti * G::TI, j0 = (long)tj * G::TJ;
    const long k_begin = (long)chunk * chunk_k;
    const long k_end = nchunks > 1 ? (k_begin + chunk_k < K ? k_begin + chunk_k : K) : K;
    const int nstage = k_end > k_begin ? (int)((k_end - k_begin + kBK - 1) / kBK) : 0;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[G::NA], rb[G::NB];
    if (nstage > 0) {
        load_slab<AKM, G::TI, G::NA, G::THREADS>(ra, A, lda, i0, I, k_begin, k_end);
        load_slab<BKM, G::TJ, G::NB, G::THREADS>(rb, B, ldb, j0, (long)J, k_begin, k_end);
        store_slab<AKM, G::TI, G::NA, G::THREADS>(ra, lds_f);
        store_slab<BKM, G::TJ, G::NB, G::THREADS>(rb, lds_f + G::A_DW);
    }
    __syncthreads();
    // The stage loop: the next stage's operand slabs are requested before a stage is multiplied and stored to the other LDS buffer
    // after it.  Two deeper forms were built on rocprofv3's "MFMA pipe busy 80 % of the SIMD cycles at 2.31 GHz" (tools/pmc_gemm.sh)
    // and measured out (tools/gemm_shape_probe.py, profiles/r04_gemm_shape_probe_*.log): (a) PIPE (opt-in, FITGNN_GEMM_DEEP=1 on the
    // small tile shapes) requests the next-but-ONE stage as well (two register sets used alternately: a small tile's stage is 1-2 us of
    // products per workgroup, about a global load's round trip): 0-10 % SLOWER on every shape (238 -> 253 us on the 165 000 x 512 x 128
    // table product, 113 -> 125 on 19 717 x 512 x 512); (b) a loop pipelined across its barrier (the last k-octet's fragments read
    // before it, multiplied after it under the next stage's first LDS reads): +-1 % -- removed.
    auto read_frags = [&](float (&fa)[MI][4], float (&fb)[NJ][4], const float *sa, const float *sb, int kk) {
#pragma unroll
        for (int i = 0; i < MI; ++i) frag<AKM, G::TI>(fa[i], sa, wm * (32 * MI) + i * 32, kk, lane);
#pragma unroll
        for (int j = 0; j < NJ; ++j) frag<BKM, G::TJ>(fb[j], sb, wn * (32 * NJ) + j * 32, kk, lane);
    };
    auto multiply = [&](const float (&fa)[MI][4], const float (&fb)[NJ][4]) {
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][st], fb[j][st], acc[i][j], 0, 0, 0);
    };
    constexpr int NKK = kBK / 8;
    auto products = [&](int s) {
        const float *sa = lds_f + (s & 1) * G::STAGE_DW;
        const float *sb = sa + G::A_DW;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            float fa[MI][4], fb[NJ][4];
            read_frags(fa, fb, sa, sb, kk);
            multiply(fa, fb);
        }
    };
    if (!PIPE) {
        for (int s = 0; s < nstage; ++s) {
            const bool more = s + 1 < nstage;   // workgroup-uniform
            if (more) {
                const long k0 = k_begin + (long)(s + 1) * kBK;
                load_slab<AKM, G::TI, G::NA, G::THREADS>(ra, A, lda, i0, I, k0, k_end);
                load_slab<BKM, G::TJ, G::NB, G::THREADS>(rb, B, ldb, j0, (long)J, k0, k_end);
            }
            products(s);
            if (more) {
                float *na = lds_f + ((s + 1) & 1) * G::STAGE_DW;
                store_slab<AKM, G::TI, G::NA, G::THREADS>(ra, na);
                store_slab<BKM, G::TJ, G::NB, G::THREADS>(rb, na + G::A_DW);
            }
            __syncthreads();
        }
    } else {
        float4 ra2[G::NA], rb2[G::NB];
        if (nstage > 1) {   // stage 1 on its way while stage 0 is multiplied
            load_slab<AKM, G::TI, G::NA, G::THREADS>(ra, A, lda, i0, I, k_begin + kBK, k_end);
            load_slab<BKM, G::TJ, G::NB, G::THREADS>(rb, B, ldb, j0, (long)J, k_begin + kBK, k_end);
        }
        // one stage: `cur` holds stage s + 1 (requested a stage ago), `nxt` receives stage s + 2
        auto body = [&](int s, float4 (&cur_a)[G::NA], float4 (&cur_b)[G::NB], float4 (&nxt_a)[G::NA], float4 (&nxt_b)[G::NB]) {
            if (s + 2 < nstage) {
                const long k0 = k_begin + (long)(s + 2) * kBK;
                load_slab<AKM, G::TI, G::NA, G::THREADS>(nxt_a, A, lda, i0, I, k0, k_end);
                load_slab<BKM, G::TJ, G::NB, G::THREADS>(nxt_b, B, ldb, j0, (long)J, k0, k_end);
            }
            products(s);
            if (s + 1 < nstage) {
                float *na = lds_f + ((s + 1) & 1) * G::STAGE_DW;
                store_slab<AKM, G::TI, G::NA, G::THREADS>(cur_a, na);
                store_slab<BKM, G::TJ, G::NB, G::THREADS>(cur_b, na + G::A_DW);
            }
            __syncthreads();
        };
        for (int s = 0; s < nstage; s += 2) {
            body(s, ra, rb, ra2, rb2);
            if (s + 1 < nstage) body(s + 1, ra2, rb2, ra, rb);
        }
    }

    // C/D layout of the 32 x 32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float *out = nchunks > 1 ? C + (long)chunk * I * J : C;
    const long ldo = nchunks > 1 ? (long)J : ldc;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const long col = j0 + wn * (32 * NJ) + j * 32 + (lane & 31);
            const long row0 = i0 + wm * (32 * MI) + i * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = row0 + (r & 3) + 8 * (r >> 2);
                if (row < I && col < J) out[row * ldo + col] = acc[i][j][r];
            }
        }
    }
}

// out = sum over chunks of partial[chunk] in a fixed order (eight running sums: eight loads in flight per lane)
__global__ __launch_bounds__(256) void sum_chunks_kernel(const float *__restrict__ partial, int nchunks, long IJ, float *__restrict__ out, int J,
                                                         long ldc) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= IJ) return;
    float part[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) part[u] = 0.f;
    int c = 0;
    for (; c + 8 <= nchunks; c += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) part[u] += partial[(long)(c + u) * IJ + q];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (c + u < nchunks) part[u] += partial[(long)(c + u) * IJ + q];
    float acc = part[0];
#pragma unroll
    for (int u = 1; u < 8; ++u) acc += part[u];
    out[(q / J) * ldc + (q % J)] = acc;
}

// Tile shapes.  256 x 256 (8 waves of 64 x 128) is the full-grid shape: one workgroup per CU (147 KB of LDS), 64 flop per operand
// byte.  Its grid is too coarse for the short operands of the configurations (S-pubmed's 19 717-row table: 154 tiles on 256 CUs; a
// 128-molecule QM9 batch: 38), so smaller tiles take over there: 128 x 128 (4 waves of 64 x 64, 74 KB: two workgroups per CU) and
// 64 x 128 (4 waves of 32 x 64, 55 KB).  64 x 512 serves the head's weight gradient (<= 64 padded class rows), 256 x 128 a narrow J.
enum Shape { S256 = 0, S256x128, S64x512, S128, S64x128, S128x64, S64x64 };
struct ShapeDim { int ti, tj, per_cu; };
constexpr ShapeDim kShape[] = {{256, 256, 1}, {256, 128, 1}, {64, 512, 1}, {128, 128, 2}, {64, 128, 2}, {128, 64, 2}, {64, 64, 4}};

struct Plan {
    int shape;           // Shape
    int tiles_i, tiles_j, nchunks;
    long chunk_k;
    // the last, mostly empty round of a long grid as a launch of its own (k-minor a only): rows [main_rows, I) split eight ways over k
    long main_rows;      // 0: one launch
    int rem_tiles_i, rem_chunks;
    long rem_chunk_k;
};

Plan make_plan(long I, int J, long K, bool akm, bool bkm) {
    Plan p;
    const bool tn = akm && bkm;
    auto tiles_of = [&](int sh) { return ((I + kShape[sh].ti - 1) / kShape[sh].ti) * ((J + kShape[sh].tj - 1) / kShape[sh].tj); };
    if (tn) {
        // split-k forms (the weight gradients): a few output tiles times many k chunks.  A short I (the head's <= 64 padded class rows)
        // takes 64 x 512; otherwise 128 x 128 tiles give the chunking four times the grid for the same partial traffic -- measured
        // faster up to k ~ 20 000 (42 vs 60 us at k = 4 861, 102 vs 121 at 19 717) and for a narrow J (204 vs 214 us on the 100-column
        // table at k = 165 000), slower beyond (744 vs 674 us at 512 x 512 x 165 000: 256 x 256 reads each operand row half as often)
        if (I <= 64) p.shape = S64x512;
        else if (J <= 128 || K < 65536) p.shape = S128;
        else p.shape = S256;
    } else if (J <= 64) {
        p.shape = S128x64;   // the head on the loss rows (47 classes): a 128-column tile multiplied mostly padding
    } else if (getenv("FITGNN_GEMM_NO_SMALL_TILES")) {
        p.shape = J <= 128 ? S256x128 : S256;
    } else {
        // One CU works through ceil(tiles / 256) tiles of TI x TJ outputs each, whatever shares the CU meanwhile; the smaller shapes
        // lose a little per flop (operands re-read more often; 128 x 128 does not fill its third round evenly) and win whenever the
        // 256 x 256 grid is coarse: 19 717 rows (154 tiles) 146 -> 111 us, 4 861 rows (38 tiles) 136 -> 45 us, a rank's 20 625 loss rows
        // 135 -> 120 us; from ~90 000 rows on 256 x 256 is the fastest again (414 vs 423 / 440 us)  [tools/gemm_shape_probe.py]
        // (64 x 64: a 128-molecule QM9 batch, 4 861 rows = 304 tiles of 64 x 128, still leaves every fifth CU with two tiles)
        const int cand[4] = {J <= 128 ? S256x128 : S256, S128, S64x128, S64x64};
        const double eff[4] = {1.0, K <= 256 ? 1.0 : 0.85, 0.94, 0.85};   // short k: two workgroups per CU hide each other's stores
        double best_cost = 0;
        for (int q = 0; q < 4; ++q) {
            const int sh = cand[q];
            const long t = tiles_of(sh);
            double rounds = (double)((t + 255) / 256);
            if (q == 0 && !akm && t / 256 >= 1 && t % 256 > 0 && t % 256 <= 96 && K >= 8 * 2 * kBK) rounds = (double)(t / 256) + 0.15;   // tail launch
            const double cost = rounds * kShape[sh].ti * kShape[sh].tj / eff[q];
            if (q == 0 || cost < best_cost) { best_cost = cost; p.shape = sh; }
        }
        // a long k on a grid of a round or two is split over k below: 128 x 128 tiles give the chunking four times the workgroups
        // for the same partial traffic (S-physics' layer 0, 34 493 x 512 x 8 448: 2.57 ms against 2.67-2.82 with 256 x 256 tiles)
        if (K >= 4096 && tiles_of(S256) <= 512 && J > 128) p.shape = S128;
    }
    if (const char *force = getenv("FITGNN_GEMM_SHAPE")) {   // experiments
        const int f = atoi(force);
        if (f >= 0 && f <= S64x64 && !(f == S64x512 && !tn)) p.shape = f;
    }
    const int TI = kShape[p.shape].ti, TJ = kShape[p.shape].tj;
    const double slots = 256.0 * kShape[p.shape].per_cu;
    p.tiles_i = (int)((I + TI - 1) / TI);
    p.tiles_j = (J + TJ - 1) / TJ;
    p.nchunks = 1;
    p.chunk_k = K;
    // Split over k where that shortens the launch.  A launch takes ceil(workgroups / slots) rounds of one tile's time each, so 270
    // full-k tiles (S-physics' layer 0: 34 493 rows) take two rounds with the second one 5 % full, and 66 tiles leave 190 CUs idle.
    // c chunks make the tiles c times shorter at the price of c + 1 passes over an [I x J] partial matrix; the model below -- fp32
    // MFMA at ~0.45 TFLOP/s per CU as measured, partials at 3 TB/s -- picks c.
    const double ntile = (double)p.tiles_i * p.tiles_j;
    const double t_tile = 2.0 * TI * TJ * (double)K / (0.45e12 / kShape[p.shape].per_cu);
    const long most = K / (4 * kBK);   // at least four stages per chunk
    // chunk counts are multiples of 8: a tile's chunks sit on consecutive block ids = one per XCD (block -> (chunk, tile) below), so
    // any other count leaves XCDs idle (measured: 2 chunks of the 34 493 x 8 448 product took 10.0 ms against 4.6 unsplit, 8: 2.8)
    double best = ceil(ntile / slots) * t_tile;
    for (long c = 8; c <= most && c <= 256; c += 8) {
        const double t = ceil(ntile * (double)c / slots) * t_tile / (double)c + (double)(c + 1) * (double)I * (double)J * 4.0 / 3.0e12 + 4e-6;
        if (t < 0.92 * best) { best = t; p.nchunks = (int)c; }
    }
    if (const char *force = getenv("FITGNN_GEMM_CHUNKS")) {   // experiments: a fixed chunk count (clamped to what K allows)
        const long c = atol(force);
        p.nchunks = (int)(c <= 1 ? 1 : (c + 7) / 8 * 8);
    }
    if (p.nchunks > 1) {
        const long per = (K + p.nchunks - 1) / p.nchunks;
        p.chunk_k = (per + kBK - 1) / kBK * kBK;   // (a last chunk may come out empty: it stores zeros)
    }
    // The tail of a long grid: 1 290 tiles (165 000 rows x 512 columns) are 5.04 rounds of 256 workgroups, and the sixth round keeps 10
    // CUs busy for a whole tile's time.  With the row-major operand on the a side the rows of that last round can be a launch of
    // their own, split eight ways over k (80 workgroups of an eighth of a tile each, their 2-MB partials summed in a fixed order):
    // 5 + 1/8 rounds instead of 6.  Not for a grid of less than one full round (its partials would be the whole output).
    p.main_rows = 0; p.rem_tiles_i = 0; p.rem_chunks = 1; p.rem_chunk_k = K;
    const long T = (long)p.tiles_i * p.tiles_j;
    const long full = T / 256, rem = T % 256;
    if (!akm && kShape[p.shape].per_cu == 1 && p.nchunks == 1 && full >= 1 && rem > 0 && rem <= 96 && K >= 8 * 2 * kBK &&
        !getenv("FITGNN_GEMM_CHUNKS") && !getenv("FITGNN_GEMM_NO_TAIL")) {
        const long main_tiles_i = full * 256 / p.tiles_j;
        if (main_tiles_i >= 1 && main_tiles_i < p.tiles_i) {
            p.main_rows = main_tiles_i * TI;
            p.rem_tiles_i = (int)(p.tiles_i - main_tiles_i);
            p.rem_chunks = 8;
            p.rem_chunk_k = ((K + 7) / 8 + kBK - 1) / kBK * kBK;
        }
    }
    return p;
}

template <int WM, int WN, int MI, int NJ, bool AKM, bool BKM, bool PIPE = false>
int launch(const Plan &p, const float *a, long lda, const float *b, long ldb, long I, int J, long K, float *c, long ldc, hipStream_t s) {
    using G = Geo<WM, WN, MI, NJ, AKM, BKM>;
    static std::atomic<uint64_t> lds_done{0};
    if (const int rc = fitgnn_lds_limit_once((const void *)gemm_f32_kernel<WM, WN, MI, NJ, AKM, BKM, PIPE>, G::LDS_BYTES, lds_done)) return rc;
    unsigned grid;
    if (p.nchunks > 1) grid = (unsigned)(p.tiles_i * p.tiles_j * ((p.nchunks + 7) / 8 * 8));   // chunk = xcd + 8 * (slot / tiles)
    else grid = (unsigned)((p.tiles_i + 7) / 8 * 8 * p.tiles_j);
    hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, MI, NJ, AKM, BKM, PIPE>), dim3(grid), dim3(G::THREADS), G::LDS_BYTES, s, a, lda, b, ldb, I, J, K,
                       c, ldc, p.tiles_i, p.tiles_j, p.nchunks, p.chunk_k);
    return (int)hipGetLastError();
}

template <bool AKM, bool BKM>
int launch_shape(const Plan &p, const float *a, long lda, const float *b, long ldb, long I, int J, long K, float *c, long ldc, hipStream_t s) {
    if constexpr (AKM && BKM) {   // make_plan picks the 64 x 512 tile for this pair only (its k-minor LDS image would not fit)
        if (p.shape == S64x512) return launch<1, 4, 2, 4, AKM, BKM>(p, a, lda, b, ldb, I, J, K, c, ldc, s);
    }
    switch (p.shape) {
        case S256x128: return launch<4, 1, 2, 4, AKM, BKM>(p, a, lda, b, ldb, I, J, K, c, ldc, s);
        case S128:
            if (getenv("FITGNN_GEMM_DEEP")) return launch<2, 2, 2, 2, AKM, BKM, true>(p, a, lda, b, ldb, I, J, K, c, ldc, s);
            return launch<2, 2, 2, 2, AKM, BKM, false>(p, a, lda, b, ldb, I, J, K, c, ldc, s);
        case S64x128:
            if (getenv("FITGNN_GEMM_DEEP")) return launch<2, 2, 1, 2, AKM, BKM, true>(p, a, lda, b, ldb, I, J, K, c, ldc, s);
            return launch<2, 2, 1, 2, AKM, BKM, false>(p, a, lda, b, ldb, I, J, K, c, ldc, s);
        case S128x64:
            if (getenv("FITGNN_GEMM_DEEP")) return launch<4, 1, 1, 2, AKM, BKM, true>(p, a, lda, b, ldb, I, J, K, c, ldc, s);
            return launch<4, 1, 1, 2, AKM, BKM, false>(p, a, lda, b, ldb, I, J, K, c, ldc, s);
        case S64x64:
            if (getenv("FITGNN_GEMM_DEEP")) return launch<2, 2, 1, 1, AKM, BKM, true>(p, a, lda, b, ldb, I, J, K, c, ldc, s);
            return launch<2, 2, 1, 1, AKM, BKM, false>(p, a, lda, b, ldb, I, J, K, c, ldc, s);
        default: return launch<4, 2, 2, 4, AKM, BKM>(p, a, lda, b, ldb, I, J, K, c, ldc, s);
    }
}

}  // namespace

extern "C" size_t fitgnn_gemm_exact_workspace_bytes(int64_t I, int32_t J, int64_t K, int32_t a_kmajor, int32_t b_kmajor) {
    if (I <= 0 || J <= 0 || K <= 0) return 0;
    const Plan p = make_plan((long)I, J, (long)K, a_kmajor != 0, b_kmajor != 0);
    if (p.main_rows > 0) return (size_t)p.rem_chunks * (size_t)(I - p.main_rows) * (size_t)J * sizeof(float);
    return p.nchunks > 1 ? (size_t)p.nchunks * (size_t)I * (size_t)J * sizeof(float) : 0;
}

extern "C" int fitgnn_gemm_exact_f32(const float *a, int64_t lda, int32_t a_kmajor, const float *b, int64_t ldb, int32_t b_kmajor,
                                     int64_t I, int32_t J, int64_t K, float *c, int64_t ldc, void *workspace, void *stream) {
    if (I <= 0 || J <= 0 || K <= 0 || ldc < J) return FITGNN_E_BADARG;
    if (!a || !b || !c) return FITGNN_E_BADARG;
    // 16-byte accesses along the contiguous dimension of each operand
    const int64_t a_inner = a_kmajor ? I : K, b_inner = b_kmajor ? (int64_t)J : K;
    if (a_inner < 4 || b_inner < 4 || (a_inner % 4) != 0 || (b_inner % 4) != 0 || lda < a_inner || ldb < b_inner || (lda % 4) != 0 ||
        (ldb % 4) != 0)
        return FITGNN_E_BADARG;
    if ((((uintptr_t)a | (uintptr_t)b) % 16) != 0) return FITGNN_E_ALIGN;
    if (a_kmajor && !b_kmajor) return FITGNN_E_BADARG;   // no caller: (k-major, k-minor) is the transpose of (k-minor, k-major)
    const Plan p = make_plan((long)I, J, (long)K, a_kmajor != 0, b_kmajor != 0);
    hipStream_t s = (hipStream_t)stream;
    if (p.main_rows > 0) {   // (k-minor a) the full rounds, then the rows of the last round split over k
        if (!workspace || ((uintptr_t)workspace % 16) != 0) return FITGNN_E_BADARG;
        Plan pm = p, pr = p;
        pm.tiles_i = (int)(p.main_rows / kShape[p.shape].ti);
        pr.tiles_i = p.rem_tiles_i; pr.nchunks = p.rem_chunks; pr.chunk_k = p.rem_chunk_k;
        const long rem_rows = (long)I - p.main_rows;
        const float *a_rem = a + p.main_rows * (long)lda;
        int rc;
        if (b_kmajor) {
            rc = launch_shape<false, true>(pm, a, (long)lda, b, (long)ldb, p.main_rows, J, (long)K, c, (long)ldc, s);
            if (!rc) rc = launch_shape<false, true>(pr, a_rem, (long)lda, b, (long)ldb, rem_rows, J, (long)K, (float *)workspace, (long)J, s);
        } else {
            rc = launch_shape<false, false>(pm, a, (long)lda, b, (long)ldb, p.main_rows, J, (long)K, c, (long)ldc, s);
            if (!rc) rc = launch_shape<false, false>(pr, a_rem, (long)lda, b, (long)ldb, rem_rows, J, (long)K, (float *)workspace, (long)J, s);
        }
        if (rc) return rc;
        const long IJ = rem_rows * J;
        hipLaunchKernelGGL(sum_chunks_kernel, dim3((unsigned)((IJ + 255) / 256)), dim3(256), 0, s, (const float *)workspace, pr.nchunks, IJ,
                           c + p.main_rows * (long)ldc, J, (long)ldc);
        return (int)hipGetLastError();
    }
    float *dst = c;
    long ldd = (long)ldc;
    if (p.nchunks > 1) {
        if (!workspace || ((uintptr_t)workspace % 16) != 0) return FITGNN_E_BADARG;
        dst = (float *)workspace;
    }
    int rc;
    if (a_kmajor) rc = launch_shape<true, true>(p, a, (long)lda, b, (long)ldb, (long)I, J, (long)K, dst, ldd, s);
    else if (b_kmajor) rc = launch_shape<false, true>(p, a, (long)lda, b, (long)ldb, (long)I, J, (long)K, dst, ldd, s);
    else rc = launch_shape<false, false>(p, a, (long)lda, b, (long)ldb, (long)I, J, (long)K, dst, ldd, s);
    if (rc) return rc;
    if (p.nchunks > 1) {
        const long IJ = (long)I * J;
        hipLaunchKernelGGL(sum_chunks_kernel, dim3((unsigned)((IJ + 255) / 256)), dim3(256), 0, s, (const float *)workspace, p.nchunks, IJ, c, J,
                           (long)ldc);
        return (int)hipGetLastError();
    }
    return 0;
}
