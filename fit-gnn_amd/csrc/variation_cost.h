// variation_cost.h -- one wavefront computes the local-variation cost of one candidate set.
//
// Restates subgraph_cost (FIT-GNN graph_coarsening/coarsening_utils.py:555-561)
//     cost(S) = || B^T L_S B ||_F / (nc-1),  B = (I - 11^T/nc) A[S,:],  L_S = diag(2 dw[S] - W_S 1) - W_S
// in the CANONICAL ARITHMETIC of DESIGN.md (binary64, one rounding per operation, no FMA):
//     mean[k] = (A[S0][k] + A[S1][k] + ...)/nc           left-to-right over the sorted members
//     B[a][k] = A[Sa][k] - mean[k]
//     rs[a]   = sum_b w_ab ; T[a][l] = sum_b (w_ab * B[b][l])    b ascending over S_b in adj(S_a), from 0.0
//     Y[a][l] = (2 dw[Sa] - rs[a]) * B[a][l] - T[a][l]
//     M[k][l] = sum_a (B[a][k] * Y[a][l])                        a ascending, from 0.0
//     p[j]    = sum over e = j, j+64, ... < K*K of M[e]^2 ; 64-lane butterfly (offsets 32..1) ; sqrt ; / (nc-1)
// Lane mapping: rows a of S on lanes (tiles of 64 rows) for B/Y, matrix entries (k,l) on lanes for M and
// the Frobenius sum -- so the wave-wide reduction IS the canonical summation tree.
//
// This translation unit must be compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "fitgnn_hip.h"

// Orders this wave's LDS traffic: later reads by any lane see earlier writes by every lane (the LDS
// pipeline is in-order per wave; the fence stops the compiler from moving accesses across it).
#define FITGNN_WAVE_SYNC()                                      \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
    } while (0)

namespace fitgnn {

constexpr int kCostTile = 64;  // rows of S per LDS tile (one per lane)

constexpr int kHitCap = 512;   // matches (adjacency entries inside the set) listed in LDS per set; more -> serial path

struct CostLds {  // per-wave LDS scratch
    double B[kCostTile * FITGNN_MAX_K];
    double Y[kCostTile * FITGNN_MAX_K];
    double mean[FITGNN_MAX_K];
    double hit_w[kHitCap];        // weight of every match, in (row, column) order
    int32_t S[kCostTile];
    int32_t rstart[kCostTile + 1];  // first flattened adjacency index of every member row
    int32_t re0[kCostTile];         // rowptr of every member
    int32_t cnt[kCostTile];         // matches per member row
    uint8_t hit_b[kHitCap];         // position in S of every match
    uint8_t hit_a[kHitCap];         // ... and the member row it belongs to (what a cached list is filtered by)
    uint8_t remap[kCostTile];       // greedy selection: position of an old member in the pruned set, 255 = dropped
};

// A set's list of matches (W_S in (row, column) order) can be kept between two costings of the same, shrinking set: the greedy
// selection re-costs a re-inserted set every time a member of it is marked, and the matches of the pruned set are the old ones
// minus those that touch a dropped member -- no adjacency list has to be scanned again (a 50-member set of degree-50 nodes: 2 500
// entries, ten rounds of dependent look-ups, 45 us; 91 % of the S-products selection).  Same list, same order: same arithmetic.
struct HitIO {
    uint16_t *pool_ab;   // (row << 8) | column per match, positions at the time of storing
    double *pool_w;      // weights (only used for a weighted graph)
    int64_t *bump;       // next free pool entry (owned by the one selecting wave)
    int64_t pool_cap;
    int64_t off;         // this set's list: pool offset (-1: none yet), in: number of stored matches (-1: none, scan)
    int32_t n;           // out: number of matches now stored (-1: the list did not fit / was not built)
    bool use_remap;      // the stored positions are to be translated through lds.remap
};

struct CostGraph {
    const int32_t *rowptr;
    const int32_t *col;
    const double *w;  // may be null: all ones
    const double *dw;
    const double *A;
    int32_t K;
    int64_t lda;
    const int32_t *node_K;  // optional: per-node column count (block-diagonal batches whose components have
                            // fewer than K spectral columns: a component of N <= K nodes has N, coarsening_utils.py:85-86)
};

// first index in sorted a[0..n) with a[i] >= v
__device__ __forceinline__ int lower_bound_i32(const int32_t *a, int n, int32_t v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// All 64 lanes of the wave must call this with identical (g, S, nc).  Returns the cost in every lane.
// S points to global memory (sorted ascending).  lds is this wave's private scratch.
// STAGED: the caller has already put the members of a set of at most kCostTile nodes into lds.S (and synchronised the wave),
// so nothing is read back from S -- the greedy selection re-costs a set it has just compacted in registers.
template <bool STAGED = false>
__device__ inline double set_cost_wave(const CostGraph &g, const int32_t *__restrict__ S, int nc, CostLds &lds, HitIO *io = nullptr) {
#pragma clang fp contract(off)
    if (nc < 2) return INFINITY;
    const int lane = threadIdx.x & 63;
    const int K = g.node_K ? __builtin_amdgcn_readfirstlane(g.node_K[STAGED ? lds.S[0] : S[0]]) : g.K;
    const int KK = K * K;
    const bool small = nc <= kCostTile;  // whole set resident in LDS
    // ---- pass 1: column means, sequential over members (tiles of 64 rows staged cooperatively) ----
    double msum = 0.0;
    int pre_e0 = 0, pre_deg = 0;   // small sets: the member's row extent and degree term ride along with the A gather
    double pre_dw = 0.0;
    for (int t0 = 0; t0 < nc; t0 += kCostTile) {
        const int rows = min(kCostTile, nc - t0);
        if (lane < rows) {
            int32_t u;
            if (STAGED) u = lds.S[lane]; else { u = S[t0 + lane]; lds.S[lane] = u; }
            if (small) { pre_e0 = g.rowptr[u]; pre_deg = g.rowptr[u + 1] - pre_e0; pre_dw = g.dw[u]; }
        }
        FITGNN_WAVE_SYNC();
        {   // the tile's rows of A: every load requested before the first is stored (one memory round trip, not one per 64 entries)
            constexpr int kIters = kCostTile * FITGNN_MAX_K / 64;
            double av[kIters];
#pragma unroll
            for (int q = 0; q < kIters; ++q) {
                const int i = lane + 64 * q;
                av[q] = 0.0;
                if (i < rows * K) { const int a = i / K, k = i - a * K; av[q] = g.A[(int64_t)lds.S[a] * g.lda + k]; }
            }
#pragma unroll
            for (int q = 0; q < kIters; ++q) {
                const int i = lane + 64 * q;
                if (i < rows * K) lds.B[i] = av[q];
            }
        }
        FITGNN_WAVE_SYNC();
        if (lane < K) {
            int a = 0;
            if (t0 == 0) { msum = lds.B[lane]; a = 1; }
            for (; a < rows; ++a) msum = msum + lds.B[a * K + lane];
        }
        FITGNN_WAVE_SYNC();
    }
    if (lane < K) lds.mean[lane] = msum / (double)nc;
    FITGNN_WAVE_SYNC();

    // ---- pass 2: per tile, rows on lanes -> B, Y ; then entries on lanes -> M ----
    double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0;
    const int ea = lane, eb = lane + 64, ec = lane + 128, ed = lane + 192;
    for (int t0 = 0; t0 < nc; t0 += kCostTile) {
        const int rows = min(kCostTile, nc - t0);
        if (!small || t0 > 0) {  // (re)stage this tile's raw rows; the small case still holds them
            if (lane < rows) lds.S[lane] = S[t0 + lane];
            FITGNN_WAVE_SYNC();
            for (int i = lane; i < rows * K; i += 64) {
                const int a = i / K, k = i - a * K;
                lds.B[i] = g.A[(int64_t)lds.S[a] * g.lda + k];
            }
            FITGNN_WAVE_SYNC();
        }
        for (int i = lane; i < rows * K; i += 64) {
            const int a = i / K, k = i - a * K;
            lds.B[i] = lds.B[i] - lds.mean[k];
        }
        FITGNN_WAVE_SYNC();
        // W_S rows: the matches adj(u) /\ S of every member u, each row in ascending column order.
        // Fast path (whole set in one tile): all 64 lanes scan the members' adjacency lists TOGETHER -- one flattened
        // index space, independent loads -- and list the matches (position in S, weight) in LDS in (row, column)
        // order; each member's lane then folds ITS matches in that same order, so the arithmetic is the serial walk's.
        bool listed = false;
        int nh_listed = 0;
        if (small && io && io->off >= 0 && io->n >= 0) {
            // ---- the stored list of this set, minus the matches that touch a dropped member ----
            lds.cnt[lane] = 0;
            FITGNN_WAVE_SYNC();
            int nh = 0;
            constexpr int kChunks = kHitCap / 64;
            uint32_t ab_q[kChunks];
            double w_q[kChunks];
#pragma unroll
            for (int q = 0; q < kChunks; ++q) {   // the whole list requested at once
                const int i = q * 64 + lane;
                ab_q[q] = i < io->n ? io->pool_ab[io->off + i] : 0u;
                w_q[q] = (i < io->n && g.w) ? io->pool_w[io->off + i] : 1.0;
            }
#pragma unroll
            for (int q = 0; q < kChunks; ++q) {
                const int base = q * 64;
                if (base >= io->n) break;   // wave-uniform
                const int i = base + lane;
                const bool valid = i < io->n;
                const uint32_t ab = ab_q[q];
                const double wv = w_q[q];
                int a = (int)(ab >> 8), b = (int)(ab & 255u);
                if (io->use_remap) { a = lds.remap[a]; b = lds.remap[b]; }
                const bool hit = valid && a != 255 && b != 255;
                const unsigned long long bal = __ballot(hit);
                if (hit) {
                    const int at = nh + __popcll(bal & ((1ull << lane) - 1ull));
                    lds.hit_a[at] = (uint8_t)a;
                    lds.hit_b[at] = (uint8_t)b;
                    lds.hit_w[at] = wv;
                    atomicAdd(&lds.cnt[a], 1);
                }
                nh += __popcll(bal);
            }
            listed = true;
            nh_listed = nh;
            FITGNN_WAVE_SYNC();
        }
        if (small && !listed) {
            const int e0 = pre_e0, deg = pre_deg;
            int incl = deg;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(incl, off, 64); if (lane >= off) incl += y; }
            const int total = __shfl(incl, 63, 64);
            lds.rstart[lane] = incl - deg;
            if (lane == 63) lds.rstart[64] = total;
            lds.re0[lane] = e0;
            lds.cnt[lane] = 0;
            FITGNN_WAVE_SYNC();
            int nh = 0;
            listed = true;
            // four 64-entry chunks per round: their column (and weight) loads are issued together, then folded in order
            for (int base = 0; base < total && listed; base += 256) {
                int a4[4], e4[4];
                int32_t c4[4];
                double w4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a4[q] = 0; e4[q] = 0;
                    if (q > 0 && base + q * 64 >= total) continue;  // wave-uniform: most re-costed sets fit one chunk
                    // clamped: every lane loads a valid entry (total >= 1 here), dead ones are masked below
                    const int ii = min(base + q * 64 + lane, total - 1);
                    int lo = 0, hi = rows;             // row a = last row whose first index is <= ii
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (lds.rstart[mid] <= ii) lo = mid; else hi = mid; }
                    a4[q] = lo;
                    e4[q] = lds.re0[lo] + (ii - lds.rstart[lo]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    c4[q] = 0; w4[q] = 1.0;
                    if (q > 0 && base + q * 64 >= total) continue;
                    c4[q] = g.col[e4[q]];
                    if (g.w) w4[q] = g.w[e4[q]];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (base + q * 64 >= total) break;  // wave-uniform
                    const bool valid = base + q * 64 + lane < total;
                    const int b = lower_bound_i32(lds.S, nc, c4[q]);
                    const bool hit = valid && b < nc && lds.S[b] == c4[q];
                    const unsigned long long bal = __ballot(hit);
                    const int nb = __popcll(bal);
                    if (nh + nb > kHitCap) { listed = false; break; }  // wave-uniform
                    if (hit) {
                        const int at = nh + __popcll(bal & ((1ull << lane) - 1ull));
                        lds.hit_a[at] = (uint8_t)a4[q];
                        lds.hit_b[at] = (uint8_t)b;
                        lds.hit_w[at] = w4[q];
                        atomicAdd(&lds.cnt[a4[q]], 1);
                    }
                    nh += nb;
                }
            }
            nh_listed = nh;
            FITGNN_WAVE_SYNC();
        }
        if (small) {
            if (io) {   // keep the list for the set's next costing (in place when it was read from the pool: it only shrinks)
                io->n = -1;
                if (listed) {
                    if (io->off < 0 && *io->bump + nh_listed <= io->pool_cap) { io->off = *io->bump; *io->bump += nh_listed; }
                    if (io->off >= 0) {
                        for (int i = lane; i < nh_listed; i += 64) {
                            io->pool_ab[io->off + i] = (uint16_t)(((uint32_t)lds.hit_a[i] << 8) | (uint32_t)lds.hit_b[i]);
                            if (g.w) io->pool_w[io->off + i] = lds.hit_w[i];
                        }
                        io->n = nh_listed;
                    }
                }
            }
            if (listed) {
                const int cn = lane < rows ? lds.cnt[lane] : 0;
                int hincl = cn;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(hincl, off, 64); if (lane >= off) hincl += y; }
                if (lane < rows) {
                    const int a = lane;
                    double T[FITGNN_MAX_K];
#pragma unroll
                    for (int l = 0; l < FITGNN_MAX_K; ++l) T[l] = 0.0;
                    double rs = 0.0;
                    for (int h = hincl - cn; h < hincl; ++h) {
                        const int b = lds.hit_b[h];
                        const double wab = lds.hit_w[h];
                        rs = rs + wab;
#pragma unroll
                        for (int l = 0; l < FITGNN_MAX_K; ++l)
                            if (l < K) { const double prod = wab * lds.B[b * K + l]; T[l] = T[l] + prod; }
                    }
                    const double d = 2.0 * pre_dw - rs;
#pragma unroll
                    for (int l = 0; l < FITGNN_MAX_K; ++l)
                        if (l < K) { const double prod = d * lds.B[a * K + l]; lds.Y[a * K + l] = prod - T[l]; }
                }
            }
        }
        if (!listed)
        if (lane < rows) {
            const int a = lane;
            const int32_t u = lds.S[a];
            const int e0 = g.rowptr[u], e1 = g.rowptr[u + 1];
            const int deg = e1 - e0;
            double T[FITGNN_MAX_K];
#pragma unroll
            for (int l = 0; l < FITGNN_MAX_K; ++l) T[l] = 0.0;
            double rs = 0.0;
            // intersection adj(u) /\ S in ascending order, walking the shorter list
            const bool walk_adj = deg <= 4 * nc;
            const int steps = walk_adj ? deg : nc;
            for (int s = 0; s < steps; ++s) {
                int e, b;
                int32_t c;
                if (walk_adj) {
                    e = e0 + s;
                    c = g.col[e];
                    b = small ? lower_bound_i32(lds.S, nc, c) : lower_bound_i32(S, nc, c);
                    const int32_t sb = (b < nc) ? (small ? lds.S[b] : S[b]) : -1;
                    if (sb != c) continue;
                } else {
                    b = s;
                    c = small ? lds.S[b] : S[b];
                    e = e0 + lower_bound_i32(g.col + e0, deg, c);
                    if (e >= e1 || g.col[e] != c) continue;
                }
                const double wab = g.w ? g.w[e] : 1.0;
                rs = rs + wab;
                if (small) {
#pragma unroll
                    for (int l = 0; l < FITGNN_MAX_K; ++l)
                        if (l < K) { const double prod = wab * lds.B[b * K + l]; T[l] = T[l] + prod; }
                } else {
#pragma unroll
                    for (int l = 0; l < FITGNN_MAX_K; ++l)
                        if (l < K) {
                            const double bb = g.A[(int64_t)c * g.lda + l] - lds.mean[l];
                            const double prod = wab * bb;
                            T[l] = T[l] + prod;
                        }
                }
            }
            const double d = 2.0 * g.dw[u] - rs;
#pragma unroll
            for (int l = 0; l < FITGNN_MAX_K; ++l)
                if (l < K) { const double prod = d * lds.B[a * K + l]; lds.Y[a * K + l] = prod - T[l]; }
        }
        FITGNN_WAVE_SYNC();
        // M[k][l] += B[a][k] * Y[a][l], a ascending
        {
            const int ka = ea / K, la = ea - ka * K;
            const int kb = eb / K, lb = eb - kb * K;
            const int kc = ec / K, lc = ec - kc * K;
            const int kd = ed / K, ld = ed - kd * K;
            for (int a = 0; a < rows; ++a) {
                const double *Br = lds.B + a * K, *Yr = lds.Y + a * K;
                if (ea < KK) { const double prod = Br[ka] * Yr[la]; m0 = m0 + prod; }
                if (eb < KK) { const double prod = Br[kb] * Yr[lb]; m1 = m1 + prod; }
                if (ec < KK) { const double prod = Br[kc] * Yr[lc]; m2 = m2 + prod; }
                if (ed < KK) { const double prod = Br[kd] * Yr[ld]; m3 = m3 + prod; }
            }
        }
        FITGNN_WAVE_SYNC();
    }
    // ---- Frobenius norm: canonical 64-lane tree ----
    double p = 0.0;
    if (ea < KK) p = m0 * m0;
    if (eb < KK) { const double q = m1 * m1; p = p + q; }
    if (ec < KK) { const double q = m2 * m2; p = p + q; }
    if (ed < KK) { const double q = m3 * m3; p = p + q; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double other = __shfl_down(p, off, 64);
        p = p + other;
    }
    p = __shfl(p, 0, 64);
    const double res = sqrt(p) / (double)(nc - 1);
    return res;
}

}  // namespace fitgnn
