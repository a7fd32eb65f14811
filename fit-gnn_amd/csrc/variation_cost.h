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
    int32_t cnt[kCostTile];         // matches per member row
    uint8_t hit_b[kHitCap];         // position in S of every match
    uint8_t hit_a[kHitCap];         // ... and the member row it belongs to (what a cached list is filtered by)
    uint8_t remap[kCostTile];       // greedy selection: position of an old member in the pruned set, 255 = dropped
};

// A set's list of matches (W_S in (row, column) order) can be kept between two costings of the same, shrinking set: the greedy
// selection re-costs a re-inserted set every time a member of it is marked, and the matches of the pruned set are the old ones
// minus those that touch a dropped member -- no adjacency list has to be scanned again (a 50-member set of degree-50 nodes: 2 500
// entries, ten rounds of dependent look-ups, 45 us; 91 % of the S-products selection).  Same list, same order: same arithmetic.
struct HitIO {           // passed by reference and never through a pointer that may be null: it has to stay in registers (a HitIO
                         // in scratch cost a memory round trip per field read, 3/4 of a re-cost)
    bool active;         // false: no list is kept for this costing
    uint16_t *pool_ab;   // (row << 8) | column per match, positions at the time of storing
    double *pool_w;      // weights (only used for a weighted graph)
    int64_t bump;        // in/out: next free pool entry of the calling wave's region
    int64_t pool_cap;
    int64_t off;         // this set's list: pool offset (-1: none yet), in: number of stored matches (-1: none, scan)
    int32_t n;           // out: number of matches now stored (-1: the list did not fit / was not built)
    bool use_remap;      // the stored positions are to be translated through lds.remap
    bool fresh_out;      // store the new list in a fresh pool region (a helper wave's speculative result must not touch the set's own)
};

#ifdef FITGNN_GREEDY_STAMPS   // phases of a re-cost on the selecting wave (tools/greedy_stamps.py)
static __device__ unsigned long long g_cost_dbg[12];
#define CSTAMP(i) do { if (io.active && io.use_remap && !io.fresh_out) { const unsigned long long now_ = __builtin_readcyclecounter(); \
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_cost_dbg[i], now_ - cst_); cst_ = now_; } } while (0)
#define CSTAMP_DECL unsigned long long cst_ = __builtin_readcyclecounter()
#else
#define CSTAMP(i)
#define CSTAMP_DECL
#endif

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
// KC: the number of spectral columns when it is known at compile time (0: read at run time).  The per-column loops below are
// written over FITGNN_MAX_K with an `l < K` guard; with K only known at run time every guarded column becomes its own branch with
// its own LDS round trip (ten dependent waits per match), with KC they are straight-line code and i / K is a multiplication.
template <bool STAGED, int KC>
__device__ inline double set_cost_wave_k(const CostGraph &g, const int32_t *__restrict__ S, int nc, CostLds &lds, HitIO &io, int K_rt) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int K = KC > 0 ? KC : K_rt;
    const int KK = K * K;
    const bool small = nc <= kCostTile;  // whole set resident in LDS
    CSTAMP_DECL;
#ifdef FITGNN_GREEDY_STAMPS
    int nh_cost_ = 0;
#endif
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
        CSTAMP(0);
        if (lane < K) {
            int a = 0;
            if (t0 == 0) { msum = lds.B[lane]; a = 1; }
#pragma unroll 8   // eight rows' LDS reads in flight; the additions stay in row order
            for (; a < rows; ++a) msum = msum + lds.B[a * K + lane];
        }
        FITGNN_WAVE_SYNC();
    }
    if (lane < K) lds.mean[lane] = msum / (double)nc;
    FITGNN_WAVE_SYNC();
    CSTAMP(1);

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
        CSTAMP(2);
        // W_S rows: the matches adj(u) /\ S of every member u, each row in ascending column order.
        // Fast path (whole set in one tile): all 64 lanes scan the members' adjacency lists TOGETHER -- one flattened
        // index space, independent loads -- and list the matches (position in S, weight) in LDS in (row, column)
        // order; each member's lane then folds ITS matches in that same order, so the arithmetic is the serial walk's.
        bool listed = false;
        int nh_listed = 0;
        if (small && io.active && io.off >= 0 && io.n >= 0) {
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
                ab_q[q] = i < io.n ? io.pool_ab[io.off + i] : 0u;
                w_q[q] = (i < io.n && g.w) ? io.pool_w[io.off + i] : 1.0;
            }
#ifdef FITGNN_GREEDY_STAMPS
            if (ab_q[0] == 0xffffffffu) lds.cnt[0] = 1;   // (never) makes the stamp below wait for the first chunk
#endif
            CSTAMP(3);
#pragma unroll
            for (int q = 0; q < kChunks; ++q) {
                const int base = q * 64;
                if (base >= io.n) break;   // wave-uniform
                const int i = base + lane;
                const bool valid = i < io.n;
                const uint32_t ab = ab_q[q];
                const double wv = w_q[q];
                int a = (int)(ab >> 8), b = (int)(ab & 255u);
                if (io.use_remap) { a = lds.remap[a]; b = lds.remap[b]; }
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
                if (q == 0) { CSTAMP(8); }
            }
#ifdef FITGNN_GREEDY_STAMPS
            if (io.active && io.use_remap && !io.fresh_out && lane == 0) atomicAdd(&g_cost_dbg[9], (unsigned long long)io.n);
#endif
            listed = true;
            nh_listed = nh;
            FITGNN_WAVE_SYNC();
        }
        if (small && !listed) {
            // Membership of a column in S: a 512-slot open-addressing table of (node, position) in the Y tile (not written before
            // the fold below) -- one LDS read per adjacency entry instead of a six-step binary search.  The rows are walked one at
            // a time, the lanes over the row's adjacency list, eight rows' column loads in flight; a row's lane-order is its
            // ascending column order, so the matches are listed in the (row, column) order the serial walk visits them in.
            unsigned long long *tab = reinterpret_cast<unsigned long long *>(lds.Y);
            constexpr int kSlots = 512;
            constexpr unsigned long long kEmpty = ~0ull;
#pragma unroll
            for (int q = 0; q < kSlots / 64; ++q) tab[q * 64 + lane] = kEmpty;
            lds.cnt[lane] = 0;
            FITGNN_WAVE_SYNC();
            auto slot_of = [](int32_t v) { return (int)(((uint32_t)v * 2654435761u) >> 23); };   // top 9 bits
            if (lane < rows) {
                const int32_t v = lds.S[lane];
                const unsigned long long mine = ((unsigned long long)(uint32_t)lane << 32) | (uint32_t)v;
                int h = slot_of(v);
                while (atomicCAS(&tab[h], kEmpty, mine) != kEmpty) h = (h + 1) & (kSlots - 1);
            }
            FITGNN_WAVE_SYNC();
            int nh = 0;
            listed = true;
            // one chunk of (at most 64) adjacency entries of row a, already loaded: look the columns up, list the matches
            auto take = [&](int a, bool valid, int32_t c, double wv) {
                int b = -1;
                if (valid) {
                    int h = slot_of(c);
                    for (;;) {
                        const unsigned long long sl = tab[h];
                        if ((int32_t)(uint32_t)sl == c && sl != kEmpty) { b = (int)(sl >> 32); break; }
                        if (sl == kEmpty) break;
                        h = (h + 1) & (kSlots - 1);
                    }
                }
                const bool hit = b >= 0;
                const unsigned long long bal = __ballot(hit);
                const int nb = __popcll(bal);
                if (nb == 0) return;
                if (nh + nb > kHitCap) { listed = false; return; }   // wave-uniform
                if (hit) {
                    const int at = nh + __popcll(bal & ((1ull << lane) - 1ull));
                    lds.hit_a[at] = (uint8_t)a;
                    lds.hit_b[at] = (uint8_t)b;
                    lds.hit_w[at] = wv;
                }
                if (lane == 0) lds.cnt[a] += nb;
                nh += nb;
            };
            constexpr int kRowsAhead = 8;
            for (int a0 = 0; a0 < rows && listed; a0 += kRowsAhead) {
                int32_t c8[kRowsAhead];
                double w8[kRowsAhead];
                int e8[kRowsAhead], d8[kRowsAhead];
#pragma unroll
                for (int q = 0; q < kRowsAhead; ++q) {
                    const int a = min(a0 + q, rows - 1);   // wave-uniform; a clamped row's loads are dropped below
                    e8[q] = __builtin_amdgcn_readlane(pre_e0, a);
                    d8[q] = a0 + q < rows ? __builtin_amdgcn_readlane(pre_deg, a) : 0;
                    c8[q] = 0; w8[q] = 1.0;
                    if (lane < d8[q]) { c8[q] = g.col[e8[q] + lane]; if (g.w) w8[q] = g.w[e8[q] + lane]; }
                }
#pragma unroll
                for (int q = 0; q < kRowsAhead; ++q) {
                    if (d8[q] == 0 || !listed) continue;   // wave-uniform
                    take(a0 + q, lane < d8[q], c8[q], w8[q]);
                    for (int base = 64; base < d8[q] && listed; base += 64) {   // a row of more than 64 neighbours
                        const bool valid = base + lane < d8[q];
                        const int32_t c = valid ? g.col[e8[q] + base + lane] : 0;
                        const double wv = (valid && g.w) ? g.w[e8[q] + base + lane] : 1.0;
                        take(a0 + q, valid, c, wv);
                    }
                }
            }
            nh_listed = nh;
            FITGNN_WAVE_SYNC();
        }
        CSTAMP(4);
#ifdef FITGNN_GREEDY_STAMPS
        nh_cost_ += nh_listed;
#endif
        if (small) {
            if (io.active) {   // keep the list for the set's next costing (in place when it was read from the pool: it only shrinks)
                io.n = -1;
                int64_t dst = io.fresh_out ? -1 : io.off;
                if (listed) {
                    if (dst < 0 && io.bump + nh_listed <= io.pool_cap) { dst = io.bump; io.bump += nh_listed; }
                    if (dst >= 0) {
                        for (int i = lane; i < nh_listed; i += 64) {
                            io.pool_ab[dst + i] = (uint16_t)(((uint32_t)lds.hit_a[i] << 8) | (uint32_t)lds.hit_b[i]);
                            if (g.w) io.pool_w[dst + i] = lds.hit_w[i];
                        }
                        io.n = nh_listed;
                    }
                }
                if (io.fresh_out || dst >= 0) io.off = dst;
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
#pragma unroll 4   // four matches' LDS reads in flight; the additions stay in list order
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
        CSTAMP(5);
        // M[k][l] += B[a][k] * Y[a][l], a ascending
        {
            // lanes beyond K * K read entry 0 and drop the product: no divergent branch between the LDS reads of one row
            const bool va = ea < KK, vb = eb < KK, vc = ec < KK, vd = ed < KK;
            const int ka = va ? ea / K : 0, la = va ? ea - ka * K : 0;
            const int kb = vb ? eb / K : 0, lb = vb ? eb - kb * K : 0;
            const int kc = vc ? ec / K : 0, lc = vc ? ec - kc * K : 0;
            const int kd = vd ? ed / K : 0, ld = vd ? ed - kd * K : 0;
            if (KK <= 128) {   // wave-uniform (K <= 11): entries 128.. do not exist
#pragma unroll 4
                for (int a = 0; a < rows; ++a) {
                    const double *Br = lds.B + a * K, *Yr = lds.Y + a * K;
                    const double pa = Br[ka] * Yr[la], pb = Br[kb] * Yr[lb];
                    const double sa = m0 + pa, sb = m1 + pb;
                    m0 = va ? sa : m0;
                    m1 = vb ? sb : m1;
                }
            } else {
#pragma unroll 2
                for (int a = 0; a < rows; ++a) {
                    const double *Br = lds.B + a * K, *Yr = lds.Y + a * K;
                    const double pa = Br[ka] * Yr[la], pb = Br[kb] * Yr[lb], pc = Br[kc] * Yr[lc], pd = Br[kd] * Yr[ld];
                    const double sa = m0 + pa, sb = m1 + pb, sc = m2 + pc, sd = m3 + pd;
                    m0 = va ? sa : m0;
                    m1 = vb ? sb : m1;
                    m2 = vc ? sc : m2;
                    m3 = vd ? sd : m3;
                }
            }
        }
        FITGNN_WAVE_SYNC();
    }
    CSTAMP(6);
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
    CSTAMP(7);
#ifdef FITGNN_GREEDY_STAMPS
    if (io.active && io.use_remap && !io.fresh_out && lane == 0) { atomicAdd(&g_cost_dbg[10], (unsigned long long)nc); atomicAdd(&g_cost_dbg[11], (unsigned long long)nh_cost_); }
#endif
    return res;
}

template <bool STAGED = false>
__device__ inline double set_cost_wave(const CostGraph &g, const int32_t *__restrict__ S, int nc, CostLds &lds, HitIO &io) {
    if (nc < 2) return INFINITY;
    const int K = g.node_K ? __builtin_amdgcn_readfirstlane(g.node_K[STAGED ? lds.S[0] : S[0]]) : g.K;
    if (K == 10) return set_cost_wave_k<STAGED, 10>(g, S, nc, lds, io, K);   // coarsening_utils.py:20 (K = 10 unless a component is smaller)
    return set_cost_wave_k<STAGED, 0>(g, S, nc, lds, io, K);
}

template <bool STAGED = false>
__device__ inline double set_cost_wave(const CostGraph &g, const int32_t *__restrict__ S, int nc, CostLds &lds) {
    HitIO none{};
    none.active = false;
    return set_cost_wave<STAGED>(g, S, nc, lds, none);
}

}  // namespace fitgnn
