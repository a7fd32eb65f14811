// spmm.hip -- CSR SpMM with LDS-staged row windows for block-diagonal subgraph batches (gfx950).
//
// Replaces the propagate step of torch_geometric's GCNConv/SAGEConv/GINConv/APPNP as called from
// FIT-GNN network.py:31,60,90,126,161,197 (gather x[row] -> multiply -> scatter-add into col), and its
// autograd backward (the same product with the transposed CSR).
//
// Mapping (MI355X): one 256-thread workgroup (4 waves) per (row tile, 64*VEC-column slab).
//   1. the tile's column window -- rows [win_begin, win_begin+win_rows) of X, slab columns only -- is
//      streamed HBM -> registers -> LDS with 16-byte coalesced loads (1 KiB per wave instruction), up to
//      8 loads in flight per lane; the tile's slice of the CSR (row pointers, column indices, values) is
//      staged in LDS next to it, so the row loop never waits on a dependent global load;
//   2. each wave owns rows of the tile round-robin; the row's (col,val) pairs are taken 64 at a time,
//      one pair per lane, and broadcast with v_readlane; the lane accumulates its VEC columns from
//      LDS (window hit, wave-uniform test) or straight from global/L2 (miss);
//   3. bias / ELU / dropout are applied in registers and the row is stored with one coalesced write.
// HBM traffic per tile ~= window rows read once + tile rows written once: the algorithmic minimum
// 4*H*(N+N) + 8*nnz for a block-diagonal batch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include <cstdlib>

#include "common.h"
#include "fitgnn_hip.h"

namespace {

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;
constexpr int kDefaultWindowRows = 16;  // LDS rows of the dense operand per workgroup (VEC=4: 1 KiB each)
constexpr int kMaxWindowRows = 96;
constexpr int kSmallWindowRows = 16;    // windows up to this size run the high-occupancy instantiation
constexpr int kLongRow = 16;            // 64-entry chunks of a row with at least this many entries take the gather path

template <int VEC> struct Pack;
template <> struct Pack<4> {
    using T = float4;
    static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ void fma(T &a, float v, const T &x) {
        a.x = fmaf(v, x.x, a.x); a.y = fmaf(v, x.y, a.y); a.z = fmaf(v, x.z, a.z); a.w = fmaf(v, x.w, a.w);
    }
    static __device__ __forceinline__ float get(const T &a, int i) { return i == 0 ? a.x : i == 1 ? a.y : i == 2 ? a.z : a.w; }
    static __device__ __forceinline__ void set(T &a, int i, float f) { if (i == 0) a.x = f; else if (i == 1) a.y = f; else if (i == 2) a.z = f; else a.w = f; }
};
template <> struct Pack<1> {
    using T = float;
    static __device__ __forceinline__ T zero() { return 0.f; }
    static __device__ __forceinline__ void fma(T &a, float v, const T &x) { a = fmaf(v, x, a); }
    static __device__ __forceinline__ float get(const T &a, int) { return a; }
    static __device__ __forceinline__ void set(T &a, int, float f) { a = f; }
};

// The epilogue applied as a row leaves the kernel.  Forward (default): + bias, ELU, dropout.  FITGNN_EPI_BACKWARD: the row is a
// GRADIENT w.r.t. a fused layer output  o = dropout(ELU(z))  and what is stored is the gradient w.r.t. z,
//   dZ = keep ? d / (1 - p) * (e > 0 ? 1 : e + 1) : 0,   e = o * (1 - p)
// (the arithmetic of epilogue_bwd_kernel, gcn_ops.hip; `prev` = o, same dropout hash), with the lane's column sums kept in cs
// for the bias gradient: a backward SpMM whose result feeds the previous layer never writes the un-transformed gradient.
struct RowEpilogue {
    uint32_t epi;
    float keep_scale, unscale;
    uint32_t thresh;
    uint64_t seed;
    const uint8_t *mask;
    const float *prev;
};
// BWD_OK == false compiles the backward mode out (the whole-subgraph kernel's forward instantiations sit at their register limit)
// epilogue_value: the row's slice as it leaves the kernel (see RowEpilogue); finish_row: + the store.
// o: the row's slice of E.prev, requested by the caller BEFORE it aggregated the row (prev_row): the load rides under the row's
// gathers instead of standing between the last FMA and the store
template <int VEC, bool BWD_OK = true, bool NOEPI = false>
__device__ __forceinline__ typename Pack<VEC>::T epilogue_value(typename Pack<VEC>::T acc, int row, int col0, int H, const float (&bv)[VEC],
                                                                const RowEpilogue &E, float (&cs)[VEC], const typename Pack<VEC>::T &o) {
    if (NOEPI) return acc;  // a plain product (the launcher saw epilogue == 0)
    using P = Pack<VEC>;
    const uint32_t epi = E.epi;
    uint64_t bits = 0;
    const uint64_t idx0 = (uint64_t)row * (uint64_t)H + (uint64_t)col0;
    if ((epi & FITGNN_EPI_DROPOUT) && !E.mask) bits = fitgnn::dropout_bits(E.seed, idx0 >> 2);
    if (BWD_OK && (epi & FITGNN_EPI_BACKWARD)) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float d = P::get(acc, i);
            if (epi & FITGNN_EPI_DROPOUT) {
                const bool keep = E.mask ? (E.mask[idx0 + i] != 0) : fitgnn::dropout_keep(bits, (int)((idx0 + i) & 3), E.thresh);
                d = keep ? d * E.keep_scale : 0.f;
            }
            if (epi & FITGNN_EPI_ELU) {
                const float e = P::get(o, i) * E.unscale;
                d = e > 0.f ? d : d * (e + 1.0f);
            }
            P::set(acc, i, d);
            cs[i] += d;
        }
        return acc;
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        float z = P::get(acc, i) + bv[i];
        if (epi & FITGNN_EPI_ELU) z = z > 0.f ? z : __expf(z) - 1.0f;
        if (epi & FITGNN_EPI_DROPOUT) {
            const bool keep = E.mask ? (E.mask[idx0 + i] != 0) : fitgnn::dropout_keep(bits, (int)((idx0 + i) & 3), E.thresh);
            z = keep ? z * E.keep_scale : 0.f;
        }
        P::set(acc, i, z);
    }
    return acc;
}

template <int VEC, bool BWD_OK = true, bool NOEPI = false>
__device__ __forceinline__ void finish_row(typename Pack<VEC>::T acc, int row, int col0, int H, float *__restrict__ Y,
                                           int64_t ldy, const float (&bv)[VEC], const RowEpilogue &E, float (&cs)[VEC],
                                           const typename Pack<VEC>::T &o) {
    using T = typename Pack<VEC>::T;
    const T out = epilogue_value<VEC, BWD_OK, NOEPI>(acc, row, col0, H, bv, E, cs, o);
#ifdef FITGNN_SPMM_NOSTORE
    if (Pack<VEC>::get(out, 0) == 12345.678f)
#endif
    *reinterpret_cast<T *>(Y + (int64_t)row * ldy + col0) = out;
}

template <int VEC, bool BWD_OK = true>
__device__ __forceinline__ typename Pack<VEC>::T prev_row(const RowEpilogue &E, int row, int col0, int H, bool live) {
    using T = typename Pack<VEC>::T;
    if (BWD_OK && (E.epi & FITGNN_EPI_BACKWARD) && live) return *reinterpret_cast<const T *>(E.prev + (uint64_t)row * (uint64_t)H + (uint64_t)col0);
    return Pack<VEC>::zero();
}

// Column sums of the rows a workgroup stored (backward epilogue): lanes own columns, the 4 waves are added in a fixed order
// through LDS (`red`: 4 x 64 x VEC floats, free at this point) into col_part[part * H + column].
template <int VEC>
__device__ __forceinline__ void write_col_part(float *red, const float (&cs)[VEC], float *__restrict__ col_part, int64_t part, int H,
                                               int col0, bool live) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VEC; ++i) red[(wave * 64 + lane) * VEC + i] = cs[i];
    __syncthreads();
    if (wave == 0 && live) {
#pragma unroll
        for (int i = 0; i < VEC; ++i)
            col_part[part * H + col0 + i] = ((red[(0 * 64 + lane) * VEC + i] + red[(1 * 64 + lane) * VEC + i]) + red[(2 * 64 + lane) * VEC + i]) +
                                            red[(3 * 64 + lane) * VEC + i];
    }
}

// B = window rows each wave keeps in flight per pass; MPR = staged CSR entries per window row (8 B each).
// <B=4, MPR=16> needs <= 64 VGPRs and ~18 KiB of LDS for a 16-row window: 8 workgroups (32 waves) per CU, which is
// what hides the fetch -> compute -> store-acknowledge latency chain of a tile (~8 us under load) at HBM rate.
// PLAIN: contiguous windows and no row indirection (lcol == win_cols == xrow == NULL), the hidden layers' production case:
// the column -> operand-row translation folds to an addition and the row loop carries no per-entry scalar branches.
template <int VEC, int B, int MPR, bool PLAIN, bool NOEPI = false>
__global__ __launch_bounds__(kThreads, (B <= 4 ? (PLAIN ? 8 : 7) : 4)) void spmm_tile_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const float *__restrict__ X, int64_t ldx, float *__restrict__ Y, int64_t ldy, int32_t H,
    const fitgnn_tile_t *__restrict__ tiles, int32_t n_tiles, int32_t tiles_per_xcd, int32_t n_slabs, int32_t lds_rows,
    const int32_t *__restrict__ lcol_arg, const int32_t *__restrict__ win_cols_arg, const int32_t *__restrict__ xrow_arg,
    const float *__restrict__ bias, uint32_t epi, float p_drop, uint64_t seed_arg, const uint8_t *__restrict__ mask,
    const float *__restrict__ prev, float *__restrict__ col_part, int32_t zero_from_arg) {
    // zero_from (with xrow): operand rows >= zero_from are rows of zeros and are not loaded (see spmm_block_kernel)
    const int zero_from = PLAIN ? -1 : zero_from_arg;
    const int32_t *__restrict__ lcol = PLAIN ? nullptr : lcol_arg;
    const int32_t *__restrict__ win_cols = PLAIN ? nullptr : win_cols_arg;
    const int32_t *__restrict__ xrow = PLAIN ? nullptr : xrow_arg;
    const uint64_t seed = fitgnn::resolve_seed(seed_arg, epi);
    using P = Pack<VEC>;
    using T = typename P::T;
    constexpr int SLAB = 64 * VEC;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    // LDS carve: [window: lds_rows x 64 T][row pointers: lds_rows+1][col: meta_cap][val: meta_cap]
    T *lds = reinterpret_cast<T *>(lds_raw);
    const int meta_cap = lds_rows * MPR;
    int32_t *s_rp = reinterpret_cast<int32_t *>(lds_raw + (size_t)lds_rows * 64 * sizeof(T));
    int32_t *s_col = s_rp + (lds_rows + 1 + 3) / 4 * 4;
    float *s_val = reinterpret_cast<float *>(s_col + meta_cap);

    // 1-D grid; block -> (tile, slab).  Blocks are dispatched in id order, round-robin over the 8 XCDs, so the
    // tile at array position p runs on XCD p % 8: the host lays the tile array out so that every XCD gets a
    // contiguous, equally heavy range of the batch (csr.arrange_tiles_for_xcds).  The slab is the FASTEST index
    // inside an XCD's sequence, so both halves of every operand/output row are in flight at the same time (a
    // slab-major order would stream bytes [0,1K) of every 2 KiB row first and [1K,2K) later).
    const int bid = blockIdx.x;
    const int seq = bid >> 3;
    const int slab = seq % n_slabs;
    const int t = (seq / n_slabs) * 8 + (bid & 7);  // tile array position p runs on XCD p % 8 (host balances it)
    if (t >= n_tiles) return;
    const fitgnn_tile_t tile = tiles[t];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col0 = slab * SLAB + lane * VEC;
    const bool live = col0 + VEC <= H;
    const float *Xs = X + (live ? col0 : max(H - VEC, 0));  // dead lanes load a valid column group (never stored)
    const bool planned = !PLAIN && lcol != nullptr;          // columns are LDS slots (>= 0) or -(global col + 1)
    const bool listed = !PLAIN && tile.reserved[0] != 0;     // window rows come from win_cols[]
    const int win_begin = tile.win_begin;
    const int win_rows = min(tile.win_rows, lds_rows);
    const int tile_rows = tile.row_end - tile.row_begin;
    const int rp_rows = min(tile_rows, lds_rows);  // rows whose pointers are staged
    const int32_t *cidx = planned ? lcol : col;

    // ---- stage: HBM -> LDS in ONE round trip: the window rows (8 x 16 B per lane in flight), the tile's row
    // ---- pointers and its (col, val) slice are all addressed from the tile descriptor alone.
    const int E0 = tile.nnz_begin;
    const int n_meta = min(tile.nnz_end - E0, meta_cap);
    {
        constexpr int MB = B <= 4 ? 1 : 4;  // CSR entries per thread in the first pass
        int mc[MB];
        float mv[MB];
        int rpv = 0;
        // this wave stages window rows wave, wave+4, ...; for a listed window fetch their operand-row ids first
        // xrow (optional): operand row r of the pattern lives at X[xrow[r]] -- union rows that are copies of the
        // same original node share one row of a de-duplicated operand table
        int wcv = 0;
        if (wave + lane * kWaves < win_rows) {
            if (listed) wcv = win_cols[win_begin + wave + lane * kWaves];
            else if (xrow) wcv = win_begin + wave + lane * kWaves;
            if (xrow) wcv = xrow[wcv];
        }
        const bool indirect = listed || xrow != nullptr;
        for (int p0 = 0; p0 == 0 || wave + p0 * kWaves < win_rows; p0 += B) {
            T v[B];
#pragma unroll
            for (int j = 0; j < B; ++j) {
                const int r = wave + (p0 + j) * kWaves;
                const int rr = min(r, max(win_rows - 1, 0));
                const int src = indirect ? __builtin_amdgcn_readlane(wcv, min(p0 + j, 63)) : win_begin + rr;
                v[j] = P::zero();
                if (r < win_rows && !(zero_from >= 0 && xrow && src >= zero_from)) v[j] = *reinterpret_cast<const T *>(Xs + (int64_t)src * ldx);
            }
            if (p0 == 0) {  // first pass: put the CSR slice in flight behind the window loads
                if ((int)threadIdx.x <= rp_rows) rpv = rowptr[tile.row_begin + threadIdx.x];
#pragma unroll
                for (int j = 0; j < MB; ++j) {
                    const int i = threadIdx.x + j * kThreads;
                    mc[j] = 0; mv[j] = 0.f;
                    if (i < n_meta) { mc[j] = cidx[E0 + i]; mv[j] = val[E0 + i]; }
                }
            }
#pragma unroll
            for (int j = 0; j < B; ++j) {
                const int r = wave + (p0 + j) * kWaves;
                if (r < win_rows) lds[r * 64 + lane] = v[j];
            }
            if (p0 == 0) {
                if ((int)threadIdx.x <= rp_rows) s_rp[threadIdx.x] = rpv;
#pragma unroll
                for (int j = 0; j < MB; ++j) {
                    const int i = threadIdx.x + j * kThreads;
                    if (i < n_meta) { s_col[i] = mc[j]; s_val[i] = mv[j]; }
                }
            }
        }
        for (int i = threadIdx.x + kThreads; i <= rp_rows; i += kThreads) s_rp[i] = rowptr[tile.row_begin + i];
        for (int i = threadIdx.x + MB * kThreads; i < n_meta; i += kThreads) { s_col[i] = cidx[E0 + i]; s_val[i] = val[E0 + i]; }
    }
    __syncthreads();

    const float keep_scale = (epi & FITGNN_EPI_DROPOUT) ? 1.0f / (1.0f - p_drop) : 1.0f;
    const uint32_t thresh = fitgnn::dropout_threshold(p_drop);
    const RowEpilogue rowepi{epi, keep_scale, (epi & FITGNN_EPI_DROPOUT) ? 1.0f - p_drop : 1.0f, thresh, seed, mask, prev};
    float cs[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) cs[i] = 0.f;
    float bv[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) bv[i] = ((epi & FITGNN_EPI_BIAS) && live) ? bias[col0 + i] : 0.f;
    // entry value c -> LDS slot (c - slot_off) or, on a miss, the global operand row
    const int slot_off = planned ? 0 : win_begin;
    auto gcol = [&](int c) -> int {  // c is already slot-relative
        if (!planned) return c + slot_off;
        if (c < 0) return -(c + 1);                                   // planner: operand row not staged
        return listed ? win_cols[win_begin + c] : win_begin + c;      // staged slot beyond a clamped LDS window
    };
    auto xsrc = [&](int g) -> int64_t { return xrow ? xrow[g] : g; };
    auto gather = [&](int g) -> T {   // operand row of pattern column g, from memory -- unless it is one of the zero rows
        const int64_t sr = xsrc(g);
        if (zero_from >= 0 && xrow && sr >= zero_from) return P::zero();
        return *reinterpret_cast<const T *>(Xs + sr * ldx);
    };
    for (int row = tile.row_begin + wave; row < tile.row_end; row += kWaves) {
        const T o_prev = prev_row<VEC>(rowepi, row, col0, H, live);
        const int lr = row - tile.row_begin;
        int e0, e1;
        if (lr < rp_rows) { e0 = s_rp[lr]; e1 = s_rp[lr + 1]; } else { e0 = rowptr[row]; e1 = rowptr[row + 1]; }
        e0 = __builtin_amdgcn_readfirstlane(e0);
        e1 = __builtin_amdgcn_readfirstlane(e1);
        T acc = P::zero();
        for (int base = e0; base < e1; base += 64) {
            const int cnt = min(64, e1 - base);
            int my_c = 0;
            float my_v = 0.f;
            if (lane < cnt) {
                const int i = base + lane - E0;
                if (i < n_meta) { my_c = s_col[i]; my_v = s_val[i]; } else { my_c = cidx[base + lane]; my_v = val[base + lane]; }
            }
            int k = 0;
            if (cnt >= kLongRow) {
                // A long row (a hub: the centre of a star-shaped subgraph references every leaf, most of them outside the
                // window): all entries as plain gathers, eight in flight, no window test -- the loads are unconditional, so
                // their waits are static counts; the one-at-a-time miss path below would pay ~100 L2 round trips in a row
                // while the other three waves of the workgroup wait at the end of the tile.  Same order of additions.
                for (; k + 8 <= cnt; k += 8) {
                    T x[8];
                    float w[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = __builtin_amdgcn_readlane(my_c, k + u) - slot_off;
                        w[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), k + u));
                        x[u] = gather(gcol(c));
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) P::fma(acc, w[u], x[u]);
                }
            }
            // groups of four entries, all four operand reads issued before the first is consumed; a last group of 1-3
            // entries is padded with copies of its first entry at weight 0 (adds +-0: the sums are unchanged) -- most rows
            // hold 2-4 entries (self loop + the cluster's centre), and taken one at a time every read, LDS or L2, would wait
            // for the one before it
            for (; k < cnt; k += 4) {
                const int k1 = min(k + 1, cnt - 1), k2 = min(k + 2, cnt - 1), k3 = min(k + 3, cnt - 1);
                const int c0 = __builtin_amdgcn_readlane(my_c, k) - slot_off;
                const int c1 = __builtin_amdgcn_readlane(my_c, k1) - slot_off;
                const int c2 = __builtin_amdgcn_readlane(my_c, k2) - slot_off;
                const int c3 = __builtin_amdgcn_readlane(my_c, k3) - slot_off;
                const float w0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), k));
                const float w1 = k + 1 < cnt ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), k1)) : 0.f;
                const float w2 = k + 2 < cnt ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), k2)) : 0.f;
                const float w3 = k + 3 < cnt ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), k3)) : 0.f;
                const bool in0 = (unsigned)c0 < (unsigned)win_rows, in1 = (unsigned)c1 < (unsigned)win_rows;
                const bool in2 = (unsigned)c2 < (unsigned)win_rows, in3 = (unsigned)c3 < (unsigned)win_rows;
                T x0, x1, x2, x3;
                if (in0 && in1 && in2 && in3) {  // wave-uniform: four LDS reads in flight
                    x0 = lds[c0 * 64 + lane]; x1 = lds[c1 * 64 + lane];
                    x2 = lds[c2 * 64 + lane]; x3 = lds[c3 * 64 + lane];
                } else {
                    const int g0 = gcol(c0), g1 = gcol(c1), g2 = gcol(c2), g3 = gcol(c3);
                    if (in0) x0 = lds[c0 * 64 + lane]; else x0 = gather(g0);
                    if (in1) x1 = lds[c1 * 64 + lane]; else x1 = gather(g1);
                    if (in2) x2 = lds[c2 * 64 + lane]; else x2 = gather(g2);
                    if (in3) x3 = lds[c3 * 64 + lane]; else x3 = gather(g3);
                }
                P::fma(acc, w0, x0); P::fma(acc, w1, x1); P::fma(acc, w2, x2); P::fma(acc, w3, x3);
            }
        }
        if (live) finish_row<VEC, true, NOEPI>(acc, row, col0, H, Y, ldy, bv, rowepi, cs, o_prev);
    }
    if (!NOEPI && (epi & FITGNN_EPI_BACKWARD) && col_part)   // wave-uniform, every wave of the workgroup gets here
        write_col_part<VEC>(reinterpret_cast<float *>(lds_raw), cs, col_part, t, H, col0, live);
}


// ---------------------------------------------------------------------------------------------------------------
// Whole-subgraph kernel for diagonal blocks LARGER than the window (fitgnn_spmm_csr_blocks_f32).
//
// Why: cut into window-sized tiles, a large subgraph re-reads its operand rows.  --extra_node subgraphs are stars (a
// cluster's few own nodes, the centres, and their many neighbours, utils.py:235-239): every leaf row references the centres
// and a centre's row references every leaf, and of those references only the ones inside the tile's own 16 rows are window
// hits.  The rest are gathered again -- by the time a centre's row is processed its leaves' rows, streamed by other
// workgroups, have left the 4-MiB L2: rocprofv3 --pmc on S-products (one ogbn-products community, 82.5 k subgraphs of ~100
// rows) counts 49.7 GB at the memory side per launch against 34.0 GB algorithmic, 6.4 TB/s of real traffic for 4.4 TB/s of
// useful work.  Here ONE workgroup walks a whole block in 16-row pieces and reads every operand row exactly once:
//   * the rows of the block's LONG rows (the centres; up to kBlkLong per block) are pinned in LDS for the whole block, so a
//     leaf's reference to a centre is an LDS hit in every piece;
//   * a long row is not computed in its own piece: the wave that owns it carries its accumulator across ALL pieces and, as
//     each piece's window sits in LDS, adds that piece's share of the row -- its CSR entries are sorted by column, so the
//     pieces consume them in order and the additions happen in exactly the order of the one-row-at-a-time kernel (same bits);
//   * the next piece (window rows, row pointers, CSR slice) is prefetched into registers while the current one is computed:
//     the workgroup is persistent over its block, so operand bytes are in flight all the time.
// Entries of short rows that point outside their piece at a row that is not pinned (leaf -- leaf edges across pieces) are
// gathered from L2 / HBM as in the tile kernel.  A "block" is any run of consecutive rows (a SEGMENT): columns outside it are
// legal and are gathered too, so a connected subgraph with many centres is cut into one segment per centre (its star) once
// the rows are laid out star by star (data.SubgraphBatch, layout="star").  H % 4 == 0 only (callers tile otherwise).
#ifdef FITGNN_SPMM_STAMPS  // make EXTRA=-DFITGNN_SPMM_STAMPS + tools/spmm_stamps.py: where a workgroup of the whole-subgraph kernel spends its cycles
static __device__ unsigned long long *g_spmm_dbg;   // [workgroups x 8], set by fitgnn_debug_spmm_buffer: plain stores, no atomics
#define SSTAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#define SACC(i, a, b) sdbg[i] += (b) - (a)
#else
#define SSTAMP(var)
#define SACC(i, a, b)
#endif
constexpr int kBlkRows = 16;   // rows per piece (== the tile kernel's default window)
constexpr int kBlkLW = 1;      // long rows carried per wave
constexpr int kBlkLong = 4 * kBlkLW;  // ... per block
constexpr int kBlkMeta = 128;  // CSR entries of a piece staged in LDS (threads 0..127 fetch one each): a piece of 16 short rows
                               // holds ~50; with 256, on stars of ~50 rows, the staging itself was 8 % of the kernel's reads

// XROW: operand row r of the pattern lives at X[xrow[r]] (a de-duplicated operand table, as in the tile kernel): the window
// rows' table indices are fetched one piece ahead of the rows themselves, so the prefetch never waits on an index.
// TWO (with XROW, NOEPI): the two-hop backward, fitgnn_spmm_two_hop_blocks_f32.  The operand of the product is dZ of the layer below,
// which is not stored: X is the side table ZT (dZ of the rows that have to be readable from anywhere), xrow[r] >= 0 names row r's
// place in it and xcol[e] that of entry e's column; a row with xrow[r] < 0 is a "simple" one -- at most one of its columns is a loss
// row -- and its dZ is made as the row is staged into the window: the prefetch brings its `prev` slice and the compact operand row
// Xc[row_p[r]] of that column, and dZ = (row_w[r] * Xc row) . ELU' / dropout' (prev) goes to LDS in place of an operand row.  From
// there on the kernel is the plain product.  The column sums of dZ (every row of a block passes through a window once) leave as
// in the backward form.
template <bool XROW, bool BWD, bool NOEPI = false, bool TWO = false>
__global__ __launch_bounds__(kThreads, BWD ? 4 : TWO ? 6 : XROW ? 7 : 7) void spmm_block_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const float *__restrict__ X, int64_t ldx, float *__restrict__ Y, int64_t ldy, int32_t H,
    const fitgnn_block_t *__restrict__ blocks, int32_t n_blocks, const int32_t *__restrict__ long_rows, int32_t n_slabs,
    const float *__restrict__ bias, uint32_t epi, float p_drop, uint64_t seed_arg, const uint8_t *__restrict__ mask,
    const int32_t *__restrict__ xrow, const int32_t *__restrict__ xcol, const float *__restrict__ prev, float *__restrict__ col_part,
    int32_t zero_from, const float *__restrict__ Xc2 = nullptr, int64_t ldxc = 0, const int32_t *__restrict__ row_p = nullptr,
    const float *__restrict__ row_w = nullptr, int32_t xc_zero_from = 0) {
    static_assert(!TWO || (XROW && !BWD && NOEPI), "the two-hop form is a plain product over a table");
    // zero_from (XROW): operand rows >= zero_from are rows of zeros (the tail of a compact operand, ops.ZERO_ROWS): they are not
    // loaded -- as window rows they are staged as zeros, as gathered entries they read the LDS slot kZeroSlot -- so an operand
    // that is zero on most rows costs LDS reads and FMAs, not a memory round trip per gathered entry (measured on the compact
    // backward of S-products: the kernel was latency-bound on those gathers at 2.6 TB/s of stores; a plain fill reaches 6.8)
    using P = Pack<4>;
    using T = float4;
    constexpr int kZeroSlot = kBlkRows + kBlkLong;
    const uint64_t seed = fitgnn::resolve_seed(seed_arg, epi);
    __shared__ T s_win[(kBlkRows + kBlkLong + 1) * 64];  // piece window, then the pinned rows, then one row of zeros
    __shared__ int32_t s_rp[kBlkRows + 4];
    __shared__ int32_t s_col[kBlkMeta];
    __shared__ float s_val[kBlkMeta];
    __shared__ int32_t s_long[kBlkLong];
    __shared__ int32_t s_hub_pos;   // TWO: compact position of the block's first long row if it is a loss row (its operand row sits in the zero slot)
    // block -> (record position, slab) as in the tile kernel: position p runs on XCD p % 8 with both of its slabs; the host
    // gives every XCD a contiguous range of the batch (records with row_begin == row_end pad the short ranges)
    const int bid = blockIdx.x;
    const int seq = bid >> 3;
    const int slab = seq % n_slabs;
    const int b = (seq / n_slabs) * 8 + (bid & 7);
    if (b >= n_blocks) return;
#ifdef FITGNN_SPMM_STAMPS
    unsigned long long sdbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    SSTAMP(s_begin);
    const fitgnn_block_t blk = blocks[b];
    if (blk.row_end <= blk.row_begin) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col0 = slab * 256 + lane * 4;
    const bool live = col0 + 4 <= H;
    const float *Xs = X + (live ? col0 : max(H - 4, 0));
    auto src = [&](int r) -> int64_t { return XROW ? (int64_t)xrow[r] : (int64_t)r; };  // operand row of pattern row / column r
    // operand row of CSR entry e (pattern column c): with the table, xcol[e] = xrow[col[e]] is listed per entry (the caller
    // builds it once per batch), so a gathered entry costs one dependent load, not two
    auto ecol = [&](int e, int c) -> int { return (TWO || (XROW && xcol)) ? xcol[e] : (XROW ? xrow[c] : c); };
    const int n_long = min(blk.n_long, kBlkLong);
    const float keep_scale = (epi & FITGNN_EPI_DROPOUT) ? 1.0f / (1.0f - p_drop) : 1.0f;
    const uint32_t thresh = fitgnn::dropout_threshold(p_drop);
    const RowEpilogue rowepi{(BWD || TWO) ? (epi | FITGNN_EPI_BACKWARD) : (epi & ~FITGNN_EPI_BACKWARD), keep_scale,
                             (epi & FITGNN_EPI_DROPOUT) ? 1.0f - p_drop : 1.0f, thresh, seed, mask, (BWD || TWO) ? prev : nullptr};
    const int colc2 = live ? col0 : max(H - 4, 0);
    float cs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cs[i] = 0.f;
    float bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = ((epi & FITGNN_EPI_BIAS) && live) ? bias[col0 + i] : 0.f;

    const bool has_zero = XROW && !TWO && zero_from >= 0;   // (the two-hop form's table has no zero rows)
    auto is_zero_row = [&](int64_t opr) { return has_zero && opr >= (int64_t)zero_from; };
    // ---- piece prefetch (registers): window rows wave, wave + 4, ...; row pointers; the piece's CSR slice ----
    T pv[4];
    T o_nx[kBlkRows / kWaves];   // BWD: the `prev` slices of this wave's rows of the piece being prefetched
    int p_rp = 0, p_c = 0, p_cx = 0, p_E0 = 0;
    float p_v = 0.f;
    int xr_next = 0;  // XROW: lane j < 4 holds the table row of window row wave + 4 j of the NEXT piece to prefetch
    int xp_next = 0, xw_next = 0;   // TWO: ... and the compact operand row / weight of a simple row's one loss column
    // TWO: the same for the piece whose rows are in pv (published at the top of the piece loop), held as scalars; the compact
    // operand row a simple row needs is its block's first long row's (the leaves of a star and their centre): kept in the LDS slot the
    // compact forms use for their row of zeros; a row with another loss column loads that row late
    int sr_pub[4] = {0, 0, 0, 0}, pp_pub[4] = {0, 0, 0, 0};
    float w_pub[4] = {0.f, 0.f, 0.f, 0.f};
    int hub_pos = 0x7fffffff;
    auto fetch_indices = [&](int r0, int r1) {
        if (XROW) {
            const int r = r0 + wave + lane * kWaves;
            xr_next = (lane < 4 && r < r1) ? xrow[r] : 0;
            if (TWO) {
                xp_next = (lane < 4 && r < r1) ? row_p[r] : 0x7fffffff;
                xw_next = (lane < 4 && r < r1) ? __float_as_int(row_w[r]) : 0;
            }
        }
    };
    auto prefetch = [&](int r0, int r1, int E0) {
        const int xr = xr_next;

#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + wave + j * kWaves;
            pv[j] = P::zero();
            const int64_t sr = XROW ? (int64_t)__builtin_amdgcn_readlane(xr, j) : (int64_t)r;
            if (TWO) {
                // one unconditional row load (the table row, or the row's `prev` slice) and the operand row of a simple row's loss column
                const int rr = min(r, r1 - 1);
                const float *rowp = sr >= 0 ? Xs + sr * ldx : prev + (uint64_t)rr * (uint64_t)H + (uint64_t)colc2;
                const int pp = __builtin_amdgcn_readlane(xp_next, j);
                const T a = *reinterpret_cast<const T *>(rowp);
                pv[j] = r < r1 ? a : P::zero();
                sr_pub[j] = (int)sr;
                pp_pub[j] = pp;
                w_pub[j] = __int_as_float(__builtin_amdgcn_readlane(xw_next, j));
            } else if (r < r1 && !is_zero_row(sr)) {
                pv[j] = *reinterpret_cast<const T *>(Xs + sr * ldx);   // wave-uniform
            }
        }
        if (XROW) fetch_indices(r1, min(r1 + kBlkRows, blk.row_end));  // the piece after: its rows are requested next time round
        if (BWD) {
#pragma unroll
            for (int j = 0; j < kBlkRows / kWaves; ++j) o_nx[j] = prev_row<4, BWD>(rowepi, min(r0 + wave + j * kWaves, r1 - 1), col0, H, live);
        }
        p_rp = 0;
        if ((int)threadIdx.x <= r1 - r0) p_rp = rowptr[r0 + threadIdx.x];
        // the slice's end is not known yet (it is rowptr[r1], in flight above): stage the next kBlkMeta entries of the block,
        // clamped to the block's last entry; entries past the piece's end are ignored by the row loop
        if ((int)threadIdx.x < kBlkMeta) {
            const int e = min(E0 + (int)threadIdx.x, blk.nnz_end - 1);
            p_c = col[e];
            p_cx = (TWO || (XROW && xcol)) ? xcol[e] : p_c;
            p_v = val[e];
        }
        p_E0 = E0;
    };
    // ---- the block's long rows: ids to LDS, their operand rows pinned, this wave's two accumulators and entry cursors ----
    fetch_indices(blk.row_begin, min(blk.row_begin + kBlkRows, blk.row_end));   // (table row ids of the first piece)
    if ((int)threadIdx.x < kBlkLong) s_long[threadIdx.x] = (int)threadIdx.x < n_long ? long_rows[blk.long_off + threadIdx.x] : -1;
    if (XROW && !TWO && (int)threadIdx.x < 64) s_win[kZeroSlot * 64 + threadIdx.x] = P::zero();
    if (TWO && threadIdx.x == 0) s_hub_pos = 0x7fffffff;
    int my_long[kBlkLW], cur[kBlkLW], end[kBlkLW], pos[kBlkLW], lc[kBlkLW], lcx[kBlkLW];
    float lv[kBlkLW];
    T acc_long[kBlkLW];
#pragma unroll
    for (int q = 0; q < kBlkLW; ++q) {
        const int slot = wave + q * kWaves;
        my_long[q] = slot < n_long ? long_rows[blk.long_off + slot] : -1;
        my_long[q] = __builtin_amdgcn_readfirstlane(my_long[q]);
        acc_long[q] = P::zero();
        cur[q] = end[q] = 0;
        pos[q] = 64;  // "chunk exhausted": the first use loads entries [cur, cur + 64)
        lc[q] = 0x7fffffff;
        lcx[q] = 0;
        lv[q] = 0.f;
    }
    // the first piece is requested HERE, between the two dependent levels of the long-row set-up (ids above, their row pointers /
    // operand rows below): issued after it, a workgroup's start was six dependent memory round trips long, now four
    prefetch(blk.row_begin, min(blk.row_begin + kBlkRows, blk.row_end), blk.nnz_begin);
#pragma unroll
    for (int q = 0; q < kBlkLW; ++q) {
        const int slot = wave + q * kWaves;
        if (my_long[q] >= 0) {
            cur[q] = __builtin_amdgcn_readfirstlane(rowptr[my_long[q]]);
            end[q] = __builtin_amdgcn_readfirstlane(rowptr[my_long[q] + 1]);
            const int64_t lr_src = src(my_long[q]);
            s_win[(kBlkRows + slot) * 64 + lane] = is_zero_row(lr_src) ? P::zero() : *reinterpret_cast<const T *>(Xs + lr_src * ldx);
            if (TWO && slot == 0 && lr_src < (int64_t)xc_zero_from) {   // wave 0: the centre is a loss row, its table row number is its compact position
                s_win[kZeroSlot * 64 + lane] = *reinterpret_cast<const T *>(Xc2 + lr_src * ldxc + colc2);
                if (lane == 0) s_hub_pos = (int)lr_src;
            }
        }
    }
    __syncthreads();
    if (TWO) hub_pos = __builtin_amdgcn_readfirstlane(s_hub_pos);
    int lid[kBlkLong];  // the long-row ids, wave-uniform
#pragma unroll
    for (int i = 0; i < kBlkLong; ++i) lid[i] = __builtin_amdgcn_readfirstlane(s_long[i]);

    const int n_pieces = (blk.row_end - blk.row_begin + kBlkRows - 1) / kBlkRows;

    // A long row's entries OUTSIDE the segment (a segment need not be a whole connected subgraph: the stars of a large
    // cluster are segments of their own, and their centres reference each other) are gathered, eight in flight, before
    // (columns < row_begin) and after (columns >= row_end) the pieces -- in CSR order, like everything else.
    auto gather_long = [&](int q, int bound) {  // consume the entries of long row q whose column is < bound
        while (cur[q] < end[q]) {
            if (pos[q] == 64) {
                const int e = cur[q] + lane;
                lc[q] = e < end[q] ? col[e] : 0x7fffffff;
                lcx[q] = (XROW && xcol && e < end[q]) ? xcol[e] : 0;
                lv[q] = e < end[q] ? val[e] : 0.f;
                pos[q] = 0;
            }
            const unsigned long long in = __ballot(lc[q] < bound) >> pos[q];
            const int n_in = in == 0 ? 0 : (int)__builtin_popcountll(in);
            constexpr int kLongBatch = (TWO || (XROW && !BWD)) ? 4 : 8;
            for (int k = pos[q]; k < pos[q] + n_in; k += kLongBatch) {
                const int last = pos[q] + n_in - 1;
                T x[kLongBatch];
                float w[kLongBatch];
#pragma unroll
                for (int u = 0; u < kLongBatch; ++u) {
                    const int kk = min(k + u, last);
                    const int64_t c = (XROW && xcol) ? (int64_t)__builtin_amdgcn_readlane(lcx[q], kk) : src(__builtin_amdgcn_readlane(lc[q], kk));
                    w[u] = k + u <= last ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lv[q]), kk)) : 0.f;
                    x[u] = is_zero_row(c) ? P::zero() : *reinterpret_cast<const T *>(Xs + c * ldx);   // wave-uniform
                }
#pragma unroll
                for (int u = 0; u < kLongBatch; ++u) P::fma(acc_long[q], w[u], x[u]);
            }
            pos[q] += n_in;
            cur[q] += n_in;
            if (pos[q] < 64) break;
        }
    };
#pragma unroll
    for (int q = 0; q < kBlkLW; ++q)
        if (my_long[q] >= 0) gather_long(q, blk.row_begin);
    SSTAMP(s_loop);
    SACC(0, s_begin, s_loop);
    for (int p = 0; p < n_pieces; ++p) {
        SSTAMP(s_p0);
        const int r0 = blk.row_begin + p * kBlkRows, r1 = min(r0 + kBlkRows, blk.row_end);
        const int rows = r1 - r0;
        // ---- publish the prefetched piece ----
        T o_pre[kBlkRows / kWaves];
#pragma unroll
        for (int j = 0; j < kBlkRows / kWaves; ++j) o_pre[j] = o_nx[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = wave + j * kWaves;
            if (r < rows) {
                T v = pv[j];
                if (TWO) {
                    if (sr_pub[j] < 0) {   // a simple row: pv is its `prev` slice
                        T u = P::zero();
                        const int pp = pp_pub[j];
                        T h = P::zero();
                        if (pp < xc_zero_from) {   // (a row without a loss column: h stays 0 -- the slot below may never have been written)
                            if (pp == hub_pos) h = s_win[kZeroSlot * 64 + lane];
                            else h = *reinterpret_cast<const T *>(Xc2 + (int64_t)pp * ldxc + colc2);
                        }
                        P::fma(u, w_pub[j], h);
                        // (the clamped column: a lane beyond H must not index an injected mask past its row; its value is never stored)
                        v = epilogue_value<4, true, false>(u, r0 + r, colc2, H, bv, rowepi, cs, pv[j]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) cs[q] += P::get(v, q);
                    }
                }
                s_win[r * 64 + lane] = v;
            }
        }
        if ((int)threadIdx.x <= rows) s_rp[threadIdx.x] = p_rp;
        const int E0 = p_E0;
        if ((int)threadIdx.x < kBlkMeta) {   // entry -> LDS row (window slot, pinned row) or -(operand row + 1): resolved once, by the thread that stages it
            int sl = p_c - r0;
            if ((unsigned)sl >= (unsigned)rows) {
                const int opr = (XROW && !TWO && !xcol) ? (int)xrow[p_c] : p_cx;   // a gathered entry: its operand row
                sl = is_zero_row(opr) ? kZeroSlot : -(opr + 1);
#pragma unroll
                for (int i = 0; i < kBlkLong; ++i)
                    if (p_c == lid[i]) sl = kBlkRows + i;
            }
            s_col[threadIdx.x] = sl;
            s_val[threadIdx.x] = p_v;
        }
        // Bare barriers inside the piece loop: __syncthreads() would drain vmcnt, i.e. wait for the NEXT piece's prefetch and
        // for the acknowledgement of this piece's row stores.  The hand-off only involves LDS: lgkmcnt(0) covers this wave's
        // ds_writes (above) / ds_reads (at the end of the piece).
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        SSTAMP(s_p1);
        SACC(1, s_p0, s_p1);
        const int E1 = __builtin_amdgcn_readfirstlane(s_rp[rows]);
        const int n_meta = min(E1 - E0, kBlkMeta);
        if (p + 1 < n_pieces) prefetch(r1, min(r1 + kBlkRows, blk.row_end), E1);

        // ---- short rows of the piece ----
        // backward epilogue: the `prev` slices of ALL this wave's rows of a piece travel with the piece's prefetch, one piece ahead --
        // one at a time, at the top of each row, every row of the wave waited out a memory round trip of its own (the
        // aggregation itself runs from LDS)
        // (o_pre: taken from the prefetch at the top of the piece, see below)
#pragma unroll
        for (int j = 0; j < kBlkRows / kWaves; ++j) {
            const int row = r0 + wave + j * kWaves;
            if (row >= r1) break;  // wave-uniform
            bool is_long = false;
#pragma unroll
            for (int i = 0; i < kBlkLong; ++i) is_long |= (row == lid[i]);
            if (is_long) continue;  // wave-uniform
            const T o_prev = o_pre[j];
            const int lr = row - r0;
            const int e0 = __builtin_amdgcn_readfirstlane(s_rp[lr]);
            const int e1 = __builtin_amdgcn_readfirstlane(s_rp[lr + 1]);
            T acc = P::zero();
            for (int base = e0; base < e1; base += 64) {
                const int cnt = min(64, e1 - base);
                int my_c = 0;
                float my_v = 0.f;
                if (lane < cnt) {
                    const int i = base + lane - E0;
                    if (i < n_meta) { my_c = s_col[i]; my_v = s_val[i]; }
                    else {  // beyond the staged slice: resolve here
                        const int c = col[base + lane];
                        my_v = val[base + lane];
                        my_c = c - r0;
                        if ((unsigned)my_c >= (unsigned)rows) {
                            const int opr = ecol(base + lane, c);
                            my_c = is_zero_row(opr) ? kZeroSlot : -(opr + 1);
#pragma unroll
                            for (int t = 0; t < kBlkLong; ++t)
                                if (c == lid[t]) my_c = kBlkRows + t;
                        }
                    }
                }
                // kRowBatch window rows in flight per step (a last group is padded with its last entry at weight 0); the two-hop form
                // takes two: a leaf's row is its own entry and its centre's, and the registers buy a workgroup per CU
                constexpr int kRowBatch = (TWO || (XROW && !BWD)) ? 2 : 4;
                for (int k = 0; k < cnt; k += kRowBatch) {
                    int cq[kRowBatch];
                    float wq[kRowBatch];
                    T xq[kRowBatch];
                    int all = 0;
#pragma unroll
                    for (int u = 0; u < kRowBatch; ++u) {
                        const int ku = min(k + u, cnt - 1);
                        cq[u] = __builtin_amdgcn_readlane(my_c, ku);
                        wq[u] = (u == 0 || k + u < cnt) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), ku)) : 0.f;
                        all |= cq[u];
                    }
                    if (all >= 0) {  // wave-uniform: every read from LDS
#pragma unroll
                        for (int u = 0; u < kRowBatch; ++u) xq[u] = s_win[cq[u] * 64 + lane];
                    } else {
#pragma unroll
                        for (int u = 0; u < kRowBatch; ++u) {
                            if (cq[u] >= 0) xq[u] = s_win[cq[u] * 64 + lane];
                            else xq[u] = *reinterpret_cast<const T *>(Xs + (int64_t)(-(cq[u] + 1)) * ldx);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kRowBatch; ++u) P::fma(acc, wq[u], xq[u]);
                }
            }
            if (live) finish_row<4, BWD, NOEPI>(acc, row, col0, H, Y, ldy, bv, rowepi, cs, o_prev);
        }

        SSTAMP(s_p2);
        SACC(2, s_p1, s_p2);
        // ---- this wave's long rows: the entries whose operand rows sit in this piece (columns < r1), in CSR order ----
#pragma unroll
        for (int q = 0; q < kBlkLW; ++q) {
            if (my_long[q] < 0) continue;  // wave-uniform
            while (cur[q] < end[q]) {
                if (pos[q] == 64) {  // next 64 entries of the row into the lanes
                    const int e = cur[q] + lane;
                    lc[q] = e < end[q] ? col[e] : 0x7fffffff;
                    lcx[q] = (XROW && xcol && e < end[q]) ? xcol[e] : 0;
                    lv[q] = e < end[q] ? val[e] : 0.f;
                    pos[q] = 0;
                }
                // entries are sorted by column: those of this piece are the lanes >= pos with col < r1
                const unsigned long long in = __ballot(lc[q] < r1) >> pos[q];
                const int n_in = in == 0 ? 0 : (int)__builtin_popcountll(in);
                constexpr int kPieceBatch = (TWO || (XROW && !BWD)) ? 2 : 4;
                for (int k = pos[q]; k < pos[q] + n_in; k += kPieceBatch) {
                    const int last = pos[q] + n_in - 1;
                    float wq[kPieceBatch];
                    T xq[kPieceBatch];
#pragma unroll
                    for (int u = 0; u < kPieceBatch; ++u) {
                        const int ku = min(k + u, last);
                        const int cu = __builtin_amdgcn_readlane(lc[q], ku) - r0;
                        wq[u] = (u == 0 || k + u <= last) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lv[q]), ku)) : 0.f;
                        xq[u] = s_win[cu * 64 + lane];
                    }
#pragma unroll
                    for (int u = 0; u < kPieceBatch; ++u) P::fma(acc_long[q], wq[u], xq[u]);
                }
                pos[q] += n_in;
                cur[q] += n_in;
                if (pos[q] < 64) break;  // the rest of the chunk belongs to later pieces
            }
        }
        SSTAMP(s_p3);
        SACC(3, s_p2, s_p3);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the window is overwritten by the next piece
        SSTAMP(s_p4);
        SACC(4, s_p3, s_p4);
    }
    SSTAMP(s_tail);
#pragma unroll
    for (int q = 0; q < kBlkLW; ++q) {
        if (my_long[q] < 0) continue;
        const T o_prev = prev_row<4, BWD>(rowepi, my_long[q], col0, H, live);
        gather_long(q, 0x7fffffff);
        if (live) finish_row<4, BWD, NOEPI>(acc_long[q], my_long[q], col0, H, Y, ldy, bv, rowepi, cs, o_prev);
    }
    if ((BWD || TWO) && col_part) write_col_part<4>(reinterpret_cast<float *>(s_win), cs, col_part, b, H, col0, live);
#ifdef FITGNN_SPMM_STAMPS
    {
        SSTAMP(s_end);
        SACC(5, s_tail, s_end);
        sdbg[6] = s_end - s_begin;
        sdbg[7] = (unsigned long long)n_pieces;
        if (threadIdx.x == 0 && g_spmm_dbg) {
            for (int i = 0; i < 8; ++i) g_spmm_dbg[(size_t)blockIdx.x * 8 + i] = sdbg[i];
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Row-streaming kernel for a COMPACT operand (fitgnn_spmm_rows_compact_f32 / _dz_f32).
//
// The last layer's backward SpMM, dX = A_hat^T dAH, reads an operand that is zero outside the rows that reach the loss
// (run.py:193-204 keeps out[mask]: with --extra_node 2 % of a union's rows): dAH is handed over in compact form -- [n_sel + zero
// rows, H] behind a row indirection, xcol[e] = the operand row of CSR entry e, rows >= zero_from are zero -- and what the launch
// really moves is its OUTPUT (4H R bytes written) and, with the previous layer's derivative in the store, `prev` (4H R read): a
// stream.  The whole-subgraph kernel runs that stream through its LDS windows, barriers and 120 registers (4 workgroups per CU,
// 9.1 ms at S-products for 34 GB).  Here every wave streams a contiguous range of rows on its own, no LDS, no barrier:
//   * the CSR entries of consecutive rows are consecutive: they pass through two 64-entry register tiles (current, next), the row
//     pointers through 64-row batches, all broadcast by v_readlane -- no dependent memory access per row;
//   * EVERY entry is multiplied and added, in CSR order (the order of the other kernels: same bits): an entry whose operand row is
//     one of the zero rows adds w * 0 from registers, a non-zero operand row is fetched once and kept while consecutive rows
//     reference it (the leaves of a star all reference their centre);
//   * the `prev` slices of four rows at a time are requested together, ahead of their rows' arithmetic (4 KB per wave in flight,
//     28 waves per CU);
//   * column sums for the bias gradient stay in registers over the wave's whole range: one partial row per range, not per block.
// <= 64 VGPR: 8 waves per SIMD.
template <bool BWD>
__global__ __launch_bounds__(kThreads, BWD ? 7 : 8) void spmm_rows_compact_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ xcol, const float *__restrict__ val, const float *__restrict__ X,
    int64_t ldx, int32_t zero_from, float *__restrict__ Y, int64_t ldy, int32_t H, int32_t n_rows, int32_t nnz, int32_t n_slabs,
    int32_t rows_per_range, int32_t n_ranges, uint32_t epi, float p_drop, uint64_t seed_arg, const uint8_t *__restrict__ mask,
    const float *__restrict__ prev, float *__restrict__ col_part) {
    using P = Pack<4>;
    using T = float4;
    const uint64_t seed = fitgnn::resolve_seed(seed_arg, epi);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * kWaves + (threadIdx.x >> 6)));
    const int slab = wave % n_slabs, range = wave / n_slabs;   // both 1-KiB halves of a row are in flight together
    if (range >= n_ranges) return;
    const int r_begin = range * rows_per_range;
    const int r_end = min(n_rows, r_begin + rows_per_range);
    const int col0 = slab * 256 + lane * 4;
    const bool live = col0 + 4 <= H;
    const int colc = live ? col0 : max(H - 4, 0);   // dead lanes read a valid column group (never stored)
    const float *Xs = X + colc;
    const float keep_scale = (epi & FITGNN_EPI_DROPOUT) ? 1.0f / (1.0f - p_drop) : 1.0f;
    const RowEpilogue rowepi{BWD ? epi : 0u, keep_scale, (epi & FITGNN_EPI_DROPOUT) ? 1.0f - p_drop : 1.0f, fitgnn::dropout_threshold(p_drop),
                             seed, mask, BWD ? prev : nullptr};
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    const float bv[4] = {0.f, 0.f, 0.f, 0.f};
    auto prev_at = [&](int row) -> T {   // rows past the range's end re-read its last row (the load stays unconditional)
        if (!BWD) return P::zero();
        return *reinterpret_cast<const T *>(prev + (uint64_t)min(row, r_end - 1) * (uint64_t)H + (uint64_t)colc);
    };
    // row pointers: lane i of rp holds rowptr[rb + i] of the current 64-row batch, rp_n the next batch's
    int rb = r_begin;
    int rp = rowptr[min(rb + lane, n_rows)], rp_n = rowptr[min(rb + 64 + lane, n_rows)];
    // CSR entries: lane i of (t_x, t_v) holds entry T0 + i, (n_x, n_v) entry T0 + 64 + i
    int T0 = __builtin_amdgcn_readfirstlane(rp);
    auto ent = [&](int e) { return min(e, max(nnz - 1, 0)); };
    int t_x = xcol[ent(T0 + lane)], n_x = xcol[ent(T0 + 64 + lane)];
    float t_v = val[ent(T0 + lane)], n_v = val[ent(T0 + 64 + lane)];
    int cached = -1;          // operand row held in xc (wave-uniform)
    T xc = P::zero();
    for (int r = r_begin; r < r_end; r += 4) {
        // The `prev` slices of the group's four rows are requested TOGETHER, ahead of the first row's arithmetic: hipcc waits with
        // s_waitcnt vmcnt(0) before the first use of a loaded register whenever a loop lies between request and use (it cannot
        // count across the entry loop), so a request made one row ahead is waited for at once -- one memory round trip per row.
        // Requested as a group the four loads share one round trip (4 KB per wave in flight, 28 waves per CU).
        T o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = prev_at(r + j);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = r + j;
            if (row >= r_end) break;   // wave-uniform
            int i = row - rb;
            if (i >= 64) {   // next batch of row pointers
                rb += 64;
                i -= 64;
                rp = rp_n;
                rp_n = rowptr[min(rb + 64 + lane, n_rows)];
            }
            const int e0 = __builtin_amdgcn_readlane(rp, i);
            const int e1 = i < 63 ? __builtin_amdgcn_readlane(rp, i + 1) : __builtin_amdgcn_readfirstlane(rp_n);
            T acc = P::zero();
            for (int e = e0; e < e1; ++e) {
                int k = e - T0;
                if (k >= 64) {   // next tile of entries
                    T0 += 64;
                    k -= 64;
                    t_x = n_x;
                    t_v = n_v;
                    n_x = xcol[ent(T0 + 64 + lane)];
                    n_v = val[ent(T0 + 64 + lane)];
                }
                const int c = __builtin_amdgcn_readlane(t_x, k);
                const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t_v), k));
                if (c < zero_from) {   // a row of the selection (wave-uniform)
                    if (c != cached) {
                        xc = *reinterpret_cast<const T *>(Xs + (int64_t)c * ldx);
                        cached = c;
                    }
                    P::fma(acc, w, xc);
                } else {
                    P::fma(acc, w, P::zero());   // a zero row: multiplied and added like every other entry, from registers
                }
            }
            if (live) finish_row<4, BWD, !BWD>(acc, row, col0, H, Y, ldy, bv, rowepi, cs, o[j]);
        }
    }
    if (BWD && col_part && live) {
#pragma unroll
        for (int i = 0; i < 4; ++i) col_part[(int64_t)range * H + col0 + i] = cs[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Segment-streaming kernel (fitgnn_spmm_csr_stream_f32 / _dz_f32): the whole-subgraph kernel's algorithm, one WAVE per run of
// segments instead of one workgroup per segment -- no LDS, no barrier.
//
// A segment is a run of consecutive rows whose FIRST row is its hub: in the star-by-star layout of an --extra_node union
// (data.assemble_subgraphs_torch, layout="star") an own node of the cluster followed by the extra nodes that are there because they
// neighbour it (utils.py:235-239).  A wave streams the rows of its segments in order; the operand row of a row passes through
// the wave's registers exactly once and serves, while it is there,
//   * the row's own entry (self loop),
//   * the hub's entry for that row: the hub's accumulator is carried across the segment and takes its entries in CSR order as the
//     rows stream by (columns inside a segment are its rows, ascending) -- the order of the one-row-at-a-time kernels: same bits;
// a row's entry for its hub reads the hub's operand row kept in registers; everything else (edges between extra nodes, the other
// own nodes of the cluster) is gathered from L2 / HBM, four of a row's entries in flight.  So every operand row is read from HBM
// once by the wave that owns it and every output row is written once, as in the whole-subgraph kernel -- without its LDS windows,
// two barriers per 16-row piece and 79-120 registers: the waves are independent, 4-6 per SIMD, each requesting the operand rows (and,
// in the backward form, the `prev` slices) of four rows at a time.  Row pointers, segment starts, the row indirection and
// the CSR entries pass through register batches / tiles broadcast by v_readlane: no dependent memory access per row.
// XROW: operand row r lives at X[xrow[r]] (layer 0 on the de-duplicated table), xcol[e] = xrow[col[e]] per entry.
template <bool XROW, bool BWD, bool NOEPI>
__global__ __launch_bounds__(kThreads, BWD ? 4 : (NOEPI && !XROW) ? 6 : 5) void spmm_stream_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val, const float *__restrict__ X,
    int64_t ldx, float *__restrict__ Y, int64_t ldy, int32_t H, int32_t n_rows, int32_t nnz, const int32_t *__restrict__ seg_ptr,
    int32_t n_seg, const int32_t *__restrict__ range_seg, int32_t n_ranges, int32_t n_slabs, const float *__restrict__ bias, uint32_t epi,
    float p_drop, uint64_t seed_arg, const uint8_t *__restrict__ mask, const int32_t *__restrict__ xrow, const int32_t *__restrict__ xcol,
    const float *__restrict__ prev, float *__restrict__ col_part) {
    using P = Pack<4>;
    using T = float4;
    const uint64_t seed = fitgnn::resolve_seed(seed_arg, epi);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * kWaves + (threadIdx.x >> 6)));
    const int slab = wave % n_slabs, range = wave / n_slabs;   // both 1-KiB halves of a row are in flight together
    if (range >= n_ranges) return;
    const int s_begin = range_seg[range], s_end = range_seg[range + 1];
    const int r_begin = seg_ptr[s_begin], r_end = seg_ptr[s_end];
    const int col0 = slab * 256 + lane * 4;
    const bool live = col0 + 4 <= H;
    const int colc = live ? col0 : max(H - 4, 0);   // dead lanes read a valid column group (never stored)
    const float *Xs = X + colc;
    const float keep_scale = (epi & FITGNN_EPI_DROPOUT) ? 1.0f / (1.0f - p_drop) : 1.0f;
    const RowEpilogue rowepi{BWD ? epi : (epi & ~FITGNN_EPI_BACKWARD), keep_scale, (epi & FITGNN_EPI_DROPOUT) ? 1.0f - p_drop : 1.0f,
                             fitgnn::dropout_threshold(p_drop), seed, mask, BWD ? prev : nullptr};
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    float bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = (!NOEPI && (epi & FITGNN_EPI_BIAS) && live) ? bias[col0 + i] : 0.f;
    if (r_begin >= r_end) {
        if (BWD && col_part && live) {
#pragma unroll
            for (int i = 0; i < 4; ++i) col_part[(int64_t)range * H + col0 + i] = 0.f;
        }
        return;
    }
    auto ent = [&](int e) { return min(e, nnz - 1); };   // nnz >= 1 (checked by the launcher)
    // ---- register batches: row pointers (and the row indirection) of 64 rows, the starts of 64 segments ----
    int rb = r_begin;
    int rp = rowptr[min(rb + lane, n_rows)], rp_n = rowptr[min(rb + 64 + lane, n_rows)];
    int xr = 0, xr_n = 0;
    if (XROW) { xr = xrow[min(rb + lane, n_rows - 1)]; xr_n = xrow[min(rb + 64 + lane, n_rows - 1)]; }
    int sb = s_begin + 1;   // sg lane i = start of segment sb + i (the first segment starts at r_begin)
    int sg = seg_ptr[min(sb + lane, n_seg)];
    int si = 0;
    int next_seg = __builtin_amdgcn_readfirstlane(sg);   // = seg_ptr[s_begin + 1]
    auto src_of = [&](int row) -> int64_t {   // operand row of union row `row` (rb <= row < rb + 128)
        if (!XROW) return (int64_t)row;
        const int i = row - rb;
        return (int64_t)(i < 64 ? __builtin_amdgcn_readlane(xr, i) : __builtin_amdgcn_readlane(xr_n, i - 64));
    };
    auto row_at = [&](int row) -> T { return *reinterpret_cast<const T *>(Xs + src_of(min(row, r_end - 1)) * ldx); };
    auto prev_at = [&](int row) -> T {
        if (!BWD) return P::zero();
        return *reinterpret_cast<const T *>(prev + (uint64_t)min(row, r_end - 1) * (uint64_t)H + (uint64_t)colc);
    };
    // ---- CSR entries of the leaf rows: two 64-entry register tiles (current, next) over the sequential entry stream ----
    int T0 = __builtin_amdgcn_readfirstlane(rp);
    int t_c, t_x = 0, n_c, n_x = 0;
    float t_v, n_v;
    auto load_tiles = [&]() {
        t_c = col[ent(T0 + lane)]; t_v = val[ent(T0 + lane)];
        n_c = col[ent(T0 + 64 + lane)]; n_v = val[ent(T0 + 64 + lane)];
        if (XROW) { t_x = xcol[ent(T0 + lane)]; n_x = xcol[ent(T0 + 64 + lane)]; }
    };
    load_tiles();
    // ---- the hub: its operand row, its accumulator, a 64-entry tile of ITS entries and the cursor into them ----
    int hub = -1, h1 = 0, cur = 0, HB = 0;
    int h_c = 0, h_x = 0;
    float h_v = 0.f;
    T xhub = P::zero(), acc_h = P::zero(), o_hub = P::zero();
    auto hub_tile = [&]() {   // entries [HB, HB + 64) of the hub's row
        const int e = HB + lane;
        h_c = e < h1 ? col[e] : 0x7fffffff;
        h_v = e < h1 ? val[e] : 0.f;
        if (XROW) h_x = e < h1 ? xcol[e] : 0;
    };
    // hub entries whose column is < bound and that lie OUTSIDE the streamed rows (other segments): gathered, four in flight
    auto hub_gather = [&](int bound) {
        while (cur < h1) {
            if (cur - HB >= 64) { HB = cur; hub_tile(); }
            const int k0 = cur - HB;
            const unsigned long long in = __ballot(h_c < bound) >> k0;
            const int n_in = in == 0 ? 0 : (int)__builtin_popcountll(in);
            for (int k = k0; k < k0 + n_in; k += 4) {
                const int last = k0 + n_in - 1;
                T x[4];
                float w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kk = min(k + u, last);
                    const int64_t c = XROW ? (int64_t)__builtin_amdgcn_readlane(h_x, kk) : (int64_t)__builtin_amdgcn_readlane(h_c, kk);
                    w[u] = k + u <= last ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h_v), kk)) : 0.f;
                    x[u] = *reinterpret_cast<const T *>(Xs + c * ldx);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) P::fma(acc_h, w[u], x[u]);
            }
            cur += n_in;
            if (k0 + n_in < 64) break;   // the rest of the tile is >= bound
        }
    };
    auto hub_take = [&](int row, const T &x) {   // the hub's entry for `row`, if it is the next one
        if (cur < h1) {
            if (cur - HB >= 64) { HB = cur; hub_tile(); }
            if (__builtin_amdgcn_readlane(h_c, cur - HB) == row) {
                P::fma(acc_h, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h_v), cur - HB)), x);
                ++cur;
            }
        }
    };
    auto finish_hub = [&]() {
        if (hub < 0) return;
        hub_gather(0x7fffffff);
        if (live) finish_row<4, BWD, NOEPI>(acc_h, hub, col0, H, Y, ldy, bv, rowepi, cs, o_hub);
    };
    for (int r = r_begin; r < r_end; r += 4) {
        if (r - rb >= 64) {          // next batch of row pointers / row indirection; a group may reach three rows into the batch after
            rb += 64;
            rp = rp_n;
            rp_n = rowptr[min(rb + 64 + lane, n_rows)];
            if (XROW) { xr = xr_n; xr_n = xrow[min(rb + 64 + lane, n_rows - 1)]; }
        }
        // The operand rows (and `prev` slices) of the group's four rows are requested TOGETHER, ahead of the first row's arithmetic:
        // hipcc waits with s_waitcnt vmcnt(0) before the first use of a loaded register whenever a loop lies between request and
        // use, so a request made one row ahead would be waited for at once (one memory round trip per row); as a group the loads
        // share one round trip -- 4 KB (8 KB with `prev`) per wave in flight, 16-24 waves per CU.
        T xq[4], oq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { xq[j] = row_at(r + j); oq[j] = prev_at(r + j); }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = r + j;
            if (row >= r_end) break;   // wave-uniform
            const T x = xq[j];
            const T o_prev = oq[j];
            const int i = row - rb;   // 0 .. 66
            const int e0 = i < 64 ? __builtin_amdgcn_readlane(rp, i) : __builtin_amdgcn_readlane(rp_n, i - 64);
            const int e1 = i < 63 ? __builtin_amdgcn_readlane(rp, i + 1) : __builtin_amdgcn_readlane(rp_n, i - 63);
            if (row == r_begin || row == next_seg) {   // ---- a segment starts: this row is its hub ----
                finish_hub();
                if (row != r_begin) {
                    if (++si >= 64) { sb += 64; si = 0; sg = seg_ptr[min(sb + lane, n_seg)]; }
                    next_seg = __builtin_amdgcn_readlane(sg, si);
                }
                hub = row;
                xhub = x;
                o_hub = o_prev;
                acc_h = P::zero();
                cur = e0;
                h1 = e1;
                HB = e0;
                hub_tile();
                hub_gather(row);   // entries left of the hub (other segments), then its own entry
                hub_take(row, x);
                continue;
            }
            // ---- a row of the hub's segment ----
            T acc = P::zero();
            for (int base = e0; base < e1; base += 4) {
                int k = base - T0;
                if (k + 3 >= 128) {          // far beyond the tiles (the hub's entries were skipped): re-base
                    T0 = base;
                    k = 0;
                    load_tiles();
                } else if (k >= 64) {        // next tile
                    T0 += 64;
                    k -= 64;
                    t_c = n_c; t_v = n_v; t_x = n_x;
                    n_c = col[ent(T0 + 64 + lane)]; n_v = val[ent(T0 + 64 + lane)];
                    if (XROW) n_x = xcol[ent(T0 + 64 + lane)];
                }
                const int cnt = min(4, e1 - base);
                T v[4];
                float w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ku = k + min(u, cnt - 1);   // a last group of 1-3 entries is padded with its last entry at weight 0
                    int c, cx;
                    float wv;
                    if (ku < 64) {
                        c = __builtin_amdgcn_readlane(t_c, ku);
                        wv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t_v), ku));
                        cx = XROW ? __builtin_amdgcn_readlane(t_x, ku) : c;
                    } else {
                        c = __builtin_amdgcn_readlane(n_c, ku - 64);
                        wv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(n_v), ku - 64));
                        cx = XROW ? __builtin_amdgcn_readlane(n_x, ku - 64) : c;
                    }
                    w[u] = u < cnt ? wv : 0.f;
                    if (c == row) v[u] = x;                 // wave-uniform
                    else if (c == hub) v[u] = xhub;
                    else v[u] = *reinterpret_cast<const T *>(Xs + (int64_t)cx * ldx);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) P::fma(acc, w[u], v[u]);
            }
            if (live) finish_row<4, BWD, NOEPI>(acc, row, col0, H, Y, ldy, bv, rowepi, cs, o_prev);
            hub_take(row, x);   // the hub's entry for this row (its entries inside the segment are consumed in row order)
        }
    }
    finish_hub();
    if (BWD && col_part && live) {
#pragma unroll
        for (int i = 0; i < 4; ++i) col_part[(int64_t)range * H + col0 + i] = cs[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Two-hop backward, side table (fitgnn_two_hop_rows_f32; the product itself is spmm_block_kernel<.., TWO>).
//
// network.py:29-33 stacks  h = dropout(ELU(A_hat (x W0^T) + b0))  under the last GCNConv; with the last layer evaluated
// aggregate-first on the rows that reach the loss (ops.FusedGCNLastLayerRows) its backward hands over dAH, zero outside those rows
// (compact form: [n_sel + zero rows, H]), and the step then needs
//     dZ = (A_hat^T dAH) (.) ELU' / dropout' (h)          [R, H]   (fitgnn_spmm_rows_compact_dz_f32: reads h, writes dZ)
//     G  = A_hat^T dZ                                     [R, H]   (a plain SpMM: reads dZ, writes G)
// -- dZ itself is needed by nothing else (its column sums are the bias gradient).  Written and re-read it costs 8 H R bytes, as
// much as the two compulsory streams (h in, G out) together.  The whole-subgraph kernel makes dZ of a row as it stages the row
// into its LDS window; only the rows it has to read from elsewhere -- the loss rows, rows seen from another piece or block -- are
// computed beforehand, into the side table ZT, by the kernel below.
// (A first, segment-streaming form -- dZ in registers, one wave per run of segments -- was bit-exact too but instruction-bound:
// 560 instructions per 1-KiB row slice, 3.5 G VALU per launch, 10.9 ms + the table against 13.5 ms for the two launches.)
constexpr int32_t kNoRow = 0x7fffffff;
// ZT[i] = dZ[rows[i]] for the n_zt table rows: one wave per (row, slab); the row's entries are taken 64 at
// a time and only those whose operand row is not a zero row are visited (a centre's entries are mostly its leaves), in CSR order.
__global__ __launch_bounds__(kThreads) void two_hop_rows_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ zcol, const float *__restrict__ val, const float *__restrict__ Xc, int64_t ldx,
    int32_t zero_from, const int64_t *__restrict__ rows, int32_t n_zt, const float *__restrict__ prev, int32_t H, int32_t n_slabs, uint32_t epi,
    float p_drop, uint64_t seed_arg, const uint8_t *__restrict__ mask, float *__restrict__ ZT, int64_t ldz) {
    using P = Pack<4>;
    using T = float4;
    const uint64_t seed = fitgnn::resolve_seed(seed_arg, epi);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * kWaves + (threadIdx.x >> 6)));
    const int slab = wave % n_slabs, i = wave / n_slabs;
    if (i >= n_zt) return;
    const int row = (int)rows[i];
    const int col0 = slab * 256 + lane * 4;
    const bool live = col0 + 4 <= H;
    const int colc = live ? col0 : max(H - 4, 0);
    const float keep_scale = (epi & FITGNN_EPI_DROPOUT) ? 1.0f / (1.0f - p_drop) : 1.0f;
    const RowEpilogue rowepi{epi | FITGNN_EPI_BACKWARD, keep_scale, (epi & FITGNN_EPI_DROPOUT) ? 1.0f - p_drop : 1.0f,
                             fitgnn::dropout_threshold(p_drop), seed, mask, prev};
    const T o = *reinterpret_cast<const T *>(prev + (uint64_t)row * (uint64_t)H + (uint64_t)colc);
    const int e0 = rowptr[row], e1 = rowptr[row + 1];
    T u = P::zero();
    for (int base = e0; base < e1; base += 64) {
        const int e = base + lane;
        const int x = e < e1 ? zcol[e] : kNoRow;
        const float w = e < e1 ? val[e] : 0.f;
        unsigned long long m = __ballot(x < zero_from);
        while (m) {   // wave-uniform
            const int k = __builtin_ctzll(m);
            m &= m - 1;
            const int xk = __builtin_amdgcn_readlane(x, k);
            const float wk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), k));
            P::fma(u, wk, *reinterpret_cast<const T *>(Xc + (int64_t)xk * ldx + colc));
        }
    }
    float none[4] = {0.f, 0.f, 0.f, 0.f};
    const float bv[4] = {0.f, 0.f, 0.f, 0.f};
    const T z = epilogue_value<4, true, false>(u, row, colc, H, bv, rowepi, none, o);   // (clamped column: see the two-hop kernel)
    if (live) *reinterpret_cast<T *>(ZT + (int64_t)i * ldz + col0) = z;
}

// Direct-gather variant for very sparse batches (few non-zeros per row, e.g. PubMed-like subgraphs with
// ~3 entries per row): no LDS phase, no barrier.  Each wave owns a CONTIGUOUS run of the tile's rows, so its
// slice of the CSR is contiguous too: one vector load brings the run's row pointers, one more its (col, val)
// pairs (<= 64 of them on the fast path), and from then on every index comes out of registers via
// v_readlane -- the only dependent memory chain is paid once per wave, not once per row.  Operand rows are
// gathered straight from L2/HBM (a row's ~nnz/N consumers run on the same CU at about the same time), with
// the gathers of row i+1 issued before the FMAs of row i.
template <int VEC>
struct RowQuad {
    typename Pack<VEC>::T x[4];
    float w[4];
    int lo, hi;  // the row's entry range, relative to the wave's slice
};

// Always issues exactly four operand-row loads (missing entries re-read the row's first operand with weight 0:
// an L1 hit), so the compiler can count outstanding loads statically and emit vmcnt(N) instead of draining
// with vmcnt(0) -- which is what lets the next row's gathers stay in flight under this row's FMAs.
template <int VEC>
__device__ __forceinline__ void issue_row(RowQuad<VEC> &q, int i, int rp_v, int E0, int my_c, float my_v, const float *Xs,
                                          int64_t ldx, bool live) {
    using P = Pack<VEC>;
    using T = typename P::T;
    q.lo = __builtin_amdgcn_readlane(rp_v, i) - E0;
    q.hi = __builtin_amdgcn_readlane(rp_v, i + 1) - E0;
    const int c_first = __builtin_amdgcn_readlane(my_c, min(q.lo, 63));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool has = q.lo + j < q.hi;
        const int idx = has ? q.lo + j : min(q.lo, 63);
        const int c = has ? __builtin_amdgcn_readlane(my_c, idx) : c_first;
        const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), idx));
        q.w[j] = has ? w : 0.f;
        q.x[j] = *reinterpret_cast<const T *>(Xs + (int64_t)c * ldx);  // unconditional: Xs is clamped for dead lanes
    }
}

template <int VEC>
__device__ __forceinline__ typename Pack<VEC>::T consume_row(const RowQuad<VEC> &q, int my_c, float my_v, const float *Xs,
                                                            int64_t ldx, bool live) {
    using P = Pack<VEC>;
    using T = typename P::T;
    T acc = P::zero();
#pragma unroll
    for (int j = 0; j < 4; ++j) P::fma(acc, q.w[j], q.x[j]);
    for (int k = q.lo + 4; k < q.hi; ++k) {  // rows with more than four entries
        const int c = __builtin_amdgcn_readlane(my_c, k);
        const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), k));
        const T x = *reinterpret_cast<const T *>(Xs + (int64_t)c * ldx);
        P::fma(acc, w, x);
    }
    return acc;
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void spmm_gather_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const float *__restrict__ X, int64_t ldx, float *__restrict__ Y, int64_t ldy, int32_t H,
    const fitgnn_tile_t *__restrict__ tiles, int32_t n_tiles, int32_t tiles_per_xcd, int32_t n_slabs,
    const float *__restrict__ bias, uint32_t epi, float p_drop, uint64_t seed_arg, const uint8_t *__restrict__ mask,
    const int32_t *__restrict__ xrow) {
    const uint64_t seed = fitgnn::resolve_seed(seed_arg, epi);
    using P = Pack<VEC>;
    using T = typename P::T;
    constexpr int SLAB = 64 * VEC;
    // xrow (optional): operand row of column c is X[xrow[c]] (a de-duplicated feature table: the gathers then hit a table
    // that stays in L2 / MALL, which is where this variant beats the LDS-window kernel).
    // 1-D grid; block -> (tile, slab).  Blocks are dispatched in id order, round-robin over the 8 XCDs: give each
    // XCD a contiguous range of tiles, and make the slab the FASTEST index inside it, so that both halves of
    // every operand/output row are in flight at the same time (a slab-major order streams bytes [0,1K) of
    // every 2 KiB row first and [1K,2K) later, i.e. keeps only half of the HBM channels busy).
    const int bid = blockIdx.x;
    const int seq = bid >> 3;
    const int slab = seq % n_slabs;
    const int t = (seq / n_slabs) * 8 + (bid & 7);  // tile array position p runs on XCD p % 8 (host balances it)
    if (t >= n_tiles) return;
    const fitgnn_tile_t tile = tiles[t];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col0 = slab * SLAB + lane * VEC;
    const bool live = col0 + VEC <= H;
    const float keep_scale = (epi & FITGNN_EPI_DROPOUT) ? 1.0f / (1.0f - p_drop) : 1.0f;
    const uint32_t thresh = fitgnn::dropout_threshold(p_drop);
    const RowEpilogue rowepi{epi & ~FITGNN_EPI_BACKWARD, keep_scale, 1.0f, thresh, seed, mask, nullptr};  // forward epilogues only (launcher)
    float cs[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) cs[i] = 0.f;
    // lanes past the last column (only when H is not a multiple of the slab) load from the last valid
    // column group instead of being branched around: keeps every load unconditional (static vmcnt counts)
    const float *Xs = X + (live ? col0 : max(H - VEC, 0));
    float bv[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) bv[i] = ((epi & FITGNN_EPI_BIAS) && live) ? bias[col0 + i] : 0.f;

    // contiguous run of rows for this wave
    const int tile_rows = tile.row_end - tile.row_begin;
    const int per = (tile_rows + kWaves - 1) / kWaves;
    const int r_lo = tile.row_begin + wave * per;
    const int r_hi = min(r_lo + per, tile.row_end);
    const int nrows = r_hi - r_lo;
    if (nrows <= 0) return;
    int E0 = 0, E1 = 0, rp_v = 0;
    if (nrows <= 63) {
        rp_v = lane <= nrows ? rowptr[r_lo + lane] : 0;
        E0 = __builtin_amdgcn_readlane(rp_v, 0);
        E1 = __builtin_amdgcn_readlane(rp_v, nrows);
    }
    if (nrows <= 63 && E1 - E0 <= 64) {
        // ---- fast path: the wave's whole CSR slice sits in two registers ----
        int my_c = 0;
        float my_v = 0.f;
        if (lane < E1 - E0) { my_c = col[E0 + lane]; my_v = val[E0 + lane]; if (xrow) my_c = xrow[my_c]; }
        RowQuad<VEC> qa, qb;
        issue_row<VEC>(qa, 0, rp_v, E0, my_c, my_v, Xs, ldx, live);
        int i = 0;
        for (; i + 1 < nrows; i += 2) {
            issue_row<VEC>(qb, i + 1, rp_v, E0, my_c, my_v, Xs, ldx, live);
            const T a0 = consume_row<VEC>(qa, my_c, my_v, Xs, ldx, live);
            if (live) finish_row<VEC>(a0, r_lo + i, col0, H, Y, ldy, bv, rowepi, cs, P::zero());
            issue_row<VEC>(qa, min(i + 2, nrows - 1), rp_v, E0, my_c, my_v, Xs, ldx, live);
            const T a1 = consume_row<VEC>(qb, my_c, my_v, Xs, ldx, live);
            if (live) finish_row<VEC>(a1, r_lo + i + 1, col0, H, Y, ldy, bv, rowepi, cs, P::zero());
        }
        if (i < nrows) {
            const T a0 = consume_row<VEC>(qa, my_c, my_v, Xs, ldx, live);
            if (live) finish_row<VEC>(a0, r_lo + i, col0, H, Y, ldy, bv, rowepi, cs, P::zero());
        }
        return;
    }
    // ---- general path: any row length ----
    for (int row = r_lo; row < r_hi; ++row) {
        const int e0 = __builtin_amdgcn_readfirstlane(rowptr[row]);
        const int e1 = __builtin_amdgcn_readfirstlane(rowptr[row + 1]);
        T acc = P::zero();
        for (int base = e0; base < e1; base += 64) {
            const int cnt = min(64, e1 - base);
            int my_c = 0;
            float my_v = 0.f;
            if (lane < cnt) { my_c = col[base + lane]; my_v = val[base + lane]; if (xrow) my_c = xrow[my_c]; }
            int k = 0;
            for (; k + 4 <= cnt; k += 4) {
                const int c0 = __builtin_amdgcn_readlane(my_c, k), c1 = __builtin_amdgcn_readlane(my_c, k + 1);
                const int c2 = __builtin_amdgcn_readlane(my_c, k + 2), c3 = __builtin_amdgcn_readlane(my_c, k + 3);
                const float w0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), k));
                const float w1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), k + 1));
                const float w2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), k + 2));
                const float w3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), k + 3));
                T x0 = P::zero(), x1 = P::zero(), x2 = P::zero(), x3 = P::zero();
                if (live) {
                    x0 = *reinterpret_cast<const T *>(Xs + (int64_t)c0 * ldx);
                    x1 = *reinterpret_cast<const T *>(Xs + (int64_t)c1 * ldx);
                    x2 = *reinterpret_cast<const T *>(Xs + (int64_t)c2 * ldx);
                    x3 = *reinterpret_cast<const T *>(Xs + (int64_t)c3 * ldx);
                }
                P::fma(acc, w0, x0); P::fma(acc, w1, x1); P::fma(acc, w2, x2); P::fma(acc, w3, x3);
            }
            for (; k < cnt; ++k) {
                const int c = __builtin_amdgcn_readlane(my_c, k);
                const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), k));
                T x = P::zero();
                if (live) x = *reinterpret_cast<const T *>(Xs + (int64_t)c * ldx);
                P::fma(acc, w, x);
            }
        }
        if (live) finish_row<VEC>(acc, row, col0, H, Y, ldy, bv, rowepi, cs, P::zero());
    }
}

inline size_t lds_bytes_for(int lds_rows, int slab_floats, int mpr) {
    return (size_t)lds_rows * slab_floats * 4 + (size_t)((lds_rows + 1 + 3) / 4 * 4) * 4 + (size_t)lds_rows * mpr * 8;
}

template <int VEC, int B, int MPR, bool PLAIN, bool NOEPI = false>
int launch_tile(const int32_t *rowptr, const int32_t *col, const float *val, const float *X, int64_t ldx, float *Y,
                int64_t ldy, int32_t H, const fitgnn_tile_t *tiles, int32_t n_tiles, const int32_t *lcol,
                const int32_t *win_cols, const int32_t *xrow, int32_t lds_rows, int n_slabs, int tiles_per_xcd,
                const float *bias, uint32_t epi, float p_drop, uint64_t seed, const uint8_t *mask, const float *prev, float *col_part,
                int32_t zero_from, hipStream_t s) {
    constexpr int SLAB = 64 * VEC;
    size_t lds_bytes = lds_bytes_for(lds_rows, SLAB, MPR);
    // the backward epilogue's column sums pass through LDS once more (write_col_part: kWaves x 64 x VEC floats): a window of
    // fewer than four rows would be smaller than that scratch
    if (epi & FITGNN_EPI_BACKWARD) lds_bytes = std::max(lds_bytes, (size_t)kWaves * 64 * VEC * sizeof(float));
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)spmm_tile_kernel<VEC, B, MPR, PLAIN, NOEPI>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid(tiles_per_xcd * 8 * n_slabs);
    hipLaunchKernelGGL((spmm_tile_kernel<VEC, B, MPR, PLAIN, NOEPI>), grid, dim3(kThreads), lds_bytes, s, rowptr, col, val, X, ldx, Y, ldy,
                       H, tiles, n_tiles, tiles_per_xcd, n_slabs, lds_rows, lcol, win_cols, xrow, bias, epi, p_drop, seed, mask, prev, col_part, zero_from);
    return (int)hipGetLastError();
}

template <int VEC>
int launch(const int32_t *rowptr, const int32_t *col, const float *val, const float *X, int64_t ldx, float *Y,
           int64_t ldy, int32_t H, const fitgnn_tile_t *tiles, int32_t n_tiles, const int32_t *lcol, const int32_t *win_cols,
           const int32_t *xrow, int32_t window_rows, const float *bias, uint32_t epi, float p_drop, uint64_t seed, const uint8_t *mask,
           const float *prev, float *col_part, int32_t zero_from, hipStream_t s) {
    constexpr int SLAB = 64 * VEC;
    const int n_slabs = (H + SLAB - 1) / SLAB;
    const int tiles_per_xcd = (n_tiles + 7) / 8;
    if (epi & FITGNN_SPMM_GATHER) {
        dim3 grid(tiles_per_xcd * 8 * n_slabs);
        hipLaunchKernelGGL(spmm_gather_kernel<VEC>, grid, dim3(kThreads), 0, s, rowptr, col, val, X, ldx, Y, ldy, H, tiles,
                           n_tiles, tiles_per_xcd, n_slabs, bias, epi, p_drop, seed, mask, xrow);
        return (int)hipGetLastError();
    }
    const int lds_rows = window_rows > 0 ? std::min(window_rows, kMaxWindowRows) : kDefaultWindowRows;
    // a plain product (no epilogue flag): the instantiation whose rows leave with a bare store
    if (lds_rows <= kSmallWindowRows && !lcol && !win_cols && !xrow &&
        (epi & (FITGNN_EPI_BIAS | FITGNN_EPI_ELU | FITGNN_EPI_DROPOUT | FITGNN_EPI_BACKWARD)) == 0)
        return launch_tile<VEC, 4, 16, true, true>(rowptr, col, val, X, ldx, Y, ldy, H, tiles, n_tiles, lcol, win_cols, xrow, lds_rows, n_slabs,
                                                   tiles_per_xcd, bias, epi, p_drop, seed, mask, prev, col_part, zero_from, s);
    if (lds_rows <= kSmallWindowRows && !lcol && !win_cols && !xrow)
        return launch_tile<VEC, 4, 16, true>(rowptr, col, val, X, ldx, Y, ldy, H, tiles, n_tiles, lcol, win_cols, xrow, lds_rows, n_slabs,
                                             tiles_per_xcd, bias, epi, p_drop, seed, mask, prev, col_part, zero_from, s);
    if (lds_rows <= kSmallWindowRows)
        return launch_tile<VEC, 4, 16, false>(rowptr, col, val, X, ldx, Y, ldy, H, tiles, n_tiles, lcol, win_cols, xrow, lds_rows, n_slabs,
                                       tiles_per_xcd, bias, epi, p_drop, seed, mask, prev, col_part, zero_from, s);
    return launch_tile<VEC, 8, 32, false>(rowptr, col, val, X, ldx, Y, ldy, H, tiles, n_tiles, lcol, win_cols, xrow, lds_rows, n_slabs,
                                   tiles_per_xcd, bias, epi, p_drop, seed, mask, prev, col_part, zero_from, s);
}

}  // namespace

extern "C" int fitgnn_spmm_default_window_rows(void) { return kDefaultWindowRows; }

extern "C" int fitgnn_spmm_max_window_rows(int32_t H) {
    (void)H;
    return kMaxWindowRows;
}

namespace {
int spmm_csr_impl(const int32_t *rowptr, const int32_t *col, const float *val, const float *X, int64_t ldx, float *Y, int64_t ldy,
                  int32_t n_rows, int32_t H, const fitgnn_tile_t *tiles, int32_t n_tiles, const int32_t *lcol, const int32_t *win_cols,
                  const int32_t *xrow, int32_t window_rows, const float *bias, uint32_t epilogue, float p_drop, uint64_t seed,
                  const uint8_t *mask, const float *prev, float *col_part, int32_t zero_from, void *stream) {
    if (n_rows < 0 || H < 0 || n_tiles < 0 || window_rows < 0) return FITGNN_E_BADARG;
    if (!xrow) zero_from = -1;
    if (zero_from >= 0) epilogue &= ~FITGNN_SPMM_GATHER;  // the direct-gather variant knows no zero rows
    if (n_rows == 0 || H == 0 || n_tiles == 0) return 0;
    // col/val may be NULL only for a matrix without non-zeros (they are then never dereferenced)
    if (!rowptr || !X || !Y || !tiles) return FITGNN_E_BADARG;
    if ((epilogue & FITGNN_EPI_BIAS) && !bias) return FITGNN_E_BADARG;
    if ((epilogue & FITGNN_EPI_DROPOUT) && !(p_drop >= 0.f && p_drop < 1.f)) return FITGNN_E_BADARG;
    if (ldx < H || ldy < H) return FITGNN_E_BADARG;
    if ((lcol == nullptr) != (win_cols == nullptr) && lcol == nullptr) return FITGNN_E_BADARG;  // win_cols needs lcol
    if (epilogue & FITGNN_EPI_BACKWARD) {  // prev is indexed like a contiguous [n_rows x H] matrix; not for the direct-gather variant
        if (!prev || (epilogue & (FITGNN_SPMM_GATHER | FITGNN_EPI_BIAS)) || ((uintptr_t)prev % 16) != 0 || (H % 4) != 0) return FITGNN_E_BADARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (H % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && (((uintptr_t)X | (uintptr_t)Y) % 16 == 0);
    if ((epilogue & FITGNN_EPI_BACKWARD) && !vec) return FITGNN_E_ALIGN;
    if (vec) return launch<4>(rowptr, col, val, X, ldx, Y, ldy, H, tiles, n_tiles, lcol, win_cols, xrow, window_rows, bias, epilogue, p_drop, seed, mask, prev, col_part, zero_from, s);
    return launch<1>(rowptr, col, val, X, ldx, Y, ldy, H, tiles, n_tiles, lcol, win_cols, xrow, window_rows, bias, epilogue, p_drop, seed, mask, prev, col_part, zero_from, s);
}

int spmm_blocks_impl(const int32_t *rowptr, const int32_t *col, const float *val, const float *X, int64_t ldx, float *Y, int64_t ldy,
                     int32_t n_rows, int32_t H, const fitgnn_block_t *blocks, int32_t n_blocks, const int32_t *long_rows,
                     const int32_t *xrow, const int32_t *xcol, const float *bias, uint32_t epilogue, float p_drop, uint64_t seed,
                     const uint8_t *mask, const float *prev, float *col_part, int32_t zero_from, void *stream) {
    if (n_rows < 0 || H < 0 || n_blocks < 0) return FITGNN_E_BADARG;
    if (xcol && !xrow) return FITGNN_E_BADARG;
    if (n_rows == 0 || H == 0 || n_blocks == 0) return 0;
    if (!rowptr || !col || !val || !X || !Y || !blocks) return FITGNN_E_BADARG;
    if ((epilogue & FITGNN_EPI_BIAS) && !bias) return FITGNN_E_BADARG;
    if ((epilogue & FITGNN_EPI_DROPOUT) && !(p_drop >= 0.f && p_drop < 1.f)) return FITGNN_E_BADARG;
    if (ldx < H || ldy < H) return FITGNN_E_BADARG;
    if ((H % 4) != 0 || (ldx % 4) != 0 || (ldy % 4) != 0) return FITGNN_E_BADARG;
    if ((((uintptr_t)X | (uintptr_t)Y) % 16) != 0) return FITGNN_E_ALIGN;
    if ((epilogue & FITGNN_EPI_BACKWARD) && (!prev || (epilogue & FITGNN_EPI_BIAS) || ((uintptr_t)prev % 16) != 0)) return FITGNN_E_BADARG;
    const int n_slabs = (H + 255) / 256;
    const dim3 grid((unsigned)((n_blocks + 7) / 8 * 8) * n_slabs);
#define FITGNN_LAUNCH_BLK(XR, BW)                                                                                                    \
    hipLaunchKernelGGL((spmm_block_kernel<XR, BW>), grid, dim3(kThreads), 0, (hipStream_t)stream, rowptr, col, val, X, ldx, Y, ldy, H, \
                       blocks, n_blocks, long_rows, n_slabs, bias, epilogue, p_drop, seed, mask, xrow, xcol, prev, col_part, zero_from)
    const bool bwd = (epilogue & FITGNN_EPI_BACKWARD) != 0;
    const bool noepi = (epilogue & (FITGNN_EPI_BIAS | FITGNN_EPI_ELU | FITGNN_EPI_DROPOUT | FITGNN_EPI_BACKWARD)) == 0;
    if (xrow) { if (bwd) FITGNN_LAUNCH_BLK(true, true); else FITGNN_LAUNCH_BLK(true, false); }
    else if (bwd) FITGNN_LAUNCH_BLK(false, true);
    else if (noepi) hipLaunchKernelGGL((spmm_block_kernel<false, false, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, rowptr, col, val, X, ldx, Y,
                                       ldy, H, blocks, n_blocks, long_rows, n_slabs, bias, epilogue, p_drop, seed, mask, xrow, xcol, prev, col_part,
                                       zero_from);
    else FITGNN_LAUNCH_BLK(false, false);
#undef FITGNN_LAUNCH_BLK
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int fitgnn_spmm_csr_f32(const int32_t *rowptr, const int32_t *col, const float *val, const float *X,
                                   int64_t ldx, float *Y, int64_t ldy, int32_t n_rows, int32_t H,
                                   const fitgnn_tile_t *tiles, int32_t n_tiles, const int32_t *lcol,
                                   const int32_t *win_cols, const int32_t *xrow, int32_t xrow_zero_from, int32_t window_rows,
                                   const float *bias, uint32_t epilogue, float p_drop, uint64_t seed, const uint8_t *mask, void *stream) {
    if (epilogue & FITGNN_EPI_BACKWARD) return FITGNN_E_BADARG;  // fitgnn_spmm_csr_dz_f32
    return spmm_csr_impl(rowptr, col, val, X, ldx, Y, ldy, n_rows, H, tiles, n_tiles, lcol, win_cols, xrow, window_rows, bias, epilogue,
                         p_drop, seed, mask, nullptr, nullptr, xrow_zero_from, stream);
}

extern "C" int fitgnn_spmm_csr_blocks_f32(const int32_t *rowptr, const int32_t *col, const float *val, const float *X,
                                          int64_t ldx, float *Y, int64_t ldy, int32_t n_rows, int32_t H,
                                          const fitgnn_block_t *blocks, int32_t n_blocks, const int32_t *long_rows,
                                          const int32_t *xrow, const int32_t *xcol, int32_t xrow_zero_from, const float *bias,
                                          uint32_t epilogue, float p_drop, uint64_t seed, const uint8_t *mask, void *stream) {
    if (epilogue & FITGNN_EPI_BACKWARD) return FITGNN_E_BADARG;  // fitgnn_spmm_csr_blocks_dz_f32
    return spmm_blocks_impl(rowptr, col, val, X, ldx, Y, ldy, n_rows, H, blocks, n_blocks, long_rows, xrow, xcol, bias, epilogue, p_drop, seed,
                            mask, nullptr, nullptr, xrow ? xrow_zero_from : -1, stream);
}

// The same products as the input gradient of a fused layer: Y = (A @ X) * dropout' * ELU'(prev), see finish_row.
extern "C" int fitgnn_spmm_csr_dz_f32(const int32_t *rowptr, const int32_t *col, const float *val, const float *X, int64_t ldx,
                                      float *Y, int64_t ldy, int32_t n_rows, int32_t H, const fitgnn_tile_t *tiles, int32_t n_tiles,
                                      const int32_t *lcol, const int32_t *win_cols, const int32_t *xrow, int32_t xrow_zero_from,
                                      int32_t window_rows, const float *prev, uint32_t epilogue, float p_drop, uint64_t seed,
                                      const uint8_t *mask, float *col_part, void *stream) {
    return spmm_csr_impl(rowptr, col, val, X, ldx, Y, ldy, n_rows, H, tiles, n_tiles, lcol, win_cols, xrow, window_rows, nullptr,
                         (epilogue & ~FITGNN_SPMM_GATHER) | FITGNN_EPI_BACKWARD, p_drop, seed, mask, prev, col_part, xrow_zero_from, stream);
}

extern "C" int fitgnn_spmm_csr_blocks_dz_f32(const int32_t *rowptr, const int32_t *col, const float *val, const float *X, int64_t ldx,
                                             float *Y, int64_t ldy, int32_t n_rows, int32_t H, const fitgnn_block_t *blocks,
                                             int32_t n_blocks, const int32_t *long_rows, const int32_t *xrow, const int32_t *xcol,
                                             int32_t xrow_zero_from, const float *prev, uint32_t epilogue, float p_drop, uint64_t seed,
                                             const uint8_t *mask, float *col_part, void *stream) {
    return spmm_blocks_impl(rowptr, col, val, X, ldx, Y, ldy, n_rows, H, blocks, n_blocks, long_rows, xrow, xcol, nullptr,
                            epilogue | FITGNN_EPI_BACKWARD, p_drop, seed, mask, prev, col_part, xrow ? xrow_zero_from : -1, stream);
}

namespace {
// ranges of consecutive rows, one per (wave, slab): enough waves to fill 256 CUs x 32 waves several times over on a large batch,
// at least 32 rows each (64 left the 90 549-row S-pubmed union with 11 waves per CU: 122 -> 93 us for its compact-operand launch; 16
// and 8 are no faster)
inline void rows_plan(int32_t n_rows, int32_t *rows_per_range, int32_t *n_ranges) {
    const int want = 8192;
    int per = (n_rows + want - 1) / want;
    int least = 32;
    if (const char *e = getenv("FITGNN_ROWS_MIN")) least = atoi(e) > 0 ? atoi(e) : 32;   // experiments
    if (per < least) per = least;
    *rows_per_range = per;
    *n_ranges = (n_rows + per - 1) / per;
}

int spmm_rows_impl(const int32_t *rowptr, const int32_t *xcol, const float *val, int64_t nnz, const float *X, int64_t ldx, int32_t zero_from,
                   float *Y, int64_t ldy, int32_t n_rows, int32_t H, const float *prev, uint32_t epilogue, float p_drop, uint64_t seed,
                   const uint8_t *mask, float *col_part, bool bwd, void *stream) {
    if (n_rows < 0 || H < 0 || nnz < 0 || nnz > 0x7fffffffLL || zero_from < 0) return FITGNN_E_BADARG;
    if (n_rows == 0 || H == 0) return 0;
    if (!rowptr || !X || !Y || (nnz > 0 && (!xcol || !val))) return FITGNN_E_BADARG;
    if ((H % 4) != 0 || (ldx % 4) != 0 || (ldy % 4) != 0 || ldx < H || ldy < H) return FITGNN_E_BADARG;
    if ((((uintptr_t)X | (uintptr_t)Y) % 16) != 0) return FITGNN_E_ALIGN;
    if ((epilogue & FITGNN_EPI_DROPOUT) && !(p_drop >= 0.f && p_drop < 1.f)) return FITGNN_E_BADARG;
    if (bwd && (!prev || ((uintptr_t)prev % 16) != 0 || (epilogue & FITGNN_EPI_BIAS))) return FITGNN_E_BADARG;
    int32_t per, n_ranges;
    rows_plan(n_rows, &per, &n_ranges);
    const int n_slabs = (H + 255) / 256;
    const dim3 grid((unsigned)(((int64_t)n_ranges * n_slabs + kWaves - 1) / kWaves));
    if (bwd)
        hipLaunchKernelGGL(spmm_rows_compact_kernel<true>, grid, dim3(kThreads), 0, (hipStream_t)stream, rowptr, xcol, val, X, ldx, zero_from, Y, ldy,
                           H, n_rows, (int32_t)nnz, n_slabs, per, n_ranges, epilogue | FITGNN_EPI_BACKWARD, p_drop, seed, mask, prev, col_part);
    else
        hipLaunchKernelGGL(spmm_rows_compact_kernel<false>, grid, dim3(kThreads), 0, (hipStream_t)stream, rowptr, xcol, val, X, ldx, zero_from, Y,
                           ldy, H, n_rows, (int32_t)nnz, n_slabs, per, n_ranges, 0u, 0.f, (uint64_t)0, (const uint8_t *)nullptr,
                           (const float *)nullptr, (float *)nullptr);
    return (int)hipGetLastError();
}
}  // namespace

namespace {
int spmm_stream_impl(const int32_t *rowptr, const int32_t *col, const float *val, int64_t nnz, const float *X, int64_t ldx, float *Y,
                     int64_t ldy, int32_t n_rows, int32_t H, const int32_t *seg_ptr, int32_t n_seg, const int32_t *range_seg, int32_t n_ranges,
                     const int32_t *xrow, const int32_t *xcol, const float *bias, uint32_t epilogue, float p_drop, uint64_t seed,
                     const uint8_t *mask, const float *prev, float *col_part, void *stream) {
    if (n_rows < 0 || H < 0 || nnz < 0 || nnz > 0x7fffffffLL || n_seg < 0 || n_ranges < 0) return FITGNN_E_BADARG;
    if ((xrow == nullptr) != (xcol == nullptr)) return FITGNN_E_BADARG;
    if (n_rows == 0 || H == 0 || n_ranges == 0) return 0;
    if (!rowptr || !col || !val || nnz == 0 || !X || !Y || !seg_ptr || !range_seg || n_seg == 0) return FITGNN_E_BADARG;
    if ((epilogue & FITGNN_EPI_BIAS) && !bias) return FITGNN_E_BADARG;
    if ((epilogue & FITGNN_EPI_DROPOUT) && !(p_drop >= 0.f && p_drop < 1.f)) return FITGNN_E_BADARG;
    if ((H % 4) != 0 || (ldx % 4) != 0 || (ldy % 4) != 0 || ldx < H || ldy < H) return FITGNN_E_BADARG;
    if ((((uintptr_t)X | (uintptr_t)Y) % 16) != 0) return FITGNN_E_ALIGN;
    const bool bwd = (epilogue & FITGNN_EPI_BACKWARD) != 0;
    if (bwd && (!prev || (epilogue & FITGNN_EPI_BIAS) || ((uintptr_t)prev % 16) != 0)) return FITGNN_E_BADARG;
    const bool noepi = (epilogue & (FITGNN_EPI_BIAS | FITGNN_EPI_ELU | FITGNN_EPI_DROPOUT | FITGNN_EPI_BACKWARD)) == 0;
    const int n_slabs = (H + 255) / 256;
    const dim3 grid((unsigned)(((int64_t)n_ranges * n_slabs + kWaves - 1) / kWaves));
#define FITGNN_LAUNCH_STREAM(XR, BW, NE)                                                                                                  \
    hipLaunchKernelGGL((spmm_stream_kernel<XR, BW, NE>), grid, dim3(kThreads), 0, (hipStream_t)stream, rowptr, col, val, X, ldx, Y, ldy, H, \
                       n_rows, (int32_t)nnz, seg_ptr, n_seg, range_seg, n_ranges, n_slabs, bias, epilogue, p_drop, seed, mask, xrow, xcol, prev,  \
                       col_part)
    if (xrow) { if (bwd) FITGNN_LAUNCH_STREAM(true, true, false); else if (noepi) FITGNN_LAUNCH_STREAM(true, false, true); else FITGNN_LAUNCH_STREAM(true, false, false); }
    else if (bwd) FITGNN_LAUNCH_STREAM(false, true, false);
    else if (noepi) FITGNN_LAUNCH_STREAM(false, false, true);
    else FITGNN_LAUNCH_STREAM(false, false, false);
#undef FITGNN_LAUNCH_STREAM
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int fitgnn_spmm_csr_stream_f32(const int32_t *rowptr, const int32_t *col, const float *val, int64_t nnz, const float *X, int64_t ldx,
                                          float *Y, int64_t ldy, int32_t n_rows, int32_t H, const int32_t *seg_ptr, int32_t n_seg,
                                          const int32_t *range_seg, int32_t n_ranges, const int32_t *xrow, const int32_t *xcol,
                                          const float *bias, uint32_t epilogue, float p_drop, uint64_t seed, const uint8_t *mask, void *stream) {
    if (epilogue & FITGNN_EPI_BACKWARD) return FITGNN_E_BADARG;  // fitgnn_spmm_csr_stream_dz_f32
    return spmm_stream_impl(rowptr, col, val, nnz, X, ldx, Y, ldy, n_rows, H, seg_ptr, n_seg, range_seg, n_ranges, xrow, xcol, bias,
                            epilogue & ~FITGNN_SPMM_GATHER, p_drop, seed, mask, nullptr, nullptr, stream);
}

extern "C" int fitgnn_spmm_csr_stream_dz_f32(const int32_t *rowptr, const int32_t *col, const float *val, int64_t nnz, const float *X,
                                             int64_t ldx, float *Y, int64_t ldy, int32_t n_rows, int32_t H, const int32_t *seg_ptr,
                                             int32_t n_seg, const int32_t *range_seg, int32_t n_ranges, const int32_t *xrow,
                                             const int32_t *xcol, const float *prev, uint32_t epilogue, float p_drop, uint64_t seed,
                                             const uint8_t *mask, float *col_part, void *stream) {
    return spmm_stream_impl(rowptr, col, val, nnz, X, ldx, Y, ldy, n_rows, H, seg_ptr, n_seg, range_seg, n_ranges, xrow, xcol, nullptr,
                            (epilogue & ~FITGNN_SPMM_GATHER) | FITGNN_EPI_BACKWARD, p_drop, seed, mask, prev, col_part, stream);
}

extern "C" int32_t fitgnn_spmm_rows_compact_parts(int32_t n_rows) {
    if (n_rows <= 0) return 0;
    int32_t per, n_ranges;
    rows_plan(n_rows, &per, &n_ranges);
    return n_ranges;
}

extern "C" int fitgnn_spmm_rows_compact_f32(const int32_t *rowptr, const int32_t *xcol, const float *val, int64_t nnz, const float *X,
                                            int64_t ldx, int32_t zero_from, float *Y, int64_t ldy, int32_t n_rows, int32_t H, void *stream) {
    return spmm_rows_impl(rowptr, xcol, val, nnz, X, ldx, zero_from, Y, ldy, n_rows, H, nullptr, 0u, 0.f, 0, nullptr, nullptr, false, stream);
}

extern "C" int fitgnn_spmm_rows_compact_dz_f32(const int32_t *rowptr, const int32_t *xcol, const float *val, int64_t nnz, const float *X,
                                               int64_t ldx, int32_t zero_from, float *Y, int64_t ldy, int32_t n_rows, int32_t H,
                                               const float *prev, uint32_t epilogue, float p_drop, uint64_t seed, const uint8_t *mask,
                                               float *col_part, void *stream) {
    return spmm_rows_impl(rowptr, xcol, val, nnz, X, ldx, zero_from, Y, ldy, n_rows, H, prev, epilogue & ~FITGNN_SPMM_GATHER, p_drop, seed, mask,
                          col_part, true, stream);
}

extern "C" int fitgnn_two_hop_rows_f32(const int32_t *rowptr, const int32_t *zcol, const float *val, const float *Xc, int64_t ldx, int32_t zero_from,
                                      const int64_t *zt_rows, int32_t n_zt, const float *prev, int32_t H, uint32_t epilogue, float p_drop,
                                      uint64_t seed, const uint8_t *mask, float *ZT, int64_t ldz, void *stream) {
    if (H < 0 || n_zt < 0 || zero_from < 0) return FITGNN_E_BADARG;
    if (n_zt == 0 || H == 0) return 0;
    if (!rowptr || !zcol || !val || !Xc || !zt_rows || !prev || !ZT) return FITGNN_E_BADARG;
    epilogue &= ~(uint32_t)FITGNN_SPMM_GATHER;
    if (epilogue & (FITGNN_EPI_BIAS | FITGNN_EPI_BACKWARD)) return FITGNN_E_BADARG;   // the FORWARD's ELU / dropout flags
    if ((epilogue & FITGNN_EPI_DROPOUT) && !(p_drop >= 0.f && p_drop < 1.f)) return FITGNN_E_BADARG;
    if ((H % 4) != 0 || (ldx % 4) != 0 || (ldz % 4) != 0 || ldx < H || ldz < H) return FITGNN_E_BADARG;
    if ((((uintptr_t)Xc | (uintptr_t)prev | (uintptr_t)ZT) % 16) != 0) return FITGNN_E_ALIGN;
    const int n_slabs = (H + 255) / 256;
    const dim3 grid0((unsigned)(((int64_t)n_zt * n_slabs + kWaves - 1) / kWaves));
    hipLaunchKernelGGL(two_hop_rows_kernel, grid0, dim3(kThreads), 0, (hipStream_t)stream, rowptr, zcol, val, Xc, ldx, zero_from, zt_rows, n_zt, prev, H,
                       n_slabs, epilogue, p_drop, seed, mask, ZT, ldz);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_spmm_two_hop_blocks_f32(const int32_t *rowptr, const int32_t *col, const float *val, const float *ZT, int64_t ldz, float *Y,
                                              int64_t ldy, int32_t n_rows, int32_t H, const fitgnn_block_t *blocks, int32_t n_blocks,
                                              const int32_t *long_rows, const int32_t *zrow, const int32_t *zcol, const float *prev,
                                              const float *Xc, int64_t ldx, int32_t zero_from, const int32_t *row_p, const float *row_w,
                                              uint32_t epilogue, float p_drop, uint64_t seed, const uint8_t *mask, float *col_part,
                                              void *stream) {
    if (n_rows < 0 || H < 0 || n_blocks < 0 || zero_from < 0) return FITGNN_E_BADARG;
    if (n_rows == 0 || H == 0 || n_blocks == 0) return 0;
    if (!rowptr || !col || !val || !ZT || !Y || !blocks || !zrow || !zcol || !prev || !Xc || !row_p || !row_w) return FITGNN_E_BADARG;
    epilogue &= ~(uint32_t)FITGNN_SPMM_GATHER;
    if (epilogue & (FITGNN_EPI_BIAS | FITGNN_EPI_BACKWARD)) return FITGNN_E_BADARG;   // the FORWARD's ELU / dropout flags
    if ((epilogue & FITGNN_EPI_DROPOUT) && !(p_drop >= 0.f && p_drop < 1.f)) return FITGNN_E_BADARG;
    if ((H % 4) != 0 || (ldx % 4) != 0 || (ldy % 4) != 0 || (ldz % 4) != 0 || ldx < H || ldy < H || ldz < H) return FITGNN_E_BADARG;
    if ((((uintptr_t)Xc | (uintptr_t)Y | (uintptr_t)prev | (uintptr_t)ZT) % 16) != 0) return FITGNN_E_ALIGN;
    const int n_slabs = (H + 255) / 256;
    const dim3 grid((unsigned)((n_blocks + 7) / 8 * 8) * n_slabs);
    hipLaunchKernelGGL((spmm_block_kernel<true, false, true, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, rowptr, col, val, ZT, ldz, Y, ldy, H,
                       blocks, n_blocks, long_rows, n_slabs, (const float *)nullptr, epilogue, p_drop, seed, mask, zrow, zcol, prev, col_part, -1, Xc,
                       ldx, row_p, row_w, zero_from);
    return (int)hipGetLastError();
}

#ifdef FITGNN_SPMM_STAMPS
extern "C" int fitgnn_debug_spmm_buffer(unsigned long long *device_buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_spmm_dbg), &device_buf, sizeof(device_buf));
}
#endif
