// assemble.hip -- the induced edges of all cluster subgraphs at once (gfx950 only): SURVEY §8 f1, the device half of
// utils.py:185-267 (`subgraph` / data.subgraph(value): an edge (x, y) of the graph belongs to cluster c's subgraph iff both ends are
// members of c -- own nodes or, with --extra_node, 1-hop neighbours).
//
// The reference's neighbour() rescans all E edges per node (utils.py:52-56).  Here a member row r = (cluster c, node x) walks x's
// adjacency list once and looks every neighbour y up in c's OWN member list -- the cluster's rows are a contiguous, ascending run
// of the membership keys, ~100 entries at S-products: a 7-step binary search that stays in L1 -- instead of searching the whole
// 8.2 M-entry key array (23 steps) through multi-GB temporaries of (row, neighbour) pairs.  One wavefront per row, the neighbours on
// its lanes; two passes (count, then fill behind an exclusive scan of the counts) so that the edge list comes out in (row, neighbour)
// order with no atomics: the same order as the torch composition it replaces (data.assemble_subgraphs_torch).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "fitgnn_hip.h"

namespace {

// position of node y in key_node[lo, hi) (ascending), or -1
__device__ __forceinline__ int64_t find_member(const int64_t *__restrict__ key_node, int64_t lo, int64_t hi, int64_t y) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int64_t v = key_node[mid];
        if (v < y) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

template <bool FILL>
__global__ __launch_bounds__(256) void induced_edges_kernel(const int64_t *__restrict__ adj_ptr, const int64_t *__restrict__ adj,
                                                            const int64_t *__restrict__ row_node, const int64_t *__restrict__ row_cluster,
                                                            const int64_t *__restrict__ cl_ptr, const int64_t *__restrict__ key_node,
                                                            const int64_t *__restrict__ inv, int64_t n_rows, int32_t *__restrict__ cnt,
                                                            const int64_t *__restrict__ off, int64_t *__restrict__ e_src,
                                                            int64_t *__restrict__ e_dst) {
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (r >= n_rows) return;
    const int64_t x = row_node[r], c = row_cluster[r];
    const int64_t a0 = adj_ptr[x], a1 = adj_ptr[x + 1];
    const int64_t lo = cl_ptr[c], hi = cl_ptr[c + 1];
    int64_t out = FILL ? off[r] : 0;
    int total = 0;
    for (int64_t base = a0; base < a1; base += 64) {
        const int64_t k = base + lane;
        int64_t pos = -1;
        if (k < a1) {
            const int64_t y = adj[k];
            const int64_t p = find_member(key_node, lo, hi, y);
            if (p < hi && key_node[p] == y) pos = p;
        }
        const unsigned long long m = __ballot(pos >= 0);
        if (FILL) {
            if (pos >= 0) {
                const int before = __popcll(m & ((1ull << lane) - 1ull));
                e_src[out + before] = r;
                e_dst[out + before] = inv ? inv[pos] : pos;
            }
            out += __popcll(m);
        } else {
            total += __popcll(m);
        }
    }
    if (!FILL && lane == 0) cnt[r] = total;
}

}  // namespace

extern "C" int fitgnn_induced_edges_count(const int64_t *adj_ptr, const int64_t *adj, const int64_t *row_node, const int64_t *row_cluster,
                                          const int64_t *cl_ptr, const int64_t *key_node, int64_t n_rows, int32_t *cnt, void *stream) {
    if (n_rows < 0) return FITGNN_E_BADARG;
    if (n_rows == 0) return 0;
    if (!adj_ptr || !adj || !row_node || !row_cluster || !cl_ptr || !key_node || !cnt) return FITGNN_E_BADARG;
    hipLaunchKernelGGL(induced_edges_kernel<false>, dim3((unsigned)((n_rows * 64 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, adj_ptr, adj,
                       row_node, row_cluster, cl_ptr, key_node, (const int64_t *)nullptr, n_rows, cnt, (const int64_t *)nullptr,
                       (int64_t *)nullptr, (int64_t *)nullptr);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_induced_edges_fill(const int64_t *adj_ptr, const int64_t *adj, const int64_t *row_node, const int64_t *row_cluster,
                                         const int64_t *cl_ptr, const int64_t *key_node, const int64_t *inv, int64_t n_rows,
                                         const int64_t *off, int64_t *e_src, int64_t *e_dst, void *stream) {
    if (n_rows < 0) return FITGNN_E_BADARG;
    if (n_rows == 0) return 0;
    if (!adj_ptr || !adj || !row_node || !row_cluster || !cl_ptr || !key_node || !off || !e_src || !e_dst) return FITGNN_E_BADARG;
    hipLaunchKernelGGL(induced_edges_kernel<true>, dim3((unsigned)((n_rows * 64 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, adj_ptr, adj,
                       row_node, row_cluster, cl_ptr, key_node, inv, n_rows, (int32_t *)nullptr, off, e_src, e_dst);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// A training batch of a graph-level dataset assembled ON THE DEVICE from the dataset's global arrays (fitgnn_batch_offsets /
// fitgnn_batch_gather): run.py:710 builds its loaders with shuffle=True, so every epoch's batches hold other graphs.  Rebuilt on the
// host that is ~3 ms per batch of 128 molecules (sort, normalisation, transposition, tile packing) and the step must run eagerly; here
// a batch is a gather: the graphs of a dataset keep their rows, CSR entries, row tiles and pooled-row lists contiguous and relative to
// their own first row, so the block-diagonal batch of ANY 128 graphs is those pieces behind four exclusive scans (rows, entries, tiles,
// pooled rows).  The buffers have fixed capacities (rows / entries / tiles past the batch's totals are empty: rows without entries,
// tiles without rows), every launch of the step keeps its shape, and ONE captured hipGraph -- these two launches, then forward, loss,
// backward, Adam -- serves every batch of every epoch; the host only uploads the epoch's permutation.
namespace {

constexpr int kBatchMaxGraphs = 1024;

// off[k * (B + 1) + i], k = 0 rows, 1 entries, 2 tiles, 3 pooled rows: exclusive scans over the batch's graphs, off[k][B] = totals;
// gid[i] = the i-th graph of the batch.  The batch is perm[step * B .. + B) with step = *step_idx, which this kernel advances.
// loss_slot / loss_sum (optional): the previous step's loss is added to the epoch's sum and the slot cleared (the step's loss kernel
// writes it again further down the same hipGraph).
__global__ __launch_bounds__(kBatchMaxGraphs) void batch_offsets_kernel(const int64_t *__restrict__ perm, int32_t *__restrict__ step_idx,
                                                                      int32_t B, const int32_t *__restrict__ g_row_ptr,
                                                                      const int32_t *__restrict__ g_nnz_ptr,
                                                                      const int32_t *__restrict__ g_tile_ptr,
                                                                      const int32_t *__restrict__ g_mem_ptr, int32_t *__restrict__ off,
                                                                      int32_t *__restrict__ gid, float *__restrict__ loss_slot,
                                                                      float *__restrict__ loss_sum) {
    __shared__ int32_t s[4][kBatchMaxGraphs];
    const int i = threadIdx.x;
    const int step = *step_idx;
    int32_t c[4] = {0, 0, 0, 0};
    int32_t g = 0;
    if (i < B) {
        g = (int32_t)perm[(int64_t)step * B + i];
        c[0] = g_row_ptr[g + 1] - g_row_ptr[g];
        c[1] = g_nnz_ptr[g + 1] - g_nnz_ptr[g];
        c[2] = g_tile_ptr[g + 1] - g_tile_ptr[g];
        c[3] = g_mem_ptr[g + 1] - g_mem_ptr[g];
        gid[i] = g;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k][i] = c[k];
    __syncthreads();
    for (int d = 1; d < kBatchMaxGraphs; d <<= 1) {   // inclusive Hillis-Steele scans, the four arrays together
        int32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = i >= d ? s[k][i - d] : 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k][i] += v[k];
        __syncthreads();
    }
    if (i < B) {
#pragma unroll
        for (int k = 0; k < 4; ++k) off[k * (B + 1) + i] = s[k][i] - c[k];
        if (i == B - 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) off[k * (B + 1) + B] = s[k][i];
        }
    }
    if (i == 0) {
        *step_idx = step + 1;
        if (loss_slot && loss_sum) {
            *loss_sum += *loss_slot;
            *loss_slot = 0.f;
        }
    }
}

// the graph i of the batch whose range [o[i], o[i + 1]) holds idx (o ascending, o[B] > idx)
__device__ __forceinline__ int batch_find(const int32_t *__restrict__ o, int B, int32_t idx) {
    int lo = 0, hi = B;   // invariant: o[lo] <= idx < o[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (o[mid] <= idx) lo = mid; else hi = mid;
    }
    return lo;
}

struct BatchGlobal {   // the dataset's arrays (device)
    const int32_t *g_row_ptr, *g_nnz_ptr, *g_tile_ptr, *g_mem_ptr;
    const int32_t *rowptr, *col;
    const float *val;
    const fitgnn_tile_t *tiles;
    const int32_t *mem;         // pooled rows (global row ids), grouped by graph
    const uint8_t *pooled;      // per row: 1 = the row is pooled
    const int32_t *mem_rank;    // per row: its rank among the pooled rows of its graph (pooled rows only)
    const float *ax;            // [rows x K] (row stride ld_ax): the first layer's aggregated input
    const float *tgt;           // [graphs x n_tgt]
};
struct BatchOut {      // the batch's fixed-capacity buffers (device)
    int32_t *rowptr, *col;
    float *val;
    fitgnn_tile_t *tiles;
    int32_t *members, *seg_off, *seg_of_row;
    float *inv_cnt, *ax, *tgt;
    int64_t *members64;         // the pooled rows once more, as the int64 row index the layers take
    int32_t *cseg;              // graph of compact (pooled) row i, -1 past the batch's pooled rows
    int32_t *pos;               // per batch row: its compact position, or M_cap + r % zero_rows (a zero row of a compact operand)
};

__global__ __launch_bounds__(256) void batch_gather_kernel(int32_t B, const int32_t *__restrict__ off, const int32_t *__restrict__ gid,
                                                           BatchGlobal G, BatchOut O, int32_t R_cap, int32_t E_cap, int32_t T_cap,
                                                           int32_t M_cap, int32_t K, int32_t ld_ax_g, int32_t ld_ax, int32_t n_tgt,
                                                           int32_t zero_rows) {
    const int32_t idx = (int32_t)(blockIdx.x * 256 + threadIdx.x);
    const int32_t *o_row = off, *o_nnz = off + (B + 1), *o_tile = off + 2 * (B + 1), *o_mem = off + 3 * (B + 1);
    const int32_t n_row = o_row[B], n_nnz = o_nnz[B], n_tile = o_tile[B], n_mem = o_mem[B];
    if (idx <= R_cap) {   // rows: row pointer, pooled-row marker, the aggregated input
        if (idx < n_row) {
            const int i = batch_find(o_row, B, idx);
            const int32_t g = gid[i];
            const int32_t src = G.g_row_ptr[g] + (idx - o_row[i]);
            O.rowptr[idx] = o_nnz[i] + (G.rowptr[src] - G.g_nnz_ptr[g]);
            const bool pl = G.pooled[src] != 0;
            O.seg_of_row[idx] = pl ? i : -1;
            if (O.pos) O.pos[idx] = pl ? o_mem[i] + G.mem_rank[src] : M_cap + idx % zero_rows;
            for (int k = 0; k < K; ++k) O.ax[(int64_t)idx * ld_ax + k] = G.ax[(int64_t)src * ld_ax_g + k];
        } else {
            O.rowptr[idx] = n_nnz;   // rows past the batch: no entries
            if (idx < R_cap) {
                O.seg_of_row[idx] = -1;
                if (O.pos) O.pos[idx] = M_cap + idx % zero_rows;
                for (int k = 0; k < K; ++k) O.ax[(int64_t)idx * ld_ax + k] = 0.f;
            }
        }
    }
    if (idx < E_cap) {    // CSR entries: columns re-based to the batch's rows
        if (idx < n_nnz) {
            const int i = batch_find(o_nnz, B, idx);
            const int32_t g = gid[i];
            const int32_t src = G.g_nnz_ptr[g] + (idx - o_nnz[i]);
            O.col[idx] = G.col[src] - G.g_row_ptr[g] + o_row[i];
            O.val[idx] = G.val[src];
        } else {
            O.col[idx] = 0;
            O.val[idx] = 0.f;
        }
    }
    if (idx < T_cap) {    // row tiles (contiguous windows: a tile never spans two graphs)
        fitgnn_tile_t t;
        t.row_begin = t.row_end = t.win_begin = t.win_rows = t.nnz_begin = t.nnz_end = 0;
        t.reserved[0] = t.reserved[1] = 0;
        if (idx < n_tile) {
            const int i = batch_find(o_tile, B, idx);
            const int32_t g = gid[i];
            t = G.tiles[G.g_tile_ptr[g] + (idx - o_tile[i])];
            const int32_t dr = o_row[i] - G.g_row_ptr[g], dn = o_nnz[i] - G.g_nnz_ptr[g];
            t.row_begin += dr; t.row_end += dr; t.win_begin += dr;
            t.nnz_begin += dn; t.nnz_end += dn;
        } else {
            // the rows past the batch are covered too, 16 to a tile: they hold no entries, and a product over the batch then WRITES
            // them (zeros through the store's epilogue) -- left unwritten they would be whatever the buffer held, and a NaN there
            // survives the backward's 0 * ELU'
            const int32_t r0 = n_row + (idx - n_tile) * 16;
            if (r0 < R_cap) {
                t.row_begin = r0; t.row_end = min(r0 + 16, R_cap);
                t.win_begin = r0; t.win_rows = t.row_end - r0;
                t.nnz_begin = t.nnz_end = n_nnz;
            }
        }
        O.tiles[idx] = t;
    }
    if (idx < M_cap) {    // pooled rows, grouped by graph
        int32_t m = 0, sg = -1;
        if (idx < n_mem) {
            const int i = batch_find(o_mem, B, idx);
            const int32_t g = gid[i];
            m = G.mem[G.g_mem_ptr[g] + (idx - o_mem[i])] - G.g_row_ptr[g] + o_row[i];
            sg = i;
        }
        O.members[idx] = m;
        if (O.members64) O.members64[idx] = m;
        if (O.cseg) O.cseg[idx] = sg;
    }
    if (idx <= B) {       // per graph: segment offsets, 1 / count, targets
        O.seg_off[idx] = o_mem[idx];
        if (idx < B) {
            const int32_t cnt = o_mem[idx + 1] - o_mem[idx];
            O.inv_cnt[idx] = 1.0f / (float)(cnt > 0 ? cnt : 1);
            const int32_t g = gid[idx];
            for (int j = 0; j < n_tgt; ++j) O.tgt[(int64_t)idx * n_tgt + j] = G.tgt[(int64_t)g * n_tgt + j];
        }
    }
}

}  // namespace

extern "C" int fitgnn_batch_offsets(const int64_t *perm, int32_t *step_idx, int32_t B, const int32_t *g_row_ptr, const int32_t *g_nnz_ptr,
                                    const int32_t *g_tile_ptr, const int32_t *g_mem_ptr, int32_t *off, int32_t *gid, float *loss_slot,
                                    float *loss_sum, void *stream) {
    if (B < 1 || B > kBatchMaxGraphs) return FITGNN_E_BADARG;
    if (!perm || !step_idx || !g_row_ptr || !g_nnz_ptr || !g_tile_ptr || !g_mem_ptr || !off || !gid) return FITGNN_E_BADARG;
    hipLaunchKernelGGL(batch_offsets_kernel, dim3(1), dim3(kBatchMaxGraphs), 0, (hipStream_t)stream, perm, step_idx, B, g_row_ptr, g_nnz_ptr,
                       g_tile_ptr, g_mem_ptr, off, gid, loss_slot, loss_sum);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_batch_gather(int32_t B, const int32_t *off, const int32_t *gid, const int32_t *g_row_ptr, const int32_t *g_nnz_ptr,
                                   const int32_t *g_tile_ptr, const int32_t *g_mem_ptr, const int32_t *rowptr, const int32_t *col,
                                   const float *val, const fitgnn_tile_t *tiles, const int32_t *mem, const uint8_t *pooled, const float *ax,
                                   int32_t ld_ax_g, const float *tgt, int32_t n_tgt, int32_t K, int32_t R_cap, int32_t E_cap, int32_t T_cap,
                                   int32_t M_cap, int32_t *b_rowptr, int32_t *b_col, float *b_val, fitgnn_tile_t *b_tiles,
                                   int32_t *b_members, int32_t *b_seg_off, int32_t *b_seg_of_row, float *b_inv_cnt, float *b_ax,
                                   int32_t ld_ax, float *b_tgt, const int32_t *mem_rank, int64_t *b_members64, int32_t *b_cseg,
                                   int32_t *b_pos, int32_t zero_rows, void *stream) {
    if (B < 1 || B > kBatchMaxGraphs || R_cap < 1 || E_cap < 1 || T_cap < 1 || M_cap < 1 || K < 1 || n_tgt < 1 || ld_ax_g < K || ld_ax < K)
        return FITGNN_E_BADARG;
    if (!off || !gid || !g_row_ptr || !g_nnz_ptr || !g_tile_ptr || !g_mem_ptr || !rowptr || !col || !val || !tiles || !mem || !pooled || !ax ||
        !tgt || !b_rowptr || !b_col || !b_val || !b_tiles || !b_members || !b_seg_off || !b_seg_of_row || !b_inv_cnt || !b_ax || !b_tgt)
        return FITGNN_E_BADARG;
    if (b_pos && (!mem_rank || zero_rows < 1)) return FITGNN_E_BADARG;
    const BatchGlobal G{g_row_ptr, g_nnz_ptr, g_tile_ptr, g_mem_ptr, rowptr, col, val, tiles, mem, pooled, mem_rank, ax, tgt};
    const BatchOut O{b_rowptr, b_col, b_val, b_tiles, b_members, b_seg_off, b_seg_of_row, b_inv_cnt, b_ax, b_tgt, b_members64, b_cseg, b_pos};
    const int64_t items = std::max<int64_t>(std::max<int64_t>((int64_t)R_cap + 1, E_cap), std::max<int64_t>(std::max<int64_t>(T_cap, M_cap), B + 1));
    hipLaunchKernelGGL(batch_gather_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, off, gid, G, O, R_cap,
                       E_cap, T_cap, M_cap, K, ld_ax_g, ld_ax, n_tgt, zero_rows > 0 ? zero_rows : 1);
    return (int)hipGetLastError();
}
