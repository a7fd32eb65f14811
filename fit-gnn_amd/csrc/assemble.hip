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
