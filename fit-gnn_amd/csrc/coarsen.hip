// coarsen.hip -- one contraction level of FIT-GNN's variation_neighborhoods coarsening on gfx950.
//
// Replaces (graph_coarsening/coarsening_utils.py): the candidate family :571-578, subgraph_cost :555-561,
// the SortedList-driven greedy selection :604-650, get_coarsening_matrix :212-254 and the level mapping
// :168-179; plus C <- iC.C (:136).  f64 arithmetic is the canonical order of DESIGN.md; the file is
// compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"
#include "fitgnn_hip.h"
#include "scan.h"
#include "variation_cost.h"

namespace {

using fitgnn::CostGraph;
using fitgnn::CostLds;

constexpr size_t kAlign = 256;
inline size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

// ------------------------------------------------------------------------------------------------
// candidate family: set i = sorted(N(i) U {i})
// ------------------------------------------------------------------------------------------------
__global__ void family_count_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, int32_t N,
                                    int32_t *__restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    const int p = fitgnn::lower_bound_i32(col + e0, e1 - e0, i);
    const bool has_self = (e0 + p < e1) && col[e0 + p] == i;
    cnt[i] = (e1 - e0) + (has_self ? 0 : 1);
}

__global__ void family_fill_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, int32_t N,
                                   const int32_t *__restrict__ set_off, int32_t *__restrict__ set_mem) {
    // one wave per node: coalesced copy of the row with i merged at its sorted position
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i >= N) return;
    const int e0 = rowptr[i], e1 = rowptr[i + 1], deg = e1 - e0;
    const int p = fitgnn::lower_bound_i32(col + e0, deg, i);
    const bool has_self = (p < deg) && col[e0 + p] == i;
    int32_t *dst = set_mem + set_off[i];
    if (has_self) {
        for (int t = lane; t < deg; t += 64) dst[t] = col[e0 + t];
    } else {
        for (int t = lane; t < deg; t += 64) dst[t + (t >= p ? 1 : 0)] = col[e0 + t];
        if (lane == 0) dst[p] = i;
    }
}

// ------------------------------------------------------------------------------------------------
// costs of many sets: one wave per set, waves stride over the sets
// ------------------------------------------------------------------------------------------------
constexpr int kCostWaves = 4;

__global__ __launch_bounds__(kCostWaves * 64) void variation_costs_kernel(CostGraph g, const int32_t *__restrict__ set_off,
                                                                         const int32_t *__restrict__ set_len,
                                                                         const int32_t *__restrict__ set_mem,
                                                                         int32_t n_sets, double *__restrict__ cost) {
    __shared__ CostLds lds[kCostWaves];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int n_waves = gridDim.x * kCostWaves;
    for (int s = blockIdx.x * kCostWaves + wave; s < n_sets; s += n_waves) {
        const int off = __builtin_amdgcn_readfirstlane(set_off[s]);
        const int nc = __builtin_amdgcn_readfirstlane(set_len[s]);
        const double c = fitgnn::set_cost_wave(g, set_mem + off, nc, lds[wave]);
        if (lane == 0) cost[s] = c;
    }
}

// ------------------------------------------------------------------------------------------------
// greedy selection: a single wavefront walks the candidates in (cost, insertion) order
// ------------------------------------------------------------------------------------------------
struct HeapItem {
    double cost;
    int64_t seq;
    int32_t cand;
    int32_t pad;
};
__device__ __forceinline__ bool item_less(const HeapItem &a, const HeapItem &b) {
    if (a.cost < b.cost) return true;
    if (b.cost < a.cost) return false;
    return a.seq < b.seq;
}

constexpr int kHeapLds = 2048;      // single graph: top of the re-insertion heap lives in LDS (48 KiB), the rest in global
constexpr int kHeapLdsBatch = 128;  // batched small components: 3 KiB per wave, 8 waves per CU

template <int HEAP_LDS>
struct Heap {
    HeapItem *lds;
    HeapItem *glob;
    __device__ __forceinline__ HeapItem get(int i) const { return i < HEAP_LDS ? lds[i] : glob[i]; }
    __device__ __forceinline__ void put(int i, const HeapItem &v) { if (i < HEAP_LDS) lds[i] = v; else glob[i] = v; }
};

template <class H>
__device__ inline void heap_push(H &h, int &n, HeapItem it) {
    int i = n++;
    while (i > 0) {
        const int p = (i - 1) >> 1;
        const HeapItem pv = h.get(p);
        if (!item_less(it, pv)) break;
        h.put(i, pv);
        i = p;
    }
    h.put(i, it);
}
template <class H>
__device__ inline HeapItem heap_pop(H &h, int &n) {
    const HeapItem top = h.get(0);
    const HeapItem last = h.get(--n);
    int i = 0;
    for (;;) {
        const int l = 2 * i + 1, r = l + 1;
        if (l >= n) break;
        HeapItem cv = h.get(l);
        int c = l;
        if (r < n) {
            const HeapItem rv = h.get(r);
            if (item_less(rv, cv)) { cv = rv; c = r; }
        }
        if (!item_less(cv, last)) break;
        h.put(i, cv);
        i = c;
    }
    if (n > 0) h.put(i, last);
    return top;
}

// The greedy selection of contract_variation_linear (:604-650) over ONE connected component, run by one wavefront.
// The component's candidates are order[head0 .. head1) (ascending (cost, node id)); node ids, set_off/mem/len/marked
// are those of the whole (possibly block-diagonal) graph.  Selected sets go to sel_mem[0 .. ) and their END
// positions to sel_end[0 .. ) (both relative to the pointers passed in).  Returns through ns / pos / the residual
// n_reduce.  Every iteration consumes one queue entry and re-insertions strictly shrink a set, so the loop is
// bounded by max_iters = candidates + members + slack (an exit every lane reaches).
#ifdef FITGNN_GREEDY_STAMPS
__device__ unsigned long long g_greedy_dbg[8];  // cycles: pop, mark check, select, prune, recost, heap push; counts: pops, recosts
#define FITGNN_STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#define FITGNN_ACC(i, a, b) if ((threadIdx.x & 63) == 0) g_greedy_dbg[i] += (b) - (a)
#else
#define FITGNN_STAMP(var)
#define FITGNN_ACC(i, a, b)
#endif

template <int HEAP_LDS>
__device__ inline void greedy_component(const CostGraph &g, CostLds &lds, Heap<HEAP_LDS> heap, int32_t head0, int32_t head1,
                                        int64_t seq, const int32_t *__restrict__ set_off, int32_t *__restrict__ mem,
                                        int32_t *__restrict__ len, uint8_t *__restrict__ marked,
                                        const int32_t *__restrict__ order, const double *__restrict__ cost0,
                                        int64_t &n_reduce, int64_t max_iters, int32_t *__restrict__ sel_end,
                                        int32_t *__restrict__ sel_mem, int32_t &ns, int32_t &pos) {
    const int lane = threadIdx.x & 63;
    int hn = 0;          // heap size (uniform)
    int head = head0;    // next unread entry of the sorted initial family (uniform)
    ns = 0; pos = 0;
    // the head of the sorted list is fetched one pop ahead (candidate, cost, set extent): the loads of the NEXT list
    // entry are in flight while the current candidate is processed.  len[] of an unprocessed list entry is its initial
    // length (only the candidate being processed is ever shrunk).
    int32_t nx_cand = 0, nx_off = 0, nx_len = 0;
    double nx_cost = 0.0;
    if (head < head1) { nx_cand = order[head]; nx_cost = cost0[nx_cand]; nx_off = set_off[nx_cand]; nx_len = len[nx_cand]; }
    for (int64_t it = 0; it < max_iters; ++it) {
        if (n_reduce <= 0) break;
        if (head >= head1 && hn == 0) break;
        // ---- pop the minimum of {sorted initial list head, heap top}: SortedList.pop(0) ----
        FITGNN_STAMP(t_a);
        int32_t cand;
        int off, nc;
        bool from_list = hn == 0;
        if (head < head1 && hn > 0) {
            HeapItem li{nx_cost, (int64_t)nx_cand, nx_cand, 0};
            const HeapItem top = heap.get(0);
            from_list = item_less(li, top);
        }
        if (from_list) {
            cand = __builtin_amdgcn_readfirstlane(nx_cand);
            off = __builtin_amdgcn_readfirstlane(nx_off);
            nc = __builtin_amdgcn_readfirstlane(nx_len);
            ++head;
            if (head < head1) { nx_cand = order[head]; nx_cost = cost0[nx_cand]; nx_off = set_off[nx_cand]; nx_len = len[nx_cand]; }
        } else {
            HeapItem top;
            if (lane == 0) top = heap_pop(heap, hn); else --hn;
            FITGNN_WAVE_SYNC();
            cand = __builtin_amdgcn_readfirstlane(__shfl(top.cand, 0, 64));
            off = __builtin_amdgcn_readfirstlane(set_off[cand]);
            nc = __builtin_amdgcn_readfirstlane(len[cand]);
        }
        int32_t *S = mem + off;
        FITGNN_STAMP(t_b);
        FITGNN_ACC(0, t_a, t_b);
        FITGNN_ACC(6, 0ull, 1ull);
        // ---- any member marked? (coarsening_utils.py:620-622) ----
        bool any = false;
        for (int t0 = 0; t0 < nc; t0 += 64) {
            const int t = t0 + lane;
            const bool mk = (t < nc) && marked[S[t]] != 0;
            any |= __ballot(mk) != 0ull;
        }
        FITGNN_STAMP(t_c);
        FITGNN_ACC(1, t_b, t_c);
        if (!any) {
            const int64_t gain = nc - 1;
            if (gain > n_reduce) continue;  // :625-626, would over-reduce: drop the set
            for (int t = lane; t < nc; t += 64) {
                const int32_t v = S[t];
                marked[v] = 1;
                sel_mem[pos + t] = v;
            }
            pos += nc;
            if (lane == 0) sel_end[ns] = pos;
            ++ns;
            n_reduce -= gain;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // marked[] writes before later reads
            FITGNN_STAMP(t_d);
            FITGNN_ACC(2, t_c, t_d);
        } else {
            // ---- drop marked members in place, keep order (:640) ----
            int m = 0;
            for (int t0 = 0; t0 < nc; t0 += 64) {
                const int t = t0 + lane;
                int32_t v = 0;
                bool keep = false;
                if (t < nc) { v = S[t]; keep = marked[v] == 0; }
                const unsigned long long bal = __ballot(keep);
                const int before = __popcll(bal & ((1ull << lane) - 1ull));
                FITGNN_WAVE_SYNC();  // all reads of this chunk done before the compacted writes (m <= t0)
                if (keep) S[m + before] = v;
                m += __popcll(bal);
            }
            FITGNN_STAMP(t_e);
            FITGNN_ACC(3, t_c, t_e);
            if (m > 1) {
                if (lane == 0) len[cand] = m;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                const double c = fitgnn::set_cost_wave(g, S, m, lds);  // :646 re-cost
                FITGNN_STAMP(t_f);
                FITGNN_ACC(4, t_e, t_f);
                FITGNN_ACC(7, 0ull, 1ull);
                if (lane == 0) heap_push(heap, hn, HeapItem{c, seq, cand, 0}); else ++hn;
                ++seq;
                FITGNN_WAVE_SYNC();
                FITGNN_STAMP(t_g);
                FITGNN_ACC(5, t_f, t_g);
            }
        }
    }
}

__global__ __launch_bounds__(64) void greedy_select_kernel(CostGraph g, int32_t N, const int32_t *__restrict__ set_off,
                                                           int32_t *__restrict__ mem, int32_t *__restrict__ len,
                                                           uint8_t *__restrict__ marked,
                                                           const int32_t *__restrict__ order,
                                                           const double *__restrict__ cost0, HeapItem *__restrict__ heap_glob,
                                                           int64_t n_reduce, int64_t max_iters,
                                                           int32_t *__restrict__ sel_off, int32_t *__restrict__ sel_mem,
                                                           int32_t *__restrict__ sel_count) {
    __shared__ CostLds lds;
    __shared__ HeapItem heap_lds[kHeapLds];
    if ((threadIdx.x & 63) == 0) sel_off[0] = 0;
    int32_t ns, pos;
    // re-inserted sets get seq = N, N+1, ... (initial family: seq = node id)
    greedy_component<kHeapLds>(g, lds, Heap<kHeapLds>{heap_lds, heap_glob}, 0, N, (int64_t)N, set_off, mem, len, marked, order,
                               cost0, n_reduce, max_iters, sel_off + 1, sel_mem, ns, pos);
    if ((threadIdx.x & 63) == 0) { sel_count[0] = ns; sel_count[1] = pos; }
}

// One wavefront (= one 64-thread workgroup) per connected component of a block-diagonal graph whose components are
// contiguous node ranges [comp_off[c], comp_off[c+1]).  Stages each component's sets in its own node range of
// stage_mem / stage_end; comp_stat[c] = {number of sets, number of members}; gain[c] = sum of (|set| - 1).
__global__ __launch_bounds__(64) void greedy_select_batch_kernel(CostGraph g, int32_t N, int32_t n_comp,
                                                                 const int32_t *__restrict__ comp_off,
                                                                 const int32_t *__restrict__ set_off, int32_t *__restrict__ mem,
                                                                 int32_t *__restrict__ len, uint8_t *__restrict__ marked,
                                                                 const int32_t *__restrict__ order,
                                                                 const double *__restrict__ cost0,
                                                                 HeapItem *__restrict__ heap_glob,
                                                                 const int64_t *__restrict__ n_reduce_in,
                                                                 int32_t *__restrict__ stage_end, int32_t *__restrict__ stage_mem,
                                                                 int32_t *__restrict__ cnt_sets, int32_t *__restrict__ cnt_mem,
                                                                 int64_t *__restrict__ gain) {
    __shared__ CostLds lds;
    __shared__ HeapItem heap_lds[kHeapLdsBatch];
    const int c = blockIdx.x;
    if (c >= n_comp) return;
    const int32_t b = comp_off[c], e = comp_off[c + 1];
    int64_t n_reduce = n_reduce_in[c];
    const int64_t budget = n_reduce;
    const int64_t max_iters = (int64_t)(e - b) + (int64_t)(set_off[e] - set_off[b]) + 8;
    int32_t ns = 0, pos = 0;
    if (e > b && n_reduce > 0)
        greedy_component<kHeapLdsBatch>(g, lds, Heap<kHeapLdsBatch>{heap_lds, heap_glob + b}, b, e, (int64_t)N, set_off, mem, len,
                                        marked, order, cost0, n_reduce, max_iters, stage_end + b, stage_mem + b, ns, pos);
    if ((threadIdx.x & 63) == 0) { cnt_sets[c] = ns; cnt_mem[c] = pos; gain[c] = budget - n_reduce; }
}

// keep[c] = gain[c] > min_gain: components whose whole level would remove <= min_gain nodes are left untouched
// (coarsening_utils.py:131-135 breaks before applying such a level)
__global__ void batch_keep_kernel(int32_t n_comp, const int64_t *__restrict__ gain, int64_t min_gain,
                                  int32_t *__restrict__ cnt_sets, int32_t *__restrict__ cnt_mem) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_comp) return;
    if (gain[c] <= min_gain) { cnt_sets[c] = 0; cnt_mem[c] = 0; }
}
// scatter the staged per-component lists into one (sel_off, sel_mem) list; set_base / mem_base = exclusive scans
__global__ __launch_bounds__(64) void batch_compact_kernel(int32_t n_comp, const int32_t *__restrict__ comp_off,
                                                           const int32_t *__restrict__ cnt_sets, const int32_t *__restrict__ cnt_mem,
                                                           const int32_t *__restrict__ set_base, const int32_t *__restrict__ mem_base,
                                                           const int32_t *__restrict__ stage_end, const int32_t *__restrict__ stage_mem,
                                                           int32_t *__restrict__ sel_off, int32_t *__restrict__ sel_mem,
                                                           int32_t *__restrict__ sel_count) {
    const int c = blockIdx.x;
    if (c >= n_comp) return;
    const int b = comp_off[c], ns = cnt_sets[c], nm = cnt_mem[c], sb = set_base[c], mb = mem_base[c];
    for (int k = threadIdx.x; k < ns; k += 64) sel_off[sb + k + 1] = mb + stage_end[b + k];
    for (int t = threadIdx.x; t < nm; t += 64) sel_mem[mb + t] = stage_mem[b + t];
    if (c == 0 && threadIdx.x == 0) {
        sel_off[0] = 0;
        sel_count[0] = set_base[n_comp];
        sel_count[1] = mem_base[n_comp];
    }
}
__global__ void comp_of_kernel(int32_t n_comp, const int32_t *__restrict__ comp_off, uint32_t *__restrict__ comp_of) {
    const int c = blockIdx.x;
    if (c >= n_comp) return;
    for (int i = comp_off[c] + threadIdx.x; i < comp_off[c + 1]; i += blockDim.x) comp_of[i] = (uint32_t)c;
}
__global__ void gather_u32_kernel(const uint32_t *__restrict__ src, const int32_t *__restrict__ idx, int32_t n,
                                  uint32_t *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

__global__ void cost_keys_kernel(const double *__restrict__ cost0, int32_t N, uint64_t *__restrict__ keys,
                                 int32_t *__restrict__ ids) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    // costs are >= +0 or +inf (NaN sorts last): their bit patterns order like the values
    uint64_t b = (uint64_t)__double_as_longlong(cost0[i]);
    if (b >> 63) b = 0;  // -0.0 / negative noise cannot occur (sqrt >= 0); clamp defensively
    keys[i] = b;
    ids[i] = i;
}

__global__ void len_init_kernel(const int32_t *__restrict__ set_off, int32_t N, int32_t *__restrict__ len) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) len[i] = set_off[i + 1] - set_off[i];
}

struct GreedyLayout {
    size_t mem, len, marked, keys_in, keys_out, ids_in, order, heap, sort_tmp, sort_tmp_bytes, total;
};
GreedyLayout greedy_layout(int32_t N, int64_t total_members) {
    GreedyLayout L{};
    size_t o = 0;
    const size_t n = (size_t)(N > 0 ? N : 1), tm = (size_t)(total_members > 0 ? total_members : 1);
    L.mem = o; o += align_up(tm * 4);
    L.len = o; o += align_up(n * 4);
    L.marked = o; o += align_up(n);
    L.keys_in = o; o += align_up(n * 8);
    L.keys_out = o; o += align_up(n * 8);
    L.ids_in = o; o += align_up(n * 4);
    L.order = o; o += align_up(n * 4);
    L.heap = o; o += align_up(n * sizeof(HeapItem));
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, (uint64_t *)nullptr, (uint64_t *)nullptr, (int32_t *)nullptr,
                                    (int32_t *)nullptr, n, 0, 64, (hipStream_t)0);
    L.sort_tmp_bytes = tmp;
    L.sort_tmp = o; o += align_up(tmp);
    L.total = o;
    return L;
}

// ------------------------------------------------------------------------------------------------
// assignment vectors from the selected sets
// ------------------------------------------------------------------------------------------------
__global__ void assign_init_kernel(int32_t N, int32_t *__restrict__ root, double *__restrict__ cval) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) { root[i] = i; cval[i] = 1.0; }
}
__global__ void assign_sets_kernel(const int32_t *__restrict__ sel_off, const int32_t *__restrict__ sel_mem,
                                   const int32_t *__restrict__ sel_count, int32_t *__restrict__ root,
                                   double *__restrict__ cval) {
#pragma clang fp contract(off)
    // one wave per selected set
    const int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (s >= sel_count[0]) return;
    const int o = sel_off[s], nc = sel_off[s + 1] - o;
    const int32_t r = sel_mem[o];  // sets are sorted: the minimum member keeps the row (:239)
    const double v = 1.0 / sqrt((double)nc);
    for (int t = lane; t < nc; t += 64) { root[sel_mem[o + t]] = r; cval[sel_mem[o + t]] = v; }
}
__global__ void survivor_flag_kernel(int32_t N, const int32_t *__restrict__ root, int32_t *__restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) flag[i] = root[i] == i ? 1 : 0;
}
__global__ void assign_final_kernel(int32_t N, const int32_t *__restrict__ root, const int32_t *__restrict__ rank,
                                    int32_t *__restrict__ assign, int32_t *__restrict__ n_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    assign[i] = rank[root[i]];
    if (i == N - 1) n_out[0] = rank[i] + (root[i] == i ? 1 : 0);
}

__global__ void compose_levels_kernel(int32_t N0, const int32_t *__restrict__ assign_l, const double *__restrict__ cval_l,
                                      int32_t *__restrict__ assign_tot, double *__restrict__ cval_tot) {
#pragma clang fp contract(off)
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N0) return;
    const int32_t prev = assign_tot[j];
    cval_tot[j] = cval_l[prev] * cval_tot[j];
    assign_tot[j] = assign_l[prev];
}

inline dim3 blocks_for(int64_t threads, int block = 256) { return dim3((unsigned)((threads + block - 1) / block)); }

}  // namespace

extern "C" int fitgnn_closed_neighbourhoods(const int32_t *rowptr, const int32_t *col, int32_t N, int32_t *set_off,
                                            int32_t *set_mem, void *stream) {
    if (N < 0 || !set_off) return FITGNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) return (int)hipMemsetAsync(set_off, 0, sizeof(int32_t), s);
    if (!rowptr || !col || !set_mem) return FITGNN_E_BADARG;
    hipLaunchKernelGGL(family_count_kernel, blocks_for(N), dim3(256), 0, s, rowptr, col, N, set_off);
    fitgnn::exclusive_scan_i32(set_off, set_off, N, s);  // counts -> offsets, in place
    hipLaunchKernelGGL(family_fill_kernel, blocks_for((int64_t)N * 64), dim3(256), 0, s, rowptr, col, N, set_off, set_mem);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_variation_costs_f64(const int32_t *rowptr, const int32_t *col, const double *w, const double *dw,
                                          const double *A, int32_t K, int64_t lda, const int32_t *set_off,
                                          const int32_t *set_len, const int32_t *set_mem, int32_t n_sets, double *cost,
                                          void *stream) {
    if (n_sets < 0 || K < 1 || K > FITGNN_MAX_K || lda < K) return FITGNN_E_BADARG;
    if (n_sets == 0) return 0;
    if (!rowptr || !col || !dw || !A || !set_off || !set_len || !set_mem || !cost) return FITGNN_E_BADARG;
    CostGraph g{rowptr, col, w, dw, A, K, lda, nullptr};
    const int blocks = (int)std::min<int64_t>(((int64_t)n_sets + kCostWaves - 1) / kCostWaves, 256 * 8);
    hipLaunchKernelGGL(variation_costs_kernel, dim3(blocks), dim3(kCostWaves * 64), 0, (hipStream_t)stream, g, set_off,
                       set_len, set_mem, n_sets, cost);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_variation_costs_batch_f64(const int32_t *rowptr, const int32_t *col, const double *w, const double *dw,
                                                const double *A, int32_t K, int64_t lda, const int32_t *node_K,
                                                const int32_t *set_off, const int32_t *set_len, const int32_t *set_mem,
                                                int32_t n_sets, double *cost, void *stream) {
    if (n_sets < 0 || K < 1 || K > FITGNN_MAX_K || lda < K) return FITGNN_E_BADARG;
    if (n_sets == 0) return 0;
    if (!rowptr || !col || !dw || !A || !set_off || !set_len || !set_mem || !cost) return FITGNN_E_BADARG;
    CostGraph g{rowptr, col, w, dw, A, K, lda, node_K};
    const int blocks = (int)std::min<int64_t>(((int64_t)n_sets + kCostWaves - 1) / kCostWaves, 256 * 8);
    hipLaunchKernelGGL(variation_costs_kernel, dim3(blocks), dim3(kCostWaves * 64), 0, (hipStream_t)stream, g, set_off,
                       set_len, set_mem, n_sets, cost);
    return (int)hipGetLastError();
}

#ifdef FITGNN_GREEDY_STAMPS
extern "C" int fitgnn_debug_greedy_counters(unsigned long long *out, int reset) {
    hipDeviceSynchronize();
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_greedy_dbg), sizeof(unsigned long long) * 8);
    rc |= (int)hipMemcpyFromSymbol(out + 8, HIP_SYMBOL(g_cost_dbg), sizeof(unsigned long long) * 8);
    if (reset) { unsigned long long z2[8] = {0}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cost_dbg), z2, sizeof(z2)); }
    if (reset) { unsigned long long z[8] = {0}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_greedy_dbg), z, sizeof(z)); }
    return rc;
}
#endif

extern "C" size_t fitgnn_greedy_select_workspace_bytes(int32_t N, int64_t total_members) {
    if (N < 0 || total_members < 0) return 0;
    return greedy_layout(N, total_members).total;
}

extern "C" int fitgnn_greedy_select(const int32_t *rowptr, const int32_t *col, const double *w, const double *dw,
                                    const double *A, int32_t K, int64_t lda, int32_t N, const int32_t *set_off,
                                    const int32_t *set_mem, const double *cost0, int64_t n_reduce, int32_t *sel_off,
                                    int32_t *sel_mem, int32_t *sel_count, void *work, size_t work_bytes, void *stream) {
    if (N < 0 || K < 1 || K > FITGNN_MAX_K || lda < K) return FITGNN_E_BADARG;
    if (!sel_off || !sel_count) return FITGNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        FITGNN_RETURN_IF_HIP(hipMemsetAsync(sel_off, 0, sizeof(int32_t), s));
        return (int)hipMemsetAsync(sel_count, 0, 2 * sizeof(int32_t), s);
    }
    if (!rowptr || !col || !dw || !A || !set_off || !set_mem || !cost0 || !sel_mem || !work) return FITGNN_E_BADARG;
    // total members = set_off[N] lives on the device; the caller sized the workspace with it.  Recover the
    // capacity the workspace was sized for from work_bytes by requiring the caller's figure to be consistent:
    // we only need an upper bound, and set_off[N] <= nnz + N, so read it back once (4 bytes, needed for the
    // copy and the iteration bound).
    int32_t total = 0;
    FITGNN_RETURN_IF_HIP(hipMemcpyAsync(&total, set_off + N, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    FITGNN_RETURN_IF_HIP(hipStreamSynchronize(s));
    const GreedyLayout L = greedy_layout(N, total);
    if (work_bytes < L.total) return FITGNN_E_WORKSPACE;
    char *base = (char *)work;
    int32_t *mem = (int32_t *)(base + L.mem);
    int32_t *len = (int32_t *)(base + L.len);
    uint8_t *marked = (uint8_t *)(base + L.marked);
    uint64_t *keys_in = (uint64_t *)(base + L.keys_in), *keys_out = (uint64_t *)(base + L.keys_out);
    int32_t *ids_in = (int32_t *)(base + L.ids_in), *order = (int32_t *)(base + L.order);
    HeapItem *heap = (HeapItem *)(base + L.heap);
    FITGNN_RETURN_IF_HIP(hipMemcpyAsync(mem, set_mem, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    FITGNN_RETURN_IF_HIP(hipMemsetAsync(marked, 0, (size_t)N, s));
    hipLaunchKernelGGL(len_init_kernel, blocks_for(N), dim3(256), 0, s, set_off, N, len);
    hipLaunchKernelGGL(cost_keys_kernel, blocks_for(N), dim3(256), 0, s, cost0, N, keys_in, ids_in);
    // stable LSD radix sort on the cost bits: ties keep ascending node id == SortedList's stable build
    size_t tmp = L.sort_tmp_bytes;
    FITGNN_RETURN_IF_HIP(rocprim::radix_sort_pairs((void *)(base + L.sort_tmp), tmp, keys_in, keys_out, ids_in, order,
                                                   (size_t)N, 0, 64, s));
    CostGraph g{rowptr, col, w, dw, A, K, lda, nullptr};
    const int64_t max_iters = (int64_t)N + (int64_t)total + 8;
    hipLaunchKernelGGL(greedy_select_kernel, dim3(1), dim3(64), 0, s, g, N, set_off, mem, len, marked, order, cost0, heap,
                       n_reduce, max_iters, sel_off, sel_mem, sel_count);
    return (int)hipGetLastError();
}

struct BatchLayout {
    GreedyLayout G;
    size_t comp_of, ckey_in, ckey_out, order1, stage_end, stage_mem, cnt_sets, cnt_mem, set_base, mem_base, total;
};
BatchLayout batch_layout(int32_t N, int64_t total_members, int32_t n_comp) {
    BatchLayout L{};
    L.G = greedy_layout(N, total_members);
    size_t o = L.G.total;
    const size_t n = (size_t)(N > 0 ? N : 1), c = (size_t)(n_comp > 0 ? n_comp : 1);
    L.comp_of = o; o += align_up(n * 4);
    L.ckey_in = o; o += align_up(n * 4);
    L.ckey_out = o; o += align_up(n * 4);
    L.order1 = o; o += align_up(n * 4);
    L.stage_end = o; o += align_up(n * 4);
    L.stage_mem = o; o += align_up(n * 4);
    L.cnt_sets = o; o += align_up((c + 1) * 4);
    L.cnt_mem = o; o += align_up((c + 1) * 4);
    L.set_base = o; o += align_up((c + 1) * 4);
    L.mem_base = o; o += align_up((c + 1) * 4);
    L.total = o;
    return L;
}

extern "C" size_t fitgnn_greedy_select_batch_workspace_bytes(int32_t N, int64_t total_members, int32_t n_comp) {
    if (N < 0 || total_members < 0 || n_comp < 0) return 0;
    return batch_layout(N, total_members, n_comp).total;
}

extern "C" int fitgnn_greedy_select_batch(const int32_t *rowptr, const int32_t *col, const double *w, const double *dw,
                                          const double *A, int32_t K, int64_t lda, int32_t N, const int32_t *set_off,
                                          const int32_t *set_mem, const double *cost0, int32_t n_comp,
                                          const int32_t *comp_off, const int64_t *n_reduce, int64_t min_gain,
                                          const int32_t *node_K, int32_t *sel_off, int32_t *sel_mem, int32_t *sel_count,
                                          int64_t *comp_gain, void *work, size_t work_bytes, void *stream) {
    if (N < 0 || n_comp < 0 || K < 1 || K > FITGNN_MAX_K || lda < K) return FITGNN_E_BADARG;
    if (!sel_off || !sel_count) return FITGNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (N == 0 || n_comp == 0) {
        FITGNN_RETURN_IF_HIP(hipMemsetAsync(sel_off, 0, sizeof(int32_t), s));
        return (int)hipMemsetAsync(sel_count, 0, 2 * sizeof(int32_t), s);
    }
    if (!rowptr || !col || !dw || !A || !set_off || !set_mem || !cost0 || !sel_mem || !work || !comp_off || !n_reduce ||
        !comp_gain)
        return FITGNN_E_BADARG;
    int32_t total = 0;
    FITGNN_RETURN_IF_HIP(hipMemcpyAsync(&total, set_off + N, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    FITGNN_RETURN_IF_HIP(hipStreamSynchronize(s));
    const BatchLayout L = batch_layout(N, total, n_comp);
    if (work_bytes < L.total) return FITGNN_E_WORKSPACE;
    char *base = (char *)work;
    int32_t *mem = (int32_t *)(base + L.G.mem);
    int32_t *len = (int32_t *)(base + L.G.len);
    uint8_t *marked = (uint8_t *)(base + L.G.marked);
    uint64_t *keys_in = (uint64_t *)(base + L.G.keys_in), *keys_out = (uint64_t *)(base + L.G.keys_out);
    int32_t *ids_in = (int32_t *)(base + L.G.ids_in), *order = (int32_t *)(base + L.G.order);
    HeapItem *heap = (HeapItem *)(base + L.G.heap);
    uint32_t *comp_of = (uint32_t *)(base + L.comp_of), *ckey_in = (uint32_t *)(base + L.ckey_in),
             *ckey_out = (uint32_t *)(base + L.ckey_out);
    int32_t *order1 = (int32_t *)(base + L.order1), *stage_end = (int32_t *)(base + L.stage_end),
            *stage_mem = (int32_t *)(base + L.stage_mem), *cnt_sets = (int32_t *)(base + L.cnt_sets),
            *cnt_mem = (int32_t *)(base + L.cnt_mem), *set_base = (int32_t *)(base + L.set_base),
            *mem_base = (int32_t *)(base + L.mem_base);
    FITGNN_RETURN_IF_HIP(hipMemcpyAsync(mem, set_mem, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    FITGNN_RETURN_IF_HIP(hipMemsetAsync(marked, 0, (size_t)N, s));
    hipLaunchKernelGGL(len_init_kernel, blocks_for(N), dim3(256), 0, s, set_off, N, len);
    hipLaunchKernelGGL(cost_keys_kernel, blocks_for(N), dim3(256), 0, s, cost0, N, keys_in, ids_in);
    // candidates in (component, cost, node id) order: stable sort on the cost bits, then a stable sort on the
    // component id (the 64-bit sort's temporary storage is large enough for the 32-bit one)
    size_t tmp = L.G.sort_tmp_bytes;
    FITGNN_RETURN_IF_HIP(rocprim::radix_sort_pairs((void *)(base + L.G.sort_tmp), tmp, keys_in, keys_out, ids_in, order1,
                                                   (size_t)N, 0, 64, s));
    hipLaunchKernelGGL(comp_of_kernel, dim3(n_comp), dim3(64), 0, s, n_comp, comp_off, comp_of);
    hipLaunchKernelGGL(gather_u32_kernel, blocks_for(N), dim3(256), 0, s, comp_of, order1, N, ckey_in);
    int bits = 1;
    while ((1ll << bits) < (long long)n_comp) ++bits;
    size_t tmp2 = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp2, ckey_in, ckey_out, order1, order, (size_t)N, 0, bits, s);
    if (tmp2 > L.G.sort_tmp_bytes) return FITGNN_E_WORKSPACE;
    FITGNN_RETURN_IF_HIP(rocprim::radix_sort_pairs((void *)(base + L.G.sort_tmp), tmp2, ckey_in, ckey_out, order1, order, (size_t)N,
                                                   0, bits, s));
    CostGraph g{rowptr, col, w, dw, A, K, lda, node_K};
    hipLaunchKernelGGL(greedy_select_batch_kernel, dim3(n_comp), dim3(64), 0, s, g, N, n_comp, comp_off, set_off, mem, len, marked,
                       order, cost0, heap, n_reduce, stage_end, stage_mem, cnt_sets, cnt_mem, comp_gain);
    hipLaunchKernelGGL(batch_keep_kernel, blocks_for(n_comp), dim3(256), 0, s, n_comp, comp_gain, min_gain, cnt_sets, cnt_mem);
    fitgnn::exclusive_scan_i32(cnt_sets, set_base, n_comp, s);
    fitgnn::exclusive_scan_i32(cnt_mem, mem_base, n_comp, s);
    hipLaunchKernelGGL(batch_compact_kernel, dim3(n_comp), dim3(64), 0, s, n_comp, comp_off, cnt_sets, cnt_mem, set_base, mem_base,
                       stage_end, stage_mem, sel_off, sel_mem, sel_count);
    return (int)hipGetLastError();
}

extern "C" size_t fitgnn_build_assignment_workspace_bytes(int32_t N) {
    if (N <= 0) return kAlign;
    return 2 * align_up(((size_t)N + 1) * sizeof(int32_t));
}

extern "C" int fitgnn_build_assignment(int32_t N, const int32_t *sel_off, const int32_t *sel_mem,
                                       const int32_t *sel_count, int32_t *assign, double *cval, int32_t *n_out, void *work,
                                       size_t work_bytes, void *stream) {
    if (N < 0 || !n_out) return FITGNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) return (int)hipMemsetAsync(n_out, 0, sizeof(int32_t), s);
    if (!sel_off || !sel_mem || !sel_count || !assign || !cval || !work) return FITGNN_E_BADARG;
    if (work_bytes < fitgnn_build_assignment_workspace_bytes(N)) return FITGNN_E_WORKSPACE;
    int32_t *root = (int32_t *)work;
    int32_t *rank = (int32_t *)((char *)work + align_up(((size_t)N + 1) * sizeof(int32_t)));
    hipLaunchKernelGGL(assign_init_kernel, blocks_for(N), dim3(256), 0, s, N, root, cval);
    // at most N/2 sets of >= 2 members (plus singletons never listed): launch for the worst case, waves past
    // sel_count[0] exit at once
    hipLaunchKernelGGL(assign_sets_kernel, blocks_for((int64_t)N * 64), dim3(256), 0, s, sel_off, sel_mem, sel_count, root, cval);
    hipLaunchKernelGGL(survivor_flag_kernel, blocks_for(N), dim3(256), 0, s, N, root, rank);
    fitgnn::exclusive_scan_i32(rank, rank, N, s);
    hipLaunchKernelGGL(assign_final_kernel, blocks_for(N), dim3(256), 0, s, N, root, rank, assign, n_out);
    return (int)hipGetLastError();
}

extern "C" int fitgnn_compose_levels(int32_t N0, const int32_t *assign_l, const double *cval_l, int32_t *assign_tot,
                                     double *cval_tot, void *stream) {
    if (N0 < 0) return FITGNN_E_BADARG;
    if (N0 == 0) return 0;
    if (!assign_l || !cval_l || !assign_tot || !cval_tot) return FITGNN_E_BADARG;
    hipLaunchKernelGGL(compose_levels_kernel, blocks_for(N0), dim3(256), 0, (hipStream_t)stream, N0, assign_l, cval_l,
                       assign_tot, cval_tot);
    return (int)hipGetLastError();
}
