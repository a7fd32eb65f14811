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
#ifdef FITGNN_GREEDY_STAMPS  // make EXTRA=-DFITGNN_GREEDY_STAMPS + tools/greedy_stamps.py: where the one wave's cycles go
static __device__ unsigned long long g_greedy_dbg[24];
// accumulated in registers and flushed once: a global read-modify-write per stamp would drain the prefetches it measures
#define GSTAMP_DECL unsigned long long gdbg[24] = {0}
#define GSTAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#define GACC(i, a, b) gdbg[i] += (b) - (a)
#define GCNT(i) gdbg[i] += 1
#define GSTAMP_FLUSH if ((threadIdx.x & 63) == 0) { for (int q = 0; q < 24; ++q) g_greedy_dbg[q] += gdbg[q]; }
#else
#define GSTAMP_DECL
#define GSTAMP(var)
#define GACC(i, a, b)
#define GCNT(i)
#define GSTAMP_FLUSH
#endif

struct HeapItem {
    double cost;
    int64_t seq;
    int32_t cand;
    int32_t off, len;  // the set's extent in mem[]: a pop needs no look-up before it can fetch the members
    int32_t pad;
};
__device__ __forceinline__ bool item_less(const HeapItem &a, const HeapItem &b) {
    if (a.cost < b.cost) return true;
    if (b.cost < a.cost) return false;
    return a.seq < b.seq;
}

constexpr int kHeapLds = 1024;      // single graph: the first slots of the re-insertion queue live in LDS (24 KiB), the rest in global
constexpr int kStateLdsBytes = 64 * 1024;  // ... and the marked-node bitmap + block minima beside them when they fit (N <= 209 000)
constexpr int kHeapLdsBatch = 128;  // batched small components: 4 KiB per wave

// ---- lane-0 binary heap (batched small components): top in LDS, the rest in global ----
template <int HEAP_LDS>
struct BinHeap {
    HeapItem *lds;
    HeapItem *glob;
    int n;  // uniform
    __device__ __forceinline__ HeapItem get(int i) const { return i < HEAP_LDS ? lds[i] : glob[i]; }
    __device__ __forceinline__ void put(int i, const HeapItem &v) { if (i < HEAP_LDS) lds[i] = v; else glob[i] = v; }
    __device__ __forceinline__ int size() const { return n; }
    __device__ inline void push(const HeapItem &it) {
        if ((threadIdx.x & 63) == 0) {
            int i = n;
            while (i > 0) {
                const int p = (i - 1) >> 1;
                const HeapItem pv = get(p);
                if (!item_less(it, pv)) break;
                put(i, pv);
                i = p;
            }
            put(i, it);
        }
        ++n;
        FITGNN_WAVE_SYNC();
    }
    __device__ inline HeapItem extract_min() {
        HeapItem top{};
        --n;
        if ((threadIdx.x & 63) == 0) {
            top = get(0);
            const HeapItem last = get(n);
            int i = 0;
            for (;;) {
                const int l = 2 * i + 1, r = l + 1;
                if (l >= n) break;
                HeapItem cv = get(l);
                int c = l;
                if (r < n) {
                    const HeapItem rv = get(r);
                    if (item_less(rv, cv)) { cv = rv; c = r; }
                }
                if (!item_less(cv, last)) break;
                put(i, cv);
                i = c;
            }
            if (n > 0) put(i, last);
        }
        FITGNN_WAVE_SYNC();
        top.cost = __shfl(top.cost, 0, 64);
        top.seq = __shfl(top.seq, 0, 64);
        top.cand = __shfl(top.cand, 0, 64);
        top.off = __shfl(top.off, 0, 64);
        top.len = __shfl(top.len, 0, 64);
        return top;
    }
};

// ---- wave-parallel tournament (single graph) ----
// A binary heap walked by one lane costs log2(n) DEPENDENT LDS round trips per operation (measured on S-pubmed, ~4 000
// live items: 6 200 cycles per pop, 2 800 per push -- a third of the whole selection).  Here the live items sit densely
// in slots [0, n), 64 slots form a block, and every block's minimum key is cached: a pop is (1) all lanes scan the block
// minima, (2) all lanes read the winning block, (3) the last slot fills the hole and the block minimum is rebuilt from the
// registers -- a fixed number of wave-wide steps whatever n is.  Keys order by (cost, seq) exactly like item_less: costs
// are >= +0 (a Frobenius norm over a positive count), so their bit patterns order like the values.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    // row_shr 1,2,4,8: lane 15 of every 16-lane row holds the row's minimum; row_bcast 15 / 31 fold the rows into lane 63
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// lane holding the smallest (cost, seq) among the lanes with valid set; -1 when there is none.  All 64 lanes call it.
__device__ __forceinline__ int wave_argmin(double cost, uint32_t seq, bool valid) {
    const uint64_t bits = (uint64_t)__double_as_longlong(cost);
    const uint32_t hi = (uint32_t)(bits >> 32), lo = (uint32_t)bits;
    bool in = valid;
    const uint32_t mh = wave_min_u32(in ? hi : 0xffffffffu);
    in = in && hi == mh;
    unsigned long long bal = __ballot(in);
    if (bal == 0ull) return -1;
    if (__popcll(bal) > 1) {  // wave-uniform; rare: the high words of two different costs seldom agree
        const uint32_t ml = wave_min_u32(in ? lo : 0xffffffffu);
        in = in && lo == ml;
        bal = __ballot(in);
        if (__popcll(bal) > 1) {
            const uint32_t ms = wave_min_u32(in ? seq : 0xffffffffu);
            in = in && seq == ms;
            bal = __ballot(in);
        }
    }
    return __builtin_ctzll(bal);
}
__device__ __forceinline__ int32_t lane_bcast(int32_t v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint32_t lane_bcast(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ double lane_bcast(double v, int l) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t h = (uint32_t)__builtin_amdgcn_readlane((int)(b >> 32), l), w = (uint32_t)__builtin_amdgcn_readlane((int)b, l);
    return __longlong_as_double((long long)(((uint64_t)h << 32) | w));
}

// LDS arrays are addressed through address-space-3 pointers: behind a plain pointer the compiler folds "slot in LDS ? a : b" into ONE
// generic pointer and a flat load (vector-memory latency even for LDS, plus a pointer table in scratch)
template <class T>
using lds_ptr = __attribute__((address_space(3))) T *;
template <class T>
__device__ __forceinline__ lds_ptr<T> to_lds(T *p) { return (lds_ptr<T>)p; }

template <int HEAP_LDS>
struct TourSlots {  // slot arrays: the first HEAP_LDS slots in LDS, the rest (same index) in global
    lds_ptr<double> l_cost; lds_ptr<uint32_t> l_seq; lds_ptr<int32_t> l_cand, l_off, l_len;
    double *g_cost; uint32_t *g_seq; int32_t *g_cand, *g_off, *g_len;
};
struct BlockMinLds { lds_ptr<double> cost; lds_ptr<uint32_t> seq; };
struct BlockMinGlobal { double *cost; uint32_t *seq; };
template <int HEAP_LDS, class BlockMin>
struct TourHeap {
    static_assert(HEAP_LDS % 64 == 0, "blocks do not straddle the LDS / global split");
    TourSlots<HEAP_LDS> a;
    BlockMin b;        // per block of 64 slots: its minimum key (LDS when it fits, launcher)
    int n;             // uniform
    __device__ __forceinline__ int size() const { return n; }
    __device__ __forceinline__ void read(int slot, double &c, uint32_t &s, int32_t &cd, int32_t &o, int32_t &l) const {
        if (slot < HEAP_LDS) { c = a.l_cost[slot]; s = a.l_seq[slot]; cd = a.l_cand[slot]; o = a.l_off[slot]; l = a.l_len[slot]; }
        else { c = a.g_cost[slot]; s = a.g_seq[slot]; cd = a.g_cand[slot]; o = a.g_off[slot]; l = a.g_len[slot]; }
    }
    __device__ __forceinline__ void write(int slot, double c, uint32_t s, int32_t cd, int32_t o, int32_t l) const {
        if (slot < HEAP_LDS) { a.l_cost[slot] = c; a.l_seq[slot] = s; a.l_cand[slot] = cd; a.l_off[slot] = o; a.l_len[slot] = l; }
        else { a.g_cost[slot] = c; a.g_seq[slot] = s; a.g_cand[slot] = cd; a.g_off[slot] = o; a.g_len[slot] = l; }
    }
    __device__ inline void push(const HeapItem &it) {
        const int slot = n++;
        if ((threadIdx.x & 63) == 0) {
            write(slot, it.cost, (uint32_t)it.seq, it.cand, it.off, it.len);
            const int j = slot >> 6;
            bool lower = (slot & 63) == 0;
            if (!lower) { const double bc = b.cost[j]; lower = it.cost < bc || (it.cost == bc && (uint32_t)it.seq < b.seq[j]); }
            if (lower) { b.cost[j] = it.cost; b.seq[j] = (uint32_t)it.seq; }
        }
        FITGNN_WAVE_SYNC();
    }
    // minimum over the slots < n of block j, from this lane's key (kc, ks) of slot j*64 + lane; stored as the block's minimum
    __device__ __forceinline__ void rebuild(int j, double kc, uint32_t ks) {
        const int lane = threadIdx.x & 63;
        const int l = wave_argmin(kc, ks, j * 64 + lane < n);
        if (l >= 0) {
            const double mc = lane_bcast(kc, l);
            const uint32_t ms = lane_bcast(ks, l);
            if (lane == 0) { b.cost[j] = mc; b.seq[j] = ms; }
        }
    }
    __device__ inline HeapItem extract_min() {  // n > 0
        const int lane = threadIdx.x & 63;
        const int nb = (n + 63) >> 6;
        // (1) the block holding the minimum
        double c = 0.0;
        uint32_t s = 0;
        int jb = -1;
        for (int j = lane; j < nb; j += 64) {
            const double cj = b.cost[j];
            const uint32_t sj = b.seq[j];
            if (jb < 0 || cj < c || (cj == c && sj < s)) { c = cj; s = sj; jb = j; }
        }
        const int l1 = wave_argmin(c, s, jb >= 0);
        jb = lane_bcast((int32_t)jb, l1);
        // (2) the slot inside it
        const int slot = jb * 64 + lane;
        double kc = 0.0;
        uint32_t ks = 0;
        int32_t pc = 0, po = 0, pl = 0;
        if (slot < n) read(slot, kc, ks, pc, po, pl);
        const int l2 = wave_argmin(kc, ks, slot < n);
        HeapItem out{lane_bcast(kc, l2), (int64_t)lane_bcast(ks, l2), lane_bcast(pc, l2), lane_bcast(po, l2), lane_bcast(pl, l2), 0};
        const int p = jb * 64 + l2;
        // (3) the last slot fills the hole
        --n;
        double lc = 0.0;
        uint32_t ls = 0;
        if (p != n) {
            int32_t lcd, lo, ll;
            read(n, lc, ls, lcd, lo, ll);  // same address in every lane
            FITGNN_WAVE_SYNC();
            if (lane == 0) write(p, lc, ls, lcd, lo, ll);
            if (lane == l2) { kc = lc; ks = ls; }
        }
        rebuild(jb, kc, ks);
        // the block the last slot left: its minimum changes only if the moved item was that minimum
        const int jl = n >> 6;
        if (p != n && jl != jb && (n & 63) != 0) {
            if (__double_as_longlong(lc) == __double_as_longlong(b.cost[jl]) && ls == b.seq[jl]) {  // wave-uniform
                const int sl = jl * 64 + lane;
                double c2 = 0.0;
                uint32_t s2 = 0;
                if (sl < n) { if (sl < HEAP_LDS) { c2 = a.l_cost[sl]; s2 = a.l_seq[sl]; } else { c2 = a.g_cost[sl]; s2 = a.g_seq[sl]; } }
                rebuild(jl, c2, s2);
            }
        }
        FITGNN_WAVE_SYNC();
        return out;
    }
};

// Marked-node set of the selection: bytes in global memory (batched components: one wave each, node ranges of any size) or a
// bitmap in LDS (single graph: 20 KiB for 165 000 nodes) -- the mark check and the prune then touch no global memory at all.
struct MarksGlobal {
    uint8_t *m;
    __device__ __forceinline__ bool get(int32_t v) const { return m[v] != 0; }
    __device__ __forceinline__ void set(int32_t v) const { m[v] = 1; }
    // mark writes before later reads (waits for every outstanding access of the wave)
    __device__ __forceinline__ void publish() const { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }
};
struct MarksLds {
    lds_ptr<uint32_t> w;
    __device__ __forceinline__ bool get(int32_t v) const { return (w[v >> 5] >> (v & 31)) & 1u; }
    __device__ __forceinline__ void set(int32_t v) const { __hip_atomic_fetch_or(&w[v >> 5], 1u << (v & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    // the LDS pipeline is in order per wave: nothing to wait for, and the prefetches in flight stay in flight
    __device__ __forceinline__ void publish() const { FITGNN_WAVE_SYNC(); }
};

// ---- speculative re-costs (single graph) ----
// 60 % of the selecting wave's time is the re-cost of sets it has just pruned, and 90 % of those are list entries meeting their
// first marked member.  Which members a list entry will keep is almost always known a few pops early (marks only grow, and a
// pop marks a handful of nodes out of thousands), so kSpecWaves helper waves of the same workgroup walk AHEAD of the list head:
// the next unclaimed entry p (a counter in LDS: whichever helper is free takes it) is pruned against the marks of that moment
// and, if something was dropped, costed with the very routine the selecting wave would use; (p, kept count, cost) goes to a
// ring in LDS.  The selecting wave
// uses it only if the kept COUNT equals its own: the helper saw a subset of the marks, so it kept a superset of the members,
// and equal counts mean equal sets -- the cost is then the one it would have computed, bit for bit.  A miss costs nothing
// (it computes, as before); the selecting wave never waits for a helper, and helpers leave when it raises `done`.
constexpr int kSpecWaves = 2;
constexpr int kSpecRing = 64;    // ring entries (entry p in slot p mod kSpecRing; > kSpecAhead + kSpecWaves, so a slot is only reused once its entry is popped)
constexpr int kSpecAhead = 24;   // helpers work on entries less than this far past the list head
struct SpecShared {
    volatile int32_t head;                        // list entries below this index have been popped
    volatile int32_t done;
    volatile int32_t next;                        // first list entry no helper has claimed
    volatile int32_t tag[kSpecRing];  // list index the entry describes (written last)
    volatile int32_t cnt[kSpecRing];
    volatile double cost[kSpecRing];
    volatile int32_t hoff[kSpecRing];   // the helper's stored match list of the pruned set (pool offset, length; -1: none)
    volatile int32_t hn[kSpecRing];
    volatile unsigned long long keep[kSpecRing];   // which of the entry's members the helper kept (bit = position)
    // the re-insertion queue's held-out minimum, published by the selecting wave for the last helper to prune and cost ahead of
    // its pop (job_seq written last), and that helper's answer (res_seq written last)
    volatile int32_t job_seq, job_cand, job_off, job_len;
    volatile int32_t res_seq, res_cnt, res_hoff, res_hn;
    volatile double res_cost;
};
// what a helper has for a set the selecting wave has just pruned to m members
enum SpecAnswer { kSpecNothing = 0, kSpecCost = 1, kSpecList = 2 };
struct SpecNone {
    __device__ __forceinline__ void publish_head(int) const {}
    __device__ __forceinline__ int lookup(int, int, double &, int32_t &, int32_t &, unsigned long long &) const { return kSpecNothing; }
    __device__ __forceinline__ void publish_top(int, int32_t, int32_t, int32_t, bool) const {}
    __device__ __forceinline__ int lookup_top(int, int, double &, int32_t &, int32_t &) const { return kSpecNothing; }
};
struct SpecRing {
    lds_ptr<SpecShared> sh;
    __device__ __forceinline__ void publish_head(int head) const { if ((threadIdx.x & 63) == 0) sh->head = head; }
    // kSpecCost: the helper pruned the entry to the same m members (marks only grow: equal counts = equal sets) -- its cost and its
    // stored match list are this set's.  kSpecList: it kept more (a member was marked after it looked): the cost is of no use, but
    // its match list is that of a superset, which the caller filters (keep = the members it kept) instead of scanning adjacency lists.
    __device__ __forceinline__ int lookup(int lp, int m, double &c, int32_t &hoff, int32_t &hn, unsigned long long &keep) const {
        const int e = lp % kSpecRing;
        if (sh->tag[e] != lp) return kSpecNothing;   // volatile: tag, then the rest (the writer's order reversed)
        const int cnt = sh->cnt[e];
        hoff = sh->hoff[e];
        hn = sh->hn[e];
        if (cnt == m) {
            c = sh->cost[e];
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // the helper's pool entries (same CU) before any later read of them
            return kSpecCost;
        }
        if (cnt < m || hoff < 0 || hn < 0) return kSpecNothing;
        keep = sh->keep[e];
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return kSpecList;
    }
    // the queue's new held-out minimum: its members and stored list are final until it is popped (only the set being processed
    // is ever changed), so a helper may prune and cost it now
    // fresh: the set was pruned in THIS iteration -- its compacted members and list-table entry may still be on their way to memory, so
    // the wave waits for them (a workgroup release is s_waitcnt vmcnt(0): a memory round trip).  A set that comes out of the queue was
    // written at least one pop earlier, and the loads that just fetched it from the queue have been waited for since (the counter
    // retires in order): no wait -- and the request for its members, issued just before, stays in flight.
    __device__ __forceinline__ void publish_top(int seq, int32_t cand, int32_t off, int32_t len, bool fresh) const {
        if (fresh) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        // not fresh: no wait is needed (above), but the COMPILER must still keep the set's earlier global stores ahead of the LDS stores
        // that publish it: a release at wavefront scope emits no instruction and pins that order (the hardware side rests on global_*
        // accesses retiring in order on vmcnt -- tests/test_abi_cpu.py checks that the kernel's ISA holds no flat_ access)
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if ((threadIdx.x & 63) == 0) { sh->job_cand = cand; sh->job_off = off; sh->job_len = len; sh->job_seq = seq; }
    }
    // (a queue set has its own stored list: a helper's list of a superset of it is of no use)
    __device__ __forceinline__ int lookup_top(int seq, int m, double &c, int32_t &hoff, int32_t &hn) const {
        if (sh->res_seq != seq) return kSpecNothing;
        if (sh->res_cnt != m) return kSpecNothing;
        c = sh->res_cost;
        hoff = sh->res_hoff;
        hn = sh->res_hn;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return kSpecCost;
    }
};

// Helper wave `w` (0-based) of the single-graph kernel; see above.  Reads mem / len of list entries that have not been popped
// (immutable until then) and the marks; writes only its ring.
template <class Marks>
__device__ inline void greedy_speculate(const CostGraph &g, CostLds &lds, lds_ptr<SpecShared> sh, int w, int32_t n_list,
                                        const int32_t *__restrict__ set_off, const int32_t *mem, const int32_t *len,
                                        Marks marks, const int32_t *__restrict__ order, uint16_t *pool_ab, double *pool_w,
                                        int64_t pool_begin, int64_t pool_end, const int32_t *hc_off, const int32_t *hc_n) {
    const int lane = threadIdx.x & 63;
    int64_t bump = pool_begin;   // this helper's own region of the match-list pool
    int last_job = 0;
    // The last helper also serves the queue: whenever the selecting wave publishes a new held-out minimum it prunes that set
    // against the marks of the moment and costs it from the set's stored match list -- the selecting wave is busy with the pop
    // before, and when it gets to this set its own prune keeps the same members unless that pop marked one of them (equal counts
    // = equal sets, as for the list entries).  The filtered list goes to a fresh region of this helper's pool.
    auto serve_queue = [&]() {
        if (w != kSpecWaves - 1 || !hc_off) return;
        const int js = sh->job_seq;
        if (js == last_job) return;
        last_job = js;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int32_t cand = sh->job_cand;
        const int off = sh->job_off, nc = sh->job_len;
        if (sh->job_seq != js) return;            // overwritten while being read: the next poll takes the newer job
        if (nc > fitgnn::kCostTile || nc < 3) return;
        const int32_t v = lane < nc ? mem[off + lane] : -1;
        const bool keep = v >= 0 && !marks.get(v);
        const unsigned long long bal = __ballot(keep);
        const int m = __popcll(bal);
        if (m == nc || m < 2) return;             // nothing marked (no re-cost will be needed) / the set will be dropped
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (keep) lds.S[before] = v;
        if (lane < nc) lds.remap[lane] = keep ? (uint8_t)before : (uint8_t)255;
        FITGNN_WAVE_SYNC();
        const int32_t s_off = __builtin_amdgcn_readfirstlane(hc_off[cand]), s_n = __builtin_amdgcn_readfirstlane(hc_n[cand]);
        fitgnn::HitIO io{true, pool_ab, pool_w, bump, pool_end, (int64_t)s_off, s_n, true, true};
        const double cost = fitgnn::set_cost_wave<true>(g, nullptr, m, lds, io);
        bump = io.bump;
        FITGNN_WAVE_SYNC();
        if (sh->job_seq != js) return;            // the minimum changed meanwhile: nobody will ask for this answer
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) {
            sh->res_cost = cost;
            sh->res_cnt = m;
            sh->res_hoff = (int32_t)io.off;
            sh->res_hn = io.n;
            sh->res_seq = js;
        }
    };
    for (;;) {
        if (sh->done) return;
        serve_queue();
        // claim the next list entry once it is inside the window (when the list is used up the queue still pops: keep serving it)
        const int h = sh->head;
        if (sh->next >= n_list || sh->next >= h + kSpecAhead) { __builtin_amdgcn_s_sleep(8); continue; }
        int p = 0;
        if (lane == 0) p = atomicAdd((int *)&sh->next, 1);
        p = __builtin_amdgcn_readfirstlane(p);
        if (p >= n_list) continue;
        if (p <= sh->head) continue;  // popped already, or being popped
        const int32_t c = order[p];
        const int off = __builtin_amdgcn_readfirstlane(set_off[c]);
        const int nc = __builtin_amdgcn_readfirstlane(len[c]);
        if (nc > fitgnn::kCostTile || nc < 3) continue;  // only sets the staged re-cost handles; a pair cannot be pruned to a set
        const int32_t v = lane < nc ? mem[off + lane] : -1;
        const bool keep = v >= 0 && !marks.get(v);
        const unsigned long long bal = __ballot(keep);
        const int m = __popcll(bal);
        if (m == nc || m < 2) continue;  // nothing marked yet / the entry will be dropped
        if (keep) lds.S[__popcll(bal & ((1ull << lane) - 1ull))] = v;
        FITGNN_WAVE_SYNC();
        // the pruned set's match list goes to the pool with its cost: if the selecting wave takes the cost (equal sets), the set's
        // next re-cost filters that list instead of scanning adjacency lists again
        fitgnn::HitIO io{pool_ab != nullptr, pool_ab, pool_w, bump, pool_end, -1, -1, false, false};
        const double cost = fitgnn::set_cost_wave<true>(g, nullptr, m, lds, io);
        bump = io.bump;
        FITGNN_WAVE_SYNC();
        // publish only if the entry is still unpopped: then every read above preceded the selecting wave's changes to it
        if (sh->head > p) continue;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the list's entries before the ring entry that names them
        if (lane == 0) {
            const int e = p % kSpecRing;
            sh->cost[e] = cost;
            sh->cnt[e] = m;
            sh->hoff[e] = (int32_t)io.off;
            sh->hn[e] = io.n;
            sh->keep[e] = bal;
            sh->tag[e] = p;
        }
    }
}

// ---- the match lists of re-inserted sets, kept between their re-costs (variation_cost.h: HitIO) ----
// Only the selecting wave reads or writes them.  hc_off / hc_n [N]: pool offset (-1: no list) and current length (-1: not valid) of
// the list of the set that grew out of candidate `cand`.
struct HitCacheNone {
    __device__ __forceinline__ void begin(fitgnn::HitIO &io, int32_t, bool) { io.active = false; }
    __device__ __forceinline__ void begin_from(fitgnn::HitIO &io, int32_t, int32_t) { io.active = false; }
    __device__ __forceinline__ void end(const fitgnn::HitIO &, int32_t) {}
    __device__ __forceinline__ void adopt(int32_t, int32_t, int32_t) {}
};
struct HitCacheGlobal {
    int32_t *hc_off, *hc_n;
    uint16_t *pool_ab;
    double *pool_w;
    int64_t pool_cap;
    int64_t bump;
    __device__ __forceinline__ void begin(fitgnn::HitIO &io, int32_t cand, bool remap_ok) {
        const int32_t off = __builtin_amdgcn_readfirstlane(hc_off[cand]);
        const int32_t n = __builtin_amdgcn_readfirstlane(hc_n[cand]);
        io.active = true;
        io.pool_ab = pool_ab; io.pool_w = pool_w; io.bump = bump; io.pool_cap = pool_cap;
        io.off = off;
        io.n = remap_ok ? n : -1;   // the translation table covers sets of at most 64 members
        io.use_remap = true;
        io.fresh_out = false;
    }
    // a helper's list of a superset of the set (its region of the pool is never reused: the filtered list replaces it in place)
    __device__ __forceinline__ void begin_from(fitgnn::HitIO &io, int32_t off, int32_t n) {
        io.active = true;
        io.pool_ab = pool_ab; io.pool_w = pool_w; io.bump = bump; io.pool_cap = pool_cap;
        io.off = off;
        io.n = n;
        io.use_remap = true;
        io.fresh_out = false;
    }
    __device__ __forceinline__ void end(const fitgnn::HitIO &io, int32_t cand) {
        bump = io.bump;
        if ((threadIdx.x & 63) == 0) { hc_off[cand] = (int32_t)io.off; hc_n[cand] = io.n; }
    }
    // a helper wave costed this very set and stored its list (SpecRing::lookup)
    __device__ __forceinline__ void adopt(int32_t cand, int32_t off, int32_t n) {
        if ((threadIdx.x & 63) == 0) { hc_off[cand] = off; hc_n[cand] = off >= 0 ? n : -1; }
    }
};

// The greedy selection of contract_variation_linear (:604-650) over ONE connected component, run by one wavefront.
// The component's candidates are order[head0 .. head1) (ascending (cost, node id)); node ids, set_off/mem/len/marks
// are those of the whole (possibly block-diagonal) graph.  Selected sets go to sel_mem[0 .. ) and their END
// positions to sel_end[0 .. ) (both relative to the pointers passed in).  Returns through ns / pos / the residual
// n_reduce.  Every iteration consumes one queue entry and re-insertions strictly shrink a set, so the loop is
// bounded by max_iters = candidates + members + slack (an exit every lane reaches).
//
// A pop is pointer chasing (list entry -> candidate -> its cost / extent -> its members -> their marks): five dependent
// accesses, each an L2 round trip for the one wave that makes them.  The sorted initial family is known in advance, so it is
// read through a three-stage software pipeline -- entry head + 2: id requested; head + 1: cost / extent requested; head:
// complete, its first 64 members in the lanes -- and every request was issued at least one pop earlier; re-inserted sets
// carry their extent in the heap item.  With the marks in LDS a pop from the list touches global memory only to prefetch.
template <class Heap, class Marks, class Spec, class Cache = HitCacheNone>
__device__ inline void greedy_component(const CostGraph &g, CostLds &lds, Heap heap, Spec spec, int32_t head0, int32_t head1,
                                        int64_t seq, const int32_t *__restrict__ set_off, int32_t *__restrict__ mem,
                                        int32_t *__restrict__ len, Marks marks,
                                        const int32_t *__restrict__ order, const double *__restrict__ cost0,
                                        int64_t &n_reduce, int64_t max_iters, int32_t *__restrict__ sel_end,
                                        int32_t *__restrict__ sel_mem, int32_t &ns, int32_t &pos, Cache cache = Cache{}) {
    const int lane = threadIdx.x & 63;
    int head = head0;    // next unread entry of the sorted initial family (uniform)
    ns = 0; pos = 0;
    GSTAMP_DECL;
    // the smallest re-inserted set is held OUT of the heap, with its first 64 members already requested: popping it needs
    // no look-up, and its successor is extracted (and its members requested) while this one is being processed
    bool has_top = false;
    HeapItem top{};
    int32_t tm = -1;
    int top_job = 0, cur_job = 0;   // the held-out minimum's number with the helper that costs it ahead of its pop (Spec)
    // list pipeline (len[] of an unprocessed list entry is its initial length and its members are untouched: only the
    // candidate being processed is ever shrunk)
    int32_t c0 = 0, o0 = 0, l0 = 0, m0 = -1;  // entry head: candidate, extent, first 64 members (m0 per lane)
    double k0 = 0.0;
    int32_t c1 = 0, o1 = 0, l1 = 0;           // entry head + 1
    double k1 = 0.0;
    int32_t c2 = 0;                           // entry head + 2
    auto entry = [&](int h) { return order[min(h, head1 - 1)]; };   // clamped: past the end the value is never used
    if (head < head1) {
        c0 = entry(head); c1 = entry(head + 1); c2 = entry(head + 2);
        k0 = cost0[c0]; o0 = set_off[c0]; l0 = len[c0];
        k1 = cost0[c1]; o1 = set_off[c1]; l1 = len[c1];
        m0 = lane < l0 ? mem[o0 + lane] : -1;
    }
    for (int64_t it = 0; it < max_iters; ++it) {
        if (n_reduce <= 0) break;
        if (head >= head1 && !has_top) break;
        // ---- pop the minimum of {sorted initial list head, smallest re-inserted set}: SortedList.pop(0) ----
        GSTAMP(g0);
        int32_t cand, mm;
        int off, nc;
        bool from_list = !has_top;
        if (head < head1 && has_top) {
            HeapItem li{k0, (int64_t)c0, c0, 0, 0, 0};
            from_list = item_less(li, top);
        }
        if (from_list) {
            cand = __builtin_amdgcn_readfirstlane(c0);
            off = __builtin_amdgcn_readfirstlane(o0);
            nc = __builtin_amdgcn_readfirstlane(l0);
            mm = m0;
            ++head;
            spec.publish_head(head - head0);
            if (head < head1) {  // advance the pipeline: every value used here was requested one pop ago or earlier
                c0 = c1; k0 = k1; o0 = o1; l0 = l1;
                m0 = lane < l0 ? mem[o0 + lane] : -1;
                c1 = c2;
                k1 = cost0[c1]; o1 = set_off[c1]; l1 = len[c1];
                c2 = entry(head + 2);
            }
        } else {
            cand = top.cand; off = top.off; nc = top.len;
            mm = tm;
            cur_job = top_job;
            has_top = heap.size() > 0;
            if (has_top) {
                top = heap.extract_min();
                tm = lane < top.len ? mem[top.off + lane] : -1;
                spec.publish_top(++top_job, top.cand, top.off, top.len, false);
            }
        }
        int32_t *S = mem + off;
        GSTAMP(g1);
        if (from_list) { GACC(0, g0, g1); GCNT(8); } else { GACC(1, g0, g1); GCNT(9); }
        // ---- any member marked? (coarsening_utils.py:620-622) ----
        bool any = __ballot(mm >= 0 && marks.get(mm)) != 0ull;
        for (int t0 = 64; t0 < nc && !any; t0 += 64) {
            const int t = t0 + lane;
            const bool mk = (t < nc) && marks.get(S[t]);
            any |= __ballot(mk) != 0ull;
        }
        GSTAMP(g2);
        GACC(2, g1, g2);
        if (!any) {
            const int64_t gain = nc - 1;
            if (gain > n_reduce) continue;  // :625-626, would over-reduce: drop the set
            for (int t = lane; t < nc; t += 64) {
                const int32_t v = t < 64 ? mm : S[t];
                marks.set(v);
                sel_mem[pos + t] = v;
            }
            pos += nc;
            if (lane == 0) sel_end[ns] = pos;
            ++ns;
            n_reduce -= gain;
            marks.publish();
            GSTAMP(g3);
            GACC(3, g2, g3); GCNT(10);
        } else {
            // ---- drop marked members in place, keep order (:640) ----
            int m = 0;
            for (int t0 = 0; t0 < nc; t0 += 64) {
                const int t = t0 + lane;
                int32_t v = 0;
                bool keep = false;
                if (t < nc) { v = t0 == 0 ? mm : S[t]; keep = !marks.get(v); }
                const unsigned long long bal = __ballot(keep);
                const int before = __popcll(bal & ((1ull << lane) - 1ull));
                FITGNN_WAVE_SYNC();  // all reads of this chunk done before the compacted writes (m <= t0)
                if (keep) {
                    S[m + before] = v;
                    if (m + before < fitgnn::kCostTile) lds.S[m + before] = v;  // staged for the re-cost
                }
                if (t0 == 0 && t < nc) lds.remap[lane] = keep ? (uint8_t)before : (uint8_t)255;  // old position -> new (sets of <= 64)
                m += __popcll(bal);
            }
            GSTAMP(g4);
            GACC(4, g2, g4); GCNT(11);
            if (m > 1) {
                if (lane == 0) len[cand] = m;
                double c;  // :646 re-cost
                int32_t xm;  // the new set's first 64 members, per lane
                if (m <= fitgnn::kCostTile) {
                    FITGNN_WAVE_SYNC();
                    int32_t h_off = -1, h_n = -1;
                    unsigned long long h_keep = 0ull;
                    const int ans = from_list ? spec.lookup(head - 1 - head0, m, c, h_off, h_n, h_keep)
                                              : (nc <= fitgnn::kCostTile ? spec.lookup_top(cur_job, m, c, h_off, h_n) : (int)kSpecNothing);
                    if (ans == kSpecCost) {
                        cache.adopt(cand, h_off, h_n);
#ifdef FITGNN_GREEDY_STAMPS
                        gdbg[from_list ? 15 : 6] += 1;
#endif
                    } else {
                        fitgnn::HitIO io{};
                        if (ans == kSpecList) {
                            // the helper numbered the members it kept 0, 1, ...: its position -> the position after this prune
                            const bool in_h = lane < nc && ((h_keep >> lane) & 1ull) != 0ull;
                            const uint8_t to = in_h ? lds.remap[lane] : (uint8_t)255;
                            const int ph = __popcll(h_keep & ((1ull << lane) - 1ull));
                            FITGNN_WAVE_SYNC();
                            if (in_h) lds.remap[ph] = to;
                            FITGNN_WAVE_SYNC();
                            cache.begin_from(io, h_off, h_n);
#ifdef FITGNN_GREEDY_STAMPS
                            gdbg[16] += 1;
#endif
                        } else {
                            cache.begin(io, cand, nc <= fitgnn::kCostTile);
#ifdef FITGNN_GREEDY_STAMPS
                            if (!(io.active && io.off >= 0 && io.n >= 0)) gdbg[from_list ? 18 : 19] += 1;   // adjacency lists scanned by the selecting wave
#endif
                        }
#ifdef FITGNN_GREEDY_STAMPS
                        if (io.active && io.off >= 0 && io.n >= 0) gdbg[7] += 1;   // re-costs answered from a stored match list
#endif
                        c = fitgnn::set_cost_wave<true>(g, S, m, lds, io);
                        cache.end(io, cand);
                    }
                    xm = lane < m ? lds.S[lane] : -1;
                } else {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // the compacted members are read back
                    c = fitgnn::set_cost_wave<false>(g, S, m, lds);
                    xm = S[lane];
                }
                GSTAMP(g5);
                GACC(5, g4, g5); GCNT(12);
                const HeapItem x{c, seq, cand, off, m, 0};
                ++seq;
                if (!has_top) {
                    top = x; tm = xm; has_top = true;
                    spec.publish_top(++top_job, cand, off, m, true);
                } else if (item_less(x, top)) {
                    heap.push(top);
                    top = x; tm = xm;
                    spec.publish_top(++top_job, cand, off, m, true);
                } else {
                    heap.push(x);
                }
#ifdef FITGNN_GREEDY_STAMPS
                if ((unsigned long long)heap.size() > gdbg[13]) gdbg[13] = heap.size();
                if (from_list) gdbg[14] += 1;
#endif
            }
        }
    }
    GSTAMP_FLUSH;
}

// dynamic LDS of the single-graph kernel when the graph is small enough (launcher): marked-node bitmap, then the block minima
__host__ __device__ inline size_t greedy_bitmap_words(int32_t N) { return ((size_t)N + 31) / 32; }
__host__ __device__ inline size_t greedy_blocks(int32_t N) { return ((size_t)N + 63) / 64 + 1; }
__host__ __device__ inline size_t greedy_dyn_lds_bytes(int32_t N) {
    return ((greedy_bitmap_words(N) * 4 + 7) / 8) * 8 + greedy_blocks(N) * 12;
}

__global__ __launch_bounds__(64 * (1 + kSpecWaves)) void greedy_select_kernel(
    CostGraph g, int32_t N, const int32_t *__restrict__ set_off, int32_t *mem, int32_t *len, uint8_t *__restrict__ marked,
    const int32_t *__restrict__ order, const double *__restrict__ cost0, char *__restrict__ heap_glob, int64_t n_reduce,
    int64_t max_iters, int32_t *__restrict__ sel_off, int32_t *__restrict__ sel_mem, int32_t *__restrict__ sel_count,
    int32_t state_in_lds, int32_t *hc_off, int32_t *hc_n, uint16_t *hc_ab, double *hc_w, int64_t hc_cap) {
    __shared__ CostLds lds[1 + kSpecWaves];
    __shared__ double h_cost[kHeapLds];
    __shared__ uint32_t h_seq[kHeapLds];
    __shared__ int32_t h_cand[kHeapLds], h_off[kHeapLds], h_len[kHeapLds];
    __shared__ SpecShared spec_sh;
    extern __shared__ double dyn_lds[];  // state_in_lds: [bitmap | block minima: cost, seq] (greedy_dyn_lds_bytes)
    const int wave = threadIdx.x >> 6;
    // wave 0 selects; with the marks in LDS the other waves speculate ahead of it, otherwise they leave at once
    uint32_t *mark_bits = (uint32_t *)dyn_lds;
    const size_t words = greedy_bitmap_words(N);
    if (state_in_lds) {
        for (int i = threadIdx.x; i < (int)words; i += blockDim.x) mark_bits[i] = 0u;
        for (int i = threadIdx.x; i < kSpecRing; i += blockDim.x) spec_sh.tag[i] = -1;
        if (threadIdx.x == 0) { spec_sh.head = 0; spec_sh.done = 0; spec_sh.next = 0; spec_sh.job_seq = 0; spec_sh.res_seq = 0; }
        __syncthreads();  // the only workgroup barrier: nothing below waits for another wave
        if (wave > 0) {
            // the match-list pool: first half the selecting wave's, the second half split between the helpers
            const int64_t half = hc_cap / 2, per = (hc_cap - half) / kSpecWaves;
            greedy_speculate(g, lds[wave], to_lds(&spec_sh), wave - 1, N, set_off, mem, len, MarksLds{to_lds(mark_bits)}, order, hc_ab, hc_w,
                             half + (wave - 1) * per, half + wave * per, hc_off, hc_n);
            return;
        }
    } else if (wave > 0) {
        return;
    }
    if (threadIdx.x == 0) sel_off[0] = 0;
    int32_t ns, pos;
    // slots beyond kHeapLds and, for a graph too large for LDS, the block minima: carved from the heap workspace (32 B / node)
    const size_t n = (size_t)N;
    TourSlots<kHeapLds> slots{to_lds(h_cost), to_lds(h_seq), to_lds(h_cand), to_lds(h_off), to_lds(h_len),
                              (double *)heap_glob, (uint32_t *)(heap_glob + 8 * n), (int32_t *)(heap_glob + 12 * n),
                              (int32_t *)(heap_glob + 16 * n), (int32_t *)(heap_glob + 20 * n)};
    // re-inserted sets get seq = N, N+1, ... (initial family: seq = node id)
    if (state_in_lds) {
        double *b_cost = dyn_lds + (words * 4 + 7) / 8;
        uint32_t *b_seq = (uint32_t *)(b_cost + greedy_blocks(N));
        greedy_component(g, lds[0], TourHeap<kHeapLds, BlockMinLds>{slots, BlockMinLds{to_lds(b_cost), to_lds(b_seq)}, 0},
                         SpecRing{to_lds(&spec_sh)}, 0, N, (int64_t)N, set_off, mem, len, MarksLds{to_lds(mark_bits)}, order, cost0,
                         n_reduce, max_iters, sel_off + 1, sel_mem, ns, pos, HitCacheGlobal{hc_off, hc_n, hc_ab, hc_w, hc_cap / 2, 0});
        if (threadIdx.x == 0) spec_sh.done = 1;
    } else {
        double *b_cost = (double *)(heap_glob + 24 * n);
        uint32_t *b_seq = (uint32_t *)(b_cost + greedy_blocks(N));
        greedy_component(g, lds[0], TourHeap<kHeapLds, BlockMinGlobal>{slots, BlockMinGlobal{b_cost, b_seq}, 0}, SpecNone{}, 0, N,
                         (int64_t)N, set_off, mem, len, MarksGlobal{marked}, order, cost0, n_reduce, max_iters, sel_off + 1, sel_mem, ns,
                         pos, HitCacheGlobal{hc_off, hc_n, hc_ab, hc_w, hc_cap, 0});
    }
    if (threadIdx.x == 0) { sel_count[0] = ns; sel_count[1] = pos; }
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
        greedy_component(g, lds, BinHeap<kHeapLdsBatch>{heap_lds, heap_glob + b, 0}, SpecNone{}, b, e, (int64_t)N, set_off, mem, len,
                         MarksGlobal{marked}, order, cost0, n_reduce, max_iters, stage_end + b, stage_mem + b, ns, pos);
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
    size_t mem, len, marked, keys_in, keys_out, ids_in, order, heap, sort_tmp, sort_tmp_bytes, hc_off, hc_n, hc_ab, hc_w, hc_cap, total;
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
    L.heap = o; o += align_up(n * sizeof(HeapItem) + 64);  // single graph: 24 B of slot arrays + 12/64 B of block minima per node
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, (uint64_t *)nullptr, (uint64_t *)nullptr, (int32_t *)nullptr,
                                    (int32_t *)nullptr, n, 0, 64, (hipStream_t)0);
    L.sort_tmp_bytes = tmp;
    L.sort_tmp = o; o += align_up(tmp);
    // match lists of re-inserted sets (HitCacheGlobal): two ints per node + a pool of one entry per family member
    L.hc_off = o; o += align_up(n * 4);
    L.hc_n = o; o += align_up(n * 4);
    L.hc_cap = tm;
    L.hc_ab = o; o += align_up(tm * 2);
    L.hc_w = o; o += align_up(tm * 8);
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
    int rc = (int)hipDeviceSynchronize();
    rc |= (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_greedy_dbg), sizeof(unsigned long long) * 24);
    if (reset) {
        unsigned long long z[24] = {0};
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_greedy_dbg), z, sizeof(z));
    }
    return rc;
}
extern "C" int fitgnn_debug_cost_counters(unsigned long long *out, int reset) {
    int rc = (int)hipDeviceSynchronize();
    rc |= (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fitgnn::g_cost_dbg), sizeof(unsigned long long) * 12);
    if (reset) {
        unsigned long long z[12] = {0};
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(fitgnn::g_cost_dbg), z, sizeof(z));
    }
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
    FITGNN_RETURN_IF_HIP(hipMemsetAsync(base + L.hc_off, 0xff, (size_t)N * 4, s));   // -1: no list
    FITGNN_RETURN_IF_HIP(hipMemsetAsync(base + L.hc_n, 0xff, (size_t)N * 4, s));
    hipLaunchKernelGGL(len_init_kernel, blocks_for(N), dim3(256), 0, s, set_off, N, len);
    hipLaunchKernelGGL(cost_keys_kernel, blocks_for(N), dim3(256), 0, s, cost0, N, keys_in, ids_in);
    // stable LSD radix sort on the cost bits: ties keep ascending node id == SortedList's stable build
    size_t tmp = L.sort_tmp_bytes;
    FITGNN_RETURN_IF_HIP(rocprim::radix_sort_pairs((void *)(base + L.sort_tmp), tmp, keys_in, keys_out, ids_in, order,
                                                   (size_t)N, 0, 64, s));
    CostGraph g{rowptr, col, w, dw, A, K, lda, nullptr};
    const int64_t max_iters = (int64_t)N + (int64_t)total + 8;
    // marked-node bitmap and the queue's block minima in LDS while they fit beside the slots and the cost scratch
    const size_t dyn_bytes = greedy_dyn_lds_bytes(N);
    const int state_in_lds = dyn_bytes <= (size_t)kStateLdsBytes ? 1 : 0;
    static std::atomic<uint64_t> lds_done{0};
    if (const int rc = fitgnn_lds_limit_once((const void *)greedy_select_kernel, kStateLdsBytes, lds_done)) return rc;
    hipLaunchKernelGGL(greedy_select_kernel, dim3(1), dim3(64 * (1 + kSpecWaves)), state_in_lds ? dyn_bytes : 0, s, g, N, set_off, mem, len, marked, order,
                       cost0, (char *)heap, n_reduce, max_iters, sel_off, sel_mem, sel_count, state_in_lds, (int32_t *)(base + L.hc_off),
                       (int32_t *)(base + L.hc_n), (uint16_t *)(base + L.hc_ab), (double *)(base + L.hc_w), (int64_t)L.hc_cap);
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
