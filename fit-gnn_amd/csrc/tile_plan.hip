// tile_plan.hip -- host-side planner of the SpMM row tiles (runs once per static batch).
//
// FIT-GNN rebuilds its block-diagonal PyG batch every epoch (run.py:336 G_DataLoader, shuffle=False); the
// batches never change, so here the tiling is planned once.  A tile is a run of consecutive output rows plus
// the SET of operand rows they reference (the window the kernel stages in LDS).  Small subgraphs pack several
// to a tile with a contiguous window; a subgraph larger than the window (hub clusters) is cut into pieces
// whose windows hold exactly the rows each piece touches (its own rows + the hub + ...), so almost every
// non-zero still finds its operand in LDS.
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "fitgnn_hip.h"

namespace {

struct Planner {
    const int32_t *rowptr, *col;
    int32_t max_rows, max_window;
    fitgnn_tile_t *tiles;
    int32_t *win_cols, *lcol;
    int32_t nt = 0, nw = 0;
    std::vector<int32_t> stamp, slot, cur;

    // One tile of rows [r0, r1) whose window is exactly the contiguous operand rows [c0, c0 + wn).
    void emit_contiguous(int32_t r0, int32_t r1, int32_t c0, int32_t wn) {
        fitgnn_tile_t t;
        t.row_begin = r0; t.row_end = r1; t.win_begin = c0; t.win_rows = wn;
        t.nnz_begin = rowptr[r0]; t.nnz_end = rowptr[r1];
        t.reserved[0] = 0; t.reserved[1] = 0;
        for (int32_t e = rowptr[r0]; e < rowptr[r1]; ++e) {
            const int32_t s = col[e] - c0;
            lcol[e] = (s >= 0 && s < wn) ? s : -(col[e] + 1);
        }
        tiles[nt++] = t;
    }

    // Rows [r0, r1): greedy tiles whose window is the SET of operand rows the tile's rows reference.
    void plan_column_sets(int32_t r0, int32_t r1) {
        int32_t r = r0;
        while (r < r1) {
            const int32_t tile_id = nt;
            cur.clear();
            int32_t r_end = r;
            while (r_end < r1 && r_end - r < max_rows) {
                int32_t add = 0;
                for (int32_t e = rowptr[r_end]; e < rowptr[r_end + 1]; ++e)
                    if (stamp[col[e]] != tile_id) { stamp[col[e]] = tile_id; ++add; cur.push_back(col[e]); }
                if ((int32_t)cur.size() > max_window && r_end > r) {  // does not fit: undo this row's admissions
                    for (int32_t k = 0; k < add; ++k) { stamp[cur.back()] = -1; cur.pop_back(); }
                    break;
                }
                ++r_end;
                if ((int32_t)cur.size() >= max_window) break;
            }
            // window = the admitted columns, ascending; a single row wider than the window keeps its first
            // max_window columns, the rest stay global
            std::sort(cur.begin(), cur.end());
            const int32_t wn = std::min<int32_t>((int32_t)cur.size(), max_window);
            for (int32_t k = 0; k < (int32_t)cur.size(); ++k) slot[cur[k]] = k < wn ? k : -1;
            const bool contiguous = wn > 0 && cur[wn - 1] - cur[0] == wn - 1;
            fitgnn_tile_t t;
            t.row_begin = r; t.row_end = r_end; t.win_rows = wn;
            t.nnz_begin = rowptr[r]; t.nnz_end = rowptr[r_end];
            t.reserved[1] = 0;
            if (contiguous || wn == 0) {
                t.win_begin = wn ? cur[0] : 0;
                t.reserved[0] = 0;
            } else {
                t.win_begin = nw;  // offset into win_cols
                t.reserved[0] = 1;
                for (int32_t k = 0; k < wn; ++k) win_cols[nw + k] = cur[k];
                nw += wn;
            }
            for (int32_t e = rowptr[r]; e < rowptr[r_end]; ++e) {
                const int32_t s = slot[col[e]];
                lcol[e] = s >= 0 ? s : -(col[e] + 1);  // miss: the global column, encoded negative
            }
            for (int32_t c : cur) stamp[c] = -1;
            tiles[nt++] = t;
            r = r_end;
        }
    }
};

}  // namespace

extern "C" int fitgnn_plan_tiles_host(const int32_t *rowptr, const int32_t *col, int32_t n_rows, int32_t n_cols,
                                      const int64_t *block_ptr, int32_t n_blocks, int32_t max_rows, int32_t max_window,
                                      fitgnn_tile_t *tiles, int32_t *n_tiles, int32_t *win_cols, int32_t *n_win,
                                      int32_t *lcol) {
    if (n_rows < 0 || n_cols < 0 || max_rows < 1 || max_window < 1 || !n_tiles || !n_win) return FITGNN_E_BADARG;
    *n_tiles = 0;
    *n_win = 0;
    if (n_rows == 0) return 0;
    if (!rowptr || !tiles || !win_cols || (rowptr[n_rows] > 0 && (!col || !lcol))) return FITGNN_E_BADARG;
    if (block_ptr && (n_blocks < 1 || block_ptr[0] != 0 || block_ptr[n_blocks] != n_rows)) return FITGNN_E_BADARG;
    Planner P{rowptr, col, max_rows, max_window, tiles, win_cols, lcol};
    P.stamp.assign((size_t)n_cols, -1);
    P.slot.assign((size_t)n_cols, 0);
    if (!block_ptr) {
        P.plan_column_sets(0, n_rows);  // unstructured pattern
    } else {
        // diagonal blocks (disjoint subgraphs): whole blocks are packed into contiguous-window tiles while they
        // fit; a block larger than the window is planned on its own with column-set windows
        const int32_t cap = std::min(max_rows, max_window);
        int32_t b = 0;
        while (b < n_blocks) {
            const int32_t s0 = (int32_t)block_ptr[b];
            if ((int32_t)block_ptr[b + 1] - s0 > cap) {
                P.plan_column_sets(s0, (int32_t)block_ptr[b + 1]);
                ++b;
                continue;
            }
            int32_t e = b + 1;
            while (e < n_blocks && (int32_t)block_ptr[e + 1] - s0 <= cap) ++e;
            P.emit_contiguous(s0, (int32_t)block_ptr[e], s0, (int32_t)block_ptr[e] - s0);
            b = e;
        }
    }
    *n_tiles = P.nt;
    *n_win = P.nw;
    return 0;
}

// ---- contiguous-window tiles and whole-subgraph block records from the diagonal-block boundaries (host; once per static batch) ----
namespace {
// consecutive blocks packed into tiles of at most max_rows rows (window == the tile's own rows); a block beyond max_rows is cut into
// max_rows-row pieces.  Appends (row_begin, row_end, win_begin, win_rows) records; returns false when `cap` records do not suffice.
bool pack_tiles(const int64_t *ptr, int64_t b0, int64_t b1, int64_t max_rows, int32_t *out, int64_t cap, int64_t &n) {
    int64_t b = b0;
    while (b < b1) {
        const int64_t start = ptr[b], size = ptr[b + 1] - start;
        if (size > max_rows) {
            for (int64_t s = start; s < start + size; s += max_rows) {
                const int64_t e = std::min(s + max_rows, start + size);
                if (n >= cap) return false;
                int32_t *t = out + 4 * n++;
                t[0] = (int32_t)s; t[1] = (int32_t)e; t[2] = (int32_t)s; t[3] = (int32_t)(e - s);
            }
            ++b;
            continue;
        }
        int64_t e = b + 1;   // furthest block end within start + max_rows
        while (e < b1 && ptr[e + 1] - start <= max_rows) ++e;
        if (n >= cap) return false;
        int32_t *t = out + 4 * n++;
        t[0] = (int32_t)start; t[1] = (int32_t)ptr[e]; t[2] = (int32_t)start; t[3] = (int32_t)(ptr[e] - start);
        b = e;
    }
    return true;
}
}  // namespace

extern "C" int fitgnn_make_tiles_host(const int64_t *ptr, int64_t n_blocks, int32_t max_rows, int32_t *tiles4, int64_t capacity,
                                      int64_t *n_tiles) {
    if (n_blocks < 0 || max_rows < 1 || capacity < 0 || !n_tiles) return FITGNN_E_BADARG;
    *n_tiles = 0;
    if (n_blocks == 0) return 0;
    if (!ptr || !tiles4) return FITGNN_E_BADARG;
    int64_t n = 0;
    if (!pack_tiles(ptr, 0, n_blocks, max_rows, tiles4, capacity, n)) return FITGNN_E_WORKSPACE;
    *n_tiles = n;
    return 0;
}

extern "C" int fitgnn_split_blocks_host(const int64_t *ptr, int64_t n_blocks, const int32_t *rowptr, int32_t cap, int64_t limit,
                                        int32_t long_row, int32_t *tiles4, int64_t tiles_capacity, int64_t *n_tiles, int32_t *blocks8,
                                        int64_t *n_large, int32_t *long_rows, int64_t long_capacity, int64_t *n_long) {
    if (n_blocks < 0 || cap < 1 || !n_tiles || !n_large || !n_long) return FITGNN_E_BADARG;
    *n_tiles = *n_large = *n_long = 0;
    if (n_blocks == 0) return 0;
    if (!ptr || !rowptr || !tiles4 || !blocks8 || !long_rows) return FITGNN_E_BADARG;
    int64_t nt = 0, nl = 0, nlong = 0;
    int64_t b = 0;
    while (b < n_blocks) {
        const int64_t size = ptr[b + 1] - ptr[b];
        const bool large = size > cap && size <= limit;
        if (!large) {   // a maximal run of other blocks -> tiles (never packed across a whole-subgraph block)
            int64_t e = b;
            while (e < n_blocks && !((ptr[e + 1] - ptr[e]) > cap && (ptr[e + 1] - ptr[e]) <= limit)) ++e;
            if (!pack_tiles(ptr, b, e, cap, tiles4, tiles_capacity, nt)) return FITGNN_E_WORKSPACE;
            b = e;
            continue;
        }
        const int64_t r0 = ptr[b], r1 = ptr[b + 1];
        int32_t *rec = blocks8 + 8 * nl++;
        rec[0] = (int32_t)r0; rec[1] = (int32_t)r1; rec[2] = rowptr[r0]; rec[3] = rowptr[r1];
        rec[4] = (int32_t)nlong; rec[6] = rec[7] = 0;
        int32_t cnt = 0;
        for (int64_t r = r0; r < r1; ++r)
            if (rowptr[r + 1] - rowptr[r] > long_row) {
                if (nlong >= long_capacity) return FITGNN_E_WORKSPACE;
                long_rows[nlong++] = (int32_t)r;
                ++cnt;
            }
        rec[5] = cnt;
        ++b;
    }
    *n_tiles = nt; *n_large = nl; *n_long = nlong;
    return 0;
}
