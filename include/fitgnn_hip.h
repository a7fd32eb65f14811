/*
 * fitgnn_hip.h -- C ABI of libfitgnn_hip.so: the MI355X (gfx950) implementation of FIT-GNN's
 * coarsen-then-train hot path.  Plain pointers and sizes only; every pointer is a DEVICE pointer
 * unless it says "host".  All entry points:
 *   - return 0 on success, a FITGNN_E_* code (<0) for argument errors, or a hipError_t (>0);
 *   - are asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - never allocate, free or retain memory: the caller owns every buffer, including workspaces,
 *     whose sizes come from the *_workspace_bytes() queries (host-side, no GPU work);
 *   - are re-entrant across streams and devices (no global mutable state).
 *
 * Each function names the reference interface (file:line under the FIT-GNN repository) it replaces.
 * Reference-side bindings (ctypes stubs a maintainer would add): INTEGRATION.md.
 */
#ifndef FITGNN_HIP_H
#define FITGNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FITGNN_ABI_VERSION 1
#define FITGNN_MAX_K 16 /* columns of the spectral matrix A (reference uses K = 10) */

#define FITGNN_E_BADARG (-1)    /* NULL pointer / negative size / unsupported shape */
#define FITGNN_E_WORKSPACE (-2) /* workspace smaller than *_workspace_bytes() */
#define FITGNN_E_ALIGN (-3)     /* pointer or leading dimension not aligned as documented */

int fitgnn_abi_version(void);
/* Human-readable text for a return code of this library (host pointer, static storage). */
const char *fitgnn_error_string(int code);

/* Measurement aid (bench.py's `roofline.copy_ceiling_GBps`): dst[0..n) = src[0..n), n % 4 == 0, both 16-byte aligned -- a
 * read-once / write-once stream in the shape the SpMM kernels move their rows (one workgroup per 16-KiB chunk, 16-byte
 * non-temporal accesses, eight loads in flight per lane: the fastest row of tools/microbench/copy_probe.hip).  2 * 4 * n bytes
 * over its duration is what an HBM-bound stream reaches on the device at hand. */
int fitgnn_stream_copy_f32(const float *src, float *dst, int64_t n, void *stream);

/* =====================================================================================
 * Train half: GCN-family message passing on block-diagonal subgraph batches
 * replaces: torch_geometric.nn.GCNConv & friends as called from network.py:31,60,90,126,161,197
 * ===================================================================================== */

/* A row tile of a CSR batch: rows [row_begin,row_end) are processed by one workgroup, which stages the
 * rows [win_begin, win_begin+win_rows) of the dense operand in LDS.  Non-zeros whose column falls outside
 * the window are fetched from global memory, so ANY tiling is correct; block-diagonal batches (disjoint
 * subgraphs, utils.py:248) make every column fall inside.  Built once per static batch on the host. */
typedef struct fitgnn_tile {
    int32_t row_begin, row_end; /* output rows of the tile */
    int32_t win_begin;          /* reserved[0]==0: first operand row of a contiguous window;
                                   reserved[0]==1: offset into win_cols[] listing the window's operand rows */
    int32_t win_rows;           /* operand rows staged in LDS */
    int32_t nnz_begin, nnz_end; /* = rowptr[row_begin], rowptr[row_end]: lets the kernel fetch the tile's CSR
                                   slice without first waiting on a row-pointer load */
    int32_t reserved[2];        /* [0]: window kind (above); [1]: must be 0 */
} fitgnn_tile_t;

/* Tile order: the kernel runs the tile at array position p on XCD p % 8 (blocks are dealt round-robin over the
 * 8 XCDs).  Any order is correct; for speed give each XCD a contiguous, equally heavy range of the batch,
 * interleaved as tiles[j*8 + k] = range_k[j], padding short ranges with empty tiles (row_begin == row_end).
 *
 * HOST function (host pointers, no GPU work): plan the tiles of a square CSR pattern.  block_ptr[0..n_blocks]
 * (may be NULL) are the row offsets of its diagonal blocks (disjoint subgraphs): whole blocks are packed into
 * tiles with contiguous windows while they fit min(max_rows, max_window) rows; a larger block -- or the whole
 * pattern when block_ptr is NULL -- is cut into runs of consecutive rows that have <= max_rows rows and
 * reference <= max_window distinct operand rows (their window is that set of rows).  Outputs:
 *   tiles    [capacity n_rows], *n_tiles
 *   win_cols [capacity nnz], *n_win      operand rows of the non-contiguous windows
 *   lcol     [nnz]   per non-zero: LDS slot of its operand row inside its tile's window (>= 0), or
 *                    -(col+1) when the operand row is not staged (read from global memory instead) */
int fitgnn_plan_tiles_host(const int32_t *rowptr, const int32_t *col, int32_t n_rows, int32_t n_cols,
                           const int64_t *block_ptr, int32_t n_blocks, int32_t max_rows, int32_t max_window,
                           fitgnn_tile_t *tiles, int32_t *n_tiles, int32_t *win_cols, int32_t *n_win, int32_t *lcol);

/* HOST functions (host pointers, no GPU work) for batches whose windows are the tiles' own rows (every column of a block-diagonal
 * batch is then a window hit): ptr[0..n_blocks] = row offsets of the diagonal blocks (the subgraphs of utils.py:248, or the stars of
 * a star-by-star layout).
 *   fitgnn_make_tiles_host   consecutive blocks packed into tiles of <= max_rows rows, a larger block cut into max_rows-row pieces;
 *                            tiles4: (row_begin, row_end, win_begin, win_rows) records, capacity >= n_blocks + rows / max_rows.
 *   fitgnn_split_blocks_host blocks of cap < rows <= limit become fitgnn_block_t records for fitgnn_spmm_csr_blocks_f32 (blocks8, in
 *                            input order, with their long rows -- more than long_row non-zeros -- listed ascending in long_rows);
 *                            the maximal runs of the other blocks are packed into tiles as above (never across a block record).
 * Both return FITGNN_E_WORKSPACE when an output capacity is too small. */
int fitgnn_make_tiles_host(const int64_t *ptr, int64_t n_blocks, int32_t max_rows, int32_t *tiles4, int64_t capacity, int64_t *n_tiles);
int fitgnn_split_blocks_host(const int64_t *ptr, int64_t n_blocks, const int32_t *rowptr, int32_t cap, int64_t limit, int32_t long_row,
                             int32_t *tiles4, int64_t tiles_capacity, int64_t *n_tiles, int32_t *blocks8, int64_t *n_large,
                             int32_t *long_rows, int64_t long_capacity, int64_t *n_long);

/* A run of consecutive rows LARGER than the SpMM window -- a diagonal block (one subgraph of a block-diagonal batch) or a
 * mostly self-contained segment of one (a star: a centre row and the rows that reference it) -- handled whole by one
 * workgroup per column slab (fitgnn_spmm_csr_blocks_f32): every operand row inside the run is read once; columns outside it
 * are gathered.  long_rows[long_off .. long_off + n_long) are the run's rows with many non-zeros (ascending row ids; the
 * first 8 are carried, see spmm.hip). */
typedef struct fitgnn_block {
    int32_t row_begin, row_end; /* rows of the run */
    int32_t nnz_begin, nnz_end; /* = rowptr[row_begin], rowptr[row_end] */
    int32_t long_off, n_long;   /* the block's long rows in long_rows[] */
    int32_t reserved[2];        /* must be 0 */
} fitgnn_block_t;

/* Record order: as for tiles, the record at array position p runs on XCD p % 8 (any order is correct; for speed give each XCD a
 * contiguous range of the batch, interleaved as blocks[j*8 + k] = range_k[j], short ranges padded with empty records).
 * Y[rows of the listed blocks] = epilogue(A @ X) for diagonal blocks larger than the window; same arguments and epilogue
 * semantics as fitgnn_spmm_csr_f32, same bits as that kernel on the same rows.  A batch is covered by ONE call of each:
 * fitgnn_spmm_csr_f32 over tiles that pack the small blocks, this one over the large blocks.  Requires H % 4 == 0, 16-byte
 * aligned rows (FITGNN_E_BADARG / FITGNN_E_ALIGN otherwise: tile those blocks instead).  xrow (may be NULL): row indirection into a
 * de-duplicated operand table, as in fitgnn_spmm_csr_f32; xcol (may be NULL; needs xrow): xcol[e] = xrow[col[e]] for every CSR
 * entry, built once per batch -- an entry whose operand row is gathered then costs one dependent load instead of two.
 * xrow_zero_from (with xrow; -1: none): operand rows with index >= xrow_zero_from are rows of zeros (the tail of a compact operand):
 * they are not loaded (window rows staged as zeros, gathered entries served from an LDS row of zeros) -- same result. */
int fitgnn_spmm_csr_blocks_f32(const int32_t *rowptr, const int32_t *col, const float *val, const float *X, int64_t ldx,
                               float *Y, int64_t ldy, int32_t n_rows, int32_t H, const fitgnn_block_t *blocks,
                               int32_t n_blocks, const int32_t *long_rows, const int32_t *xrow, const int32_t *xcol,
                               int32_t xrow_zero_from, const float *bias, uint32_t epilogue, float p_drop, uint64_t seed,
                               const uint8_t *mask, void *stream);

/* The same two products as the input gradient of a fused layer output  prev = dropout(ELU(z))  (network.py:32-33): what is stored is
 *   Y[row] = keep ? (A @ X)[row] / (1 - p) * (e > 0 ? 1 : e + 1) : 0,   e = prev[row] * (1 - p)
 * -- fitgnn_epilogue_bwd_f32's arithmetic with the forward's flags (FITGNN_EPI_ELU and/or FITGNN_EPI_DROPOUT), seed and mask --
 * so that a backward SpMM whose result is the gradient w.r.t. the previous layer's output writes that layer's dZ directly and the
 * un-transformed gradient is never written and re-read.  prev: contiguous [n_rows x H], 16-byte aligned; H % 4 == 0.
 * col_part (may be NULL): [n_tiles (resp. n_blocks) x H], ZEROED by the caller; row t receives the column sums of the rows tile /
 * block t stored, in a fixed order -- sum them (fitgnn_colsum_partials_f32) for the bias gradient. */
int fitgnn_spmm_csr_dz_f32(const int32_t *rowptr, const int32_t *col, const float *val, const float *X, int64_t ldx, float *Y,
                           int64_t ldy, int32_t n_rows, int32_t H, const fitgnn_tile_t *tiles, int32_t n_tiles, const int32_t *lcol,
                           const int32_t *win_cols, const int32_t *xrow, int32_t xrow_zero_from, int32_t window_rows, const float *prev,
                           uint32_t epilogue, float p_drop, uint64_t seed, const uint8_t *mask, float *col_part, void *stream);
int fitgnn_spmm_csr_blocks_dz_f32(const int32_t *rowptr, const int32_t *col, const float *val, const float *X, int64_t ldx, float *Y,
                                  int64_t ldy, int32_t n_rows, int32_t H, const fitgnn_block_t *blocks, int32_t n_blocks,
                                  const int32_t *long_rows, const int32_t *xrow, const int32_t *xcol, int32_t xrow_zero_from,
                                  const float *prev, uint32_t epilogue, float p_drop, uint64_t seed, const uint8_t *mask,
                                  float *col_part, void *stream);

/* One step of the thick-restart Lanczos iteration of the spectral prelude (coarsening_utils.py:83-90 hands T = 2 max(dw) I - L to
 * ARPACK; fitgnn_amd.coarsening.lanczos_smallest iterates on the device), float64, over a COLUMN-major basis
 * V [m + 1][ldv] (basis vector c = V + c * ldv, ldv >= n):
 *   fitgnn_lanczos_spmv_f64     y = alpha (A x) + beta x for a CSR matrix (int32 indices, f64 values): with A = L, alpha = -1,
 *                               beta = 2 max(dw) the reference's shifted operator, never built;
 *   fitgnn_lanczos_project_f64  w -= sum_{c < ncol} V[c] h_in[c] (h_in NULL: no subtraction);  part_out[b][c] = workgroup b's share of
 *                               V[c] . w for c < ncol and, in column ncol, of w . w (row stride ncol + 1);
 *                               fitgnn_lanczos_parts(n) workgroups;  ncol <= 128;
 *   fitgnn_lanczos_reduce_f64   out[c] = sum over the n_part partial rows of part[.][c], c < ncol1 (= ncol + 1), in a fixed order;
 *   fitgnn_lanczos_finish_f64   beta = sqrt(hc[j + 1]);  V[j + 1] = w / max(beta, 1e-300);  H[c][j] = ha[c] + hb[c] for c <= j (the
 *                               coefficients of the two orthogonalisation passes), H[j + 1][j] = beta;  H row-major, row stride ldh;
 *   fitgnn_lanczos_rotate_f64   out[c] = sum_{j < m} S[j * nk + c] V[j], c < nk <= 16 (a restart's Ritz vectors), out row stride ldo.
 * No atomics: every sum runs in a fixed order, the iteration is reproducible. */
int32_t fitgnn_lanczos_parts(int32_t n);
int fitgnn_lanczos_spmv_f64(const int32_t *rowptr, const int32_t *col, const double *val, const double *x, double *y, int32_t n,
                            double alpha, double beta, void *stream);
int fitgnn_lanczos_project_f64(const double *V, int64_t ldv, int32_t ncol, double *w, int32_t n, const double *h_in, double *part_out,
                               void *stream);
int fitgnn_lanczos_reduce_f64(const double *part, int32_t n_part, int32_t ncol1, double *out, void *stream);
int fitgnn_lanczos_finish_f64(double *V, int64_t ldv, int32_t j, const double *w, int32_t n, const double *ha, const double *hb,
                              const double *hc, double *H, int32_t ldh, void *stream);
int fitgnn_lanczos_rotate_f64(const double *V, int64_t ldv, int32_t m, const double *S, int32_t nk, double *out, int64_t ldo, int32_t n,
                              void *stream);

/* The same products by the segment-streaming kernel (csrc/spmm.hip: spmm_stream_kernel): the whole-subgraph kernel's algorithm
 * with one WAVE per run of segments and no LDS.  seg_ptr [n_seg + 1] (ascending, seg_ptr[0] = 0, seg_ptr[n_seg] = n_rows) cuts the
 * rows into segments whose FIRST row is the hub (star-by-star layout of an --extra_node union: an own node followed by the extra
 * nodes it brought in, utils.py:235-239); range_seg [n_ranges + 1] (ascending segment indices, range_seg[0] = 0,
 * range_seg[n_ranges] = n_seg) gives every wave its run of whole segments.  A row's operand row is read once, by the wave that
 * streams it; the hub's accumulator is carried across its segment in CSR order (same bits as fitgnn_spmm_csr_f32).  xrow / xcol
 * (both or neither): operand row of union row r is X[xrow[r]], of CSR entry e X[xcol[e]] (= xrow[col[e]]).  _dz_: the store
 * applies the previous layer's ELU' / dropout' (as fitgnn_spmm_csr_dz_f32); col_part (may be NULL) receives one partial row of
 * column sums of dZ per range, [n_ranges x H], every element written.  H, ldx, ldy multiples of 4; X, Y, prev 16-byte aligned. */
int fitgnn_spmm_csr_stream_f32(const int32_t *rowptr, const int32_t *col, const float *val, int64_t nnz, const float *X, int64_t ldx,
                               float *Y, int64_t ldy, int32_t n_rows, int32_t H, const int32_t *seg_ptr, int32_t n_seg,
                               const int32_t *range_seg, int32_t n_ranges, const int32_t *xrow, const int32_t *xcol, const float *bias,
                               uint32_t epilogue, float p_drop, uint64_t seed, const uint8_t *mask, void *stream);
int fitgnn_spmm_csr_stream_dz_f32(const int32_t *rowptr, const int32_t *col, const float *val, int64_t nnz, const float *X, int64_t ldx,
                                  float *Y, int64_t ldy, int32_t n_rows, int32_t H, const int32_t *seg_ptr, int32_t n_seg,
                                  const int32_t *range_seg, int32_t n_ranges, const int32_t *xrow, const int32_t *xcol, const float *prev,
                                  uint32_t epilogue, float p_drop, uint64_t seed, const uint8_t *mask, float *col_part, void *stream);

/* The same two products for a COMPACT operand, by a row-streaming kernel (csrc/spmm.hip: spmm_rows_compact_kernel): X holds the
 * operand rows of a selection followed by rows of zeros (rows >= zero_from), and xcol[e] names the operand row of CSR entry e
 * (xcol[e] = xrow[col[e]] of the entry points above).  This is the last layer's backward SpMM of the train step, whose operand is
 * zero outside the rows that reach the loss (run.py:193-204 keeps out[mask]): every entry is still multiplied and added, in CSR
 * order (same bits as the kernels above), zero rows from registers; each wave streams a contiguous range of rows, no LDS.
 * _dz_: the store applies the previous layer's ELU' / dropout' as fitgnn_spmm_csr_dz_f32 does; col_part (may be NULL) receives
 * fitgnn_spmm_rows_compact_parts(n_rows) partial rows [parts x H] of column sums of dZ (every element written: no zeroing needed),
 * to be folded by fitgnn_colsum_partials_f32.  H, ldx, ldy multiples of 4; X, Y, prev 16-byte aligned. */
int32_t fitgnn_spmm_rows_compact_parts(int32_t n_rows);
int fitgnn_spmm_rows_compact_f32(const int32_t *rowptr, const int32_t *xcol, const float *val, int64_t nnz, const float *X, int64_t ldx,
                                 int32_t zero_from, float *Y, int64_t ldy, int32_t n_rows, int32_t H, void *stream);
int fitgnn_spmm_rows_compact_dz_f32(const int32_t *rowptr, const int32_t *xcol, const float *val, int64_t nnz, const float *X, int64_t ldx,
                                    int32_t zero_from, float *Y, int64_t ldy, int32_t n_rows, int32_t H, const float *prev,
                                    uint32_t epilogue, float p_drop, uint64_t seed, const uint8_t *mask, float *col_part, void *stream);

/* Two backward products in one pass on the whole-subgraph kernel (csrc/spmm.hip: spmm_block_kernel<.., TWO>), for the layer right
 * below a last GCN layer that was evaluated on the loss rows (network.py:29-33: h = dropout(ELU(conv(x))) feeding the last GCNConv;
 * run.py:193-204 keeps out[mask]):
 *     dZ = (A^T Xc) (.) ELU' / dropout' (prev)         fitgnn_spmm_rows_compact_dz_f32's product, NOT stored as a whole
 *     Y  = A^T dZ                                      fitgnn_spmm_csr_blocks_f32's plain product over it, for the listed blocks
 * rowptr / col / val: the CSR of A^T; Xc: the compact operand ([zero_from + zero rows] x H; rows >= zero_from are zero); prev: the
 * layer's forward output [n_rows x H] (contiguous); epilogue / p_drop / seed / mask: the FORWARD's ELU / dropout flags (no
 * FITGNN_EPI_BIAS).  A row's dZ enters the product from the LDS window its piece is staged in; what the kernel cannot serve from
 * LDS it reads from the side table ZT [n_zt x H], filled beforehand by
 *   fitgnn_two_hop_rows_f32: ZT[i] = dZ[zt_rows[i]]  (zt_rows int64; the first zero_from of them are the loss rows in compact order;
 *                            zcol [nnz]: table row of every entry's column, 0x7fffffff if it has none -- a value < zero_from is
 *                            also the column's operand row in Xc).
 * zrow [n_rows]: table row of row r, or -1 for a "simple" row, whose dZ is made while it is staged: at most one of its columns is a
 * loss row, row_p[r] its compact position (0x7fffffff: none) and row_w[r] the entry's value.  The caller's index must put into ZT:
 * the loss rows, the blocks' carried long rows (the first 4 of a block's long_rows), every row with two or more loss columns, and
 * the column of every entry (r, c) that is not served from LDS -- served are: r and c in the same 16-row piece of the same block, c
 * a carried long row of r's block, r a carried long row and c in its block (ops._two_hop_block_index builds exactly this).  Rows
 * outside the listed blocks are not touched: run them through fitgnn_spmm_csr_f32 with X = ZT, xrow = zrow.
 * col_part (may be NULL): [n_blocks x H], ZEROED by the caller, one partial row of column sums of dZ per block.
 * Same bits as fitgnn_spmm_rows_compact_dz_f32 followed by fitgnn_spmm_csr_blocks_f32.  H, ldx, ldy, ldz multiples of 4; 16-byte
 * aligned Xc, Y, prev, ZT. */
int fitgnn_two_hop_rows_f32(const int32_t *rowptr, const int32_t *zcol, const float *val, const float *Xc, int64_t ldx, int32_t zero_from,
                            const int64_t *zt_rows, int32_t n_zt, const float *prev, int32_t H, uint32_t epilogue, float p_drop,
                            uint64_t seed, const uint8_t *mask, float *ZT, int64_t ldz, void *stream);
int fitgnn_spmm_two_hop_blocks_f32(const int32_t *rowptr, const int32_t *col, const float *val, const float *ZT, int64_t ldz, float *Y,
                                   int64_t ldy, int32_t n_rows, int32_t H, const fitgnn_block_t *blocks, int32_t n_blocks,
                                   const int32_t *long_rows, const int32_t *zrow, const int32_t *zcol, const float *prev, const float *Xc,
                                   int64_t ldx, int32_t zero_from, const int32_t *row_p, const float *row_w, uint32_t epilogue,
                                   float p_drop, uint64_t seed, const uint8_t *mask, float *col_part, void *stream);


/* LDS window sizes of the SpMM kernel (rows of the dense operand staged per workgroup): the default used
 * when window_rows == 0, and the largest accepted value.  Tiles should be built with win_rows <= the
 * window_rows later passed to fitgnn_spmm_csr_f32 (larger windows are clamped: still correct, slower). */
int fitgnn_spmm_default_window_rows(void);
int fitgnn_spmm_max_window_rows(int32_t H);

/* gcn_norm of PyG's GCNConv (network.py:31 -> GCNConv.forward): the CSR holds, per TARGET row i, the
 * incoming edges j->i INCLUDING the self loops (added by the caller as add_remaining_self_loops does).
 *   deg[i] = sum_e w[e];  dinv = deg^-1/2 (0 where deg == 0);  val[e] = dinv[i] * w[e] * dinv[col[e]]
 * w == NULL means all-ones.  dinv: f32[n_rows] scratch/output. */
int fitgnn_gcn_norm_csr_f32(const int32_t *rowptr, const int32_t *col, const float *w, float *val, float *dinv,
                            int32_t n_rows, void *stream);

/* epilogue flags of fitgnn_spmm_csr_f32 */
#define FITGNN_EPI_BIAS 1u    /* + bias[h]                                   (GCNConv bias) */
#define FITGNN_EPI_ELU 2u     /* ELU(alpha=1)                                 (network.py:32 F.elu) */
#define FITGNN_EPI_DROPOUT 4u /* inverted dropout with keep-prob 1-p          (network.py:33 F.dropout) */
/* the `seed` argument is a DEVICE pointer to the uint64 seed (read by the kernel), not the seed itself: a step
 * captured in a hipGraph keeps drawing fresh dropout patterns, its seeds are advanced by a kernel of the graph */
#define FITGNN_EPI_SEED_DEVICE 8u
/* internal to the *_dz_* entry points below (the store epilogue is the DERIVATIVE of ELU / dropout); rejected elsewhere */
#define FITGNN_EPI_BACKWARD 0x10u
/* kernel-variant hint carried in the same word: skip the LDS window, gather operand rows straight from
 * L2/HBM -- faster when rows hold only a few non-zeros (identical results) */
#define FITGNN_SPMM_GATHER 0x100u

/* Y[n_rows x H] = epilogue( A . X ),  A in CSR (int32 indices, f32 values), X,Y row-major f32 with leading
 * dimensions ldx,ldy (elements).  This is GCNConv.propagate (gather + scatter-add over edge_index) done as a
 * segmented reduction per target row; the backward pass is the same call on the transposed CSR.
 * lcol / win_cols: outputs of fitgnn_plan_tiles_host (device copies).  Both NULL = every tile has a contiguous
 * window and `col` is used as is.
 * xrow (may be NULL): operand row r of the pattern is read from X[xrow[r]] instead of X[r] -- lets the many union rows
 * that are copies of one original node (extra nodes of the subgraphs, utils.py:235-239) share a single row of a
 * de-duplicated operand table.  xrow_zero_from (with xrow; -1: none): table rows with index >= xrow_zero_from are rows of zeros
 * (the tail of a compact operand) and are not loaded; the direct-gather variant is then not used.
 * window_rows: LDS rows per workgroup (0 = default); see fitgnn_spmm_default_window_rows().
 * Dropout: element (row,h) is kept iff mask[row*H+h] != 0 when `mask` is given, else iff 16 bits of a
 * counter-based hash of (seed, (row*H+h)/4) are >= floor(p*65536); kept values are scaled by 1/(1-p).  16-byte aligned X/Y rows (H%4==0 and
 * ld%4==0) take the vector path; anything else takes the scalar path. */
int fitgnn_spmm_csr_f32(const int32_t *rowptr, const int32_t *col, const float *val, const float *X, int64_t ldx,
                        float *Y, int64_t ldy, int32_t n_rows, int32_t H, const fitgnn_tile_t *tiles,
                        int32_t n_tiles, const int32_t *lcol, const int32_t *win_cols, const int32_t *xrow,
                        int32_t xrow_zero_from, int32_t window_rows, const float *bias, uint32_t epilogue, float p_drop,
                        uint64_t seed, const uint8_t *mask, void *stream);

/* out[s] = sum of X[members[m]] for m in [seg_off[s], seg_off[s+1]) in that order (f32, fixed order: reproducible).
 * The adjoint of the row indirection above: gradients of duplicated union rows summed back per original node. */
int fitgnn_segment_sum_f32(const int32_t *seg_off, const int32_t *members, int32_t n_seg, const float *X, int64_t ldx,
                           int32_t F, float *out, int64_t ldo, void *stream);

/* Graph-level pooling over SORTED segments (torch_geometric.nn.global_mean_pool / global_max_pool as called from network.py:93,131,
 * 164,202; a batch's rows are grouped by graph).  members (int32, may be NULL = the identity) lists the pooled rows -- the row mask
 * x[mask] of the *_gs models (network.py:129,200) is folded into the pool: no gathered copy.
 *   mean / sum: fitgnn_segment_sum_f32 above (+ a per-graph scale);
 *   max: out[s][c] = max_m X[members[m]][c], arg[s][c] = that row (first on a tie; -inf / -1 for an empty segment);
 *   backward of mean / sum: fitgnn_segment_expand_f32 writes EVERY row of dst [n_rows x F]: scale[seg] * src[seg] for seg =
 *     seg_of_row[r] >= 0 and zeros otherwise (F % 4 == 0);
 *   backward of max: fitgnn_segment_max_bwd_f32 stores g[s][c] at dst[arg[s][c]][c] (dst zeroed by the caller; segments disjoint). */
int fitgnn_segment_max_f32(const int32_t *seg_off, const int32_t *members, int32_t n_seg, const float *X, int64_t ldx, int32_t F,
                           float *out, int32_t *arg, void *stream);
int fitgnn_segment_max_bwd_f32(const float *g, const int32_t *arg, int32_t n_seg, int32_t F, float *dst, int64_t ldd, void *stream);
int fitgnn_segment_expand_f32(const float *src, const int32_t *seg_of_row, const float *scale, int64_t n_rows, int32_t F, float *dst,
                              void *stream);

/* lt1(global_mean_pool(x[rows])) in one launch (Regress_graph_gs / _gc, network.py:164-166, :200-204): graph s pools the rows
 * members[seg_off[s] .. seg_off[s+1]) of X (row stride ldx, F columns), scaled by inv_cnt[s]; pooled [n_seg x F] (contiguous) keeps
 * the pooled rows for the backward, y [n_seg x C] = pooled W^T + b (W [C x F] contiguous, b may be NULL).  One workgroup per graph,
 * fixed summation order.  fitgnn_pool_head_supported(F, C): F % 4 == 0, F / 4 divides 256, C <= 8.
 * Backward: dx [n_rows x F] (contiguous; EVERY row written: dx[r] = inv_cnt[s] sum_c dy[s][c] W[c] for seg_of_row[r] = s >= 0, zeros
 * otherwise), dW [C x F] = dy^T pooled and db [C] = column sums of dy (either may be NULL), sums over ascending s. */
int fitgnn_pool_head_supported(int32_t F, int32_t C);
int fitgnn_pool_head_f32(const int32_t *seg_off, const int32_t *members, int32_t n_seg, const float *X, int64_t ldx, int32_t F,
                         const float *inv_cnt, const float *W, const float *b, int32_t C, float *pooled, float *y, void *stream);
int fitgnn_pool_head_bwd_f32(const float *dy, const float *W, int32_t C, const float *pooled, const int32_t *seg_of_row,
                             const float *inv_cnt, int64_t n_rows, int32_t n_seg, int32_t F, float *dx, float *dW, float *db,
                             void *stream);

/* out[W] = sum over b < B of part[b][W] in a fixed order (W % 4 == 0, 16-byte aligned): combines the partial products
 * of the split-K weight-gradient GEMM dH^T @ X (the library has no deterministic split-K for K = number of rows). */
int fitgnn_sum_leading_f32(const float *part, int32_t B, int64_t W, float *out, void *stream);

/* out [M x N] = a^T @ b for tall row-major operands a [R x M] (row stride lda), b [R x N] (row stride ldb): the
 * weight-gradient product grad_W = grad_h^T @ x that `loss.backward()` (run.py:207, :246) runs for every Linear on the
 * path (GCNConv.lin, lt1; network.py:31-33).  Hand-written split-K MFMA kernel: each fp32 operand is split into two
 * bf16 (hi by truncation, lo = bf16(x - hi)) and the product is hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16 with
 * fp32 accumulation (~5e-6 relative error against fp64); the row chunks' partial tiles go to `workspace`
 * (fitgnn_gemm_atb_workspace_bytes) and are added in a fixed order -- reproducible, no atomics.
 * M, N, lda, ldb multiples of 4; all pointers 16-byte aligned. */
size_t fitgnn_gemm_atb_workspace_bytes(int64_t R, int32_t M, int32_t N);
int fitgnn_gemm_atb_f32(const float *a, int64_t lda, const float *b, int64_t ldb, int64_t R, int32_t M, int32_t N,
                        float *out, void *workspace, void *stream);

/* c [R x N] (row stride ldc) = a [R x K] @ b [N x K]^T, row-major fp32, K a multiple of 32, lda/ldb multiples of 4,
 * a and b 16-byte aligned: the forward of GCNConv's bias-free Linear, h = x W^T (network.py:31 via torch_geometric), and
 * with b = W^T the input gradient grad_x = grad_h @ W of its backward.  Same three-product bf16 split on the MFMA pipe
 * as fitgnn_gemm_atb_f32 (~5e-6 relative error against fp64), 256 x 256 output tiles, no split over K. */
int fitgnn_gemm_nt_f32(const float *a, int64_t lda, const float *b, int64_t ldb, int64_t R, int32_t N, int32_t K,
                       float *c, int64_t ldc, void *stream);

/* The same three products of a Linear in the REFERENCE's own arithmetic -- fp32 operands, exact fp32 products, fp32 accumulation
 * (PyG's Linear inside GCNConv, network.py:13-31, is an fp32 GEMM) -- on v_mfma_f32_32x32x2_f32 (csrc/gemm_f32.hip):
 *     c[i][j] = sum_k A(i,k) * B(j,k),  c [I x J] row-major with row stride ldc,
 *     A(i,k) = a[i * lda + k] (a_kmajor == 0: k contiguous) or a[k * lda + i] (a_kmajor != 0: k is the row index); B likewise.
 * forward h = x W^T: (0, 0);  grad_x = grad_h W: (0, 1) with b = W (no transposed copy);  grad_W = grad_h^T x: (1, 1) with
 * a = grad_h, b = x and K = the number of rows.  (1, 0) is E_BADARG (swap the operands and transpose the result).
 * A long reduction into few output tiles is split over k: the partial tiles go to `workspace`
 * (fitgnn_gemm_exact_workspace_bytes; 0 = none needed, workspace may then be NULL) and are added in a fixed order -- reproducible,
 * no atomics.  The contiguous extent of each operand (K if k-minor, I or J if k-major), lda and ldb are multiples of 4; a and b
 * are 16-byte aligned; any I, J, K otherwise (ragged tiles are masked, k beyond K reads as zero). */
size_t fitgnn_gemm_exact_workspace_bytes(int64_t I, int32_t J, int64_t K, int32_t a_kmajor, int32_t b_kmajor);
int fitgnn_gemm_exact_f32(const float *a, int64_t lda, int32_t a_kmajor, const float *b, int64_t ldb, int32_t b_kmajor,
                          int64_t I, int32_t J, int64_t K, float *c, int64_t ldc, void *workspace, void *stream);

/* Pre-split b operand for the tall-GEMM kernels: b [N x K] given by element strides (b[n * stride_n + k * stride_k]; so
 * b = W^T needs no transposed copy) is converted ONCE per call into bf16 hi/lo fragments laid out as the kernel's LDS image,
 * per (256-column tile, 32-wide k stage); b has K_valid <= K columns, the image is zero from there to K (a multiple of 32:
 * a feature width like 8 415 runs against an `a` operand zero-padded to 8 448 columns).  fitgnn_gemm_nt_pre_f32 then stages that side by LDS-DMA (no registers, no
 * conversion in the loop).  fitgnn_gemm_nt_epilogue_bwd_f32 takes such an image as `b` when ldb == 0. */
size_t fitgnn_gemm_nt_presplit_bytes(int32_t N, int32_t K);
int fitgnn_gemm_nt_presplit_f32(const float *b, int64_t stride_n, int64_t stride_k, int32_t N, int32_t K, int32_t K_valid,
                                void *image, void *stream);
int fitgnn_gemm_nt_pre_f32(const float *a, int64_t lda, const void *b_image, int64_t R, int32_t N, int32_t K, float *c,
                           int64_t ldc, void *stream);

/* The same product taken as the gradient dOut of a fused layer output out = dropout(ELU(z)) (network.py:32-33), with
 * fitgnn_epilogue_bwd_f32's transformation applied to the accumulators before they are stored:
 *   dZ [R x N] = keep ? (a @ b^T) / (1 - p) * (o > 0 ? 1 : o + 1) : 0,  o = out * (1 - p);  db[n] = sum_rows dZ (may be NULL).
 * dOut itself is never written.  out, dZ and mask are contiguous [R x N]; flags/seed/mask as in the forward. */
size_t fitgnn_gemm_nt_epilogue_bwd_workspace_bytes(int64_t R, int32_t N);
int fitgnn_gemm_nt_epilogue_bwd_f32(const float *a, int64_t lda, const float *b, int64_t ldb, int64_t R, int32_t N, int32_t K,
                                    const float *out, float *dZ, uint32_t epilogue, float p_drop, uint64_t seed,
                                    const uint8_t *mask, float *db, void *work, size_t work_bytes, void *stream);

/* The output head lt1 (network.py:34) on selected rows only: y[rows[i]][c] = sum_h out[rows[i]][h] * Wl[c][h] + bl[c]
 * (bl may be NULL) for i < n_rows, accumulated in ascending h; the other rows of y are not touched.  For the train step, whose
 * loss keeps out[mask] (run.py:193-204): one pass over the kept rows instead of a [R x H] @ [H x C] product over all of them.
 * Wl [C x H] contiguous is held in LDS: fitgnn_head_rows_lds_bytes(H, C) must not exceed 160 KiB (else E_BADARG).
 * out_compact != 0: `out` holds the selected rows only (row i of out is row rows[i] of y). */
size_t fitgnn_head_rows_lds_bytes(int32_t H, int32_t C);
int fitgnn_head_rows_f32(const float *out, int64_t ldo, const int64_t *rows, int32_t n_rows, const float *Wl, const float *bl,
                         int32_t C, int32_t H, float *y, int64_t ldy, int32_t out_compact, void *stream);

/* out[c] = sum over rows of x[row][c] for a tall matrix with C <= 64 columns (row stride ldx), fixed order: the bias gradient
 * of the output head lt1 (network.py:34), grad_b = sum_rows grad_y.  torch's dim-0 reduction of such a matrix takes 19-50 us. */
size_t fitgnn_colsum_narrow_workspace_bytes(int32_t n_rows, int32_t C);
int fitgnn_colsum_narrow_f32(const float *x, int64_t ldx, int32_t n_rows, int32_t C, float *out, void *work, size_t work_bytes,
                             void *stream);

/* out[h] = sum over c < n_chunks of partial[c][h] in a fixed order (the second pass of the bias-gradient reductions). */
int fitgnn_colsum_partials_f32(const float *partial, int32_t n_chunks, int32_t H, float *out, void *stream);

/* loss[0] = scale * sum_t NLL(log_softmax(z[idx[t]]), labels[t]) over n selected rows (Classify_node's log_softmax,
 * network.py:35, followed by NLLLoss, run.py:341; scale = 1/n for reduction='mean', 1/global count under data
 * parallelism), and dz [n_rows x ldz] = its gradient w.r.t. the logits z (zero on rows that are not selected).
 * One pass over the selected rows instead of log_softmax + gather + nll_loss and their three backward kernels. */
/* loss[0] = scale * sum_i |out[i] - tgt[i]| and grad[i] = scale * sign(out[i] - tgt[i]) over n contiguous values: L1Loss of the
 * regression tasks (run.py:518,716; scale = 1/n for reduction='mean') with its gradient, one launch (a fixed-order sum). */
int fitgnn_l1_loss_f32(const float *out, const float *tgt, int32_t n, float scale, float *loss, float *grad, void *stream);

size_t fitgnn_softmax_nll_workspace_bytes(int32_t n);
int fitgnn_softmax_nll_f32(const float *z, int64_t ldz, int32_t n_rows, int32_t C, const int64_t *idx,
                           const int64_t *labels, int32_t n, float scale, float *loss, float *dz, void *work,
                           size_t work_bytes, void *stream);

/* One Adam update (torch.optim.Adam semantics: L2 weight decay folded into the gradient, bias-corrected moments;
 * run.py:344 Adam(lr, weight_decay=5e-4)) over a FLAT parameter buffer of n floats (n % 4 == 0, 16-byte aligned) and its
 * equally laid out gradient / moment buffers.  step: device counter of completed updates (read, then advanced). */
int fitgnn_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                         float beta1, float beta2, float eps, float weight_decay, float *step, void *stream);

/* The same update for a step whose backward wrote its weight gradients into a buffer of their own, grad_new (laid out like grad_acc;
 * NULL: none), instead of adding them to grad_acc tensor by tensor: the gradient used is grad_acc + grad_new, stored back to grad_acc,
 * and grad_new is cleared
 * (run.py:254-304 never clears the gradients inside an epoch: they accumulate over the batch steps).  state: float[2] -- the step
 * count, then a word the kernel uses as a ticket counter and leaves at zero (zero-initialise both).  The last workgroup to finish
 * advances the count and adds seed_stride (mod 2^64) to each of the n_seeds device-resident dropout seeds (seeds may be NULL with
 * n_seeds == 0): ONE launch per optimiser step of a captured step sequence. */
int fitgnn_adam_step_acc_f32(float *param, float *grad_acc, float *grad_new, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, float *state, uint64_t *seeds, int32_t n_seeds,
                             uint64_t seed_stride, void *stream);

/* Backward of the fused epilogue  out = dropout(ELU(z)):  given dOut and the forward OUTPUT `out`
 *   dZ = keep ? dOut * 1/(1-p) * (o > 0 ? 1 : o + 1) : 0,   o = out*(1-p) (pre-dropout ELU value)
 * and the bias gradient db[h] = sum_rows dZ[row][h] (deterministic two-pass reduction through `work`).
 * flags: FITGNN_EPI_ELU and/or FITGNN_EPI_DROPOUT as used in the forward; db may be NULL. */
size_t fitgnn_epilogue_bwd_workspace_bytes(int32_t n_rows, int32_t H);
int fitgnn_epilogue_bwd_f32(const float *dOut, const float *out, float *dZ, int32_t n_rows, int32_t H,
                            uint32_t epilogue, float p_drop, uint64_t seed, const uint8_t *mask, float *db,
                            void *work, size_t work_bytes, void *stream);

/* The same, with the backward of the output head lt1 (network.py:34, y = out Wl^T + bl) folded in: dOut is not
 * read but formed on the fly as dy[n_rows x C] @ Wl[C x H], 1 <= C <= fitgnn_head_max_classes().  Replaces the
 * K = num_classes GEMM dy @ Wl and the [n_rows x H] matrix it would write and this kernel would re-read.
 * dWl (f32[C x H], may be NULL) receives the head's weight gradient dy^T @ out, accumulated while `out` streams by
 * (the library GEMM for this [C x rows] @ [rows x H] shape takes 220-340 us on its own).  With dWl == NULL the head may be
 * as wide as fitgnn_head_max_classes_wide() (ogbn-products: 47 classes); its weight gradient is then a separate product. */
int fitgnn_head_max_classes(void);
int fitgnn_head_max_classes_wide(void);
/* 1 when fitgnn_epilogue_bwd_head_f32 takes a head of C classes on H hidden columns (with / without dWl): besides the two
 * limits above, C may not exceed the number of lanes that own columns of the last 256-column slab (H = 16 -> 4 classes). */
int fitgnn_epilogue_bwd_head_supported(int32_t H, int32_t C, int32_t with_dWl);
size_t fitgnn_epilogue_bwd_head_workspace_bytes(int32_t n_rows, int32_t H, int32_t C);
int fitgnn_epilogue_bwd_head_f32(const float *dy, const float *Wl, int32_t C, const float *out, float *dZ,
                                 int32_t n_rows, int32_t H, uint32_t epilogue, float p_drop, uint64_t seed,
                                 const uint8_t *mask, float *db, float *dWl, void *work, size_t work_bytes, void *stream);

/* The same over selected rows only, in compact form: for i < n_sel, ORIGINAL row rows[i] gives row i of dZc [n_sel x H]; db / dWl
 * sum over those rows.  The mask entry / dropout hash of row i is that of row rows[i].  inputs_compact == 0: dy [R x C] and
 * out [R x H] are the full matrices, read at row rows[i]; != 0: they are compact themselves ([n_sel x C], [n_sel x H], row i).
 * For a loss that keeps out[mask] (run.py:193-204) dy is zero on every other row, hence so is dZ there: the caller appends zero
 * rows to dZc and hands the backward SpMM a row indirection instead of a [R x H] matrix that is 98 % zeros.
 * Workspace: fitgnn_epilogue_bwd_head_workspace_bytes(n_sel, H, C). */
int fitgnn_epilogue_bwd_head_rows_f32(const float *dy, const float *Wl, int32_t C, const float *out, const int64_t *rows,
                                      int32_t n_sel, int32_t inputs_compact, float *dZc, int32_t H, uint32_t epilogue, float p_drop,
                                      uint64_t seed, const uint8_t *mask, float *db, float *dWl, void *work, size_t work_bytes,
                                      void *stream);

/* fitgnn_epilogue_bwd_f32 over selected rows, compact: dZc [n_sel x H] row i = dOut row . ELU' / dropout' of ORIGINAL row rows[i] (its
 * mask entry / dropout hash), db = column sums; inputs_compact != 0: dOut and out hold the selected rows only (row i), else they are the
 * full matrices read at row rows[i].  Serves a GCN layer evaluated aggregate-first on the rows its consumer reads (the pooled rows of
 * the *_graph_gs models, network.py:129,200: x[mask]).  Workspace: fitgnn_epilogue_bwd_workspace_bytes(n_sel, H). */
int fitgnn_epilogue_bwd_rows_f32(const float *dOut, const float *out, const int64_t *rows, int32_t n_sel, int32_t inputs_compact, float *dZc,
                                 int32_t H, uint32_t epilogue, float p_drop, uint64_t seed, const uint8_t *mask, float *db, void *work,
                                 size_t work_bytes, void *stream);

/* z[i] <- dropout(ELU(z[i] + bias)) for the n rows of a compact matrix z (row stride ldz), row i standing for ORIGINAL row
 * rows[i] (rows == NULL: i) whose dropout hash / mask entry it takes: the store epilogue of fitgnn_spmm_csr_f32 (same flags,
 * same arithmetic) for a layer whose dense part is evaluated on selected rows only.  H % 4 == 0, z 16-byte aligned. */
int fitgnn_epilogue_fwd_rows_f32(float *z, int64_t ldz, const int64_t *rows, int32_t n, int32_t H, const float *bias,
                                 uint32_t epilogue, float p_drop, uint64_t seed, const uint8_t *mask, void *stream);

/* A layer's dense part on a FEW input columns: out = dropout(ELU(a W^T + bias)), a [n x K] (row stride lda), W [H x K] (row
 * stride ldw: torch's Linear weight), 1 <= K <= 32, H % 4 == 0, out 16-byte aligned with ldo % 4 == 0; flags, dropout hash and
 * mask indexing as fitgnn_spmm_csr_f32's store epilogue (element (row, column) of an [n x H] matrix).  Serves GCNConv on an input
 * narrower than the layer (network.py:189-204 on QM9's 11 atom features) evaluated aggregate-first, (A_hat x) W^T, with a = A_hat x
 * formed once per batch: per step the layer is this one pass over its output.  The K products of an element are added in
 * ascending k.  fitgnn_dense_narrow_k_lds_bytes: the kernel's LDS need, 0 when (K, H) is not supported (W^T must fit 64 KiB). */
size_t fitgnn_dense_narrow_k_lds_bytes(int32_t K, int32_t H);
int fitgnn_dense_narrow_k_f32(const float *a, int64_t lda, const float *W, int64_t ldw, int32_t n, int32_t K, int32_t H,
                              const float *bias, uint32_t epilogue, float p_drop, uint64_t seed, const uint8_t *mask, float *out,
                              int64_t ldo, void *stream);

/* ... and that layer's backward in one pass over the incoming gradient d [n x H] (row stride ldd, 16-byte aligned):
 * dW [H x K] (contiguous) = dZ^T a and db [H] = column sums of dZ, a [n x K] as above.  prev == NULL: d is dZ itself.
 * prev != NULL ([n x H] contiguous, the layer's output o = dropout(ELU(z))): d is the gradient w.r.t. o and
 * dZ = fitgnn_epilogue_bwd_f32's arithmetic with the forward's flags (FITGNN_EPI_ELU / _DROPOUT), seed and mask, formed in registers
 * (the layer's input needs no gradient, so dZ is never written).  Per-block partial sums reduced in a fixed order (reproducible).
 * H / 4 must divide 256.  Workspace: fitgnn_narrow_atb_workspace_bytes(n, K, H) (0 = shape not supported). */
size_t fitgnn_narrow_atb_workspace_bytes(int32_t n, int32_t K, int32_t H);
int fitgnn_narrow_atb_f32(const float *d, int64_t ldd, const float *prev, uint32_t epilogue, float p_drop, uint64_t seed,
                          const uint8_t *mask, const float *a, int64_t lda, int32_t n, int32_t K, int32_t H, float *dW, float *db,
                          void *work, size_t work_bytes, void *stream);

/* Backward SpMM with the epilogue backward folded in: dH = A^T dZ with dZ (above) formed while the operand rows are
 * staged, never written to memory; db / dWl reduced over tiles in a fixed order.  (rowptr, col, val, tiles) describe
 * the TRANSPOSED pattern; tiles must have contiguous windows that cover their own rows (fitgnn_amd.csr.make_tiles) and
 * window_rows <= 16.  dOut == NULL selects the head form (dy, Wl, C as in fitgnn_epilogue_bwd_head_f32, C <= 4).
 * fitgnn_spmm_epilogue_bwd_supported() tells whether a shape is covered; otherwise use the two separate calls. */
int fitgnn_spmm_epilogue_bwd_supported(int32_t H, int32_t C, int32_t window_rows);
size_t fitgnn_spmm_epilogue_bwd_workspace_bytes(int32_t n_tiles, int32_t H, int32_t C);
int fitgnn_spmm_epilogue_bwd_f32(const int32_t *rowptr, const int32_t *col, const float *val, const fitgnn_tile_t *tiles,
                                 int32_t n_tiles, int32_t window_rows, const float *dOut, const float *dy, const float *Wl,
                                 int32_t C, const float *out, float *dH, int32_t n_rows, int32_t H, uint32_t epilogue,
                                 float p_drop, uint64_t seed, const uint8_t *mask, float *db, float *dWl, void *work,
                                 size_t work_bytes, void *stream);

/* ---- graph attention (torch_geometric.nn.GATConv as network.py:13 constructs it: heads = 1, negative_slope 0.2,
 * add_self_loops, bias; no attention dropout).  CSR rows = target nodes, self loops included by the caller.
 *   a_src[j] = h_j . att_src,  a_dst[i] = h_i . att_dst
 *   alpha_e  = softmax over CSR row i of LeakyReLU(a_src[col[e]] + a_dst[i])
 *   out      = fitgnn_spmm_csr_f32 with val = alpha (+ bias)
 * backward:  dalpha_e = dOut_i . h_col[e] (SDDMM);  ds_e = alpha_e (dalpha_e - sum_k alpha_k dalpha_k) LeakyReLU'(s_e);
 *            da_dst[i] = sum over row i of ds;  da_src = row sums of ds on the transposed order. */
int fitgnn_gat_scores_f32(const float *h, int64_t ldh, int32_t n, int32_t C, const float *att_src, const float *att_dst,
                          float *a_src, float *a_dst, void *stream);
int fitgnn_gat_edge_softmax_f32(const int32_t *rowptr, const int32_t *col, const float *a_src, const float *a_dst,
                                float negative_slope, int32_t n, float *alpha, void *stream);
int fitgnn_sddmm_csr_f32(const int32_t *rowptr, const int32_t *col, const float *dOut, int64_t ldo, const float *h,
                         int64_t ldh, int32_t n, int32_t C, float *dalpha, void *stream);
int fitgnn_gat_softmax_bwd_f32(const int32_t *rowptr, const int32_t *col, const float *a_src, const float *a_dst,
                               const float *alpha, const float *dalpha, float negative_slope, int32_t n, float *ds,
                               float *da_dst, void *stream);
/* The same two passes over the rows sel[0 .. n_sel) only (int64 row ids), for a gradient that is zero outside them (the last GAT layer
 * of a step whose loss keeps out[mask], run.py:193-204): dOut_c is COMPACT (row i belongs to row sel[i]); only the entries of those rows
 * are written to dalpha / ds and only their da_dst -- the caller zeroes the three arrays. */
int fitgnn_sddmm_csr_rows_f32(const int32_t *rowptr, const int32_t *col, const float *dOut_c, int64_t ldo, const float *h, int64_t ldh,
                              const int64_t *sel, int32_t n_sel, int32_t C, float *dalpha, void *stream);
int fitgnn_gat_softmax_bwd_rows_f32(const int32_t *rowptr, const int32_t *col, const float *a_src, const float *a_dst,
                                    const float *alpha, const float *dalpha, float negative_slope, const int64_t *sel, int32_t n_sel,
                                    float *ds, float *da_dst, void *stream);
int fitgnn_csr_row_sum_f32(const int32_t *rowptr, const float *v, int32_t n, float *y, void *stream);

/* Propagation on narrow signals (APPNP, Baselines/SGGC/APPNP/networks.py:11,23: z <- (1-alpha) A_hat z + alpha z0 on
 * [rows x num_classes]): Y = beta * (A X) + gamma * Z0 (Z0 may be NULL), and optionally ACC += delta * X (the backward
 * pass accumulates d z0 while it propagates).  X, Y, Z0, ACC: dense row-major [n_rows x H], leading dimension H; A is
 * square.  Meant for H well below 64, where the tiled kernel would idle most of a wavefront. */
int fitgnn_spmm_narrow_f32(const int32_t *rowptr, const int32_t *col, const float *val, const float *X, float *Y,
                           int32_t n_rows, int32_t H, float beta, const float *Z0, float gamma, float *ACC, float delta,
                           void *stream);
/* The same product on a signal whose rows are padded to whole float4s: X, Y, Z0, ACC are [n_rows x 4 * h4] (1 <= h4 <= 16, pad
 * columns zero, 16-byte aligned) -- the layout APPNP's K steps run in (ops.APPNPPropagate): a wave packs 64 / h4 consecutive rows,
 * an operand row is one contiguous access, long rows (a star's centre) are split over the wave.  Fixed summation order. */
int fitgnn_spmm_narrow_padded_f32(const int32_t *rowptr, const int32_t *col, const float *val, const float *X, float *Y,
                                  int32_t n_rows, int32_t h4, float beta, const float *Z0, float gamma, float *ACC, float delta,
                                  void *stream);

/* dst [n_rows x 4 h4] (contiguous, 16-byte aligned) = the rows src[index[r]] (index int32, NULL = the identity; H <= 4 h4 floats each, row
 * stride lds) padded with zeros: the layout fitgnn_spmm_narrow_padded_f32 / fitgnn_appnp_*_f32 run in, made in one pass from the model's
 * class-wide output -- x.index_select(0, x_index) of a de-duplicated table (Baselines/SGGC/APPNP/networks.py:21-23 runs lin2's output
 * through prop1 directly; the indirection is this library's) + zero fill + strided copy. */
int fitgnn_gather_rows_padded_f32(const float *src, int64_t lds, int32_t H, const int32_t *index, int64_t n_rows, float *dst, int32_t h4,
                                  void *stream);

/* APPNP's K propagation steps z_{k+1} = (1 - alpha) A z_k + alpha z_0 (Baselines/SGGC/APPNP/networks.py:11,23) with the signal resident
 * in LDS, for the "units" of a block-diagonal batch: units [n_units x 2] = (row_begin, row_end) of runs of consecutive rows that are
 * CLOSED under the pattern (every column of their rows lies inside the run: whole cluster subgraphs), at most
 * fitgnn_appnp_unit_rows(h4) rows (768 / h4) and fitgnn_appnp_unit_entries() CSR entries each.  X, Y: [rows x 4 h4] padded signals as in fitgnn_spmm_narrow_padded_f32 (only
 * the units' rows are read / written).  One wavefront per unit loads it once, steps it K times between two LDS buffers and stores it
 * once -- one launch instead of K passes over the signal.  backward != 0: hand in the TRANSPOSED pattern; computes
 * alpha sum_{k<K} g_k + g_K with g_0 = X, g_{k+1} = (1 - alpha) A^T g_k (the gradient w.r.t. z_0).  Rows outside the units are the
 * caller's (the per-step kernel on their sub-matrix).  max_rows / max_entries: the largest unit of this launch (its LDS is sized by them:
 * a batch of 50-row subgraphs runs five wavefronts per CU, one with a 700-row unit one). */
int fitgnn_appnp_unit_rows(int32_t h4);
int fitgnn_appnp_unit_entries(void);
int fitgnn_appnp_units_f32(const int32_t *rowptr, const int32_t *col, const float *val, const int32_t *units, int32_t n_units, int32_t max_rows,
                           int32_t max_entries, const float *X, float *Y, int32_t h4, int32_t K, float alpha, int32_t backward, void *stream);

/* The subgraphs beyond a unit: `blocks` [n_blocks][2] = closed diagonal blocks (row ranges) of at most fitgnn_appnp_block_rows() rows and
 * fitgnn_appnp_block_entries() CSR entries (max_rows / max_entries: the largest of THIS list, which sizes the launch's LDS).  One
 * workgroup per block runs all K >= 1 steps: the block's CSR slice staged in LDS, the steps ping-pong between the scratch signals T1 / T2
 * (same shape as X; only the blocks' rows are touched; they stay in L2), separated by workgroup barriers.  Forward: Y = z_K,
 * z_{k+1} = (1 - alpha) A z_k + alpha X.  backward != 0 (hand in the transposed pattern): Y = g_K + alpha * sum_{k<K} g_k,
 * g_{k+1} = (1 - alpha) A^T g_k, g_0 = X.  Row arithmetic = fitgnn_spmm_narrow_padded_f32's, bit for bit.  X, Y, T1, T2 distinct,
 * 16-byte aligned, rows of 4 * h4 floats.  Replaces K calls of APPNP.propagate per direction (Baselines/SGGC/APPNP/networks.py:30-38,
 * torch_geometric.nn.APPNP) on those rows. */
int fitgnn_appnp_block_rows(void);
int fitgnn_appnp_block_entries(void);
int fitgnn_appnp_blocks_f32(const int32_t *rowptr, const int32_t *col, const float *val, const int32_t *blocks, int32_t n_blocks, int32_t max_rows,
                            int32_t max_entries, const float *X, float *Y, float *T1, float *T2, int32_t h4, int32_t K, float alpha,
                            int32_t backward, void *stream);

/* The K steps in LDS by COLUMN SLICES, for closed row ranges (`ranges` [n_ranges][2]) of any size that fits: a workgroup of `threads`
 * (64 .. 1024) stages its range's CSR slice once (columns re-based, 16 bit), then for one slice of <= `slice` (1, 2 or 4) float4 columns
 * after the other loads the slice of its rows, steps it K times between two LDS buffers and stores it (the recurrence never mixes
 * columns).  A thread owns <= fitgnn_appnp_lds_items_per_thread() (row, slice column) items: max_rows * slice <= that * threads; a row of
 * more than 16 entries is summed by a whole wavefront.  LDS: fitgnn_appnp_lds_bytes(max_rows, max_entries, slice) <=
 * fitgnn_appnp_lds_max_bytes(); max_rows / max_entries = the largest range of THIS list (a larger one is skipped by its workgroup).
 * Semantics of X, Y, K, alpha, backward as fitgnn_appnp_units_f32 (same reference lines).  Sums of a short row in CSR order; a long
 * row's in a fixed tree: results do not depend on the launch. */
int64_t fitgnn_appnp_lds_bytes(int32_t max_rows, int32_t max_entries, int32_t slice);
int fitgnn_appnp_lds_max_bytes(void);
int fitgnn_appnp_lds_items_per_thread(void);
int fitgnn_appnp_lds_f32(const int32_t *rowptr, const int32_t *col, const float *val, const int32_t *ranges, int32_t n_ranges, int32_t max_rows,
                         int32_t max_entries, const float *X, float *Y, int32_t h4, int32_t K, float alpha, int32_t backward, int32_t threads,
                         int32_t slice, void *stream);

/* =====================================================================================
 * Coarsen half: one contraction level of variation_neighborhoods
 * replaces: graph_coarsening/coarsening_utils.py contract_variation_linear :530-650,
 *           get_coarsening_matrix :212-254, coarsen_matrix :201-205 (+ graph_utils.zero_diag :82-90),
 *           and the feature pooling C.dot(X) of utils.py:161,393,738,827
 * All f64 arithmetic follows the canonical operation order documented in DESIGN.md (no FMA).
 * ===================================================================================== */

/* Candidate family (coarsening_utils.py:571-578): set i = sorted(N(i) U {i}).
 * set_off: int32[N+1];  set_mem: int32[nnz + N] capacity. */
int fitgnn_closed_neighbourhoods(const int32_t *rowptr, const int32_t *col, int32_t N, int32_t *set_off,
                                 int32_t *set_mem, void *stream);

/* Local-variation cost of every candidate set (subgraph_cost, coarsening_utils.py:555-561):
 *   cost = || B^T L_S B ||_F / (nc-1),  B = (I - 11^T/nc) A[S,:],  L_S = diag(2 dw[S] - W_S 1) - W_S.
 * W: symmetric CSR with ascending columns (w == NULL: all ones), dw f64[N], A f64[N x K] row-major with
 * leading dimension lda, 1 <= K <= FITGNN_MAX_K.  Set s = set_mem[set_off[s] .. +set_len[s]) sorted ascending. */
int fitgnn_variation_costs_f64(const int32_t *rowptr, const int32_t *col, const double *w, const double *dw,
                               const double *A, int32_t K, int64_t lda, const int32_t *set_off,
                               const int32_t *set_len, const int32_t *set_mem, int32_t n_sets, double *cost,
                               void *stream);

/* The same over a block-diagonal batch whose components carry different numbers of spectral columns: node_K
 * int32[N] gives, per node, the K of its component (a component of N_c <= K nodes has N_c columns,
 * coarsening_utils.py:85-86); A is stored with the common leading dimension lda >= max K.  node_K == NULL: uniform K. */
int fitgnn_variation_costs_batch_f64(const int32_t *rowptr, const int32_t *col, const double *w, const double *dw,
                                     const double *A, int32_t K, int64_t lda, const int32_t *node_K,
                                     const int32_t *set_off, const int32_t *set_len, const int32_t *set_mem,
                                     int32_t n_sets, double *cost, void *stream);

/* Greedy minimum-cost disjoint selection (coarsening_utils.py:604-650) run entirely on the device:
 * candidates are visited in (cost, insertion order) order -- sortedcontainers.SortedList semantics --,
 * sets with marked members are filtered, re-costed with the same arithmetic as above and re-inserted.
 *   set_off int32[N+1], set_mem int32[set_off[N]] : the family (NOT modified; copied into the workspace)
 *   cost0 f64[N]   : initial costs (fitgnn_variation_costs_f64 output)
 *   n_reduce       : floor(r*N) computed by the caller in double, as np.floor(r * N) (:612)
 * Outputs: sel_off int32[N+1], sel_mem int32[N], sel_count int32[2] = {number of sets, number of members}. */
size_t fitgnn_greedy_select_workspace_bytes(int32_t N, int64_t total_members);
int fitgnn_greedy_select(const int32_t *rowptr, const int32_t *col, const double *w, const double *dw,
                         const double *A, int32_t K, int64_t lda, int32_t N, const int32_t *set_off,
                         const int32_t *set_mem, const double *cost0, int64_t n_reduce, int32_t *sel_off,
                         int32_t *sel_mem, int32_t *sel_count, void *work, size_t work_bytes, void *stream);

/* The same selection run independently on every connected component of a block-diagonal graph (one wavefront per
 * component; components are the contiguous node ranges [comp_off[c], comp_off[c+1]), n_comp of them).  This is the
 * reference's per-component / per-dataset-graph loop (utils.py:154-159, utils.py:386-391, main.py:370) as ONE launch:
 * 130 831 QM9 molecules, or the components of a node-level dataset.  n_reduce[c] = floor(r_cur_c * N_c) per component
 * (0 = leave the component alone).  Components whose selection would remove <= min_gain nodes are dropped from the
 * output (coarsening_utils.py:131-135 breaks before applying a level that removes <= 2 nodes: pass min_gain = 2);
 * comp_gain[c] reports the nodes removed BEFORE that filter.  Output: one combined list of sets in the format of
 * fitgnn_greedy_select (node ids are those of the whole graph), ready for fitgnn_build_assignment.  node_K as in
 * fitgnn_variation_costs_batch_f64 (may be NULL). */
size_t fitgnn_greedy_select_batch_workspace_bytes(int32_t N, int64_t total_members, int32_t n_comp);
int fitgnn_greedy_select_batch(const int32_t *rowptr, const int32_t *col, const double *w, const double *dw,
                               const double *A, int32_t K, int64_t lda, int32_t N, const int32_t *set_off,
                               const int32_t *set_mem, const double *cost0, int32_t n_comp, const int32_t *comp_off,
                               const int64_t *n_reduce, int64_t min_gain, const int32_t *node_K, int32_t *sel_off,
                               int32_t *sel_mem, int32_t *sel_count, int64_t *comp_gain, void *work, size_t work_bytes,
                               void *stream);

/* get_coarsening_matrix (:212-254) and the level mapping (:168-179) as vectors:
 *   assign[i] = row of C holding column i = rank of the cluster's minimum member among surviving rows,
 *   cval[i]   = that non-zero = 1/sqrt(|cluster|),  n_out[0] = number of clusters. */
size_t fitgnn_build_assignment_workspace_bytes(int32_t N);
int fitgnn_build_assignment(int32_t N, const int32_t *sel_off, const int32_t *sel_mem, const int32_t *sel_count,
                            int32_t *assign, double *cval, int32_t *n_out, void *work, size_t work_bytes,
                            void *stream);

/* C <- iC . C across levels (coarsening_utils.py:136): assign_tot[j] = assign_l[assign_tot[j]],
 * cval_tot[j] = cval_l[assign_tot_old[j]] * cval_tot[j]. In place on assign_tot / cval_tot (length N0). */
int fitgnn_compose_levels(int32_t N0, const int32_t *assign_l, const double *cval_l, int32_t *assign_tot,
                          double *cval_tot, void *stream);

/* Adjacency lift Wc = zero_diag(Pinv^T W Pinv), then (Wc + Wc^T)/2 (coarsening_utils.py:138-139, :201-205),
 * with SciPy's summation order (bit-identical to the reference).  Outputs a CSR with ascending
 * columns: rowptr_c int32[n+1], col_c/w_c capacity nnz(W), nnz_c int32[1]. */
size_t fitgnn_lift_adjacency_workspace_bytes(int32_t N, int64_t nnz, int32_t n);
int fitgnn_lift_adjacency(int32_t N, const int32_t *rowptr, const int32_t *col, const double *w,
                          const int32_t *assign, const double *cval, int32_t n, int32_t *rowptr_c, int32_t *col_c,
                          double *w_c, int32_t *nnz_c, void *work, size_t work_bytes, void *stream);

/* Subgraph assembly (utils.py:185-267 `subgraph`, :235-239 --extra_node): the induced edges of ALL cluster subgraphs.
 * The caller holds the membership list (one row per (cluster, member node)): key_node int64 [R] = the member nodes in KEY order
 * (clusters ascending, nodes ascending inside a cluster), cl_ptr int64 [n_clusters + 1] = the clusters' row ranges in that order;
 * the LAYOUT the union is emitted in may be any permutation of it (star by star): row_node / row_cluster int64 [R] = node and
 * cluster of layout row r, inv int64 [R] (may be NULL = identity) = layout row of key position p.  adj_ptr int64 [N + 1] / adj int64
 * [2E]: the graph's adjacency lists (CSR by source node, neighbours ascending).
 *   _count: cnt[r] = number of neighbours of row r's node that are members of its cluster;
 *   _fill : behind off = exclusive scan of cnt (int64 [R + 1]): e_src / e_dst [off[R]] = the subgraphs' directed edges as layout
 *           rows, ordered by (row, neighbour): one wavefront per row, a binary search inside the row's own cluster per neighbour. */
int fitgnn_induced_edges_count(const int64_t *adj_ptr, const int64_t *adj, const int64_t *row_node, const int64_t *row_cluster,
                               const int64_t *cl_ptr, const int64_t *key_node, int64_t n_rows, int32_t *cnt, void *stream);
int fitgnn_induced_edges_fill(const int64_t *adj_ptr, const int64_t *adj, const int64_t *row_node, const int64_t *row_cluster,
                              const int64_t *cl_ptr, const int64_t *key_node, const int64_t *inv, int64_t n_rows, const int64_t *off,
                              int64_t *e_src, int64_t *e_dst, void *stream);

/* A training batch of a graph-level dataset assembled on the device (run.py:710: shuffle=True loaders -- every epoch's batches hold
 * other graphs).  The dataset keeps, per graph g, contiguous ranges of union rows g_row_ptr[g .. g+1], CSR entries g_nnz_ptr (=
 * rowptr[g_row_ptr[g]]), row tiles g_tile_ptr (no tile spans two graphs) and pooled rows g_mem_ptr (mem: global row ids of the rows
 * that are pooled, grouped by graph; pooled: the same as one byte per row); the batch of graphs perm[step B .. step B + B), step =
 * *step_idx, is those pieces re-based behind four exclusive scans.
 * fitgnn_batch_offsets (one workgroup): off [4 x (B + 1)] int32 -- rows, entries, tiles, pooled rows; off[k][B] = the batch's totals --
 * and gid [B] = the batch's graphs; *step_idx is advanced; loss_slot / loss_sum (both or neither): *loss_sum += *loss_slot, then the
 * slot is cleared (the previous step's loss joins the epoch's sum inside the next step's first launch).  B <= 1024.
 * fitgnn_batch_gather: fills the batch's fixed-capacity buffers: CSR b_rowptr [R_cap + 1] / b_col / b_val [E_cap], b_tiles [T_cap],
 * the pool's b_members [M_cap] / b_seg_off [B + 1] / b_seg_of_row [R_cap] (graph of a pooled row, else -1) / b_inv_cnt [B], the first
 * layer's aggregated input b_ax [R_cap x K] (row stride ld_ax) gathered from ax (row stride ld_ax_g), the targets b_tgt [B x n_tgt].
 * Rows past the batch's total hold no entries and zeros in b_ax and are covered by tiles of their own (16 rows each: T_cap must
 * leave room for them); entries / pooled rows past the totals are zeros.  The caller guarantees that the totals fit the capacities
 * (it knows every graph's sizes).  Optional (NULL to skip), for a last layer evaluated on the pooled rows only (compact operands):
 * b_members64 [M_cap] = b_members as int64, b_cseg [M_cap] = the graph of compact row i (-1 past the batch's pooled rows),
 * b_pos [R_cap] = a batch row's compact position -- mem_rank[row] = its rank among its graph's pooled rows -- or, for a row that is
 * not pooled, M_cap + r % zero_rows (a zero row of a compact operand that is never loaded). */
int fitgnn_batch_offsets(const int64_t *perm, int32_t *step_idx, int32_t B, const int32_t *g_row_ptr, const int32_t *g_nnz_ptr,
                         const int32_t *g_tile_ptr, const int32_t *g_mem_ptr, int32_t *off, int32_t *gid, float *loss_slot, float *loss_sum,
                         void *stream);
int fitgnn_batch_gather(int32_t B, const int32_t *off, const int32_t *gid, const int32_t *g_row_ptr, const int32_t *g_nnz_ptr,
                        const int32_t *g_tile_ptr, const int32_t *g_mem_ptr, const int32_t *rowptr, const int32_t *col, const float *val,
                        const fitgnn_tile_t *tiles, const int32_t *mem, const uint8_t *pooled, const float *ax, int32_t ld_ax_g,
                        const float *tgt, int32_t n_tgt, int32_t K, int32_t R_cap, int32_t E_cap, int32_t T_cap, int32_t M_cap,
                        int32_t *b_rowptr, int32_t *b_col, float *b_val, fitgnn_tile_t *b_tiles, int32_t *b_members, int32_t *b_seg_off,
                        int32_t *b_seg_of_row, float *b_inv_cnt, float *b_ax, int32_t ld_ax, float *b_tgt, const int32_t *mem_rank,
                        int64_t *b_members64, int32_t *b_cseg, int32_t *b_pos, int32_t zero_rows, void *stream);

/* Feature pooling Xc = C . X (utils.py:161,393,738,827): f64 accumulation over each cluster's members in
 * ascending node order, rounded once to f32 (utils.py:738 torch.FloatTensor).  X f32[N x F] (ldx), Xc
 * f32[n x F] (ldxc).  Xc64 (f64[n x F], leading dimension F) may be NULL. */
size_t fitgnn_pool_rows_workspace_bytes(int32_t N, int32_t n);
int fitgnn_pool_rows_f32(const int32_t *assign, const double *cval, int32_t N, int32_t n, const float *X,
                         int64_t ldx, int32_t F, float *Xc, int64_t ldxc, double *Xc64, void *work,
                         size_t work_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FITGNN_HIP_H */
