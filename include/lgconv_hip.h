/*
 * lgconv_hip.h -- C ABI of the MI355X (gfx950) LightGCN propagation library.
 *
 * This is the drop-in boundary for the hot path of happykygo/GNN-eCommerce
 * (BASELINE.json: north_star).  The reference has no FFI of its own: the path is reached
 * through two Python call surfaces.  Each entry point below names
 * the reference interface it stands in for (paths relative to the reference tree); the
 * ctypes stubs a maintainer would add are in INTEGRATION.md and are what
 * gnn-ecommerce_amd/_native.py contains.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it says host
 *   - all buffers are owned by the caller (PyTorch's caching allocator in practice); the
 *     library borrows them for the duration of the launches it enqueues, keeps no global
 *     state and allocates nothing
 *   - every function enqueues on `stream` (a hipStream_t passed as void*) and returns
 *     without synchronising
 *   - return value: 0 = ok; >0 = hipError_t from the runtime; <0 = LGC_E_* argument error
 *   - `status` words are device int32 the kernels OR error bits into (LGC_ST_*); the
 *     caller zeroes them and reads them back when it wants to know
 *   - values fp32, indices int32 inside the library; the reference's int64 COO is converted
 *     by lgc_build_csr
 */
#ifndef LGCONV_HIP_H
#define LGCONV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LGC_ABI_VERSION 12

/* argument errors (negative return values) */
#define LGC_E_INVAL      (-1)  /* null pointer, negative size, bad flag                    */
#define LGC_E_DIM        (-2)  /* embedding width outside 1..256 (see lgc_dim_ok)           */
#define LGC_E_WORKSPACE  (-3)  /* workspace smaller than lgc_build_workspace_bytes()        */
#define LGC_E_RANGE      (-4)  /* node or edge count does not fit int32                     */
#define LGC_E_ALIGN      (-5)  /* pointer / stride alignment requirement violated           */

/* bits OR-ed into device status words */
#define LGC_ST_INDEX_OOB 1     /* an index was outside [0, n_nodes); that element was skipped */

/* One adjacency entry as the kernels read it: 8 bytes, {source column, fp32 value}. */
typedef struct lgc_entry {
    int32_t col;
    float   val;
} lgc_entry;

/* One unit of work for rows longer than `short_max` (see lgc_spmm). */
typedef struct lgc_chunk {
    int32_t row;    /* output row                                                   */
    int32_t begin;  /* first entry (index into entries[])                            */
    int32_t end;    /* one past the last entry                                       */
    int32_t slot;   /* >=0: write the raw sum to partials[slot]; -1: finish the row  */
} lgc_chunk;

/* A row that was cut into several chunks: its partial sums sit in slots [begin, end). */
typedef struct lgc_multi_row {
    int32_t row;
    int32_t slot_begin;
    int32_t slot_end;
    int32_t reserved;
} lgc_multi_row;

int lgc_abi_version(void);

/* Human-readable text for a negative return code (host pointer, static storage). */
const char *lgc_error_string(int code);

/* 1 if lgc_spmm / lgc_pair_dot accept this embedding width. */
int lgc_dim_ok(int32_t dim);

/* ---------------------------------------------------------------------------------------
 * Graph build: dense COO -> CSR with normalised values.
 *
 * Replaces what PyG's LGConv.forward -> gcn_norm recomputes on every layer of every call
 * (called from src/lightgcn.py:96; COO produced by src/utils_v2.py:146-165, copy at
 * torchserve/lightgcn_handler.py:112-131).
 *
 *   edge_index  int64 [2, n_edges] contiguous: row 0 = source j, row 1 = target i
 *   edge_weight fp32 [n_edges] or NULL (= all ones, as upstream)
 *   by_source   0: rows of the CSR are targets, columns sources  (the forward operator A)
 *               1: rows are sources, columns targets              (A^T, for the backward pass)
 *   normalize   1: val = dis[j] * w * dis[i], dis = deg^-1/2 (inf -> 0), deg = weighted
 *                  in-degree by target accumulated SEQUENTIALLY IN EDGE ORDER in fp32 --
 *                  the order the reference's CPU scatter uses
 *               0: val = w
 *   dis_in      normalize=1 only: NULL -> compute deg/dis here (requires by_source=0) and
 *               write them to deg_out/dis_out; non-NULL -> use these (the A^T build)
 *   rowptr      out int32 [n_nodes + 1]
 *   entries     out lgc_entry [n_edges]; inside a row the entries keep the edge order
 *   edge_val    out fp32 [n_edges] in ORIGINAL edge order, or NULL
 *   deg_out, dis_out  out fp32 [n_nodes] or NULL
 * ------------------------------------------------------------------------------------- */
size_t lgc_build_workspace_bytes(int64_t n_nodes, int64_t n_edges);

int lgc_build_csr(const int64_t *edge_index, const float *edge_weight,
                  int64_t n_nodes, int64_t n_edges,
                  int32_t by_source, int32_t normalize,
                  const float *dis_in,
                  int32_t *rowptr, lgc_entry *entries,
                  float *edge_val, float *deg_out, float *dis_out,
                  void *workspace, size_t workspace_bytes,
                  int32_t *status, void *stream);

/* The user|item split of an edge list in the layout of src/utils_v2.py:146-165 (ids below n_users are users): out2[0] =
 * max over edges of min(source, target), out2[1] = min over edges of max(source, target) (int64 [2], device).  The graph is
 * user|item with split = out2[0] + 1 iff out2[0] < out2[1].  One pass over the COO. */
int lgc_bipartite_split(const int64_t *edge_index, int64_t n_edges, int64_t *out2, void *stream);

/* The work list of lgc_spmm's long rows, built on the device: every row of [row_begin, row_end) with more than `short_max`
 * entries is cut into ceil(deg / chunk_len) near-equal chunks, listed in row order; a row with several chunks gets
 * consecutive partial-sum slots and an lgc_multi_row.  lgc_row_plan_count leaves per-row prefix sums in `workspace`
 * (>= lgc_row_plan_workspace_bytes(n_rows)) and writes totals[3] = {chunks, multi-chunk rows, slots} (int32, device);
 * the caller sizes `chunks` and `multi` from them and calls lgc_row_plan_fill with the same arguments and workspace. */
size_t lgc_row_plan_workspace_bytes(int64_t n_rows);
int lgc_row_plan_count(const int32_t *rowptr, int32_t row_begin, int32_t row_end, int32_t short_max, int32_t chunk_len,
                       void *workspace, size_t workspace_bytes, int32_t *totals, void *stream);
int lgc_row_plan_fill(const int32_t *rowptr, int32_t row_begin, int32_t row_end, int32_t short_max, int32_t chunk_len,
                      const void *workspace, lgc_chunk *chunks, lgc_multi_row *multi, void *stream);

/* ---------------------------------------------------------------------------------------
 * Tiled rows: rows with at most 32 entries, listed in a caller-chosen processing order.
 *
 * Same arithmetic as lgc_spmm's short rows -- the gather -> scale -> scatter-add of one PyG LGConv layer,
 * called from src/lightgcn.py:96 (and the epilogue of src/lightgcn.py:93,97) -- for the rows of `order`,
 * entries in edge order, products rounded before the add.  lgc_build_tiles copies those rows into the tile
 * layout: 1 KiB pieces (2 KiB for width 32) that one coalesced access per wavefront fetches, holding
 * 128 / width (width 32: 8) whole rows each, padding entries col = -1 -- the hop reads no row pointer.
 *   order    int32 [n_slots]: row ids, -1 = empty slot; n_slots a multiple of the rows per tile R
 *            (16 for width 8, 8 for width 16 and 32); every listed row must have at most `width` entries.
 *            Slot g * (R/4) + bt of a tile is processed by lane group g in batch bt.
 *   meta     int32 [n_tiles]: byte bt = the largest entry count among the four rows of batch bt (upper bounds
 *            are allowed), or NULL.  With meta, a 61..64- or 68..128-wide table and tables below 4 GiB the kernel takes
 *            the path without divergent control flow (DPP broadcasts, buffer addressing); otherwise a generic one.
 *   width    8, 16 or 32 entries per row
 *   y[row] = a * sum_k val_k * x[col_k] + b * r[row]   (r may be NULL)
 */
int lgc_build_tiles(const int32_t *rowptr, const lgc_entry *entries, const int32_t *order, int64_t n_slots,
                    int32_t width, lgc_entry *slab, void *stream);

/* The processing order itself, on the device (what the host code otherwise derives with two dozen index operations):
 * lgc_tile_classes sorts the rows of [row_begin, row_end) by (width class, key, row id) -- class 0 / 1 / 2 = at most
 * min(8 | 16 | 32, max_len) entries (a row goes to the narrowest class that holds it; 0 entries -> class 0), class 3 = every
 * longer row; key = (popularity of the row's least-gathered column, that column) when `cold`, else nothing (row order) --
 * into sorted_rows (int32 [n_rows]) and counts the classes (class_count uint64 [4], device).  lgc_tile_pack turns one
 * class's slice of sorted_rows into `order` (int32 [n_tiles * R], -1 padded; inside a tile the longest rows first, rank
 * rho in slot (rho % 4) * R / 4 + rho / 4) and `meta` (int32 [n_tiles], byte bt = longest row of batch bt), n_tiles =
 * ceil(n_rows / R), R = 16 (width 8) or 8.  table_rows bounds the column ids; workspace >= the _workspace_bytes answer.
 * One-time planning calls, not hop launches: with `cold`, lgc_tile_classes reads the entry range of the rows back and
 * synchronises `stream` once (like lgc_sweep_plan_upload, lgc_sweep_dplan_create and lgc_sweep_dplan_fill, which say so). */
size_t lgc_tile_classes_workspace_bytes(int64_t n_rows, int64_t table_rows);
int lgc_tile_classes(const int32_t *rowptr, const lgc_entry *entries, int32_t row_begin, int32_t row_end, int32_t max_len,
                     int32_t cold, int64_t table_rows, void *workspace, size_t workspace_bytes, int32_t *sorted_rows,
                     uint64_t *class_count, void *stream);
int lgc_tile_pack(const int32_t *rowptr, const int32_t *sorted_rows, int64_t n_rows, int32_t width, int32_t *order, int32_t *meta,
                  void *stream);

int lgc_spmm_tiles(const int32_t *order, const int32_t *meta, const lgc_entry *slab, int32_t n_tiles, int32_t width,
                   int32_t tiles_per_wave, int64_t table_rows, const float *x, int64_t x_stride,
                   float *y, int64_t y_stride, const float *r, int64_t r_stride, float a, float b, int32_t dim,
                   void *stream);

/* ---------------------------------------------------------------------------------------
 * One propagation hop:  y[row] = a * sum_k entries[k].val * x[entries[k].col] + b * r[row]
 * for row in [row_begin, row_end) and for the rows named by `chunks`.
 *
 * Replaces one LGConv.forward (src/lightgcn.py:96) and, through a/b/r, the layer sum of
 * src/lightgcn.py:93,97 folded into the epilogue (r may be NULL: then y = a * sum).
 *
 *   rows with at most `short_max` entries in [row_begin,row_end) are done by one lane group
 *   each, accumulating in entry order; longer rows there are skipped and must appear in
 *   `chunks` (one wavefront per chunk).  Chunks with slot >= 0 write raw sums to
 *   partials[slot * dim ...]; `multi` rows then add their slots in order and apply the
 *   epilogue, which makes the result independent of scheduling.
 *
 *   table_rows  number of rows of the x / y / r tables (every row and column index is below it)
 *   x, y, r     fp32, row strides in floats (>= dim); any dword-aligned rows are accepted, 16-byte
 *               aligned rows (dim % 4 == 0) are the fast case;  y must not alias x
 *   partials    fp32 [n_slots, dim] or NULL when no chunk has slot >= 0
 * ------------------------------------------------------------------------------------- */
int lgc_spmm(const int32_t *rowptr, const lgc_entry *entries,
             int32_t row_begin, int32_t row_end, int32_t short_max,
             const lgc_chunk *chunks, int32_t n_chunks,
             const lgc_multi_row *multi, int32_t n_multi, float *partials,
             int64_t table_rows,
             const float *x, int64_t x_stride,
             float *y, int64_t y_stride,
             const float *r, int64_t r_stride,
             float a, float b, int32_t dim, void *stream);

/* The hop of lgc_spmm for a short LIST of rows:  y[row] = a * sum_k val_k * x[col_k] + b * r[row]  for row in row_ids
 * (int64 [n_ids], device; ids outside [row_begin, row_end) are skipped, repeats are harmless), every other row of y left
 * untouched.  One wavefront per listed row; rows of up to 32 entries are summed in entry order (the bits of lgc_spmm /
 * lgc_spmm_tiles), longer ones strided over the lane groups.  What it is for: `LightGCN.forward` in a training step scores
 * 2B label pairs (src/lightgcn.py:123-125, src/train_lightgcn.py:138), so of the last user step's output only the rows
 * of the batch's users are ever read -- a few thousand gathers instead of the whole 1.6 M-row step. */
int lgc_spmm_rows(const int32_t *rowptr, const lgc_entry *entries, int32_t row_begin, int32_t row_end, const int64_t *row_ids,
                  int64_t n_ids, int64_t table_rows, const float *x, int64_t x_stride, float *y, int64_t y_stride, const float *r,
                  int64_t r_stride, float a, float b, int32_t dim, void *stream);

/* The same for a list that names LONG rows (ABI 12).  What it is for: the scores of a training step read the propagated
 * table at the batch's 2B item rows too (src/lightgcn.py:123-125: `out[edge_label_index[1]]`), and nothing else of the last
 * item step's [n_items, D] block is ever read -- but an item row holds 186 entries on average and a hub beyond 10^5, which
 * one wavefront per row cannot balance.  The listed rows are cut into chunks ON THE DEVICE (no host round trip, fixed launch
 * shapes: the step can be recorded as a HIP graph): a one-workgroup planning launch (entries of the listed rows -> chunk
 * length: 256 entries, or longer when `partial_rows` would not hold that many chunks -> first chunk number of every list
 * position), a fixed grid of wavefronts striding over the chunk numbers (rows of up to 32 entries finish in entry order
 * with the bits of lgc_spmm_tiles; longer single-chunk rows finish directly; the others leave partial rows), and one
 * wavefront per cut position adding its partial rows in chunk order: deterministic, no float atomics.
 *   y_rows        rows of y (compact: >= n_ids; else > every listed id of the operator half, i.e. >= row_end)
 *   compact       0: y[row_ids[m]] is written (as lgc_spmm_rows); 1: y[m] is written -- a [n_ids, dim] table in list order
 *                 (positions whose id lies outside [row_begin, row_end) are left untouched), e.g. the block a rank of a
 *                 partition all-reduces instead of the whole item block
 *   work          int32 [n_ids + 2] device scratch
 *   partials      fp32 [partial_rows, dim] device scratch, partial_rows > n_ids (LGC_E_RANGE otherwise); n_ids + 16384
 *                 keeps 256-entry chunks up to 4 M listed entries
 * r (optional epilogue rows) is indexed by row id in both modes.  Three launches on `stream`. */
int lgc_spmm_rows_split(const int32_t *rowptr, const lgc_entry *entries, int32_t row_begin, int32_t row_end,
                        const int64_t *row_ids, int64_t n_ids, int64_t table_rows, const float *x, int64_t x_stride, float *y,
                        int64_t y_stride, int64_t y_rows, const float *r, int64_t r_stride, float a, float b, int32_t dim,
                        int32_t compact, int32_t *work, float *partials, int64_t partial_rows, void *stream);

/* ---------------------------------------------------------------------------------------
 * Band sweep: long rows over a gathered table far larger than the caches (the item step of a user|item graph).
 *
 * Same hop as lgc_spmm's chunked rows (one LGConv layer, src/lightgcn.py:96, epilogue of :93,97), organised so that a
 * row of the gathered table crosses the memory fabric about once per pass instead of once per use: the columns
 * [col_lo, col_hi) are cut into n_bands ranges (one per XCD), every (row, band) pair gets an accumulator in LDS, and
 * all wavefronts of a band walk its columns in ascending order together.  Sums are taken per piece in column order
 * and per row in band order: deterministic, different association from lgc_spmm (inside the 1e-5 gate).
 *
 * The plan is built on the HOST from host copies of the CSR (the only entry points that take host pointers):
 *   lgc_sweep_plan_create  rows [row_begin, row_end) of the CSR; returns NULL and sets *code on error
 *   lgc_sweep_plan_dims    sizes of the arrays below
 *   lgc_sweep_plan_export  copies them into caller-provided HOST buffers:
 *       slabs uint32 [n_slabs * 64 * groups], wave_slab_ptr int32 [n_waves + 1], wave_npieces int32 [n_waves],
 *       piece_slot int32 [n_waves * row_cap], multi lgc_multi_row [n_rows]
 *   lgc_sweep_plan_free
 * The caller uploads the arrays and passes device pointers to lgc_spmm_sweep, with a scratch `partials` of
 * n_slots * dim floats.  The rows of `multi` may be split over two lists: `multi` (one lane group per row, for rows
 * with few slots) and `multi_wide` (one wavefront per row, for rows cut into many pieces); every row in exactly one.  lgc_sweep_ok says whether a table qualifies (< 2^24 - 1 rows, < 4 GiB) and with which plan: 4 (entries per step) for 61..64 columns and for 97..128 columns (the same plan run twice, columns [0, 64) and [64, dim)), 2 for 68..96 columns, 0 = no sweep.
 */
typedef struct lgc_sweep_cfg {
    int32_t n_bands;               /* 8: one band per XCD (blockIdx % 8)                               */
    int32_t waves_per_band_round;  /* 256: 32 CUs x 8 wavefronts                                       */
    int32_t row_cap;               /* 78: accumulators per wavefront (8 x 79 rows x 256 B = 158 KiB; 51 for 68..96 columns) */
    int32_t piece_cap;             /* 64: longest run of one row's entries inside one wavefront's list; raised in
                                      steps of 16 (up to 4x) while that saves a whole round                */
    int32_t lookahead;             /* 64: how far the step builder looks for an entry of another piece  */
    int32_t groups;                /* entries per step = table rows a wavefront gathers per instruction: 4 (or 0) for tables
                                      of 61..64 columns (a 16-lane group per row, 1 KiB slabs), 2 for 68..96 columns (two
                                      DPP rows per row, 512-byte slabs)                                                   */
    int32_t round_order;           /* which pieces of a band share a round.  0: every round gets the same mix (pieces dealt
                                      heaviest-first over all wavefronts of the band); 1: the heaviest pieces fill round 0,
                                      the next round 1, ... (dealt heaviest-first inside the round).  A column's table row
                                      is fetched once per round that holds a piece using it, so grouping the long pieces
                                      leaves the later rounds touching few columns: 576 -> 551 us per hop on the
                                      1.6 M x 54 k graph; 2: as 1, and the odd rounds walk their band in descending column
                                      order (a round starts where the previous one ended, on rows still in the Infinity
                                      Cache: another 2 us per hop).  Entries of a piece are then accumulated in
                                      descending column order in odd rounds (still one fixed order per plan)              */
} lgc_sweep_cfg;

typedef struct lgc_sweep_dims {
    int32_t n_bands, rounds, row_cap, piece_cap, n_rows, groups;
    int64_t n_waves, n_slabs, n_slots, n_entries, n_steps, n_padding;
} lgc_sweep_dims;

typedef struct lgc_sweep_plan lgc_sweep_plan;

lgc_sweep_plan *lgc_sweep_plan_create(const int32_t *rowptr_host, const lgc_entry *entries_host, int32_t row_begin,
                                      int32_t row_end, int32_t col_lo, int32_t col_hi, const lgc_sweep_cfg *cfg,
                                      int *code);
int lgc_sweep_plan_dims(const lgc_sweep_plan *plan, lgc_sweep_dims *dims);
int lgc_sweep_plan_export(const lgc_sweep_plan *plan, uint32_t *slabs, int32_t *wave_slab_ptr, int32_t *wave_npieces,
                          int32_t *piece_slot, lgc_multi_row *multi);
/* The four big arrays straight into caller-provided DEVICE buffers of the sizes lgc_sweep_plan_dims reports (slabs,
 * wave_slab_ptr, wave_npieces, piece_slot; `multi` stays with lgc_sweep_plan_export: the host splits it by slot count);
 * synchronises `stream` before returning, so the plan may be freed right away.  Spares a host copy of ~90 MB. */
int lgc_sweep_plan_export_multi(const lgc_sweep_plan *plan, lgc_multi_row *multi);      /* HOST buffer, [n_rows] */
int lgc_sweep_plan_upload(const lgc_sweep_plan *plan, uint32_t *slabs, int32_t *wave_slab_ptr, int32_t *wave_npieces,
                          int32_t *piece_slot, void *stream);
void lgc_sweep_plan_free(lgc_sweep_plan *plan);

/* The same plan -- bit for bit -- with the bulk of the work on the device (rows [row_begin, row_end) of a CSR that is
 * ALREADY on the device; first_entry / n_entries = rowptr[row_begin] and the entry count of the range, which the caller
 * knows): column histogram, every row sorted by column (one radix sort), the run length of every (row, band); on the host
 * only the piece list and the deal over the wavefronts (from the run lengths, ~2 MB); back on the device every
 * wavefront's merged column-sorted list (a second radix sort) and the conflict-free step builder, one wavefront per list
 * (its look-ahead window is the wavefront's 64 lanes).  The 81 MB of entries are never copied to the host and the 87 MB of
 * slabs never from it.  Needs cfg->piece_cap <= 64 and cfg->lookahead <= 64 (LGC_E_RANGE otherwise: use the host planner).
 *   lgc_sweep_dplan_create   everything up to the sizes (returns NULL and sets *code on error); `workspace` (device,
 *                            >= lgc_sweep_dplan_workspace_bytes) must stay untouched until _fill has returned
 *   lgc_sweep_dplan_dims     as lgc_sweep_plan_dims
 *   lgc_sweep_dplan_fill     writes the four arrays into caller-provided DEVICE buffers of those sizes; synchronises
 *   lgc_sweep_dplan_export_multi   `multi` into a HOST buffer [n_rows]; multi[].row are the CSR's own row ids
 *   lgc_sweep_dplan_free */
typedef struct lgc_sweep_dplan lgc_sweep_dplan;
size_t lgc_sweep_dplan_workspace_bytes(int64_t n_entries, int64_t n_rows, int64_t n_cols, const lgc_sweep_cfg *cfg);
lgc_sweep_dplan *lgc_sweep_dplan_create(const int32_t *rowptr, const lgc_entry *entries, int32_t row_begin, int32_t row_end,
                                        int64_t first_entry, int64_t n_entries, int32_t col_lo, int32_t col_hi,
                                        const lgc_sweep_cfg *cfg, void *workspace, size_t workspace_bytes, void *stream,
                                        int *code);
int lgc_sweep_dplan_dims(const lgc_sweep_dplan *plan, lgc_sweep_dims *dims);
int lgc_sweep_dplan_fill(const lgc_sweep_dplan *plan, uint32_t *slabs, int32_t *wave_slab_ptr, int32_t *wave_npieces,
                         int32_t *piece_slot, void *stream);
int lgc_sweep_dplan_export_multi(const lgc_sweep_dplan *plan, lgc_multi_row *multi);
void lgc_sweep_dplan_free(lgc_sweep_dplan *plan);

int lgc_sweep_ok(int32_t dim, int64_t table_rows, int64_t x_stride);

int lgc_spmm_sweep(const uint32_t *slabs, const int32_t *wave_slab_ptr, const int32_t *wave_npieces,
                   const int32_t *piece_slot, int64_t n_waves, int32_t row_cap, int32_t groups, const lgc_multi_row *multi,
                   int32_t n_rows, const lgc_multi_row *multi_wide, int32_t n_wide, float *partials, int64_t table_rows, const float *x, int64_t x_stride, float *y,
                   int64_t y_stride, const float *r, int64_t r_stride, float a, float b, int32_t dim, void *stream);
/* ---------------------------------------------------------------------------------------
 * One operator half as a single argument, and the per-hop exchange hook.
 *
 * lgc_operator gathers everything one `y[rows] = a * A[rows, :] x + b * r[rows]` needs -- the chunk plan of
 * lgc_spmm, the tile classes of lgc_spmm_tiles and, when the half qualifies, the band-sweep arrays -- so that a
 * host applies it with ONE call (lgc_apply picks the path exactly as the separate entry points would:
 * sweep if `sweep` is set and lgc_sweep_ok(dim, table_rows, x_stride); else chunks + tiles; tiles need dim >= 4,
 * narrower tables take lgc_spmm's row part over [row_begin, row_end)).
 *
 * lgc_hop_exchange is one LGConv layer (src/lightgcn.py:96) of a user|item graph PARTITIONED over several devices
 * (SURVEY.md 8e): item step (partial sums of all item rows from this rank's own users) -> `exchange(block, rows,
 * row_stride, dim, stream, user)` -> user step (this rank's users from the replicated, now reduced, item rows).
 * The callback must leave the SUM over ranks in `block` (y rows [exchange_row_begin, +exchange_rows), row stride
 * y_stride floats) in stream order -- e.g. ncclAllReduce(block, block, rows * row_stride, ncclFloat, ncclSum, comm,
 * stream) when y_stride == dim -- and return 0; it is the only place where the library needs the host's
 * communicator, which it never sees.  A non-zero return aborts the hop and is handed back.
 * `item_epilogue`: the exchanged block is a SUM over ranks, so the b * r term of the item rows may enter it once only:
 * pass 1 on exactly one rank (rank 0) and 0 on the others -- with the same r and b everywhere; the user step applies
 * the epilogue of this rank's own user rows on every rank.
 */
typedef struct lgc_tile_class {
    const int32_t   *order;      /* device, [n_tiles * R]            */
    const int32_t   *meta;       /* device, [n_tiles] or NULL        */
    const lgc_entry *slab;       /* device, lgc_build_tiles layout   */
    int32_t n_tiles, width;
} lgc_tile_class;

typedef struct lgc_sweep_arrays {
    const uint32_t *slabs;
    const int32_t  *wave_slab_ptr, *wave_npieces, *piece_slot;
    const lgc_multi_row *multi, *multi_wide;
    float   *partials;           /* [n_slots, dim] scratch            */
    int64_t  n_waves;
    int32_t  row_cap, n_rows, n_wide;
    int32_t  groups;             /* 4 (or 0): plan for 61..64 columns; 2: plan for 68..96 columns */
} lgc_sweep_arrays;

typedef struct lgc_operator {
    const int32_t   *rowptr;
    const lgc_entry *entries;
    const lgc_chunk *chunks;
    const lgc_multi_row *multi;
    float   *partials;           /* scratch of the chunk plan, [n_slots, dim] or NULL */
    const lgc_sweep_arrays *sweep;   /* HOST pointer or NULL */
    lgc_tile_class tiles[3];
    int32_t row_begin, row_end, short_max, n_chunks, n_multi, n_tile_classes, tiles_per_wave, reserved;
} lgc_operator;

int lgc_apply(const lgc_operator *op, int64_t table_rows, const float *x, int64_t x_stride, float *y, int64_t y_stride,
              const float *r, int64_t r_stride, float a, float b, int32_t dim, void *stream);

typedef int (*lgc_exchange_fn)(float *block, int64_t rows, int64_t row_stride, int32_t dim, void *stream, void *user);

int lgc_hop_exchange(const lgc_operator *item_op, const lgc_operator *user_op, int64_t table_rows, const float *x,
                     int64_t x_stride, float *y, int64_t y_stride, const float *r, int64_t r_stride, float a, float b,
                     int32_t dim, int32_t exchange_row_begin, int32_t exchange_rows, int32_t item_epilogue,
                     lgc_exchange_fn exchange, void *user, void *stream);

/* Fixed-order sum of runs -- how the sparse gradient of a scoring step (src/lightgcn.py:123-125 scores 2B pairs; what
 * autograd's index_select backward does with float atomics at src/train_lightgcn.py:146) is accumulated deterministically:
 * `key_sorted` int64 [n] is sorted; position t is a head when key[t] != key[t - 1]; for every head with 0 <= dest[t] < y_rows
 *   y[dest[t]] = (accumulate ? y[dest[t]] : 0) + scale * (vals[t] + vals[t + 1] + ... over the run, in that order)
 * vals fp32 [n, dim] dense.  One lane group owns a destination row: no atomics, the same bits on every run. */
int lgc_segment_sum(const int64_t *key_sorted, const int64_t *dest, const float *vals, const int32_t *vals_index, int64_t n,
                    float scale, float *y, int64_t y_stride, int64_t y_rows, int32_t dim, int32_t accumulate, void *stream);
/* vals_index (int32 [n] or NULL): the value row of sorted position t is vals[vals_index[t]] -- the permutation a sort
 * returned, so that the value table is never copied into sorted order. */

/* ---------------------------------------------------------------------------------------
 * The glue of one training step (src/train_lightgcn.py:137-147) around the propagation, as a handful of launches instead
 * of the ~150 small torch kernels the same arithmetic costs on the host side: on a rank of an 8-way partition the step was
 * bound by the host's launch rate (3.0 ms wall for 1.5 ms of kernels).
 *
 * lgc_pair_dot_rows  lgc_pair_dot that also keeps what autograd's backward of src/lightgcn.py:123-125 needs: the two
 *                    gathered rows of every pair (rows0, rows1: fp32 [n_pairs, dim] dense, or NULL) and a validity byte
 *                    (ok: uint8 [n_pairs] or NULL; 0 for an out-of-range pair, whose rows are zeros and whose score is NaN).
 * lgc_bpr_loss       `recommendation_loss(pos, neg, 0) * size` (src/lightgcn.py:262-286, src/train_lightgcn.py:141) for
 *                    scores [2 * n_triples] = [pos | neg] and its gradient: loss[0] = -sum_{mask} log sigmoid(pos - neg) /
 *                    size, grad [2 * n_triples].  mask (uint8 [n_triples] or NULL = all): the triples this caller owns --
 *                    a rank of a partition scores the triples of its own users; `size` is the GLOBAL batch size.
 * lgc_pair_seed_vals the sparse gradient of the scores with respect to the propagated table, as rows of a value table:
 *                    vals[m] = g_m * rows1[m], vals[n_pairs + m] = g_m * rows0[m], g_m = mask[m] ? grad_scores[m] *
 *                    (*grad_scale) : 0 (grad_scale: DEVICE scalar or NULL; mask uint8 [n_pairs] or NULL).
 * lgc_seed_prepare   sorts the m <= LGC_SEED_MAX node ids `rows` (int64; ids outside [0, n_nodes) count as "no row" and
 *                    come out as -1, first; two launches: every id ranked by counting, then one thread per sorted position)
 *                    and derives, per sorted position t: rows_sorted[t]; perm[t] (int32, the input
 *                    position: a stable sort); dest_item[t] = the row if t heads a run of an item row (row >= split) else
 *                    -1; dest_slot[t] = t, dest_user[t] = the row if t heads a run of a user row (row < split) else -1 --
 *                    the three destination lists of lgc_segment_sum for the item block of the seed table, the compact
 *                    table of seed users and the user rows of the result.  With col_flag / col_slot (both or neither):
 *                    col_flag[row] = 1, col_slot[row] = t for every user head -- lgc_seed_pull's column map.
 * lgc_seed_flags     col_flag[row] = value for the user rows of a sorted list (value 0 takes the flags back).
 * ------------------------------------------------------------------------------------- */
#define LGC_SEED_MAX 8192
int lgc_pair_dot_rows(const float *emb, int64_t stride, int32_t dim, int64_t n_nodes, const int64_t *idx0, const int64_t *idx1,
                      int64_t n_pairs, float *scores, float *rows0, float *rows1, uint8_t *ok, int32_t *status, void *stream);
int lgc_bpr_loss(const float *scores, const uint8_t *mask, int64_t n_triples, int64_t size, float *loss, float *grad,
                 void *stream);
/* The regulariser of src/utils_v2.py:193-211 (called at src/train_lightgcn.py:142) in one launch (ABI 12):
 *   value[0] = scale * (|w[ids0]|_F^2 + |w[ids1]|_F^2 + |w[ids2]|_F^2)      scale = decay / (2 * batch_size)
 * each norm as (sqrt(sum of squares))^2 like `init_embed[batch].norm().pow(2)`; ids int64 device lists of m0 / m1 / m2
 * entries (a list may be empty), negative ids wrap like torch's indexing.  rows_out (int64 [m0 + m1 + m2] or NULL): the ids
 * as row numbers in list order -- the rows the regulariser's gradient decay / size * w[row] goes to (lgc_segment_sum); an id
 * outside [-n_rows, n_rows) contributes nothing, comes out as -1 ("no row") and sets LGC_ST_INDEX_OOB in `status` (upstream's
 * gather raises IndexError).  One workgroup, fixed reduction order: the same bits on every run. */
int lgc_reg_rows(const float *w, int64_t stride, int32_t dim, int64_t n_rows, const int64_t *ids0, int64_t m0, const int64_t *ids1,
                 int64_t m1, const int64_t *ids2, int64_t m2, float scale, float *value, int64_t *rows_out, int32_t *status,
                 void *stream);

int lgc_pair_seed_vals(const float *grad_scores, const uint8_t *mask, const float *grad_scale, const float *rows0,
                       const float *rows1, int64_t n_pairs, int32_t dim, float *vals, void *stream);
int lgc_seed_prepare(const int64_t *rows, int64_t m, int64_t split, int64_t n_nodes, int64_t *rows_sorted, int32_t *perm,
                     int64_t *dest_item, int64_t *dest_slot, int64_t *dest_user, uint8_t *col_flag, int32_t *col_slot,
                     uint64_t *scratch /* [m], device */, void *stream);
int lgc_seed_flags(const int64_t *rows_sorted, int64_t m, int64_t split, uint8_t *col_flag, int32_t value, void *stream);

/* Seeded pull (first hop of the backward pass, loss.backward() at src/train_lightgcn.py:146): the incoming gradient has
 * non-zero rows only at a few thousand seed columns, given as a compact table.  The hop of lgc_spmm over rows
 * [row_begin, row_end) + chunks, in which an entry counts only if col_flag[col] != 0 and then gathers row col_slot[col] of
 * `seed_vals` (fp32 [n_seed, seed_stride]):   y[row] = sum_{k : col_flag[col_k]} val_k * seed_vals[col_slot[col_k]].
 * Same fixed summation order as lgc_spmm (entry order per lane group, chunk slots in order): deterministic, where a push
 * along the seeds' rows needs float atomics.  col_flag uint8 [table_rows], col_slot int32 [table_rows] (read only where
 * the flag is set). */
int lgc_seed_pull(const int32_t *rowptr, const lgc_entry *entries, int32_t row_begin, int32_t row_end, int32_t short_max,
                  const lgc_chunk *chunks, int32_t n_chunks, const lgc_multi_row *multi, int32_t n_multi, float *partials,
                  const uint8_t *col_flag, const int32_t *col_slot, const uint8_t *row_mark, const float *seed_vals,
                  int64_t seed_stride, int64_t table_rows, float *y, int64_t y_stride, int32_t dim, void *stream);

/* Which rows the seeded pull has to read at all: a batch of B users touches ~6 B of the 54 k item rows, and scanning the
 * other rows' 10 M entries for flags that are not there was 227 of the pull's 240 us.  lgc_seed_mark walks the rows listed
 * in `seed_rows` (int64 [n_seed], sorted; ids outside [row_begin, row_end) are skipped, repeats are harmless) of the
 * operator half that HOLDS the seeds -- the user rows, for a seed of users -- and stores `value` (0..255) in mark[col] for
 * each of their columns col < mark_len.  lgc_seed_pull given `row_mark` (uint8 [table_rows], NULL = read every row)
 * writes rows whose mark is 0 as zeros without reading their entries; value 0 takes the marks back after the pull. */
int lgc_seed_mark(const int32_t *rowptr, const lgc_entry *entries, int32_t row_begin, int32_t row_end, const int64_t *seed_rows,
                  int64_t n_seed, uint8_t *mark, int64_t mark_len, int32_t value, void *stream);

/* y[i, :dim] = sum_t coef[t] * src[t][i, :dim]  for i < n_rows, terms added in index order, each product
 * rounded before its add -- the order of the reference's running layer sum `out = out + x * alpha`
 * (src/lightgcn.py:93,97).  `src`, `src_stride`, `coef` are HOST arrays of n_terms (1..LGC_MAX_TERMS) entries;
 * src[t] are device pointers.  y may alias one of the sources (element-wise). */
#define LGC_MAX_TERMS 8
int lgc_lincomb(float *y, int64_t y_stride, const float *const *src, const int64_t *src_stride,
                const float *coef, int32_t n_terms, int64_t n_rows, int32_t dim, void *stream);

/* ---------------------------------------------------------------------------------------
 * Dense Adam step over a contiguous fp32 table in ONE pass (w, g, m, v read once; w, m, v written once): what
 * `optimizer.step()` of `torch.optim.Adam(model.parameters(), lr)` does at src/train_lightgcn.py:58,147 (amsgrad off, no
 * weight decay):  m <- m + (g - m)(1 - beta1);  v <- beta2 v + (1 - beta2) g^2;
 *                 w <- w - step_size * m / (sqrt(v) / bias_correction2_sqrt + eps)
 * with step_size = lr / (1 - beta1^t) and bias_correction2_sqrt = sqrt(1 - beta2^t) computed by the host (t = step
 * count), and (1 - beta1), (1 - beta2) handed over as the host rounds them from double, like torch's own scalars
 * (1.0f - 0.999f is 4.7e-5 away from 0.001f).  n elements each; the four pointers dword-aligned and at the SAME offset
 * inside a 16-byte line (LGC_E_ALIGN otherwise) -- 16-byte aligned tables, or the same row range of same-shaped tables:
 * a rank of a partitioned run updates only the rows it owns (its users + the item block), two calls per step.
 * ------------------------------------------------------------------------------------- */
int lgc_adam_step(float *w, const float *g, float *m, float *v, int64_t n, float one_minus_beta1, float beta2,
                  float one_minus_beta2, float eps, float step_size, float bias_correction2_sqrt, void *stream);
/* The same with the six scalars read from DEVICE memory: hyper = {1 - beta1, beta2, 1 - beta2, eps, step_size,
 * bias_correction2_sqrt} (fp32 [6]).  For launches captured in a HIP graph and replayed every step: step_size and
 * bias_correction2_sqrt change with the step count, so the host refreshes those 24 bytes before each replay. */
int lgc_adam_step_hp(float *w, const float *g, float *m, float *v, int64_t n, const float *hyper, void *stream);

/* ---------------------------------------------------------------------------------------
 * Pair scoring: scores[m] = <emb[idx0[m]], emb[idx1[m]]>.
 * Replaces src/lightgcn.py:123-125.  idx are the rows of edge_label_index
 * (src/utils_v2.py:184-190), int64.  Out-of-range pairs score NaN and set LGC_ST_INDEX_OOB.
 * ------------------------------------------------------------------------------------- */
int lgc_pair_dot(const float *emb, int64_t stride, int32_t dim, int64_t n_nodes,
                 const int64_t *idx0, const int64_t *idx1, int64_t n_pairs,
                 float *scores, int32_t *status, void *stream);

/* Serving tail of LightGCN.recommendK (src/lightgcn.py:175-177; called per request from
 * torchserve/lightgcn_handler.py:91): masked = scores * (1 - seen), then per row the k largest by
 * (value descending, index ascending), entirely on the device -- upstream copies the [rows, n_cols] score matrix
 * to the host first.  k <= 256 (LGC_E_RANGE beyond).  The mask comes in one of two forms (or neither: NULL, NULL):
 *   dense   seen fp32 [n_rows, n_cols], any values -- what upstream's handler builds per request
 *           (index_select on the sparse purchase matrix + to_dense, lightgcn_handler.py:88);
 *   lists   list_ptr int64 [n_users + 1], list_items int64 (a CSR of the purchase matrix, seen = 1 for listed
 *           columns), list_rows int64 [n_rows] = the user of each score row (NULL: row r is user r): the kernel
 *           builds a bit per column in LDS, the dense mask never exists (n_cols <= 983,040).
 *   scores fp32 [n_rows, n_cols] (row stride in floats), out_index int64 [n_rows, k], out_value fp32 [n_rows, k] or NULL. */
int lgc_mask_topk(const float *scores, int64_t score_stride, const float *seen, int64_t seen_stride,
                  const int64_t *list_ptr, const int64_t *list_items, const int64_t *list_rows, int64_t n_rows,
                  int32_t n_cols, int32_t k, int64_t *out_index, float *out_value, void *stream);

/* ---------------------------------------------------------------------------------------
 * Mini-batch sampler: for each of the `n` given users one positive and one negative item.
 * Replaces the per-row Python of batch_loader (src/utils_v2.py:168-181; its caller
 * src/train_lightgcn.py:132): `p = random.choice(item_id_idx_list)`,
 * `n = rejection-sample random.randint(0, n_items-1) + n_users until not in ignor_neg_list`.
 * The users themselves (random.sample without replacement, :174) are drawn by the caller.
 *
 *   users       int64 [n]   user ids (rows of the two CSRs below)
 *   pos_ptr/pos_items   int32 [n_users+1] / int64 [..]  each user's positive item ids (already offset by
 *                       n_users, duplicates allowed: the choice is uniform over list entries, as upstream)
 *   ign_ptr/ign_items   int32 [n_users+1] / int64 [..]  each user's ignore set, SORTED ascending per user
 *   seed, step  the draw is a pure function of (seed, step, position in the batch): counter-based RNG
 *   pos_out, neg_out    int64 [n]
 *   status      LGC_ST_INDEX_OOB for a user id outside [0, n_users) or without positives;
 *               LGC_ST_SAMPLER_EXHAUSTED when 256 draws all hit the ignore set (then neg = last draw)
 * ------------------------------------------------------------------------------------- */
#define LGC_ST_SAMPLER_EXHAUSTED 2
int lgc_sample_triples(const int64_t *users, int64_t n,
                       const int32_t *pos_ptr, const int64_t *pos_items,
                       const int32_t *ign_ptr, const int64_t *ign_items,
                       int64_t n_users, int64_t n_items, uint64_t seed, uint64_t step,
                       int64_t *pos_out, int64_t *neg_out, int32_t *status, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LGCONV_HIP_H */
