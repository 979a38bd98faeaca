/*
 * wgnn.h - C ABI of the MI355X (gfx950) weighted-GNN aggregation library
 *          (libwgnn_hip.so, built from scdeepsort_amd/csrc/).
 *
 * This is the drop-in boundary for scDeepSort's hot path.  The reference has no
 * FFI of its own; the operator boundary it replaces is the Python call
 *
 *     nf.block_compute(i, self.message_func, fn.mean('m', 'neigh'), layer)
 *                                                  (reference models/gnn.py:65)
 *
 * i.e. per-edge message  m_e = h[src]*alpha[k(e)]*w_e   (models/gnn.py:47-56),
 * mean over in-edges incl. the self-loop  [DGL 0.4.3 fn.mean], then
 * NodeUpdate = Linear + ReLU  (models/gnn.py:18-25), plus autograd's backward of
 * the same (train.py:84) and the graph-operand normalisation
 * normalize_weight (utils/preprocess_internal.py:15-23).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory unless the
 *     parameter name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - functions enqueue work on `stream` and return immediately: they never
 *     synchronise, never allocate persistent memory, never throw; they are
 *     re-entrant across streams, threads and devices (scratch is passed in by
 *     the caller; the only library state - "dynamic-LDS limit already raised"
 *     marks - is kept per device in atomics).  The CURRENT device
 *     (hipGetDevice) must be the one that owns `stream` and the pointers;
 *   - return value: 0 = ok, negative = WGNN_ERR_* (see wgnn_last_error_string);
 *   - CSR is destination-major: row r lists the in-edges of destination r,
 *     `col` = source index, `val` = normalised edge weight.  Self-loops are
 *     IMPLICIT (weight 1, added by the kernels), matching the reference's
 *     "normalise, then add self-loops" order (preprocess_internal.py:211-214).
 *   - feature matrices are row-major with a leading dimension in ELEMENTS;
 *     D and every ld must be multiples of 4 (16-byte rows for f32).
 */
#ifndef WGNN_H_
#define WGNN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WGNN_VERSION 206           /* 0.2.x - INCOMPATIBLE with 0.1.x binders: `neigh_sum` was inserted before `n_out` in
                                      wgnn_agg_fwd / wgnn_agg_fwd_tiled (0.1.1, should have been a major bump then - a 0.1.0
                                      caller would pass n_out in a pointer slot); 0.2.0 adds int64 row pointers
                                      (WGNN_FLAG_ROWPTR_I64, wgnn_normalize_rows_i64), WGNN_FLAG_SRC_PRESCALED and
                                      wgnn_linear_fwd_ex.  Binders must check wgnn_version() / 100 == 2.
                                      0.2.1: tile-plan entries may mark shared pairs (see wgnn_agg_fwd_tiled); a 0.2.0
                                      library would misread the marks, so a plan that carries them needs >= 201.
                                      0.2.2: WGNN_FLAG_OUT_SCALE_ALPHA (an older library ignores the bit: callers that set it
                                      need >= 202).  0.2.3: wgnn_agg_bwd_prepare, wgnn_ce_sum_fwd_bwd; wgnn_agg_bwd_src_tiled
                                      takes col_scale == NULL (pre-scaled gradient rows).  0.2.4: WGNN_PLAN_TALL (tall tile
                                      plans: 8 waves x 49 rows),
                                      wgnn_tile_plan_count / wgnn_tile_plan_fill.  0.2.5: a tile-plan segment with an odd number
                                      of entries is ALWAYS padded to even (0.2.4: only when shared pairs follow), so its size
                                      depends on its entry count alone: wgnn_tile_plan_count no longer reports pair counts
                                      (`seg_pairs` is ignored, may be NULL) and wgnn_tile_plan_fill takes the padded offsets;
                                      the aggregation kernels read either layout.  0.2.6: wgnn_csr_transpose_* (additive). */

/* Tile-plan geometry, OR-ed into the `block_rows` argument of wgnn_agg_fwd_tiled / wgnn_agg_bwd_src_tiled /
 * wgnn_agg_bwd_alpha_tiled (0.2.4; an older library rejects the bit with WGNN_ERR_PLAN): the plan was built for the TALL tile -
 * 8 waves x 49 destination rows = 392 item slots per tile, seg_ptr with 8 segments per (tile, block), 6-bit slot fields in the
 * entries - instead of 16 waves x 16 rows = 256 item slots, 16 segments, 4-bit slots. */
#define WGNN_PLAN_TALL (1 << 16)

/* error codes */
#define WGNN_OK                 0
#define WGNN_ERR_BAD_ARG       -1  /* NULL pointer / negative size / bad enum             */
#define WGNN_ERR_ALIGNMENT     -2  /* D or ld not a multiple of 4, or pointer not 16-B aligned */
#define WGNN_ERR_UNSUPPORTED   -3  /* dtype / width combination not built                  */
#define WGNN_ERR_WORKSPACE     -4  /* workspace too small                                  */
#define WGNN_ERR_LAUNCH        -5  /* hipLaunch / hipMemsetAsync failed                    */
#define WGNN_ERR_PLAN          -6  /* plan blob malformed or built for another CSR         */

/* which end of an edge is the gene (selects the alpha index rule, gnn.py:49-53) */
#define WGNN_SRC_IS_GENE  0   /* gene->cell edges: k(e) = source gene id   (gnn.py:51); rows are cells  */
#define WGNN_DST_IS_GENE  1   /* cell->gene edges: k(e) = dest   gene id   (gnn.py:52); rows are genes  */
#define WGNN_NO_ALPHA     2   /* plain weighted sum (cell-feature build, preprocess_internal.py:197-199;
                                 multi-GPU partial sums)                                                 */

/* element types of feature matrices */
#define WGNN_F32 0
#define WGNN_F16 1            /* storage only; accumulation is always fp32 */

/* flags for wgnn_agg_fwd */
#define WGNN_FLAG_RELU     1u   /* out = max(out, 0) after bias (NodeUpdate activation, gnn.py:21-22)      */
#define WGNN_FLAG_NO_MEAN  2u   /* skip the 1/(deg+1) division (partial sums that are all-reduced first)  */
#define WGNN_FLAG_NO_SELF  4u   /* skip the self-loop term                                                */
#define WGNN_FLAG_SELF_COMPACT 8u /* h_self holds one row per OUTPUT SLOT (h_self[i]) instead of per CSR row (h_self[r]);
                                     used with row_ids for seed mini-batches (train.py:71-81)                */
#define WGNN_FLAG_ROWPTR_I64 16u  /* `rowptr` points to int64_t[R+1] (scipy / torch CSR of large matrices) instead of int32_t[R+1].
                                     SURVEY 8b: "rowptr[R+1] i32 or i64".  The kernels read rowptr only for the inv_deg == NULL
                                     fallback (row length); non-zero OFFSETS stay 32-bit (plan items), so an operand still has
                                     < 2^31 non-zeros per GPU (see wgnn_plan_build_host_i64)                    */
#define WGNN_FLAG_SRC_PRESCALED 32u /* wgnn_agg_fwd_tiled, WGNN_SRC_IS_GENE: h_src already holds alpha[s]*h[s] (written by
                                     wgnn_linear_fwd_ex's scaled output); no scale pass, src_scratch may be NULL */

#define WGNN_FLAG_OUT_SCALE_ALPHA 64u /* wgnn_agg_fwd / wgnn_agg_fwd_tiled, WGNN_DST_IS_GENE (rows are genes): the finished row
                                     (after mean / self-loop / bias / ReLU) is multiplied by alpha[row] once more, i.e. the
                                     output is the NEXT layer's alpha-folded gene table, (h*alpha) of gnn.py:54, ready for
                                     WGNN_FLAG_SRC_PRESCALED - used when nothing else reads the unscaled gene rows (the
                                     layer below a cells-only last layer).  Forward entries only                */

int         wgnn_version(void);
const char* wgnn_last_error_string(int code);

/* Bytes of caller-provided scratch one aggregation call needs (the library never allocates):
 *   *partials_bytes    = n_partials * D * 4                      (long-row / column-split partial sums)
 *   *src_scratch_bytes = n_src * D * 4 for the tiled kernels that fold a per-row factor into the source table
 *                        (wgnn_agg_fwd_tiled with WGNN_SRC_IS_GENE, wgnn_agg_bwd_src_tiled's g_scratch), else 0. */
int wgnn_agg_workspace_bytes(int64_t n_partials, int64_t n_src, int32_t D, int alpha_mode, int tiled,
                             int64_t* partials_bytes, int64_t* src_scratch_bytes);

/* ---------------------------------------------------------------------------
 * Execution plan: splits long rows into fixed-size chunks so that a hub gene
 * with ~C in-edges does not serialise on one wavefront.  Built once per CSR
 * (host side, from a host copy of rowptr) and uploaded by the caller.
 *
 *   items_host : int32[4 * n_items]  {row_slot, nnz_begin, nnz_end, partial_slot | -1}
 *   long_host  : int32[4 * n_long]   {row_slot, first_partial_slot, n_partials, 0}
 * `row_slot` indexes row_ids (or is the row itself when row_ids == NULL).
 * Call once with items_host == NULL to obtain the counts.
 * ------------------------------------------------------------------------- */
int wgnn_plan_build_host(const int32_t* rowptr_host, const int32_t* row_ids_host, int64_t n_rows,
                         int32_t chunk_nnz,
                         int32_t* items_host, int32_t* long_host,
                         int64_t* n_items, int64_t* n_long, int64_t* n_partials);

/* Same, from a 64-bit row-pointer array (scipy / torch CSR of large matrices).  The kernels address non-zeros with
 * 32-bit offsets (items hold {begin, end} as int32; col/val of 2^31 edges would be 17 GB per direction): when
 * rowptr_host[r+1] > INT32_MAX for any planned row this returns WGNN_ERR_UNSUPPORTED - the caller shards the cell
 * axis (one CSR per GPU / per shard) so that every shard stays below 2^31 non-zeros. */
int wgnn_plan_build_host_i64(const int64_t* rowptr_host, const int32_t* row_ids_host, int64_t n_rows,
                             int32_t chunk_nnz,
                             int32_t* items_host, int32_t* long_host,
                             int64_t* n_items, int64_t* n_long, int64_t* n_partials);

/* ---------------------------------------------------------------------------
 * K1  forward:  replaces message_func + fn.mean (+ optional bias/ReLU epilogue)
 *               (models/gnn.py:47-56,65 and :20-22)
 *
 *   for each output slot i (row r = row_ids ? row_ids[i] : i):
 *     SRC_IS_GENE: out[i] = ( sum_j val_j*alpha[col_j]*h_src[col_j] + alpha[self_idx]*h_self[r] ) * inv_deg[r]
 *     DST_IS_GENE: out[i] = ( alpha[r]*sum_j val_j*h_src[col_j]    + alpha[self_idx]*h_self[r] ) * inv_deg[r]
 *     NO_ALPHA   : out[i] = ( sum_j val_j*h_src[col_j] [+ h_self[r]] ) * inv_deg[r]
 *   then  out[i] += bias (if bias) ; out[i] = relu(out[i]) (if WGNN_FLAG_RELU).
 *   inv_deg == NULL  =>  1/(rowptr[r+1]-rowptr[r]+1)   (in-degree counts the self-loop).
 *   self_idx is gene_num+1 for cell rows and gene_num for gene rows (gnn.py:42,49,53).
 *
 *   items/long_rows come from wgnn_plan_build_host (device copies).  `partials`
 *   must hold n_partials*D floats (may be NULL when n_long == 0).
 *   neigh_sum (optional, f32 [n_out, D] contiguous, NULL to skip): receives sum_j val_j*[alpha[col_j]*]h_src[col_j] per
 *   slot BEFORE the row factor / self-loop / mean / bias / ReLU.  Training saves it for DST_IS_GENE rows: the
 *   gradient of alpha[r] is then inv_deg[r]*<g[r], neigh_sum[r]> - a row dot product instead of a second pass over the
 *   edges (K3).
 * ------------------------------------------------------------------------- */
int wgnn_agg_fwd(const void* rowptr /* int32_t[R+1]; int64_t[R+1] with WGNN_FLAG_ROWPTR_I64 */, const int32_t* col, const float* val,
                 const float* alpha, int alpha_mode, int32_t self_idx,
                 const void* h_src, int64_t ld_src,
                 const void* h_self, int64_t ld_self,
                 const int32_t* row_ids, const float* inv_deg, const float* bias,
                 void* out, int64_t ld_out, float* neigh_sum,
                 int64_t n_out, int32_t D, int dtype_in, int dtype_out, uint32_t flags,
                 const int32_t* items, int64_t n_items,
                 const int32_t* long_rows, int64_t n_long,
                 float* partials, int64_t n_partials,
                 void* stream);

/* ---------------------------------------------------------------------------
 * K1t forward, LDS-streamed variant of K1 (same arithmetic, same outputs) for D <= 256, f32,
 *     h_src contiguous (leading dimension == D).  One 1024-thread workgroup per TILE of up to 256
 *     destination rows (16 waves x 16 rows); the source table is streamed through LDS in blocks of
 *     `block_rows` rows.  The tile plan is a re-ordering of the CSR (built once per graph):
 *
 *   tile_hdr   : int32[n_tiles * 2]        {col_begin, col_end} = source range the tile reduces over
 *   tile_items : int32[n_tiles * 256 * 4]  per tile, wave-major: {row_slot | -1 (padding), -, -, partial_slot | -1};
 *                                          a wave's 16 slots are filled from slot 0.  LOADER WAVES: the leading waves of a tile
 *                                          whose slot 0 is empty own no rows - the kernel makes them issue the tile's whole
 *                                          global->LDS stream while the other waves only compute (a plan that gives every
 *                                          wave rows keeps all 16 waves streaming their share; same results either way)
 *   entries    : int32[n_entries * 2]      {meta, weight (f32 bits)}, grouped by (tile, block, wave) segment;
 *                                          block = (col - col_begin) / block_rows.  meta = dst_slot_in_wave << 8 (bits 8..13) |
 *                                          src_row_in_block (bits 0..7), any order inside a segment, plus optionally
 *                                          SHARED PAIRS: two entries of a segment on the same source row may be marked
 *                                          (bit 31 on both, the first also carrying the second's slot in bits 16..21);
 *                                          the kernel then stages that source row once for both.  Marked pairs must be the
 *                                          LAST entries of their segment and start at an even offset from the segment's
 *                                          begin (pad the unmarked run with a zero-weight copy of its last entry; bit 30
 *                                          marks such a filler for tools, kernels ignore it).  n_entries = nnz + fillers.
 *   seg_ptr    : int32[n_tiles*nblk_max*16 + 1]  entry offsets per (tile, block, wave)
 *   block_rows : source rows per LDS block the plan was built for (16..255; 2*block_rows*D*4 B <= 160 KiB,
 *                at D == 256 minus 4 KiB for the per-wave weight strips, i.e. <= 78)
 *   long_rows / partials as in wgnn_agg_fwd (every row of a column-split plan is a "long row").
 *   src_scratch: float[n_src * D], required for WGNN_SRC_IS_GENE: alpha is folded into the source rows
 *                once ((h*alpha), gnn.py:54) instead of once per edge.
 * ------------------------------------------------------------------------- */
int wgnn_agg_fwd_tiled(const void* rowptr /* int32_t[R+1] | int64_t[R+1] (WGNN_FLAG_ROWPTR_I64) | NULL with inv_deg */, const float* alpha, int alpha_mode, int32_t self_idx,
                       const float* h_src, int64_t n_src, float* src_scratch,
                       const float* h_self, int64_t ld_self,
                       const int32_t* row_ids, const float* inv_deg, const float* bias,
                       float* out, int64_t ld_out, float* neigh_sum, int64_t n_out, int32_t D, uint32_t flags,
                       const int32_t* entries, const int32_t* seg_ptr, int32_t nblk_max, int32_t block_rows,
                       const int32_t* tile_items, const int32_t* tile_hdr, int64_t n_tiles,
                       const int32_t* long_rows, int64_t n_long, float* partials, int64_t n_partials,
                       void* stream);

/* ---------------------------------------------------------------------------
 * K2  backward w.r.t. the source rows (autograd of K1, train.py:84):
 *     runs over the TRANSPOSED structure (row s lists the destinations r that s feeds,
 *     t_val = the same normalised weights re-ordered).
 *
 *     SRC_IS_GENE: T[s] = sum_r t_val*inv_deg[r]*g[r];  dh_src[s] (+)= alpha[s]*T[s];
 *                  dalpha[s] (+)= <h_src[s], T[s]>      (if dalpha && h_src)
 *     DST_IS_GENE: dh_src[s] (+)= sum_r t_val*alpha[r]*inv_deg[r]*g[r]
 *     NO_ALPHA   : dh_src[s] (+)= sum_r t_val*inv_deg[r]*g[r]
 *   `accumulate` != 0 adds into dh_src instead of overwriting.
 *   inv_deg is REQUIRED here (it belongs to the destination rows).
 * ------------------------------------------------------------------------- */
int wgnn_agg_bwd_src(const int32_t* t_rowptr, const int32_t* t_col, const float* t_val,
                     const float* alpha, int alpha_mode,
                     const float* inv_deg_dst,
                     const float* g, int64_t ld_g,
                     const float* h_src, int64_t ld_src,
                     float* dh_src, int64_t ld_dh, float* dalpha, int accumulate,
                     int64_t n_src, int32_t D,
                     const int32_t* items, int64_t n_items,
                     const int32_t* long_rows, int64_t n_long,
                     float* partials, int64_t n_partials,
                     void* stream);

/* ---------------------------------------------------------------------------
 * K3  backward w.r.t. alpha for DST_IS_GENE rows and for the self-loop scalars:
 *     dalpha_row[i]  = inv_deg[r] * < g[i], sum_j val_j*h_src[col_j] >       (DST_IS_GENE only, else untouched)
 *     dself_row[i]   = inv_deg[r] * < g[i], h_self[r] >                      (per-row partial of dalpha[self_idx])
 *   The caller adds dalpha_row into dalpha[r] and sums dself_row into dalpha[self_idx].
 * ------------------------------------------------------------------------- */
int wgnn_agg_bwd_alpha(const int32_t* rowptr, const int32_t* col, const float* val,
                       const float* inv_deg, const int32_t* row_ids,
                       const float* g, int64_t ld_g,
                       const float* h_src, int64_t ld_src,
                       const float* h_self, int64_t ld_self,
                       float* dalpha_row, float* dself_row,
                       int64_t n_out, int32_t D, uint32_t flags,   /* WGNN_FLAG_SELF_COMPACT only */
                       const int32_t* items, int64_t n_items,
                       const int32_t* long_rows, int64_t n_long,
                       float* partials, int64_t n_partials,
                       void* stream);

/* ---------------------------------------------------------------------------
 * K2t / K3t  LDS-streamed variants of K2 / K3 (D <= 256, f32, contiguous rows), same tile-plan layout as K1t.
 *   K2t runs over the tile plan of the TRANSPOSED structure; `col_scale[r]` = inv_deg[r] (x alpha[r] for
 *   WGNN_DST_IS_GENE) is folded into the gradient rows once (g_scratch: float[n_dst*D]); col_scale == NULL (0.2.3): `g`
 *   already carries that factor (written so by wgnn_agg_bwd_prepare) and is read as the source table, no scale pass.
 *   K3t runs over the forward structure's tile plan.
 * ------------------------------------------------------------------------- */
int wgnn_agg_bwd_src_tiled(const float* alpha, int alpha_mode, const float* col_scale,
                           const float* g, int64_t n_dst, float* g_scratch,
                           const float* h_src, int64_t ld_src, float* dh_src, int64_t ld_dh, float* dalpha,
                           int accumulate, int64_t n_src, int32_t D,
                           const int32_t* entries, const int32_t* seg_ptr, int32_t nblk_max, int32_t block_rows,
                           const int32_t* tile_items, const int32_t* tile_hdr, int64_t n_tiles,
                           const int32_t* long_rows, int64_t n_long, float* partials, int64_t n_partials,
                           void* stream);
int wgnn_agg_bwd_alpha_tiled(const float* inv_deg, const float* g, int64_t ld_g,
                             const float* h_src, const float* h_self, int64_t ld_self,
                             float* dalpha_row, float* dself_row, int64_t n_out, int32_t D,
                             const int32_t* entries, const int32_t* seg_ptr, int32_t nblk_max, int32_t block_rows,
                             const int32_t* tile_items, const int32_t* tile_hdr, int64_t n_tiles,
                             const int32_t* long_rows, int64_t n_long, float* partials, int64_t n_partials,
                             void* stream);

/* ---------------------------------------------------------------------------
 * K4  graph-operand normalisation: normalize_weight (preprocess_internal.py:15-23)
 *     val_out[j] = deg_r * val_in[j] / sum_{j in row r} val_in[j]      for rows with >= 1 entry
 *     inv_deg[r] = 1 / (deg_r + 1)                                     (if inv_deg != NULL)
 * ------------------------------------------------------------------------- */
int wgnn_normalize_rows(const int32_t* rowptr, const float* val_in, float* val_out, float* inv_deg,
                        int64_t n_rows, void* stream);
/* the same over a 64-bit row-pointer array ("rowptr[R+1] i32 or i64") */
int wgnn_normalize_rows_i64(const int64_t* rowptr, const float* val_in, float* val_out, float* inv_deg,
                            int64_t n_rows, void* stream);

/* ---------------------------------------------------------------------------
 * K5  seeded neighbour subsampling (train.py:37-40,71-78: NeighborSampler(expand_factor = num_neighbors,
 *     neighbor_type = 'in')): for each of n_rows destination rows (row_ids or 0..n_rows-1) draw min(k, deg + 1) of its
 *     deg + 1 in-edges - the deg CSR entries plus the unit self-loop the reference's graph holds explicitly
 *     (preprocess_internal.py:213-214) - uniformly without replacement.  Output in ELL form (static shapes):
 *       out_col / out_val [n_rows * k] : row i owns [i*k, i*k + out_cnt[i]) (drawn real edges, parent col / val)
 *       out_cnt  [n_rows]              : number of real edges drawn
 *       out_self [n_rows]              : 1.0 where the self-loop was among the draws
 *       out_inv  [n_rows]              : 1 / (number of drawn edges)   - fn.mean's divisor
 *     Random numbers = hash(seed, *step, stream_id, row, draw): `step` is a DEVICE counter the caller advances on the
 *     stream (a captured hipGraph therefore draws a new sample at every replay); no state is read back.  k <= 256.
 * ------------------------------------------------------------------------- */
int wgnn_sample_rows(const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* row_ids,
                     int64_t n_rows, int32_t k, uint64_t seed, const int64_t* step, int32_t stream_id,
                     int32_t* out_col, float* out_val, int32_t* out_cnt, float* out_self, float* out_inv,
                     void* stream);

/* ---------------------------------------------------------------------------
 * Dense half of a layer on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, an fmaf chain in k order):
 *     out[M, N] = act( x[M, K] . w[N, K]^T + bias[N] )
 * = NodeUpdate.forward's `activation(fc_neigh(neigh))` (models/gnn.py:18-25; w / bias in nn.Linear's layout) and the
 * classifier head `linear(h)` (models/gnn.py:66-67).  flags: WGNN_FLAG_RELU or 0.  K, ld_x, ld_w multiples of 4,
 * x / w 16-byte aligned; M, N arbitrary.  bias may be NULL.  Makes the ABI self-sufficient for one whole layer.
 * ------------------------------------------------------------------------- */
/* Extended form (0.2.0):
 *   x_dtype WGNN_F32 | WGNN_F16: fp16-STORED features (BASELINE cfg5) are widened in registers on their way into LDS - fp16-
 *     rounded inputs, fp32 multiply-accumulate, no fp32 copy of x in HBM (x 8-byte aligned then);
 *   row_scale / out_scaled (both or neither): out_scaled[m, :] = row_scale[m] * out[m, :], written from the same
 *     accumulators.  With row_scale = alpha[0:G] this is the alpha-folded gene table (h*alpha, models/gnn.py:54) that
 *     wgnn_agg_fwd_tiled takes under WGNN_FLAG_SRC_PRESCALED - no separate scale pass.  `out` may be NULL then. */
int wgnn_linear_fwd_ex(const void* x, int x_dtype, int64_t ld_x, const float* w, int64_t ld_w, const float* bias,
                       float* out, int64_t ld_out, const float* row_scale, float* out_scaled, int64_t ld_out_scaled,
                       int64_t M, int32_t N, int32_t K, uint32_t flags, void* stream);
int wgnn_linear_fwd(const float* x, int64_t ld_x, const float* w, int64_t ld_w, const float* bias,
                    float* out, int64_t ld_out, int64_t M, int32_t N, int32_t K, uint32_t flags, void* stream);

/* Weight gradient of the same Linear (autograd of fc_neigh / linear, train.py:84):
 *     dW[N, K] (+)= sum_m g[m, N] * x[m, K]
 * The node axis M (1e5 at cfg3) is the reduction: it is cut into n_slabs slabs whose partial [N, K] products are folded in
 * fixed order (deterministic).  Query n_slabs / workspace bytes with wgnn_linear_wgrad_workspace; N, K, ld_g, ld_x
 * multiples of 4. */
int wgnn_linear_wgrad_workspace(int64_t M, int32_t N, int32_t K, int64_t* n_slabs, int64_t* bytes);
int wgnn_linear_wgrad(const float* g, int64_t ld_g, const float* x, int64_t ld_x, float* dW, int64_t ld_dw,
                      int64_t M, int32_t N, int32_t K, int accumulate, float* workspace, int64_t n_slabs, void* stream);

/* ---------------------------------------------------------------------------
 * One reference layer on a block in the reference's literal order (SURVEY 8b's optional "fused variant" - here a COMPOSED
 * entry: two launches through the caller's neigh_scratch, see DESIGN.md section 8 for why the fusion itself does not pay):
 *     neigh = nf.block_compute(i, message_func, fn.mean('m','neigh'))   (models/gnn.py:47-56,65)  = wgnn_agg_fwd (f32)
 *     out   = relu(fc_neigh(neigh))                                      (models/gnn.py:18-25)     = wgnn_linear_fwd
 * Arguments up to n_partials as wgnn_agg_fwd (f32 in/out, no bias, agg_flags without WGNN_FLAG_RELU);
 * neigh_scratch: float[n_out * D] (caller-owned); W: float[H, ld_w] (nn.Linear layout), bias: float[H] or NULL;
 * lin_flags: WGNN_FLAG_RELU or 0; out: float[n_out, ld_out].
 * ------------------------------------------------------------------------- */
int wgnn_agg_linear_relu_fwd(const void* rowptr /* as wgnn_agg_fwd */, const int32_t* col, const float* val,
                             const float* alpha, int alpha_mode, int32_t self_idx,
                             const float* h_src, int64_t ld_src, const float* h_self, int64_t ld_self,
                             const int32_t* row_ids, const float* inv_deg,
                             int64_t n_out, int32_t D, uint32_t agg_flags,
                             const int32_t* items, int64_t n_items, const int32_t* long_rows, int64_t n_long,
                             float* partials, int64_t n_partials,
                             float* neigh_scratch,
                             const float* W, int64_t ld_w, const float* bias, int32_t H, uint32_t lin_flags,
                             float* out, int64_t ld_out, void* stream);

/* ---------------------------------------------------------------------------
 * Fused glue of the training step (0.2.3; reference train.py:80-87 through torch autograd).
 *
 * wgnn_agg_bwd_prepare: everything the backward of ONE aggregation pass derives from the upstream gradient, in one read of it
 * (full passes: one gradient row per CSR row, f32):
 *     g             = gout * (out > 0)              out = the saved forward output (NodeUpdate's ReLU, gnn.py:21-22) or NULL
 *     g_scaled[r]   = inv_deg[r] * (alpha[r] for WGNN_DST_IS_GENE) * g[r]      float[n_rows, D]: K2t's source table (col_scale NULL)
 *     dh_self[r]    = alpha[self_idx] * inv_deg[r] * g[r]                       gradient of the self rows (alpha = 1 for WGNN_NO_ALPHA)
 *     dalpha_row[r] = inv_deg[r] * < g[r], neigh_sum[r] >                       neigh_sum: the raw sums K1 saved (float[n_rows, D])
 *     dself_row[r]  = inv_deg[r] * < g[r], h_self[r] >                          the caller sums it into dalpha[self_idx]
 *     dbias[c]      = sum_r g[r, c]                                              block partials in `workspace`, folded in fixed order
 *   Any of g_scaled / dh_self / dalpha_row / dself_row / dbias may be NULL.  inv_deg NULL = 1.  D <= 1024, multiples of 4.
 *   workspace: wgnn_agg_bwd_prepare_workspace floats (needed for dbias only).
 *
 * wgnn_ce_sum_fwd_bwd: CrossEntropyLoss(reduction='sum') (train.py:36) over float logits[n_rows, n_classes] and int64 labels:
 *     *loss_sum = sum_r ( logsumexp(logits[r]) - logits[r, labels[r]] ) ;  dlogits[r] = softmax(logits[r]) - onehot(labels[r])
 *   (dlogits may be NULL).  workspace: wgnn_ce_sum_workspace floats.  Deterministic (fixed-order folds, no atomics).
 *   A label of -100 (torch's default ignore_index) contributes 0 to the loss and a zero dlogits row, as in the framework call
 *   this replaces.  Any other label outside [0, n_classes) - tested on the 64-bit value - is never used as an index: its
 *   row's loss term and dlogits row are NaN (the framework call raises a device-side assertion; a kernel that never
 *   synchronises cannot, NaN is its loud answer).  expf / logf, not the fast intrinsics.
 * ------------------------------------------------------------------------- */
int wgnn_agg_bwd_prepare_workspace(int64_t n_rows, int32_t D, int64_t* floats);
int wgnn_agg_bwd_prepare(const float* gout, int64_t ld_gout, const float* out, int64_t ld_out,
                         const float* inv_deg, const float* alpha, int alpha_mode, int32_t self_idx,
                         float* g_scaled, const float* h_self, int64_t ld_self, float* dh_self, int64_t ld_dh,
                         const float* neigh_sum, float* dalpha_row, float* dself_row, float* dbias,
                         int64_t n_rows, int32_t D, float* workspace, int64_t workspace_floats, void* stream);
int wgnn_ce_sum_workspace(int64_t n_rows, int64_t* floats);
int wgnn_ce_sum_fwd_bwd(const float* logits, int64_t ld_logits, const int64_t* labels, int64_t n_rows, int32_t n_classes,
                        float* loss_sum, float* dlogits, int64_t ld_dlogits, float* workspace, int64_t workspace_floats,
                        void* stream);

/* ---------------------------------------------------------------------------
 * Tile-plan construction on the device (0.2.4, revised 0.2.5).  `entries` / `seg_ptr` of a tile plan (see wgnn_agg_fwd_tiled) from
 * the CSR and the row -> (tile, wave, slot) assignment, without a sort: one wavefront per (tile, wave), lane = destination slot,
 * steps through its <= 64 destination rows (their non-zeros are sorted by column) along the tile's source range.  Two passes:
 *   wgnn_tile_plan_count : seg_total[s] = entries of segment s  (s = (tile * nblk_max + block) * waves + wave; array
 *                          zero-initialised by the caller; every lane counts along its own row, no group walk)
 *   caller               : seg_ptr = exclusive prefix sum of seg_total + (seg_total & 1)   [n_seg + 1 offsets]
 *   wgnn_tile_plan_fill  : entries[seg_ptr[s] .. seg_ptr[s + 1]) = [unshared][pad, iff the count is odd][shared pairs]: the walk in
 *                          lock step (wave minimum of the pending columns -> ballot = the group on that source row) writes the
 *                          unshared entries upwards from the segment's start and the pairs downwards from its end
 *   seg_pairs : ignored since 0.2.5 (may be NULL)
 *   slot_vrow : int32[n_row_tiles * waves * rpw]   virtual row of (row tile, wave, slot) | -1
 *   vrow_*    : per virtual row: CSR row, part j, parts k  (the row's non-zeros j, j + k, j + 2k, ...; k = 1: the whole row)
 *   flat_t    : int32[n_tiles]  row tile of the tile launched at position f;  tile_hdr as in wgnn_agg_fwd_tiled
 *   waves x rpw : 16 x 16 or 8 x 49 (WGNN_PLAN_TALL); block_rows: source rows per LDS block (<= 255)
 * Deterministic, no atomics, no allocation; col / rowptr are int32 (an operand holds < 2^31 non-zeros per GPU).
 * ------------------------------------------------------------------------- */
int wgnn_tile_plan_count(const int32_t* rowptr, const int32_t* col, const int32_t* slot_vrow,
                         const int32_t* vrow_row, const int32_t* vrow_part, const int32_t* vrow_k,
                         const int32_t* flat_t, const int32_t* tile_hdr, int64_t n_tiles, int32_t waves, int32_t rpw,
                         int32_t nblk_max, int32_t block_rows, int32_t* seg_total, int32_t* seg_pairs, void* stream);
int wgnn_tile_plan_fill(const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* slot_vrow,
                        const int32_t* vrow_row, const int32_t* vrow_part, const int32_t* vrow_k,
                        const int32_t* flat_t, const int32_t* tile_hdr, int64_t n_tiles, int32_t waves, int32_t rpw,
                        int32_t nblk_max, int32_t block_rows, const int32_t* seg_total, const int32_t* seg_pairs,
                        const int32_t* seg_ptr, int32_t* entries, void* stream);

/* ---------------------------------------------------------------------------
 * Stable CSR transpose on the device (0.2.6): the gene-major copy of the (cells x genes) expression CSR - every stored value
 * gives a cell->gene AND a gene->cell edge (preprocess_internal.py:170-173), and normalize_weight is per destination (:17-23), so
 * the genes<-cells direction needs the RAW values re-ordered by gene, cells ascending inside a gene - without a sort: an entry's
 * place in its gene's row is the number of earlier cells that express the gene (per-chunk LDS histograms, a prefix over the
 * chunks, then every chunk walks its cells in order).  Deterministic; n_cols <= 32768 (else WGNN_ERR_UNSUPPORTED: the caller
 * sorts).  Precondition: a row lists a column at most once.
 *   wgnn_csr_transpose_workspace : n_chunks and the byte size of `counts` (int32 [n_chunks * n_cols]) for an operand
 *   wgnn_csr_transpose_count     : counts[chunk][col], t_count[col] = entries of column col (rows with row_keep[r] == 0 dropped;
 *                                  row_keep == NULL keeps every row)
 *   caller                       : t_rowptr[0] = 0, t_rowptr[c + 1] = t_rowptr[c] + t_count[c]
 *   wgnn_csr_transpose_fill      : t_col[t_rowptr[c] ..) = the rows that list column c, ascending; t_val their values
 *                                  (overwrites `counts`)
 * ------------------------------------------------------------------------- */
int wgnn_csr_transpose_workspace(int64_t n_rows, int32_t n_cols, int64_t* n_chunks, int64_t* bytes);
int wgnn_csr_transpose_count(const int32_t* rowptr, const int32_t* col, const uint8_t* row_keep, int64_t n_rows, int32_t n_cols,
                             int64_t n_chunks, int32_t* counts, int32_t* t_count, void* stream);
int wgnn_csr_transpose_fill(const int32_t* rowptr, const int32_t* col, const float* val, const uint8_t* row_keep, int64_t n_rows,
                            int32_t n_cols, int64_t n_chunks, int32_t* counts, const int32_t* t_rowptr, int32_t* t_col, float* t_val,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WGNN_H_ */
