/*
 * pna_amd.h -- C ABI of libpna_amd.so, the MI355X (gfx950) implementation of the PNA
 * message-passing hot path of lukecavabarrett/pna.
 *
 * The reference has no FFI of its own (it is pure Python/PyTorch); the boundary it exposes is the
 * `reduce_func` / aggregator / scaler operator set that DGL's `update_all` schedules, plus the
 * dense `nn.Linear` that follows it.  Each entry point below names the reference code it
 * replaces (paths relative to the reference root).
 *
 * Conventions
 *   - plain C, no ownership transfer: every pointer is a DEVICE pointer owned by the caller
 *     (except where marked host), valid until the stream reaches the end of the call's work;
 *   - stream-ordered and asynchronous: nothing here synchronises the device;
 *   - re-entrant: no global mutable state except the per-thread last-error string;
 *   - return value 0 = success, negative = error (see PNA_E_*); pna_last_error() gives the text;
 *   - all data is fp32 and all arithmetic is fp32 (pna_posttrans_x3_f32 evaluates each fp32 product as six exact
 *     bf16 partial products accumulated in fp32 -- fp32-level accuracy, see there); indices are int32; row strides
 *     (ld*) are in floats.
 */
#ifndef PNA_AMD_H
#define PNA_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNA_ABI_VERSION 21 /* 21: the GUARDED fp16 x 2 contraction of the one-kernel layer (round 6): + pna_fused_degree_args.{w_img_x3,
                                  image_stride_x3, guard_ws, guard_ws_bytes, arith}, PNA_FD_ARITH_*, pna_fused_image_bytes, pna_fused_pack_f32,
                                  pna_fused_degree_guard_bytes; the fp16 x 2 images carry a 1024-byte tail (column scales + the guard's column
                                  thresholds); image_stride must EQUAL pna_fused_image_bytes; row scale bound in [2^14, 2^15).
                              20: - pna_fused_roles_{supported,image_bytes,grid,f32} (ABI 19's one-kernel layer with gather / multiply wavefront
                                  roles: parity-green, 2.1-2.7x slower than pna_fused_degree_f32, never on a product path -- removed from the
                                  library in round 5; the source lives on as tools/ubench/fused_roles.hip, the result in DESIGN.md 4.9).
                                  + pna_fused_degree_args.tile_counter (dynamic tile schedule), pna_fused_degree_tile_rows.
                                  pna_fused_degree_{image_bytes,pack_f32,f32}: images of two fp16 terms + a 512-byte tail of column scales
                                  (an opaque format between the pack function and the kernel of ONE library; the byte count changed).
                                  The trailing fields added "inside 19" are part of 20's structs: a binding that knows them no longer
                                  passes the version check of a library that does not.  Every args struct carries struct_size first.
                              19: struct_size first in every args struct; + pna_posttrans_dw_f32 / pna_posttrans_dw_grouped_f32
                                  (+ _workspace_bytes), pna_tower_layer_args.{edge_type, edge_table, ld_edge_table, n_edge_types, no_self_panel},
                                  pna_segreduce_bwd_args.{stat_row_of, stat_node_of, stat_rows}, the packed / in-place rows of
                                  pna_segreduce_bwd_pull_f32, 64-bit source-row addressing (x_rows < 2^32, any pitch >= F).
                              18: + pna_bn_tail_{workspace_bytes,fwd_f32,bwd_f32}: batch-statistics BatchNorm + ReLU + residual of the training path.
                              17: + pna_fused_degree_args.spare_workgroups (the rest-row launches beside the persistent kernel).
                              16: + pna_segreduce_bwd_pull_f32 (the backward's max / min terms inside the pull: no scatter atomics).
                              15: pna_fused_degree_*: wide shapes (F in 113..128 and / or N in 81..128).
                              14: pna_segreduce_args.edge_type / n_edge_types (edge terms from a table of edge types); the hand-scheduled gather takes
                                  edge terms (per edge or per type).
                              13: pna_fused_degree_args.x_dst / h_self / row_post + pna_fused_tower_{image_bytes,pack_f32}: the one-kernel layer
                                  for PNALayer with one tower.
                              12: pna_segreduce_args.out_row_of (the tower layers' aggregate written in degree order).
                              11: - pna_posttrans_x3w_* (ABI 8's contraction on 32x32 tiles: parity-green, 5 % slower than the 16x16 kernel on every
                                  shape measured, never selected -- removed, round 3).
                              10: + pna_fused_degree_{image_bytes,pack_f32,f32} (gather + degree-grouped contraction in one kernel).
                              9: pna_posttrans_args.row_perm / tile_image / image_stride (degree-grouped contraction),
                                pna_segreduce_args.heavy_out_rows.
                             8: + pna_posttrans_x3w_* (the bf16x3 contraction on 32x32 matrix-core tiles).
                             7: + pna_small_*, pna_tower_post_*, pna_tower_layer_f32 (the molecule-batch tower layer).
                             6: pna_posttrans_args: act_slope (LeakyReLU), n_tower + tower strides.
                             5: pna_posttrans_args.pipeline; the hand-scheduled gather takes dst_term; + pna_pack_rows_f32.
                             4: + pna_posttrans_x3_*, pna_fused_simple_f32, pna_collate_*, PNA_AGG_VAR_RAW */

#define PNA_OK 0
#define PNA_E_INVALID (-1)   /* bad argument (null pointer, unsupported size, unknown code) */
#define PNA_E_LAUNCH (-2)    /* the HIP runtime refused a launch / copy */
#define PNA_E_NODEVICE (-3)  /* no gfx950 device visible */

typedef void* pna_stream_t; /* a hipStream_t (0 = the null stream) */

/* Every *_args struct starts with `struct_size` = sizeof of the struct in the header the CALLER was compiled against (ABI 19, round 4:
 * fields had been appended through 18 ABI versions with only pna_abi_version() between a stale binding and a wild pointer).  An entry
 * point refuses a struct shorter than its own (PNA_E_INVALID, pna_last_error names the two sizes); fields appended in later ABI
 * versions therefore never read past a caller's struct.  PNA_ARGS_INIT zero-fills and stamps a struct in C. */
#define PNA_ARGS_INIT(type) ((type){ .struct_size = (uint32_t)sizeof(type) })

/* Aggregator codes -- models/dgl/aggregators.py:54-56 (AGGREGATORS dict) and
 * models/pytorch/pna/aggregators.py:149-152.  The order of codes in `aggr[]` is the order of the
 * F-wide blocks in the output, exactly like the order of names on the reference's command line. */
enum {
  PNA_AGG_MEAN = 0, /* dgl/aggregators.py:6-7    sum / D                                  */
  PNA_AGG_SUM = 1,  /* dgl/aggregators.py:50-51                                            */
  PNA_AGG_MAX = 2,  /* dgl/aggregators.py:10-11  NaN-propagating, like torch.max           */
  PNA_AGG_MIN = 3,  /* dgl/aggregators.py:14-15                                            */
  PNA_AGG_STD = 4,  /* dgl/aggregators.py:18-19  sqrt(relu(E[x^2]-E[x]^2) + 1e-5)          */
  PNA_AGG_VAR = 5,  /* dgl/aggregators.py:22-26  relu(E[x^2]-E[x]^2)                       */
  PNA_AGG_VAR_RAW = 6 /* pytorch_geometric/aggregators.py:25-28  E[x^2]-E[x]^2, NOT clamped (forward only) */
};
#define PNA_MAX_AGGR 8
#define PNA_MAX_SCALER 8

/* Optional launch tuning; all-zero = library defaults. */
typedef struct pna_tuning {
  int32_t lanes_per_row;   /* lanes of a 64-wide wavefront that share one destination row (each lane owns
                              `vec` consecutive features); 0 = auto = min(64, ceil(F/vec))             */
  int32_t unroll;          /* in-edges of one row gathered per loop trip (2,4,8); 0 = auto            */
  int32_t rows_per_group;  /* destination rows processed back to back by one lane group; 0 = auto    */
  int32_t vec;             /* features per lane: 4 (dwordx4 gathers) or 1; 0 = auto                   */
  int32_t nt_store;        /* 1 = non-temporal output stores; 0 = auto(1); -1 = plain stores          */
  int32_t prefetch;        /* 1 = fetch the next row's source ids one row ahead; 0 = auto(1); -1 = off */
  int32_t reserved[2];     /* [0]: ignored by the shipped library (bench-experiment knobs that exist only in the
                                   separate -DPNA_AMD_EXPERIMENTS build of tools/build_experiments.sh);
                              [1]: 1 = force the compiler-scheduled kernel instead of the hand-scheduled one;
                                   2 = REQUIRE the hand-scheduled kernel (PNA_E_INVALID if the call does not qualify):
                                   a work list that covers only part of the rows is only honoured by that kernel */
} pna_tuning;

/*
 * Fused gather + multi-aggregator segment-reduce + degree scalers over a destination-sorted CSR.
 *
 * Replaces, in one launch: DGL `update_all(message, reduce_func)` with
 *   reduce_func            models/dgl/pna_layer.py:45-50 (PNATower) and :189-194 (PNASimpleLayer)
 *   the aggregators        models/dgl/aggregators.py:6-26,50-51
 *   the scalers            models/dgl/scalers.py:7-19
 * and, for the dense variant, models/pytorch/pna/layer.py:43-44 with aggregators.py:17-84 and
 * scalers.py:7-38 (edge_weight = adjacency weight).
 *
 * Message of CSR edge k (k in [rowptr[v], rowptr[v+1]) for destination v):
 *     m_k = x[col[k]]  (+ dst_term[v])  (+ edge_term[k])             each an F-vector
 *   col == NULL  -> x is edge-resident, m_k = x[k] (already-materialised per-edge messages, e.g. the
 *                   output of a multi-layer pretrans MLP, models/dgl/pna_layer.py:35-40);
 *   dst_term/edge_term are the h_dst / edge-feature halves of a 1-layer (affine) pretrans that has
 *   been factorised to node level: W[h_src|h_dst|ef]+b = (W_a h_src) + (W_b h_dst + b) + (W_e ef).
 * Output row v, block (s * n_aggr + a), feature f:
 *     out[v*ldo + (s*n_aggr + a)*block_stride + f] = aggr[a](m_k : k in row v)[f] * row_scale[s][v]
 *   (row_scale[s] == NULL means the identity scaler).  Scaler-major / aggregator-minor is the
 *   reference's concatenation order (pna_layer.py:48-49).
 * Rows with no in-edges get 0 in every block (DGL's zero initialiser; undefined in the reference).
 * Arithmetic: fp32, no FMA contraction, sums in CSR edge order (hub rows: per 128-edge segment, then over segments).
 * mean = s / D and E[x^2] = q / D are the reference's divisions bit for bit (one IEEE division per row for 1 / D, then
 * Markstein's fma correction per feature; round 1 used s * (1/D), 1 ulp off, which the cancellation in
 * std = sqrt(E[x^2] - E[x]^2 + 1e-5) amplifies to percents when all neighbours are equal); max, min and the degree
 * scalers are bit-exact; mean / std differ from the reference only through the summation ORDER of s and q.
 * edge_weight (nullable, [E]): mean/sum/std/var use sum_k w_k m_k and D = sum_k w_k; max/min use
 *   only edges with w_k > 0.
 * argmax/argmin (nullable, (V, ld_arg) int32): CSR edge position k of the first max / min, -1 for
 *   empty rows -- what the backward pass needs.
 *
 * Heavy rows: destinations with more than `heavy_threshold` in-edges are not walked by one lane
 * group; they are listed in `heavy_rows` and cut into segments of `seg_len` edges that are reduced
 * in parallel into `partials` and combined in segment order by a second small kernel, so results do
 * not depend on the launch geometry.  heavy_threshold <= 0 disables the split.
 */
typedef struct pna_segreduce_args {
  uint32_t struct_size;    /* sizeof(pna_segreduce_args) of the CALLER's header (ABI 19): a shorter struct is refused with PNA_E_INVALID */
  uint32_t _abi_reserved;  /* 0 */
  const int32_t* rowptr; /* [V+1] */
  const int32_t* col;    /* [E] or NULL */
  int32_t V;
  int32_t F;
  const float* x;
  int64_t ldx;
  int64_t x_rows;        /* rows of x (source nodes, or E when col == NULL); 0 = unknown (64-bit addressing) */
  const float* dst_term; /* nullable (V, ld_dst) */
  int64_t ld_dst;
  const float* edge_term; /* nullable (E, ld_edge) */
  int64_t ld_edge;
  const float* edge_weight; /* nullable [E] */

  /* Towers: the same graph reduced over n_tower independent feature slices of width F in one launch
   * (the reference loops over towers in Python, models/dgl/pna_layer.py:133-139).  Tower t reads
   * columns [t*tower_stride_in, +F) of x / dst_term / edge_term (and argmax/argmin) and writes its
   * n_scaler*n_aggr blocks starting at column t*tower_stride_out of out.  n_tower <= 1: one slice,
   * strides ignored. */
  int32_t n_tower;
  int32_t _pad_t;
  int64_t tower_stride_in;
  int64_t tower_stride_out;

  int32_t n_aggr;
  int32_t aggr[PNA_MAX_AGGR];
  int32_t n_scaler;
  int32_t _pad0;
  const float* row_scale[PNA_MAX_SCALER]; /* each nullable, [V] */

  float* out;
  int64_t ldo;
  int32_t block_stride;
  int32_t _pad1;
  int32_t* argmax; /* nullable */
  int32_t* argmin; /* nullable */
  int64_t ld_arg;

  /* heavy-row schedule (all optional; produced once per graph by the host, see pna_amd/graph.py) */
  int32_t heavy_threshold;   /* rows with degree > threshold are skipped by the main walk          */
  int32_t seg_len;           /* edges per heavy segment                                            */
  int32_t n_heavy;           /* number of heavy rows                                               */
  int32_t n_seg;             /* total number of segments                                           */
  const int32_t* heavy_rows; /* [n_heavy] row ids, ascending                                       */
  const int32_t* heavy_segptr; /* [n_heavy+1] first segment of each heavy row                      */
  const int32_t* seg_heavy;  /* [n_seg] index into heavy_rows of the row each segment belongs to   */
  float* partials;           /* workspace, pna_segreduce_partials_bytes(n_seg, F, n_tower) bytes   */

  /* Optional work list for the hand-scheduled kernel (used when the call is the 4-aggregator gather
   * mean|max|min|std with the identity scaler, messages x[col[k]] or x[col[k]] + dst_term[v] -- i.e. what
   * PNASimpleLayer and the PNATower layers issue; ABI 14: also with edge_term, per edge or per edge type): n_work_items records {row, beg, end, slot} (int32 x 4).  slot < 0:
   * the record is a whole row [beg,end) = [rowptr[row], rowptr[row+1]) whose result is finalised and stored;
   * slot >= 0: the record is heavy segment number `slot` (its partials go to partials[slot]; the segments of a
   * heavy row must be the ones heavy_segptr describes).  Every row must be covered exactly once, either by one
   * whole-row record or by its segments.  The host orders the records by length (pna_amd/graph.py) so that the
   * lane groups of a wavefront walk equally long records; results do not depend on the order.  NULL = the
   * compiler-scheduled kernel in natural row order. */
  const int32_t* work_items;
  int32_t n_work_items;
  int32_t _pad2;
  int64_t n_edges; /* = rowptr[V] (length of col); required with work_items */
  pna_tuning tune;
  /* ABI 9: where the aggregate of heavy row i (heavy_rows[i]) is written, as a row index of `out` (nullable: the row itself).
   * Together with work_items whose `row` field holds the OUTPUT row of a whole-row record (the hand-scheduled kernel uses that
   * field for nothing else when there is no dst_term) this lets a caller have the aggregate written in any row order -- e.g.
   * grouped by in-degree for pna_posttrans_args.row_perm.  `out` must then have as many rows as the largest index + 1. */
  const int32_t* heavy_out_rows;
  /* ABI 12: with dst_term (where the work list's `row` must stay the node), the row of `out` that receives node v's aggregate:
   * out_row_of[v] for whole-row records (heavy rows: heavy_out_rows).  Nullable; hand-scheduled kernel with work_items only. */
  const int32_t* out_row_of;
  /* ABI 14: edge terms from a table.  When the edge features are an embedding of an edge TYPE (the molecule nets: bond type,
   * realworld_benchmark/nets/molecules_graph_regression/pna_net.py with --edge_feat True), W_e . ef has one distinct row per type:
   * edge_term is then (n_edge_types, ld_edge) and the term of CSR edge k is row edge_type[k] (int32 [E], CSR order).  NULL: edge_term
   * has one row per edge.  n_edge_types <= 4 keeps the table in registers of the hand-scheduled kernel. */
  const int32_t* edge_type;
  int32_t n_edge_types;
  int32_t _pad3;
} pna_segreduce_args;

/* Launches the kernels described above on `stream`. */
int pna_segreduce_fwd_f32(const pna_segreduce_args* args, pna_stream_t stream);

/* Bytes of `partials` workspace needed for n_seg heavy segments of n_tower slices of width F. */
int64_t pna_segreduce_partials_bytes(int32_t n_seg, int32_t F, int32_t n_tower);

/*
 * Backward of pna_segreduce_fwd_f32 (what autograd derives through the reference's per-bucket torch ops,
 * models/dgl/aggregators.py:6-26).  `gagg` is the gradient w.r.t. the UNSCALED aggregates in the forward's
 * identity-scaler layout: row v, tower t, aggregator i (aggr[i]) at column t*tower_stride_g + i*F of an
 * (V, ld_g) matrix (the caller folds the degree scalers in: G_i = sum_s row_scale[s] * dOut[s, i]).
 * The kernel rebuilds every message m_k = x[col[k]] (+ dst_term[v]) (+ edge_term[k]) and scatters
 *     dL/dm_k = G_mean/D + G_sum + [k = argmax] G_max + [k = argmin] G_min
 *             + (G_var + G_std/(2 std)) [var > 0] (2/D) (m_k - mean)
 * into grad_x[col[k]] (fp32 hardware atomics: zero-fill grad_x first; plain stores when col == NULL),
 * grad_dst[v] (sum over the row's edges; zero-fill first) and grad_edge[k].  Any of the three may be NULL.
 * mean / stdv / var are the forward results (identity-scaled), (V, ld_stat) with tower stride
 * tower_stride_stat -- needed only when std or var is among aggr[] (stdv or var, one is enough);
 * argmax / argmin are the forward's (V, ld_arg) outputs, tower stride tower_stride_in.
 * x / dst_term / edge_term are only read when std or var is among aggr[].
 */
typedef struct pna_segreduce_bwd_args {
  uint32_t struct_size;    /* sizeof(pna_segreduce_bwd_args) of the CALLER's header (ABI 19): a shorter struct is refused with PNA_E_INVALID */
  uint32_t _abi_reserved;  /* 0 */
  const int32_t* rowptr;
  const int32_t* col; /* nullable: x edge-resident */
  int32_t V;
  int32_t F;
  const float* x;
  int64_t ldx;
  const float* dst_term;
  int64_t ld_dst;
  const float* edge_term;
  int64_t ld_edge;
  int32_t n_tower;
  int32_t n_aggr;
  int64_t tower_stride_in;
  int32_t aggr[PNA_MAX_AGGR];
  const float* gagg;
  int64_t ld_g;
  int64_t tower_stride_g;
  const float* mean;
  const float* stdv;
  const float* var;
  int64_t ld_stat;
  int64_t tower_stride_stat;
  const int32_t* argmax;
  const int32_t* argmin;
  int64_t ld_arg;
  float* grad_x;
  int64_t ld_gx;
  float* grad_dst;
  int64_t ld_gd;
  float* grad_edge;
  int64_t ld_ge;
  int32_t heavy_threshold;
  int32_t seg_len;
  int32_t n_heavy;
  int32_t n_seg;
  const int32_t* heavy_rows;
  const int32_t* heavy_segptr;
  const int32_t* seg_heavy;
  const int32_t* stat_row_of;  /* nullable [V] (pna_segreduce_bwd_rowprep_f32 / _pull_f32 only): mean / stdv / var / argmax / argmin hold node
                                * v's values in row stat_row_of[v] -- a forward that wrote them in a degree plan's row order */
  const int32_t* stat_node_of; /* nullable [stat_rows], the inverse map (takes precedence): the rowprep pass then WALKS the statistics' rows in
                                * their own order -- row r belongs to node stat_node_of[r], < 0 = a padding row -- and touches the per-node
                                * tensors (gagg, the table) at node rows: sequential reads of the big tensors, every node exactly once */
  int64_t stat_rows;
} pna_segreduce_bwd_args;

int pna_segreduce_bwd_f32(const pna_segreduce_bwd_args* args, pna_stream_t stream);

/*
 * Pull formulation of the same backward for messages WITHOUT a per-edge term (m_k = x[col_k] + dst_term[v]):
 *     dL/dm_k = R1[v] + R2[v] * x[col_k] + [k = argmax] G_max[v] + [k = argmin] G_min[v]
 *     R2 = (G_var + G_std / (2 std)) [var > 0] (2/D),   R1 = G_mean/D + G_sum + R2 * (dst_term[v] - mean[v])
 * so grad_x[u] = sum_{out-edges (u,v)} R1[v] + x[u] * sum_{out-edges} R2[v] + the max/min terms: the two sums are
 * pna_segreduce_fwd_f32 ("sum", 2 * n_tower towers) over the TRANSPOSED graph on the table written by
 * pna_segreduce_bwd_rowprep_f32 -- table[v] = [R1 (n_tower*F) | R2 (n_tower*F, only if std/var is among aggr[])] --
 * and pna_segreduce_bwd_argscatter_f32 adds the max/min terms with V*n_tower*F atomics each (instead of one atomic
 * per edge and feature).  rowprep also writes grad_dst[v] = D*(G_mean/D + G_sum) + G_max + G_min when grad_dst != NULL
 * (the variance term sums to zero over a row).  Same args struct; fields the formulation does not need are ignored.
 */
int pna_segreduce_bwd_rowprep_f32(const pna_segreduce_bwd_args* args, float* table, int64_t ld_table, pna_stream_t stream);
int pna_segreduce_bwd_argscatter_f32(const pna_segreduce_bwd_args* args, pna_stream_t stream);

/* ABI 16: the same gradient with the max / min terms INSIDE the pull -- no atomics except for the segments of hub SOURCE rows.
 * After pna_segreduce_bwd_rowprep_f32 (table = [R1 | R2]) -- or with run_rowprep set, instead of it:
 *   grad_x[u] = sum over out-edges (u -> v), the k-th in-edge of v, of  R1[v] + [k = argmax[v] - rowptr[v]] G_max[v] + [k = argmin[v] - rowptr[v]] G_min[v]
 *               + x[u] * sum R2[v]
 * base: rowptr (forward CSR), argmax / argmin, gagg / aggr[] (max, min and std or var among them), n_tower, F, V, x (the source
 * table) and grad_x (n_src rows; rows of hub sources -- work-list records with slot >= 0 -- must be ZERO on entry, every other row is
 * overwritten).  col_t / rank_t: destination v and in-list rank k of every edge of the TRANSPOSED CSR (edges sorted by source);
 * items_t: its work list {source row, beg, end, slot} (slot < 0: whole row; slot >= 0: a segment, added atomically).  ranks:
 * workspace (V, ld_rank >= 2 T F) of uint16 -- in-degrees up to 65534.  4 <= F <= 256.  * PACKED rows (round 4): with run_rowprep != 0, ld_table >= 5 T F and ranks == (uint16_t*)(table + 4 T F), the rowprep pass also
 * copies G_max | G_min to table[v][2 T F .. 4 T F) and the pull reads ONE contiguous row [R1 | R2 | G_max | G_min | ranks] per
 * out-edge (12 cache lines at F = 75 with a 1536-byte pitch instead of ~14.7 for three separate pieces); ld_rank then = 2 ld_table.
 * IN PLACE: packed rows with base->gagg == table, one tower and base->aggr[] = {mean, std, max, min} (ld_g == ld_table): the caller's
 * d agg contraction has written [G_mean | G_std | G_max | G_min] straight into the rows; rowprep overwrites the first two blocks
 * with R1 | R2 and copies nothing.
 */
typedef struct pna_segreduce_bwd_pull_args {
  uint32_t struct_size;    /* sizeof(pna_segreduce_bwd_pull_args) of the CALLER's header (ABI 19): a shorter struct is refused with PNA_E_INVALID */
  uint32_t _abi_reserved;  /* 0 */
  const pna_segreduce_bwd_args* base;
  const float* table;
  int64_t ld_table;
  const int32_t* col_t;
  const int32_t* rank_t;
  const int32_t* items_t;
  int32_t n_items_t;
  int32_t run_rowprep;   /* != 0: `table` is a WORKSPACE; the call runs the rowprep pass itself (table, base->grad_dst and the ranks in one sweep) */
  uint16_t* ranks;
  int64_t ld_rank;
  /* round 6 (ABI 21), optional: the pull over PER-EDGE rows.  edge_rows != NULL: a (n_edges, ld_edge >= F) workspace; the call first walks the
   * FORWARD work list `items` (n_items records {row, beg, end, slot}, pna_segreduce_args.work_items) and writes, in CSR order,
   *     P[e] = R1[v] + [e = argmax[v]] G_max[v] + [e = argmin[v]] G_min[v]        and R2[v] into table (V, ld_table >= F),
   * then pulls per out-edge P[pos_t[j]] and R2[col_t[j]] (pos_t: position of every transposed edge in the forward CSR): 2 x 4F bytes per
   * edge instead of 20F + ranks; the same additions in the same order: bit-identical gradients.  One tower, no dst_term, aggr[] = mean, std,
   * max, min (any order); rank_t / ranks / run_rowprep are not used. */
  float* edge_rows;
  int64_t ld_edge;
  const int32_t* pos_t;
  const int32_t* items;
  int32_t n_items;
  int32_t _pad_e;
} pna_segreduce_bwd_pull_args;
int pna_segreduce_bwd_pull_f32(const pna_segreduce_bwd_pull_args* args, pna_stream_t stream);

/*
 * Per-row degree scalers of the DGL variant -- models/dgl/scalers.py:12-19 evaluated with the
 * reference's exact fp32 rounding sequence (np.log in float64, rounded to fp32, then
 * Tensor.__rtruediv__ = reciprocal()*scalar for amplification and a true division for attenuation):
 *     amp[v] = fl32( fl32(1/avg_log) * fl32(log(D_v + 1)) ),   att[v] = fl32( avg_log / fl32(log(D_v + 1)) )
 * with D_v = rowptr[v+1]-rowptr[v].  Rows with D_v = 0 get amp = 0, att = 0 (their aggregates are 0).
 * amp / att nullable.
 */
int pna_degree_scalers_f32(const int32_t* rowptr, int32_t V, float avg_log, float* amp, float* att,
                           pna_stream_t stream);

/*
 * Post-aggregation tower contraction on the fp32 matrix cores -- replaces the `posttrans` nn.Linear
 * applied to the concatenated [self | scaler-major aggregate] row (models/dgl/pna_layer.py:65-68,
 * :206; models/pytorch/pna/layer.py:47-48) WITHOUT materialising the (V, A*S*F) operand, and -- in
 * eval mode -- the elementwise tail that follows it (graph-norm :71-72, BatchNorm with running stats
 * :73-74 / :209-210, ReLU :211, residual :212-213):
 *     z[v]  = bias + W_self . h[v] + sum_s row_scale[s][v] * ( W_s . a[v] )
 *     y[v]  = residual[v] + act( (z[v] * row_post[v]) * col_scale + col_shift )
 * where `a` is the (M, K = n_aggr*F)-wide identity-scaler output of pna_segreduce_fwd_f32 and W_s the
 * column block of the reference weight that multiplies scaler s; row_post / col_scale+col_shift /
 * relu / residual are each optional (NULL / 0 = skipped).  Uses v_mfma_f32_16x16x4_f32 (exact fp32
 * products, fp32 accumulate).  K >= 4 and (when h is given) Kh >= 4: narrower operands are padded with zero
 * columns by the caller (pna_amd/functional.py does).
 *
 * The weight is consumed in a packed, zero-padded tile image produced once per weight update by
 * pna_posttrans_pack_f32 from the reference layout: w_ref is the nn.Linear weight, (N, ldw_ref)
 * row-major, input columns ordered [h (Kh) | scaler 0 (K) | scaler 1 (K) | ...] exactly as the
 * reference concatenates them (pna_layer.py:48-49,:65).  pna_posttrans_packed_floats returns the
 * number of floats of w_img (and of wh_img through *wh_floats).
 */
int64_t pna_posttrans_packed_floats(int32_t K, int32_t N, int32_t n_scaler, int32_t Kh, int64_t* wh_floats);
int pna_posttrans_pack_f32(const float* w_ref, int64_t ldw_ref, int32_t N, int32_t K, int32_t n_scaler, int32_t Kh,
                           float* w_img, float* wh_img /* nullable when Kh == 0 */, pna_stream_t stream);

typedef struct pna_posttrans_args {
  uint32_t struct_size;    /* sizeof(pna_posttrans_args) of the CALLER's header (ABI 19): a shorter struct is refused with PNA_E_INVALID */
  uint32_t _abi_reserved;  /* 0 */
  const float* a;   /* (M, lda), K columns used */
  int64_t lda;
  int32_t M;
  int32_t K;
  int32_t N;
  int32_t n_scaler; /* 1..5 */
  const float* row_scale[PNA_MAX_SCALER]; /* each [M] or NULL = identity */
  const float* w_img;  /* packed aggregate weight */
  const float* h;      /* nullable (M, ldh): the node's own features (tower variant) */
  int64_t ldh;
  int32_t Kh;
  int32_t relu;        /* activation after the affine column map: 0 = none, 1 = ReLU, 2 = LeakyReLU (act_slope) */
  const float* wh_img; /* packed self weight (when h != NULL) */
  const float* bias;   /* nullable [N] */
  const float* row_post;  /* nullable [M]: graph-norm factor snorm_n */
  const float* col_scale; /* nullable [N]: gamma / sqrt(running_var + eps)            (eval BatchNorm) */
  const float* col_shift; /* nullable [N]: beta - running_mean * col_scale            (with col_scale) */
  const float* residual;  /* nullable (M, ld_res): added after the activation */
  int64_t ld_res;
  float* y;            /* (M, ldy), N columns */
  int64_t ldy;
  int32_t pipeline;    /* pna_posttrans_x3_f32 only: 0 = library default; 2 = two weight buffers in LDS, one barrier at every
                          chunk boundary; 3 = three weight buffers, one barrier in the middle of every chunk (wavefronts
                          cross chunk boundaries unsynchronised).  Same arithmetic, bit-identical results. */
  float act_slope;     /* relu == 2: LeakyReLU, act(v) = v < 0 ? act_slope * v : v (the FCLayer 'LeakyReLU' of the tower layers'
                          mixing network, models/layers.py:157, slope 0.01); ignored for relu == 0 / 1 */
  /* Towers (models/dgl/pna_layer.py:133-139 runs the towers' posttrans one after the other): n_tower > 1 evaluates n_tower
   * independent contractions of the same shape in ONE call.  Tower t reads a + t*tower_stride_a and h + t*tower_stride_h
   * (0 = all towers share h), the images w_img + t*tower_stride_w / wh_img + t*tower_stride_wh (floats for pna_posttrans_f32,
   * BYTES for pna_posttrans_x3_f32: t-th image of a buffer of n_tower images packed one after the other), bias / col_scale /
   * col_shift + t*N, and writes y + t*tower_stride_y (column slices of one row-major matrix).  row_scale / row_post are shared;
   * residual must be NULL.  n_tower <= 1: one contraction, strides ignored. */
  int32_t n_tower;
  int32_t _pad_t;
  int64_t tower_stride_a;
  int64_t tower_stride_h;
  int64_t tower_stride_w;
  int64_t tower_stride_wh;
  int64_t tower_stride_y;
  /* Degree-grouped rows (ABI 9, pna_posttrans_x3_f32 only; NULL = off).  Every PNA scaler is a function of the in-degree
   * alone (models/dgl/scalers.py:7-19), so for the rows of ONE degree D the three scaler blocks of the posttrans weight
   * collapse into one:  sum_s scale_s(D) (W_s a) = (sum_s scale_s(D) W_s) a  -- a third of the multiply-adds.  The caller
   * orders the rows by degree: `a`, `row_scale[]`, `row_post` are indexed by a VIRTUAL row v in [0, M) (M a multiple of the
   * workgroup tile R = 128 rows for n_scaler = 1, 192 for n_scaler = 3: every degree group padded to whole tiles), row_perm[v] is
   * the row of `y` / `residual` that virtual row v is (or -1: padding, nothing is stored), and tile t (virtual rows [R t, R t + R)) multiplies by the weight image number
   * tile_image[t] of a buffer of images packed one after the other, image_stride BYTES apart (pna_posttrans_x3_pack_f32 of a
   * (G * 80, K) matrix whose rows [80 g, 80 g + N) are group g's combined weight: G column blocks = G images).  tile_image
   * NULL: every tile uses w_img as it is (the rows no group holds, with their per-row scalers: n_scaler = 3).
   * Needs N <= 80 (one 80-column block whatever N: the grouped kernel is bound by its A stream, not by the matrix pipe), no h
   * panel, n_tower <= 1, n_scaler 1 or 3 -- or 80 < N <= 128 with n_scaler = 1 (one 128-column
   * block; its images are packed one by one, pna_posttrans_x3_pack_f32 of a (128, K) matrix each, image_stride =
   * pna_posttrans_x3_packed_bytes(K, 128, 1, 0); the caller sends the rows no group holds through an ordinary call).
   * pna_segreduce_args.work_items (row = output row) and heavy_out_rows write the aggregate in that virtual order in the
   * first place. */
  const int32_t* row_perm;
  const int32_t* tile_image;
  int64_t image_stride;
} pna_posttrans_args;

int pna_posttrans_f32(const pna_posttrans_args* args, pna_stream_t stream);

/* ---- the same contraction on the bf16 matrix pipe at fp32 accuracy ("bf16x3") ---------------------
 *
 * Same operation, arguments and epilogue as pna_posttrans_f32 (n_scaler <= 3).  Every fp32 operand is
 * cut exactly into three bf16 terms (8+8+8 mantissa bits, by truncation) and each product is evaluated
 * as its six partial products of weight >= 2^-16, accumulated in fp32 by v_mfma_f32_16x16x32_bf16; the
 * dropped partial products are below 2^-23 of the product (the size of one fp32 rounding).  Inputs and
 * outputs stay fp32; results agree with pna_posttrans_f32 to fp32 summation-order noise.
 * Non-finite operands: NaN propagates; a +-Inf in `a` / `h` / the weight keeps its row's results +-Inf / NaN exactly as
 * in pna_posttrans_f32 when the other operand's three bf16 terms are all non-zero (any generic fp32 value), and gives
 * NaN instead of +-Inf where the other operand is exactly bf16-representable (its residual terms are 0: Inf * 0) --
 * non-finite wherever the fp32 contraction is non-finite, never finite garbage (before the ReLU of the epilogue,
 * which maps -Inf to 0 but keeps NaN).  Subnormal bf16 terms (operands below 2^-126, residual terms of operands below
 * ~2^-110) may be flushed by the matrix pipe: an absolute error of at most 2^-126 * |other operand| per product.
 * tests/test_gpu_posttrans_x3.py pins all of this (mixed 1e-20..1e20 magnitudes, cancellation between scaler blocks,
 * K = 900 equal-sign sums, subnormals, single infinities).  w_img / wh_img of the args are the images made by pna_posttrans_x3_pack_f32 (a
 * different format from pna_posttrans_pack_f32's; sizes in BYTES from pna_posttrans_x3_packed_bytes).
 */
int64_t pna_posttrans_x3_packed_bytes(int32_t K, int32_t N, int32_t n_scaler, int32_t Kh, int64_t* wh_bytes);
int pna_posttrans_x3_pack_f32(const float* w_ref, int64_t ldw_ref, int32_t N, int32_t K, int32_t n_scaler, int32_t Kh,
                              void* w_img, void* wh_img /* nullable when Kh == 0 */, pna_stream_t stream);
int pna_posttrans_x3_f32(const pna_posttrans_args* args, pna_stream_t stream);

/* ---- fused PNASimpleLayer forward (inference) -----------------------------------------------------
 *
 * One launch for the whole of models/dgl/pna_layer.py:197-216 with aggregators "mean max min std":
 *   a[v]  = [mean | max | min | std] over in-edges of x[col[e]]          (:168-187, :189-194)
 *   y[v]  = residual[v] + relu((bias + sum_s row_scale[s][v] * (W_s . a[v])) * col_scale + col_shift)
 * The (V, 4F) aggregate never reaches HBM: it is built per 32-row tile in LDS and contracted from there.
 * w_img is the pna_posttrans_pack_f32 image of the Linear weight re-laid with each aggregator block
 * zero-padded from F to B4 = round_up(F, 4) input columns (K = 4*B4 per scaler, Kh = 0).
 * Rows with degree > heavy_threshold (0 = never) are reduced by the whole workgroup.
 * Supported: 4 <= F <= 80, N <= 80, n_scaler <= 3.  Same numerics as the two-kernel path except for the
 * summation order over hub rows' edges and over k (tolerance-level, not bit-level, agreement).
 */
typedef struct pna_fused_simple_args {
  uint32_t struct_size;    /* sizeof(pna_fused_simple_args) of the CALLER's header (ABI 19): a shorter struct is refused with PNA_E_INVALID */
  uint32_t _abi_reserved;  /* 0 */
  const int32_t* rowptr; /* [V+1] CSR by destination */
  const int32_t* col;    /* [E] source row of x per edge */
  const float* x;        /* (x_rows, ldx) source features */
  int64_t ldx;
  int32_t V;
  int32_t F;
  int32_t N;
  int32_t n_scaler;
  const float* row_scale[PNA_MAX_SCALER]; /* each [V] or NULL = identity */
  const float* w_img;
  const float* bias;      /* nullable [N] */
  const float* col_scale; /* nullable [N] */
  const float* col_shift; /* nullable [N] (with col_scale) */
  const float* residual;  /* nullable (V, ld_res) */
  int64_t ld_res;
  float* y;               /* (V, ldy) */
  int64_t ldy;
  int32_t relu;
  int32_t heavy_threshold;
} pna_fused_simple_args;

int pna_fused_simple_f32(const pna_fused_simple_args* args, pna_stream_t stream);

/* ---- PNASimpleLayer forward on degree-ordered rows: gather + scalers + posttrans in ONE kernel (ABI 10) ------------------
 *
 * Replaces, for inference on large graphs, the whole of models/dgl/pna_layer.py:197-216 with aggregators "mean max min std"
 * (reduce_func :189-194, posttrans :206, BatchNorm :209-210, ReLU :211, residual :212-213).  The (V, 4F) aggregate -- the
 * reference materialises (V, 12F) -- never reaches HBM.  Every PNA scaler is a function of the destination's in-degree alone
 * (models/dgl/scalers.py:7-19), so all rows of one in-degree D share ONE combined weight  W_D = sum_s scale_s(D) W_s :
 *
 *   a[v]        = [mean | max | min | std] over the in-edges of x[src]                     (bit-identical to pna_segreduce_fwd_f32
 *                                                                                             for rows one lane group walks alone)
 *   y[perm[v]]  = residual[perm[v]] + act((bias + W_D(v) . a[v]) * col_scale + col_shift)   (fp32 in and out, fp32 accumulation; see ARITHMETIC)
 *
 * ARITHMETIC (`arith`, ABI 21).  PNA_FD_ARITH_X3: every statistic and every weight as three bf16 terms, six partial products (the bf16x3
 * arithmetic of pna_posttrans_x3_f32, rounds 3-4): componentwise fp32-accurate for operands of any dynamic range.  PNA_FD_ARITH_H2 (round 5):
 * TWO fp16 terms, x = h0 + h1 + r with h0 = fp16(x), h1 = fp16(x - h0) (round to nearest), and three partial products (h1 w0, h0 w1, h0 w0;
 * v_mfma_f32_16x16x32_f16, fp32 accumulation): half the matrix instructions and two thirds of the weight stream.  fp16's range is narrow, so
 * the operands are first multiplied by powers of two (exact): row v's statistics by 2^s(v), chosen in the kernel so that twice the row's
 * largest message magnitude lies in [2^14, 2^15), and column n of the weights by 2^t(n), chosen by the pack function from the largest
 * |W_D[n][k]| over k and over all images into [2^13, 2^14) (2^-t(n) rides in each image's tail); the accumulator is multiplied by 2^-(s + t)
 * in the bias' fma.  In these units |r| <= max(2^-22 |x|, 2^-25): an operand below 2^-3 -- more than 2^17 below its row's (2^16 below its
 * column's) bound -- sits on fp16's subnormal grid with its second term and carries an ABSOLUTE error of up to 2^-25, a relative one of
 * 2^-25 / |x|: the form is normwise-, not componentwise-accurate (a statistic 1e7 x the rest of its row with a zero weight on it: 5.6e-5
 * relative on the outputs; at 1e13 the small statistics vanish).  PNA_FD_ARITH_GUARDED (the default) = the same fp16 x 2 launch with that
 * FLOOR error bounded per output: the kernel adds up the exact floor errors fs(v) of a row's statistics (the part of |r| above 2^-22 |x|;
 * zero for almost every row), the pack function those of a column's weights fw(n), and an output is certified when
 *     |acc| >= 2^34 fs(v) + 2^35 fw(n)          (floor error <= 2^-20 |acc|: everything else of the split's error is relative term by term);
 * a 64-row (wide shapes: 64-row) workgroup tile with an uncertified output is appended to a device-side list and a second launch computes
 * exactly those tiles again in bf16 x 3 (no host round trip; on Gaussian features and weights nothing is handed over, the launch finds an
 * empty list and costs ~6 us).  In tower mode the row's scale also covers its own x_dst / h_self strips: it is lowered when they arrive,
 * mid-row, and the accumulator with it (exact).  Measured against float64 on BASELINE configs[2] / [4] shapes: 1.25 x the error of bf16x3, a
 * fifth of an fp32 GEMM's.
 * Non-finite operands: NaN propagates; a row that holds an infinite statistic (column: an infinite weight) is non-finite in every output
 * it reaches, as in fp32 -- but its FINITE elements are scaled against FLT_MAX and may underflow, so where fp32 gives +-Inf the result
 * may be +-Inf or NaN, never finite garbage (such rows are not the guard's: a non-finite accumulator certifies itself).
 *
 * The caller (pna_amd/degree_groups.py) orders the rows by in-degree into VIRTUAL rows v in [0, M):
 *   row_perm[v]   node of virtual row v, or -1 = padding (nothing is stored); M a multiple of pna_fused_degree_tile_rows(F, N) (64 or 128); every aligned block of 16
 *                 virtual rows holds rows of ONE in-degree (padding aside), every aligned block of 64 rows uses ONE weight image;
 *   tile_desc     [M / 16][4] = {first record, in-degree D, weight image, 0} of every 16-row block;
 *   tile_ids      n_records records of 16 int32: record (first + e)[i] = source row (row of x) of the e-th in-edge of the block's
 *                 i-th row, e in [0, D), in the row's edge order; a padding row repeats the block's first row; a block owns
 *                 max(4, round_up(D, 4)) records, the ones past D being copies of record D - 1 (D = 0: any valid row);
 *   w_img         n_img images, image_stride == pna_fused_image_bytes(F, N, tower, 0) bytes apart (the pack function's layout), from pna_fused_pack_f32
 *                 (w_img_x3 / image_stride_x3: the bf16 x 3 images, x3 = 1):
 *                 w_ref is the Linear weight (N, n_scaler * 4F) in the reference's column order [scaler][aggregator][feature]
 *                 (pna_layer.py:192-193), scale (n_img, n_scaler) the scalers' values for image i (NULL with n_scaler = 1:
 *                 the weight itself); W_D is formed in fp32 in scaler order, like the reference's blocks.
 * x: (x_rows, ldx) rows, 16-byte aligned, ldx % 4 == 0, ldx >= round_up(F, 8) (round_up(F, 4) when F % 32 is in 1..16),
 * x_rows < 2^24, table < 4 GiB; y and residual: n_nodes rows, each table < 4 GiB.
 * F in 17..80, or (ABI 15; 97..112 since ABI 21) 97..128 -- the features in two gather passes (BASELINE configs[4]: 128 -> 128);
 * N in 4..80, or (ABI 15) 81..128 -- the output columns in two panels of 64 (needs F in 49..64 or 97..128).  pna_fused_degree_image_bytes(F, N) > 0
 * is the authoritative test.
 * relu: 0 none / 1 ReLU / 2 LeakyReLU(act_slope).  agg_out (nullable): (M, ld_agg) receives the
 * statistics the contraction consumed, [mean | max | min | std] x F per virtual row (verification; a slower instantiation).
 * Rows of degrees too rare to fill a tile, and hub rows, are the caller's: pna_segreduce_fwd_f32 + pna_posttrans_x3_f32 over
 * their compact list.
 *
 * TOWER MODE (ABI 13; x_dst, h_self and row_post non-null together): PNALayer.forward with ONE tower, 1-layer pretrans / posttrans,
 * no edge features (models/dgl/pna_layer.py:33-76, :130-145), inference.  The pretrans Linear of [h_u | h_v] is x_src[u] + x_dst[v]
 * (two node-level projections, the caller's): x = x_src, and with a[v] the statistics of x_src over the in-edges as above
 *
 *   y[perm[v]] = residual + act(((bias + W_D . a[v] + [deg > 0] (W_D,mean + W_D,max + W_D,min) . x_dst[perm[v]] + W_self . h_self[perm[v]])
 *                                 * row_post[v]) * col_scale + col_shift)
 *
 * -- mean / max / min of (a_u + b) are those of a_u plus b (max / min: the same bits, rounding is monotone), the std does not see
 * the shift.  x_dst, h_self: (n_nodes, ld) tables in node order under x's alignment rules; row_post: [M] in VIRTUAL row order
 * (graph norm; ones without); residual additionally 16-byte aligned with ld_res % 4 == 0.  49 <= F <= 80.  The images come from
 * pna_fused_tower_pack_f32: w_ref (N, n_scaler * 5F) in scaler blocks [4F aggregators | F self panel (block 0 only)] -- the
 * posttrans Linear with its h columns moved behind the first block, or posttrans . BatchNorm . mixing Linear collapsed into one
 * weight (pna_amd/functional.py::_tower_collapsed_weights).  No agg_out in this mode.
 */
typedef struct pna_fused_degree_args {
  uint32_t struct_size;    /* sizeof(pna_fused_degree_args) of the CALLER's header (ABI 19): a shorter struct is refused with PNA_E_INVALID */
  uint32_t _abi_reserved;  /* 0 */
  const int32_t* tile_desc;
  const int32_t* tile_ids;
  int64_t n_records;
  const float* x;
  int64_t ldx;
  int64_t x_rows;
  int32_t F;
  int32_t N;
  const int32_t* row_perm;
  int64_t M;
  int64_t n_nodes;        /* rows of y / residual */
  const void* w_img;
  int64_t image_stride;
  const float* bias;      /* nullable [N] */
  const float* col_scale; /* nullable [N] */
  const float* col_shift; /* nullable [N] (with col_scale) */
  const float* residual;  /* nullable (n_nodes, ld_res), node order */
  int64_t ld_res;
  float* y;               /* (n_nodes, ldy), node order */
  int64_t ldy;
  int32_t relu;
  float act_slope;
  float* agg_out;         /* nullable (M, ld_agg) */
  int64_t ld_agg;
  const float* x_dst;     /* tower mode (ABI 13): (n_nodes, ld_xdst) */
  int64_t ld_xdst;
  const float* h_self;    /* (n_nodes, ld_h) */
  int64_t ld_h;
  const float* row_post;  /* [M], virtual row order */
  int32_t spare_workgroups; /* ABI 17: workgroups to leave OUT of the launch (0: two per CU, the whole device).  The kernel is
                             * persistent and books every register of a CU, so launches on another stream wait until it retires;
                             * with a few workgroups left out they run beside it (the caller's rest-row launches: DESIGN.md 4.8.9). */
  int32_t _pad4;
  int32_t* tile_counter;  /* ABI 20, nullable: TWO device int32 the call owns while it runs, ZERO before the first launch (the kernel
                           * leaves them zero: the last workgroup to finish resets them, so back-to-back launches on one stream need
                           * nothing in between).  Non-null: DYNAMIC tile schedule -- a workgroup's first four tiles are b, b + G,
                           * b + 2 G, b + 3 G (G = the launch's grid), every later tile index is 4 G + a claim from the first word,
                           * so a workgroup that falls behind simply takes fewer tiles; tiles are started in tile_desc order
                           * device-wide.  Null: the static schedule b, b + G, b + 2 G, ...  The results do not depend on it
                           * (bit-identical). */
  /* ABI 21 (round 6): the arithmetic of the contraction, see ARITHMETIC above */
  const void* w_img_x3;     /* bf16 x 3 images (pna_fused_pack_f32(.., x3 = 1)), image_stride_x3 == pna_fused_image_bytes(F, N, tower, 1) apart:
                             * PNA_FD_ARITH_GUARDED and PNA_FD_ARITH_X3; w_img / image_stride: the fp16 x 2 images (GUARDED and H2) */
  int64_t image_stride_x3;
  void* guard_ws;           /* PNA_FD_ARITH_GUARDED: pna_fused_degree_guard_bytes(M) bytes, 16-byte aligned, owned by the call while it runs, its first 16
                             * bytes ZERO before the first launch (the library leaves words 0..1 zero behind every call).  int32 words: [0] tiles
                             * handed over by the running call, [1] internal, [2] tiles handed over since the caller last zeroed it, [3] calls */
  int64_t guard_ws_bytes;
  int32_t arith;            /* PNA_FD_ARITH_GUARDED (0, the default) | PNA_FD_ARITH_X3 | PNA_FD_ARITH_H2 */
  int32_t _pad5;
  const float* pre_add;     /* tower mode, nullable: (n_nodes, ld_pre_add) rows, node order, added to the biased accumulator IN FRONT of the row factor:
                             *   y = residual + act(((bias + W_D . a + .. + pre_add[perm[v]]) * row_post[v]) * col_scale + col_shift)
                             * -- a PNALayer of T towers over the whole input (divide_input=False, models/dgl/pna_layer.py:137-139) is T launches, one per
                             * tower's 75 message features, the partial sums carried from launch to launch; only the last one applies bias / row factor /
                             * BatchNorm / activation / residual.  pre_add must NOT alias y under PNA_FD_ARITH_GUARDED: the second launch computes
                             * a handed-over tile again from the same pre_add rows.  Also taken by the layer proper (with row_post as its
                             * optional row factor): a layer in feature panels, FusedMultiTowerCall's launches */
  int64_t ld_pre_add;
  int32_t y_cols_writable;  /* 0 (= N), or N <= y_cols_writable <= min(ldy, 16 ceil(N / 16)): the columns of a row of y the kernel may WRITE; the ones
                             * behind N receive zeros.  With the padding of its own output buffer writable, a row of N = 75 floats is ten whole
                             * 32-byte sectors (no partially written sector for the memory side to complete by reading it first) and the row's last
                             * window one 16-byte store instead of three scalar ones.  Never set it for a y that is a view of wider rows. */
  int32_t _pad6;
} pna_fused_degree_args;

#define PNA_FD_ARITH_GUARDED 0 /* fp16 x 2 with the floor-error guard: tiles it cannot certify are computed again in bf16 x 3 (a second launch) */
#define PNA_FD_ARITH_X3 1      /* bf16 x 3 everywhere: componentwise fp32-accurate for operands of any dynamic range (rounds 3-4) */
#define PNA_FD_ARITH_H2 2      /* fp16 x 2 without the guard (round 5): operands of moderate dynamic range only; the verification instantiation */
int64_t pna_fused_image_bytes(int32_t F, int32_t N, int32_t tower, int32_t x3);   /* bytes of one image = the stride between images; 0 = unsupported shape */
int pna_fused_pack_f32(const float* w_ref, int64_t ldw_ref, int32_t N, int32_t F, int32_t n_scaler, const float* scale, int32_t n_img, void* img,
                       int32_t tower, int32_t x3, pna_stream_t stream);          /* the images of either arithmetic, layer or tower mode */
int64_t pna_fused_degree_guard_bytes(int64_t M);               /* size of pna_fused_degree_args.guard_ws for M virtual rows */
int64_t pna_fused_degree_image_bytes(int32_t F, int32_t N);   /* = pna_fused_image_bytes(F, N, 0, 0); 0 = unsupported shape */
int32_t pna_fused_degree_tile_rows(int32_t F, int32_t N);     /* rows of a workgroup tile for the shape (M must be a multiple; all of
                                                                  one degree and one weight image): 64, or 128 for the wide shapes when the
                                                                  library runs them as 8-wavefront workgroups; 0 = unsupported shape */
int pna_fused_degree_pack_f32(const float* w_ref, int64_t ldw_ref, int32_t N, int32_t F, int32_t n_scaler, const float* scale,
                              int32_t n_img, void* img, pna_stream_t stream);
int pna_fused_degree_f32(const pna_fused_degree_args* args, pna_stream_t stream);
int64_t pna_fused_tower_image_bytes(int32_t F, int32_t N);    /* 0 = unsupported shape */
int pna_fused_tower_pack_f32(const float* w_ref, int64_t ldw_ref, int32_t N, int32_t F, int32_t n_scaler, const float* scale,
                             int32_t n_img, void* img, pna_stream_t stream);

/* ---- the tower layer of molecule-sized batches: one call, two launches (BASELINE.json configs[1]) -------------------
 *
 * At ~3 k nodes / ~6 k edges per batch the layer is bound by launch and load LATENCY, so the work is cut by destination rows
 * instead of by operator (pna_tower_fused.hip).  pna_tower_layer_f32 evaluates, in eval mode, all of
 * models/dgl/pna_layer.py:133-148 (PNALayer.forward) with 1-layer pretrans / posttrans and no edge features:
 *   launch 1   x_cat[v] = [ W_a,t h_t[v] | W_b,t h_t[v] + b_t ]_t          the pretrans Linear(2 Fi -> Fi) of every tower,
 *                                                                          factorised to node level (:35-40)
 *   launch 2   per 16 destination rows, never leaving the CU:
 *              a_t[v]   = [mean | max | min | std] over in-edges (u -> v) of x_src,t[u] + x_dst,t[v]      (:42-56)
 *              z_t[v]   = ((b_t + W_h,t h_t[v] + sum_s row_scale[s][v] (W_s,t a_t[v])) * row_post[v]) * col_scale + col_shift   (:65-74)
 *              y[v]     = residual[v] + act(W_mix [z_0 .. z_T-1][v] + b_mix)                               (:141-147)
 *   (mix_img == NULL: y = [z_0 .. z_T-1], the plain concatenation; PNATower on its own.)
 * h_t = h[:, t*Fi .. (t+1)*Fi) when divide_input, else all of h (then Fin = Fi).  Exact fp32 products on
 * v_mfma_f32_16x16x4_f32; the aggregation is the light-row arithmetic of pna_segreduce_fwd_f32 (edge order, s / D correctly
 * rounded), for every row whatever its degree; results agree with the large-graph path to fp32 summation-order noise.
 * Aggregators are fixed to mean|max|min|std, n_scaler <= 3.  Returns PNA_E_INVALID when one tower's 16-row tile does not fit
 * the LDS (Fi > ~570): callers use the large-graph kernels then.
 *
 * Weight images (made once per weight update):
 *   pna_small_pack_f32       a Linear weight (N, K) for pna_small_linear_f32: proj_img = pack of [W_a ; W_b] (2*T*Fi, Fin)
 *                            (block-diagonal in Fin when divide_input), mix_img = pack of the mixing weight (No, T*Fo)
 *   pna_tower_post_pack_f32  ONE tower's posttrans weight (Fo, Fi + n_scaler*4*Fi) in the reference's column order
 *                            [h | scaler 0: mean max min std | scaler 1 ...] (:65); post_img = the T images one after the other,
 *                            each pna_tower_post_packed_floats(Fi, Fo, n_scaler) floats
 * pna_small_linear_f32 is launch 1 on its own: y = residual + act(x W^T + bias) for row counts where a library GEMM is
 * launch-bound; any M, K <= 2500.
 */
int64_t pna_small_packed_floats(int32_t N, int32_t K);
int pna_small_pack_f32(const float* w_ref, int64_t ldw_ref, int32_t N, int32_t K, float* img, pna_stream_t stream);

typedef struct pna_small_linear_args {
  uint32_t struct_size;    /* sizeof(pna_small_linear_args) of the CALLER's header (ABI 19): a shorter struct is refused with PNA_E_INVALID */
  uint32_t _abi_reserved;  /* 0 */
  const float* x;        /* (M, ldx), K columns used */
  int64_t ldx;
  int32_t M;
  int32_t K;
  int32_t N;
  int32_t act;           /* 0 = none, 1 = ReLU, 2 = LeakyReLU(act_slope) */
  const float* img;      /* pna_small_pack_f32 image of the (N, K) weight */
  const float* bias;     /* nullable [N] */
  float act_slope;
  int32_t _pad;
  const float* residual; /* nullable (M, ld_res): added after the activation */
  int64_t ld_res;
  float* y;              /* (M, ldy) */
  int64_t ldy;
} pna_small_linear_args;

int pna_small_linear_f32(const pna_small_linear_args* args, pna_stream_t stream);

int64_t pna_tower_post_packed_floats(int32_t Fi, int32_t Fo, int32_t n_scaler);
int pna_tower_post_pack_f32(const float* w_ref, int64_t ldw_ref, int32_t Fi, int32_t Fo, int32_t n_scaler, float* img,
                            pna_stream_t stream);

typedef struct pna_tower_layer_args {
  uint32_t struct_size;    /* sizeof(pna_tower_layer_args) of the CALLER's header (ABI 19): a shorter struct is refused with PNA_E_INVALID */
  uint32_t _abi_reserved;  /* 0 */
  const int32_t* rowptr;  /* [V+1] CSR by destination */
  const int32_t* col;     /* [E] source node per CSR edge */
  int32_t V;
  int32_t n_tower;        /* T */
  int32_t Fi;             /* per-tower input width (= pretrans output width) */
  int32_t Fo;             /* per-tower output width */
  int32_t divide_input;   /* 0: every tower reads all of h (ldh >= Fi); 1: tower t reads columns [t*Fi, (t+1)*Fi) (ldh >= T*Fi) */
  int32_t n_scaler;       /* 1..3 */
  const float* h;         /* (V, ldh) node features */
  int64_t ldh;
  float* x_cat;           /* (V, ldx) workspace, 2*T*Fi columns: written by launch 1, read by launch 2 */
  int64_t ldx;
  const float* proj_img;  /* pna_small_pack_f32 image of [W_a ; W_b] (2*T*Fi, Fin) */
  const float* proj_bias; /* [2*T*Fi] = [0 ; b], nullable */
  const float* row_scale[PNA_MAX_SCALER]; /* each [V] or NULL = identity */
  const float* post_img;  /* T pna_tower_post_pack_f32 images */
  const float* post_bias; /* nullable [T*Fo] */
  const float* row_post;  /* nullable [V]: graph-norm factor */
  const float* col_scale; /* nullable [T*Fo]: eval BatchNorm scale of every tower, concatenated */
  const float* col_shift; /* nullable [T*Fo] (with col_scale) */
  const float* mix_img;   /* nullable: pna_small_pack_f32 image of the mixing weight (No, T*Fo) */
  const float* mix_bias;  /* nullable [No] */
  int32_t No;             /* mixing output width (ignored without mix_img: the output is T*Fo wide) */
  int32_t mix_act;        /* 0 = none, 1 = ReLU, 2 = LeakyReLU(mix_slope) */
  float mix_slope;
  int32_t _pad;
  const float* residual;  /* nullable (V, ld_res): added after the mixing activation */
  int64_t ld_res;
  float* y;               /* (V, ldy) */
  int64_t ldy;
  /* edge features that are an embedding of <= 4 edge types (the molecule nets: pna_net.py `e = self.embedding_e(e)`; the W_e . ef
   * part of the factorised pretrans, models/dgl/pna_layer.py:35-40): message = (W_a h_u + (W_b h_v + b)) + edge_table[edge_type[k]].
   * All NULL / 0: a layer without edge features. */
  const int32_t* edge_type;  /* nullable [E]: type of every CSR edge, 0 <= type < n_edge_types (not checked on the device) */
  const float* edge_table;   /* (n_edge_types, ld_edge_table): row t = W_e . ef_t of every tower, concatenated (n_tower * Fi columns) */
  int64_t ld_edge_table;
  int32_t n_edge_types;      /* 1..4 with edge_type */
  int32_t no_self_panel;     /* != 0: the PNASimpleLayer form (models/dgl/pna_layer.py:197-206: messages are the raw source features, the
                              * posttrans never reads the node's own h): no destination term (the second half of x_cat is not read) and zeros
                              * instead of h against the image's (zero) self block -- so a node whose own feature is Inf / NaN keeps the
                              * finite output the reference gives it instead of 0 * Inf = NaN */
} pna_tower_layer_args;

int pna_tower_layer_f32(const pna_tower_layer_args* args, pna_stream_t stream);

/* ---- weight / bias gradient of the posttrans contraction (round 4) ------------------------------------------
 * replaces: the autograd node of `self.posttrans(torch.cat([h, scaled aggregate], dim=1))` (models/dgl/pna_layer.py:206; the
 * training loop's loss.backward(), realworld_benchmark/train/train_molecules_graph_regression.py:29-32) for the Linear's
 * weight and bias -- rounds 2-3 used the vendor GEMM library here:
 *
 *   grad_w[n, Kh + s K + k] = sum_m row_scale[s][m] gy[m, n] a[m, k]        (row_scale[s] == NULL: 1)
 *   grad_w[n, j]            = sum_m gy[m, n] h[m, j]                        j < Kh
 *   grad_b[n]               = sum_m gy[m, n]                                (grad_b nullable)
 *
 * bf16x3 arithmetic (each fp32 operand cut exactly into three bf16 terms, six partial products, fp32 accumulation per slab of
 * rows, float64 across the slabs in a fixed order: deterministic).  Shapes: 1 <= n_scaler <= 3, n_scaler * N <= 240,
 * K + Kh + 1 <= 384 (pna_posttrans_dw_workspace_bytes returns -1 otherwise: the caller keeps its library route); rows 4-byte
 * aligned; with an h panel or grad_b, row_scale[0] must be NULL (they are formed from the first copy of gy -- the identity scaler
 * comes first in every reference configuration).  Non-finite inputs give NaN in the columns / rows they touch.  workspace: no
 * initialisation needed.
 */
typedef struct pna_posttrans_dw_args {
  uint32_t struct_size;    /* sizeof(pna_posttrans_dw_args) of the CALLER's header (ABI 19): a shorter struct is refused with PNA_E_INVALID */
  uint32_t _abi_reserved;  /* 0 */
  const float* gy;         /* (M, ldg): gradient of the contraction's output, N columns */
  int64_t ldg;
  int64_t M;
  int32_t N;
  int32_t n_scaler;
  const float* a;          /* (M, lda): the aggregate the forward multiplied, K columns (identity-scaled) */
  int64_t lda;
  int32_t K;
  int32_t Kh;              /* columns of h (0: no self panel) */
  const float* h;          /* nullable (M, ldh) */
  int64_t ldh;
  const float* row_scale[PNA_MAX_SCALER];  /* each [M] or NULL = identity */
  float* grad_w;           /* (N, ldw): [h panel | scaler block 0 | 1 | 2], the Linear's weight layout */
  int64_t ldw;
  float* grad_b;           /* nullable [N] */
  void* workspace;
  int64_t workspace_bytes; /* >= pna_posttrans_dw_workspace_bytes(M, N, n_scaler, K, Kh) */
} pna_posttrans_dw_args;

int64_t pna_posttrans_dw_workspace_bytes(int64_t M, int32_t N, int32_t n_scaler, int32_t K, int32_t Kh);
int pna_posttrans_dw_f32(const pna_posttrans_dw_args* args, pna_stream_t stream);

/* The same gradient for the rows of a DEGREE PLAN (pna_amd/degree_groups.py: virtual rows in 128-row tiles of one in-degree).  The
 * degree scalers are functions of the in-degree alone (models/dgl/scalers.py:7-19), so with the rows walked in plan order
 *   grad_w block s = sum over degree runs  group_scale[group][s] * sum_{m in run} gy[m]^T a[m]
 * needs one unscaled copy of gy and a third of the multiply-adds; the rows are loaded through row_perm (no packing pass).
 * Covers the plan's group rows only: the caller adds the (few) rest rows' part.  Workgroup w walks tiles [wg_range[2w],
 * wg_range[2w+1]) and writes one partial product per degree run it meets into workspace entries wg_entry[w], wg_entry[w] + 1, ...;
 * entry_group[e] = the degree group of entry e.  N <= 80, K + Kh + 1 <= 384, n_scaler in 1..3. */
typedef struct pna_posttrans_dw_grouped_args {
  uint32_t struct_size;    /* sizeof(pna_posttrans_dw_grouped_args) of the CALLER's header (ABI 19): a shorter struct is refused with PNA_E_INVALID */
  uint32_t _abi_reserved;  /* 0 */
  const float* gy;         /* (n_nodes, ldg) */
  int64_t ldg;
  int32_t N;
  int32_t n_scaler;
  const float* a;          /* (n_nodes, lda): aggregate in NODE order, K columns */
  int64_t lda;
  int32_t K;
  int32_t Kh;
  const float* h;          /* nullable (n_nodes, ldh) */
  int64_t ldh;
  const int32_t* row_perm;    /* [128 * n_tiles]: node of every virtual row, -1 = padding */
  const int32_t* tile_group;  /* [n_tiles]: degree group of every 128-row tile, runs of equal values */
  const int32_t* wg_range;    /* [n_workgroups][2] */
  const int32_t* wg_entry;    /* [n_workgroups] */
  int32_t n_workgroups;
  int32_t n_entries;
  const int32_t* entry_group; /* [n_entries] */
  const float* group_scale;   /* [n_groups][n_scaler] */
  float* grad_w;           /* (N, ldw) */
  int64_t ldw;
  float* grad_b;           /* nullable [N] */
  void* workspace;
  int64_t workspace_bytes; /* >= pna_posttrans_dw_grouped_workspace_bytes(N, K, Kh, n_entries) */
  int32_t a_plan_order;    /* != 0: `a` is already in the plan's row order (row r of `a` = virtual row r): read in sequence, not through row_perm */
  int32_t _pad;
} pna_posttrans_dw_grouped_args;

int64_t pna_posttrans_dw_grouped_workspace_bytes(int32_t N, int32_t K, int32_t Kh, int32_t n_entries);
int pna_posttrans_dw_grouped_f32(const pna_posttrans_dw_grouped_args* args, pna_stream_t stream);

/* ---- batching + destination-sorted CSR on the device ----------------------------------
 *
 * Replaces dgl.batch (realworld_benchmark/data/molecules.py:153-164: offset + concatenate the member graphs' edge
 * lists) and the in-edge CSR DGL builds inside update_all (models/dgl/pna_layer.py:64,:202).
 *   global ids:  dst[k] + node_offset[edge_graph[k]]  (same for src; both NULL = ids are already global)
 *   rowptr[V+1], col[E] (source per CSR edge), eid[E] (original edge per CSR edge), row[E] (destination per CSR edge)
 * grouped by destination and STABLE inside a group (the original edge order: DGL's mailbox order).  One radix sort
 * over the bits the node count needs; workspace from pna_collate_workspace_bytes (-1 if a size leaves int32).
 */
int64_t pna_collate_workspace_bytes(int64_t n_edges, int32_t n_nodes);
int pna_collate_csr_i32(const int32_t* src, const int32_t* dst, int64_t n_edges, int32_t n_nodes,
                        const int32_t* edge_graph /* nullable [E] */, const int32_t* node_offset /* nullable [n_graphs] */,
                        int32_t* rowptr, int32_t* col, int32_t* eid, int32_t* row,
                        void* workspace, int64_t workspace_bytes, pna_stream_t stream);

/* ---- halo packing for the destination-sharded multi-GPU path (the reference has no distributed code) ------
 *
 *   out[i, 0:F] = x[idx[i], 0:F]      i in [0, n)
 * One launch packs the feature rows every peer asked for (idx = the peers' request lists, concatenated in peer order) into
 * the contiguous send buffer of the per-layer all-to-all; pna_amd/shard.py is the caller.  Rows need 4-byte alignment only.
 */
int pna_pack_rows_f32(const float* x, int64_t ldx, const int32_t* idx, int64_t n, int32_t F, float* out, int64_t ldo,
                      pna_stream_t stream);

/* ---- node-level projection with a wide output (ABI 21) ------------------------------------------------------------------
 * replaces: the source half of `self.pretrans` of every tower of a PNALayer whose towers all read the whole input
 * (models/dgl/pna_layer.py:137-139 divide_input=False; models/dgl/pna_layer.py:36-44 pretrans_edges), factorised to node level:
 *
 *   y[m, 0:N] = sum_k x[m, k] w[n, k]            x (M, K) fp32, 4 <= K <= 128;  w (N, K) row-major (nn.Linear's layout), N <= 512
 *
 * fp32 operands, exact fp32 products (v_mfma_f32_16x16x4_f32), fp32 accumulation: an fp32 GEMM up to the order of the sum.  The whole
 * weight stays in LDS (16 ceil(K / 16) x (80 ceil(N / 80) + 4) floats <= 160 KB, else PNA_E_INVALID), x is read once and every row
 * of y is written once, whatever N -- the product is bound by its output stream.  Columns [N, ldy) of y are left alone.
 */
int pna_project_f32(const float* x, int64_t ldx, int64_t M, int32_t K, const float* w, int64_t ldw, int32_t N, float* y, int64_t ldy,
                    pna_stream_t stream);

/* The scaled form (ABI 21): the part of a multi-tower PNALayer that is linear in the row's OWN features -- the destination half of every
 * tower's pretrans (models/dgl/pna_layer.py:36-44: mean / max / min of (a_u + b_v) = those of a_u, + b_v), through the scalers
 * (models/dgl/scalers.py:11-26) and the collapsed posttrans . mixing weight, plus the self panel of posttrans (pna_layer.py:69-72):
 *
 *   y[m, n] = [self_block] sum_k x[m, k] w[n, k]  +  sum_{s < n_scaled} scales[m, s] (sum_k x[m, k] w[n, (self_block + s) K + k] + beta[s, n])
 *
 * w (N <= 80, blocks x K) row-major, blocks = self_block + n_scaled in 1..4 (four: K <= 80) side by side along a row; scales (M, n_scaled) (zero for a row
 * without in-edges); beta (n_scaled, N) nullable.  Same arithmetic and residency as pna_project_f32; one pass, every block's accumulators
 * in registers, the combination lane-local.  pna_amd/functional.py::FusedMultiTowerCall.dense_term is the caller.
 */
int pna_project_scaled_f32(const float* x, int64_t ldx, int64_t M, int32_t K, const float* w, int64_t ldw, int32_t N, int32_t n_scaled,
                           int32_t self_block, const float* scales, int64_t ld_scales, const float* beta /* nullable */, int64_t ld_beta,
                           float* y, int64_t ldy, pna_stream_t stream);

/* The grouped form (ABI 21): one weight per DEGREE GROUP of a degree plan, rows through the plan's permutation --
 *
 *   y[node, 0:N] = sum_k x[node, k] w_groups[g][n, k]        node = row_perm[v] (-1: padding, skipped), g = tile_group[v / 128], v in [0, M)
 *
 * the backward's d agg = gy W_D^T of a PNASimpleLayer in training (autograd of models/dgl/pna_layer.py:197-206): every scaler is a function
 * of the in-degree alone (models/dgl/scalers.py:7-19), so for the rows of one degree the three scaler blocks collapse into one 4F x N
 * matrix.  w_groups: n_groups dense (N, K) matrices group_stride FLOATS apart; M a multiple of 128.  A workgroup walks a contiguous piece of
 * the tiles and refills its LDS image when the group changes: list the tiles sorted by group (they are independent -- any order of
 * (128 entries of row_perm, one entry of tile_group) pairs is the same product).
 * Same arithmetic and limits as pna_project_f32.  Rows of y that no virtual row names are left alone.
 */
int pna_project_grouped_f32(const float* x, int64_t ldx, int64_t x_rows, int32_t K, const float* w_groups, int64_t group_stride, int32_t n_groups,
                            int32_t N, const int32_t* row_perm, int64_t M, const int32_t* tile_group, float* y, int64_t ldy, pna_stream_t stream);

/* ---- the tail of PNASimpleLayer's TRAINING forward, and its backward (ABI 18) ----------------------------------------
 * replaces: models/dgl/pna_layer.py:207-213 in training mode -- `h = self.batchnorm_h(h)` (nn.BatchNorm1d, batch statistics),
 * `h = F.relu(h)`, `h = h_in + h` -- and their autograd nodes:
 *
 *   mean_c, var_c = batch statistics of y[:, c] (biased variance);  save_mean = mean, save_invstd = 1 / sqrt(var + eps)
 *   out = residual + act((y - mean) * (gamma * save_invstd) + beta)                   act = ReLU when relu != 0
 *   running_mean = (1 - momentum) running_mean + momentum mean;  running_var likewise with var * M / (M - 1)
 *                                                                 (both skipped when running_mean is NULL or momentum < 0)
 *   backward (g' = grad_out where the activation passed, xhat = (y - mean) save_invstd):
 *   grad_beta = sum_r g',  grad_gamma = sum_r g' xhat,  grad_y = gamma save_invstd (g' - mean_r g' - xhat mean_r(g' xhat))
 *   (the residual's gradient is grad_out itself: the caller's)
 *
 * Two streaming passes each way (column sums, then the element-wise pass) instead of the ~10 library passes; column sums are
 * taken of y - y[0, c] in fp32 per 512 rows and in float64 across them.  M >= 2 rows (nn.BatchNorm1d raises below that: the
 * caller's), 1 <= N <= 128, rows 4-byte aligned, ld >= N.  workspace: pna_bn_tail_workspace_bytes(M, N) bytes, no
 * initialisation needed, not kept between the two calls.
 */
typedef struct pna_bn_tail_args {
  uint32_t struct_size;    /* sizeof(pna_bn_tail_args) of the CALLER's header (ABI 19): a shorter struct is refused with PNA_E_INVALID */
  uint32_t _abi_reserved;  /* 0 */
  const float* y;          /* (M, ldy): the posttrans output */
  int64_t ldy;
  int64_t M;
  int32_t N;
  int32_t relu;
  const float* gamma;      /* nullable [N] (affine=False: 1) */
  const float* beta;       /* nullable [N] */
  float eps;
  float momentum;          /* < 0: leave the running statistics alone */
  float* running_mean;     /* nullable [N]; with running_var */
  float* running_var;
  const float* residual;   /* fwd, nullable (M, ld_res) */
  int64_t ld_res;
  float* out;              /* fwd (M, ld_out) */
  int64_t ld_out;
  float* save_mean;        /* [N]: written by fwd, read by bwd */
  float* save_invstd;      /* [N] */
  void* workspace;
  int64_t workspace_bytes;
  const float* grad_out;   /* bwd (M, ld_go) */
  int64_t ld_go;
  float* grad_y;           /* bwd (M, ld_gy) */
  int64_t ld_gy;
  float* grad_gamma;       /* bwd, nullable [N] */
  float* grad_beta;        /* bwd, nullable [N] */
} pna_bn_tail_args;

int64_t pna_bn_tail_workspace_bytes(int64_t M, int32_t N);
int pna_bn_tail_fwd_f32(const pna_bn_tail_args* args, pna_stream_t stream);
int pna_bn_tail_bwd_f32(const pna_bn_tail_args* args, pna_stream_t stream);

const char* pna_last_error(void);
int pna_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PNA_AMD_H */
