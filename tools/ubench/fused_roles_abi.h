// The argument block and entry points pna_fused_roles.hip exported through include/pna_amd.h in ABI 19 (removed in ABI 20):
// kept here so that the archived kernel still compiles as a standalone micro-benchmark library.
#pragma once
#include "pna_amd.h"
#ifdef __cplusplus
extern "C" {
#endif
/* ---- the same layer as ONE kernel of SPECIALISED wavefronts (ABI 19; pna_fused_roles.hip) --------------------
 *
 * pna_fused_roles_f32 computes what pna_fused_degree_f32 computes for the rows of the degree groups (PNASimpleLayer.forward,
 * models/dgl/pna_layer.py:197-216 over reduce_func :189-194; same statistics bit for bit) with a different division of labour on
 * the CU: per SIMD one wavefront that only gathers (a register ring of edge packets kept full across tile boundaries) and one
 * that only multiplies (the bf16x3 contraction of pna_fused_degree_f32 over the same weight images: pna_fused_degree_pack_f32); the
 * statistics pass through LDS.  Tables for it (pna_amd/degree_groups.py::DegreePlan.roles_tables):
 *   tile_desc   int32 [n_tiles][4]  = {first id record, in-degree D, weight image, 0} per 64-row TILE of the virtual row order
 *                                     (rows 64 t .. 64 t + 64 of row_perm; all of one in-degree; an all-padding tile has D = 0)
 *   tile_ids    int32, FOUR arrays ids_stride bytes apart, one per 16-row block b of a tile: record (first + e)[i] = source row
 *               of the e-th in-edge of row 64 t + 16 b + i (a padding row repeats the tile's first row); max(D, 1) records per
 *               tile (NO rounding: the ring does not care), tiles back to back, so the records of ANY contiguous tile range
 *               are one linear stream; n_records of them per array, followed by >= 24 padding records of valid ids (0)
 *   wg_range    int32 [n_workgroups][2]: workgroup b takes the tiles [wg_range[2 b], wg_range[2 b + 1]) -- disjoint ranges that
 *               cover [0, n_tiles), chosen by the caller (balanced by edges; pna_fused_roles_grid() workgroups fill the device)
 * x: (x_rows, ldx) fp32, 4-byte aligned rows of any pitch >= round_up(F, 8) (round_up(F, 4) when F % 32 is in 1..16): a
 * CONTIGUOUS (V, F) table qualifies when its storage is readable up to the last row's rounded-up strip; rows are addressed with
 * 64-bit lane addresses (no 4 GiB / 2^24-row limit).  F in 17..80, N in 4..80.  err: int32 [1] on the device, zero before the
 * call; the kernel sets it when one of its bounded spins gave up (never observed; the output is then incomplete).
 */
typedef struct pna_fused_roles_args {
  uint32_t struct_size;   /* sizeof(pna_fused_roles_args) of the CALLER's header: a shorter struct is refused */
  int32_t F;
  int32_t N;
  int32_t relu;           /* 0 none, 1 ReLU, 2 LeakyReLU(act_slope) */
  const int32_t* tile_desc;
  const int32_t* tile_ids;
  int64_t ids_stride;     /* bytes */
  int64_t n_records;
  int64_t n_tiles;
  const int32_t* wg_range;
  int32_t n_workgroups;
  float act_slope;
  const float* x;
  int64_t ldx;
  int64_t x_rows;
  const int32_t* row_perm; /* [64 n_tiles]: node of every virtual row, -1 = padding */
  int64_t n_nodes;         /* rows of y / residual */
  const void* w_img;
  int64_t image_stride;
  const float* bias;       /* nullable [N] */
  const float* col_scale;  /* nullable [N] */
  const float* col_shift;  /* nullable [N] (with col_scale) */
  const float* residual;   /* nullable (n_nodes, ld_res), node order */
  int64_t ld_res;
  float* y;                /* (n_nodes, ldy), node order */
  int64_t ldy;
  float* agg_out;          /* nullable (64 n_tiles, ld_agg): the statistics [mean | max | min | std] x F per virtual row (verification) */
  int64_t ld_agg;
  int32_t* err;            /* device int32 [1] */
} pna_fused_roles_args;

int32_t pna_fused_roles_supported(int32_t F, int32_t N);      /* 1 when pna_fused_roles_f32 has an instantiation for the shape */
int64_t pna_fused_roles_image_bytes(int32_t F, int32_t N);    /* = pna_fused_degree_image_bytes for the shapes it serves; 0 = unsupported */
int32_t pna_fused_roles_grid(int32_t spare_units);            /* workgroups that fill the current device, minus spare_units CUs */
int pna_fused_roles_f32(const pna_fused_roles_args* args, pna_stream_t stream);

#ifdef __cplusplus
}
#endif
