// pna_segreduce.hip -- fused gather + {mean,sum,max,min,std,var} segment-reduce + degree scalers
// for gfx950 (MI355X, CDNA4).  Implements pna_segreduce_fwd_f32 / pna_degree_scalers_f32 of
// include/pna_amd.h; see that header for the reference code each entry point replaces.
//
// Execution model (DESIGN.md "segment-reduce kernel"):
//   * a 64-lane wavefront is cut into G = 64/L lane groups; one group owns one destination row
//     at a time and each of its L lanes owns VEC (=4) consecutive features, so a row gather is one
//     global_load_dwordx4 per lane and no cross-lane reduction is ever needed: every lane keeps
//     its own (sum, sum of squares, max, min) in registers for its features;
//   * the source ids of a row are fetched L at a time by the group's lanes (one coalesced load) and
//     broadcast with ds_bpermute, then U gathers are issued back to back before the first is used
//     (memory-level parallelism is what bounds this kernel, not VALU);
//   * features that do not fill the last dwordx4 are handled by sliding the last lane's window back
//     to [F-4, F): the overlap recomputes identical values, so no padding of x or out is required
//     and any 4-byte aligned leading dimension works;
//   * rows with more than `heavy_threshold` in-edges are cut into fixed segments that other lane
//     groups reduce in parallel (blocks at the front of the grid, so the long poles start first);
//     a second tiny kernel folds the per-segment partials in segment order => results are
//     independent of the launch geometry.
// All arithmetic is fp32 without FMA contraction (-ffp-contract=off), mirroring the op-by-op
// rounding of the reference's torch code (x*x rounded, then summed; mean*mean rounded, then
// subtracted).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "pna_amd.h"
#include "pna_internal.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) f4u { f4 v; };   // dwordx4 at 4-byte alignment
typedef int i4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) i4u { i4 v; };

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kNQ = 7;   // partial quantities per heavy segment: s, q, mx, mn, amx, amn, wsum

struct KArgs {
  const int32_t* rowptr; const int32_t* col;
  const float* x; const float* dst_term; const float* edge_term; const float* ew;
  const float* row_scale[PNA_MAX_SCALER];
  float* out; int32_t* argmax; int32_t* argmin;
  const int32_t* heavy_rows; const int32_t* heavy_segptr; const int32_t* seg_heavy; float* partials;
  long ldx, ld_dst, ld_edge, ldo, ld_arg, ts_in, ts_out;
  int V, F, n_aggr, n_scaler, block_stride;
  int aggr[PNA_MAX_AGGR];
  int heavy_threshold, seg_len, n_heavy, n_seg;
  int L, G, R, n_heavy_blocks, pstride, nt, T, tiles;
};

template <int VEC> struct Ld;
template <> struct Ld<4> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    f4 t = reinterpret_cast<const f4u*>(p)->v;
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4], bool nt) {
    f4 t = {v[0], v[1], v[2], v[3]};
    typedef f4 f4a4 __attribute__((aligned(4)));
    if (nt) __builtin_nontemporal_store(t, reinterpret_cast<f4a4*>(p));
    else reinterpret_cast<f4u*>(p)->v = t;
  }
  static __device__ __forceinline__ void store_i(int32_t* p, const int (&v)[4]) {
    i4 t = {v[0], v[1], v[2], v[3]};
    reinterpret_cast<i4u*>(p)->v = t;
  }
};
template <> struct Ld<1> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[1]) { v[0] = *p; }
  static __device__ __forceinline__ void store(float* p, const float (&v)[1], bool nt) {
    if (nt) __builtin_nontemporal_store(v[0], p); else *p = v[0];
  }
  static __device__ __forceinline__ void store_i(int32_t* p, const int (&v)[1]) { *p = v[0]; }
};

template <int VEC, bool EXTRA> struct Acc {
  float s[VEC], q[VEC], mx[VEC], mn[VEC];
  int amx[VEC], amn[VEC];   // only live when EXTRA (dead-code eliminated otherwise)
  float wsum;
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int k = 0; k < VEC; ++k) { s[k] = 0.f; q[k] = 0.f; mx[k] = -INFINITY; mn[k] = INFINITY; }
#pragma unroll
    for (int k = 0; k < VEC; ++k) { amx[k] = -1; amn[k] = -1; }
    wsum = 0.f;
  }
};

// One message m (already gathered) of CSR edge position e folded into the accumulators.
template <int VEC, bool EXTRA>
__device__ __forceinline__ void fold(Acc<VEC, EXTRA>& a, const float (&m)[VEC], int e, float w, bool has_w) {
  if constexpr (!EXTRA) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const float v = m[k];
      a.s[k] = a.s[k] + v;
      a.q[k] = a.q[k] + v * v;
      a.mx[k] = (v > a.mx[k] || v != v) ? v : a.mx[k];   // NaN-propagating like torch.max
      a.mn[k] = (v < a.mn[k] || v != v) ? v : a.mn[k];
    }
  } else {
    const bool on = !has_w || w > 0.f;                     // max/min: adjacency is a mask (A.5)
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const float v = m[k];
      if (has_w) {
        a.s[k] = a.s[k] + v * w;                           // torch.mul(X, adj) then sum
        a.q[k] = a.q[k] + (v * v) * w;                     // torch.mul(torch.mul(X, X), adj)
      } else {
        a.s[k] = a.s[k] + v;
        a.q[k] = a.q[k] + v * v;
      }
      const bool gx = on && (v > a.mx[k] || (v != v && a.mx[k] == a.mx[k]));
      const bool gn = on && (v < a.mn[k] || (v != v && a.mn[k] == a.mn[k]));
      a.mx[k] = gx ? v : a.mx[k]; a.amx[k] = gx ? e : a.amx[k];
      a.mn[k] = gn ? v : a.mn[k]; a.amn[k] = gn ? e : a.amn[k];
    }
    a.wsum = a.wsum + (has_w ? w : 1.f);
  }
}

// B consecutive edges [e0, e0+B) of one row: issue every gather first, then fold in edge order.
template <int VEC, int B, bool EXTRA>
__device__ __forceinline__ void batch(const KArgs& a, Acc<VEC, EXTRA>& acc, int myidx, int src_lane0, int e0,
                                      long off, const float (&dterm)[VEC]) {
  int id[B];
  float v[B][VEC];
#pragma unroll
  for (int u = 0; u < B; ++u) id[u] = a.col ? __shfl(myidx, src_lane0 + u) : (e0 + u);
#pragma unroll
  for (int u = 0; u < B; ++u) Ld<VEC>::load(a.x + (size_t)id[u] * a.ldx + off, v[u]);
  if constexpr (!EXTRA) {
#pragma unroll
    for (int u = 0; u < B; ++u) fold<VEC, false>(acc, v[u], e0 + u, 1.f, false);
  } else {
    float et[B][VEC];
    float w[B];
    if (a.edge_term) {
#pragma unroll
      for (int u = 0; u < B; ++u) Ld<VEC>::load(a.edge_term + (size_t)(e0 + u) * a.ld_edge + off, et[u]);
    }
    if (a.ew) {
#pragma unroll
      for (int u = 0; u < B; ++u) w[u] = a.ew[e0 + u];
    }
#pragma unroll
    for (int u = 0; u < B; ++u) {
      float m[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        m[k] = v[u][k];
        if (a.dst_term) m[k] = m[k] + dterm[k];
        if (a.edge_term) m[k] = m[k] + et[u][k];
      }
      fold<VEC, true>(acc, m, e0 + u, a.ew ? w[u] : 1.f, a.ew != nullptr);
    }
  }
}

// All lanes of a group walk CSR positions [beg, end) of destination `row`.
template <int VEC, int U, bool EXTRA>
__device__ __forceinline__ void walk(const KArgs& a, Acc<VEC, EXTRA>& acc, int row, int beg, int end, int c,
                                     int grp_lane0, long off) {
  float dterm[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) dterm[k] = 0.f;
  if (EXTRA && a.dst_term && end > beg) Ld<VEC>::load(a.dst_term + (size_t)row * a.ld_dst + off, dterm);
  const int L = a.L;
  for (int cb = beg; cb < end; cb += L) {
    const int nidx = min(L, end - cb);
    int myidx = 0;
    if (a.col && c < nidx) myidx = a.col[cb + c];
    int j = 0;
    for (; j + U <= nidx; j += U) batch<VEC, U, EXTRA>(a, acc, myidx, grp_lane0 + j, cb + j, off, dterm);
    if (U >= 8 && nidx - j >= 4) { batch<VEC, 4, EXTRA>(a, acc, myidx, grp_lane0 + j, cb + j, off, dterm); j += 4; }
    if (U >= 4 && nidx - j >= 2) { batch<VEC, 2, EXTRA>(a, acc, myidx, grp_lane0 + j, cb + j, off, dterm); j += 2; }
    if (U >= 2 && nidx - j >= 1) { batch<VEC, 1, EXTRA>(a, acc, myidx, grp_lane0 + j, cb + j, off, dterm); j += 1; }
  }
}

// mean/std/... from the raw accumulators, times the row scalers, written in the reference's
// scaler-major / aggregator-minor block order.
template <int VEC, bool EXTRA>
__device__ __forceinline__ void finalize_store(const KArgs& a, const Acc<VEC, EXTRA>& acc, int row, int deg,
                                               long offi, long offo) {
  float mean[VEC], var[VEC], sd[VEC];
  const bool empty = deg <= 0;
  const float D = (EXTRA && a.ew) ? acc.wsum : (float)deg;
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    mean[k] = acc.s[k] / D;                               // torch.mean = sum / n
    const float msq = acc.q[k] / D;
    const float t = msq - mean[k] * mean[k];
    var[k] = (t < 0.f) ? 0.f : t;                         // relu (keeps NaN)
    sd[k] = sqrtf(var[k] + 1e-5f);
  }
  for (int s = 0; s < a.n_scaler; ++s) {
    const float sc = a.row_scale[s] ? a.row_scale[s][row] : 1.f;
    for (int i = 0; i < a.n_aggr; ++i) {
      float o[VEC];
      const int code = a.aggr[i];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        float v;
        switch (code) {
          case PNA_AGG_MEAN: v = mean[k]; break;
          case PNA_AGG_SUM: v = acc.s[k]; break;
          case PNA_AGG_MAX: v = acc.mx[k]; break;
          case PNA_AGG_MIN: v = acc.mn[k]; break;
          case PNA_AGG_STD: v = sd[k]; break;
          default: v = var[k]; break;
        }
        o[k] = empty ? 0.f : v * sc;
      }
      Ld<VEC>::store(a.out + (size_t)row * a.ldo + (size_t)(s * a.n_aggr + i) * a.block_stride + offo, o, a.nt != 0);
    }
  }
  if constexpr (EXTRA) {
    if (a.argmax) Ld<VEC>::store_i(a.argmax + (size_t)row * a.ld_arg + offi, acc.amx);
    if (a.argmin) Ld<VEC>::store_i(a.argmin + (size_t)row * a.ld_arg + offi, acc.amn);
  }
}

template <int VEC, int U, bool EXTRA>
__global__ __launch_bounds__(kBlock) void k_segreduce(const KArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int L = a.L;
  const int grp = lane / L;
  if (grp >= a.G) return;                                  // spare lanes of the wavefront
  const int c = lane - grp * L;
  const int grp_lane0 = grp * L;
  const int nchunks = (a.F + VEC - 1) / VEC;
  const int tower = blockIdx.y / a.tiles;                  // independent feature slice (PNA tower)
  const int chunk = (blockIdx.y - tower * a.tiles) * L + c;
  const bool lane_ok = chunk < nchunks;
  int off = min(chunk, nchunks - 1) * VEC;
  if (VEC == 4) off = min(off, a.F - 4);                   // slide the last window back inside the row
  const long offi = (long)tower * a.ts_in + off;           // column in x / dst_term / edge_term / arg*
  const long offo = (long)tower * a.ts_out + off;          // column in out (before the block offset)
  const int NG = kWaves * a.G;
  const int gid = wave * a.G + grp;

  Acc<VEC, EXTRA> acc;
  if ((int)blockIdx.x < a.n_heavy_blocks) {
    // ---- heavy segments: one lane group per segment, raw partials to the workspace -------------
    const int seg = blockIdx.x * NG + gid;
    if (seg >= a.n_seg) return;
    const int hi = a.seg_heavy[seg];
    const int row = a.heavy_rows[hi];
    const int sidx = seg - a.heavy_segptr[hi];
    const int rbeg = a.rowptr[row], rend = a.rowptr[row + 1];
    const int beg = rbeg + sidx * a.seg_len;
    const int end = min(beg + a.seg_len, rend);
    acc.init();
    walk<VEC, U, EXTRA>(a, acc, row, beg, end, c, grp_lane0, offi);
    if (lane_ok) {
      float* p = a.partials + ((size_t)seg * a.T + tower) * kNQ * a.pstride + off;
      Ld<VEC>::store(p, acc.s, false);
      Ld<VEC>::store(p + a.pstride, acc.q, false);
      Ld<VEC>::store(p + 2 * a.pstride, acc.mx, false);
      Ld<VEC>::store(p + 3 * a.pstride, acc.mn, false);
      if constexpr (EXTRA) {
        Ld<VEC>::store_i(reinterpret_cast<int32_t*>(p + 4 * a.pstride), acc.amx);
        Ld<VEC>::store_i(reinterpret_cast<int32_t*>(p + 5 * a.pstride), acc.amn);
        if (chunk == 0) p[6 * a.pstride - off] = acc.wsum;
      }
    }
    return;
  }
  // ---- ordinary rows: R rows per lane group, interleaved across the block's groups -------------
  const long base = (long)(blockIdx.x - a.n_heavy_blocks) * NG * a.R;
  for (int r = 0; r < a.R; ++r) {
    const long row_l = base + (long)r * NG + gid;
    if (row_l >= a.V) break;
    const int row = (int)row_l;
    const int beg = a.rowptr[row], end = a.rowptr[row + 1];
    const int deg = end - beg;
    if (a.heavy_threshold > 0 && deg > a.heavy_threshold) continue;   // done by the segment blocks
    acc.init();
    walk<VEC, U, EXTRA>(a, acc, row, beg, end, c, grp_lane0, offi);
    if (lane_ok) finalize_store<VEC, EXTRA>(a, acc, row, deg, offi, offo);
  }
}

// Fold the per-segment partials of each heavy row in segment order and finish the row.
template <int VEC, bool EXTRA>
__global__ __launch_bounds__(kBlock) void k_heavy_finalize(const KArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int L = a.L;
  const int grp = lane / L;
  if (grp >= a.G) return;
  const int c = lane - grp * L;
  const int nchunks = (a.F + VEC - 1) / VEC;
  const int tower = blockIdx.y / a.tiles;
  const int chunk = (blockIdx.y - tower * a.tiles) * L + c;
  if (chunk >= nchunks) return;
  int off = chunk * VEC;
  if (VEC == 4) off = min(off, a.F - 4);
  const long offi = (long)tower * a.ts_in + off;
  const long offo = (long)tower * a.ts_out + off;
  const int NG = kWaves * a.G;
  const int hi = blockIdx.x * NG + wave * a.G + grp;
  if (hi >= a.n_heavy) return;
  const int row = a.heavy_rows[hi];
  const int s0 = a.heavy_segptr[hi], s1 = a.heavy_segptr[hi + 1];
  Acc<VEC, EXTRA> acc;
  acc.init();
  for (int seg = s0; seg < s1; ++seg) {
    const float* p = a.partials + ((size_t)seg * a.T + tower) * kNQ * a.pstride + off;
    float ps[VEC], pq[VEC], pmx[VEC], pmn[VEC];
    Ld<VEC>::load(p, ps);
    Ld<VEC>::load(p + a.pstride, pq);
    Ld<VEC>::load(p + 2 * a.pstride, pmx);
    Ld<VEC>::load(p + 3 * a.pstride, pmn);
    int pax[VEC], pan[VEC];
    if constexpr (EXTRA) {
      float t[VEC];
      Ld<VEC>::load(p + 4 * a.pstride, t);
#pragma unroll
      for (int k = 0; k < VEC; ++k) pax[k] = __float_as_int(t[k]);
      Ld<VEC>::load(p + 5 * a.pstride, t);
#pragma unroll
      for (int k = 0; k < VEC; ++k) pan[k] = __float_as_int(t[k]);
      acc.wsum = acc.wsum + p[6 * a.pstride - off];
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      acc.s[k] = acc.s[k] + ps[k];
      acc.q[k] = acc.q[k] + pq[k];
      const bool gx = pmx[k] > acc.mx[k] || (pmx[k] != pmx[k] && acc.mx[k] == acc.mx[k]);
      const bool gn = pmn[k] < acc.mn[k] || (pmn[k] != pmn[k] && acc.mn[k] == acc.mn[k]);
      acc.mx[k] = gx ? pmx[k] : acc.mx[k];
      acc.mn[k] = gn ? pmn[k] : acc.mn[k];
      if constexpr (EXTRA) { acc.amx[k] = gx ? pax[k] : acc.amx[k]; acc.amn[k] = gn ? pan[k] : acc.amn[k]; }
    }
  }
  finalize_store<VEC, EXTRA>(a, acc, row, a.rowptr[row + 1] - a.rowptr[row], offi, offo);
}

// models/dgl/scalers.py:12-19 with the reference's rounding sequence (see pna_amd.h).
__global__ void k_degree_scalers(const int32_t* rowptr, int V, float avg_log, float inv_avg, float* amp, float* att) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const int d = rowptr[v + 1] - rowptr[v];
  float fa = 0.f, ft = 0.f;
  if (d > 0) {
    const float lg = (float)log((double)d + 1.0);          // np.log(D + 1) in float64, then fp32
    fa = inv_avg * lg;                                     // avg.reciprocal() * scalar
    ft = avg_log / lg;                                     // avg / scalar
  }
  if (amp) amp[v] = fa;
  if (att) att[v] = ft;
}

template <int VEC, int U>
int launch_u(const KArgs& k, bool extra, dim3 grid, hipStream_t st) {
  if (extra) hipLaunchKernelGGL((k_segreduce<VEC, U, true>), grid, dim3(kBlock), 0, st, k);
  else hipLaunchKernelGGL((k_segreduce<VEC, U, false>), grid, dim3(kBlock), 0, st, k);
  return 0;
}

template <int VEC>
int launch_v(const KArgs& k, int U, bool extra, dim3 grid, hipStream_t st) {
  switch (U) {
    case 1: return launch_u<VEC, 1>(k, extra, grid, st);
    case 2: return launch_u<VEC, 2>(k, extra, grid, st);
    case 4: return launch_u<VEC, 4>(k, extra, grid, st);
    case 8: return launch_u<VEC, 8>(k, extra, grid, st);
  }
  return -1;
}

}  // namespace

extern "C" int64_t pna_segreduce_partials_bytes(int32_t n_seg, int32_t F, int32_t n_tower) {
  const int64_t ps = ((int64_t)F + 3) / 4 * 4;
  return (int64_t)n_seg * (n_tower > 1 ? n_tower : 1) * kNQ * ps * (int64_t)sizeof(float);
}

extern "C" int pna_segreduce_fwd_f32(const pna_segreduce_args* p, pna_stream_t stream) {
  if (!p) return pna_set_error(PNA_E_INVALID, "pna_segreduce_fwd_f32: null args");
  if (p->V < 0 || p->F <= 0) return pna_set_error(PNA_E_INVALID, "pna_segreduce_fwd_f32: bad V/F");
  if (p->V == 0) return PNA_OK;
  if (!p->rowptr || !p->x || !p->out) return pna_set_error(PNA_E_INVALID, "pna_segreduce_fwd_f32: rowptr/x/out must be non-null");
  if (p->n_aggr <= 0 || p->n_aggr > PNA_MAX_AGGR || p->n_scaler <= 0 || p->n_scaler > PNA_MAX_SCALER)
    return pna_set_error(PNA_E_INVALID, "pna_segreduce_fwd_f32: n_aggr/n_scaler out of range");
  for (int i = 0; i < p->n_aggr; ++i)
    if (p->aggr[i] < PNA_AGG_MEAN || p->aggr[i] > PNA_AGG_VAR)
      return pna_set_error(PNA_E_INVALID, "pna_segreduce_fwd_f32: unknown aggregator code");
  const int T = p->n_tower > 1 ? p->n_tower : 1;
  const int64_t ts_in = T > 1 ? p->tower_stride_in : 0, ts_out = T > 1 ? p->tower_stride_out : 0;
  if (T > 1 && (ts_in < p->F || ts_out < (int64_t)p->block_stride * (p->n_aggr * p->n_scaler - 1) + p->F))
    return pna_set_error(PNA_E_INVALID, "pna_segreduce_fwd_f32: tower strides too small");
  const int64_t in_w = ts_in * (T - 1) + p->F;             // columns touched in x / dst_term / edge_term
  const int64_t out_w = ts_out * (T - 1) + (int64_t)p->block_stride * (p->n_aggr * p->n_scaler - 1) + p->F;
  if (p->ldx < in_w || p->block_stride < p->F || p->ldo < out_w)
    return pna_set_error(PNA_E_INVALID, "pna_segreduce_fwd_f32: leading dimensions too small");
  if ((p->dst_term && p->ld_dst < in_w) || (p->edge_term && p->ld_edge < in_w) ||
      ((p->argmax || p->argmin) && p->ld_arg < in_w))
    return pna_set_error(PNA_E_INVALID, "pna_segreduce_fwd_f32: leading dimensions too small");
  const bool heavy = p->heavy_threshold > 0 && p->n_heavy > 0;
  if (heavy && (!p->heavy_rows || !p->heavy_segptr || !p->seg_heavy || !p->partials || p->seg_len <= 0 || p->n_seg <= 0))
    return pna_set_error(PNA_E_INVALID, "pna_segreduce_fwd_f32: incomplete heavy-row schedule");

  KArgs k;
  memset(&k, 0, sizeof(k));
  k.rowptr = p->rowptr; k.col = p->col; k.x = p->x; k.dst_term = p->dst_term; k.edge_term = p->edge_term;
  k.ew = p->edge_weight;
  for (int s = 0; s < p->n_scaler; ++s) k.row_scale[s] = p->row_scale[s];
  k.out = p->out; k.argmax = p->argmax; k.argmin = p->argmin;
  k.heavy_rows = p->heavy_rows; k.heavy_segptr = p->heavy_segptr; k.seg_heavy = p->seg_heavy; k.partials = p->partials;
  k.ldx = p->ldx; k.ld_dst = p->ld_dst; k.ld_edge = p->ld_edge; k.ldo = p->ldo; k.ld_arg = p->ld_arg;
  k.V = p->V; k.F = p->F; k.n_aggr = p->n_aggr; k.n_scaler = p->n_scaler; k.block_stride = p->block_stride;
  for (int i = 0; i < p->n_aggr; ++i) k.aggr[i] = p->aggr[i];
  k.heavy_threshold = heavy ? p->heavy_threshold : (p->heavy_threshold > 0 ? p->heavy_threshold : 0);
  k.seg_len = p->seg_len; k.n_heavy = heavy ? p->n_heavy : 0; k.n_seg = heavy ? p->n_seg : 0;
  // a threshold without a schedule would silently drop rows: only honour it when the lists exist
  if (!heavy) k.heavy_threshold = 0;

  const pna_tuning& t = p->tune;
  int vec = t.vec ? t.vec : (p->F >= 4 ? 4 : 1);
  if (vec != 1 && vec != 4) return pna_set_error(PNA_E_INVALID, "pna_segreduce_fwd_f32: tune.vec must be 1 or 4");
  if (vec == 4 && p->F < 4) vec = 1;
  const int nchunks = (p->F + vec - 1) / vec;
  int L = t.lanes_per_row;
  int tiles;
  if (L <= 0) {
    tiles = (nchunks + 63) / 64;
    L = (nchunks + tiles - 1) / tiles;
  } else {
    if (L > 64) return pna_set_error(PNA_E_INVALID, "pna_segreduce_fwd_f32: tune.lanes_per_row > 64");
    tiles = (nchunks + L - 1) / L;
  }
  k.L = L; k.G = 64 / L; k.T = T; k.tiles = tiles; k.ts_in = ts_in; k.ts_out = ts_out;
  int U = t.unroll ? t.unroll : 4;
  if (U != 1 && U != 2 && U != 4 && U != 8) return pna_set_error(PNA_E_INVALID, "pna_segreduce_fwd_f32: tune.unroll must be 1,2,4,8");
  k.R = t.rows_per_group > 0 ? t.rows_per_group : 4;
  k.nt = t.nt_store >= 0 ? 1 : 0;
  k.pstride = (p->F + 3) / 4 * 4;
  const int NG = kWaves * k.G;
  k.n_heavy_blocks = heavy ? (k.n_seg + NG - 1) / NG : 0;
  const long rows_per_block = (long)NG * k.R;
  const long light_blocks = (p->V + rows_per_block - 1) / rows_per_block;
  if (light_blocks + k.n_heavy_blocks > 0x7fffffffL) return pna_set_error(PNA_E_INVALID, "pna_segreduce_fwd_f32: grid too large");
  const bool extra = p->dst_term || p->edge_term || p->edge_weight || p->argmax || p->argmin;
  dim3 grid((unsigned)(light_blocks + k.n_heavy_blocks), (unsigned)(tiles * T));
  hipStream_t st = (hipStream_t)stream;
  int rc = vec == 4 ? launch_v<4>(k, U, extra, grid, st) : launch_v<1>(k, U, extra, grid, st);
  if (rc != 0) return pna_set_error(PNA_E_INVALID, "pna_segreduce_fwd_f32: no kernel for this tuning");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  if (heavy) {
    dim3 g2((unsigned)((k.n_heavy + NG - 1) / NG), (unsigned)(tiles * T));
    if (vec == 4) {
      if (extra) hipLaunchKernelGGL((k_heavy_finalize<4, true>), g2, dim3(kBlock), 0, st, k);
      else hipLaunchKernelGGL((k_heavy_finalize<4, false>), g2, dim3(kBlock), 0, st, k);
    } else {
      if (extra) hipLaunchKernelGGL((k_heavy_finalize<1, true>), g2, dim3(kBlock), 0, st, k);
      else hipLaunchKernelGGL((k_heavy_finalize<1, false>), g2, dim3(kBlock), 0, st, k);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  }
  return PNA_OK;
}

extern "C" int pna_degree_scalers_f32(const int32_t* rowptr, int32_t V, float avg_log, float* amp, float* att,
                                      pna_stream_t stream) {
  if (V < 0 || (V > 0 && !rowptr)) return pna_set_error(PNA_E_INVALID, "pna_degree_scalers_f32: bad arguments");
  if (V == 0) return PNA_OK;
  const float inv = 1.0f / avg_log;                        // Tensor.reciprocal() in fp32
  hipLaunchKernelGGL(k_degree_scalers, dim3((V + 255) / 256), dim3(256), 0, (hipStream_t)stream, rowptr, V, avg_log, inv,
                     amp, att);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}
