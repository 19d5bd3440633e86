// pna_segreduce_bwd.hip -- backward of the fused gather + multi-aggregator segment-reduce.
// Implements pna_segreduce_bwd_f32 of include/pna_amd.h.
//
// The reference gets this gradient from autograd through its per-bucket torch ops (torch.mean / max / min /
// sqrt(relu(E[x^2]-E[x]^2)+eps), models/dgl/aggregators.py:6-26).  With G_a[v,f] the gradient w.r.t. the
// UNSCALED aggregate a of destination v (the caller folds the degree scalers in: G_a = sum_s scale_s * dOut_sa),
// D the in-degree and m_k the message of in-edge k:
//     dL/dm_k = G_mean/D + G_sum + [k = argmax] G_max + [k = argmin] G_min
//             + (G_var + G_std / (2 std)) * [var > 0] * (2/D) * (m_k - mean)
// (relu'(0) = 0 and max/min route the gradient to the single index torch.max/min return).  m_k = x[col_k]
// + dst_term[v] + edge_term[k] is rebuilt by re-gathering; dL/dm_k is then scattered:
//     grad_x[col_k] += dL/dm_k   (hardware fp32 atomics; plain store when x is edge-resident)
//     grad_dst[v]   += sum_k dL/dm_k,    grad_edge[k] = dL/dm_k.
// Same lane mapping as the forward kernel (G lane groups per wavefront, one destination row per group, 4
// features per lane) and the same heavy-row segmentation, so hub rows are spread over many lane groups.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "pna_amd.h"
#include "pna_internal.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) f4u { f4 v; };
typedef int i4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) i4u { i4 v; };

constexpr int kBlock = 256;
constexpr int kWaves = 4;

struct BArgs {
  const int32_t* rowptr; const int32_t* col;
  const float* x; const float* dst_term; const float* edge_term;
  const float* g[6];                 // gradient block per aggregator CODE (mean,sum,max,min,std,var) or null
  const float* mean; const float* stdv; const float* var;
  const int32_t* argmax; const int32_t* argmin;
  float* grad_x; float* grad_dst; float* grad_edge;
  const int32_t* heavy_rows; const int32_t* heavy_segptr; const int32_t* seg_heavy;
  long ldx, ld_dst, ld_edge, ld_g, ld_stat, ld_arg, ld_gx, ld_gd, ld_ge, ts_in, ts_g, ts_stat;
  int V, F, heavy_threshold, seg_len, n_heavy, n_seg, L, G, R, n_heavy_blocks, T, tiles;
};

template <int VEC> struct V;
template <> struct V<4> {
  static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
    f4 t = reinterpret_cast<const f4u*>(p)->v; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void ldi(const int32_t* p, int (&v)[4]) {
    i4 t = reinterpret_cast<const i4u*>(p)->v; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[4]) {
    f4 t = {v[0], v[1], v[2], v[3]}; reinterpret_cast<f4u*>(p)->v = t;
  }
};
template <> struct V<1> {
  static __device__ __forceinline__ void ld(const float* p, float (&v)[1]) { v[0] = *p; }
  static __device__ __forceinline__ void ldi(const int32_t* p, int (&v)[1]) { v[0] = *p; }
  static __device__ __forceinline__ void st(float* p, const float (&v)[1]) { *p = v[0]; }
};

template <int VEC>
__global__ __launch_bounds__(kBlock) void k_segreduce_bwd(const BArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int L = a.L;
  const int grp = lane / L;
  if (grp >= a.G) return;
  const int c = lane - grp * L;
  const int grp_lane0 = grp * L;
  const int nchunks = (a.F + VEC - 1) / VEC;
  const int tower = blockIdx.y / a.tiles;
  const int chunk = (blockIdx.y - tower * a.tiles) * L + c;
  // The sliding last window overlaps its neighbour: lanes must not double-count overlapped features.  Feature
  // ownership: a lane owns component k iff its global feature index is >= VEC*chunk (its nominal start).
  const bool lane_ok = chunk < nchunks;
  int off = min(chunk, nchunks - 1) * VEC;
  const int nominal = off;
  if (VEC == 4) off = min(off, a.F - 4);
  const int skipk = lane_ok ? nominal - off : VEC;          // components [0, skipk) belong to the previous lane
  const long oin = (long)tower * a.ts_in + off;
  const long og = (long)tower * a.ts_g + off;
  const long ostat = (long)tower * a.ts_stat + off;
  const int NG = kWaves * a.G;
  const int gid = wave * a.G + grp;

  int row, beg, end;
  bool heavy_item = false;
  auto process = [&]() {
    const int rbeg = a.rowptr[row], rend = a.rowptr[row + 1];
    const float D = (float)(rend - rbeg);
    float base[VEC], cvar[VEC], mean[VEC], gmx[VEC], gmn[VEC], gd[VEC];
    int amx[VEC], amn[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { base[k] = 0.f; cvar[k] = 0.f; mean[k] = 0.f; gmx[k] = 0.f; gmn[k] = 0.f; gd[k] = 0.f; amx[k] = -1; amn[k] = -1; }
    float t[VEC];
    const size_t grow = (size_t)row * a.ld_g + og;
    if (a.g[PNA_AGG_MEAN]) { V<VEC>::ld(a.g[PNA_AGG_MEAN] + grow, t);
#pragma unroll
      for (int k = 0; k < VEC; ++k) base[k] = t[k] / D; }
    if (a.g[PNA_AGG_SUM]) { V<VEC>::ld(a.g[PNA_AGG_SUM] + grow, t);
#pragma unroll
      for (int k = 0; k < VEC; ++k) base[k] = base[k] + t[k]; }
    if (a.g[PNA_AGG_MAX]) { V<VEC>::ld(a.g[PNA_AGG_MAX] + grow, gmx); V<VEC>::ldi(a.argmax + (size_t)row * a.ld_arg + oin, amx); }
    if (a.g[PNA_AGG_MIN]) { V<VEC>::ld(a.g[PNA_AGG_MIN] + grow, gmn); V<VEC>::ldi(a.argmin + (size_t)row * a.ld_arg + oin, amn); }
    if (a.g[PNA_AGG_STD] || a.g[PNA_AGG_VAR]) {
      const size_t srow = (size_t)row * a.ld_stat + ostat;
      V<VEC>::ld(a.mean + srow, mean);
      float sd[VEC], vr[VEC], gs[VEC], gv[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) { sd[k] = 1.f; vr[k] = 0.f; gs[k] = 0.f; gv[k] = 0.f; }
      if (a.stdv) V<VEC>::ld(a.stdv + srow, sd);
      if (a.var) V<VEC>::ld(a.var + srow, vr);
      if (a.g[PNA_AGG_STD]) V<VEC>::ld(a.g[PNA_AGG_STD] + grow, gs);
      if (a.g[PNA_AGG_VAR]) V<VEC>::ld(a.g[PNA_AGG_VAR] + grow, gv);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float v = a.var ? vr[k] : sd[k] * sd[k] - 1e-5f;        // relu'(raw var): 0 at and below 0
        const float gsv = a.g[PNA_AGG_STD] ? gs[k] / (2.f * sd[k]) : 0.f;
        cvar[k] = v > 0.f ? (gv[k] + gsv) * (2.f / D) : 0.f;
      }
    }
    float dterm[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) dterm[k] = 0.f;
    if (a.dst_term) V<VEC>::ld(a.dst_term + (size_t)row * a.ld_dst + oin, dterm);
    const bool need_m = a.g[PNA_AGG_STD] || a.g[PNA_AGG_VAR];
    for (int cb = beg; cb < end; cb += L) {
      const int nidx = min(L, end - cb);
      int myidx = cb + c;
      if (a.col) myidx = c < nidx ? a.col[cb + c] : 0;
      for (int j = 0; j < nidx; j += 4) {
        int id[4];
        float m[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; ++u) id[u] = __shfl(myidx, grp_lane0 + min(j + u, nidx - 1));
        if (need_m) {
#pragma unroll
          for (int u = 0; u < 4; ++u) V<VEC>::ld(a.x + (size_t)id[u] * a.ldx + oin, m[u]);
          if (a.edge_term) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              float e[VEC];
              V<VEC>::ld(a.edge_term + (size_t)(cb + min(j + u, nidx - 1)) * a.ld_edge + oin, e);
#pragma unroll
              for (int k = 0; k < VEC; ++k) m[u][k] = (m[u][k] + dterm[k]) + e[k];
            }
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int k = 0; k < VEC; ++k) m[u][k] = m[u][k] + dterm[k];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (j + u >= nidx) break;
          const int e = cb + j + u;
          float gm[VEC];
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            float v = base[k];
            if (need_m) v = v + cvar[k] * (m[u][k] - mean[k]);
            if (e == amx[k]) v = v + gmx[k];
            if (e == amn[k]) v = v + gmn[k];
            gm[k] = v;
            gd[k] = gd[k] + v;
          }
          if (lane_ok) {
            if (a.grad_edge) {
              // overlapped components carry identical values; both lanes may store them
              V<VEC>::st(a.grad_edge + (size_t)e * a.ld_ge + oin, gm);
            }
            if (a.grad_x) {
              float* px = a.grad_x + (size_t)id[u] * a.ld_gx + oin;
              if (a.col) {
#pragma unroll
                for (int k = 0; k < VEC; ++k)
                  if (k >= skipk) unsafeAtomicAdd(px + k, gm[k]);
              } else {
                V<VEC>::st(px, gm);
              }
            }
          }
        }
      }
    }
    if (a.grad_dst && lane_ok) {
      float* pd = a.grad_dst + (size_t)row * a.ld_gd + oin;
      if (heavy_item) {
#pragma unroll
        for (int k = 0; k < VEC; ++k)
          if (k >= skipk) unsafeAtomicAdd(pd + k, gd[k]);
      } else {
        V<VEC>::st(pd, gd);
      }
    }
  };

  if ((int)blockIdx.x < a.n_heavy_blocks) {
    const int seg = blockIdx.x * NG + gid;
    if (seg >= a.n_seg) return;
    const int hi = a.seg_heavy[seg];
    row = a.heavy_rows[hi];
    const int sidx = seg - a.heavy_segptr[hi];
    beg = a.rowptr[row] + sidx * a.seg_len;
    end = min(beg + a.seg_len, a.rowptr[row + 1]);
    heavy_item = true;
    process();
    return;
  }
  const long base_row = (long)(blockIdx.x - a.n_heavy_blocks) * NG * a.R + gid;
  for (int r = 0; r < a.R; ++r) {
    const long row_l = base_row + (long)r * NG;
    if (row_l >= a.V) break;
    row = (int)row_l;
    beg = a.rowptr[row]; end = a.rowptr[row + 1];
    if (a.heavy_threshold > 0 && end - beg > a.heavy_threshold) continue;
    if (end == beg) {
      if (a.grad_dst && lane_ok) {
        float z[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) z[k] = 0.f;
        V<VEC>::st(a.grad_dst + (size_t)row * a.ld_gd + oin, z);
      }
      continue;
    }
    process();
  }
}


// ---- pull formulation (messages without a per-edge term) -------------------------------------------------------------
// dL/dm_k = R1[v] + R2[v] * x[u_k] + max/min terms, R2 = cvar, R1 = base + cvar * (dst_term - mean): one thread per
// (row, tower*F + f).  table[v] = [R1 (T*F) | R2 (T*F, only when std/var is present)]; grad_dst[v] = D*base + G_max + G_min.
struct PArgs {
  const int32_t* rowptr; const int32_t* col;
  const float* g[6]; const float* mean; const float* stdv; const float* var; const float* dst_term;
  const int32_t* argmax; const int32_t* argmin;
  float* table; float* grad_dst; float* grad_x;
  unsigned short* ranks;             // nullable: (V, ld_rank) [rank of argmax | rank of argmin] in the row's in-edge list (0xFFFF: none)
  const int32_t* row_of;             // nullable: row of mean / std / argmax / argmin that belongs to node v (a forward in degree-plan order)
  const int32_t* node_of; long n_rows;   // nullable: the pass walks the n_rows rows of mean / std / arg* IN THEIR ORDER, row r = node node_of[r] (< 0: padding)
  int in_place;                      // gagg == table (pna_segreduce_bwd_pull_f32, in place)
  float* gcopy;                      // nullable: packed table rows -- G_max | G_min copied to table[v][2 T F .. 4 T F) (pna_segreduce_bwd_pull_f32)
  long ld_g, ld_stat, ld_dst, ld_table, ld_gd, ld_arg, ld_gx, ts_in, ts_g, ts_stat, ld_rank;
  int V, F, T, has_var;
};

__device__ __forceinline__ void rowprep_elem(const PArgs& a, int v, long vs, int t, int f) {      // vs: the row of the statistics / arg indices
  const int TF = a.T * a.F, c = t * a.F + f;
  const float D = (float)(a.rowptr[v + 1] - a.rowptr[v]);
  const size_t og = (size_t)v * a.ld_g + (size_t)t * a.ts_g + f;
  float base = 0.f;
  if (D > 0.f) {
    if (a.g[PNA_AGG_MEAN]) base = a.g[PNA_AGG_MEAN][og] / D;
    if (a.g[PNA_AGG_SUM]) base = base + a.g[PNA_AGG_SUM][og];
  }
  float r1 = base;
  if (a.has_var) {
    float cvar = 0.f;
    const size_t os = (size_t)vs * a.ld_stat + (size_t)t * a.ts_stat + f;
    if (D > 0.f) {
      const float sd = a.stdv ? a.stdv[os] : 1.f;
      const float vr = a.var ? a.var[os] : sd * sd - 1e-5f;              // relu'(raw var): 0 at and below 0
      float gs = a.g[PNA_AGG_VAR] ? a.g[PNA_AGG_VAR][og] : 0.f;
      if (a.g[PNA_AGG_STD]) gs = gs + a.g[PNA_AGG_STD][og] / (2.f * sd);
      cvar = vr > 0.f ? gs * (2.f / D) : 0.f;
      const float dt = a.dst_term ? a.dst_term[(size_t)v * a.ld_dst + (size_t)t * a.ts_in + f] : 0.f;
      r1 = base + cvar * (dt - a.mean[os]);
    }
    a.table[(size_t)v * a.ld_table + TF + c] = cvar;
  }
  a.table[(size_t)v * a.ld_table + c] = r1;
  if (a.grad_dst) {
    float gd = D * base;
    if (D > 0.f) {
      if (a.g[PNA_AGG_MAX]) gd = gd + a.g[PNA_AGG_MAX][og];
      if (a.g[PNA_AGG_MIN]) gd = gd + a.g[PNA_AGG_MIN][og];
    }
    a.grad_dst[(size_t)v * a.ld_gd + c] = gd;
  }
  if (a.gcopy) {                     // packed rows: the pull then reads ONE contiguous 20 T F-byte row per out-edge instead of three pieces
    a.gcopy[(size_t)v * a.ld_table + c] = a.g[PNA_AGG_MAX] ? a.g[PNA_AGG_MAX][og] : 0.f;
    a.gcopy[(size_t)v * a.ld_table + TF + c] = a.g[PNA_AGG_MIN] ? a.g[PNA_AGG_MIN][og] : 0.f;
  }
  if (a.ranks) {                     // for pna_segreduce_bwd_pull_f32: where in the row's in-edge list argmax / argmin sit
    const size_t oa = (size_t)vs * a.ld_arg + (size_t)t * a.ts_in + f;
    const int beg = a.rowptr[v];
    const int ex = a.argmax ? a.argmax[oa] : -1, en = a.argmin ? a.argmin[oa] : -1;
    a.ranks[(size_t)v * a.ld_rank + c] = ex < 0 ? (unsigned short)0xFFFF : (unsigned short)(ex - beg);
    a.ranks[(size_t)v * a.ld_rank + TF + c] = en < 0 ? (unsigned short)0xFFFF : (unsigned short)(en - beg);
  }
}

__global__ __launch_bounds__(kBlock) void k_bwd_rowprep(const PArgs a) {
  const int TF = a.T * a.F;
  const long i = (long)blockIdx.x * kBlock + threadIdx.x;
  const long R = a.node_of ? a.n_rows : (long)a.V;
  if (i >= R * TF) return;
  const long r = i / TF;
  const int c = (int)(i - r * TF);
  const int v = a.node_of ? a.node_of[r] : (int)r;
  if (v < 0) return;
  const int t = c / a.F;
  rowprep_elem(a, v, a.node_of ? r : (a.row_of ? (long)a.row_of[v] : (long)v), t, c - t * a.F);
}

// The same pass with FOUR features per thread (F >= 4; 16-byte accesses, the row's last window slides back to [F - 4, F) and rewrites
// the columns it shares with its neighbour with the same values): the one-element kernel above issues 14 dword memory
// instructions per element and ran at 4.0 TB/s on the 3.9 GB of the packed C3 pass (0.97 ms); the same arithmetic, op by op.
__global__ __launch_bounds__(kBlock) void k_bwd_rowprep4(const PArgs a) {
  const int nch = (a.F + 3) / 4, TC = a.T * nch;
  const long i = (long)blockIdx.x * kBlock + threadIdx.x;
  const long R = a.node_of ? a.n_rows : (long)a.V;
  if (i >= R * TC) return;
  const long r = i / TC;
  const int ct = (int)(i - r * TC);
  const int v = a.node_of ? a.node_of[r] : (int)r;
  if (v < 0) return;
  const long vs = a.node_of ? r : (a.row_of ? (long)a.row_of[v] : (long)v);
  const int t = ct / nch, f = min((ct - t * nch) * 4, a.F - 4);
  if (a.in_place && (ct - t * nch) * 4 + 4 > a.F) {
    // in place (gagg == table) the sliding last window would read columns its neighbour has already overwritten: the tail's own
    // 1..3 columns one by one instead
    for (int ff = (ct - t * nch) * 4; ff < a.F; ++ff) rowprep_elem(a, v, vs, t, ff);
    return;
  }
  const int TF = a.T * a.F, c = t * a.F + f;
  struct __attribute__((packed, aligned(4))) i4u { i4 v; };
  typedef unsigned short us4 __attribute__((ext_vector_type(4)));
  struct __attribute__((packed, aligned(2))) us4u { us4 v; };
  auto ld = [](const float* p) -> f4 { return reinterpret_cast<const f4u*>(p)->v; };
  auto st = [](float* p, f4 x) { reinterpret_cast<f4u*>(p)->v = x; };
  const float D = (float)(a.rowptr[v + 1] - a.rowptr[v]);
  const size_t og = (size_t)v * a.ld_g + (size_t)t * a.ts_g + f;
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  f4 base = zero;
  if (D > 0.f) {
    if (a.g[PNA_AGG_MEAN]) { const f4 gm = ld(a.g[PNA_AGG_MEAN] + og); for (int q = 0; q < 4; ++q) base[q] = gm[q] / D; }
    if (a.g[PNA_AGG_SUM]) { const f4 gs = ld(a.g[PNA_AGG_SUM] + og); for (int q = 0; q < 4; ++q) base[q] = base[q] + gs[q]; }
  }
  f4 r1 = base;
  if (a.has_var) {
    f4 cvar = zero;
    const size_t os = (size_t)vs * a.ld_stat + (size_t)t * a.ts_stat + f;
    if (D > 0.f) {
      const f4 one = {1.f, 1.f, 1.f, 1.f};
      const f4 sd = a.stdv ? ld(a.stdv + os) : one;
      f4 vr;
      if (a.var) vr = ld(a.var + os);
      else for (int q = 0; q < 4; ++q) vr[q] = sd[q] * sd[q] - 1e-5f;
      f4 gs = a.g[PNA_AGG_VAR] ? ld(a.g[PNA_AGG_VAR] + og) : zero;
      if (a.g[PNA_AGG_STD]) { const f4 gd = ld(a.g[PNA_AGG_STD] + og); for (int q = 0; q < 4; ++q) gs[q] = gs[q] + gd[q] / (2.f * sd[q]); }
      const f4 dt = a.dst_term ? ld(a.dst_term + (size_t)v * a.ld_dst + (size_t)t * a.ts_in + f) : zero;
      const f4 mean = ld(a.mean + os);
      for (int q = 0; q < 4; ++q) {
        cvar[q] = vr[q] > 0.f ? gs[q] * (2.f / D) : 0.f;
        r1[q] = base[q] + cvar[q] * (dt[q] - mean[q]);
      }
    }
    st(a.table + (size_t)v * a.ld_table + TF + c, cvar);
  }
  st(a.table + (size_t)v * a.ld_table + c, r1);
  f4 gmx = zero, gmn = zero;
  if (a.g[PNA_AGG_MAX] && (a.gcopy || (a.grad_dst && D > 0.f))) gmx = ld(a.g[PNA_AGG_MAX] + og);
  if (a.g[PNA_AGG_MIN] && (a.gcopy || (a.grad_dst && D > 0.f))) gmn = ld(a.g[PNA_AGG_MIN] + og);
  if (a.grad_dst) {
    f4 gd;
    for (int q = 0; q < 4; ++q) {
      gd[q] = D * base[q];
      if (D > 0.f) {
        if (a.g[PNA_AGG_MAX]) gd[q] = gd[q] + gmx[q];
        if (a.g[PNA_AGG_MIN]) gd[q] = gd[q] + gmn[q];
      }
    }
    st(a.grad_dst + (size_t)v * a.ld_gd + c, gd);
  }
  if (a.gcopy) {
    st(a.gcopy + (size_t)v * a.ld_table + c, gmx);
    st(a.gcopy + (size_t)v * a.ld_table + TF + c, gmn);
  }
  if (a.ranks) {
    const size_t oa = (size_t)vs * a.ld_arg + (size_t)t * a.ts_in + f;
    const int beg = a.rowptr[v];
    const i4 none = {-1, -1, -1, -1};
    const i4 ex = a.argmax ? reinterpret_cast<const i4u*>(a.argmax + oa)->v : none;
    const i4 en = a.argmin ? reinterpret_cast<const i4u*>(a.argmin + oa)->v : none;
    us4 kx, kn;
    for (int q = 0; q < 4; ++q) {
      kx[q] = ex[q] < 0 ? (unsigned short)0xFFFF : (unsigned short)(ex[q] - beg);
      kn[q] = en[q] < 0 ? (unsigned short)0xFFFF : (unsigned short)(en[q] - beg);
    }
    reinterpret_cast<us4u*>(a.ranks + (size_t)v * a.ld_rank + c)->v = kx;
    reinterpret_cast<us4u*>(a.ranks + (size_t)v * a.ld_rank + TF + c)->v = kn;
  }
}

void launch_rowprep(PArgs k, hipStream_t st, bool in_place = false) {
  k.in_place = in_place ? 1 : 0;
  const long R = k.node_of ? k.n_rows : (long)k.V;
  if (R <= 0) return;
  if (k.F >= 4) {
    const long n = R * k.T * ((k.F + 3) / 4);
    hipLaunchKernelGGL(k_bwd_rowprep4, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, k);
  } else {
    const long n = R * k.T * k.F;
    hipLaunchKernelGGL(k_bwd_rowprep, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, k);
  }
}

// grad_x[col[arg[v,c]]][c] += G[v,c] for max and min: V*T*F atomics each instead of E*T*F
__global__ __launch_bounds__(kBlock) void k_bwd_argscatter(const PArgs a) {
  const int TF = a.T * a.F;
  const long i = (long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (long)a.V * TF) return;
  const int v = (int)(i / TF), c = (int)(i - (long)v * TF);
  const int t = c / a.F, f = c - t * a.F;
  const size_t og = (size_t)v * a.ld_g + (size_t)t * a.ts_g + f;
  const size_t oa = (size_t)v * a.ld_arg + (size_t)t * a.ts_in + f;
  if (a.g[PNA_AGG_MAX]) {
    const int e = a.argmax[oa];
    if (e >= 0) unsafeAtomicAdd(a.grad_x + (size_t)a.col[e] * a.ld_gx + c, a.g[PNA_AGG_MAX][og]);
  }
  if (a.g[PNA_AGG_MIN]) {
    const int e = a.argmin[oa];
    if (e >= 0) unsafeAtomicAdd(a.grad_x + (size_t)a.col[e] * a.ld_gx + c, a.g[PNA_AGG_MIN][og]);
  }
}

// ---- the max / min terms WITHOUT atomics (round 3): one pull over the transposed graph does everything ---------------------
// grad_x[u] = sum over out-edges (u -> v), the k-th in-edge of v:  R1[v] + [k = rank of argmax[v]] G_max[v] + [k = rank of argmin[v]] G_min[v]
//             + x[u] * sum R2[v].
// Per out-edge a lane reads its 4 features of R1, R2 (the rowprep table), G_max, G_min (the aggregate's gradient) and the two
// 16-bit ranks: 1500 bytes per edge at F = 75 instead of the 600 of the sums-only pull -- against 150 M scattered atomics.  Lane
// mapping of the forward kernel: G lane groups per wavefront, one work-list record per group (a source row, or a segment of a hub
// source whose partial result is added atomically into the pre-zeroed row), 4 features per lane.
struct LArgs {
  const int32_t* items; const int32_t* col_t; const int32_t* rank_t;
  const float* table; const float* gmax; const float* gmin; const unsigned short* ranks; const float* x; float* gx;
  long ld_table, ld_g, ld_rank, ldx, ld_gx, ts_g;
  int n_items, F, T, TF, L, G;
};

__global__ __launch_bounds__(kBlock) void k_bwd_ranks(const int32_t* rowptr, const int32_t* argmax, const int32_t* argmin, long ld_arg, long ts_in,
                                                      int V, int F, int T, unsigned short* ranks, long ld_rank) {
  const int TF = T * F;
  const long i = (long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (long)V * TF) return;
  const int v = (int)(i / TF), c = (int)(i - (long)v * TF);
  const int t = c / F, f = c - t * F;
  const size_t oa = (size_t)v * ld_arg + (size_t)t * ts_in + f;
  const int beg = rowptr[v];
  const int ex = argmax ? argmax[oa] : -1, en = argmin ? argmin[oa] : -1;
  ranks[(size_t)v * ld_rank + c] = ex < 0 ? (unsigned short)0xFFFF : (unsigned short)(ex - beg);
  ranks[(size_t)v * ld_rank + TF + c] = en < 0 ? (unsigned short)0xFFFF : (unsigned short)(en - beg);
}

template <int U>
__global__ __launch_bounds__(kBlock) void k_bwd_pull(const LArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane / a.L;
  if (grp >= a.G) return;
  const int c = lane - grp * a.L;
  const int nchunks = (a.F + 3) / 4;
  const int tower = blockIdx.y;
  const bool lane_ok = c < nchunks;
  const int off = min(min(c, nchunks - 1) * 4, a.F - 4);
  const long item = ((long)blockIdx.x * kWaves + wave) * a.G + grp;
  if (item >= a.n_items) return;
  const i4 rec = reinterpret_cast<const i4*>(a.items)[item];
  const int row = rec.x, beg = rec.y, end = rec.z, slot = rec.w;
  const long ot = (long)tower * a.F + off;                  // column inside a [T F] half of the table / ranks / x / grad_x
  const long og = (long)tower * a.ts_g + off;               // column inside the aggregate's gradient (per aggregator block)
  f4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
  typedef unsigned short us4 __attribute__((ext_vector_type(4)));
  struct __attribute__((packed, aligned(2))) us4u { us4 v; };
  for (int e = beg; e < end; e += U) {
    int v[U], k[U];
    f4 r1[U], r2[U], gx[U], gn[U];
    us4 kx[U], kn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = min(e + u, end - 1);
      v[u] = a.col_t[j];
      k[u] = a.rank_t[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float* tr = a.table + (size_t)v[u] * a.ld_table + ot;
      r1[u] = reinterpret_cast<const f4u*>(tr)->v;
      r2[u] = reinterpret_cast<const f4u*>(tr + a.TF)->v;
      gx[u] = reinterpret_cast<const f4u*>(a.gmax + (size_t)v[u] * a.ld_g + og)->v;
      gn[u] = reinterpret_cast<const f4u*>(a.gmin + (size_t)v[u] * a.ld_g + og)->v;
      const unsigned short* rr = a.ranks + (size_t)v[u] * a.ld_rank + ot;
      kx[u] = reinterpret_cast<const us4u*>(rr)->v;
      kn[u] = reinterpret_cast<const us4u*>(rr + a.TF)->v;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (e + u < end) {                                    // (group-uniform)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float t = r1[u][q];
          t = t + ((int)kx[u][q] == k[u] ? gx[u][q] : 0.f);
          t = t + ((int)kn[u][q] == k[u] ? gn[u][q] : 0.f);
          s1[q] = s1[q] + t;
          s2[q] = s2[q] + r2[u][q];
        }
      }
    }
  }
  if (!lane_ok) return;
  const f4 xu = reinterpret_cast<const f4u*>(a.x + (size_t)row * a.ldx + ot)->v;
  f4 res;
#pragma unroll
  for (int q = 0; q < 4; ++q) res[q] = s1[q] + xu[q] * s2[q];
  float* o = a.gx + (size_t)row * a.ld_gx + ot;
  if (slot < 0) {
    reinterpret_cast<f4u*>(o)->v = res;                     // (a row's last, overlapping window recomputes the same values)
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (off + q >= c * 4) unsafeAtomicAdd(o + q, res[q]);   // hub segment: every column once (the last window overlaps its neighbour)
  }
}

// ---- round 6: the pull over PER-EDGE rows ------------------------------------------------------------------------------------
// The ranked pull above reads, per out-edge (u -> v), R1 | R2 | G_max | G_min of v and two rank rows: 1 568 bytes at F = 75, of which the
// max / min part (G rows + ranks: 900 bytes) only decides which ONE in-edge of v per feature receives G_max / G_min.  Here that decision is
// taken where it is cheap -- on the destination's side, walking v's in-edges in sequence (the forward work list) --:
//     P[e] = R1[v] + [e = argmax[v]] G_max[v] + [e = argmin[v]] G_min[v]        one F-float row per edge, written in CSR order
// and the pull reads per out-edge P[position of the edge] and R2[v]: 2 x 4F bytes.  The same additions in the same order as the ranked pull
// (t = R1; t += max term; t += min term; s1 += t; s2 += R2): bit-identical gradients.  One tower, no destination term (PNASimpleLayer).
#ifndef PNA_EDGE_ROWS_WHOLE
#define PNA_EDGE_ROWS_WHOLE 1          // (0: rows written through the load windows -- the A/B build of tools/fastbuild.sh _nowhole -DPNA_EDGE_ROWS_WHOLE=0)
#endif
struct EArgs {
  const int32_t* items; const int32_t* rowptr; const int32_t* row_of;
  const float* gmean; const float* gstd; const float* gmax; const float* gmin; const float* mean; const float* stdv;
  const int32_t* amx; const int32_t* amn;
  float* r2; float* P;
  long ld_g, ld_stat, ld_arg, ld_r2, ld_p;
  int n_items, F, L, G;
};
__global__ __launch_bounds__(kBlock) void k_bwd_edge_rows(const EArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane / a.L;
  if (grp >= a.G) return;
  const int c = lane - grp * a.L;
  const int nchunks = (a.F + 3) / 4;
  // whole_rows (L = ld_p / 4 lanes): every lane stores the 16 ALIGNED bytes at column 4 c, the row's padding columns included (zeros) -- an edge
  // row is then ld_p x 4 contiguous, fully written bytes.  With the windows of the loads (the last one slid back to end at F) a 300-byte row
  // of pitch 320 left 20 bytes unwritten: one partially written 32-byte sector per edge, which the memory side completes by reading it first.
  const bool whole_rows = a.L > nchunks;
  if (c >= nchunks && !whole_rows) return;
  const int off = min(min(c, nchunks - 1) * 4, a.F - 4);
  const int sc = whole_rows ? c * 4 : off, sd_ = sc - off;   // store column; how far the load window sits below it (0..3; >= 4: pure padding)
  const long item = ((long)blockIdx.x * kWaves + wave) * a.G + grp;
  if (item >= a.n_items) return;
  const i4 rec = reinterpret_cast<const i4*>(a.items)[item];
  const int row = rec.x, beg = rec.y, end = rec.z;
  const int rbeg = a.rowptr[row];
  const float D = (float)(a.rowptr[row + 1] - rbeg);
  const long vs = a.row_of ? (long)a.row_of[row] : (long)row;
  struct __attribute__((packed, aligned(4))) i4u { i4 v; };
  auto ld = [](const float* p) -> f4 { return reinterpret_cast<const f4u*>(p)->v; };
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  f4 r1 = zero, cvar = zero, gmx = zero, gmn = zero;
  i4 ex = {-1, -1, -1, -1}, en = ex;
  if (D > 0.f) {
    const size_t og = (size_t)row * a.ld_g + off, os = (size_t)vs * a.ld_stat + off, oa = (size_t)vs * a.ld_arg + off;
    const f4 gm = ld(a.gmean + og), gd = ld(a.gstd + og), sd = ld(a.stdv + os), mean = ld(a.mean + os);
    gmx = ld(a.gmax + og); gmn = ld(a.gmin + og);
    ex = reinterpret_cast<const i4u*>(a.amx + oa)->v; en = reinterpret_cast<const i4u*>(a.amn + oa)->v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {                           // (k_bwd_rowprep4's arithmetic, op by op)
      const float base = gm[q] / D;
      const float vr = sd[q] * sd[q] - 1e-5f;
      const float gs = 0.f + gd[q] / (2.f * sd[q]);
      cvar[q] = vr > 0.f ? gs * (2.f / D) : 0.f;
      r1[q] = base + cvar[q] * (0.f - mean[q]);
    }
  }
  if (beg == rbeg && c < nchunks) reinterpret_cast<f4u*>(a.r2 + (size_t)row * a.ld_r2 + off)->v = cvar;      // (a hub row's first segment writes the row's R2)
  if (whole_rows) {
    // the window's values moved up to the store column ONCE, in front of the edge loop (zeros / no arg index behind column F): the loop is
    // two compares, two selects and two additions per column and one aligned 16-byte store
    auto upf = [&](f4 t, int q) { const int j = q + sd_; return (j < 4 && sc + q < a.F) ? (j == 0 ? t.x : j == 1 ? t.y : j == 2 ? t.z : t.w) : 0.f; };
    auto upi = [&](i4 t, int q) { const int j = q + sd_; return (j < 4 && sc + q < a.F) ? (j == 0 ? t.x : j == 1 ? t.y : j == 2 ? t.z : t.w) : -1; };
    const f4 r1s = {upf(r1, 0), upf(r1, 1), upf(r1, 2), upf(r1, 3)}, gxs = {upf(gmx, 0), upf(gmx, 1), upf(gmx, 2), upf(gmx, 3)},
             gns = {upf(gmn, 0), upf(gmn, 1), upf(gmn, 2), upf(gmn, 3)};
    const i4 exs = {upi(ex, 0), upi(ex, 1), upi(ex, 2), upi(ex, 3)}, ens = {upi(en, 0), upi(en, 1), upi(en, 2), upi(en, 3)};
    float* pe = a.P + (size_t)beg * a.ld_p + sc;
    for (int e = beg; e < end; ++e, pe += a.ld_p) {
      f4 t;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v = r1s[q];
        v = v + (exs[q] == e ? gxs[q] : 0.f);
        v = v + (ens[q] == e ? gns[q] : 0.f);
        t[q] = v;
      }
      *reinterpret_cast<f4*>(pe) = t;
    }
    return;
  }
  for (int e = beg; e < end; ++e) {
    f4 t;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v = r1[q];
      v = v + (ex[q] == e ? gmx[q] : 0.f);
      v = v + (en[q] == e ? gmn[q] : 0.f);
      t[q] = v;
    }
    reinterpret_cast<f4u*>(a.P + (size_t)e * a.ld_p + off)->v = t;
  }
}

struct LArgs2 {
  const int32_t* items; const int32_t* col_t; const int32_t* pos_t;
  const float* P; const float* r2; const float* x; float* gx;
  long ld_p, ld_r2, ldx, ld_gx;
  int n_items, F, L, G;
};
template <int U>
__global__ __launch_bounds__(kBlock) void k_bwd_pull_rows(const LArgs2 a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane / a.L;
  if (grp >= a.G) return;
  const int c = lane - grp * a.L;
  const int nchunks = (a.F + 3) / 4;
  const bool lane_ok = c < nchunks;
  const int off = min(min(c, nchunks - 1) * 4, a.F - 4);
  const long item = ((long)blockIdx.x * kWaves + wave) * a.G + grp;
  if (item >= a.n_items) return;
  const i4 rec = reinterpret_cast<const i4*>(a.items)[item];
  const int row = rec.x, beg = rec.y, end = rec.z, slot = rec.w;
  f4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
  int v[U], ps[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    v[u] = 0; ps[u] = 0;
    if (beg < end) {                                        // (a row without out-edges reads no id: beg may be the end of the arrays)
      const int j = min(beg + u, end - 1);
      v[u] = a.col_t[j];
      ps[u] = a.pos_t[j];
    }
  }
  for (int e = beg; e < end; e += U) {
    f4 pr[U], r2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      pr[u] = reinterpret_cast<const f4u*>(a.P + (size_t)ps[u] * a.ld_p + off)->v;
      r2[u] = reinterpret_cast<const f4u*>(a.r2 + (size_t)v[u] * a.ld_r2 + off)->v;
    }
    // the NEXT batch's ids, requested behind this batch's rows: they have landed by the time these rows are summed (round 6: the ids of a
    // batch used to be requested at its top -- two dependent round trips per four edges; 1.39 -> 1.29 ms at C3, same sums in the same order)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = min(e + U + u, end - 1);
      v[u] = a.col_t[j];
      ps[u] = a.pos_t[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (e + u < end) {                                    // (group-uniform)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          s1[q] = s1[q] + pr[u][q];
          s2[q] = s2[q] + r2[u][q];
        }
      }
    }
  }
  if (!lane_ok) return;
  const f4 xu = reinterpret_cast<const f4u*>(a.x + (size_t)row * a.ldx + off)->v;
  f4 res;
#pragma unroll
  for (int q = 0; q < 4; ++q) res[q] = s1[q] + xu[q] * s2[q];
  float* o = a.gx + (size_t)row * a.ld_gx + off;
  if (slot < 0) {
    reinterpret_cast<f4u*>(o)->v = res;                     // (a row's last, overlapping window recomputes the same values)
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (off + q >= c * 4) unsafeAtomicAdd(o + q, res[q]);   // hub segment: every column once (the last window overlaps its neighbour)
  }
}

int fill_pull_args(const pna_segreduce_bwd_args* p, PArgs& k, const char* who) {
  memset(&k, 0, sizeof(k));
  if (!p) return pna_set_error(PNA_E_INVALID, "null args");
  if (int rc_ss = pna_check_struct_size(who, p->struct_size, sizeof(*p))) return rc_ss;
  if (p->V < 0 || p->F <= 0 || !p->rowptr || !p->gagg || p->n_aggr <= 0 || p->n_aggr > PNA_MAX_AGGR) return pna_set_error(PNA_E_INVALID, who);
  const int T = p->n_tower > 1 ? p->n_tower : 1;
  for (int i = 0; i < p->n_aggr; ++i) {
    const int code = p->aggr[i];
    if (code < PNA_AGG_MEAN || code > PNA_AGG_VAR || k.g[code]) return pna_set_error(PNA_E_INVALID, who);
    k.g[code] = p->gagg + (int64_t)i * p->F;
  }
  k.has_var = k.g[PNA_AGG_STD] || k.g[PNA_AGG_VAR];
  if (k.has_var && (!p->mean || !(p->stdv || p->var) || (k.g[PNA_AGG_STD] && !p->stdv))) return pna_set_error(PNA_E_INVALID, who);
  k.rowptr = p->rowptr; k.col = p->col; k.mean = p->mean; k.stdv = p->stdv; k.var = p->var; k.dst_term = p->dst_term;
  k.argmax = p->argmax; k.argmin = p->argmin; k.grad_dst = p->grad_dst; k.grad_x = p->grad_x;
  k.ld_g = p->ld_g; k.ld_stat = p->ld_stat; k.ld_dst = p->ld_dst; k.ld_gd = p->ld_gd; k.ld_arg = p->ld_arg; k.ld_gx = p->ld_gx;
  k.ts_in = T > 1 ? p->tower_stride_in : 0; k.ts_g = T > 1 ? p->tower_stride_g : 0; k.ts_stat = T > 1 ? p->tower_stride_stat : 0;
  k.V = p->V; k.F = p->F; k.T = T;
  k.row_of = p->stat_row_of;
  k.node_of = p->stat_node_of; k.n_rows = (long)p->stat_rows;
  if (p->stat_node_of && p->stat_rows < 0) return pna_set_error(PNA_E_INVALID, who);
  return PNA_OK;
}

}  // namespace

extern "C" int pna_segreduce_bwd_f32(const pna_segreduce_bwd_args* p, pna_stream_t stream) {
  if (!p) return pna_set_error(PNA_E_INVALID, "pna_segreduce_bwd_f32: null args");
  if (int rc_ss = pna_check_struct_size("pna_segreduce_bwd_f32", p->struct_size, sizeof(*p))) return rc_ss;
  if (p->V < 0 || p->F <= 0) return pna_set_error(PNA_E_INVALID, "pna_segreduce_bwd_f32: bad V/F");
  if (p->stat_row_of || p->stat_node_of) return pna_set_error(PNA_E_INVALID, "pna_segreduce_bwd_f32: stat_row_of / stat_node_of are honoured by the rowprep / pull entry points only");
  if (p->V == 0) return PNA_OK;
  if (!p->rowptr || !p->gagg) return pna_set_error(PNA_E_INVALID, "pna_segreduce_bwd_f32: rowptr/gagg must be non-null");
  if (p->n_aggr <= 0 || p->n_aggr > PNA_MAX_AGGR) return pna_set_error(PNA_E_INVALID, "pna_segreduce_bwd_f32: n_aggr out of range");
  BArgs k;
  memset(&k, 0, sizeof(k));
  const int T = p->n_tower > 1 ? p->n_tower : 1;
  bool need_stat = false, need_x = false;
  for (int i = 0; i < p->n_aggr; ++i) {
    const int code = p->aggr[i];
    if (code < PNA_AGG_MEAN || code > PNA_AGG_VAR) return pna_set_error(PNA_E_INVALID, "pna_segreduce_bwd_f32: unknown aggregator code");
    if (k.g[code]) return pna_set_error(PNA_E_INVALID, "pna_segreduce_bwd_f32: duplicate aggregator");
    k.g[code] = p->gagg + (int64_t)i * p->F;
    if (code == PNA_AGG_STD || code == PNA_AGG_VAR) need_stat = need_x = true;
    if (code == PNA_AGG_MAX && !p->argmax) return pna_set_error(PNA_E_INVALID, "pna_segreduce_bwd_f32: max needs argmax");
    if (code == PNA_AGG_MIN && !p->argmin) return pna_set_error(PNA_E_INVALID, "pna_segreduce_bwd_f32: min needs argmin");
  }
  if (need_stat && (!p->mean || !(p->stdv || p->var) || (k.g[PNA_AGG_STD] && !p->stdv)))
    return pna_set_error(PNA_E_INVALID, "pna_segreduce_bwd_f32: std/var need the forward mean and std (or var)");
  if (need_x && !p->x) return pna_set_error(PNA_E_INVALID, "pna_segreduce_bwd_f32: std/var need the forward input x");
  const bool heavy = p->heavy_threshold > 0 && p->n_heavy > 0;
  if (heavy && (!p->heavy_rows || !p->heavy_segptr || !p->seg_heavy || p->seg_len <= 0 || p->n_seg <= 0))
    return pna_set_error(PNA_E_INVALID, "pna_segreduce_bwd_f32: incomplete heavy-row schedule");
  k.rowptr = p->rowptr; k.col = p->col; k.x = p->x; k.dst_term = need_x ? p->dst_term : nullptr;
  k.edge_term = need_x ? p->edge_term : nullptr;
  k.mean = p->mean; k.stdv = p->stdv; k.var = p->var; k.argmax = p->argmax; k.argmin = p->argmin;
  k.grad_x = p->grad_x; k.grad_dst = p->grad_dst; k.grad_edge = p->grad_edge;
  k.heavy_rows = p->heavy_rows; k.heavy_segptr = p->heavy_segptr; k.seg_heavy = p->seg_heavy;
  k.ldx = p->ldx; k.ld_dst = p->ld_dst; k.ld_edge = p->ld_edge; k.ld_g = p->ld_g; k.ld_stat = p->ld_stat; k.ld_arg = p->ld_arg;
  k.ld_gx = p->ld_gx; k.ld_gd = p->ld_gd; k.ld_ge = p->ld_ge;
  k.ts_in = T > 1 ? p->tower_stride_in : 0; k.ts_g = T > 1 ? p->tower_stride_g : 0; k.ts_stat = T > 1 ? p->tower_stride_stat : 0;
  k.V = p->V; k.F = p->F; k.T = T;
  k.heavy_threshold = heavy ? p->heavy_threshold : 0; k.seg_len = p->seg_len; k.n_heavy = heavy ? p->n_heavy : 0;
  k.n_seg = heavy ? p->n_seg : 0;
  const int vec = p->F >= 4 ? 4 : 1;
  const int nchunks = (p->F + vec - 1) / vec;
  const int tiles = (nchunks + 63) / 64;
  k.L = (nchunks + tiles - 1) / tiles; k.G = 64 / k.L; k.tiles = tiles; k.R = 4;
  const int NG = kWaves * k.G;
  k.n_heavy_blocks = heavy ? (k.n_seg + NG - 1) / NG : 0;
  const long light_blocks = (p->V + (long)NG * k.R - 1) / ((long)NG * k.R);
  dim3 grid((unsigned)(light_blocks + k.n_heavy_blocks), (unsigned)(tiles * T));
  if (vec == 4) hipLaunchKernelGGL((k_segreduce_bwd<4>), grid, dim3(kBlock), 0, (hipStream_t)stream, k);
  else hipLaunchKernelGGL((k_segreduce_bwd<1>), grid, dim3(kBlock), 0, (hipStream_t)stream, k);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}

extern "C" int pna_segreduce_bwd_rowprep_f32(const pna_segreduce_bwd_args* p, float* table, int64_t ld_table, pna_stream_t stream) {
  PArgs k;
  int rc = fill_pull_args(p, k, "pna_segreduce_bwd_rowprep_f32: bad arguments");
  if (rc != PNA_OK) return rc;
  if (p->V == 0) return PNA_OK;
  const long TF = (long)k.T * k.F;
  if (!table || ld_table < TF * (k.has_var ? 2 : 1)) return pna_set_error(PNA_E_INVALID, "pna_segreduce_bwd_rowprep_f32: table too narrow");
  if (p->edge_term) return pna_set_error(PNA_E_INVALID, "pna_segreduce_bwd_rowprep_f32: the pull formulation has no per-edge term");
  k.table = table; k.ld_table = ld_table;
  const long n = (long)k.V * TF;
  launch_rowprep(k, (hipStream_t)stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}

extern "C" int pna_segreduce_bwd_pull_f32(const pna_segreduce_bwd_pull_args* q, pna_stream_t stream) {
  if (!q || !q->base) return pna_set_error(PNA_E_INVALID, "pna_segreduce_bwd_pull_f32: null args");
  if (int rc_ss = pna_check_struct_size("pna_segreduce_bwd_pull_f32", q->struct_size, sizeof(*q))) return rc_ss;
  const pna_segreduce_bwd_args* p = q->base;
  PArgs k;
  int rc = fill_pull_args(p, k, "pna_segreduce_bwd_pull_f32: bad arguments");
  if (rc != PNA_OK) return rc;
  if (p->V == 0 || q->n_items_t == 0) return PNA_OK;
  const int T = k.T, F = p->F, TF = T * F;
  if (!k.g[PNA_AGG_MAX] || !k.g[PNA_AGG_MIN] || !p->argmax || !p->argmin || !k.has_var || F < 4 || F > 256 || !p->x || !p->grad_x || !q->table ||
      !q->col_t || !q->items_t || p->ldx < TF || p->ld_gx < TF || q->n_items_t < 0 ||
      (!q->edge_rows && (!q->rank_t || !q->ranks || q->ld_table < 2 * TF || q->ld_rank < 2 * TF)))
    return pna_set_error(PNA_E_INVALID, "pna_segreduce_bwd_pull_f32: needs max + min + std/var among aggr[], argmax / argmin, x, grad_x, the rowprep table "
                                        "(ld >= 2 T F), the transposed graph (col_t, rank_t, items_t), a ranks workspace (ld >= 2 T F) and 4 <= F <= 256");
  hipStream_t st = (hipStream_t)stream;
  if (q->edge_rows) {
    // round 6: per-edge rows (see k_bwd_edge_rows): table = the (V, ld_table >= F) rows of R2
    if (T != 1 || p->dst_term || !k.g[PNA_AGG_MEAN] || !k.g[PNA_AGG_STD] || k.g[PNA_AGG_SUM] || k.g[PNA_AGG_VAR] || !p->stdv || !q->pos_t || !q->items ||
        q->n_items < 0 || q->ld_edge < F || q->ld_table < F || p->stat_node_of)
      return pna_set_error(PNA_E_INVALID, "pna_segreduce_bwd_pull_f32: edge_rows needs one tower, no dst_term, aggr[] = mean / std / max / min, pos_t, the forward "
                                          "work list (items), ld_edge >= F and ld_table >= F");
    EArgs e;
    memset(&e, 0, sizeof(e));
    e.items = q->items; e.rowptr = p->rowptr; e.row_of = p->stat_row_of;
    e.gmean = k.g[PNA_AGG_MEAN]; e.gstd = k.g[PNA_AGG_STD]; e.gmax = k.g[PNA_AGG_MAX]; e.gmin = k.g[PNA_AGG_MIN]; e.mean = p->mean; e.stdv = p->stdv;
    e.amx = p->argmax; e.amn = p->argmin; e.r2 = const_cast<float*>(q->table); e.P = q->edge_rows;
    e.ld_g = p->ld_g; e.ld_stat = p->ld_stat; e.ld_arg = p->ld_arg; e.ld_r2 = q->ld_table; e.ld_p = q->ld_edge;
    e.n_items = q->n_items; e.F = F;
    e.L = (F + 3) / 4 > 64 ? 64 : (F + 3) / 4;
    // whole rows (k_bwd_edge_rows): the pitch a multiple of 4 floats, at most 2 chunks of padding, the buffer 16-byte aligned, as many groups per wavefront
    if (PNA_EDGE_ROWS_WHOLE && q->ld_edge % 4 == 0 && q->ld_edge / 4 > e.L && q->ld_edge / 4 <= e.L + 2 && q->ld_edge / 4 <= 64 && ((uintptr_t)q->edge_rows & 15) == 0 &&
        64 / (int)(q->ld_edge / 4) == 64 / e.L)
      e.L = (int)(q->ld_edge / 4);
    e.G = 64 / e.L;
    const long groups = (long)kWaves * e.G;
    if (q->n_items > 0)
      hipLaunchKernelGGL(k_bwd_edge_rows, dim3((unsigned)((q->n_items + groups - 1) / groups)), dim3(kBlock), 0, st, e);
    LArgs2 a;
    memset(&a, 0, sizeof(a));
    a.items = q->items_t; a.col_t = q->col_t; a.pos_t = q->pos_t; a.P = q->edge_rows; a.r2 = q->table; a.x = p->x; a.gx = p->grad_x;
    a.ld_p = q->ld_edge; a.ld_r2 = q->ld_table; a.ldx = p->ldx; a.ld_gx = p->ld_gx;
    a.n_items = q->n_items_t; a.F = F; a.L = e.L; a.G = e.G;
    hipLaunchKernelGGL((k_bwd_pull_rows<4>), dim3((unsigned)((q->n_items_t + groups - 1) / groups)), dim3(kBlock), 0, st, a);
    hipError_t er = hipGetLastError();
    if (er != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(er));
    return PNA_OK;
  }
  const long n = (long)p->V * TF;
  // PACKED rows (round 4): ld_table >= 5 T F and ranks == table + 4 T F floats -- one row [R1 | R2 | G_max | G_min | ranks] per node,
  // 20 T F bytes contiguous: an out-edge touches 12 cache lines at F = 75 (pitch 1536 B) instead of ~14.7 for the three separate
  // pieces (R1 | R2 in the table, G_max | G_min inside the aggregate's gradient row, the rank row).  Needs run_rowprep (the copy).
  const bool packed = q->run_rowprep && q->ld_table >= 5L * TF && (const void*)q->ranks == (const void*)(q->table + 4L * TF);
  if (q->run_rowprep) {              // rowprep (table, grad_dst) and the ranks in ONE pass over the rows
    k.table = const_cast<float*>(q->table); k.ld_table = q->ld_table; k.ranks = q->ranks; k.ld_rank = q->ld_rank;
    // IN PLACE: base->gagg == table with aggr[] = {mean, std, max, min} -- the caller's d agg contraction wrote its output straight
    // into the packed rows ([G_mean | G_std | G_max | G_min] at [0, 4 T F)): rowprep overwrites the first two blocks with R1 | R2,
    // nothing is copied (one-tower layers; every thread reads its own columns before it writes them)
    const bool in_place = packed && p->gagg == q->table;
    if (in_place && !(T == 1 && p->n_aggr == 4 && p->aggr[0] == PNA_AGG_MEAN && p->aggr[1] == PNA_AGG_STD && p->aggr[2] == PNA_AGG_MAX &&
                      p->aggr[3] == PNA_AGG_MIN && p->ld_g == q->ld_table))
      return pna_set_error(PNA_E_INVALID, "pna_segreduce_bwd_pull_f32: gagg == table (in place) needs one tower, aggr[] = {mean, std, max, min} and ld_g == ld_table");
    k.gcopy = (packed && !in_place) ? const_cast<float*>(q->table) + 2L * TF : nullptr;
    launch_rowprep(k, st, in_place);
  } else {
    hipLaunchKernelGGL(k_bwd_ranks, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, p->rowptr, p->argmax, p->argmin, (long)p->ld_arg,
                       (long)k.ts_in, p->V, F, T, q->ranks, (long)q->ld_rank);
  }
  LArgs a;
  memset(&a, 0, sizeof(a));
  a.items = q->items_t; a.col_t = q->col_t; a.rank_t = q->rank_t; a.table = q->table; a.gmax = k.g[PNA_AGG_MAX]; a.gmin = k.g[PNA_AGG_MIN];
  a.ranks = q->ranks; a.x = p->x; a.gx = p->grad_x;
  a.ld_table = q->ld_table; a.ld_g = p->ld_g; a.ld_rank = q->ld_rank; a.ldx = p->ldx; a.ld_gx = p->ld_gx; a.ts_g = k.ts_g;
  if (packed) { a.gmax = q->table + 2L * TF; a.gmin = q->table + 3L * TF; a.ld_g = q->ld_table; a.ts_g = F; }
  a.n_items = q->n_items_t; a.F = F; a.T = T; a.TF = TF;
  a.L = (F + 3) / 4 > 64 ? 64 : (F + 3) / 4; a.G = 64 / a.L;
  const long groups = (long)kWaves * a.G;
  dim3 grid((unsigned)((q->n_items_t + groups - 1) / groups), (unsigned)T);
  hipLaunchKernelGGL((k_bwd_pull<4>), grid, dim3(kBlock), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}

extern "C" int pna_segreduce_bwd_argscatter_f32(const pna_segreduce_bwd_args* p, pna_stream_t stream) {
  PArgs k;
  int rc = fill_pull_args(p, k, "pna_segreduce_bwd_argscatter_f32: bad arguments");
  if (rc != PNA_OK) return rc;
  if (p->stat_row_of || p->stat_node_of) return pna_set_error(PNA_E_INVALID, "pna_segreduce_bwd_argscatter_f32: stat_row_of / stat_node_of are honoured by the rowprep / pull entry points only");
  if (p->V == 0 || (!k.g[PNA_AGG_MAX] && !k.g[PNA_AGG_MIN])) return PNA_OK;
  if (!p->col || !p->grad_x || (k.g[PNA_AGG_MAX] && !p->argmax) || (k.g[PNA_AGG_MIN] && !p->argmin))
    return pna_set_error(PNA_E_INVALID, "pna_segreduce_bwd_argscatter_f32: col / grad_x / argmax / argmin missing");
  const long n = (long)k.V * k.T * k.F;
  hipLaunchKernelGGL(k_bwd_argscatter, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, k);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}
