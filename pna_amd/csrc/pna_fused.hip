// pna_fused.hip -- the whole PNASimpleLayer forward (inference) in ONE kernel: gather + mean|max|min|std
// segment-reduce of a 32-row tile into LDS, then the posttrans contraction of that tile on the fp32 matrix
// cores straight out of LDS, then degree scalers / BatchNorm / ReLU / residual in the epilogue.
// Implements pna_fused_simple_f32 of include/pna_amd.h (models/dgl/pna_layer.py:197-216 in one launch).
//
// Why: run as two kernels, the (V, 4F) aggregate is written to HBM by the gather kernel (1.2 GB on the roofline
// workload) and read back by the contraction (another 1.2 GB) -- a third of the layer's HBM traffic.  Here the
// aggregate of a tile lives only in LDS.
//
// Workgroup = 8 wavefronts = 32 destination rows; LDS = A tile 32 x (4*B4 + 4) floats (B4 = F rounded up to 4;
// aggregator blocks at stride B4, pad columns zero) + double-buffered weight image (same packed format as
// pna_posttrans_f32).  ~72 KB per workgroup -> two workgroups per CU, whose phases interleave: one gathers
// (HBM/latency bound, matrix pipe idle) while the other contracts (matrix pipe busy, memory idle).
//   phase 1  lane groups (L = B4/4 lanes, G = 64/L per wavefront, 8G per workgroup) walk one row each, 4 gathers
//            in flight per group, fold in registers, finalise, write mean|max|min|std to the LDS tile;
//            hub rows (degree > heavy_threshold) are then walked by ALL lane groups of the workgroup and their
//            partial (sum, sumsq, max, min) combined in lane-group order through the LDS row (deterministic);
//   phase 2  v_mfma_f32_16x16x4_f32: wavefront w owns row tile (w & 1) and the accumulators c = (w>>1) + 4j of the
//            S*NT (scaler, column tile) combinations; A fragments by ds_read_b128 from the tile, B fragments from
//            the staged weight image; K in chunks of 16 with one barrier per chunk;
//   phase 3  accumulators -> LDS, every thread finishes output elements:
//            y = residual + relu((bias + sum_s scale_s[row] * acc_s) * bn_scale + bn_shift).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "pna_amd.h"
#include "pna_internal.h"
#include "pna_rowstats.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) f4u { f4 v; };

constexpr int kThreads = 512;
constexpr int kWaves = 8;
constexpr int kRows = 32;                    // destination rows per workgroup
constexpr int kKC = 16;
constexpr int kMaxNT = 5;
constexpr int kNW = kMaxNT * 16;
constexpr int kNP = kNW + 4;
constexpr int kPanel = kKC * kNP;            // must match pna_posttrans.hip (packed image format)

struct UArgs {
  const int32_t* rowptr; const int32_t* col; const float* x; const float* w_img; const float* bias;
  const float* row_scale[3]; const float* col_scale; const float* col_shift; const float* residual;
  float* y;
  long ldx, ld_res, ldy;
  int V, F, N, heavy_threshold, relu, a_floats;
};

using pna_dev::vmax;
using pna_dev::vmin;

struct Acc4 {
  f4 s, q, mx, mn;
  __device__ __forceinline__ void init() {
    s = (f4){0.f, 0.f, 0.f, 0.f}; q = s;
    mx = (f4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    mn = (f4){INFINITY, INFINITY, INFINITY, INFINITY};
  }
  __device__ __forceinline__ void fold(const f4 v, bool on) {       // off slots hold a copy of a folded edge
    const f4 z = (f4){0.f, 0.f, 0.f, 0.f};
    const f4 vm = on ? v : z;
    s = s + vm;
    q = q + vm * vm;
    mx.x = vmax(mx.x, v.x); mx.y = vmax(mx.y, v.y); mx.z = vmax(mx.z, v.z); mx.w = vmax(mx.w, v.w);
    mn.x = vmin(mn.x, v.x); mn.y = vmin(mn.y, v.y); mn.z = vmin(mn.z, v.z); mn.w = vmin(mn.w, v.w);
  }
};

template <int S>
__global__ __launch_bounds__(kThreads, 4) void k_fused_simple(const UArgs a) {
  extern __shared__ float lds[];
  const int F = a.F;
  const int B4 = (F + 3) & ~3;               // aggregator block stride in the tile
  const int Kp = 4 * B4;                     // contraction length (multiple of 16)
  const int AP = Kp + 4;                     // tile row pitch (floats): 16-byte aligned rows, A reads conflict-free
  float* const At = lds;                     // [kRows][AP]
  float* const Wt = lds + a.a_floats;        // [2][S][kPanel]; a_floats = max(tile, phase-3 accumulator dump)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int tile0 = blockIdx.x * kRows;
  const int NT = (a.N + 15) >> 4;

  // ---- weight staging (issued early: chunk 0 lands while the tile is being gathered) ---------------------
  const int nca = Kp / kKC;
  const f4* img = reinterpret_cast<const f4*>(a.w_img);
  constexpr int SV = (S * kPanel / 4 + kThreads - 1) / kThreads;
  f4 wreg[SV];
  auto stage_load = [&](int c) {
    const f4* src = img + (size_t)c * (S * kPanel / 4);
#pragma unroll
    for (int i = 0; i < SV; ++i) wreg[i] = src[min(tid + i * kThreads, S * kPanel / 4 - 1)];
  };
  auto stage_write = [&](int buf) {
    f4* dst = reinterpret_cast<f4*>(Wt + buf * S * kPanel);
#pragma unroll
    for (int i = 0; i < SV; ++i) {
      const int idx = tid + i * kThreads;
      if (idx < S * kPanel / 4) dst[idx] = wreg[i];
    }
  };
  stage_load(0);
  __shared__ unsigned hub_mask;              // bit rl = tile row rl is a hub row (filled in phase 1)
  if (tid == 0) hub_mask = 0u;
  __syncthreads();

  // ---- phase 1: aggregate the tile's rows into LDS ---------------------------------------------------------
  const int L = B4 >> 2;                     // lanes per row
  const int G = 64 / L;
  const int NG = kWaves * G;
  const int grp = lane / L;
  const bool in_grp = grp < G;
  const int c = lane - grp * L;
  const int grp_lane0 = grp * L;
  const int gid = wave * G + grp;
  const int off = min(4 * c, F - 4);         // sliding last window (F >= 4)
  const bool tail = 4 * c + 4 > F;           // this lane's window was slid back / has pad columns after it
  const int HT = a.heavy_threshold;

  auto walk = [&](Acc4& acc, int beg, int end) {
    for (int cb = beg; cb < end; cb += L) {
      const int nidx = min(L, end - cb);
      const int idx = a.col[cb + min(c, nidx - 1)];
      for (int j = 0; j < nidx; j += 4) {
        const int nv = min(4, nidx - j);
        int id[4];
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) id[u] = __shfl(idx, grp_lane0 + j + min(u, nv - 1));
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const f4u*>(a.x + (size_t)id[u] * a.ldx + off)->v;
#pragma unroll
        for (int u = 0; u < 4; ++u) acc.fold(v[u], u < nv);
      }
    }
  };
  auto put4 = [&](float* p, const f4 v) {    // 16-byte aligned unless this is the slid-back window
    if (!tail || (off & 3) == 0) *reinterpret_cast<f4*>(p) = v;
    else { p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w; }
  };
  auto write_blocks = [&](int rl, const f4 b0, const f4 b1, const f4 b2, const f4 b3) {
    float* r = At + rl * AP;
    put4(r + off, b0); put4(r + B4 + off, b1); put4(r + 2 * B4 + off, b2); put4(r + 3 * B4 + off, b3);
    if (tail)
      for (int k = F; k < B4; ++k) { r[k] = 0.f; r[B4 + k] = 0.f; r[2 * B4 + k] = 0.f; r[3 * B4 + k] = 0.f; }   // pad columns
  };
  auto finalize = [&](int rl, const Acc4& acc, int deg) {   // mean | max | min | std of one row -> tile
    const f4 z = (f4){0.f, 0.f, 0.f, 0.f};
    if (deg <= 0) { write_blocks(rl, z, z, z, z); return; }
    // s / D and q / D correctly rounded from ONE division per row (Markstein's fma correction, see pna_segreduce.hip div_rn)
    const float Dg = (float)deg, invD = 1.0f / Dg;
    auto div4 = [&](const f4 a) -> f4 {
      const f4 q0 = a * invD;
      f4 q;
      q.x = __builtin_fmaf(__builtin_fmaf(-Dg, q0.x, a.x), invD, q0.x); q.y = __builtin_fmaf(__builtin_fmaf(-Dg, q0.y, a.y), invD, q0.y);
      q.z = __builtin_fmaf(__builtin_fmaf(-Dg, q0.z, a.z), invD, q0.z); q.w = __builtin_fmaf(__builtin_fmaf(-Dg, q0.w, a.w), invD, q0.w);
      q.x = (q.x == q.x && __builtin_fabsf(q.x) != INFINITY) ? q.x : q0.x; q.y = (q.y == q.y && __builtin_fabsf(q.y) != INFINITY) ? q.y : q0.y;
      q.z = (q.z == q.z && __builtin_fabsf(q.z) != INFINITY) ? q.z : q0.z; q.w = (q.w == q.w && __builtin_fabsf(q.w) != INFINITY) ? q.w : q0.w;
      return q;
    };
    const f4 mean = div4(acc.s);
    f4 var = div4(acc.q) - mean * mean;
    var.x = var.x < 0.f ? 0.f : var.x; var.y = var.y < 0.f ? 0.f : var.y;
    var.z = var.z < 0.f ? 0.f : var.z; var.w = var.w < 0.f ? 0.f : var.w;
    f4 mx, mn, sd;
    mx.x = acc.q.x != acc.q.x ? acc.q.x : acc.mx.x; mx.y = acc.q.y != acc.q.y ? acc.q.y : acc.mx.y;
    mx.z = acc.q.z != acc.q.z ? acc.q.z : acc.mx.z; mx.w = acc.q.w != acc.q.w ? acc.q.w : acc.mx.w;
    mn.x = acc.q.x != acc.q.x ? acc.q.x : acc.mn.x; mn.y = acc.q.y != acc.q.y ? acc.q.y : acc.mn.y;
    mn.z = acc.q.z != acc.q.z ? acc.q.z : acc.mn.z; mn.w = acc.q.w != acc.q.w ? acc.q.w : acc.mn.w;
    sd.x = sqrtf(var.x + 1e-5f); sd.y = sqrtf(var.y + 1e-5f); sd.z = sqrtf(var.z + 1e-5f); sd.w = sqrtf(var.w + 1e-5f);
    write_blocks(rl, mean, mx, mn, sd);
  };

  if (in_grp) {
    for (int rl = gid; rl < kRows; rl += NG) {
      const int row = tile0 + rl;
      Acc4 acc;
      acc.init();
      int deg = 0;
      if (row < a.V) {
        const int beg = a.rowptr[row], end = a.rowptr[row + 1];
        deg = end - beg;
        if (HT > 0 && deg > HT) { deg = 0; if (c == 0) atomicOr(&hub_mask, 1u << rl); }   // zeros for now, done cooperatively below
        else walk(acc, beg, end);
      }
      finalize(rl, acc, deg);
    }
  }
  __syncthreads();
  // hub rows of this tile: every lane group takes a contiguous slice of the row's edges; partials are combined in
  // lane-group order through the row's LDS slots (raw s | q | max | min), then group 0 finalises in place.
  static_assert(kRows <= 32, "hub_mask is one word");
  {
    for (unsigned m = hub_mask; m; m &= m - 1) {                     // workgroup-uniform
      const int rl = __ffs(m) - 1;
      const int row = tile0 + rl;
      const int beg = a.rowptr[row], end = a.rowptr[row + 1];
      const int deg = end - beg;
      const int per = ((deg + NG - 1) / NG + 3) & ~3;
      Acc4 acc;
      acc.init();
      if (in_grp) {
        const int b = min(beg + gid * per, end), e = min(b + per, end);
        walk(acc, b, e);
      }
      float* r = At + rl * AP;
      for (int g2 = 0; g2 < NG; ++g2) {
        if (in_grp && gid == g2) {
          f4 os, oq, ox, on;
          if (g2 > 0) {
            os = (f4){r[off], r[off + 1], r[off + 2], r[off + 3]};
            oq = (f4){r[B4 + off], r[B4 + off + 1], r[B4 + off + 2], r[B4 + off + 3]};
            ox = (f4){r[2 * B4 + off], r[2 * B4 + off + 1], r[2 * B4 + off + 2], r[2 * B4 + off + 3]};
            on = (f4){r[3 * B4 + off], r[3 * B4 + off + 1], r[3 * B4 + off + 2], r[3 * B4 + off + 3]};
            acc.s = os + acc.s; acc.q = oq + acc.q;
            acc.mx.x = vmax(ox.x, acc.mx.x); acc.mx.y = vmax(ox.y, acc.mx.y); acc.mx.z = vmax(ox.z, acc.mx.z); acc.mx.w = vmax(ox.w, acc.mx.w);
            acc.mn.x = vmin(on.x, acc.mn.x); acc.mn.y = vmin(on.y, acc.mn.y); acc.mn.z = vmin(on.z, acc.mn.z); acc.mn.w = vmin(on.w, acc.mn.w);
          }
          if (g2 == NG - 1) {
            finalize(rl, acc, deg);
          } else {
            // overlapped lanes (slid window) hold identical values for the shared features: benign double write
            for (int k = 0; k < 4; ++k) {
              r[off + k] = acc.s[k]; r[B4 + off + k] = acc.q[k]; r[2 * B4 + off + k] = acc.mx[k]; r[3 * B4 + off + k] = acc.mn[k];
            }
          }
        }
        __syncthreads();
      }
    }
  }

  // ---- phase 2: contraction of the tile on the matrix cores --------------------------------------------------
  const int li = lane & 15, lg = lane >> 4;
  const int rt = wave % (kRows / 16), q = wave / (kRows / 16);
  const int ncombo = S * NT;
  constexpr int QW = kWaves / (kRows / 16);  // wavefronts sharing a row tile
  constexpr int JN = (S * kMaxNT + QW - 1) / QW;
  f4 acc2[JN];
  int boff[JN];
#pragma unroll
  for (int j = 0; j < JN; ++j) {
    acc2[j] = (f4){0.f, 0.f, 0.f, 0.f};
    const int cmb = q + QW * j < ncombo ? q + QW * j : 0;       // dead slots recompute combination 0 (never stored)
    const int s = cmb / NT;
    boff[j] = s * kPanel + (cmb - s * NT) * 16;
  }
  const float* arow = At + (rt * 16 + li) * AP + 4 * lg;
  stage_write(0);
  __syncthreads();
  for (int ch = 0; ch < nca; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nca) stage_load(ch + 1);
    const float* base = Wt + buf * S * kPanel;
    const f4 av4 = *reinterpret_cast<const f4*>(arow + ch * kKC);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float av = t == 0 ? av4.x : t == 1 ? av4.y : t == 2 ? av4.z : av4.w;
      const float* brow = base + (4 * lg + t) * kNP + li;
#pragma unroll
      for (int j = 0; j < JN; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, brow[boff[j]], acc2[j], 0, 0, 0);
    }
    if (ch + 1 < nca) stage_write(buf ^ 1);
    __syncthreads();
  }

  // ---- phase 3: accumulators -> LDS (the A tile is dead now), epilogue ---------------------------------------
  float* Ct = At;                             // [ncombo][kRows/16][16][16]
#pragma unroll
  for (int j = 0; j < JN; ++j) {
    const int cmb = q + QW * j;
    if (cmb < ncombo) {
      float* t = Ct + ((cmb * (kRows / 16) + rt) * 16) * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) t[(lg * 4 + r) * 16 + li] = acc2[j][r];
    }
  }
  __syncthreads();
  const int N = a.N;
  for (int idx = tid; idx < kRows * N; idx += kThreads) {
    const int rl = idx / N, n = idx - rl * N;
    const int row = tile0 + rl;
    if (row >= a.V) continue;
    const int nt = n >> 4, cl = n & 15, rt2 = rl >> 4, r16 = rl & 15;
    float v = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float sc = a.row_scale[s] ? a.row_scale[s][row] : 1.f;
      v = v + sc * Ct[(((s * NT + nt) * (kRows / 16) + rt2) * 16 + r16) * 16 + cl];
    }
    if (a.col_scale) v = v * a.col_scale[n] + a.col_shift[n];
    if (a.relu) v = v > 0.f ? v : (v != v ? v : 0.f);
    if (a.residual) v = a.residual[(size_t)row * a.ld_res + n] + v;
    a.y[(size_t)row * a.ldy + n] = v;
  }
}

}  // namespace

extern "C" int pna_fused_simple_f32(const pna_fused_simple_args* p, pna_stream_t stream) {
  if (!p) return pna_set_error(PNA_E_INVALID, "pna_fused_simple_f32: null args");
  if (int rc_ss = pna_check_struct_size("pna_fused_simple_f32", p->struct_size, sizeof(*p))) return rc_ss;
  if (p->V < 0 || p->F < 4 || p->F > 80 || p->N < 1 || p->N > kNW || p->n_scaler < 1 || p->n_scaler > 3)
    return pna_set_error(PNA_E_INVALID, "pna_fused_simple_f32: supported range is 4 <= F <= 80, N <= 80, 1..3 scalers");
  if (p->V == 0) return PNA_OK;
  if (!p->rowptr || !p->col || !p->x || !p->w_img || !p->y) return pna_set_error(PNA_E_INVALID, "pna_fused_simple_f32: null pointer");
  if (p->ldx < p->F || p->ldy < p->N || (p->residual && p->ld_res < p->N))
    return pna_set_error(PNA_E_INVALID, "pna_fused_simple_f32: leading dimensions too small");
  if ((p->col_scale == nullptr) != (p->col_shift == nullptr))
    return pna_set_error(PNA_E_INVALID, "pna_fused_simple_f32: col_scale and col_shift come together");
  UArgs u;
  memset(&u, 0, sizeof(u));
  u.rowptr = p->rowptr; u.col = p->col; u.x = p->x; u.w_img = p->w_img; u.bias = p->bias;
  for (int s = 0; s < p->n_scaler; ++s) u.row_scale[s] = p->row_scale[s];
  u.col_scale = p->col_scale; u.col_shift = p->col_shift; u.residual = p->residual; u.y = p->y;
  u.ldx = p->ldx; u.ld_res = p->ld_res; u.ldy = p->ldy;
  u.V = p->V; u.F = p->F; u.N = p->N; u.heavy_threshold = p->heavy_threshold; u.relu = p->relu;
  const int B4 = (p->F + 3) & ~3, AP = 4 * B4 + 4, NT = (p->N + 15) / 16;
  u.a_floats = kRows * AP > p->n_scaler * NT * kRows * 16 ? kRows * AP : p->n_scaler * NT * kRows * 16;
  const size_t lds = ((size_t)u.a_floats + (size_t)2 * p->n_scaler * kPanel) * sizeof(float);
  const dim3 grid((unsigned)((p->V + kRows - 1) / kRows));
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipSuccess;
  switch (p->n_scaler) {
    case 1:
      e = hipFuncSetAttribute((const void*)k_fused_simple<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e == hipSuccess) hipLaunchKernelGGL((k_fused_simple<1>), grid, dim3(kThreads), lds, st, u);
      break;
    case 2:
      e = hipFuncSetAttribute((const void*)k_fused_simple<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e == hipSuccess) hipLaunchKernelGGL((k_fused_simple<2>), grid, dim3(kThreads), lds, st, u);
      break;
    default:
      e = hipFuncSetAttribute((const void*)k_fused_simple<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e == hipSuccess) hipLaunchKernelGGL((k_fused_simple<3>), grid, dim3(kThreads), lds, st, u);
      break;
  }
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}
