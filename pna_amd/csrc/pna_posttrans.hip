// pna_posttrans.hip -- post-aggregation tower contraction on the gfx950 fp32 matrix cores.
// Implements pna_posttrans_f32 of include/pna_amd.h (replaces the `posttrans` nn.Linear of
// models/dgl/pna_layer.py:65-68,:206 and models/pytorch/pna/layer.py:47-48).
//
//   y[v] = bias + Wh . h[v] + sum_s scale_s[v] * (W_s . a[v])
//
// The reference materialises the scaled aggregate [id*a | amp*a | att*a] (V x A*S*F floats) and
// multiplies it by one weight; here the S per-row scalers are pulled out of the contraction, so the
// operand read from HBM is only the (V x A*F) identity aggregate and each scaler block gets its own
// accumulator tile that is combined in the epilogue.
//
// Tiling (v_mfma_f32_16x16x4_f32, exact fp32 fma chain, 32-cycle issue):
//   * a workgroup = 4 wavefronts = 64 output rows; each wavefront owns 16 rows x (NT x 16) columns
//     x S scalers  => S*NT independent accumulators, every A fragment feeds S*NT MFMAs;
//   * K is consumed 16 at a time: lane (i = l&15, g = l>>4) loads A[row0+i][k0+4g .. k0+4g+3] with one
//     dwordx4; k-step t of the chunk lets lane group g multiply physical k = k0 + 4g + t (a
//     permutation of the summation order that A and B agree on), so no transpose is needed;
//   * the matching W rows (16 x S x N floats) are staged once per workgroup in LDS (double
//     buffered) and read back as B fragments with conflict-free ds_read_b32 (row pitch = 4 mod 8).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "pna_amd.h"
#include "pna_internal.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) f4u { f4 v; };

constexpr int kBlock = 256;
constexpr int kKC = 16;          // k values per chunk
constexpr int kMaxNT = 5;        // 16-column tiles per workgroup (80 columns)
constexpr int kNP = kMaxNT * 16 + 4;   // LDS row pitch in floats, = 4 (mod 8): conflict-free B reads

struct GArgs {
  const float* a; const float* w; const float* h; const float* wh; const float* bias;
  const float* row_scale[PNA_MAX_SCALER];
  float* y;
  long lda, ldw, ldh, ldy;
  int M, K, N, Kh;
};

// Stage rows [k0, k0+16) x columns [n0, n0 + NT*16) of one K-major weight panel into LDS.
// Out-of-range rows / columns are written as zeros, so the MFMA loop needs no masks.
__device__ __forceinline__ void stage_w(float* dst, const float* w, long ldw, int k0, int kmax, int n0, int nmax) {
  // 16 rows x 80 columns = 1280 floats; 256 threads x 5 floats
  for (int idx = threadIdx.x; idx < kKC * kMaxNT * 16; idx += kBlock) {
    const int kk = idx / (kMaxNT * 16);
    const int n = idx - kk * (kMaxNT * 16);
    const int k = k0 + kk, col = n0 + n;
    float v = 0.f;
    if (k < kmax && col < nmax) v = w[(size_t)k * ldw + col];
    dst[kk * kNP + n] = v;
  }
}

template <int S, bool HAS_H>
__global__ __launch_bounds__(kBlock) void k_posttrans(const GArgs g) {
  // LDS: 2 buffers x (S [+1]) panels x 16 x kNP floats
  constexpr int P = S + (HAS_H ? 1 : 0);
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int row0 = blockIdx.x * 64 + wave * 16;
  const int n0 = blockIdx.y * (kMaxNT * 16);
  const int nt_act = min(kMaxNT, (g.N - n0 + 15) / 16);
  const int arow = min(row0 + li, g.M - 1);

  f4 acc[P][kMaxNT];
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int n = 0; n < kMaxNT; ++n) acc[p][n] = (f4){0.f, 0.f, 0.f, 0.f};

  // The K loop runs over the aggregate operand `a` (chunks [0, nca)) and then, for the tower
  // variant, over the node's own features `h` (chunks [nca, nca + nch)), which only feed panel S.
  const int nca = (g.K + kKC - 1) / kKC;
  const int nch = HAS_H ? (g.Kh + kKC - 1) / kKC : 0;
  const int nc = nca + nch;
  const int panel = kKC * kNP;

  auto stage = [&](int c, int buf) {
    float* base = lds + buf * P * panel;
    if (c < nca) {
#pragma unroll
      for (int s = 0; s < S; ++s) stage_w(base + s * panel, g.w + (size_t)s * g.K * g.ldw, g.ldw, c * kKC, g.K, n0, g.N);
    } else if (HAS_H) {
      stage_w(base + S * panel, g.wh, g.ldw, (c - nca) * kKC, g.Kh, n0, g.N);
    }
  };
  auto load_a = [&](int c) -> f4 {
    const float* src; long ld; int kmax, k;
    if (c < nca) { src = g.a; ld = g.lda; kmax = g.K; k = c * kKC + 4 * lg; }
    else { src = g.h; ld = g.ldh; kmax = g.Kh; k = (c - nca) * kKC + 4 * lg; }
    f4 v = (f4){0.f, 0.f, 0.f, 0.f};
    const float* p = src + (size_t)arow * ld + k;
    if (k + 3 < kmax) v = reinterpret_cast<const f4u*>(p)->v;
    else {
      if (k < kmax) v.x = p[0];
      if (k + 1 < kmax) v.y = p[1];
      if (k + 2 < kmax) v.z = p[2];
    }
    return v;
  };

  stage(0, 0);
  f4 a_cur = load_a(0);
  __syncthreads();
  for (int c = 0; c < nc; ++c) {
    const int buf = c & 1;
    f4 a_nxt = (f4){0.f, 0.f, 0.f, 0.f};
    if (c + 1 < nc) { stage(c + 1, buf ^ 1); a_nxt = load_a(c + 1); }
    const float* base = lds + buf * P * panel;
    const bool is_h = c >= nca;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float av = t == 0 ? a_cur.x : t == 1 ? a_cur.y : t == 2 ? a_cur.z : a_cur.w;
      const float* brow = base + (4 * lg + t) * kNP + li;
      if (!is_h) {
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
          for (int n = 0; n < kMaxNT; ++n)
            if (n < nt_act)
              acc[s][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, brow[s * panel + n * 16], acc[s][n], 0, 0, 0);
      } else if (HAS_H) {
#pragma unroll
        for (int n = 0; n < kMaxNT; ++n)
          if (n < nt_act)
            acc[P - 1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, brow[S * panel + n * 16], acc[P - 1][n], 0, 0, 0);
      }
    }
    a_cur = a_nxt;
    __syncthreads();
  }

  // epilogue: C/D layout of 16x16 tiles: col = lane & 15, row = (lane >> 4) * 4 + reg
  float sc[S][4];
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = min(row0 + lg * 4 + r, g.M - 1);
      sc[s][r] = g.row_scale[s] ? g.row_scale[s][row] : 1.f;
    }
#pragma unroll
  for (int n = 0; n < kMaxNT; ++n) {
    if (n >= nt_act) continue;
    const int col = n0 + n * 16 + li;
    if (col >= g.N) continue;
    const float b = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + lg * 4 + r;
      if (row >= g.M) continue;
      float v = b;
      if (HAS_H) v = v + acc[P - 1][n][r];
#pragma unroll
      for (int s = 0; s < S; ++s) v = v + sc[s][r] * acc[s][n][r];
      g.y[(size_t)row * g.ldy + col] = v;
    }
  }
}

template <int S>
void launch_s(const GArgs& g, bool has_h, dim3 grid, hipStream_t st) {
  const size_t lds = (size_t)2 * (S + (has_h ? 1 : 0)) * kKC * kNP * sizeof(float);
  if (has_h) hipLaunchKernelGGL((k_posttrans<S, true>), grid, dim3(kBlock), lds, st, g);
  else hipLaunchKernelGGL((k_posttrans<S, false>), grid, dim3(kBlock), lds, st, g);
}

}  // namespace

extern "C" int pna_posttrans_f32(const pna_posttrans_args* p, pna_stream_t stream) {
  if (!p) return pna_set_error(PNA_E_INVALID, "pna_posttrans_f32: null args");
  if (p->M < 0 || p->K <= 0 || p->N <= 0) return pna_set_error(PNA_E_INVALID, "pna_posttrans_f32: bad M/K/N");
  if (p->M == 0) return PNA_OK;
  if (!p->a || !p->w || !p->y) return pna_set_error(PNA_E_INVALID, "pna_posttrans_f32: a/w/y must be non-null");
  if (p->n_scaler < 1 || p->n_scaler > 5) return pna_set_error(PNA_E_INVALID, "pna_posttrans_f32: n_scaler must be 1..5");
  if (p->lda < p->K || p->ldw < p->N || p->ldy < p->N) return pna_set_error(PNA_E_INVALID, "pna_posttrans_f32: leading dimensions too small");
  const bool has_h = p->h != nullptr && p->Kh > 0;
  if (has_h && (!p->wh || p->ldh < p->Kh)) return pna_set_error(PNA_E_INVALID, "pna_posttrans_f32: h given without wh / ldh too small");
  GArgs g;
  memset(&g, 0, sizeof(g));
  g.a = p->a; g.w = p->w; g.h = has_h ? p->h : nullptr; g.wh = has_h ? p->wh : nullptr; g.bias = p->bias;
  for (int s = 0; s < p->n_scaler; ++s) g.row_scale[s] = p->row_scale[s];
  g.y = p->y; g.lda = p->lda; g.ldw = p->ldw; g.ldh = p->ldh; g.ldy = p->ldy;
  g.M = p->M; g.K = p->K; g.N = p->N; g.Kh = has_h ? p->Kh : 0;
  dim3 grid((unsigned)((p->M + 63) / 64), (unsigned)((p->N + kMaxNT * 16 - 1) / (kMaxNT * 16)));
  hipStream_t st = (hipStream_t)stream;
  switch (p->n_scaler) {
    case 1: launch_s<1>(g, has_h, grid, st); break;
    case 2: launch_s<2>(g, has_h, grid, st); break;
    case 3: launch_s<3>(g, has_h, grid, st); break;
    case 4: launch_s<4>(g, has_h, grid, st); break;
    default: launch_s<5>(g, has_h, grid, st); break;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}
