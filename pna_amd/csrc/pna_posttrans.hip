// pna_posttrans.hip -- post-aggregation tower contraction on the gfx950 fp32 matrix cores.
// Implements pna_posttrans_f32 / pna_posttrans_pack_f32 of include/pna_amd.h (replaces the `posttrans`
// nn.Linear of models/dgl/pna_layer.py:65-68,:206 and models/pytorch/pna/layer.py:47-48, and in eval
// mode the graph-norm / BatchNorm / ReLU / residual tail of :71-75 and :209-213).
//
//   y[v] = epilogue( bias + Wh . h[v] + sum_s scale_s[v] * (W_s . a[v]) )
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
//   * the weight is pre-packed (pna_posttrans_pack_f32, once per weight update) into the exact LDS
//     image of every K-chunk: [column tile][chunk][panel][16 k][84 floats], zero padded, row pitch
//     84 = 4 (mod 8) so that the B-fragment ds_read_b32 of the four lane groups hit disjoint banks;
//   * software pipeline per chunk (async-STAGE split): issue the dwordx4 loads of chunk c+1's image
//     and A fragment -> run chunk c's MFMAs from LDS -> write the staged registers to the other LDS
//     buffer -> one barrier.  HBM/L2 latency hides under the 60 MFMAs (1920 cycles) of a chunk.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "pna_amd.h"
#include "pna_internal.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) f4u { f4 v; };

constexpr int kBlock = 256;
constexpr int kKC = 16;                       // k values per chunk
constexpr int kMaxNT = 5;                     // 16-column tiles per workgroup (80 columns)
constexpr int kNW = kMaxNT * 16;              // 80 output columns per workgroup
constexpr int kNP = kNW + 4;                  // LDS row pitch in floats, = 4 (mod 8)
constexpr int kPanel = kKC * kNP;             // floats per (chunk, panel) image = 1344 (multiple of 4)

struct GArgs {
  const float* a; const float* w_img; const float* h; const float* wh_img; const float* bias;
  const float* row_scale[PNA_MAX_SCALER];
  const float* row_post; const float* col_scale; const float* col_shift; const float* residual;
  float* y;
  long lda, ldh, ldy, ld_res;
  long ts_a, ts_h, ts_w, ts_wh, ts_y;          // tower strides (blockIdx.z = tower), floats
  int M, K, N, Kh, relu;
  float slope;
};

// ---- weight packing ------------------------------------------------------------------------------
// w_ref: the reference nn.Linear weight (N, ldw) with input columns [h (Kh) | scaler 0 (K) | scaler 1 (K) ...].
// w_img[ny][c][s][kk][n]  = w_ref[ny*80 + n][Kh + s*K + c*16 + kk]   (0 outside K / N)
// wh_img[ny][c][kk][n]    = w_ref[ny*80 + n][c*16 + kk]              (0 outside Kh / N)
__global__ void k_pack(const float* w_ref, long ldw, int N, int K, int S, int Kh, float* w_img, float* wh_img) {
  const int nca = (K + kKC - 1) / kKC, nch = (Kh + kKC - 1) / kKC, nty = (N + kNW - 1) / kNW;
  const long total_w = (long)nty * nca * S * kPanel, total_h = (long)nty * nch * kPanel;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total_w + total_h; i += (long)gridDim.x * blockDim.x) {
    if (i < total_w) {
      long r = i;
      const int n = r % kNP; r /= kNP;
      const int kk = r % kKC; r /= kKC;
      const int s = r % S; r /= S;
      const int c = r % nca; r /= nca;
      const int ny = (int)r;
      const int col = ny * kNW + n, k = c * kKC + kk;
      w_img[i] = (n < kNW && col < N && k < K) ? w_ref[(long)col * ldw + Kh + (long)s * K + k] : 0.f;
    } else {
      long r = i - total_w;
      const int n = r % kNP; r /= kNP;
      const int kk = r % kKC; r /= kKC;
      const int c = r % nch; r /= nch;
      const int ny = (int)r;
      const int col = ny * kNW + n, k = c * kKC + kk;
      wh_img[i - total_w] = (n < kNW && col < N && k < Kh) ? w_ref[(long)col * ldw + k] : 0.f;
    }
  }
}

// NT = number of live 16-column tiles (compile time: a run-time guard around every MFMA costs a scalar branch
// per MFMA and breaks the ds_read / MFMA interleave).
template <int S, bool HAS_H, int NT>
__global__ __launch_bounds__(kBlock, (S + (HAS_H ? 1 : 0)) * NT <= 15 ? 4 : 2) void k_posttrans(const GArgs g0) {
  // blockIdx.z = tower: the same contraction on the tower's slices (models/dgl/pna_layer.py:133-139 in one launch)
  GArgs g = g0;
  {
    const long tw = blockIdx.z;
    g.a += tw * g0.ts_a; g.w_img += tw * g0.ts_w; g.y += tw * g0.ts_y;
    if (g0.h) g.h += tw * g0.ts_h;
    if (g0.wh_img) g.wh_img += tw * g0.ts_wh;
    if (g0.bias) g.bias += tw * g0.N;
    if (g0.col_scale) { g.col_scale += tw * g0.N; g.col_shift += tw * g0.N; }
  }
  constexpr int P = S + (HAS_H ? 1 : 0);       // panels resident in LDS per buffer
  constexpr int SV = (S * kPanel / 4 + kBlock - 1) / kBlock;     // dwordx4 per thread to stage S panels
  extern __shared__ float lds[];               // 2 buffers x P panels x 16 x 84 floats
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int row0 = blockIdx.x * 64 + wave * 16;
  const int n0 = blockIdx.y * kNW;
  const int arow = min(row0 + li, g.M - 1);

  f4 acc[P][NT];
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[p][n] = (f4){0.f, 0.f, 0.f, 0.f};

  // K chunks [0, nca) run over the aggregate `a`; chunks [nca, nca+nch) over the node's own features `h`.
  const int nca = (g.K + kKC - 1) / kKC;
  const int nch = HAS_H ? (g.Kh + kKC - 1) / kKC : 0;
  const int nc = nca + nch;
  const f4* img_a = reinterpret_cast<const f4*>(g.w_img + (size_t)blockIdx.y * nca * S * kPanel);
  const f4* img_h = HAS_H ? reinterpret_cast<const f4*>(g.wh_img + (size_t)blockIdx.y * nch * kPanel) : nullptr;

  // ---- staging of the next chunk -------------------------------------------------------------------------------
  // !HAS_H (the simple layer, the hot case): every load of the next chunk (weight image + A fragment) is an
  // UNCONDITIONAL inline-asm load waited for by hand right before it is written to LDS / rotated in.  hipcc cannot
  // count loads issued under per-thread range checks and would put `s_waitcnt vmcnt(0)` in front of the first
  // MFMA of every chunk, stalling the matrix pipe for a full L2/HBM round trip per chunk (1.38 -> 1.16 ms on C3).
  // Rules that keep this safe: an asm-loaded register is written by asm on EVERY path (no control-flow merge with
  // another definition -- the copy hipcc inserts at a merge would read the register before the load lands): range
  // checks are replaced by clamped addresses (redundant loads), the last iteration re-loads the last chunk.
  // HAS_H (tower variant, two images): compiler-scheduled loads, same structure.
  constexpr int SH = (kPanel / 4 + kBlock - 1) / kBlock;
  f4 st[SV];                                   // staged image of the next chunk
#pragma unroll
  for (int i = 0; i < SV; ++i) st[i] = (f4){0.f, 0.f, 0.f, 0.f};
  auto a_src = [&](int c, const float*& src, long& ld, int& kmax, int& k) {
    if (c < nca) { src = g.a; ld = g.lda; kmax = g.K; k = c * kKC + 4 * lg; }
    else { src = g.h; ld = g.ldh; kmax = g.Kh; k = (c - nca) * kKC + 4 * lg; }
  };
  auto stage_load = [&](int c) {
    if constexpr (!HAS_H) {
      const f4* src = img_a + (size_t)c * (S * kPanel / 4);
#pragma unroll
      for (int i = 0; i < SV; ++i) {
        const int idx = min((int)threadIdx.x + i * kBlock, S * kPanel / 4 - 1);
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(st[i]) : "v"((unsigned)idx * 16u), "s"(src) : "memory");
      }
    } else {
      if (c < nca) {
        const f4* src = img_a + (size_t)c * (S * kPanel / 4);
#pragma unroll
        for (int i = 0; i < SV; ++i) {
          const int idx = threadIdx.x + i * kBlock;
          if (idx < S * kPanel / 4) st[i] = src[idx];
        }
      } else {
        const f4* src = img_h + (size_t)(c - nca) * (kPanel / 4);
#pragma unroll
        for (int i = 0; i < SH; ++i) {
          const int idx = threadIdx.x + i * kBlock;
          if (idx < kPanel / 4) st[i] = src[idx];
        }
      }
    }
  };
  auto stage_write = [&](int c, int buf) {
    f4* base = reinterpret_cast<f4*>(lds + buf * P * kPanel);
    if (c < nca) {
#pragma unroll
      for (int i = 0; i < SV; ++i) {
        const int idx = threadIdx.x + i * kBlock;
        if (idx < S * kPanel / 4) base[idx] = st[i];
      }
    } else if (HAS_H) {
      f4* bh = base + S * kPanel / 4;
#pragma unroll
      for (int i = 0; i < SH; ++i) {
        const int idx = threadIdx.x + i * kBlock;
        if (idx < kPanel / 4) bh[idx] = st[i];
      }
    }
  };
  // A fragment of chunk c for this lane: floats [k, k+4) of its row, k = 16 c + 4 lg, read as ONE load from a
  // window clamped inside the row (floats [kk, kk+4), kk = min(k, kmax-4); the launcher guarantees kmax >= 4);
  // `fix_a` shifts / zero-fills after the wait.
  auto load_a = [&](int c) -> f4 {                     // !HAS_H: async, only valid after wait_staged()
    const float* src; long ld; int kmax, k;
    a_src(c, src, ld, kmax, k);
    const int kk = max(0, min(k, kmax - 4));
    const float* p = src + (size_t)arow * ld + kk;
    f4 v;
    if constexpr (!HAS_H) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    else v = reinterpret_cast<const f4u*>(p)->v;
    return v;
  };
  auto fix_a = [&](int c, f4 t) -> f4 {                // window [kk,kk+4) -> elements [k,k+4), 0 beyond kmax
    const float* src; long ld; int kmax, k;
    a_src(c, src, ld, kmax, k);
    const int d = k - max(0, min(k, kmax - 4));        // 0 except in the last, partial group of a row
    f4 v;
    v.x = d == 0 ? t.x : d == 1 ? t.y : d == 2 ? t.z : t.w;
    v.y = d == 0 ? t.y : d == 1 ? t.z : t.w;
    v.z = d == 0 ? t.z : t.w;
    v.w = t.w;
    v.x = (k < kmax && d <= 3) ? v.x : 0.f;
    v.y = (k + 1 < kmax && d <= 2) ? v.y : 0.f;
    v.z = (k + 2 < kmax && d <= 1) ? v.z : 0.f;
    v.w = (k + 3 < kmax && d == 0) ? v.w : 0.f;
    return v;
  };
  auto wait_staged = [&](f4& a_reg) {
    if constexpr (!HAS_H) {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(a_reg) : : "memory");
#pragma unroll
      for (int i = 0; i < SV; ++i) asm volatile("" : "+v"(st[i]));
    }
  };

  stage_load(0);
  f4 a_cur = load_a(0);
  wait_staged(a_cur);
  a_cur = fix_a(0, a_cur);
  stage_write(0, 0);
  __syncthreads();
  for (int c = 0; c < nc; ++c) {
    const int buf = c & 1;
    const int cn = min(c + 1, nc - 1);                   // the last iteration re-loads its own chunk (never written)
    stage_load(cn);                                      // issue only; consumed after the MFMAs
    f4 a_nxt = load_a(cn);
    const float* base = lds + buf * P * kPanel;
    const bool is_h = c >= nca;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float av = t == 0 ? a_cur.x : t == 1 ? a_cur.y : t == 2 ? a_cur.z : a_cur.w;
      const float* brow = base + (4 * lg + t) * kNP + li;
      if (!is_h) {
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[s][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, brow[s * kPanel + n * 16], acc[s][n], 0, 0, 0);
      } else if (HAS_H) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[P - 1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, brow[S * kPanel + n * 16], acc[P - 1][n], 0, 0, 0);
      }
    }
    wait_staged(a_nxt);
    a_nxt = fix_a(cn, a_nxt);
    if (c + 1 < nc) stage_write(c + 1, buf ^ 1);
    a_cur = a_nxt;
    __syncthreads();
  }

  // epilogue: C/D layout of 16x16 tiles: col = lane & 15, row = (lane >> 4) * 4 + reg
  float sc[S][4], rp[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = min(row0 + lg * 4 + r, g.M - 1);
    rp[r] = g.row_post ? g.row_post[row] : 1.f;
#pragma unroll
    for (int s = 0; s < S; ++s) sc[s][r] = g.row_scale[s] ? g.row_scale[s][row] : 1.f;
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = n0 + n * 16 + li;
    if (col >= g.N) continue;
    const float b = g.bias ? g.bias[col] : 0.f;
    const float cs = g.col_scale ? g.col_scale[col] : 1.f;
    const float ct = g.col_shift ? g.col_shift[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + lg * 4 + r;
      if (row >= g.M) continue;
      float v = b;
      if (HAS_H) v = v + acc[P - 1][n][r];
#pragma unroll
      for (int s = 0; s < S; ++s) v = v + sc[s][r] * acc[s][n][r];
      if (g.row_post) v = v * rp[r];                                   // graph-norm (pna_layer.py:71-72)
      if (g.col_scale) v = v * cs + ct;                                // eval-mode BatchNorm folded to an affine map
      if (g.relu) v = v > 0.f ? v : (v != v ? v : (g.relu == 2 ? g.slope * v : 0.f));   // ReLU / LeakyReLU (keep NaN)
      if (g.residual) v = g.residual[(size_t)row * g.ld_res + col] + v;  // h_in + h (pna_layer.py:212-213)
      g.y[(size_t)row * g.ldy + col] = v;
    }
  }
}

template <int S, bool HAS_H, int NT>
int launch_k(const GArgs& g, dim3 grid, hipStream_t st) {
  const size_t lds = (size_t)2 * (S + (HAS_H ? 1 : 0)) * kPanel * sizeof(float);
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute((const void*)k_posttrans<S, HAS_H, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return -1;
  hipLaunchKernelGGL((k_posttrans<S, HAS_H, NT>), grid, dim3(kBlock), lds, st, g);
  return 0;
}

template <int S, bool HAS_H>
int launch_nt(const GArgs& g, int nt, dim3 grid, hipStream_t st) {
  if (nt <= 1) return launch_k<S, HAS_H, 1>(g, grid, st);
  if (nt <= 3) return launch_k<S, HAS_H, 3>(g, grid, st);
  return launch_k<S, HAS_H, 5>(g, grid, st);
}

template <int S>
int launch_s(const GArgs& g, bool has_h, int nt, dim3 grid, hipStream_t st) {
  return has_h ? launch_nt<S, true>(g, nt, grid, st) : launch_nt<S, false>(g, nt, grid, st);
}

}  // namespace

extern "C" int64_t pna_posttrans_packed_floats(int32_t K, int32_t N, int32_t n_scaler, int32_t Kh, int64_t* wh_floats) {
  const int64_t nca = (K + kKC - 1) / kKC, nch = (Kh + kKC - 1) / kKC, nty = (N + kNW - 1) / kNW;
  if (wh_floats) *wh_floats = nty * nch * kPanel;
  return nty * nca * n_scaler * kPanel;
}

extern "C" int pna_posttrans_pack_f32(const float* w_ref, int64_t ldw, int32_t N, int32_t K, int32_t n_scaler, int32_t Kh,
                                      float* w_img, float* wh_img, pna_stream_t stream) {
  if (!w_ref || !w_img || N <= 0 || K <= 0 || n_scaler < 1 || n_scaler > 5 || Kh < 0 || (Kh > 0 && !wh_img) ||
      ldw < (int64_t)Kh + (int64_t)n_scaler * K)
    return pna_set_error(PNA_E_INVALID, "pna_posttrans_pack_f32: bad arguments");
  int64_t nh = 0;
  const int64_t nw = pna_posttrans_packed_floats(K, N, n_scaler, Kh, &nh);
  const int blocks = (int)((nw + nh + 255) / 256 > 4096 ? 4096 : (nw + nh + 255) / 256);
  hipLaunchKernelGGL(k_pack, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_ref, (long)ldw, N, K, n_scaler, Kh, w_img, wh_img);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}

extern "C" int pna_posttrans_f32(const pna_posttrans_args* p, pna_stream_t stream) {
  if (!p) return pna_set_error(PNA_E_INVALID, "pna_posttrans_f32: null args");
  if (int rc_ss = pna_check_struct_size("pna_posttrans_f32", p->struct_size, sizeof(*p))) return rc_ss;
  if (p->M < 0 || p->K <= 0 || p->N <= 0) return pna_set_error(PNA_E_INVALID, "pna_posttrans_f32: bad M/K/N");
  if (p->M == 0) return PNA_OK;
  if (!p->a || !p->w_img || !p->y) return pna_set_error(PNA_E_INVALID, "pna_posttrans_f32: a/w_img/y must be non-null");
  if (p->n_scaler < 1 || p->n_scaler > 5) return pna_set_error(PNA_E_INVALID, "pna_posttrans_f32: n_scaler must be 1..5");
  if (p->lda < p->K || p->ldy < p->N) return pna_set_error(PNA_E_INVALID, "pna_posttrans_f32: leading dimensions too small");
  if (p->K < 4 || (p->h != nullptr && p->Kh > 0 && p->Kh < 4))
    return pna_set_error(PNA_E_INVALID, "pna_posttrans_f32: K and Kh must be >= 4 (pad the operand and the weight with zero columns)");
  const bool has_h = p->h != nullptr && p->Kh > 0;
  if (has_h && (!p->wh_img || p->ldh < p->Kh)) return pna_set_error(PNA_E_INVALID, "pna_posttrans_f32: h given without wh_img / ldh too small");
  if (p->residual && p->ld_res < p->N) return pna_set_error(PNA_E_INVALID, "pna_posttrans_f32: ld_res too small");
  if ((p->col_scale == nullptr) != (p->col_shift == nullptr))
    return pna_set_error(PNA_E_INVALID, "pna_posttrans_f32: col_scale and col_shift come together");
  GArgs g;
  memset(&g, 0, sizeof(g));
  g.a = p->a; g.w_img = p->w_img; g.h = has_h ? p->h : nullptr; g.wh_img = has_h ? p->wh_img : nullptr; g.bias = p->bias;
  for (int s = 0; s < p->n_scaler; ++s) g.row_scale[s] = p->row_scale[s];
  g.row_post = p->row_post; g.col_scale = p->col_scale; g.col_shift = p->col_shift; g.residual = p->residual;
  g.y = p->y; g.lda = p->lda; g.ldh = p->ldh; g.ldy = p->ldy; g.ld_res = p->ld_res;
  g.M = p->M; g.K = p->K; g.N = p->N; g.Kh = has_h ? p->Kh : 0; g.relu = p->relu;
  if (p->relu < 0 || p->relu > 2) return pna_set_error(PNA_E_INVALID, "pna_posttrans_f32: relu must be 0, 1 or 2");
  g.slope = p->relu == 2 ? p->act_slope : 0.f;
  const int T = p->n_tower > 1 ? p->n_tower : 1;
  if (T > 1) {
    if (p->residual) return pna_set_error(PNA_E_INVALID, "pna_posttrans_f32: residual is not supported with n_tower > 1");
    if (T > 65535) return pna_set_error(PNA_E_INVALID, "pna_posttrans_f32: too many towers");
    g.ts_a = p->tower_stride_a; g.ts_h = p->tower_stride_h; g.ts_w = p->tower_stride_w; g.ts_wh = p->tower_stride_wh; g.ts_y = p->tower_stride_y;
  }
  dim3 grid((unsigned)((p->M + 63) / 64), (unsigned)((p->N + kNW - 1) / kNW), (unsigned)T);
  hipStream_t st = (hipStream_t)stream;
  // live 16-column tiles per workgroup (all column tiles of the grid use the same instantiation; columns >= N
  // are zero in the packed image and masked at the store)
  const int nt = p->N >= kNW ? kMaxNT : (p->N + 15) / 16;
  int rc;
  switch (p->n_scaler) {
    case 1: rc = launch_s<1>(g, has_h, nt, grid, st); break;
    case 2: rc = launch_s<2>(g, has_h, nt, grid, st); break;
    case 3: rc = launch_s<3>(g, has_h, nt, grid, st); break;
    case 4: rc = launch_s<4>(g, has_h, nt, grid, st); break;
    default: rc = launch_s<5>(g, has_h, nt, grid, st); break;
  }
  if (rc != 0) return pna_set_error(PNA_E_LAUNCH, "pna_posttrans_f32: could not reserve LDS");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}
