// pna_posttrans_x3w.hip -- the bf16x3 posttrans contraction on 32x32 matrix-core tiles ("wide" form).
// Implements pna_posttrans_x3w_{supported,packed_bytes,pack_f32,f32} of include/pna_amd.h.
//
//   y[v] = epilogue( bias + sum_s scale_s[v] * (W_s . a[v]) )                          (models/dgl/pna_layer.py:206)
//
// Same arithmetic as pna_posttrans_x3.hip (every fp32 operand cut exactly into three bf16 terms, six partial products per
// multiply, fp32 accumulation), restructured around v_mfma_f32_32x32x16_bf16:
//  * the 32x32 shape sustains ~15 % more flops per cycle than 16x16x32 (2382 vs 2075 TF/s in a pure-MFMA loop) and issues
//    half as many instructions per flop -- the 16x16 kernel was co-bound by instruction issue;
//  * the S scaler blocks of the weight are PACKED side by side into 32-column tiles: for S = 3, N = 75 the 225 (scaler,
//    column) pairs fill 7 tiles + 1 column, where 16-column tiles per scaler computed 3 x 80 = 240 and K padded to 320; the
//    odd column is evaluated on the VALU (8 fma per lane per K step, beside the operand split); K advances 16 at a time, so
//    K = 300 pads to 304;
//  * the epilogue goes through LDS: every wavefront deposits its 32 x N tile (scalers applied; packed columns of the same
//    output column combined in a fixed order) and reads it back row-major, so bias / graph-norm / BatchNorm / activation /
//    residual run on 16-byte vectors and every global load and store of the tail is a coalesced 16-byte access
//    (the 16x16 kernel stored 4 bytes per lane to 64-byte row fragments: 20 stores per 16 rows).
//
// Tiling: one persistent workgroup per CU, 8 wavefronts (2 per SIMD, <= 256 registers), each owning 32 rows x NT packed
// tiles (112-128 accumulator registers).  The weight is pre-split and pre-packed into the LDS image of every K step
// ([term][tile][k half][32 columns][8 k] bf16, 3 KB per tile), streamed by global_load_lds_dwordx4 through 3 buffers with ONE
// barrier in the middle of each step (as pna_posttrans_x3.hip's pipeline 3); a B fragment is one conflict-free ds_read_b128.
// A is fetched three steps ahead into a 3-slot register ring, split one step ahead.  The step loop is unrolled by 3 so that
// the LDS buffer and the register slot of a step are compile-time.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "pna_amd.h"
#include "pna_internal.h"
#include "pna_x3_split.h"

#ifndef X3W_FINE
#define X3W_FINE 1     // the operand split in 12 pieces, one behind each MFMA of group H (0: 6 pieces, one behind each MFMA pair)
#endif
#ifndef X3W_SKIP
#define X3W_SKIP 0     // development (tools/x3w_variants.sh; results are garbage): 1 cheap epilogue, 2 no MFMAs, 4 no B reads,
#endif                 // 8 no weight copies, 16 no A loads, 32 no operand split, 64 no soft barrier

namespace {

using namespace pna_x3;
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8m __attribute__((ext_vector_type(8)));

constexpr int kWaves = 8, kThreads = kWaves * 64, kWaveRows = 32, kTileRows = kWaves * kWaveRows;
constexpr int kNBuf = 4;           // weight step images in LDS
constexpr int kHalfRows = 16;      // rows of a wavefront's tile that pass through the epilogue's staging area at a time

// How N output columns x S scalers are laid into 32-column tiles (host and device agree through these numbers).
struct Plan {
  int ok;     // shape supported
  int NB;     // output columns per workgroup column block
  int ny;     // column blocks
  int ntf;    // 32-column groups of output columns whose S scaler tiles combine in-lane
  int r;      // remaining output columns NB - 32 * ntf; their S * r packed columns fill the mixed tiles in (s, j) order
  int nmix;   // mixed tiles
  int nvc;    // 1 = the last packed column (s = S-1, j = NB-1) is evaluated on the VALU
  int nt;     // S * ntf + nmix
};
inline Plan make_plan(int N, int S) {
  Plan p;
  memset(&p, 0, sizeof(p));
  if (N <= 0 || S < 1 || S > 3) return p;
  if (N <= 80) { p.NB = N; p.ny = 1; }
  else if (N % 64 == 0) { p.NB = 64; p.ny = N / 64; }
  else return p;
  p.ntf = p.NB / 32;
  p.r = p.NB % 32;
  const int q = S * p.r;
  p.nvc = (q % 32 == 1 && q > 1) ? 1 : 0;
  p.nmix = (q - p.nvc + 31) / 32;
  p.nt = S * p.ntf + p.nmix;
  p.ok = p.nt >= 1 && p.nt <= 8;
  return p;
}
inline int steps_of(int K) { return (K + 15) / 16; }         // K advances 16 at a time
// k of element 0 of lane half g in step s: k = 16 s + 8 g -- the two halves of a row read 64 contiguous bytes per step (one
// sector; with k = 32 c + 16 g + 8 o they read two 32-byte pieces 64 bytes apart, every sector was requested by two steps
// and the A stream ran at 2.7 TB/s).  A and B agree on it, nothing else sees it.
__host__ __device__ inline int k_of(int s, int g, int K) {
  (void)K;
  return 16 * s + 8 * g;
}

struct WArgs {
  const float* a; const unsigned char* w_img; const float* wv_img; const float* bias;
  const float* row_scale[3];
  const float* row_post; const float* col_scale; const float* col_shift; const float* residual;
  float* y;
  long lda, ldy, ld_res;
  int M, K, NB, r, YP, NS, relu, vec_ok;
  float slope;
  unsigned long long* dbg;     // -DPNA_X3W_TIMERS builds only (tools/x3w_timers.py): raw time stamps of the first steps
};

// ---- weight packing ------------------------------------------------------------------------------------------------
// w_img[by][s_step][term][tile][g][c][e] = term(w_ref[by*NB + j][sc*K + k_of(s_step, g) + e]),  (tile, c) -> (sc, j):
//   tile <  S*ntf : sc = tile / ntf, j = 32 * (tile % ntf) + c
//   tile >= S*ntf : q = 32 * (tile - S*ntf) + c;  q < S*r - nvc ? (sc = q / r, j = 32*ntf + q % r) : zero column
// wv_img[by][s_step][g][e] (fp32) = w_ref[by*NB + NB-1][(S-1)*K + k_of(s_step, g) + e]   (only when nvc)
__global__ void k_pack_x3w(const float* w_ref, long ldw, int K, int S, Plan p, int NS, unsigned short* w_img, float* wv_img) {
  const long per_step = 3L * p.nt * 2 * 32 * 8;
  const long total_w = (long)p.ny * NS * per_step, total_v = p.nvc ? (long)p.ny * NS * 16 : 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total_w + total_v; i += (long)gridDim.x * blockDim.x) {
    if (i < total_w) {
      long x = i;
      const int e = x % 8; x /= 8;
      const int c = x % 32; x /= 32;
      const int g = x % 2; x /= 2;
      const int tile = x % p.nt; x /= p.nt;
      const int term = x % 3; x /= 3;
      const int st = x % NS; x /= NS;
      const int by = (int)x;
      int sc = -1, j = 0;
      if (tile < S * p.ntf) { sc = tile / p.ntf; j = 32 * (tile % p.ntf) + c; }
      else {
        const int q = 32 * (tile - S * p.ntf) + c;
        if (q < S * p.r - p.nvc) { sc = q / p.r; j = 32 * p.ntf + q % p.r; }
      }
      const int k = k_of(st, g, K) + e;
      float w = 0.f;
      if (sc >= 0 && k < K) w = w_ref[(long)(by * p.NB + j) * ldw + (long)sc * K + k];
      w_img[i] = weight_term(w, term);
    } else {
      long x = i - total_w;
      const int e = x % 8; x /= 8;
      const int g = x % 2; x /= 2;
      const int st = x % NS; x /= NS;
      const int by = (int)x;
      const int k = k_of(st, g, K) + e;
      wv_img[i - total_w] = k < K ? w_ref[(long)(by * p.NB + p.NB - 1) * ldw + (long)(S - 1) * K + k] : 0.f;
    }
  }
}

template <int S, int NTF, int NMIX, int NVC>
__global__ __launch_bounds__(kThreads, 1) void k_posttrans_x3w(const WArgs g) {
  constexpr int NT = S * NTF + NMIX;
  static_assert(NT >= 4 && NT <= 8, "tile groups below assume 4 .. 8 packed tiles");
  constexpr int NP = 4;                          // tile groups of 2 or 1 tiles: the MFMAs of a group alternate between its accumulators
  constexpr int STEPV = NT * 192;                // 16-byte pieces of one step image
  constexpr int STEPB = STEPV * 16;
  constexpr int NI = (STEPV + kThreads - 1) / kThreads;   // global_load_lds instructions per wavefront per step
  constexpr int NFULL = 32 * NTF;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 31, lh = lane >> 5;
  const int NS = g.NS, YP = g.YP, NB = g.NB;
  const int n0 = blockIdx.y * NB;
  const int ntiles = (g.M + kTileRows - 1) / kTileRows;
  if ((int)blockIdx.x >= ntiles) return;
  const unsigned char* img = g.w_img + (size_t)blockIdx.y * NS * STEPB;
  // LDS: kNBuf step images | per wavefront a 16-row x YP staging tile of the epilogue | column constants | row factors | VALU column
  float* const ytile_w = reinterpret_cast<float*>(lds + kNBuf * STEPB) + (size_t)wave * kHalfRows * YP;   // wave-uniform
  float* const colc = reinterpret_cast<float*>(lds + kNBuf * STEPB) + (size_t)kWaves * kHalfRows * YP;   // [3][YP]: bias | scale | shift
  float* const rowf = colc + 3 * YP;                                                          // [8 waves][S + 1][32] per-row factors
  float* const wv = rowf + kWaves * (S + 1) * 32;                                             // [NS][2][8] (NVC only)
  unsigned* const arrived = reinterpret_cast<unsigned*>(wv + (NVC ? NS * 16 : 0));            // the soft barrier's arrival counter

  f16v acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  float vacc = 0.f, vpend = 0.f;                  // the VALU column: this lane's k half of its row | the next step's share

  // ---- column constants, row-factor defaults and the VALU column's weights -> LDS -------------------------------------
  for (int i = tid; i < YP; i += kThreads) {
    const int col = n0 + i;
    const bool in = i < NB;
    colc[i] = (g.bias && in) ? g.bias[col] : 0.f;
    colc[YP + i] = (g.col_scale && in) ? g.col_scale[col] : 1.f;
    colc[2 * YP + i] = (g.col_shift && in) ? g.col_shift[col] : 0.f;
  }
  for (int i = tid; i < kWaves * (S + 1) * 32; i += kThreads) rowf[i] = 1.f;
  if (tid == 0) *arrived = 0u;
  if constexpr (NVC) {
    for (int i = tid; i < NS * 16; i += kThreads) wv[i] = g.wv_img[(size_t)blockIdx.y * NS * 16 + i];
  }

  // step image -> LDS buffer, asynchronously: every wavefront issues exactly NI 1 KB copies per step (a slot past the
  // image re-copies an earlier piece: same bytes to the same address), destination = wave-uniform base + lane * 16
  // The copies are issued through inline asm: an LDS-DMA copy hipcc can SEE makes it drain vmcnt(0) in front of the next LDS
  // access it can see as well (the copy might alias it) -- at the top of every step, which serialised the whole prefetch.
  auto lds_dma16 = [&](const void* src, const void* dst_wave_base) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                 : : "s"((unsigned)(size_t)(const __attribute__((address_space(3))) void*)dst_wave_base), "v"(src) : "memory", "m0");
  };
  auto lds_dma4 = [&](const void* src, const void* dst_wave_base) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off"
                 : : "s"((unsigned)(size_t)(const __attribute__((address_space(3))) void*)dst_wave_base), "v"(src) : "memory", "m0");
  };
  auto stage_piece = [&](int st, int buf, int i) __attribute__((always_inline)) {
    const unsigned char* src = img + (size_t)st * STEPB;
    unsigned char* dst = lds + (size_t)buf * STEPB;
    int w0 = (i * kWaves + wave) * 64;
    if (w0 >= STEPV) w0 = w0 % STEPV;
    lds_dma16(src + (size_t)(w0 + lane) * 16, dst + (size_t)w0 * 16);
  };

  // ---- A operand -----------------------------------------------------------------------------------------------------
  // lane (row lc, half lh) of (tile t, step s): floats [k, k+8), k = k_of(s, lh), as two clamped 16-byte windows fixed
  // up after the load (only a row's last step can reach beyond K).  Rows beyond M re-read row M-1 (never stored).
  f4 raw[3][2];                                  // register ring: step j lives in slot j % 3
  u4 T[3], Tn[3];                                // the three bf16 terms (8 x bf16 each) of the step being multiplied | of the next
  auto load_a = [&](f4 (&dst)[2], int t, int s) __attribute__((always_inline)) {
#if X3W_SKIP & 128       // development: every tile reads the first tile's rows (A comes from L2)
    const int row = min(0 * t + wave * kWaveRows + lc, g.M - 1);
#else
    const int row = min(t * kTileRows + wave * kWaveRows + lc, g.M - 1);
#endif
    const int k = k_of(s, lh, g.K);
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const int kk = max(0, min(k + 4 * w, g.K - 4));
      const float* ptr = g.a + (size_t)row * g.lda + kk;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[w]) : "v"(ptr) : "memory");
    }
  };
  // slot -> the three bf16 terms (the slot's loads are known to have landed: see the counted wait of the step loop).
  // take_fast is straight-line code, so that it can be scheduled between the MFMAs of a step; the rare cases it does not
  // handle -- a row's last K step reaching beyond K, an infinite element -- are redone by take_slow behind the MFMA stream.
  auto vcol = [&](const f4 lo4, const f4 hi4, int s) -> float __attribute__((always_inline)) {
    const f4 w0 = *reinterpret_cast<const f4*>(wv + (s * 2 + lh) * 8), w1 = *reinterpret_cast<const f4*>(wv + (s * 2 + lh) * 8 + 4);
    float v = 0.f;             // (added to vacc at the top of ITS step: the epilogue of the previous tile runs in between)
#pragma unroll
    for (int e = 0; e < 4; ++e) v = __builtin_fmaf(lo4[e], w0[e], v);
#pragma unroll
    for (int e = 0; e < 4; ++e) v = __builtin_fmaf(hi4[e], w1[e], v);
    return v;
  };
  // chunk c of take_fast: c < 4 splits elements 2c, 2c+1 (dword c of each term); 4: is the slow path needed (wave-uniform);
  // 5: the VALU column's share
  auto take_chunk = [&](f4 (&cur)[2], int s, u4 (&Tu)[3], int c, bool& redo) __attribute__((always_inline)) {
    if (c == 0) {
      asm volatile("" : "+v"(cur[0]), "+v"(cur[1]) : : "memory");
      // beyond K (only a row's last step reaches it; K % 4 == 0, so a 16-byte window is inside or outside as a whole):
      // the clamped load returned other columns -- zero them (straight-line: 2 compares + 8 selects per step)
      const int k = k_of(s, lh, g.K);
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const bool in = k + 4 * w < g.K;
#pragma unroll
        for (int e = 0; e < 4; ++e) cur[w][e] = in ? cur[w][e] : 0.f;
      }
    }
    if (c < 4) {
      const float xe = cur[c >> 1][(2 * c) & 3], xo = cur[c >> 1][(2 * c + 1) & 3];
      const float re = xe - top16(xe), ro = xo - top16(xo);
      const float se = re - top16(re), so = ro - top16(ro);
      Tu[0][c] = pack_hi(xe, xo);
      Tu[1][c] = pack_hi(re, ro);
      Tu[2][c] = pack_hi(se, so);
      asm volatile("" : "+v"(Tu[0][c]), "+v"(Tu[1][c]), "+v"(Tu[2][c]));      // (pins the arithmetic HERE: the compiler would sink it to the use)
    } else if (c == 4) {
      redo = __builtin_amdgcn_ballot_w64(absmax8(cur[0], cur[1]) == INFINITY) != 0;        // wave-uniform
    } else {
      if constexpr (NVC) {
        vpend = vcol(cur[0], cur[1], s);
        asm volatile("" : "+v"(vpend));
      }
    }
  };
  // The same work in 12 pieces, one behind EACH of the 12 MFMAs of a two-tile group (2 MFMAs + 11 VALU left the pipe idle for
  // the last VALU of every piece when the SIMD's other wavefront was not in its MFMA stream): 0/1 zero the windows beyond K and
  // request the VALU column's weights, 2..9 the four element pairs in two halves each, 10 the infinity test, 11 the VALU column
  struct TakeState { float re, ro, se, so; f4 w0, w1; };
  auto take_micro = [&](f4 (&cur)[2], int s, u4 (&Tu)[3], int m, bool& redo, TakeState& ts) __attribute__((always_inline)) {
    if (m < 2) {
      if (m == 0) {
        asm volatile("" : "+v"(cur[0]), "+v"(cur[1]) : : "memory");
        if constexpr (NVC) {
          ts.w0 = *reinterpret_cast<const f4*>(wv + (s * 2 + lh) * 8);
          ts.w1 = *reinterpret_cast<const f4*>(wv + (s * 2 + lh) * 8 + 4);
        }
      }
      const bool in = k_of(s, lh, g.K) + 4 * m < g.K;
#pragma unroll
      for (int e = 0; e < 4; ++e) cur[m][e] = in ? cur[m][e] : 0.f;
      asm volatile("" : "+v"(cur[m]));
    } else if (m < 10) {
      const int c = (m - 2) >> 1;
      const float xe = cur[c >> 1][(2 * c) & 3], xo = cur[c >> 1][(2 * c + 1) & 3];
      if (((m - 2) & 1) == 0) {
        ts.re = xe - top16(xe); ts.ro = xo - top16(xo);
        ts.se = ts.re - top16(ts.re); ts.so = ts.ro - top16(ts.ro);
        asm volatile("" : "+v"(ts.re), "+v"(ts.ro), "+v"(ts.se), "+v"(ts.so));
      } else {
        Tu[0][c] = pack_hi(xe, xo);
        Tu[1][c] = pack_hi(ts.re, ts.ro);
        Tu[2][c] = pack_hi(ts.se, ts.so);
        asm volatile("" : "+v"(Tu[0][c]), "+v"(Tu[1][c]), "+v"(Tu[2][c]));
      }
    } else if (m == 10) {
      redo = __builtin_amdgcn_ballot_w64(absmax8(cur[0], cur[1]) == INFINITY) != 0;        // wave-uniform
    } else {
      if constexpr (NVC) {
        float v = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) v = __builtin_fmaf(cur[0][e], ts.w0[e], v);
#pragma unroll
        for (int e = 0; e < 4; ++e) v = __builtin_fmaf(cur[1][e], ts.w1[e], v);
        vpend = v;
        asm volatile("" : "+v"(vpend));
      }
    }
  };
  // an infinite element (a pathological aggregate): the fragment is fetched AGAIN, synchronously, and split by the form that
  // keeps infinities (the ring slot it came from already receives a later step's loads)
  auto take_slow = [&](int t, int s, u4 (&Tu)[3]) __attribute__((always_inline)) {
    f4 cur[2];
    load_a(cur, t, s);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]) : : "memory");
    const int k = k_of(s, lh, g.K);
    const f4 lo4 = fix4(k, g.K, cur[0]), hi4 = fix4(k + 4, g.K, cur[1]);
    if constexpr (NVC) vpend = vcol(lo4, hi4, s);
    bf8 t0, t1, t2;
    split8_inf(lo4, hi4, t0, t1, t2);
    Tu[0] = __builtin_bit_cast(u4, t0); Tu[1] = __builtin_bit_cast(u4, t1); Tu[2] = __builtin_bit_cast(u4, t2);
  };

  // ---- epilogue ------------------------------------------------------------------------------------------------------
  // C/D layout of v_mfma_f32_32x32x16: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
  // The per-row factors of a tile (S scalers + the graph-norm factor) are copied into LDS when the tile STARTS (tile_begin,
  // global_load_lds_dword by lanes 0..31; an absent factor stays at the 1.0 written above), so the epilogue reads them
  // with no global round trip.  The tile goes through the staging area in two halves of 16 rows (registers 0..7, then 8..15).
  constexpr int NQMAX = NMIX == 0 ? NFULL / 4 : (NFULL + 16) / 4;   // 16-byte slots per tile row, at most (r <= 16 with mixed tiles)
  float* const myf_w = rowf + wave * (S + 1) * 32;                // [S + 1][32]: scaler 0 .. S-1 | graph norm (wave-uniform)
  auto tile_begin = [&](int t) __attribute__((always_inline)) {
    int lane_ = lane;
    asm volatile("" : "+v"(lane_));
    const int row = min(t * kTileRows + wave * kWaveRows + (lane_ & 31), g.M - 1);
    if (lane_ < 32) {
#pragma unroll
      for (int f = 0; f <= S; ++f) {
        const float* src = f < S ? g.row_scale[f] : g.row_post;
        if (src) lds_dma4(src + row, myf_w + f * 32);             // (wave-uniform)
      }
    }
  };
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    vacc = 0.f;
  };
#ifdef PNA_X3W_TIMERS
  int dbg_j = 0;
  auto estamp = [&](int idx) __attribute__((always_inline)) {
    if (g.dbg && dbg_j < 64) {
      const unsigned long long tm = clock64();
      if (lane == 0) g.dbg[(((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kWaves + wave) * 64 + dbg_j) * 8 + idx] = tm;
    }
  };
#define X3W_ESTAMP(hf, idx) if (hf == 0) estamp(idx)
#else
#define X3W_ESTAMP(hf, idx)
#endif
  // Read-back geometry.  Fast form (y / residual rows 16-byte aligned): the NB / 4 whole 16-byte slots of the half tile's 16
  // rows are dealt to the lanes in row-major order, NITF iterations of one ds_read_b128 + one 16-byte store per lane with no
  // per-element predicate; the NB % 4 columns beyond them (3 for N = 75) take ONE more iteration of 4-byte accesses.
  constexpr int NITF = (kHalfRows * NQMAX + 63) / 64;
  auto finish1 = [&](float v, float cb, float cs, float ct, float rp, float res, float lo, bool leaky) -> float __attribute__((always_inline)) {
    float x = (v + cb) * rp;
    x = __builtin_fmaf(x, cs, ct);
    x = x < lo ? (leaky ? x * g.slope : 0.f) : x;            // ReLU / LeakyReLU / none (lo = -inf); NaN < lo is false: kept
    return res + x;
  };
  auto epilogue = [&](int t) __attribute__((always_inline)) {
    const int row0 = t * kTileRows + wave * kWaveRows;
    if (row0 >= g.M) { zero_acc(); return; }
    // the lane ids go through an empty asm so that everything derived from them is rebuilt HERE instead of being hoisted
    // out of the step loop and kept in registers (or scratch) across the MFMA stream
    int lane_ = lane;
    asm volatile("" : "+v"(lane_));
    const int lc = lane_ & 31, lh = lane_ >> 5;
    float* const ytile = ytile_w;
    float* const myf = myf_w;
    const int r = g.r;
    const float lo = g.relu ? 0.f : -INFINITY;
    const bool leaky = g.relu == 2;
    float vt = 0.f;
    if constexpr (NVC) {
      const float other = __shfl_xor(vacc, 32);
      vt = ((lh == 0 ? vacc : other) + (lh == 0 ? other : vacc)) * myf[(S - 1) * 32 + lc];
    }
    // read-back coordinates of the fast form
    const int NQF = max(NB >> 2, 1), REM = NB & 3;
    const int dq = 64 % NQF, drow = 64 / NQF;                         // (launcher: NB >= 4 when vec_ok)
    const int rrow0 = lane_ / NQF, rq0 = lane_ - rrow0 * NQF;
    const int xrow = REM == 1 ? lane_ : REM == 2 ? lane_ >> 1 : (lane_ * 171) >> 9, xe = lane_ - xrow * REM;   // remainder columns: lane / REM
    // The residual: the loads of a half are issued ahead of its deposit and waited for ONCE, before the half's first store.
    // (A wait the compiler places itself is vmcnt(0) at every control-flow merge, and from the second one on it also
    // waits for the stores issued in between: one store round trip per slot.)
    f4 res[NITF];
    float resx;
    auto load_res = [&](int hf) __attribute__((always_inline)) {
#pragma unroll
      for (int it = 0; it < NITF; ++it) res[it] = (f4){0.f, 0.f, 0.f, 0.f};
      resx = 0.f;
      if (g.vec_ok && g.residual) {
        int rr = rrow0, rq = rq0;
#pragma unroll
        for (int it = 0; it < NITF; ++it) {
          const int row = row0 + kHalfRows * hf + rr;
          if (rr < kHalfRows && row < g.M) res[it] = *reinterpret_cast<const f4*>(g.residual + (size_t)row * g.ld_res + n0 + 4 * rq);
          rq += dq; rr += drow;
          if (rq >= NQF) { rq -= NQF; ++rr; }
        }
        const int row = row0 + kHalfRows * hf + xrow;
        if (REM != 0 && xrow < kHalfRows && row < g.M) resx = g.residual[(size_t)row * g.ld_res + n0 + 4 * NQF + xe];
      }
    };
    // mixed tiles: packed column qi = 32 m + lc belongs to scaler qi / r and output column NFULL + qi % r.  The lanes of
    // scaler 0 (qi < r: all in mixed tile 0) gather the other scalers' products of their output column from the lanes that
    // hold them (ds_bpermute: a register exchange through the LDS crossbar, no memory) and add them in scaler order.
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      load_res(hf);
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int q = 2 * hf + qq;
        // the factors of the lane's 4 adjacent rows 8 q + 4 lh + {0..3}
        f4 sc[S];
#pragma unroll
        for (int s = 0; s < S; ++s) sc[s] = *reinterpret_cast<const f4*>(myf + s * 32 + 8 * q + 4 * lh);
        float* const yl = ytile + (8 * qq + 4 * lh) * YP;
        // full tiles: the S scaler tiles of the same 32 output columns sit in the same lane
#pragma unroll
        for (int jt = 0; jt < NTF; ++jt)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = sc[0][e] * acc[jt][4 * q + e];
#pragma unroll
            for (int s = 1; s < S; ++s) x = __builtin_fmaf(sc[s][e], acc[s * NTF + jt][4 * q + e], x);
            yl[e * YP + 32 * jt + lc] = x;
          }
        if constexpr (NMIX > 0) {
          float xm[NMIX][4];
#pragma unroll
          for (int m = 0; m < NMIX; ++m) {
            const int sm = (32 * m + lc) / r;                      // this lane's scaler in mixed tile m (>= S: an unused column)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float scl = sc[0][e];
              if constexpr (S > 1) scl = sm == 1 ? sc[1][e] : scl;
              if constexpr (S > 2) scl = sm == 2 ? sc[2][e] : scl;
              xm[m][e] = scl * acc[S * NTF + m][4 * q + e];
            }
          }
          float tot[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) tot[e] = xm[0][e];
#pragma unroll
          for (int s = 1; s < S; ++s) {
            const int qs = s * r + lc;                              // packed column of (scaler s, this lane's output column)
            const int src = ((qs & 31) + 32 * lh) * 4;              // its lane (same half), as a ds_bpermute byte index
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, xm[0][e])));
              if constexpr (NMIX > 1) {
                const float v1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, xm[1][e])));
                v = (qs >> 5) == 1 ? v1 : v;
              }
              if (!(NVC && s == S - 1) || lc != r - 1) tot[e] = tot[e] + v;     // (the VALU column's lane: added below)
            }
          }
          if (lc < r) {
#pragma unroll
            for (int e = 0; e < 4; ++e) yl[e * YP + NFULL + lc] = tot[e];
          }
        }
      }
      if constexpr (NVC) {                                   // the VALU column: row lc's product, added last (scaler order)
        if (lh == 0 && (lc >> 4) == hf) {
          float* p = ytile + (lc & 15) * YP + NB - 1;
          *p = *p + vt;
        }
      }
      asm volatile("" ::: "memory");
      X3W_ESTAMP(hf, 6);
#pragma unroll
      for (int it = 0; it < NITF; ++it) asm volatile("" : "+v"(res[it]));   // (unconditional: on every path the loads are waited for HERE)
      asm volatile("" : "+v"(resx));
      if (g.vec_ok) {
        X3W_ESTAMP(hf, 7);
        int rr = rrow0, rq = rq0;
#pragma unroll
        for (int it = 0; it < NITF; ++it) {
          const int row = row0 + kHalfRows * hf + rr;
          if (rr < kHalfRows && row < g.M) {
            f4 v = *reinterpret_cast<const f4*>(ytile + rr * YP + 4 * rq);
            const f4 cb = *reinterpret_cast<const f4*>(colc + 4 * rq), cs = *reinterpret_cast<const f4*>(colc + YP + 4 * rq),
                     ct = *reinterpret_cast<const f4*>(colc + 2 * YP + 4 * rq);
            const float rp = myf[S * 32 + kHalfRows * hf + rr];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = finish1(v[e], cb[e], cs[e], ct[e], rp, res[it][e], lo, leaky);
            *reinterpret_cast<f4*>(g.y + (size_t)row * g.ldy + n0 + 4 * rq) = v;
          }
          rq += dq; rr += drow;
          if (rq >= NQF) { rq -= NQF; ++rr; }
        }
        {
          const int row = row0 + kHalfRows * hf + xrow;
          if (REM != 0 && xrow < kHalfRows && row < g.M) {
            const int c = 4 * NQF + xe;
            const float v = finish1(ytile[xrow * YP + c], colc[c], colc[YP + c], colc[2 * YP + c], myf[S * 32 + kHalfRows * hf + xrow], resx, lo, leaky);
            g.y[(size_t)row * g.ldy + n0 + c] = v;
          }
        }
      } else {
        // any alignment: one element per lane and trip (a plain loop: this form is not the one to be fast)
        for (int id = lane_; id < kHalfRows * NB; id += 64) {
          const int rr = id / NB, c = id - rr * NB, row = row0 + kHalfRows * hf + rr;
          if (row < g.M) {
            const float rs = g.residual ? g.residual[(size_t)row * g.ld_res + n0 + c] : 0.f;
            g.y[(size_t)row * g.ldy + n0 + c] =
                finish1(ytile[rr * YP + c], colc[c], colc[YP + c], colc[2 * YP + c], myf[S * 32 + kHalfRows * hf + rr], rs, lo, leaky);
          }
        }
      }
      asm volatile("" ::: "memory");
    }
    zero_acc();
  };

  // ---- the step pipeline -----------------------------------------------------------------------------------------------
  // Global step j multiplies K step s_j = j mod NS of tile t_j, reads weight buffer j % 4 and A slot j % 3.  ONE barrier per
  // step, in the middle of its MFMA stream (B_j).  Behind B_j every wavefront has left step j-1, so buffer (j+3) % 4 is
  // free: the copies of step j+3's image are issued there and waited for -- by the wavefront that issued them, with a COUNTED
  // vmcnt -- ahead of B_{j+2}, the last barrier before step j+3 reads them: two whole steps for an L2 -> LDS copy, and the
  // same wait leaves in flight the A loads of the last two steps (fetched three steps ahead).  (With three buffers the wait
  // had to sit ONE step behind the copies and covered A loads issued one step earlier: every step stalled on HBM latency.)
  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int nsteps = my_tiles * NS;
  int t = blockIdx.x, s = 0;                       // position of step j
  int t3 = t, s3 = 0;                              // position of step j + 3 (weight image) ...
  int t4, s4;                                      // ... and of step j + 4 (A fragment)
  auto adv = [&](int& tt, int& ss) __attribute__((always_inline)) {
    if (++ss == NS) { ss = 0; tt += gridDim.x; }
  };
#pragma unroll
  for (int b = 0; b < 3; ++b) {
#pragma unroll
    for (int i = 0; i < NI; ++i) stage_piece(b % NS, b, i);
  }
  load_a(raw[0], t3, s3); adv(t3, s3);
  load_a(raw[1], t3, s3); adv(t3, s3);
  load_a(raw[2], t3, s3); adv(t3, s3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                 // colc / rowf / wv are written, the first three images have landed
  {
    bool redo0 = false;
#pragma unroll
    for (int c = 0; c < 6; ++c) take_chunk(raw[0], 0, T, c, redo0);
    if (redo0) take_slow(t, 0, T);
  }
  load_a(raw[0], t3, s3);                          // step 3's fragment: slot 0 is free again
  t4 = t3; s4 = s3; adv(t4, s4);

  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + (unsigned)lane * 16u;
  int buf = 0;                                     // j % 4
  // tile groups: 2 tiles while more tiles than groups remain, then 1 (6 tiles: 2 2 1 1; 7: 2 2 2 1; 8: 2 2 2 2)
  auto gstart = [](int gi) constexpr -> int { int st = 0; for (int i = 0; i < gi; ++i) st += (NT - st > NP - i) ? 2 : 1; return st; };
  auto gcnt = [gstart](int gi) constexpr -> int { return (NT - gstart(gi) > NP - gi) ? 2 : 1; };
  // B fragment (term tm, tile n): bytes (tm * NT + n) * 1024 + lane * 16 of the buffer; read by hand with counted
  // lgkmcnt (LDS returns in order): the reads of group p+1 are issued behind those of group p
  auto load_b = [&](bf8 (&B)[2][2][3], int gi, int slot, unsigned base) __attribute__((always_inline)) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      if (jj < gcnt(gi)) {
#pragma unroll
        for (int tm = 0; tm < 3; ++tm)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(B[slot][jj][tm]) : "v"(base), "n"((tm * NT + gstart(gi) + jj) * 1024));
      }
    }
  };
  // Soft barrier: ARRIVE (one LDS atomic by lane 0) and WAIT (poll) are half a step apart.  A wavefront signals barrier j+1
  // as soon as it (i) has read its last B fragment of step j and (ii) its own copies of step j+2's image have landed; it
  // waits for all eight signals in the middle of step j+1, before it overwrites step j's buffer and before its first read of
  // image j+2.  A wavefront that is late -- in its epilogue, or starved of the matrix pipe by its SIMD neighbour -- has
  // signalled long before the others ask: with s_barrier the early wavefronts of every step idled ~1500 cycles.
  const unsigned arr_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)arrived;
  auto arrive = [&]() __attribute__((always_inline)) {
    unsigned long long keep;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_add_u32 %1, %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(keep) : "v"(arr_addr), "v"(1u) : "memory");
  };
  auto wait_all = [&](unsigned target) __attribute__((always_inline)) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(arr_addr) : "memory");
    while (__builtin_amdgcn_readfirstlane(v) < target) {
      __builtin_amdgcn_s_sleep(1);
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(arr_addr) : "memory");
    }
  };
#ifdef PNA_X3W_TIMERS
  auto stamp = [&](int j, int idx) __attribute__((always_inline)) {
    if (g.dbg && j < 64) {
      const unsigned long long tm = clock64();
      if (lane == 0) g.dbg[(((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kWaves + wave) * 64 + j) * 8 + idx] = tm;
    }
  };
#define X3W_STAMP(j, idx) stamp(j, idx)
#else
#define X3W_STAMP(j, idx)
#endif
  auto step = [&](auto slot_c, int j) __attribute__((always_inline)) {
    constexpr int R = decltype(slot_c)::value;     // j % 3: register slot of this step
    constexpr int R1 = (R + 1) % 3;
    X3W_STAMP(j, 0);
    if (s == 0) tile_begin(t);                     // (older than the loads below: covered by the counted waits)
    if constexpr (NVC) vacc += vpend;
    const int simg = s3;                           // image index of step j + 3
    bool redo = false;
    TakeState tks;
    int s1 = s + 1; if (s1 >= NS) s1 -= NS;
    const int buf3 = (buf + 3) & 3;
    const unsigned ba = lds_base + (unsigned)(buf * STEPB);
    bf8 B[2][2][3];
    constexpr int H = 1;                           // the barrier follows the B prefetch of group H
    constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};   // smallest partial products first
#if !(X3W_SKIP & 4)
    load_b(B, 0, 0, ba);
#endif
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const bool two = gcnt(p) == 2;
      // the B fragments of the NEXT group are requested behind those of this one.  (Requesting the next STEP's first group
      // behind the last one was measured: no gain, and 24 more registers live across the epilogue.)
#if !(X3W_SKIP & 4)
      if (p + 1 < NP) {
        load_b(B, p + 1, (p + 1) & 1, ba);
        if (gcnt(p + 1) == 2) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
#endif
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        if (jj < gcnt(p)) asm volatile("" : "+v"(B[p & 1][jj][0]), "+v"(B[p & 1][jj][1]), "+v"(B[p & 1][jj][2]));
#if !(X3W_SKIP & 64)
      if (p == H) wait_all((unsigned)kWaves * (unsigned)j);      // barrier j: signalled by every wavefront at the end of its step j-1
#endif
      if (p == NP - 1) {
        // barrier j+1: this wavefront's reads of step j's buffer are complete (the wait above) and so are its copies of step
        // j+2's image (issued in step j-1): in flight may stay what was issued behind them -- the 2 A loads of step j-1, the NI
        // copies and the 2 A loads of this step.  Everything older has landed, the fragment of step j+2 (fetched in step j-2,
        // 2.25 steps ago) included.  VMEM returns in order: a wait that covered younger A loads made every step wait for HBM.
#if !(X3W_SKIP & 64)
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(4 + NI) : "memory");
        arrive();
#endif
      }
      if (p >= H && p < NP - 1) {                  // always NI copies per step: the counted wait relies on it (beyond the
#pragma unroll                                     // last step they fill a buffer nobody reads); all issued AHEAD of the arrive
        for (int i = 0; i < NI; ++i)
          if (H + (i * (NP - 1 - H)) / NI == p && !(X3W_SKIP & 8)) stage_piece(simg, buf3, i);
      }
#pragma unroll
      for (int pp = 0; pp < 6; ++pp) {
#if X3W_SKIP & 2
        if (pp == 0) asm volatile("" : "+v"(acc[gstart(p)]), "+v"(acc[gstart(p) + (two ? 1 : 0)]) : "v"(T[0]), "v"(B[p & 1][0][0]));
        if (j < 0)
#endif
        acc[gstart(p)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8m, T[TA[pp]]), __builtin_bit_cast(bf8m, B[p & 1][0][TB[pp]]), acc[gstart(p)], 0, 0, 0);
#if X3W_FINE
        if (p == H) {
          __builtin_amdgcn_sched_barrier(0);
          if (!(X3W_SKIP & 32)) take_micro(raw[R1], s1, Tn, 2 * pp, redo, tks);
          __builtin_amdgcn_sched_barrier(0);
        }
#endif
#if X3W_SKIP & 2
        if (j < 0)
#endif
        if (two)
          acc[gstart(p) + (two ? 1 : 0)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8m, T[TA[pp]]), __builtin_bit_cast(bf8m, B[p & 1][1][TB[pp]]), acc[gstart(p) + (two ? 1 : 0)], 0, 0, 0);
        // The wavefront's own VALU work is PINNED between the MFMAs behind the barrier (the fences keep the compiler from
        // collecting it behind the last MFMA): the split of the next step's fragment (its loads landed before the counted
        // wait) behind pair H, the address arithmetic and the loads of step j+3's fragment behind the pair that follows.
        // A wavefront then keeps the matrix pipe busy on its own, whatever its SIMD neighbour is doing.
        if (p == H) {
          __builtin_amdgcn_sched_barrier(0);
#if X3W_FINE
          if (!(X3W_SKIP & 32)) take_micro(raw[R1], s1, Tn, 2 * pp + 1, redo, tks);
#else
          if (!(X3W_SKIP & 32)) take_chunk(raw[R1], s1, Tn, pp, redo);
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
        if (p == (H + 1 < NP ? H + 1 : H) && pp == (H + 1 < NP ? 0 : 5)) {
          __builtin_amdgcn_sched_barrier(0);
          // step j+4's fragment, into the slot the split above has just emptied: 2.25 steps ahead of the counted wait that
          // covers it (rows clamp to M-1 beyond the last tile: harmless)
          if (!(X3W_SKIP & 16)) load_a(raw[R1], t4, s4);
          adv(t3, s3); adv(t4, s4);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    X3W_STAMP(j, 4);
#ifdef PNA_X3W_TIMERS
    dbg_j = j;
#endif
    if (redo && !(X3W_SKIP & 32)) {
      int t1 = t, sx = s;
      adv(t1, sx);
      take_slow(t1, s1, Tn);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) T[i] = Tn[i];
    if (s == NS - 1) {
      __builtin_amdgcn_sched_barrier(0);
#if defined(X3W_SKIP) && (X3W_SKIP & 1)      // development: the main loop alone (tools/x3w_variants.sh); results are garbage
      {
        float sum = vacc;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int i = 0; i < 16; ++i) sum += acc[n][i];
        if (sum == 12345.678f) g.y[lane] = sum;
        zero_acc();
      }
#else
      epilogue(t);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    X3W_STAMP(j, 5);
    adv(t, s);
    buf = (buf + 1) & 3;
  };
  for (int j = 0; j < nsteps; j += 3) {
    step(std::integral_constant<int, 0>{}, j);
    if (j + 1 < nsteps) step(std::integral_constant<int, 1>{}, j + 1);
    if (j + 2 < nsteps) step(std::integral_constant<int, 2>{}, j + 2);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // the copies, re-loads and B reads issued for a step that does not exist
}

template <int S, int NTF, int NMIX, int NVC>
int launch_w(const WArgs& g, const Plan& p, int cus, hipStream_t st) {
  constexpr int NT = S * NTF + NMIX;
  const size_t lds = (size_t)kNBuf * NT * 3072 + (size_t)kWaves * kHalfRows * g.YP * 4 + (size_t)3 * g.YP * 4 + (size_t)kWaves * (S + 1) * 128 +
                     (NVC ? (size_t)g.NS * 64 : 0) + 16;
  if (lds > 160 * 1024) return -2;
  if (hipFuncSetAttribute((const void*)k_posttrans_x3w<S, NTF, NMIX, NVC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return -1;
  const int ntiles = (g.M + kTileRows - 1) / kTileRows;
  const int gx = (cus + p.ny - 1) / p.ny;
  const dim3 grid((unsigned)(ntiles < gx ? ntiles : gx), (unsigned)p.ny);
  hipLaunchKernelGGL((k_posttrans_x3w<S, NTF, NMIX, NVC>), grid, dim3(kThreads), lds, st, g);
  return 0;
}

// the instantiations that exist (everything else stays on pna_posttrans_x3_f32)
bool have_kernel(int S, const Plan& p) {
  if (!p.ok || S != 3) return false;
  if (p.ntf == 2 && p.nmix == 1 && p.nvc == 1) return true;     // N = 75
  if (p.ntf == 2 && p.nmix == 2 && p.nvc == 0) return true;     // N = 76 .. 80
  if (p.ntf == 2 && p.nmix == 1 && p.nvc == 0) return true;     // N = 65 .. 74
  if (p.ntf == 2 && p.nmix == 0 && p.nvc == 0) return true;     // N = 64, 128, 192, ...
  return false;
}

}  // namespace

extern "C" int pna_posttrans_x3w_supported(int32_t K, int32_t N, int32_t n_scaler, int32_t Kh) {
  if (K < 4 || K % 4 != 0 || Kh != 0) return 0;          // (K % 4: the in-stream tail handling of take_chunk)
  return have_kernel(n_scaler, make_plan(N, n_scaler)) ? 1 : 0;
}

extern "C" int64_t pna_posttrans_x3w_packed_bytes(int32_t K, int32_t N, int32_t n_scaler) {
  const Plan p = make_plan(N, n_scaler);
  if (!p.ok || K <= 0) return 0;
  const int64_t NS = steps_of(K);
  return (int64_t)p.ny * NS * p.nt * 3072 + (int64_t)p.ny * NS * 64;      // bf16 tile images | the VALU column's fp32 weights
}

extern "C" int pna_posttrans_x3w_pack_f32(const float* w_ref, int64_t ldw, int32_t N, int32_t K, int32_t n_scaler, void* w_img,
                                          pna_stream_t stream) {
  const Plan p = make_plan(N, n_scaler);
  if (!w_ref || !w_img || K <= 0 || K % 4 != 0 || !have_kernel(n_scaler, p) || ldw < (int64_t)n_scaler * K)
    return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3w_pack_f32: bad arguments / unsupported shape (see pna_posttrans_x3w_supported)");
  const int NS = steps_of(K);
  const int64_t nw = (int64_t)p.ny * NS * p.nt * 3072;
  const int64_t elems = nw / 2 + (int64_t)p.ny * NS * 16;
  const int blocks = (int)((elems + 255) / 256 > 4096 ? 4096 : (elems + 255) / 256);
  hipLaunchKernelGGL(k_pack_x3w, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_ref, (long)ldw, K, n_scaler, p, NS,
                     (unsigned short*)w_img, (float*)((unsigned char*)w_img + nw));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}

extern "C" int pna_posttrans_x3w_f32(const pna_posttrans_args* a, pna_stream_t stream) {
  if (!a) return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3w_f32: null args");
  if (a->M < 0 || a->K < 4 || a->N <= 0) return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3w_f32: bad M/K/N (K >= 4)");
  if (a->M == 0) return PNA_OK;
  if (!a->a || !a->w_img || !a->y) return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3w_f32: a/w_img/y must be non-null");
  if (a->h != nullptr || a->n_tower > 1)
    return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3w_f32: the h panel and n_tower > 1 are served by pna_posttrans_x3_f32");
  const Plan p = make_plan(a->N, a->n_scaler);
  if (!have_kernel(a->n_scaler, p) || a->K % 4 != 0) return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3w_f32: unsupported shape (see pna_posttrans_x3w_supported)");
  if (a->lda < a->K || a->ldy < a->N) return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3w_f32: leading dimensions too small");
  if (a->residual && a->ld_res < a->N) return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3w_f32: ld_res too small");
  if ((a->col_scale == nullptr) != (a->col_shift == nullptr))
    return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3w_f32: col_scale and col_shift come together");
  if (a->relu < 0 || a->relu > 2) return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3w_f32: relu must be 0, 1 or 2");
  WArgs g;
  memset(&g, 0, sizeof(g));
  const int NS = steps_of(a->K);
  g.a = a->a; g.w_img = (const unsigned char*)a->w_img;
  g.wv_img = (const float*)((const unsigned char*)a->w_img + (size_t)p.ny * NS * p.nt * 3072);
  g.bias = a->bias;
  for (int s = 0; s < a->n_scaler; ++s) g.row_scale[s] = a->row_scale[s];
  g.row_post = a->row_post; g.col_scale = a->col_scale; g.col_shift = a->col_shift; g.residual = a->residual;
  g.y = a->y; g.lda = a->lda; g.ldy = a->ldy; g.ld_res = a->ld_res;
  g.M = a->M; g.K = a->K; g.NB = p.NB; g.r = p.r; g.YP = (p.NB + 3) / 4 * 4; g.NS = NS; g.relu = a->relu;
  g.slope = a->relu == 2 ? a->act_slope : 0.f;
  // 16-byte accesses of the tail: y (and the residual) rows must be 16-byte aligned
  g.vec_ok = ((uintptr_t)a->y % 16 == 0 && a->ldy % 4 == 0) &&
             (!a->residual || ((uintptr_t)a->residual % 16 == 0 && a->ld_res % 4 == 0)) && (p.ny == 1 || p.NB % 4 == 0) && p.NB >= 4;
#ifdef PNA_X3W_TIMERS
  if (const char* e = getenv("PNA_X3W_DBG_PTR")) g.dbg = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    return pna_set_error(PNA_E_NODEVICE, "pna_posttrans_x3w_f32: cannot query the device");
  hipStream_t st = (hipStream_t)stream;
  int rc = -3;
  if (p.ntf == 2 && p.nmix == 1 && p.nvc == 1) rc = launch_w<3, 2, 1, 1>(g, p, cus, st);
  else if (p.ntf == 2 && p.nmix == 2 && p.nvc == 0) rc = launch_w<3, 2, 2, 0>(g, p, cus, st);
  else if (p.ntf == 2 && p.nmix == 1 && p.nvc == 0) rc = launch_w<3, 2, 1, 0>(g, p, cus, st);
  else if (p.ntf == 2 && p.nmix == 0 && p.nvc == 0) rc = launch_w<3, 2, 0, 0>(g, p, cus, st);
  if (rc != 0) return pna_set_error(PNA_E_LAUNCH, "pna_posttrans_x3w_f32: could not reserve LDS");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}
