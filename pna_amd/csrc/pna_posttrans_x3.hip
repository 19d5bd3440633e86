// pna_posttrans_x3.hip -- the posttrans contraction of pna_posttrans.hip on the bf16 matrix pipe, at fp32 accuracy.
// Implements pna_posttrans_x3_{packed_bytes,pack_f32,f32} of include/pna_amd.h.
//
//   y[v] = epilogue( bias + Wh . h[v] + sum_s scale_s[v] * (W_s . a[v]) )        (models/dgl/pna_layer.py:65-68,:206)
//
// The f32-input MFMA runs at the fp32 vector rate (157 TF/s): the contraction, not HBM, bounded the layer.  The
// bf16 pipe is 16x faster per instruction.  Every fp32 operand is cut EXACTLY into three bf16 terms by truncation,
//   x = x0 + x1 + x2,   x0 = top 16 bits of x,  x1 = top 16 bits of (x - x0),  x2 = x - x0 - x1   (8+8+8 mantissa bits)
// and the product a*b is evaluated as the six partial products of weight >= 2^-16,
//   a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0),
// each exact in fp32 (8 x 8 mantissa bits), accumulated in fp32 by v_mfma_f32_16x16x32_bf16.  The three dropped
// products (a1 b2, a2 b1, a2 b2) are below 2^-23 |a b|: the same order as ONE fp32 rounding of the product, far below
// the fp32 summation-order noise of a K = 900 contraction (tests/test_gpu_posttrans_x3.py measures both paths against
// float64).  6 instructions x 16 cycles per K = 32 vs 8 x 32 cycles on the f32 MFMA: 2.7x less matrix-pipe time.
// +-Inf operands (`a`, `h`, and weights at pack time) stay infinities (split8_inf) and NaN propagates, so the Inf/NaN
// pattern of the result is the fp32 contraction's (except Inf * Inf at the same k, which gives NaN); finite inputs never overflow (truncation, not rounding).  bf16-subnormal terms (|x| < 2^-126, and the residual terms of
// |x| < ~2^-110) may be flushed by the matrix pipe: an ABSOLUTE error of at most 2^-126 |w| per product, see the tests.
//
// Tiling: persistent workgroups (one per CU) of WAVES wavefronts; each wavefront owns RT row tiles of 16 rows x (NT x 16)
// columns x P panels (P = S scalers [+ the h panel]; 12 wavefronts x 1 row tile by default); K is consumed 32 at a time.
// The weight is pre-split and pre-packed (pna_posttrans_x3_pack_f32) into the exact LDS image of every chunk,
// [term][panel][lane group][80 cols][8 k] bf16, so that (i) the chunk is copied global -> LDS by
// global_load_lds_dwordx4 (no staging registers, no ds_write), double buffered, one barrier per chunk, and (ii) a B
// fragment is one conflict-free ds_read_b128.  The A fragment (lane (i, g): floats [k0+8g, k0+8g+8) of row i) is
// loaded one chunk ahead and split in registers once per chunk.  The kernel is bound by instruction issue (one
// instruction per 4 cycles per SIMD, MFMA and VALU together) and by its synchronized non-MFMA phases (epilogue, split,
// barrier skew), not by the matrix pipe: see DESIGN.md 4.2b for the phase-timer breakdown.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "pna_amd.h"
#include "pna_internal.h"
#include "pna_x3_split.h"

#ifndef X3_GRP_NBUF
#define X3_GRP_NBUF 3      // weight buffers of the one-block grouped kernel (development: 4 = fragments and images 3 steps ahead; measured equal)
#endif

namespace {

using namespace pna_x3;       // f4 / bf8 / u4 / f4u, split8 / split8_inf / absmax8 / top16 / pack_hi: the exact fp32 -> 3 x bf16 split (pna_x3_split.h)

constexpr int kKC = 32;                       // k values per chunk
constexpr int kDefaultNBuf = 3;              // weight buffers of the default pipeline (pna_posttrans_args.pipeline = 0)
// Output columns per workgroup (template parameter NW): 80 (5 column tiles: N <= 80, or N > 128 in 80-column blocks) or 128
// (8 tiles: 80 < N <= 128, e.g. BASELINE configs[4]'s F = 128 -- as two 80-column blocks it computed 10 tiles for 8 and read the
// aggregate twice).  The packed image and everything below is laid out per NW.
__host__ __device__ constexpr int nw_of(int N) { return (N > 80 && N <= 128) ? 128 : 80; }
__host__ __device__ constexpr int panel_bytes(int NW) { return 4 * NW * 8 * 2; }   // one (term, panel) image: [4 lane groups][NW][8] bf16

struct XArgs {
  const float* a; const void* w_img; const float* h; const void* wh_img; const float* bias;
  const float* row_scale[PNA_MAX_SCALER];
  const float* row_post; const float* col_scale; const float* col_shift; const float* residual;
  float* y;
  long lda, ldh, ldy, ld_res;
  int M, K, N, Kh, relu, grid_x;
  float slope;                 // LeakyReLU slope (relu == 2), 0 for ReLU
  unsigned long long* dbg;     // experiments build only: per-wavefront phase timers (null otherwise)
  // grouped mode (GRP kernels: pna_posttrans_args.row_perm): rows of `a` / row_scale / row_post are in a VIRTUAL order,
  // perm[virtual row] = row of y / residual (or -1: padding, nothing stored); tile t multiplies by weight image tile_image[t]
  const int* perm; const int* tile_image; long img_stride;
};

// ---- weight packing ------------------------------------------------------------------------------------------------
// w_img[ny][c][term][s][g][n][e]  = term(w_ref[ny*80 + n][Kh + s*K + c*32 + kperm(g, e)])   (0 outside K / N)
// wh_img[ny][c][term][g][n][e]    = term(w_ref[ny*80 + n][c*32 + kperm(g, e)])              (0 outside Kh / N)
// kperm(g, e) = 4g + e for e < 4, 16 + 4g + (e - 4) otherwise: element e of lane group g.  A and B agree on it, and it
// makes each of the two A loads of a chunk read 64 contiguous bytes per row (lane groups 0..3 x 16 B).
template <int kNW>
__global__ void k_pack_x3(const float* w_ref, long ldw, int N, int K, int S, int Kh, unsigned short* w_img, unsigned short* wh_img) {
  const int nca = (K + kKC - 1) / kKC, nch = (Kh + kKC - 1) / kKC, nty = (N + kNW - 1) / kNW;
  const long per = 4L * kNW * 8;
  const long total_w = (long)nty * nca * 3 * S * per, total_h = (long)nty * nch * 3 * per;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total_w + total_h; i += (long)gridDim.x * blockDim.x) {
    const bool is_h = i >= total_w;
    long r = is_h ? i - total_w : i;
    const int e = r % 8; r /= 8;
    const int n = r % kNW; r /= kNW;
    const int g = r % 4; r /= 4;
    int s = 0;
    if (!is_h) { s = r % S; r /= S; }
    const int term = r % 3; r /= 3;
    const int nc = is_h ? nch : nca;
    const int c = r % nc; r /= nc;
    const int ny = (int)r;
    const int col = ny * kNW + n, k = c * kKC + 8 * g + e;
    const int kmax = is_h ? Kh : K;
    float w = 0.f;
    if (col < N && k < kmax) w = w_ref[(long)col * ldw + (is_h ? 0 : Kh + (long)s * K) + k];
    const bool winf = __builtin_fabsf(w) == INFINITY;           // an infinite weight is carried by its lowest term alone (see split8_inf)
    const float wf = winf ? 0.f : w;
    const float r1 = wf - top16(wf), r2 = winf ? w : r1 - top16(r1);
    const float t = term == 0 ? wf : term == 1 ? r1 : r2;
    (is_h ? wh_img + (i - total_w) : w_img + i)[0] = (unsigned short)(fbits(t) >> 16);
  }
}

// GEN: the generic epilogue (any M; rows and columns predicated per element) -- used only for the < 16 rows a matrix has
// beyond a multiple of 16; otherwise the straight-line one, with wavefront tiles entirely past M skipped.  (Both in one
// kernel made the compiler drain vmcnt at the top of every chunk: its scoreboard merges the two paths conservatively.)
template <int S, bool HAS_H, int kNW, int NT, int RT, int WAVES, int NBUF, bool GEN, bool GRP = false>
__global__ __launch_bounds__(WAVES * 64, 1) void k_posttrans_x3(const XArgs g) {
  static_assert(!GRP || (NBUF >= 3 && !GEN && !HAS_H), "grouped mode: 3- or 4-buffer pipeline, straight-line epilogue, no h panel");
  static_assert(NBUF >= 2 && NBUF <= 4, "2, 3 or 4 weight buffers");
  constexpr int kPanelB = panel_bytes(kNW), kPanelV = kPanelB / 16;
  constexpr int kThreads = WAVES * 64;
  constexpr int kChunkV = 3 * S * kPanelV;     // 16-byte pieces of an aggregate chunk image (the h chunk is 3 * kPanelV)
  constexpr int kTileRows = WAVES * 16 * RT;   // rows per workgroup tile
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 buffers x kChunkV x 16 B | column constants
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform, in an SGPR
  const int li = lane & 15, lg = lane >> 4;
  const int n0 = blockIdx.y * kNW;
  const int ntiles = (g.M + kTileRows - 1) / kTileRows;

  // One accumulator set per scaler block (and one for the h block): the per-row scalers are applied in the epilogue,
  // so the A operand is split ONCE per chunk -- the kernel is bound by instruction issue (one instruction per 4 cycles
  // per SIMD, shared by MFMA and VALU), and re-scaling + re-splitting A per scaler tripled its VALU count.
  constexpr int P = S + (HAS_H ? 1 : 0);
  f4 acc[RT][P][NT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[r][p][n] = (f4){0.f, 0.f, 0.f, 0.f};

  const int nca = (g.K + kKC - 1) / kKC;
  const int nch = HAS_H ? (g.Kh + kKC - 1) / kKC : 0;
  const int nc = nca + nch;
  const unsigned char* img_a = reinterpret_cast<const unsigned char*>(g.w_img) + (size_t)blockIdx.y * nca * kChunkV * 16;
  const unsigned char* img_h = reinterpret_cast<const unsigned char*>(HAS_H ? g.wh_img : g.w_img) + (size_t)blockIdx.y * nch * 3 * kPanelV * 16;

  // chunk image -> LDS buffer, asynchronously; every wavefront copies whole 1 KB pieces (the piece counts are
  // multiples of 64), destination = wave-uniform base + lane * 16
  constexpr int NI = (kChunkV + kThreads - 1) / kThreads;   // global_load_lds instructions per wavefront per chunk
  auto stage_piece = [&](int c, int buf, int i, long ib = 0) __attribute__((always_inline)) {   // ib: byte offset of the tile's image (GRP)
    const unsigned char* src = img_a + ib + (size_t)c * kChunkV * 16;
    int pieces = kChunkV;
    if constexpr (HAS_H) {
      if (c >= nca) { src = img_h + (size_t)(c - nca) * 3 * kPanelV * 16; pieces = 3 * kPanelV; }
    }
    unsigned char* dst = lds + (size_t)buf * kChunkV * 16;
    int w0 = (i * WAVES + wave) * 64;                    // first piece of this wavefront (wave-uniform)
    if constexpr (NBUF >= 3) {
      // the 3- / 4-buffer pipeline waits with COUNTED vmcnt: every wavefront issues exactly NI copies per chunk; a slot past
      // the image re-copies an earlier piece (same bytes to the same LDS address: harmless)
      if (w0 >= pieces) w0 = w0 % pieces;
    } else {
      if (w0 >= pieces) return;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(w0 + lane) * 16),
                                     (__attribute__((address_space(3))) void*)(dst + (size_t)w0 * 16), 16, 0, 0);
  };
  auto stage = [&](int c, int buf, long ib = 0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NI; ++i) stage_piece(c, buf, i, ib);
  };

  // A operand of (tile t, chunk c), row tile r: floats [k, k+4) and [k+16, k+20) of the lane's row, k = 32c + 4 lg, as two
  // clamped 16-byte windows ([kk, kk+4), kk = min(k, kmax-4); the launcher guarantees kmax >= 4), fixed up after the load.
  auto a_src = [&](int c, const float*& src, long& ld, int& kmax, int& k) __attribute__((always_inline)) {
    src = g.a; ld = g.lda; kmax = g.K; k = c * kKC + 8 * lg;
    if constexpr (HAS_H) {
      if (c >= nca) { src = g.h; ld = g.ldh; kmax = g.Kh; k = (c - nca) * kKC + 8 * lg; }
    }
  };
  auto fix4 = [&](int k, int kmax, f4 t) -> f4 __attribute__((always_inline)) {        // window [kk,kk+4) -> elements [k,k+4), 0 beyond kmax
    const int d = k - max(0, min(k, kmax - 4));
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
  f4 nxt[NBUF == 4 ? 3 : 2][RT][2];            // A fragments in flight: one set (2-buffer pipeline), NBUF - 1 (3 / 4 buffers: that many chunks ahead)
  bf8 A[3][RT];                                // the three bf16 terms of the chunk being multiplied
  // Issued by hand (hipcc would hoist plain loads into one burst at the top of the interval, and a burst of scattered-row
  // loads from all wavefronts stalls them at the TA).  Rules for a register written by an asm load: it is written on EVERY
  // path (the caller passes a valid tile even when there is no next one), no control flow lies between the load and the
  // wait that covers it (a register with two reaching definitions may be copied by the compiler at the merge, i.e. while
  // the load is still in flight), and it is read only after the "+v" anchor that follows the wait in take().
  typedef f4 aset_t[RT][2];
  auto load_a_piece = [&](aset_t& dst, int t, int c, int r, int w) __attribute__((always_inline)) {
    const float* src; long ld; int kmax, k;
    a_src(c, src, ld, kmax, k);
#if defined(X3_DEV_A_FROM_L2)     // development: every tile reads the first tile's rows (how much of the kernel is A latency?)
    const int row = min((0 * t + wave) * (16 * RT) + 16 * r + li, g.M - 1);
#elif defined(X3_DEV_A_WRAP_ROWS) // development: the A rows wrap around a window that fits the Infinity Cache
    const int row = min(((t * WAVES + wave) * (16 * RT) + 16 * r + li) % (X3_DEV_A_WRAP_ROWS), g.M - 1);
#else
    const int row = min((t * WAVES + wave) * (16 * RT) + 16 * r + li, g.M - 1);
#endif
    const int kk = max(0, min(k + 4 * w, kmax - 4));
    const float* ptr = src + (size_t)row * ld + kk;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[r][w]) : "v"(ptr) : "memory");
  };
  auto load_a = [&](aset_t& dst, int t, int c) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int w = 0; w < 2; ++w) load_a_piece(dst, t, c, r, w);
  };
  // next -> current: fix the windows, split.  wait: 0 = vmcnt(0) first, 1 = vmcnt(NI) first (the NI weight copies issued AFTER
  // these A loads may stay in flight: VMEM returns in order), 2 = none (an earlier counted wait already covered the loads).
  // The waits carry NO register operands; the registers are tied by ONE anchor after them.  (Two "+v" waits in two branches
  // made the compiler copy the fragment into fresh registers at the top of one branch -- BEFORE that branch's wait, while the
  // loads were still in flight: stale A fragments on short K.)
  auto take = [&](aset_t& cur, int c, int wait) __attribute__((always_inline)) {
    const float* src; long ld; int kmax, k;
    a_src(c, src, ld, kmax, k);
    const bool tail = c * kKC + kKC > (c < nca ? g.K : nca * kKC + g.Kh);   // wave-uniform: only a row's last chunk needs fixing
    if (wait == 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NI) : "memory");
    else if (wait == 0) asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
#pragma unroll
    for (int r = 0; r < RT; ++r) asm volatile("" : "+v"(cur[r][0]), "+v"(cur[r][1]) : : "memory");
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      f4 lo4 = cur[r][0], hi4 = cur[r][1];
      if (tail) { lo4 = fix4(k, kmax, lo4); hi4 = fix4(k + 4, kmax, hi4); }
      // +-Inf anywhere in the wavefront's fragment (5 VALU to find out): the slower split that keeps it an infinity
      if (__builtin_amdgcn_ballot_w64(absmax8(lo4, hi4) == INFINITY) != 0) split8_inf(lo4, hi4, A[0][r], A[1][r], A[2][r]);
      else split8(lo4, hi4, A[0][r], A[1][r], A[2][r]);
    }
  };

  // Epilogue, from the MFMA's C/D layout (col = lane & 15, row = (lane >> 4) * 4 + reg): the scalers of a lane's four
  // adjacent rows come as one 16-byte load each, the column constants from LDS; every load is unconditional (clamped
  // row / column) and issued before its first use, only the store is predicated.  (Transposing the tile through LDS
  // to store 16 bytes per lane was measured: no faster -- the epilogue is three dependent memory round trips, not
  // instruction count -- and 16-byte accesses to rows of N = 75 floats are unaligned for 3 rows in 4, which is slower.)
  float* const colc = reinterpret_cast<float*>(lds + (size_t)NBUF * kChunkV * 16);         // [3][80]: bias | scale | shift
  for (int i = tid; i < kNW; i += kThreads) {
    const int col = n0 + i;
    colc[i] = (g.bias && col < g.N) ? g.bias[col] : 0.f;
    colc[kNW + i] = (g.col_scale && col < g.N) ? g.col_scale[col] : 1.f;
    colc[2 * kNW + i] = (g.col_shift && col < g.N) ? g.col_shift[col] : 0.f;
  }
  auto scalers_cd = [&](int rb, f4 (&scv)[S]) {  // the lane's 4 adjacent rows: one 16-byte load per scaler
#pragma unroll
    for (int s = 0; s < S; ++s) {
      scv[s] = (f4){1.f, 1.f, 1.f, 1.f};
      if (g.row_scale[s]) {
        if (rb + 3 < g.M) scv[s] = reinterpret_cast<const f4u*>(g.row_scale[s] + rb)->v;
        else
          for (int r = 0; r < 4; ++r) scv[s][r] = g.row_scale[s][min(rb + r, g.M - 1)];
      }
    }
  };
  auto finish = [&](float v, float rp, float cb, float cs, float ct) -> float __attribute__((always_inline)) {
    v = v + cb;
    if (g.row_post) v = v * rp;                                        // graph-norm (pna_layer.py:71-72)
    if (g.col_scale) v = v * cs + ct;                                  // eval-mode BatchNorm folded to an affine map
    if (g.relu) v = v > 0.f ? v : (v != v ? v : (g.relu == 2 ? g.slope * v : 0.f));   // ReLU / LeakyReLU (keep NaN)
    return v;
  };
  auto epilogue = [&](int t) __attribute__((always_inline)) {
    const int row0 = (t * WAVES + wave) * (16 * RT);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      f4 scv[S];
      scalers_cd(row0 + 16 * rt + 4 * lg, scv);
      f4 rpv = (f4){1.f, 1.f, 1.f, 1.f};       // graph-norm factors of the lane's 4 rows: one load, not one per element
      if (g.row_post) {
        const int rb = row0 + 16 * rt + 4 * lg;
        if (rb + 3 < g.M) rpv = reinterpret_cast<const f4u*>(g.row_post + rb)->v;
        else
          for (int r = 0; r < 4; ++r) rpv[r] = g.row_post[min(rb + r, g.M - 1)];
      }
      float res[NT][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = min(row0 + 16 * rt + lg * 4 + r, g.M - 1);
#pragma unroll
        for (int n = 0; n < NT; ++n)
          res[n][r] = g.residual ? g.residual[(size_t)row * g.ld_res + min(n0 + n * 16 + li, g.N - 1)] : 0.f;
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int cl = n * 16 + li, col = n0 + cl;
        const float cb = colc[cl], cs = colc[kNW + cl], ct = colc[2 * kNW + cl];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row0 + 16 * rt + lg * 4 + r;
          float v = HAS_H ? acc[rt][P - 1][n][r] : 0.f;
#pragma unroll
          for (int s = 0; s < S; ++s) v = v + scv[s][r] * acc[rt][s][n][r];
          v = finish(v, rpv[r], cb, cs, ct);
          if (g.residual) v = res[n][r] + v;                               // h_in + h (pna_layer.py:212-213)
          if (row < g.M && col < g.N) g.y[(size_t)row * g.ldy + col] = v;
#pragma unroll
          for (int p = 0; p < P; ++p) acc[rt][p][n][r] = 0.f;
        }
      }
    }
  };
  // The same epilogue for a wavefront whose 16*RT rows all exist (every tile but the matrix's last): straight-line.
  // The generic epilogue above is ~1100 instructions per row tile (a uniform branch per optional operand per element,
  // 64-bit address arithmetic per element, row/column predicates around every store) and the kernel is bound by
  // instruction issue while all wavefronts sit in it together -- it cost 15 k of the 74 k cycles of a tile period.
  // Here absent operands are neutral constants (x*1, +0 are exact), ReLU is `v < lo ? 0 : v` with lo = 0 or -inf (keeps
  // NaN like the generic form), every load is issued before the first use, rows need no predicate and only a column
  // tile that crosses N predicates its stores.
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int p = 0; p < P; ++p)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[r][p][n] = (f4){0.f, 0.f, 0.f, 0.f};
  };
  auto epilogue_full = [&](int t) __attribute__((always_inline)) {
    // Addresses are (wave-uniform 64-bit base in SGPRs) + (32-bit lane offset): one VALU per address instead of a
    // 64-bit multiply-add chain, and nothing loop-invariant for the compiler to hoist into the MFMA loop's registers
    // (the lane ids go through an empty asm so that values derived from them are rebuilt here).
    int li_ = li, lg_ = lg;
    asm volatile("" : "+v"(li_), "+v"(lg_));
    const int row0 = (t * WAVES + wave) * (16 * RT);
    const float lo = g.relu ? 0.f : -INFINITY;
    const bool leaky = g.relu == 2;
    const f4 ones = (f4){1.f, 1.f, 1.f, 1.f};
    const unsigned ldyb = (unsigned)g.ldy * 4u, ldrb = (unsigned)g.ld_res * 4u;       // (launcher: pitches < 2^24 floats)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const unsigned rl = 16u * rt + 4u * (unsigned)lg_;                              // first of the lane's 4 rows, tile-local
      f4 scv[S];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        scv[s] = ones;
        if (g.row_scale[s])
          scv[s] = reinterpret_cast<const f4u*>(reinterpret_cast<const char*>(g.row_scale[s] + row0) + rl * 4u)->v;
      }
      f4 rpv = ones;
      if (g.row_post) rpv = reinterpret_cast<const f4u*>(reinterpret_cast<const char*>(g.row_post + row0) + rl * 4u)->v;
      float res[NT][4];
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) res[n][r] = 0.f;
      // GRP: the lane's four rows of y / residual are perm[row0 + rl .. + 3] (one 16-byte load; -1 = padding: loads re-read
      // row 0, stores are skipped)
      typedef int i4 __attribute__((ext_vector_type(4)));
      i4 pr = (i4){0, 0, 0, 0};
      if constexpr (GRP) pr = *reinterpret_cast<const i4*>(g.perm + row0 + (int)rl);
      if (g.residual) {
        const char* rbase = reinterpret_cast<const char*>(g.residual + (GRP ? (size_t)0 : (size_t)row0 * g.ld_res));
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const unsigned cc = (unsigned)min(n0 + n * 16 + li_, g.N - 1);
          const unsigned vo = (GRP ? 0u : rl * ldrb) + cc * 4u;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            res[n][r] = *reinterpret_cast<const float*>(rbase + (GRP ? (size_t)(unsigned)max(pr[r], 0) * ldrb : (size_t)r * ldrb) + vo);
        }
      }
      // Every load above is waited for HERE, once, ahead of the first (predicated) store: otherwise the compiler
      // sinks the wait into the first predicated block and has to repeat it -- as vmcnt(0), which then also waits for
      // the stores issued meanwhile -- at every control-flow merge that follows.
#pragma unroll
      for (int s = 0; s < S; ++s) asm volatile("" : "+v"(scv[s]));
      asm volatile("" : "+v"(rpv));
#pragma unroll
      for (int n = 0; n < NT; ++n) asm volatile("" : "+v"(res[n][0]), "+v"(res[n][1]), "+v"(res[n][2]), "+v"(res[n][3]));
      asm volatile("" : "+v"(pr));
      char* const ybase = reinterpret_cast<char*>(g.y + (GRP ? (size_t)0 : (size_t)row0 * g.ldy));
      const unsigned yo = (GRP ? 0u : rl * ldyb) + (unsigned)(n0 + li_) * 4u;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int cl = n * 16 + li_;
        const float cb = colc[cl], cs = colc[kNW + cl], ct = colc[2 * kNW + cl];
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = HAS_H ? __builtin_fmaf(scv[0][r], acc[rt][0][n][r], acc[rt][P - 1][n][r]) : scv[0][r] * acc[rt][0][n][r];
#pragma unroll
          for (int s = 1; s < S; ++s) x = __builtin_fmaf(scv[s][r], acc[rt][s][n][r], x);
          x = (x + cb) * rpv[r];
          x = __builtin_fmaf(x, cs, ct);
          x = x < lo ? (leaky ? x * g.slope : 0.f) : x;  // ReLU / LeakyReLU / none (lo = -inf); NaN < lo is false: NaN is kept
          v[r] = res[n][r] + x;
#pragma unroll
          for (int p = 0; p < P; ++p) acc[rt][p][n][r] = 0.f;
        }
        if constexpr (GRP) {
          if (n0 + cl < g.N) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (pr[r] >= 0) *reinterpret_cast<float*>(ybase + (size_t)(unsigned)pr[r] * ldyb + (yo + n * 64u)) = v[r];
          }
        } else if (n0 + (n + 1) * 16 <= g.N) {                 // wave-uniform: the whole column tile exists
#pragma unroll
          for (int r = 0; r < 4; ++r) *reinterpret_cast<float*>(ybase + (size_t)r * ldyb + (yo + n * 64u)) = v[r];
        } else if (n0 + cl < g.N) {
#pragma unroll
          for (int r = 0; r < 4; ++r) *reinterpret_cast<float*>(ybase + (size_t)r * ldyb + (yo + n * 64u)) = v[r];
        }
      }
    }
  };
  if constexpr (NBUF == 2) {
    // Persistent workgroup: tiles t = blockIdx.x, blockIdx.x + gridDim.x, ...; the (tile, chunk) pairs form ONE software
    // pipeline, so the first chunk of the next tile is already in flight while this tile's epilogue runs.
    int t = blockIdx.x, c = 0, buf = 0;
    if (t >= ntiles) return;
    stage(0, 0);
    load_a(nxt[0], t, 0);
    take(nxt[0], 0, 0);                           // (waits vmcnt(0): chunk 0 of the image has landed too)
    __syncthreads();
    while (true) {
      int tn = t, cn = c + 1;
      if (cn == nc) { tn = t + gridDim.x; cn = 0; }
      const bool more = tn < ntiles;
      const int ta = more ? tn : t, ca = more ? cn : c;   // A prefetch target (a harmless re-load when there is no next chunk)
      const bool is_h = HAS_H && c >= nca;
      const unsigned baddr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + (unsigned)(buf * kChunkV + lg * kNW + li) * 16u;   // LDS byte address of this lane's fragment column
      // The chunk's S*NT (panel, column tile) groups form one software pipeline: while the 6*RT MFMAs of group g run,
      // the three B fragments of group g+1 are read from LDS and one piece of the NEXT chunk's weight image is sent on
      // its way to LDS -- spread over the groups, because a burst of VMEM issue from all wavefronts at once stalls
      // every one of them at the TA.
      // B fragment (term, panel p, column tile n): piece ((term*np + p)*4 + lg)*80 + n*16 + li;
      // MFMAs ordered smallest partial products first, row tiles alternating (no MFMA waits for its predecessor).
      auto run = [&](auto npanel_c, auto p0_c, unsigned ba0) __attribute__((always_inline)) {
        constexpr int p0 = decltype(p0_c)::value;
        constexpr int NPN = decltype(npanel_c)::value;   // panels in this chunk's image; they accumulate into acc[.][p0 + p]
        constexpr int NG = NPN * NT;
        bf8 B[2][3];
        // B fragments are read by hand (inline asm + counted lgkmcnt): hipcc sinks its own ds_reads next to their use
        // and waits lgkmcnt(0), exposing one LDS round trip per group.  LDS returns in order, so with the three reads
        // of group g+1 issued behind those of group g, `lgkmcnt(3)` means group g has landed.
        auto load_b = [&](unsigned ba, int gi, int slot) __attribute__((always_inline)) {
          const int p = gi / NT, n = gi % NT;
  #pragma unroll
          for (int tm = 0; tm < 3; ++tm)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(B[slot][tm]) : "v"(ba + (unsigned)(tm * NPN * 4 * kNW * 16)), "n"(((p * 4) * kNW + n * 16) * 16));
        };
        load_b(ba0, 0, 0);
        constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
  #pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
          const int p = gi / NT, n = gi % NT;
          if (gi + 1 < NG) {
            load_b(ba0, gi + 1, (gi + 1) & 1);
            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(B[gi & 1][0]), "+v"(B[gi & 1][1]), "+v"(B[gi & 1][2]));
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(B[gi & 1][0]), "+v"(B[gi & 1][1]), "+v"(B[gi & 1][2]));
          }
            if (more) {                                                 // the L2-resident weight image: spread over the rest
  #pragma unroll
            for (int i = 0; i < NI; ++i)
              if ((NG > 2 * RT ? 2 * RT + (i * (NG - 2 * RT)) / NI : NG - 1) == gi) stage_piece(cn, buf ^ 1, i);
          }
  #pragma unroll
          for (int pp = 0; pp < 6; ++pp)
  #pragma unroll
            for (int r = 0; r < RT; ++r)
              acc[r][p0 + p][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[TA[pp]][r], B[gi & 1][TB[pp]], acc[r][p0 + p][n], 0, 0, 0);
        }
      };
      load_a(nxt[0], ta, ca);                            // HBM latency: first (and outside the is_h branches, see load_a_piece)
      if (!is_h) {
        run(std::integral_constant<int, S>{}, std::integral_constant<int, 0>{}, baddr);
      } else if (HAS_H) {
        run(std::integral_constant<int, 1>{}, std::integral_constant<int, P - 1>{}, baddr);
      }
      if (c == nc - 1) {
        __builtin_amdgcn_sched_barrier(0);         // keep the epilogue's loads out of the MFMA stream (register pressure)
        if constexpr (GEN) epilogue(t);
        else if ((t * WAVES + wave) * (16 * RT) < g.M) epilogue_full(t);          // (launcher: M is a multiple of 16 * RT)
        else zero_acc();
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!more) break;
      take(nxt[0], cn, 0);                               // waits vmcnt(0): this wavefront's share of the next chunk is in LDS
      __syncthreads();                                   // ... everyone's is, and everyone is done reading this chunk
      buf ^= 1; t = tn; c = cn;
    }
  } else {
    // ---- 3 weight buffers, ONE barrier per chunk placed in the MIDDLE of the chunk's MFMA stream -------------------------
    // With the barrier at the chunk boundary every wavefront drains its loads, splits its next A fragment and waits for the
    // slowest of the 12 at the same moment -- the matrix pipe idles through all of it, ten times per tile.  Here step k
    // (tile, chunk) reads buffer k % 3; barrier B_k sits between group H and H+1 of step k.  At B_k every wavefront has
    // finished step k-1, so buffer (k+2) % 3 = (k-1) % 3 is free: the copies of step k+2's image are issued AFTER B_k and
    // waited for (counted vmcnt, they are older than the A loads of the following step) BEFORE B_{k+1}; step k+2 starts
    // after B_{k+1}.  At a chunk boundary a wavefront only waits for ITS OWN next A fragment and splits it while the other
    // wavefronts of its SIMD keep the matrix pipe busy: wavefronts are synchronised mid-chunk only.
    // The A fragment is fetched TWO chunks ahead into two alternating register sets: the loads of step k+2 are issued at
    // the top of step k (into the set that held step k's fragment), so the counted wait before B_k -- which lets only
    // those youngest 2*RT loads stay in flight -- has already covered step k+1's fragment (issued a whole chunk earlier):
    // a wavefront never waits for HBM at a chunk boundary, and the latency it tolerates is 1.5 chunk times.
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nsteps = my_tiles * nc;
    int t = blockIdx.x, c = 0, buf = 0;
    if (t >= ntiles) return;
    // GRP: the weight image of a tile is tile_image[tile]; the ids of this workgroup's tiles sit in an LDS table that the
    // step loop reads through inline asm when the tile of step k+2 changes (an LDS read the compiler can see would make it
    // drain vmcnt(0) first: the LDS-DMA copies in flight might alias it)
    constexpr int D = NBUF - 1;                  // pipeline depth: fragments and images are fetched D steps ahead
    int* const timg = reinterpret_cast<int*>(colc + 3 * kNW);
    long ib2 = 0;                                // byte offset of the image of step k+D's tile
    int ti = 0, ti2_seen = -1;                   // local index of step k's tile | of the tile ib2 belongs to
    if constexpr (GRP) {
      for (int i = tid; i < my_tiles; i += kThreads) timg[i] = g.tile_image ? g.tile_image[blockIdx.x + i * (int)gridDim.x] : 0;
    }
    // steps 0 .. D-1: their images into buffers 0 .. D-1; the fragments of steps 0 .. D-2 before the vmcnt(0), step D-1's after
    // it (a step that does not exist re-loads step 0's fragment: harmless)
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (d < nsteps) {
        const int tl = d / nc;
        long ib = 0;
        if constexpr (GRP) ib = g.tile_image ? (long)g.tile_image[(int)blockIdx.x + tl * (int)gridDim.x] * g.img_stride : 0;
        stage(d % nc, d, ib);
      }
    }
    auto load_step = [&](aset_t& dst, int d) __attribute__((always_inline)) {
      const bool ex = d < nsteps;
      load_a(dst, ex ? t + (d / nc) * (int)gridDim.x : t, ex ? d % nc : 0);
    };
#pragma unroll
    for (int d = 0; d + 1 < D; ++d) load_step(nxt[d], d);
    take(nxt[0], 0, 0);                         // vmcnt(0): the images and the first D-1 fragments have landed
    load_step(nxt[D - 1], D - 1);
    __syncthreads();
#ifdef PNA_AMD_EXPERIMENTS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // phase timers, tools/x3_timers.py [removed in round 5: git history]
#endif
    auto step = [&](auto par_c, int k) __attribute__((always_inline)) {
      constexpr int PAR = decltype(par_c)::value;           // k % D
      aset_t& mine = nxt[PAR];              // free now (held step k's fragment): receives step k+D's
      aset_t& other = nxt[(PAR + 1) % D];   // holds step k+1's
      int tn = t, cn = c + 1;
      if (cn == nc) { tn = t + gridDim.x; cn = 0; }
      int t2 = tn, c2 = cn, wraps = cn == 0 ? 1 : 0;
#pragma unroll
      for (int d = 1; d < D; ++d) {                         // (tile, chunk) of step k+D
        if (++c2 == nc) { t2 += gridDim.x; c2 = 0; ++wraps; }
      }
      const bool more1 = k + 1 < nsteps, more2 = k + D < nsteps;
      if constexpr (GRP) {
        const int ti2 = min(ti + wraps, my_tiles - 1);
        if (ti2 != ti2_seen) {                             // (wave-uniform; once per tile)
          int v;
          asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v)
                       : "v"((unsigned)(size_t)(__attribute__((address_space(3))) int*)(timg + ti2)) : "memory");
          ib2 = (long)v * g.img_stride;
          ti2_seen = ti2;
        }
      }
      const int buf2 = buf == 0 ? NBUF - 1 : buf - 1;      // (k + D) % NBUF
      const bool is_h = HAS_H && c >= nca;
      const unsigned baddr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + (unsigned)(buf * kChunkV + lg * kNW + li) * 16u;
      load_a(mine, more2 ? t2 : t, more2 ? c2 : c);       // (a harmless re-load of this step's fragment near the end)
#ifdef PNA_AMD_EXPERIMENTS
      const unsigned long long tm0 = clock64();
#endif
#ifdef PNA_AMD_EXPERIMENTS
      unsigned long long tm1 = 0, tm1b = 0, tm2 = 0;
#endif
      auto run3 = [&](auto npanel_c, auto p0_c, unsigned ba0) __attribute__((always_inline)) {
        constexpr int p0 = decltype(p0_c)::value;       // compile-time: a run-time panel index puts the accumulators in scratch
        constexpr int NPN = decltype(npanel_c)::value;
        constexpr int NG = NPN * NT;
        constexpr int H = (NG - 1) / 2;                    // the barrier follows the B prefetch of group H
        constexpr int kPend = (D - 1) * 2 * RT + (D - 2) * NI;   // VMEM operations that may stay in flight across the barrier
        bf8 B[2][3];
        auto load_b = [&](unsigned ba, int gi, int slot) __attribute__((always_inline)) {
          const int p = gi / NT, n = gi % NT;
#pragma unroll
          for (int tm = 0; tm < 3; ++tm)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(B[slot][tm]) : "v"(ba + (unsigned)(tm * NPN * 4 * kNW * 16)), "n"(((p * 4) * kNW + n * 16) * 16));
        };
        load_b(ba0, 0, 0);
        constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
          const int p = gi / NT, n = gi % NT;
          if (gi + 1 < NG) {
            load_b(ba0, gi + 1, (gi + 1) & 1);
            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(B[gi & 1][0]), "+v"(B[gi & 1][1]), "+v"(B[gi & 1][2]));
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(B[gi & 1][0]), "+v"(B[gi & 1][1]), "+v"(B[gi & 1][2]));
          }
          if (gi == H) {
            // D = 2: everything older than the 2*RT A loads issued at the top of this step has landed: this wavefront's
            // copies of step k+1's image AND step k+1's A fragment.  D = 3: the loads of this step's top, the NI copies of
            // step k-1 and the loads of step k-1's top may stay in flight (issued in that order, returned in order): what is
            // older -- the image copied during step k-2 (step k+1's) and the fragment loaded at step k-2's top (step k+1's)
            // -- has landed.  The counts are exact: with D = 3 every step issues its NI copies, needed or not.
#ifdef PNA_AMD_EXPERIMENTS
            tm1 = clock64();
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(kPend) : "memory");
            tm1b = clock64();
            asm volatile("s_barrier" : : : "memory");
            tm2 = clock64();
#else
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" : : "n"(kPend) : "memory");
#endif
          }
          if (more2 || D > 2) {                           // (D = 3, no step k+D: the step's own image again, into a free buffer)
#pragma unroll
            for (int i = 0; i < NI; ++i)
              if (gi >= H && (NG - 1 > H ? H + 1 + (i * (NG - 1 - H)) / NI : H) == gi) stage_piece(more2 ? c2 : c, buf2, i, GRP ? ib2 : 0);
          }
#pragma unroll
          for (int pp = 0; pp < 6; ++pp)
#pragma unroll
            for (int r = 0; r < RT; ++r)
              acc[r][p0 + p][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[TA[pp]][r], B[gi & 1][TB[pp]], acc[r][p0 + p][n], 0, 0, 0);
        }
      };
      if (!is_h) {
        run3(std::integral_constant<int, S>{}, std::integral_constant<int, 0>{}, baddr);
      } else if (HAS_H) {
        run3(std::integral_constant<int, 1>{}, std::integral_constant<int, P - 1>{}, baddr);
      }
#ifdef PNA_AMD_EXPERIMENTS
      const unsigned long long tm3 = clock64();
#endif
      // the next fragment first, then the epilogue (the fragment landed before this step's barrier: no wait here)
      if (more1) take(other, cn, 2);
#ifdef PNA_AMD_EXPERIMENTS
      const unsigned long long tm4 = clock64();
#endif
      if (c == nc - 1) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (GEN) epilogue(t);
        else if ((t * WAVES + wave) * (16 * RT) < g.M) epilogue_full(t);        // (launcher: M is a multiple of 16 * RT)
        else zero_acc();
        __builtin_amdgcn_sched_barrier(0);
      }
#ifdef PNA_AMD_EXPERIMENTS
      if (g.dbg) {
        const unsigned long long tm5 = clock64();
        tacc[0] += tm1 - tm0; tacc[1] += tm1b - tm1; tacc[2] += tm2 - tm1b; tacc[3] += tm3 - tm2; tacc[4] += tm4 - tm3; tacc[5] += tm5 - tm4;
        tacc[6] += 1;
      }
#endif
      if (GRP && cn == 0) ++ti;
      t = tn; c = cn; buf = buf == NBUF - 1 ? 0 : buf + 1;
    };
#ifdef PNA_AMD_EXPERIMENTS
    const unsigned long long tstart = clock64();
#endif
    for (int k = 0; k < nsteps; k += D) {
      step(std::integral_constant<int, 0>{}, k);
      if (k + 1 < nsteps) step(std::integral_constant<int, 1>{}, k + 1);
      if constexpr (D > 2) {
        if (k + 2 < nsteps) step(std::integral_constant<int, 2>{}, k + 2);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the re-loads of the last steps
#ifdef PNA_AMD_EXPERIMENTS
    if (g.dbg && lane == 0) {
      tacc[7] = clock64() - tstart;
      unsigned long long* o = g.dbg + ((size_t)blockIdx.x * WAVES + wave) * 8;
      for (int i = 0; i < 8; ++i) o[i] = tacc[i];
    }
#endif
  }
}

template <int S, bool HAS_H, int kNW, int NT, int RT, int WAVES, int NBUF, bool GEN>
int launch_v(const XArgs& g, hipStream_t st) {
  const size_t lds = (size_t)NBUF * 3 * S * panel_bytes(kNW) + (size_t)(3 * kNW) * sizeof(float);
  if (hipFuncSetAttribute((const void*)k_posttrans_x3<S, HAS_H, kNW, NT, RT, WAVES, NBUF, GEN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return -1;
  const int ntiles = (g.M + WAVES * 16 * RT - 1) / (WAVES * 16 * RT);
  const dim3 grid((unsigned)(ntiles < g.grid_x ? ntiles : g.grid_x), (unsigned)((g.N + kNW - 1) / kNW));
  hipLaunchKernelGGL((k_posttrans_x3<S, HAS_H, kNW, NT, RT, WAVES, NBUF, GEN>), grid, dim3(WAVES * 64), lds, st, g);
  return 0;
}

// grouped mode (XArgs.perm): M is a multiple of the workgroup tile, N in (64, 80], no h panel
template <int S, int RT, int WAVES, int NBUF = 3, int kNW = 80, int NT = 5>
int launch_grouped(const XArgs& g, hipStream_t st) {
  const int ntiles = g.M / (WAVES * 16 * RT);
  // 8-wavefront workgroups of the 80-column block (117 registers, 50 KB of LDS): two per CU; the 128-column block (79 KB, > 128
  // registers) and the 12-wavefront three-block kernel: one
  const int wgs = (WAVES <= 8 && kNW == 80) ? 2 * g.grid_x : g.grid_x;
  const int gx = ntiles < wgs ? ntiles : wgs;
  const int per_wg = (ntiles + gx - 1) / gx;
  const size_t lds = (size_t)NBUF * 3 * S * panel_bytes(kNW) + (size_t)(3 * kNW) * sizeof(float) + (size_t)per_wg * sizeof(int);
  if (lds > 160 * 1024 ||
      hipFuncSetAttribute((const void*)k_posttrans_x3<S, false, kNW, NT, RT, WAVES, NBUF, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return -1;
  hipLaunchKernelGGL((k_posttrans_x3<S, false, kNW, NT, RT, WAVES, NBUF, false, true>), dim3((unsigned)gx, 1), dim3(WAVES * 64), lds, st, g);
  return 0;
}

template <int S, bool HAS_H, int kNW, int NT, int RT, int WAVES>
int launch_split(const XArgs& g, int nbuf, hipStream_t st) {
  // rows [0, Mf): wavefront tiles of 16 * RT whole rows, the straight-line epilogue; the rows beyond: a second launch with the
  // generic one
  const int Mf = g.M - g.M % (16 * RT);
  if (Mf > 0) {
    XArgs m = g;
    m.M = Mf;
    int rc;
    if constexpr (kNW == 80) rc = nbuf == 3 ? launch_v<S, HAS_H, kNW, NT, RT, WAVES, 3, false>(m, st) : launch_v<S, HAS_H, kNW, NT, RT, WAVES, 2, false>(m, st);
    else rc = launch_v<S, HAS_H, kNW, NT, RT, WAVES, 2, false>(m, st);      // 128 columns: three 74 KB buffers do not fit the LDS
    if (rc != 0) return rc;
  }
  if (g.M > Mf) {
    XArgs r = g;
    r.M = g.M - Mf;
    r.a = g.a + (size_t)Mf * g.lda;
    if (g.h) r.h = g.h + (size_t)Mf * g.ldh;
    for (int s = 0; s < S; ++s) if (g.row_scale[s]) r.row_scale[s] = g.row_scale[s] + Mf;
    if (g.row_post) r.row_post = g.row_post + Mf;
    if (g.residual) r.residual = g.residual + (size_t)Mf * g.ld_res;
    r.y = g.y + (size_t)Mf * g.ldy;
    constexpr int WT = (S + (HAS_H ? 1 : 0)) * NT > 15 ? 8 : 12;
    return launch_v<S, HAS_H, kNW, NT, 1, WT, 2, true>(r, st);
  }
  return 0;
}

template <int S, bool HAS_H, int kNW, int NT>
int launch_k(const XArgs& g, int nbuf, int shape, hipStream_t st) {
  // One row tile per wavefront, 12 wavefronts (3 per SIMD, <= 168 registers each) while the P * NT accumulator tiles fit
  // that budget, else 8 (256 registers).  
  constexpr int WAVES = (S + (HAS_H ? 1 : 0)) * NT > 15 ? 8 : 12;
  (void)shape;   // 8 wavefronts x 2 row tiles was measured in round 1 (slower) and spills with the 3-buffer pipeline: not built
  return launch_split<S, HAS_H, kNW, NT, 1, WAVES>(g, nbuf, st);
}

template <int S, bool HAS_H>
int launch_nt(const XArgs& g, int nt, int nbuf, int shape, hipStream_t st) {
  if (nw_of(g.N) == 128) return launch_k<S, HAS_H, 128, 8>(g, nbuf, shape, st);      // 80 < N <= 128: one 8-tile block
  if (nt <= 1) return launch_k<S, HAS_H, 80, 1>(g, nbuf, shape, st);
  if (nt <= 3) return launch_k<S, HAS_H, 80, 3>(g, nbuf, shape, st);
  return launch_k<S, HAS_H, 80, 5>(g, nbuf, shape, st);
}

template <int S>
int launch_s(const XArgs& g, bool has_h, int nt, int nbuf, int shape, hipStream_t st) {
  return has_h ? launch_nt<S, true>(g, nt, nbuf, shape, st) : launch_nt<S, false>(g, nt, nbuf, shape, st);
}

}  // namespace

extern "C" int64_t pna_posttrans_x3_packed_bytes(int32_t K, int32_t N, int32_t n_scaler, int32_t Kh, int64_t* wh_bytes) {
  const int kNW = nw_of(N);
  const int64_t nca = (K + kKC - 1) / kKC, nch = (Kh + kKC - 1) / kKC, nty = (N + kNW - 1) / kNW;
  if (wh_bytes) *wh_bytes = nty * nch * 3 * panel_bytes(kNW);
  return nty * nca * 3 * n_scaler * panel_bytes(kNW);
}

extern "C" int pna_posttrans_x3_pack_f32(const float* w_ref, int64_t ldw, int32_t N, int32_t K, int32_t n_scaler, int32_t Kh,
                                         void* w_img, void* wh_img, pna_stream_t stream) {
  if (!w_ref || !w_img || N <= 0 || K <= 0 || n_scaler < 1 || n_scaler > 3 || Kh < 0 || (Kh > 0 && !wh_img) ||
      ldw < (int64_t)Kh + (int64_t)n_scaler * K)
    return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3_pack_f32: bad arguments (n_scaler must be 1..3)");
  int64_t nh = 0;
  const int64_t nw = pna_posttrans_x3_packed_bytes(K, N, n_scaler, Kh, &nh);
  const int64_t elems = (nw + nh) / 2;
  const int blocks = (int)((elems + 255) / 256 > 4096 ? 4096 : (elems + 255) / 256);
  if (nw_of(N) == 128)
    hipLaunchKernelGGL(k_pack_x3<128>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_ref, (long)ldw, N, K, n_scaler, Kh,
                       (unsigned short*)w_img, (unsigned short*)wh_img);
  else
    hipLaunchKernelGGL(k_pack_x3<80>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_ref, (long)ldw, N, K, n_scaler, Kh,
                       (unsigned short*)w_img, (unsigned short*)wh_img);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}

extern "C" int pna_posttrans_x3_f32(const pna_posttrans_args* p, pna_stream_t stream) {
  if (!p) return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3_f32: null args");
  if (int rc_ss = pna_check_struct_size("pna_posttrans_x3_f32", p->struct_size, sizeof(*p))) return rc_ss;
  if (p->M < 0 || p->K <= 0 || p->N <= 0) return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3_f32: bad M/K/N");
  if (p->M == 0) return PNA_OK;
  if (!p->a || !p->w_img || !p->y) return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3_f32: a/w_img/y must be non-null");
  if (p->n_scaler < 1 || p->n_scaler > 3) return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3_f32: n_scaler must be 1..3");
  if (p->lda < p->K || p->ldy < p->N) return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3_f32: leading dimensions too small");
  if (p->K < 4 || (p->h != nullptr && p->Kh > 0 && p->Kh < 4))
    return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3_f32: K and Kh must be >= 4 (pad the operand and the weight with zero columns)");
  const bool has_h = p->h != nullptr && p->Kh > 0;
  if (has_h && (!p->wh_img || p->ldh < p->Kh)) return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3_f32: h given without wh_img / ldh too small");
  if (p->residual && p->ld_res < p->N) return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3_f32: ld_res too small");
  if ((p->col_scale == nullptr) != (p->col_shift == nullptr))
    return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3_f32: col_scale and col_shift come together");
  XArgs g;
  memset(&g, 0, sizeof(g));
  g.a = p->a; g.w_img = p->w_img; g.h = has_h ? p->h : nullptr; g.wh_img = has_h ? p->wh_img : nullptr; g.bias = p->bias;
  for (int s = 0; s < p->n_scaler; ++s) g.row_scale[s] = p->row_scale[s];
  g.row_post = p->row_post; g.col_scale = p->col_scale; g.col_shift = p->col_shift; g.residual = p->residual;
  g.y = p->y; g.lda = p->lda; g.ldh = p->ldh; g.ldy = p->ldy; g.ld_res = p->ld_res;
  g.M = p->M; g.K = p->K; g.N = p->N; g.Kh = has_h ? p->Kh : 0; g.relu = p->relu;
  if (p->relu < 0 || p->relu > 2) return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3_f32: relu must be 0, 1 or 2");
  g.slope = p->relu == 2 ? p->act_slope : 0.f;
  const int T = p->n_tower > 1 ? p->n_tower : 1;
  if (T > 1 && p->residual) return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3_f32: residual is not supported with n_tower > 1");
#ifdef PNA_AMD_EXPERIMENTS
  if (const char* e = getenv("PNA_X3_DBG_PTR")) g.dbg = (unsigned long long*)strtoull(e, nullptr, 0);   // device buffer: 8 counters per wavefront
#endif
  {   // one persistent workgroup per CU (and per column tile)
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      return pna_set_error(PNA_E_NODEVICE, "pna_posttrans_x3_f32: cannot query the device");
    const int kNW = nw_of(p->N);
    const int ny = (p->N + kNW - 1) / kNW;
    g.grid_x = (cus + ny - 1) / ny;
  }
  hipStream_t st = (hipStream_t)stream;
  if (p->row_perm) {
    // rows in a virtual order (degree groups): see include/pna_amd.h
    // one scaler block: 115 registers and 50 KB of LDS per workgroup, so TWO workgroups of 8 wavefronts per CU (independent
    // barriers), workgroup tile 128 rows (measured: 12 wavefronts 1.27 ms/step, 16 wavefronts 1.30, 12 x 2 row tiles 1.31 with
    // spills); three blocks: 12 wavefronts, 192 rows
    const int tile_rows = p->n_scaler == 1 ? 128 : 192;
    const bool wide = p->N > 80;                 // one 128-column block (one scaler block only: the caller sends its rest rows through the ordinary path)
    if (has_h || T > 1 || p->N < 1 || p->N > 128 || (wide && p->n_scaler != 1) || p->M % tile_rows != 0 || (p->n_scaler != 1 && p->n_scaler != 3) || (p->pipeline != 0 && p->pipeline != 3) ||
        (p->tile_image && p->image_stride <= 0) || (int64_t)p->ldy * 4 >= (1ll << 32) || (p->residual && (int64_t)p->ld_res * 4 >= (1ll << 32)))
      return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3_f32: row_perm needs N <= 80 (1 or 3 scalers) or 80 < N <= 128 (1 scaler), M % 128 == 0 (1 scaler) / M % 192 == 0 (3 scalers), no h panel / towers");
    g.perm = p->row_perm; g.tile_image = p->tile_image; g.img_stride = p->tile_image ? p->image_stride : 0;
    // (the short rest list of a degree plan -- ~26 tiles, one step-latency-bound tile per workgroup, 44 us -- on 4-wavefront
    // workgroups of 64 rows over 3x the CUs: measured, layer 1.245 vs 1.230 ms, not kept)
    const int rc2 = wide ? launch_grouped<1, 1, 8, 3, 128, 8>(g, st)
                    : p->n_scaler == 1 ? launch_grouped<1, 1, 8, X3_GRP_NBUF>(g, st) : launch_grouped<3, 1, 12>(g, st);
    if (rc2 != 0) return pna_set_error(PNA_E_LAUNCH, "pna_posttrans_x3_f32: could not reserve LDS (grouped mode)");
    hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e2));
    return PNA_OK;
  }
  const int nt = p->N >= 80 ? 5 : (p->N + 15) / 16;      // (80-column blocks; launch_nt switches to one 128-column block itself)
  int rc;
  const int pl = p->pipeline, shape = 0;
  if (pl != 0 && pl != 2 && pl != 3)
    return pna_set_error(PNA_E_INVALID, "pna_posttrans_x3_f32: pipeline must be 0 (default), 2 or 3");
  const int nbuf = pl ? pl : kDefaultNBuf;
  rc = 0;
  for (int t = 0; t < T && rc == 0; ++t) {           // towers: one persistent launch each (large M: launch cost is immaterial)
    XArgs gt = g;
    if (T > 1) {
      gt.a = g.a + (size_t)t * p->tower_stride_a;
      gt.w_img = (const unsigned char*)g.w_img + (size_t)t * p->tower_stride_w;
      if (has_h) { gt.h = g.h + (size_t)t * p->tower_stride_h; gt.wh_img = (const unsigned char*)g.wh_img + (size_t)t * p->tower_stride_wh; }
      if (g.bias) gt.bias = g.bias + (size_t)t * p->N;
      if (g.col_scale) { gt.col_scale = g.col_scale + (size_t)t * p->N; gt.col_shift = g.col_shift + (size_t)t * p->N; }
      gt.y = g.y + (size_t)t * p->tower_stride_y;
    }
    switch (p->n_scaler) {
      case 1: rc = launch_s<1>(gt, has_h, nt, nbuf, shape, st); break;
      case 2: rc = launch_s<2>(gt, has_h, nt, nbuf, shape, st); break;
      default: rc = launch_s<3>(gt, has_h, nt, nbuf, shape, st); break;
    }
  }
  if (rc != 0) return pna_set_error(PNA_E_LAUNCH, "pna_posttrans_x3_f32: could not reserve LDS");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}
