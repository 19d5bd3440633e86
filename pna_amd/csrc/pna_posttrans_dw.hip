// pna_posttrans_dw.hip -- the WEIGHT GRADIENT of the posttrans contraction for gfx950 (pna_posttrans_dw_f32, include/pna_amd.h).
//
// Replaces the autograd node of `self.posttrans(torch.cat([h, scaled aggregate], dim=1))` (models/dgl/pna_layer.py:206,
// realworld_benchmark/train/train_molecules_graph_regression.py:29-32: loss.backward()) for the weight and the bias:
//
//   grad_w[n, Kh + s K + k] = sum_m scale_s[m] gy[m, n] a[m, k]      grad_w[n, j < Kh] = sum_m gy[m, n] h[m, j]
//   grad_b[n]               = sum_m gy[m, n]
//
// i.e. C = L^T R with L = [gy | scale_1 . gy | scale_2 . gy] (M x S N) and R = [a | h | 1] (M x (K + Kh + 1)): a product whose
// REDUCTION runs over the M = 1e6 rows and whose output is 225 x 376.  Rounds 2-3 gave it to the vendor library (one GEMM: 44 TF/s,
// 3.06 ms; slab-batched bmm + sum: 1.3-1.5 ms + 0.3 ms to write the scaled copies of gy) -- the last library call on the training
// step's hot path.
//
// Here: the bf16x3 arithmetic of the forward contraction (pna_x3_split.h: every fp32 operand cut exactly into three bf16 terms,
// six partial products on v_mfma_f32_16x16x32_bf16, fp32 accumulate).  The MFMA's k index is the ROW m, so both operands are needed
// "k-major": a lane must hold 8 consecutive rows of ONE column.  Lanes therefore load columns (one dword per row: a wavefront
// reads 256 contiguous bytes of a row per instruction), scale and split their 8 values in registers and write one 16-byte
// fragment cell per term into LDS, [term][8-row group][column] -- a fragment read is one conflict-free ds_read_b128.
//
// Tiling: a workgroup owns a SLAB of rows and ONE THIRD of R's columns (8 tiles of 16 = 128 columns) against all of L (15 tiles):
// 120 output tiles in the registers of 4 wavefronts (32 tiles = 128 accumulator VGPRs each), 69 KB of LDS -> two workgroups per
// CU: while one converts the next 32 rows, the other multiplies (the structure of the one-kernel layer).  L is
// converted by every third (3 x 8 of 16 conversion units per step are L's): the price of keeping R's conversions -- the bigger
// operand -- unshared.  Partial tiles go to a workspace [slab][S N][K + Kh + 1]; a second kernel adds the slabs in float64, in a
// fixed order (deterministic), and scatters the sums into grad_w / grad_b.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "pna_amd.h"
#include "pna_internal.h"
#include "pna_x3_split.h"

namespace {
using namespace pna_x3;

constexpr int kThreads = 256;          // 4 wavefronts
constexpr int kMA = 240;               // columns of L a workgroup holds (15 tiles)
constexpr int kNB = 128;               // columns of R a workgroup holds (8 tiles: one third of <= 384)
constexpr int kMaxR = 384;
constexpr int kLeftBytes = 3 * 4 * kMA * 16, kRightBytes = 3 * 4 * kNB * 16;
constexpr int kLdsBytes = kLeftBytes + kRightBytes;     // 70 656

struct DArgs {
  const float* gy; long ldg;
  const float* a; long lda;
  const float* h; long ldh;
  const float* scale[3];
  long M;
  int N, S, K, Kh;
  int MT;            // tiles of L in use: ceil(S N / 16)
  int NTH;           // thirds of R in use
  long slab;         // rows per slab (a multiple of 32)
  float* part;       // [n_slab][kMA][kMaxR]
};

__global__ __launch_bounds__(kThreads, 2) void k_posttrans_dw(const DArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int third = blockIdx.x % g.NTH;
  const long sl = blockIdx.x / g.NTH;
  const long r_beg = sl * g.slab, r_end = min(g.M, r_beg + g.slab);
  constexpr unsigned left0 = 0, right0 = kLeftBytes;       // byte offsets into lds[]
  auto lds_write16 = [&](unsigned off, bf8 v) __attribute__((always_inline)) { *reinterpret_cast<bf8*>(lds + off) = v; };
  auto lds_read16 = [&](unsigned off) __attribute__((always_inline)) { return *reinterpret_cast<const bf8*>(lds + off); };
  // padding columns / rows are zero for the whole launch
  for (int i = tid; i < kLdsBytes / 16; i += kThreads) reinterpret_cast<u4*>(lds)[i] = u4{0u, 0u, 0u, 0u};
  __syncthreads();

  // ---- this wavefront's conversion units: L units {wave, wave + 4} and R units {wave, wave + 4} of 8 each; unit u = (column block
  // u & 1 of 64 columns, row group u >> 1 of 8 rows) ----
  const int N = g.N, S = g.S, K = g.K, Kh = g.Kh;
  int cbL[2], kgL[2];
  const float* pL[2]; bool okL[2];
  const float* pR[2]; unsigned ldR[2]; int kindR[2];       // 0: column of a / h, 1: the ones column, 2: padding (stays zero)
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int u = wave + 4 * q;
    cbL[q] = u & 1; kgL[q] = u >> 1;
    const int n = 64 * cbL[q] + lane;
    okL[q] = n < N;
    pL[q] = g.gy + min(n, N - 1);
    const int j = kNB * third + 64 * cbL[q] + lane;
    if (j < K) { pR[q] = g.a + j; ldR[q] = (unsigned)g.lda; kindR[q] = 0; }
    else if (j < K + Kh) { pR[q] = g.h + (j - K); ldR[q] = (unsigned)g.ldh; kindR[q] = 0; }
    else { pR[q] = g.gy; ldR[q] = 0; kindR[q] = j == K + Kh ? 1 : 2; }
  }
  float rawL[2][8], rawR[2][8];
  long row_staged = r_beg;                                 // first row of the step the raw registers belong to
  auto prefetch = [&](long row0) __attribute__((always_inline)) {
    row_staged = row0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        // wavefront-uniform row (the L and R units of a wavefront cover the same rows).  No branches: a row past the slab's end
        // re-reads the last row and is masked to +0 bits; the ones / padding columns read a dummy address and select afterwards
        const long rl = row0 + 8 * kgL[q] + r, rc = min(rl, r_end - 1);
        const unsigned keep = rl < r_end ? 0xFFFFFFFFu : 0u;
        const float vl = pL[q][rc * g.ldg], vr = pR[q][(size_t)rc * ldR[q]];
        rawL[q][r] = bfloat(fbits(vl) & keep);
        rawR[q][r] = bfloat(fbits(kindR[q] == 0 ? vr : 1.f) & keep);
      }
    }
  };
  auto stage = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (okL[q]) {
        const int n = 64 * cbL[q] + lane;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          if (s < S) {
            float x[8];
            if (g.scale[s]) {                                // the unit's 8 scaler values: wavefront-uniform scalar loads, here and not in
              const float* const sp = g.scale[s];            // prefetch() -- 48 more live SGPRs through the multiply phase spilled
#pragma unroll
              for (int r = 0; r < 8; ++r) {
                const long rl = min(row_staged + 8 * kgL[q] + r, r_end - 1);      // (rows past the end hold zeros already)
                x[r] = rawL[q][r] * sp[rl];
              }
            } else {                                         // NULL = the identity scaler: no multiply, like the forward
#pragma unroll
              for (int r = 0; r < 8; ++r) x[r] = rawL[q][r];
            }
            bf8 t0, t1, t2;
            split8(f4{x[0], x[1], x[2], x[3]}, f4{x[4], x[5], x[6], x[7]}, t0, t1, t2);
            const unsigned cell = (unsigned)(kgL[q] * kMA + s * N + n) * 16u;
            lds_write16(left0 + cell, t0);
            lds_write16(left0 + 4 * kMA * 16 + cell, t1);
            lds_write16(left0 + 8 * kMA * 16 + cell, t2);
          }
        }
      }
      if (kindR[q] != 2) {
        bf8 t0, t1, t2;
        split8(f4{rawR[q][0], rawR[q][1], rawR[q][2], rawR[q][3]}, f4{rawR[q][4], rawR[q][5], rawR[q][6], rawR[q][7]}, t0, t1, t2);
        const unsigned cell = (unsigned)(kgL[q] * kNB + 64 * cbL[q] + lane) * 16u;
        lds_write16(right0 + cell, t0);
        lds_write16(right0 + 4 * kNB * 16 + cell, t1);
        lds_write16(right0 + 8 * kNB * 16 + cell, t2);
      }
    }
  };

  // ---- this wavefront's output tiles: L tiles [8 mh, 8 mh + 8) x R tiles [4 ng, 4 ng + 4) of the third ----
  const int mh = wave & 1, ng = wave >> 1;
  f4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  const int MT = g.MT;
  constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};     // small partial products first (pna_posttrans_x3.hip)

  if (r_beg < r_end) {
    prefetch(r_beg);
    for (long row0 = r_beg; row0 < r_end; row0 += 32) {
      stage();
      __syncthreads();
      if (row0 + 32 < r_end) prefetch(row0 + 32);
      // two halves of the wavefront's four R tiles: the B fragments of two tiles (24 registers) stay, the A fragments are read once
      // per half (the LDS has the bandwidth: 60 x 16-byte reads against 192 MFMAs per wavefront and step)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        bf8 B[2][3];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int p = 0; p < 3; ++p)
            B[nt][p] = lds_read16(right0 + (unsigned)((p * 4 + lg) * kNB + (4 * ng + 2 * half + nt) * 16 + li) * 16u);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
          const int m = 8 * mh + mt;
          if (m < MT) {
            bf8 A[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) A[p] = lds_read16(left0 + (unsigned)((p * 4 + lg) * kMA + m * 16 + li) * 16u);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int pp = 0; pp < 6; ++pp)
                acc[mt][2 * half + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[TA[pp]], B[nt][TB[pp]], acc[mt][2 * half + nt], 0, 0, 0);
          }
        }
      }
      __syncthreads();
    }
  }
  // ---- partial tiles -> workspace: D[i = 4 lg + e][j = li] of tile (m, nt) ----
  float* const part = g.part + (size_t)sl * kMA * kMaxR;
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const int m = 8 * mh + mt;
    if (m < MT) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int j = kNB * third + (4 * ng + nt) * 16 + li;
#pragma unroll
        for (int e = 0; e < 4; ++e) part[(size_t)(m * 16 + 4 * lg + e) * kMaxR + j] = acc[mt][nt][e];
      }
    }
  }
}

// grad_w / grad_b = sum over the slabs (float64, ascending slab order), scattered to the reference's column order
struct RArgs {
  const float* part; long n_slab;
  int N, S, K, Kh;
  float* gw; long ldw; float* gb;
};

__global__ void k_posttrans_dw_reduce(const RArgs g) {
  const int R = g.K + g.Kh + 1;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)g.S * g.N * R;
  if (idx >= total) return;
  const int i = (int)(idx / R), j = (int)(idx - (long)i * R);
  const int s = i / g.N, n = i - s * g.N;
  if (j >= g.K && s > 0) return;                           // (the h / ones columns against the scaled copies of gy: not part of the gradient)
  double sum = 0.0;
  const float* p = g.part + (size_t)i * kMaxR + j;
  for (long sl = 0; sl < g.n_slab; ++sl) sum += (double)p[(size_t)sl * kMA * kMaxR];
  if (j < g.K) g.gw[(size_t)n * g.ldw + g.Kh + (size_t)s * g.K + j] = (float)sum;
  else if (j < g.K + g.Kh) g.gw[(size_t)n * g.ldw + (j - g.K)] = (float)sum;
  else if (g.gb) g.gb[n] = (float)sum;
}

// ---- the degree-grouped form (large graphs with a degree plan: pna_amd/degree_groups.py) -----------------------------------
// The scalers are functions of the in-degree alone (models/dgl/scalers.py:7-19), and the plan's virtual row order puts the rows of
// one degree into 128-row tiles.  Walking the rows in THAT order, 32 consecutive rows share their scaler values, so
//   block s of grad_w = sum over degree runs  scale_s(D) * P_run,     P_run = sum_{m in run} gy[m]^T [a | h | 1][m]
// needs ONE unscaled copy of gy (80 columns of L instead of 240) and a third of the multiply-adds; the scaling moves to the
// reduction pass.  (Round 3 tried this through the library -- packing the rows into slab order cost what it saved; here the rows
// are simply LOADED through the permutation.)  One workgroup of 8 wavefronts per CU holds all of R (384 columns, 74 KB of LDS) and
// all 5 x 24 output tiles (15 per wavefront); every workgroup walks a contiguous, equally long range of tiles and writes one
// partial product per degree run it meets.
constexpr int kGThreads = 512;
constexpr int kGL = 80;                                    // columns of L (unscaled gy): 5 tiles
constexpr int kGLeftBytes = 3 * 4 * kGL * 16, kGRightBytes = 3 * 4 * kMaxR * 16;
constexpr int kGLdsBytes = kGLeftBytes + kGRightBytes;    // 89 088

struct GArgs {
  const float* gy; long ldg;
  const float* a; long lda;
  const float* h; long ldh;
  const int32_t* row_perm;       // [128 * n_tiles]: node of every virtual row, -1 = padding
  const int32_t* tile_group;     // [n_tiles]
  const int32_t* wg_range;       // [n_wg][2]: tiles [lo, hi)
  const int32_t* wg_entry;       // [n_wg]: first workspace entry of the workgroup
  int N, K, Kh, MT;
  int a_plan;                    // `a` is in plan order: row = the virtual row
  float* part;                   // [n_entries][kGL][kMaxR]
};

__global__ __launch_bounds__(kGThreads, 1) void k_posttrans_dw_grouped(const GArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int t0 = g.wg_range[2 * blockIdx.x], t1 = g.wg_range[2 * blockIdx.x + 1];
  if (t0 >= t1) return;
  constexpr unsigned left0 = 0, right0 = kGLeftBytes;
  auto lds_write16 = [&](unsigned off, bf8 v) __attribute__((always_inline)) { *reinterpret_cast<bf8*>(lds + off) = v; };
  auto lds_read16 = [&](unsigned off) __attribute__((always_inline)) { return *reinterpret_cast<const bf8*>(lds + off); };
  for (int i = tid; i < kGLdsBytes / 16; i += kGThreads) reinterpret_cast<u4*>(lds)[i] = u4{0u, 0u, 0u, 0u};
  __syncthreads();

  // conversion units of this wavefront: rows 8 kg .. 8 kg + 7 of every step; L column block cbL (of 2), R column blocks cbL, cbL + 2,
  // cbL + 4 (of 6)
  const int N = g.N, K = g.K, Kh = g.Kh;
  const int kg = wave & 3, cbL = wave >> 2;
  const int nL = 64 * cbL + lane;
  const bool okL = nL < N;
  const float* const pL = g.gy + min(nL, N - 1);
  const float* pR[3]; unsigned ldR[3]; int kindR[3];       // 0: column of a / h, 1: the ones column, 2: padding (stays zero)
  bool seqR[3];                                             // the column's rows are addressed by the VIRTUAL row (a in plan order)
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int j = 64 * (cbL + 2 * q) + lane;
    seqR[q] = g.a_plan && j < K;
    if (j < K) { pR[q] = g.a + j; ldR[q] = (unsigned)g.lda; kindR[q] = 0; }
    else if (j < K + Kh) { pR[q] = g.h + (j - K); ldR[q] = (unsigned)g.ldh; kindR[q] = 0; }
    else { pR[q] = g.gy; ldR[q] = 0; kindR[q] = j == K + Kh ? 1 : 2; }
  }
  float rawL[8], rawR[3][8];
  auto prefetch = [&](long vrow0) __attribute__((always_inline)) {
    const int32_t* const ip = g.row_perm + vrow0 + 8 * kg;             // wavefront-uniform: eight scalar loads
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int id = ip[r];
      const unsigned keep = id >= 0 ? 0xFFFFFFFFu : 0u;                // a padding row reads node 0 and is masked to +0 bits
      const size_t idc = (size_t)max(id, 0);
      rawL[r] = bfloat(fbits(pL[idc * (size_t)g.ldg]) & keep);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const size_t rq = seqR[q] ? (size_t)(vrow0 + 8 * kg + r) : idc;
        const float v = pR[q][rq * ldR[q]];
        rawR[q][r] = bfloat(fbits(kindR[q] == 0 ? v : 1.f) & keep);
      }
    }
  };
  auto stage = [&]() __attribute__((always_inline)) {
    if (okL) {
      bf8 t0_, t1_, t2_;
      split8(f4{rawL[0], rawL[1], rawL[2], rawL[3]}, f4{rawL[4], rawL[5], rawL[6], rawL[7]}, t0_, t1_, t2_);
      const unsigned cell = (unsigned)(kg * kGL + nL) * 16u;
      lds_write16(left0 + cell, t0_);
      lds_write16(left0 + 4 * kGL * 16 + cell, t1_);
      lds_write16(left0 + 8 * kGL * 16 + cell, t2_);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      if (kindR[q] != 2) {
        bf8 t0_, t1_, t2_;
        split8(f4{rawR[q][0], rawR[q][1], rawR[q][2], rawR[q][3]}, f4{rawR[q][4], rawR[q][5], rawR[q][6], rawR[q][7]}, t0_, t1_, t2_);
        const unsigned cell = (unsigned)(kg * kMaxR + 64 * (cbL + 2 * q) + lane) * 16u;
        lds_write16(right0 + cell, t0_);
        lds_write16(right0 + 4 * kMaxR * 16 + cell, t1_);
        lds_write16(right0 + 8 * kMaxR * 16 + cell, t2_);
      }
    }
  };
  // output tiles of this wavefront: all MT tiles of L x R tiles 3 wave .. 3 wave + 2
  f4 acc[5][3];
  auto reset = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  };
  auto flush = [&](int entry) __attribute__((always_inline)) {
    float* const part = g.part + (size_t)entry * kGL * kMaxR;
#pragma unroll
    for (int mt = 0; mt < 5; ++mt)
      if (mt < g.MT) {
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
          const int j = (3 * wave + nt) * 16 + li;
#pragma unroll
          for (int e = 0; e < 4; ++e) part[(size_t)(mt * 16 + 4 * lg + e) * kMaxR + j] = acc[mt][nt][e];
        }
      }
  };
  constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
  reset();
  int entry = g.wg_entry[blockIdx.x];
  int cur = g.tile_group[t0];
  const long v_end = (long)t1 * 128;
  prefetch((long)t0 * 128);
  for (int t = t0; t < t1; ++t) {
    const int grp = g.tile_group[t];
    if (grp != cur) { flush(entry); ++entry; reset(); cur = grp; }     // (workgroup-uniform: registers only, no barrier)
    for (int ks = 0; ks < 4; ++ks) {
      const long vrow0 = (long)t * 128 + 32 * ks;
      stage();
      __syncthreads();
      if (vrow0 + 32 < v_end) prefetch(vrow0 + 32);
      bf8 B[3][3];
#pragma unroll
      for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          B[nt][p] = lds_read16(right0 + (unsigned)((p * 4 + lg) * kMaxR + (3 * wave + nt) * 16 + li) * 16u);
#pragma unroll
      for (int mt = 0; mt < 5; ++mt) {
        if (mt < g.MT) {
          bf8 A[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) A[p] = lds_read16(left0 + (unsigned)((p * 4 + lg) * kGL + mt * 16 + li) * 16u);
#pragma unroll
          for (int pp = 0; pp < 6; ++pp)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[TA[pp]], B[nt][TB[pp]], acc[mt][nt], 0, 0, 0);
        }
      }
      __syncthreads();
    }
  }
  flush(entry);
}

// grad_w / grad_b of the grouped form: sum over the workspace entries (float64, ascending), block s weighted by the scaler value of
// the entry's degree group
struct GRArgs {
  const float* part; int n_entries;
  const int32_t* entry_group; const float* group_scale; int S;       // group_scale: [n_groups][S]
  int N, K, Kh;
  float* gw; long ldw; float* gb;
};

// block = 64 elements x 8 entry lanes: lane y adds entries y, y + 8, ... (float64), the eight partial sums are added in a fixed order
__global__ __launch_bounds__(512) void k_posttrans_dw_grouped_reduce(const GRArgs g) {
  __shared__ double red[3][8][64];
  const int R = g.K + g.Kh + 1;
  const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
  const long idx = (long)blockIdx.x * 64 + x;
  const bool live = idx < (long)g.N * R;
  const int n = live ? (int)(idx / R) : 0, j = live ? (int)(idx - (long)n * R) : 0;
  const float* p = g.part + (size_t)n * kMaxR + j;
  double sum[3] = {0.0, 0.0, 0.0};
  if (live) {
    if (j < g.K) {
      for (int e = y; e < g.n_entries; e += 8) {
        const double v = (double)p[(size_t)e * kGL * kMaxR];
        const float* const sc = g.group_scale + (size_t)g.entry_group[e] * g.S;
        for (int s = 0; s < g.S; ++s) sum[s] += v * (double)sc[s];
      }
    } else {
      for (int e = y; e < g.n_entries; e += 8) sum[0] += (double)p[(size_t)e * kGL * kMaxR];
    }
  }
  for (int s = 0; s < 3; ++s) red[s][y][x] = sum[s];
  __syncthreads();
  if (y == 0 && live) {
    double t[3] = {0.0, 0.0, 0.0};
    for (int k = 0; k < 8; ++k)
      for (int s = 0; s < 3; ++s) t[s] += red[s][k][x];
    if (j < g.K) {
      for (int s = 0; s < g.S; ++s) g.gw[(size_t)n * g.ldw + g.Kh + (size_t)s * g.K + j] = (float)t[s];
    } else if (j < g.K + g.Kh) g.gw[(size_t)n * g.ldw + (j - g.K)] = (float)t[0];
    else if (g.gb) g.gb[n] = (float)t[0];
  }
}

int plan(int64_t M, int N, int S, int K, int Kh, long* slab, long* n_slab, int* nth) {
  if (M < 1 || N < 1 || S < 1 || S > 3 || K < 1 || Kh < 0 || S * N > kMA || K + Kh + 1 > kMaxR) return 0;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  *nth = (K + Kh + 1 + kNB - 1) / kNB;
  long want = (2L * cus) / *nth;                           // two workgroups per CU in one wave of the grid
  if (want < 1) want = 1;
  long rows = (M + want - 1) / want;
  rows = (rows + 31) / 32 * 32;
  if (rows < 256) rows = 256;                              // (small M: fewer, longer slabs -- the reduction pass reads every slab)
  *slab = rows;
  *n_slab = (M + rows - 1) / rows;
  return 1;
}
}  // namespace

extern "C" int64_t pna_posttrans_dw_workspace_bytes(int64_t M, int32_t N, int32_t n_scaler, int32_t K, int32_t Kh) {
  long slab, n_slab; int nth;
  if (!plan(M, N, n_scaler, K, Kh, &slab, &n_slab, &nth)) return -1;
  return (int64_t)n_slab * kMA * kMaxR * 4;
}

extern "C" int pna_posttrans_dw_f32(const pna_posttrans_dw_args* p, pna_stream_t stream) {
  if (!p) return pna_set_error(PNA_E_INVALID, "pna_posttrans_dw_f32: null args");
  if (int rc_ss = pna_check_struct_size("pna_posttrans_dw_f32", p->struct_size, sizeof(*p))) return rc_ss;
  if (!p->gy || !p->a || !p->grad_w || !p->workspace || (p->Kh > 0 && !p->h))
    return pna_set_error(PNA_E_INVALID, "pna_posttrans_dw_f32: null argument");
  long slab, n_slab; int nth;
  if (!plan(p->M, p->N, p->n_scaler, p->K, p->Kh, &slab, &n_slab, &nth))
    return pna_set_error(PNA_E_INVALID, "pna_posttrans_dw_f32: unsupported shape (n_scaler in 1..3, n_scaler * N <= 240, K + Kh + 1 <= 384; "
                                        "pna_posttrans_dw_workspace_bytes returns -1 for it)");
  if (p->ldg < p->N || p->lda < p->K || (p->Kh > 0 && p->ldh < p->Kh) || p->ldw < p->Kh + (int64_t)p->n_scaler * p->K)
    return pna_set_error(PNA_E_INVALID, "pna_posttrans_dw_f32: leading dimension too small");
  if (p->row_scale[0] && (p->Kh > 0 || p->grad_b))
    return pna_set_error(PNA_E_INVALID, "pna_posttrans_dw_f32: the h panel and grad_b are formed from the first copy of gy: row_scale[0] must be NULL with them");
  if (p->workspace_bytes < (int64_t)n_slab * kMA * kMaxR * 4) return pna_set_error(PNA_E_INVALID, "pna_posttrans_dw_f32: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  DArgs g;
  g.gy = p->gy; g.ldg = (long)p->ldg; g.a = p->a; g.lda = (long)p->lda; g.h = p->h; g.ldh = (long)p->ldh;
  for (int s = 0; s < 3; ++s) g.scale[s] = s < p->n_scaler ? p->row_scale[s] : nullptr;
  g.M = (long)p->M; g.N = p->N; g.S = p->n_scaler; g.K = p->K; g.Kh = p->Kh;
  g.MT = (p->n_scaler * p->N + 15) / 16; g.NTH = nth; g.slab = slab; g.part = (float*)p->workspace;
  if (hipFuncSetAttribute((const void*)k_posttrans_dw, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes) != hipSuccess)
    return pna_set_error(PNA_E_LAUNCH, "pna_posttrans_dw_f32: LDS attribute refused");
  hipLaunchKernelGGL(k_posttrans_dw, dim3((unsigned)(n_slab * nth)), dim3(kThreads), kLdsBytes, st, g);
  if (hipGetLastError() != hipSuccess) return pna_set_error(PNA_E_LAUNCH, "pna_posttrans_dw_f32: launch failed");
  RArgs r;
  r.part = (const float*)p->workspace; r.n_slab = n_slab; r.N = p->N; r.S = p->n_scaler; r.K = p->K; r.Kh = p->Kh;
  r.gw = p->grad_w; r.ldw = (long)p->ldw; r.gb = p->grad_b;
  const long total = (long)p->n_scaler * p->N * (p->K + p->Kh + 1);
  hipLaunchKernelGGL(k_posttrans_dw_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, r);
  if (hipGetLastError() != hipSuccess) return pna_set_error(PNA_E_LAUNCH, "pna_posttrans_dw_f32: launch failed");
  return PNA_OK;
}

extern "C" int64_t pna_posttrans_dw_grouped_workspace_bytes(int32_t N, int32_t K, int32_t Kh, int32_t n_entries) {
  if (N < 1 || N > kGL || K < 1 || Kh < 0 || K + Kh + 1 > kMaxR || n_entries < 1) return -1;
  return (int64_t)n_entries * kGL * kMaxR * 4;
}

extern "C" int pna_posttrans_dw_grouped_f32(const pna_posttrans_dw_grouped_args* p, pna_stream_t stream) {
  if (!p) return pna_set_error(PNA_E_INVALID, "pna_posttrans_dw_grouped_f32: null args");
  if (int rc_ss = pna_check_struct_size("pna_posttrans_dw_grouped_f32", p->struct_size, sizeof(*p))) return rc_ss;
  if (!p->gy || !p->a || !p->grad_w || !p->workspace || (p->Kh > 0 && !p->h) || !p->row_perm || !p->tile_group || !p->wg_range || !p->wg_entry ||
      !p->entry_group || !p->group_scale)
    return pna_set_error(PNA_E_INVALID, "pna_posttrans_dw_grouped_f32: null argument");
  const int64_t need = pna_posttrans_dw_grouped_workspace_bytes(p->N, p->K, p->Kh, p->n_entries);
  if (need < 0 || p->n_scaler < 1 || p->n_scaler > 3 || p->n_workgroups < 1)
    return pna_set_error(PNA_E_INVALID, "pna_posttrans_dw_grouped_f32: unsupported shape (N <= 80, K + Kh + 1 <= 384, n_scaler in 1..3)");
  if (p->ldg < p->N || p->lda < p->K || (p->Kh > 0 && p->ldh < p->Kh) || p->ldw < p->Kh + (int64_t)p->n_scaler * p->K)
    return pna_set_error(PNA_E_INVALID, "pna_posttrans_dw_grouped_f32: leading dimension too small");
  if (p->workspace_bytes < need) return pna_set_error(PNA_E_INVALID, "pna_posttrans_dw_grouped_f32: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  GArgs g;
  g.gy = p->gy; g.ldg = (long)p->ldg; g.a = p->a; g.lda = (long)p->lda; g.h = p->h; g.ldh = (long)p->ldh;
  g.row_perm = p->row_perm; g.tile_group = p->tile_group; g.wg_range = p->wg_range; g.wg_entry = p->wg_entry;
  g.N = p->N; g.K = p->K; g.Kh = p->Kh; g.MT = (p->N + 15) / 16; g.part = (float*)p->workspace;
  g.a_plan = p->a_plan_order != 0;
  if (hipFuncSetAttribute((const void*)k_posttrans_dw_grouped, hipFuncAttributeMaxDynamicSharedMemorySize, kGLdsBytes) != hipSuccess)
    return pna_set_error(PNA_E_LAUNCH, "pna_posttrans_dw_grouped_f32: LDS attribute refused");
  hipLaunchKernelGGL(k_posttrans_dw_grouped, dim3((unsigned)p->n_workgroups), dim3(kGThreads), kGLdsBytes, st, g);
  if (hipGetLastError() != hipSuccess) return pna_set_error(PNA_E_LAUNCH, "pna_posttrans_dw_grouped_f32: launch failed");
  GRArgs r;
  r.part = (const float*)p->workspace; r.n_entries = p->n_entries; r.entry_group = p->entry_group; r.group_scale = p->group_scale; r.S = p->n_scaler;
  r.N = p->N; r.K = p->K; r.Kh = p->Kh; r.gw = p->grad_w; r.ldw = (long)p->ldw; r.gb = p->grad_b;
  const long total = (long)p->N * (p->K + p->Kh + 1);
  hipLaunchKernelGGL(k_posttrans_dw_grouped_reduce, dim3((unsigned)((total + 63) / 64)), dim3(512), 0, st, r);
  if (hipGetLastError() != hipSuccess) return pna_set_error(PNA_E_LAUNCH, "pna_posttrans_dw_grouped_f32: launch failed");
  return PNA_OK;
}
