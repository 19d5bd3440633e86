// degree_fused.hip -- EXPERIMENT, NOT PART OF THE SHIPPED LIBRARY (DESIGN.md 4.7 point 7): gather + degree-grouped contraction of
// PNASimpleLayer in ONE kernel, the 4F aggregate never in HBM.  First version, kept for the next round:
//   * as built by default (two workgroups per CU, the running sums folded by single v_add_f32 / v_mul_f32 instructions): statistics
//     = the production gather's bits, y within 1.1e-7 of max|y| of the two-kernel path on five shapes, C3 layer 1.257 ms against
//     1.245 ms for the shipped two-kernel degree-grouped path on the same box: correct, level, not yet faster;
//   * with the fold written as plain C++ (-DDF_PACKED_FOLD: hipcc vectorises it into v_pk_add_f32 / v_pk_mul_f32 / v_pk_mov_b32
//     with op_sel swizzles) whole 16-row wavefront tiles come out wrong, differently from run to run, whenever a second
//     wavefront shares the SIMD (two 4-wavefront workgroups per CU, or one of 8: -DDF_WAVES=8); exact with one wavefront per
//     SIMD (DF_WGS=1, 1.84 ms).  Replacing only the inline-asm v_max / v_min by fmaxf / fminf changes nothing.  The shipped
//     kernels contain packed fp32 ops next to MFMAs too and do NOT show this (tests/test_gpu_determinism.py); what exactly
//     the failing combination is, is open.  tools/df_check.py drives all of it (DF_WGS, DF_LIB, DF_DEBUG_AGG).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC -Iinclude -Ipna_amd/csrc tools/ubench/degree_fused.hip -o tools/ubench/libdegree_fused.so
//
//   y[perm[v]] = epilogue( bias + W_D . [mean | max | min | std](messages into perm[v]) ),   W_D = sum_s scale_s(D) W_s
//   (models/dgl/pna_layer.py:186-216 with the scaler blocks combined per in-degree, DESIGN.md 4.2d / 4.7 point 7)
//
// Rows come in the degree plan's order (pna_amd/degree_groups.py): a wavefront owns 16 rows of ONE in-degree D, so its gather
// loop is uniform -- D iterations for every lane, no tail.  Lane (li = lane & 15, lg = lane >> 4) keeps the running sum, sum of
// squares, max and min of features fb * 32 + lg * 8 .. + 8 of row li for every feature block fb (the four lanes of a row read one
// 128-byte strip of a source row per block, all blocks of a row back to back: the row's DRAM page is touched once -- a
// prototype that made one pass per block ran 0.73 instead of 0.55 ms, tools/gather_strip_time.py).  The fold is the production
// gather's (pna_segreduce.hip: s += m, q += m * m, v_max / v_min, edge order; mean = s / D correctly rounded): the statistics
// are the same bits.  They ARE the MFMA A operand: with K ordered (feature block, aggregator) -- the weight image is packed
// to match -- chunk c = 4 fb + a multiplies the lane's eight values of aggregator a, split into three bf16 terms as in
// pna_posttrans_x3.hip.  Weights: one combined image per degree group, streamed through three LDS buffers by
// global_load_lds, one barrier per chunk in the middle of the chunk's MFMA stream (the pipeline of pna_posttrans_x3.hip).
// Workgroups of 4 wavefronts (64 rows), two per CU: while one multiplies, the other gathers.
// Rows that no degree group holds (rare degrees, hubs) stay on the two-kernel path over their compact list.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "pna_amd.h"
#include "pna_rowstats.h"
#include "pna_x3_split.h"

// (would become pna_degree_fused_args of include/pna_amd.h)
//   row_perm[v], v in [0, M): node of virtual row v (or -1: padding); M a multiple of 64; every aligned block of tile_rows (64 or
//     128) virtual rows has one in-degree (padding aside) and uses weight image tile_image[v / tile_rows].
//   w_img: images image_stride bytes apart, each pna_posttrans_x3_pack_f32 (n_scaler = 1, Kh = 0, 80-column block) of the combined
//     weight with K reordered as chunks of 32: chunk 4 fb + a = features [32 fb, 32 fb + 32) of aggregator a, zero beyond F.
//   x: 16-byte aligned rows of pitch ldx >= round_up(F, 8) floats (ldx % 4 == 0); 1 <= F <= 96, 1 <= N <= 80.
extern "C" {
typedef struct degree_fused_args {
  const int32_t* rowptr; const int32_t* col; const float* x; int64_t ldx; int32_t F; int32_t N;
  const int32_t* row_perm; int64_t M; const int32_t* tile_image; int32_t tile_rows; int32_t relu;
  const void* w_img; int64_t image_stride; const float* bias; const float* col_scale; const float* col_shift;
  const float* residual; int64_t ld_res; float* y; int64_t ldy; float act_slope; int32_t workgroups_per_cu;
} degree_fused_args;
}

namespace {

using namespace pna_x3;
using pna_dev::div_rn;
using pna_dev::vmax;
using pna_dev::vmin;

struct DFArgs {
  const int* rowptr; const int* col; const float* x; long ldx; int F;
  const int* perm; const int* tile_image; const unsigned char* w_img; long img_stride;
  const float* bias; const float* col_scale; const float* col_shift; const float* residual; float* y;
  long ldy, ld_res;
  int M, N, relu, tile_shift;
  float slope;
  float* agg_dbg; long ld_dbg;       // development: the statistics as the contraction sees them, [mean | max | min | std] x F per virtual row
};
float* g_agg_dbg = nullptr;
long g_ld_dbg = 0;

#ifndef DF_WAVES
#define DF_WAVES 4          // wavefronts per workgroup (development: 8 = one 128-row workgroup per CU instead of two of 64 rows)
#endif
constexpr int kNW = 80, kNT = 5, kWaves = DF_WAVES, kThreads = 64 * DF_WAVES, kNBuf = 3;
constexpr int kChunkV = 3 * 4 * kNW;                      // 16-byte pieces of one chunk image: [term][lane group][80 cols][8 k] bf16
constexpr int kNI = (kChunkV + kThreads - 1) / kThreads;  // global_load_lds instructions per wavefront per chunk
#ifndef DF_KU
#define DF_KU 2
#endif
#ifndef DF_ATTR
#define DF_ATTR
#endif
constexpr int kU = DF_KU;                                     // edges per gather batch (2 x NFB x 2 loads of 16 bytes per lane in flight)

template <int NFB>
__global__ __launch_bounds__(kThreads, 2) DF_ATTR void k_degree_fused(const DFArgs g) {
  constexpr int NC = 4 * NFB;                             // chunks of 32 k values per tile: (feature block, aggregator)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int ntiles = g.M / (kWaves * 16);

  float* const colc = reinterpret_cast<float*>(lds + (size_t)kNBuf * kChunkV * 16);       // [3][80]: bias | scale | shift
  for (int i = tid; i < kNW; i += kThreads) {
    colc[i] = (g.bias && i < g.N) ? g.bias[i] : 0.f;
    colc[kNW + i] = (g.col_scale && i < g.N) ? g.col_scale[i] : 1.f;
    colc[2 * kNW + i] = (g.col_shift && i < g.N) ? g.col_shift[i] : 0.f;
  }

  f4 acc[kNT];
#pragma unroll
  for (int n = 0; n < kNT; ++n) acc[n] = (f4){0.f, 0.f, 0.f, 0.f};

  // ---- weight chunks: global -> LDS, asynchronously (every wavefront issues exactly kNI copies per chunk) ----------------
  auto stage = [&](int c, int buf, long ib) __attribute__((always_inline)) {
    const unsigned char* src = g.w_img + ib + (size_t)c * kChunkV * 16;
    unsigned char* dst = lds + (size_t)buf * kChunkV * 16;
#pragma unroll
    for (int i = 0; i < kNI; ++i) {
      int w0 = (i * kWaves + wave) * 64;
      if (w0 >= kChunkV) w0 = w0 % kChunkV;              // a slot past the image re-copies an earlier piece (same bytes, same address)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(w0 + lane) * 16),
                                       (__attribute__((address_space(3))) void*)(dst + (size_t)w0 * 16), 16, 0, 0);
    }
  };
  auto image_of = [&](int t) -> long {                    // byte offset of workgroup tile t's weight image (wave-uniform)
    const int tt = min(t, ntiles - 1);
    return (long)g.tile_image[tt >> g.tile_shift] * g.img_stride;
  };

  int t = blockIdx.x;                                     // the workgroup's current tile
  // ---- the gather: running statistics of the wavefront's 16 rows ----------------------------------------------------------
  float S_[NFB][8], Q_[NFB][8], MX[NFB][8], MN[NFB][8];
  int deg = 0;                                            // in-degree of the tile's rows (wave-uniform)
  int f0[NFB];
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb) {
    f0[fb] = fb * 32 + lg * 8;
    if (f0[fb] >= g.F) f0[fb] = (g.F - 1) / 8 * 8;       // a strip past the row: re-read the row's last strip (values dropped in frag())
  }
  auto gather = [&](int t) __attribute__((always_inline)) {
    const int r0 = (t * kWaves + wave) * 16;
    const int first = __builtin_amdgcn_readfirstlane(g.perm[r0]);
    int node = g.perm[r0 + li];
    if (node < 0) node = max(first, 0);                   // padding rows repeat the tile's first row (nothing is stored for them)
    const int beg = g.rowptr[node];
    const int D = first < 0 ? 0 : __builtin_amdgcn_readfirstlane(g.rowptr[node + 1] - beg);
    deg = D;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int j = 0; j < 8; ++j) { S_[fb][j] = 0.f; Q_[fb][j] = 0.f; MX[fb][j] = -INFINITY; MN[fb][j] = INFINITY; }
    for (int e = 0; e < D; e += kU) {
      int idx[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) idx[u] = g.col[beg + min(e + u, D - 1)];
      f4 v[kU][NFB][2];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const float* p = g.x + (size_t)idx[u] * g.ldx;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) {
          v[u][fb][0] = *reinterpret_cast<const f4*>(p + f0[fb]);
          v[u][fb][1] = *reinterpret_cast<const f4*>(p + f0[fb] + 4);
        }
      }
      // branch-free tail (as the production gather's partial batch): a slot past the row holds a copy of the row's last edge --
      // idempotent for max / min, replaced by +0 for the sums -- so that all kU edges' loads stay in flight together
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const bool on = e + u < D;                        // (wave-uniform)
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float m = v[u][fb][j >> 2][j & 3];
            const float ms = on ? m : 0.f;
#if !defined(DF_PACKED_FOLD)     // the sums as single VALU instructions the compiler cannot pack (see the header: with the plain C++ below
                                 // hipcc emits v_pk_add_f32 / v_pk_mul_f32 / v_pk_mov_b32 with op_sel swizzles, and the sums come out wrong
                                 // in a few wavefront tiles per launch when a second wavefront shares the SIMD)
            float s1, q1, p1;
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(S_[fb][j]), "v"(ms));
            asm volatile("v_mul_f32 %0, %1, %1" : "=v"(p1) : "v"(ms));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(q1) : "v"(Q_[fb][j]), "v"(p1));
            S_[fb][j] = s1; Q_[fb][j] = q1;
#else
            S_[fb][j] = S_[fb][j] + ms;
            Q_[fb][j] = Q_[fb][j] + ms * ms;
#endif
#if defined(DF_MINMAX_C)         // development: no inline-asm VALU in the gather (NaN handling differs: inputs here are finite)
            MX[fb][j] = __builtin_fmaxf(MX[fb][j], m);
            MN[fb][j] = __builtin_fminf(MN[fb][j], m);
#else
            MX[fb][j] = vmax(MX[fb][j], m);
            MN[fb][j] = vmin(MN[fb][j], m);
#endif
          }
      }
    }
  };

  // ---- chunk c = 4 fb + a: the lane's eight values of aggregator a (0 mean, 1 max, 2 min, 3 std), split into three bf16 terms ----
  bf8 A[3];
  auto frag = [&](auto c_c) __attribute__((always_inline)) {
    constexpr int c = decltype(c_c)::value, fb = c / 4, a = c % 4;
    const float D = (float)deg, invD = 1.0f / D;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = S_[fb][j], q = Q_[fb][j];
      float r;
      if (a == 0) {
        r = div_rn(s, D, invD);
      } else if (a == 3) {
        const float mean = div_rn(s, D, invD), msq = div_rn(q, D, invD);
        float var = msq - mean * mean;
        var = var < 0.f ? 0.f : var;
        r = sqrtf(var + 1e-5f);
      } else {
        const float e = a == 1 ? MX[fb][j] : MN[fb][j];
        r = q != q ? q : e;                               // v_max / v_min drop NaN; q is NaN iff a message is (pna_rowstats.h)
      }
      if (deg <= 0) r = 0.f;                              // rows without in-edges: DGL leaves them at zero
      if (fb == NFB - 1 && fb * 32 + lg * 8 + j >= g.F) r = 0.f;      // padding features of the last block (their weights are 0; the
      v[j] = r;                                                       // table's padding columns may hold anything)
    }
    if (g.agg_dbg && t < ntiles) {
      const int row = (t * kWaves + wave) * 16 + li;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (fb * 32 + lg * 8 + j < g.F) g.agg_dbg[(size_t)row * g.ld_dbg + a * g.F + fb * 32 + lg * 8 + j] = v[j];
    }
    const f4 lo4 = (f4){v[0], v[1], v[2], v[3]}, hi4 = (f4){v[4], v[5], v[6], v[7]};
    if (__builtin_amdgcn_ballot_w64(absmax8(lo4, hi4) == INFINITY) != 0) split8_inf(lo4, hi4, A[0], A[1], A[2]);
    else split8(lo4, hi4, A[0], A[1], A[2]);
  };

  // ---- epilogue: BatchNorm / ReLU / residual, rows scattered to node order through perm -----------------------------------
  auto epilogue = [&](int t) __attribute__((always_inline)) {
    const int row0 = (t * kWaves + wave) * 16;
    const float lo = g.relu ? 0.f : -INFINITY;
    const bool leaky = g.relu == 2;
    const unsigned ldyb = (unsigned)g.ldy * 4u, ldrb = (unsigned)g.ld_res * 4u;
    typedef int i4 __attribute__((ext_vector_type(4)));
    const i4 pr = *reinterpret_cast<const i4*>(g.perm + row0 + 4 * lg);     // the lane's four rows of y / residual (-1: padding)
    float res[kNT][4];
#pragma unroll
    for (int n = 0; n < kNT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) res[n][r] = 0.f;
    if (g.residual) {
      const char* rbase = reinterpret_cast<const char*>(g.residual);
#pragma unroll
      for (int n = 0; n < kNT; ++n) {
        const unsigned cc = (unsigned)min(n * 16 + li, g.N - 1) * 4u;
#pragma unroll
        for (int r = 0; r < 4; ++r) res[n][r] = *reinterpret_cast<const float*>(rbase + (size_t)(unsigned)max(pr[r], 0) * ldrb + cc);
      }
    }
    char* const ybase = reinterpret_cast<char*>(g.y);
#pragma unroll
    for (int n = 0; n < kNT; ++n) {
      const int cl = n * 16 + li;
      const float cb = colc[cl], cs = colc[kNW + cl], ct = colc[2 * kNW + cl];
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = acc[n][r] + cb;
        x = __builtin_fmaf(x, cs, ct);
        x = x < lo ? (leaky ? x * g.slope : 0.f) : x;    // ReLU / LeakyReLU / none (lo = -inf); NaN < lo is false: NaN is kept
        v[r] = res[n][r] + x;
        acc[n][r] = 0.f;
      }
      if (cl < g.N) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (pr[r] >= 0) *reinterpret_cast<float*>(ybase + (size_t)(unsigned)pr[r] * ldyb + (unsigned)cl * 4u) = v[r];
      }
    }
  };

  // ---- the (tile, chunk) pipeline: step k reads LDS buffer k % 3; barrier B_k sits in the middle of step k; after B_k every
  //      wavefront has finished step k-1, so buffer (k+2) % 3 is free: step k+2's image is copied then and waited for (vmcnt(0))
  //      before B_{k+1} ------------------------------------------------------------------------------------------------------
  if (t >= ntiles) return;
  long ib_cur = image_of(t), ib_next = image_of(t + (int)gridDim.x);
  int buf = 0;
  stage(0, 0, ib_cur);
  stage(1, 1, ib_cur);
  gather(t);
  frag(std::integral_constant<int, 0>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  auto step = [&](auto c_c) __attribute__((always_inline)) {
    constexpr int c = decltype(c_c)::value;
    const unsigned ba0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + (unsigned)(buf * kChunkV + lg * kNW + li) * 16u;
    const int buf2 = buf == 0 ? kNBuf - 1 : buf - 1;     // (k + 2) % 3
    bf8 B[2][3];                                         // B fragments of column tile n (slot n & 1): one ds_read_b128 per term
    constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
    constexpr int H = (kNT - 1) / 2;
#if defined(DF_COUNTED_B)
    // B fragments double buffered in registers, counted wait (the scheme of pna_posttrans_x3.hip): tile n+1's three reads are
    // issued, then lgkmcnt(3) lets exactly those stay in flight behind tile n's MFMAs
#pragma unroll
    for (int tm = 0; tm < 3; ++tm)
      asm volatile("ds_read_b128 %0, %1" : "=v"(B[0][tm]) : "v"(ba0 + (unsigned)(tm * 4 * kNW * 16)) : "memory");
#pragma unroll
    for (int n = 0; n < kNT; ++n) {
      const int slot = n & 1;
      if (n + 1 < kNT) {
#pragma unroll
        for (int tm = 0; tm < 3; ++tm)
          asm volatile("ds_read_b128 %0, %1" : "=v"(B[slot ^ 1][tm]) : "v"(ba0 + (unsigned)(tm * 4 * kNW * 16 + (n + 1) * 256)) : "memory");
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(B[slot][0]), "+v"(B[slot][1]), "+v"(B[slot][2]) : : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(B[slot][0]), "+v"(B[slot][1]), "+v"(B[slot][2]) : : "memory");
      }
#else
    // B fragments: column tile n+1's three reads are issued before tile n's MFMAs, but the wavefront waits for ALL of them
    // (lgkmcnt(0)) before it issues an MFMA: no LDS read lands while this wavefront's MFMAs are in flight.  The counted wait of
    // pna_posttrans_x3.hip (lgkmcnt(3): only tile n's fragments) gave errors of 2^-16 in whole 16-row tiles here, different
    // from run to run, when two of these 4-wavefront workgroups shared a CU -- exact with one workgroup per CU, with this wait,
    // or with 32 cycles of s_nop behind each tile's MFMAs; waiting with lgkmcnt(0) BEFORE issuing the next reads was not enough.
    // (That was BEFORE the packed-fp32 fold was found to be what produced the wrong sums: -DDF_COUNTED_B restores the counted wait.)
#pragma unroll
    for (int tm = 0; tm < 3; ++tm)
      asm volatile("ds_read_b128 %0, %1" : "=v"(B[0][tm]) : "v"(ba0 + (unsigned)(tm * 4 * kNW * 16)) : "memory");
#pragma unroll
    for (int n = 0; n < kNT; ++n) {
      const int slot = n & 1;
      if (n + 1 < kNT) {
#pragma unroll
        for (int tm = 0; tm < 3; ++tm)
          asm volatile("ds_read_b128 %0, %1" : "=v"(B[slot ^ 1][tm]) : "v"(ba0 + (unsigned)(tm * 4 * kNW * 16 + (n + 1) * 256)) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(B[0][0]), "+v"(B[0][1]), "+v"(B[0][2]), "+v"(B[1][0]), "+v"(B[1][1]), "+v"(B[1][2]) : : "memory");
#endif
      if (n == H) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (c + 2 < NC) stage(c + 2, buf2, ib_cur);
        else stage(c + 2 - NC, buf2, ib_next);
      }
#pragma unroll
      for (int pp = 0; pp < 6; ++pp) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[TA[pp]], B[slot][TB[pp]], acc[n], 0, 0, 0);
    }
    buf = buf == kNBuf - 1 ? 0 : buf + 1;
    if constexpr (c + 1 < NC) {
      frag(std::integral_constant<int, c + 1>{});
    } else {
      epilogue(t);
      t += (int)gridDim.x;
      ib_cur = ib_next;
      ib_next = image_of(t + (int)gridDim.x);
      if (t < ntiles) gather(t);                          // (wave-uniform; the last tile's statistics feed a fragment nobody multiplies)
      frag(std::integral_constant<int, 0>{});
    }
  };
  while (true) {
    const bool last = t + (int)gridDim.x >= ntiles;       // (t changes inside the last step)
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{});
    if constexpr (NFB >= 2) {
      step(std::integral_constant<int, NFB >= 2 ? 4 : 0>{});
      step(std::integral_constant<int, NFB >= 2 ? 5 : 0>{});
      step(std::integral_constant<int, NFB >= 2 ? 6 : 0>{});
      step(std::integral_constant<int, NFB >= 2 ? 7 : 0>{});
    }
    if constexpr (NFB >= 3) {
      step(std::integral_constant<int, NFB >= 3 ? 8 : 0>{});
      step(std::integral_constant<int, NFB >= 3 ? 9 : 0>{});
      step(std::integral_constant<int, NFB >= 3 ? 10 : 0>{});
      step(std::integral_constant<int, NFB >= 3 ? 11 : 0>{});
    }
    if (last) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the copies issued for steps that do not exist
}

template <int NFB>
int launch(const DFArgs& g, int wgs, hipStream_t st) {
  const size_t lds = (size_t)kNBuf * kChunkV * 16 + (size_t)(3 * kNW) * sizeof(float);
  if (hipFuncSetAttribute((const void*)k_degree_fused<NFB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
  hipLaunchKernelGGL((k_degree_fused<NFB>), dim3((unsigned)wgs), dim3(kThreads), lds, st, g);
  return 0;
}

}  // namespace

extern "C" int degree_fused_f32(const degree_fused_args* p, void* stream) {
  if (!p) return PNA_E_INVALID;
  if (p->M == 0) return PNA_OK;
  if (!p->rowptr || !p->col || !p->x || !p->row_perm || !p->tile_image || !p->w_img || !p->y)
    return PNA_E_INVALID;
  if (p->F < 1 || p->F > 96 || p->N < 1 || p->N > 80)
    return PNA_E_INVALID;
  const int nfb = (p->F + 31) / 32;
  if (p->ldx < (p->F + 7) / 8 * 8 || p->ldx % 4 != 0 || ((uintptr_t)p->x & 15) != 0)
    return PNA_E_INVALID;
  if (p->M < 0 || p->M % (kWaves * 16) != 0 || (p->tile_rows != 64 && p->tile_rows != 128))
    return PNA_E_INVALID;
  if (p->ldy < p->N || (int64_t)p->ldy * 4 >= (1ll << 32) || (p->residual && (p->ld_res < p->N || (int64_t)p->ld_res * 4 >= (1ll << 32))) || p->image_stride <= 0)
    return PNA_E_INVALID;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    return PNA_E_NODEVICE;
  DFArgs g;
  g.rowptr = p->rowptr; g.col = p->col; g.x = p->x; g.ldx = p->ldx; g.F = p->F;
  g.perm = p->row_perm; g.tile_image = p->tile_image; g.w_img = (const unsigned char*)p->w_img; g.img_stride = p->image_stride;
  g.bias = p->bias; g.col_scale = p->col_scale; g.col_shift = p->col_shift; g.residual = p->residual; g.y = p->y;
  g.ldy = p->ldy; g.ld_res = p->ld_res; g.M = (int)p->M; g.N = p->N; g.relu = p->relu; g.slope = p->act_slope;
  g.tile_shift = (p->tile_rows == 128 ? 1 : 0) - (kWaves == 8 ? 1 : 0);
  if (g.tile_shift < 0) return PNA_E_INVALID;
  g.agg_dbg = g_agg_dbg; g.ld_dbg = g_ld_dbg;
  const int ntiles = (int)(p->M / (kWaves * 16));
  const int per_cu = p->workgroups_per_cu > 0 ? p->workgroups_per_cu : 2;
  const int wgs = ntiles < per_cu * cus ? ntiles : per_cu * cus;
  hipStream_t st = (hipStream_t)stream;
  const int rc = nfb == 1 ? launch<1>(g, wgs, st) : nfb == 2 ? launch<2>(g, wgs, st) : launch<3>(g, wgs, st);
  if (rc != 0) return PNA_E_LAUNCH;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return PNA_E_LAUNCH;
  return PNA_OK;
}

extern "C" void degree_fused_debug_agg(float* agg, int64_t ld) { g_agg_dbg = agg; g_ld_dbg = ld; }   // development only
