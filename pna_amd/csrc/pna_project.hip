// pna_project.hip -- node-level projection with a WIDE output: y = x W^T, x (M, K <= 128) fp32, W (N, K), N up to 512 columns.
// Implements pna_project_f32 of include/pna_amd.h.
//
// Where it is used: a PNALayer with T towers over the whole input (models/dgl/pna_layer.py:137-139) factorises its pretrans Linear to
// node level -- x_src = h [W_a,0 | .. | W_a,T-1]^T, T x 80 output columns from K = 75 inputs (pna_amd/functional.py::FusedMultiTowerCall).
// That product WRITES 1.6 GB from 0.3 GB at C3 and has 60 GFLOP: it is bound by its output stream.  The contraction kernels of this
// library cut N into 80-column blocks and read x once per block (5 x 0.3 GB on top of the 1.6: 1.07 / 1.21 ms), the library GEMM does no
// better (0.99 ms).  Here a workgroup keeps the WHOLE weight in LDS (K x N fp32: 126 KB at 75 x 400, one workgroup per CU), reads its 128
// rows of x once and writes every output row as whole 64-byte pieces.
//
// v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation: the result is an fp32 GEMM's up to the summation order), computed
// TRANSPOSED like the one-kernel layer's contraction: the weight fragment is the A operand (lane (i, g): column 16 n + i, k = g), the rows'
// the B operand (lane (i, g): row i, k = g), so lane (i, g) ends up with columns 16 n + 4 g .. + 4 of row i -- one 16-byte store.  K is
// consumed 16 at a time: lane (i, g) loads x[row i][16 c + 4 g .. + 4) with one dwordx4 and step s of the chunk multiplies physical
// k = 16 c + 4 g + s (a permutation of the summation order that both operands agree on); the row's last window slides back to end at K and is
// realigned (pna_x3::fix4): no read leaves a row of any pitch >= K, nothing is made of padding.
//
// Measured (1 M rows, K = 75, N = 400; tools/ubench/project_variants.sh, profiles/r06_project_variants.txt): 0.70 ms against the library
// GEMM's 1.03.  Its parts alone: the MFMA stream without stores 0.56-0.59 ms (without the LDS reads too: 0.53 -- 120 TFLOP/s of the
// 157 fp32 MFMA peak, the clock under this load), loads + stores without one MFMA 0.50 (a CU issues 16-byte stores at ~12 B/clk: 7.4 TB/s for
// the chip even into L2).  256 / 512 / 1024 threads, weights prefetched one K chunk ahead in registers, the windows requested at the top of a
// tile and waited for with the tile's 25 stores still in flight (a counted vmcnt): all within 2 % of each other.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pna_amd.h"
#include "pna_internal.h"
#include "pna_x3_split.h"

namespace {

using pna_x3::f4;
using pna_x3::f4u;
using pna_x3::fix4;

#ifndef PNA_PROJECT_BLOCK
#define PNA_PROJECT_BLOCK 512
#endif
constexpr int kBlock = PNA_PROJECT_BLOCK, kWaves = kBlock / 64, kRowsWG = 16 * kWaves;     // 2 wavefronts per SIMD: one's stores under the other's MFMAs

struct JArgs {
  const float* x; const float* w; float* y;
  long ldx, ldw, ldy;
  int M, K, N, NP;                                          // NP: LDS row pitch of the weight image (floats)
};

// NCH: 16-wide chunks of K (= ceil(K / 16): compile time, so that the row's windows are all in flight at once and nothing in the MFMA loop
// is conditional); NTT: 16-column tiles per pass (the accumulators live in registers; the LDS image holds whole passes, zero beyond N)
template <int NCH, int NTT>
__global__ __launch_bounds__(kBlock) void k_project(const JArgs g) {
  extern __shared__ float lds[];                            // [16 NCH][NP]: w^T, zero beyond K / N
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  constexpr int Kp = NCH * 16;
  for (int i = tid; i < Kp * g.NP; i += kBlock) lds[i] = 0.f;
  __syncthreads();
  for (int i0 = tid; i0 < g.N * g.K; i0 += 4 * kBlock) {    // consecutive threads: consecutive k of one column (coalesced reads of w)
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(i0 + u * kBlock, g.N * g.K - 1), n = i / g.K;
      v[u] = g.w[(long)n * g.ldw + (i - n * g.K)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(i0 + u * kBlock, g.N * g.K - 1), n = i / g.K;
      lds[(i - n * g.K) * g.NP + n] = v[u];
    }
  }
  __syncthreads();
  const int npass = (g.N + 16 * NTT - 1) / (16 * NTT);
  const long stride = (long)gridDim.x * kRowsWG;
  long row0 = ((long)blockIdx.x * kWaves + wave) * 16;
  if (row0 >= g.M) return;
  // the row's K values: one 16-byte window per chunk (at 16 c + 4 lg); the last chunk's slides back to end at K and is realigned.
  // Requested by hand (asm) and waited for by hand: s_waitcnt vmcnt counts loads and stores in ONE order, and the compiler's own wait for
  // windows requested at the top of a tile drains every store of that tile first (measured: the launch 0.70 ms, 0.49 of it without a
  // single MFMA).  Here the next tile's windows are requested before the LAST pass's MFMAs and taken after them, before its stores:
  // the wait leaves nothing younger than one pass of MFMAs (>= 1.3 us) outstanding.
  f4 raw[NCH];
  auto fetch = [&](long r0) {
    const float* const xr = g.x + min(r0 + li, (long)g.M - 1) * g.ldx;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float* const a = xr + max(0, min(16 * c + 4 * lg, g.K - 4));
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[c]) : "v"(a) : "memory");
    }
  };
  const float* const wl = lds + (size_t)(4 * lg) * g.NP + li;
  f4 xa[NCH];
  auto take = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      asm volatile("" : "+v"(raw[c]));                         // (the values exist from here on: nothing that reads them moves above the wait)
      xa[c] = c + 1 < NCH ? raw[c] : fix4(16 * c + 4 * lg, g.K, raw[c]);
    }
  };
  const int nfull = g.N >> 4, nrem = g.N & 15;
  f4 acc[NTT];
  auto mfmas = [&](int p) {
#pragma unroll
    for (int n = 0; n < NTT; ++n) acc[n] = (f4){0.f, 0.f, 0.f, 0.f};
    const float* const wp = wl + p * (16 * NTT);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float* const wr = wp + (size_t)(16 * c + s) * g.NP;
#pragma unroll
        for (int n = 0; n < NTT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[16 * n], xa[c][s], acc[n], 0, 0, 0);
      }
    }
  };
  auto stores = [&](int p, float* o, bool live) {
#pragma unroll
    for (int n = 0; n < NTT; ++n) {
      const int t = p * NTT + n;                               // (wave-uniform: whole tiles take one 16-byte store per lane)
      if (t < nfull) {
        if (live) { f4u w; w.v = acc[n]; *reinterpret_cast<f4u*>(o + t * 16) = w; }
      } else if (t == nfull && nrem && live) {
        for (int r = 0; r < 4; ++r)
          if (4 * lg + r < nrem) o[t * 16 + r] = acc[n][r];
      }
    }
  };
  fetch(row0);
  take();
  for (; row0 < g.M; row0 += stride) {
    float* const o = g.y + (row0 + li) * g.ldy + 4 * lg;
    const bool live = row0 + li < g.M;
    for (int p = 0; p + 1 < npass; ++p) {
      mfmas(p);
      stores(p, o, live);
    }
    fetch(row0 + stride);                                     // (clamped to the last row: never out of bounds, unused past the end)
    __builtin_amdgcn_sched_barrier(0);
    mfmas(npass - 1);
    __builtin_amdgcn_sched_barrier(0);
    f4 last[NTT];
#pragma unroll
    for (int n = 0; n < NTT; ++n) last[n] = acc[n];
    take();
#pragma unroll
    for (int n = 0; n < NTT; ++n) acc[n] = last[n];
    stores(npass - 1, o, live);
  }
}

// ---- the scaled form: y = [x W_0^T] + sum_s r_s (x W_s^T + beta_s), every block of the weight resident, ONE pass ---------------------
// The dense term of the multi-tower layer (pna_amd/functional.py::FusedMultiTowerCall.dense_term): the row's own features against the self
// panel and the S collapsed destination-term blocks, each scaled by the row's degree scaler.  A lane holds the same 4 columns of every block
// (tile j of block b = accumulator 5 b + j), so the combination is lane-local; beta sits behind the weight image in LDS.
struct SArgs {
  const float* x; const float* w; const float* scales; const float* beta; float* y;
  long ldx, ldw, lds_, ldb, ldy;
  int M, K, N, S, self;
};

constexpr int kPanel = 80, kTPP = kPanel / 16;              // columns (tiles) of one block in the image: N <= 80

template <int NCH, int NPAN>
__global__ __launch_bounds__(kBlock) void k_project_scaled(const SArgs g) {
  extern __shared__ float lds[];                            // [16 NCH][NP]: block b's w^T at columns 80 b; then beta [S][80]
  constexpr int NP = NPAN * kPanel + 4, Kp = NCH * 16, NT = NPAN * kTPP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  float* const lbeta = lds + Kp * NP;
  for (int i = tid; i < Kp * NP + 3 * kPanel; i += kBlock) lds[i] = 0.f;
  __syncthreads();
  const int KW = NPAN * g.K;                                // the weight's row: NPAN blocks of K
  for (int i0 = tid; i0 < g.N * KW; i0 += 4 * kBlock) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(i0 + u * kBlock, g.N * KW - 1), n = i / KW;
      v[u] = g.w[(long)n * g.ldw + (i - n * KW)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(i0 + u * kBlock, g.N * KW - 1), n = i / KW, kk = i - n * KW, b = kk / g.K;
      lds[(kk - b * g.K) * NP + b * kPanel + n] = v[u];
    }
  }
  if (g.beta)
    for (int i = tid; i < g.S * g.N; i += kBlock) lbeta[(i / g.N) * kPanel + i % g.N] = g.beta[(long)(i / g.N) * g.ldb + i % g.N];
  __syncthreads();
  const long stride = (long)gridDim.x * kRowsWG;
  long row0 = ((long)blockIdx.x * kWaves + wave) * 16;
  if (row0 >= g.M) return;
  f4 raw[NCH], xa[NCH];
  auto fetch = [&](long r0) {
    const float* const xr = g.x + min(r0 + li, (long)g.M - 1) * g.ldx;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float* const a = xr + max(0, min(16 * c + 4 * lg, g.K - 4));
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[c]) : "v"(a) : "memory");
    }
  };
  auto take = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      asm volatile("" : "+v"(raw[c]));
      xa[c] = c + 1 < NCH ? raw[c] : fix4(16 * c + 4 * lg, g.K, raw[c]);
    }
  };
  const float* const wl = lds + (size_t)(4 * lg) * NP + li;
  const int nfull = g.N >> 4, nrem = g.N & 15;
  fetch(row0);
  take();
  for (; row0 < g.M; row0 += stride) {
    const long row = min(row0 + li, (long)g.M - 1);
    const bool live = row0 + li < g.M;
    float rs[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (q < g.S) rs[q] = g.scales[row * g.lds_ + q];
    fetch(row0 + stride);
    __builtin_amdgcn_sched_barrier(0);
    f4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float* const wr = wl + (size_t)(16 * c + s) * NP;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[16 * t], xa[c][s], acc[t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // y = [block 0] + sum_q r_q (block self + q, + beta_q): in this order, every operation rounded once (no contraction)
    f4 out[kTPP];
#pragma unroll
    for (int j = 0; j < kTPP; ++j) {
      out[j] = g.self ? acc[j] : (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (q < NPAN && q < g.S) {                            // block self + q (S + self = NPAN: both candidates are compile-time indices)
          const f4 bq = *reinterpret_cast<const f4*>(lbeta + q * kPanel + 16 * j + 4 * lg);
          const f4 a = g.self ? acc[(q + 1 < NPAN ? q + 1 : NPAN - 1) * kTPP + j] : acc[q * kTPP + j];
          out[j] = out[j] + rs[q] * (a + bq);
        }
      }
    }
    take();
    if (live) {
      float* const o = g.y + row * g.ldy + 4 * lg;
#pragma unroll
      for (int j = 0; j < kTPP; ++j) {
        if (j < nfull) {
          f4u w; w.v = out[j]; *reinterpret_cast<f4u*>(o + j * 16) = w;
        } else if (j == nfull && nrem) {
          for (int r = 0; r < 4; ++r)
            if (4 * lg + r < nrem) o[j * 16 + r] = out[j][r];
        }
      }
    }
  }
}

template <int NCH, int NPAN>
hipError_t launch_scaled2(const SArgs& g, unsigned grid, size_t lds, hipStream_t stream) {
  auto* fn = k_project_scaled<NCH, NPAN>;
  hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kBlock), lds, stream, g);
  return hipGetLastError();
}

template <int NCH>
hipError_t launch_scaled(const SArgs& g, int npan, unsigned grid, size_t lds, hipStream_t stream) {
  switch (npan) {
    case 1: return launch_scaled2<NCH, 1>(g, grid, lds, stream);
    case 2: return launch_scaled2<NCH, 2>(g, grid, lds, stream);
    case 3: return launch_scaled2<NCH, 3>(g, grid, lds, stream);
    default:
      if constexpr (NCH <= 5) return launch_scaled2<NCH, 4>(g, grid, lds, stream);   // (four blocks' accumulators + six chunks of windows: over the register budget)
      return hipErrorInvalidValue;
  }
}

// ---- the grouped form: one weight per DEGREE GROUP, rows through the degree plan's permutation ------------------------------------------
// y[node, 0:N] = x[node, 0:K] W_g^T for node = row_perm[v], g = tile_group[v / 128]: the backward's d agg = gy W_D^T of a PNASimpleLayer in
// training (pna_amd/autograd.py::SimpleLayerPlanFn.backward) -- every scaler is a function of the in-degree alone, so the three scaler blocks
// of a degree group's rows collapse into ONE 75 x 300 weight (a third of the multiply-adds of the three-block contraction, which took
// 1.11 ms at C3).  A workgroup walks a contiguous piece of the tiles and refills its LDS image only when the group changes: the caller lists
// the tiles sorted by group (tiles are independent: any order of (row_perm, tile_group) pairs is the same product).
struct GArgs {
  const float* x; const float* w; float* y;
  const int* perm; const int* tile_group;
  long ldx, ldy, wstride;                                    // wstride: floats between two groups' (N, K) matrices
  int ntiles, K, N, NP;
};

template <int NCH, int NTT>
__global__ __launch_bounds__(kBlock) void k_project_grouped(const GArgs g) {
  static_assert(kBlock == 512, "one 128-row tile of the degree plan per workgroup pass: 8 wavefronts x 16 rows");
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  constexpr int Kp = NCH * 16;
  const int per = (g.ntiles + gridDim.x - 1) / gridDim.x;
  const int s0 = blockIdx.x * per, s1 = min(s0 + per, g.ntiles);
  if (s0 >= s1) return;                                     // (whole workgroups: no barrier is left waiting)
  for (int i = tid; i < Kp * g.NP; i += kBlock) lds[i] = 0.f;
  const int npass = (g.N + 16 * NTT - 1) / (16 * NTT);
  const int nfull = g.N >> 4, nrem = g.N & 15;
  auto fill = [&](int grp) {                                // (between two barriers: nobody reads the image meanwhile)
    const float* const w = g.w + (size_t)grp * g.wstride;
    for (int i0 = tid; i0 < g.N * g.K; i0 += 4 * kBlock) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = w[min(i0 + u * kBlock, g.N * g.K - 1)];      // (group matrices are dense (N, K) rows)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = min(i0 + u * kBlock, g.N * g.K - 1), n = i / g.K;
        lds[(i - n * g.K) * g.NP + n] = v[u];
      }
    }
  };
  // node of this lane's row in tile s (-1: padding) and the tile's group, requested by hand TWO / ONE tiles ahead as VECTOR loads: they land by
  // the `vmcnt(0)` of the tile before.  (As scalar loads in the loop they would sit in lgkmcnt in front of every wait for an LDS read.)
  int node_a, node_b, grp_v;                                  // tiles s + 1 and s + 2 while tile s is multiplied; group of tile s + 1
  auto ask_node = [&](int s, int& dst) {
    const int* const p = g.perm + (size_t)min(s, s1 - 1) * 128 + wave * 16 + li;
    asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
  };
  auto ask_group = [&](int s) {
    const int* const p = g.tile_group + min(s, s1 - 1);
    asm volatile("global_load_dword %0, %1, off" : "=v"(grp_v) : "v"(p) : "memory");
  };
  f4 raw[NCH], xa[NCH];
  auto fetch = [&](int node) {
    const float* const xr = g.x + (long)max(node, 0) * g.ldx;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float* const a = xr + max(0, min(16 * c + 4 * lg, g.K - 4));
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[c]) : "v"(a) : "memory");
    }
  };
  auto take = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      asm volatile("" : "+v"(raw[c]));
      xa[c] = c + 1 < NCH ? raw[c] : fix4(16 * c + 4 * lg, g.K, raw[c]);
    }
  };
  const float* const wl = lds + (size_t)(4 * lg) * g.NP + li;
  f4 acc[NTT];
  auto mfmas = [&](int p) {
#pragma unroll
    for (int n = 0; n < NTT; ++n) acc[n] = (f4){0.f, 0.f, 0.f, 0.f};
    const float* const wp = wl + p * (16 * NTT);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float* const wr = wp + (size_t)(16 * c + s) * g.NP;
#pragma unroll
        for (int n = 0; n < NTT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[16 * n], xa[c][s], acc[n], 0, 0, 0);
      }
    }
  };
  auto stores = [&](int p, float* o, bool live) {
#pragma unroll
    for (int n = 0; n < NTT; ++n) {
      const int t = p * NTT + n;
      if (t < nfull) {
        if (live) { f4u w; w.v = acc[n]; *reinterpret_cast<f4u*>(o + t * 16) = w; }
      } else if (t == nfull && nrem && live) {
        for (int r = 0; r < 4; ++r)
          if (4 * lg + r < nrem) o[t * 16 + r] = acc[n][r];
      }
    }
  };
  int node;
  ask_node(s0, node);
  ask_group(s0);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(node), "+v"(grp_v) : : "memory");
  int grp = __builtin_amdgcn_readfirstlane(grp_v);
  ask_node(s0 + 1, node_a);
  ask_group(s0 + 1);
  fetch(node);
  take();                                                     // (node_a and the next group have landed too)
  asm volatile("" : "+v"(node_a), "+v"(grp_v));
  int cur = -1;
  for (int s = s0; s < s1; ++s) {
    if (grp != cur) {                                         // (workgroup-uniform)
      __syncthreads();
      fill(grp);
      __syncthreads();
      cur = grp;
    }
    grp = __builtin_amdgcn_readfirstlane(grp_v);              // tile s + 1's
    float* const o = g.y + (long)max(node, 0) * g.ldy + 4 * lg;
    const bool live = node >= 0;
    for (int p = 0; p + 1 < npass; ++p) {
      mfmas(p);
      stores(p, o, live);
    }
    ask_node(s + 2, node_b);
    ask_group(s + 2);
    fetch(node_a);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(npass - 1);
    __builtin_amdgcn_sched_barrier(0);
    f4 last[NTT];
#pragma unroll
    for (int n = 0; n < NTT; ++n) last[n] = acc[n];
    take();
    asm volatile("" : "+v"(node_b), "+v"(grp_v));
#pragma unroll
    for (int n = 0; n < NTT; ++n) acc[n] = last[n];
    stores(npass - 1, o, live);
    node = node_a;
    node_a = node_b;
  }
}

template <int NCH>
hipError_t launch_grouped(const GArgs& g, unsigned grid, size_t lds, hipStream_t stream) {
  auto* fn = k_project_grouped<NCH, 5>;
  hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kBlock), lds, stream, g);
  return hipGetLastError();
}

constexpr int kNTT = 5;

template <int NCH>
hipError_t launch_project(const JArgs& g, unsigned grid, size_t lds, hipStream_t stream) {
  auto* fn = k_project<NCH, kNTT>;
  hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kBlock), lds, stream, g);
  return hipGetLastError();
}

}  // namespace

extern "C" int pna_project_f32(const float* x, int64_t ldx, int64_t M, int32_t K, const float* w, int64_t ldw, int32_t N, float* y, int64_t ldy,
                               pna_stream_t stream) {
  if (M == 0) return PNA_OK;
  if (!x || !w || !y || M < 0 || M >= (1ll << 31) || K < 4 || K > 128 || N < 1 || N > 512 || ldx < K || ldw < K || ldy < N ||
      ((uintptr_t)x & 3) || ((uintptr_t)y & 3))
    return pna_set_error(PNA_E_INVALID, "pna_project_f32: 4 <= K <= 128, 1 <= N <= 512, ldx >= K, ldw >= K, ldy >= N, 4-byte aligned x / y");
  const int nch = (K + 15) / 16, Kp = nch * 16;
  // pitch: whole passes of 16-column tiles + 4: one read takes 16 consecutive columns of the rows k, k + 4, k + 8, k + 12 (the four lane
  // groups), which 4 x pitch = 16 (mod 64) floats apart puts on 64 different banks
  const int NP = (N + 16 * kNTT - 1) / (16 * kNTT) * (16 * kNTT) + 4;
  const size_t lds = (size_t)Kp * NP * sizeof(float);
  if (lds > 160 * 1024)
    return pna_set_error(PNA_E_INVALID, "pna_project_f32: the weight (16 ceil(K / 16) x (80 ceil(N / 80) + 4) floats) must fit 160 KB of LDS");
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    return pna_set_error(PNA_E_NODEVICE, "pna_project_f32: no device");
  JArgs g;
  g.x = x; g.w = w; g.y = y; g.ldx = ldx; g.ldw = ldw; g.ldy = ldy; g.M = (int)M; g.K = K; g.N = N; g.NP = NP;
  const long wgs_needed = (M + kRowsWG - 1) / kRowsWG;
  const unsigned grid = (unsigned)(wgs_needed < cus ? wgs_needed : cus);      // persistent: the weight image is loaded once per workgroup
  hipError_t e = hipSuccess;
  switch (nch) {
    case 1: e = launch_project<1>(g, grid, lds, (hipStream_t)stream); break;
    case 2: e = launch_project<2>(g, grid, lds, (hipStream_t)stream); break;
    case 3: e = launch_project<3>(g, grid, lds, (hipStream_t)stream); break;
    case 4: e = launch_project<4>(g, grid, lds, (hipStream_t)stream); break;
    case 5: e = launch_project<5>(g, grid, lds, (hipStream_t)stream); break;
    case 6: e = launch_project<6>(g, grid, lds, (hipStream_t)stream); break;
    case 7: e = launch_project<7>(g, grid, lds, (hipStream_t)stream); break;
    default: e = launch_project<8>(g, grid, lds, (hipStream_t)stream); break;
  }
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}

extern "C" int pna_project_scaled_f32(const float* x, int64_t ldx, int64_t M, int32_t K, const float* w, int64_t ldw, int32_t N, int32_t n_scaled,
                                      int32_t self_block, const float* scales, int64_t ld_scales, const float* beta, int64_t ld_beta, float* y,
                                      int64_t ldy, pna_stream_t stream) {
  if (M == 0) return PNA_OK;
  const int npan = n_scaled + (self_block ? 1 : 0);
  if (!x || !w || !y || M < 0 || M >= (1ll << 31) || K < 4 || K > 128 || N < 1 || N > kPanel || n_scaled < 0 || n_scaled > 3 || npan < 1 || ldx < K ||
      ldw < (int64_t)npan * K || ldy < N || (n_scaled && (!scales || ld_scales < n_scaled)) || (beta && ld_beta < N) || ((uintptr_t)x & 3) ||
      ((uintptr_t)y & 3))
    return pna_set_error(PNA_E_INVALID,
                         "pna_project_scaled_f32: 4 <= K <= 128, 1 <= N <= 80, n_scaled <= 3, at least one block, ldx >= K, ldw >= blocks x K, ldy >= N, "
                         "ld_scales >= n_scaled, ld_beta >= N");
  const int nch = (K + 15) / 16, Kp = nch * 16;
  if (npan == 4 && nch > 5) return pna_set_error(PNA_E_INVALID, "pna_project_scaled_f32: four blocks need K <= 80 (the accumulators' register budget)");
  const size_t lds = ((size_t)Kp * (npan * kPanel + 4) + 3 * kPanel) * sizeof(float);
  if (lds > 160 * 1024)
    return pna_set_error(PNA_E_INVALID, "pna_project_scaled_f32: the weight (16 ceil(K / 16) x (80 blocks + 4) floats) must fit 160 KB of LDS");
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    return pna_set_error(PNA_E_NODEVICE, "pna_project_scaled_f32: no device");
  SArgs g;
  g.x = x; g.w = w; g.scales = scales; g.beta = beta; g.y = y;
  g.ldx = ldx; g.ldw = ldw; g.lds_ = ld_scales; g.ldb = ld_beta; g.ldy = ldy;
  g.M = (int)M; g.K = K; g.N = N; g.S = n_scaled; g.self = self_block ? 1 : 0;
  const long wgs_needed = (M + kRowsWG - 1) / kRowsWG;
  const unsigned grid = (unsigned)(wgs_needed < cus ? wgs_needed : cus);
  hipError_t e = hipSuccess;
  switch (nch) {
    case 1: e = launch_scaled<1>(g, npan, grid, lds, (hipStream_t)stream); break;
    case 2: e = launch_scaled<2>(g, npan, grid, lds, (hipStream_t)stream); break;
    case 3: e = launch_scaled<3>(g, npan, grid, lds, (hipStream_t)stream); break;
    case 4: e = launch_scaled<4>(g, npan, grid, lds, (hipStream_t)stream); break;
    case 5: e = launch_scaled<5>(g, npan, grid, lds, (hipStream_t)stream); break;
    case 6: e = launch_scaled<6>(g, npan, grid, lds, (hipStream_t)stream); break;
    case 7: e = launch_scaled<7>(g, npan, grid, lds, (hipStream_t)stream); break;
    default: e = launch_scaled<8>(g, npan, grid, lds, (hipStream_t)stream); break;
  }
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}

extern "C" int pna_project_grouped_f32(const float* x, int64_t ldx, int64_t x_rows, int32_t K, const float* w_groups, int64_t group_stride,
                                       int32_t n_groups, int32_t N, const int32_t* row_perm, int64_t M, const int32_t* tile_group,
                                       float* y, int64_t ldy, pna_stream_t stream) {
  if (M == 0) return PNA_OK;
  if (!x || !w_groups || !y || !row_perm || !tile_group || M < 0 || M % 128 || M >= (1ll << 31) || x_rows < 1 || K < 4 || K > 128 || N < 1 ||
      N > 512 || n_groups < 1 || group_stride < (int64_t)N * K || ldx < K || ldy < N || ((uintptr_t)x & 3) || ((uintptr_t)y & 3))
    return pna_set_error(PNA_E_INVALID,
                         "pna_project_grouped_f32: M a multiple of 128, 4 <= K <= 128, 1 <= N <= 512, group_stride >= N K, ldx >= K, ldy >= N");
  const int nch = (K + 15) / 16, Kp = nch * 16;
  const int NP = (N + 79) / 80 * 80 + 4;
  const size_t lds = (size_t)Kp * NP * sizeof(float);
  if (lds > 160 * 1024)
    return pna_set_error(PNA_E_INVALID, "pna_project_grouped_f32: a group's weight (16 ceil(K / 16) x (80 ceil(N / 80) + 4) floats) must fit 160 KB of LDS");
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    return pna_set_error(PNA_E_NODEVICE, "pna_project_grouped_f32: no device");
  GArgs g;
  g.x = x; g.w = w_groups; g.y = y; g.perm = row_perm; g.tile_group = tile_group;
  g.ldx = ldx; g.ldy = ldy; g.wstride = group_stride; g.ntiles = (int)(M / 128); g.K = K; g.N = N; g.NP = NP;
  const unsigned grid = (unsigned)(g.ntiles < cus ? g.ntiles : cus);
  hipError_t e = hipSuccess;
  switch (nch) {
    case 1: e = launch_grouped<1>(g, grid, lds, (hipStream_t)stream); break;
    case 2: e = launch_grouped<2>(g, grid, lds, (hipStream_t)stream); break;
    case 3: e = launch_grouped<3>(g, grid, lds, (hipStream_t)stream); break;
    case 4: e = launch_grouped<4>(g, grid, lds, (hipStream_t)stream); break;
    case 5: e = launch_grouped<5>(g, grid, lds, (hipStream_t)stream); break;
    case 6: e = launch_grouped<6>(g, grid, lds, (hipStream_t)stream); break;
    case 7: e = launch_grouped<7>(g, grid, lds, (hipStream_t)stream); break;
    default: e = launch_grouped<8>(g, grid, lds, (hipStream_t)stream); break;
  }
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}
