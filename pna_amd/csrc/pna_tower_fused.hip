// pna_tower_fused.hip -- the tower layer of the molecule-sized batches (BASELINE.json configs[1]: 128 ZINC graphs, ~3 k nodes,
// ~6 k edges) as ONE C call and TWO launches for gfx950.  Implements pna_small_linear_f32 / pna_tower_layer_f32 of
// include/pna_amd.h (models/dgl/pna_layer.py:35-75,:133-148 in eval mode).
//
// Why a second code path: at this size every kernel of the large-graph path (projection GEMM, gather, batched tower
// contraction, mixing contraction) runs for 10-30 us, of which most is launch latency, barriers and exposed load latency, and
// the host issues four launches plus the tensor glue between them -- the layer is bound by LATENCY, not by HBM or MFMA.
// So here the work is cut by destination ROWS instead of by operator: after the node-level projection (launch 1, which every
// row's neighbours need and therefore has to be complete first) a workgroup owns 16 destination rows and does everything else
// for them without leaving the CU:
//     gather + mean|max|min|std of the 16 rows, all towers            -> LDS   (16 x T*4*Fi floats, <= 128 KB)
//     tower contraction  z_t = b_t + W_h,t h + sum_s scale_s (W_s,t a_t), graph-norm, eval BatchNorm   -> LDS (16 x T*Fo)
//     mixing network     y = h + LeakyReLU(W_mix [z_0 .. z_T-1] + b_mix)                               -> HBM
// The contractions run on v_mfma_f32_16x16x4_f32 (exact fp32 products): a 16-row tile is the M of one MFMA; one (tower,
// 16-column tile) "unit" is owned by one wavefront (or by 2/4/8 wavefronts splitting K when there are fewer units than
// wavefronts, their partial tiles summed in wavefront order through LDS: deterministic).  Weights are read from L2 in MFMA
// fragment order (pna_small_pack_f32 / pna_tower_post_pack_f32 images: one coalesced dwordx4 per lane per 16 k), four fragments
// ahead of their use; nothing else of the layer touches HBM more than once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "pna_amd.h"
#include "pna_internal.h"
#include "pna_rowstats.h"
#ifdef PNA_AMD_EXPERIMENTS
#include <stdlib.h>
// phase timers (tools/tf_timers.py): every wavefront's lane 0 stores clock64() / wall_clock64() at the marks below
#define TF_MARK(k) do { if (g.dbg && lane == 0) { unsigned long long* o_ = g.dbg + ((size_t)blockIdx.x * kWaves + wave) * 16; o_[k] = clock64(); if ((k) == 0) o_[14] = wall_clock64(); o_[15] = wall_clock64(); } } while (0)
#else
#define TF_MARK(k) do { } while (0)
#endif

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kRows = 16;        // destination rows of a workgroup = the M of one MFMA tile
constexpr int kMaxLds = 160 * 1024;

__host__ __device__ constexpr int quads(int k) { return (k + 15) / 16; }
__host__ __device__ constexpr int pitch_of(int q) { return q * 16 + 4; }   // LDS row pitch (floats) of an operand of q quads

// ---- weight images ----------------------------------------------------------------------------------------------------
// Fragment (n tile nt, quad q) of a weight block W[N][K]: 64 lanes x 4 floats, lane (i = l & 15, g = l >> 4) holds
// W[nt*16 + i][16 q + 4 g .. + 3] (0 outside N / K) -- the B operand of the quad's four MFMAs (k-step t multiplies physical
// k = 16 q + 4 g + t, the same permutation the A operand's ds_read_b128 applies).  frag(nt, q) lives at
// img + (nt*nt_stride + q*q_stride) * 256 floats: strides let several blocks interleave in one image.
__global__ void k_small_pack(const float* w, long ldw, int N, int K, int NT, int Q, int q_stride, int nt_stride, float* img) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)NT * Q * 256) return;
  const int i = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
  const int fq = (int)(idx >> 8), q = fq % Q, nt = fq / Q;
  const int n = nt * 16 + (lane & 15), k = 16 * q + 4 * (lane >> 4) + i;
  img[((size_t)nt * nt_stride + (size_t)q * q_stride) * 256 + lane * 4 + i] = (n < N && k < K) ? w[(size_t)n * ldw + k] : 0.f;
}

int launch_pack(const float* w, long ldw, int N, int K, int Q, int q_stride, int nt_stride, float* img, hipStream_t st) {
  const int NT = (N + 15) / 16;                           // Q quads are written (zeros past K)
  const long total = (long)NT * Q * 256;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(k_small_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, ldw, N, K, NT, Q, q_stride, nt_stride, img);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- MFMA helpers -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ void quad_fma(f4& acc, const f4 a, const f4 b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
}

// One fragment = one global_load_dwordx4 per lane, issued through inline asm so that the wait can be COUNTED: hipcc's own
// s_waitcnt placement drains the whole queue (vmcnt(0)) at every loop header, which turns a ring of D loads in flight into
// D loads, a full L2 round trip, D quads of MFMA, the next round trip, ... (measured: 18.5 k cycles for 7.5 k cycles of MFMA).
// Rules kept (cdna_hip_programming.md 5.7; pna_segreduce.hip does the same): a ring slot has ONE asm site that writes it and is
// only read after the "+v" anchor that follows its counted wait; every position issues exactly one load (indices past the end
// are clamped), so the number of younger loads at a wait is a constant; the ring is drained before its registers die.
__device__ __forceinline__ void aload128(f4& dst, const void* base, unsigned voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}

// y tile (16 rows x 16 columns) = A[16 x 16 Q] . W_nt for the column tiles nt = wave, wave + 8, ... < NT of one Linear
// (image frag(nt, q) at (nt*Q + q)*64 float4), as ONE fragment stream per wavefront: the ring of D loads runs across tile
// boundaries, so a wavefront with several tiles pays the L2 latency once, not once per tile.  epi(nt, acc) stores a tile
// (its stores count in vmcnt too: they only make the counted wait conservative, loads return in order among themselves).
struct Stream {
  const void* img; int Q, L, wave, lane;
  __device__ __forceinline__ Stream(const void* img_, int Q_, int NT, int wave_, int lane_) : img(img_), Q(Q_), wave(wave_), lane(lane_) {
    L = (NT > wave ? (NT - wave + kWaves - 1) / kWaves : 0) * Q;
  }
};

template <int D, typename Epi>
__device__ __forceinline__ void stream_tiles(const Stream& st, const float* a_lds, Epi epi) {
  if (st.L == 0) return;
  f4 b[D];
  f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
  int q = 0, i = 0;                                       // consuming cursor: quad within the tile, tile number of this wavefront
  int ql = 0, jl = 0;                                     // loading cursor, D fragments ahead (stays on the last fragment at the end)
  unsigned off = (unsigned)(st.wave * st.Q * 64 + st.lane) * 16u;
  const unsigned tile_skip = (unsigned)((kWaves - 1) * st.Q * 64) * 16u;    // from a tile's last fragment to the next tile's first, less 1 KB
  for (int j0 = -D; j0 < st.L; j0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int j = j0 + d;
      if (j >= 0) {
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(D - 1) : "memory");     // D - 1 younger loads: fragment j has landed
        asm volatile("" : "+v"(b[d]));
        if (j < st.L) {
          quad_fma(acc, *reinterpret_cast<const f4*>(a_lds + q * 16), b[d]);
          if (++q == st.Q) {
            epi(st.wave + kWaves * i, acc);
            acc = (f4){0.f, 0.f, 0.f, 0.f};
            q = 0; ++i;
          }
        }
      }
      aload128(b[d], st.img, off);
      if (jl + 1 < st.L) {
        ++jl; off += 1024u;
        if (++ql == st.Q) { ql = 0; off += tile_skip; }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" : : : "memory");      // the clamped tail loads still target b[]: drain before it dies
}

// ---- launch 1 (and a general small-M Linear): y = act(x W^T + b) (+ residual) --------------------------------------------
struct LArgs {
  const float* x; long ldx;
  int M, K, N;
  const float* img;        // pna_small_pack_f32 image: frag(nt, q) at (nt*Q + q)*256
  const float* bias;
  int act; float slope;    // 0 none, 1 ReLU, 2 LeakyReLU
  const float* residual; long ld_res;
  float* y; long ldy;
#ifdef PNA_AMD_EXPERIMENTS
  unsigned long long* dbg;
#endif
};

__device__ __forceinline__ float activate(float v, int act, float slope) {
  if (act == 0) return v;
  return v > 0.f ? v : (v != v ? v : (act == 2 ? slope * v : 0.f));
}

__global__ __launch_bounds__(kThreads) void k_small_linear(const LArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int Q = quads(g.K), P = pitch_of(Q), NT = (g.N + 15) / 16;
  const int r0 = blockIdx.x * kRows;
  float* const BL = lds + kRows * P;                      // [NT*16] bias (0 past N)
  TF_MARK(0);
  const Stream st(g.img, Q, NT, wave, lane);
  for (int i = tid; i < kRows * P; i += kThreads) {
    const int r = i / P, k = i - r * P;
    lds[i] = (r0 + r < g.M && k < g.K) ? g.x[(size_t)(r0 + r) * g.ldx + k] : 0.f;
  }
  for (int i = tid; i < NT * 16; i += kThreads) BL[i] = (g.bias && i < g.N) ? g.bias[i] : 0.f;
  __syncthreads();
  TF_MARK(1);
  stream_tiles<10>(st, lds + li * P + 4 * lg, [&](int nt, const f4 acc) __attribute__((always_inline)) {
    const int n = nt * 16 + li;
    if (n < g.N) {
      const float bn = BL[n];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = r0 + 4 * lg + i;
        if (row < g.M) {
          float v = activate(acc[i] + bn, g.act, g.slope);
          if (g.residual) v = g.residual[(size_t)row * g.ld_res + n] + v;
          g.y[(size_t)row * g.ldy + n] = v;
        }
      }
    }
  });
  TF_MARK(2);
}

// ---- launch 2: everything after the projection, per 16 destination rows ---------------------------------------------------
struct TArgs {
  const int32_t* rowptr; const int32_t* col;
  int V, T, Fi, Fo, TG;          // TG: towers gathered per pass (T when the aggregate tile fits the LDS)
  const float* xcat; long ldx;   // (V, 2*T*Fi): [W_a h | W_b h + b] of every tower (launch 1)
  const float* h; long ldh; int h_stride;   // tower t's own features: h[:, t*h_stride .. + Fi)  (0: all towers see h whole)
  const float* scale[3];         // per-row scaler factors (NULL = identity)
  const float* post_img;         // pna_tower_post_pack_f32 images, tower after tower
  const float* post_bias;        // [T*Fo]
  const float* row_post;         // [V] or NULL
  const float* col_scale; const float* col_shift;   // [T*Fo] or NULL
  const float* mix_img; const float* mix_bias;      // mixing Linear(T*Fo -> No) image / bias; NULL: y = the towers' concatenation
  int No, mix_act; float mix_slope;
  const float* residual; long ld_res;
  float* y; long ldy;
  int no_self;                   // the own-features operand tile is zeros (no [h] panel in this posttrans)
  const int32_t* etype; const float* etab; long ldet; int n_types;   // edge features that are <= 4 types: type per CSR edge, (n_types, T*Fi) table of W_e . ef
#ifdef PNA_AMD_EXPERIMENTS
  unsigned long long* dbg;
#endif
};

constexpr int kMaxCh = 6;      // 64-column chunks of a gathered row handled at once by a wavefront (more: another pass)
constexpr int kEU = 4;         // edges of a row whose gathers are issued together (a molecule's atom has <= 4 bonds: one round trip)
constexpr int kMaxItems = 16;  // (unit, K part) items of one tower group whose partial tiles fit the RED buffer
constexpr int kRing = 12;      // fragments in flight per wavefront in the tower contraction (a multiple of S + 1 for S = 1, 2, 3)

// The tower contraction of one wavefront as ONE stream of "super-steps" over its items (unit = (tower, 16-column tile), K part kp
// of KS): super-step q = kp, kp + KS, ... < QA of an item is S + 1 consecutive fragments of the unit's image -- the S scaler
// blocks' quad q, then the own-features block's quad q (all zero for q >= QH: 4 Fi >= Fi, so the aggregate's quads cover
// them).  Two cursors walk the stream: the loading one kRing / (S + 1) super-steps ahead of the consuming one.
template <int S>
struct StepCursor {
  int it, q;                        // item (stays on the last super-step of the last item at the end), quad
  int kp, tl, nt;                   // of the current item
  unsigned base;                    // byte offset of the unit's image
  int KS, NTo, QA, t0, upt_b, n_items;
  __device__ __forceinline__ void set_item(int item) {
    it = item;
    const int u = it / KS;
    kp = it - u * KS; tl = u / NTo; nt = u - tl * NTo;
    q = kp;                                                // KS <= QA: every item has a super-step
    base = (unsigned)(((t0 + tl) * NTo + nt) * upt_b);
  }
  __device__ __forceinline__ unsigned voff(int lane) const { return base + (unsigned)(q * (S + 1) * 64 + lane) * 16u; }
  __device__ __forceinline__ bool last() const { return q + KS >= QA; }
  __device__ __forceinline__ void advance() {
    if (q + KS < QA) q += KS;
    else if (it + kWaves < n_items) set_item(it + kWaves);
  }
};

template <int S>
__global__ __launch_bounds__(kThreads) void k_tower_rows(const TArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int T = g.T, Fi = g.Fi, Fo = g.Fo, TFi = T * Fi, TFo = T * Fo;
  const int QA = quads(4 * Fi), QH = quads(Fi), QM = quads(TFo), NTo = (Fo + 15) / 16, NTm = (g.No + 15) / 16;
  const int PA = pitch_of(g.TG * QA), PH = pitch_of(QH), PM = pitch_of(QM);
  const int TH = g.h_stride ? T : 1;
  // LDS carve-up (floats)
  float* const A = lds;                                   // [16][PA]   aggregate tile of the current tower group
  float* const HL = A + kRows * PA;                       // [TH][16][PH] own features, per tower when the input is divided
  float* const ZC = HL + TH * kRows * PH;                 // [16][PM]   towers' outputs, concatenated (the mixing network's input)
  float* const RED = ZC + kRows * PM;                     // [16][256]  partial tiles of the items that split one unit's K
  float* const CB = RED + kMaxItems * 256;                // [3][QM*16] posttrans bias | BatchNorm scale | shift per output column
  float* const MB = CB + 3 * QM * 16;                     // [NTm*16]   mixing bias
  float* const SC = MB + NTm * 16;                        // [4][16]    scale_0..2 | row_post of the tile's rows
  const int r0 = blockIdx.x * kRows;
  const int nrows = min(kRows, g.V - r0);
  TF_MARK(0);
  // ---- per-tile operands that do not depend on the graph: own features, row scales, column constants; and zeros in the K
  // padding columns of the three MFMA operand tiles (they multiply zero weights, but 0 * NaN garbage would not be 0).  Rows
  // past the matrix's end (last tile) stay garbage: an MFMA row only feeds its own output row, which is never stored ----
  for (int i = tid; i < TH * kRows * Fi; i += kThreads) {
    const int t = i / (kRows * Fi), j = i - t * kRows * Fi, r = j / Fi, k = j - r * Fi;
    HL[(t * kRows + r) * PH + k] = (r < nrows && !g.no_self) ? g.h[(size_t)(r0 + r) * g.ldh + t * g.h_stride + k] : 0.f;
  }
  {
    const int pa = QA * 16 - 4 * Fi, ph = QH * 16 - Fi, pm = QM * 16 - TFo;
    for (int i = tid; i < kRows * g.TG * pa; i += kThreads) {
      const int r = i / (g.TG * pa), j = i - r * g.TG * pa, tl = j / pa, k = j - tl * pa;
      A[r * PA + tl * QA * 16 + 4 * Fi + k] = 0.f;
    }
    for (int i = tid; i < TH * kRows * ph; i += kThreads) HL[(i / ph) * PH + Fi + i % ph] = 0.f;
    for (int i = tid; i < kRows * pm; i += kThreads) ZC[(i / pm) * PM + TFo + i % pm] = 0.f;
  }
  if (tid < 4 * kRows) {
    const int s = tid >> 4, r = tid & 15;
    const float* p = s < 3 ? (s < S ? g.scale[s] : nullptr) : g.row_post;
    SC[tid] = (p && r < nrows) ? p[r0 + r] : 1.f;
  }
  for (int i = tid; i < QM * 16; i += kThreads) {
    const bool in = i < TFo;
    CB[i] = (in && g.post_bias) ? g.post_bias[i] : 0.f;
    CB[QM * 16 + i] = (in && g.col_scale) ? g.col_scale[i] : 1.f;
    CB[2 * QM * 16 + i] = (in && g.col_scale) ? g.col_shift[i] : 0.f;
  }
  for (int i = tid; i < NTm * 16; i += kThreads) MB[i] = (g.mix_bias && i < g.No) ? g.mix_bias[i] : 0.f;
  TF_MARK(1);

  const int upt = QA * (S + 1) * 64;                      // float4 per unit image: QA super-steps of S + 1 fragments
  for (int t0 = 0; t0 < T; t0 += g.TG) {
    const int tg = min(g.TG, T - t0), CF = tg * Fi, c0 = t0 * Fi;
    TF_MARK(2);
    // ---- gather + reduce, wavefront-local (no barrier before it): a wavefront owns rows (wave, wave + 8) and walks them
    // together -- rowptr, then the rows' source ids (one coalesced load per row, broadcast by v_readlane), then the gathers.
    // A lane owns columns lane, lane + 64, ... of the group's CF columns and keeps their running sums in registers, so the
    // loads of ALL its columns for two edges of both rows are in flight at once: the phase is three dependent round trips
    // to memory and nothing else (~2 edges per row in a molecule) ----
    for (int rb = wave; rb < nrows; rb += 2 * kWaves) {
      const bool hasB = rb + kWaves < nrows;
      const int rw[2] = {rb, hasB ? rb + kWaves : rb};
      const int* const rp = g.rowptr + r0;
      const int eb[2] = {rp[rw[0]], rp[rw[1]]};
      const int deg[2] = {rp[rw[0] + 1] - eb[0], hasB ? rp[rw[1] + 1] - eb[1] : 0};   // (no second row: no edges)
      const int maxdeg = max(deg[0], deg[1]);
      const float* const xs = g.xcat + c0;
      for (int cb = 0; cb < CF; cb += 64 * kMaxCh) {
        float s[2][kMaxCh], q[2][kMaxCh], mx[2][kMaxCh], mn[2][kMaxCh], dt[2][kMaxCh];
        float et[4][kMaxCh];                              // the lane's columns of every edge type's term (edge-feature layers only)
        int cc[kMaxCh];
#pragma unroll
        for (int j = 0; j < kMaxCh; ++j) {
          cc[j] = min(cb + 64 * j + lane, CF - 1);        // (lanes past the last column redo it; their result is not stored)
          if (g.etype) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
              et[t][j] = (t < g.n_types && cb + 64 * j < CF) ? g.etab[(size_t)t * g.ldet + c0 + cc[j]] : 0.f;
          }
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            s[r][j] = 0.f; q[r][j] = 0.f; mx[r][j] = -INFINITY; mn[r][j] = INFINITY;
            dt[r][j] = (cb + 64 * j < CF && !g.no_self) ? g.xcat[(size_t)(r0 + rw[r]) * g.ldx + TFi + c0 + cc[j]] : 0.f;
          }
        }
        for (int base = 0; base < maxdeg; base += 64) {
          int myid[2], myty[2], n[2];
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            n[r] = min(64, deg[r] - base);
            myid[r] = lane < n[r] ? g.col[eb[r] + base + lane] : 0;
            myty[r] = (g.etype && lane < n[r]) ? g.etype[eb[r] + base + lane] : 0;      // (the same round trip as the ids)
          }
          for (int k = 0; k < max(n[0], n[1]); k += kEU) {
            float v[2][kEU][kMaxCh];
            bool on[2][kEU];
            int ty[2][kEU];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
              for (int kk = 0; kk < kEU; ++kk) {
                on[r][kk] = k + kk < n[r];                // wavefront-uniform
                ty[r][kk] = 0;
                if (on[r][kk]) {
                  if (g.etype) ty[r][kk] = __builtin_amdgcn_readlane(myty[r], k + kk);
                  const size_t o = (size_t)__builtin_amdgcn_readlane(myid[r], k + kk) * g.ldx;
#pragma unroll
                  for (int j = 0; j < kMaxCh; ++j)
                    if (cb + 64 * j < CF) v[r][kk][j] = xs[o + cc[j]];
                }
              }
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
              for (int kk = 0; kk < kEU; ++kk)
                if (on[r][kk]) {
#pragma unroll
                  for (int j = 0; j < kMaxCh; ++j)
                    if (cb + 64 * j < CF) {
                      float m = v[r][kk][j] + dt[r][j];
                      if (g.etype) {                      // (x_src + x_dst) + W_e . ef: the order of the large-graph gather (pna_segreduce.hip Drain)
                        const int t = ty[r][kk];          // wavefront-uniform
                        m = m + (t == 0 ? et[0][j] : t == 1 ? et[1][j] : t == 2 ? et[2][j] : et[3][j]);
                      }
                      s[r][j] = s[r][j] + m; q[r][j] = q[r][j] + m * m;
                      mx[r][j] = pna_dev::vmax(mx[r][j], m); mn[r][j] = pna_dev::vmin(mn[r][j], m);
                    }
                }
          }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          if (r == 1 && !hasB) break;
#pragma unroll
          for (int j = 0; j < kMaxCh; ++j) {
            const int c = cb + 64 * j + lane;
            if (c < CF) {
              const int tl = c / Fi, f = c - tl * Fi;
              float mean, omx, omn, sd;
              pna_dev::row_stats(s[r][j], q[r][j], mx[r][j], mn[r][j], deg[r], mean, omx, omn, sd);
              float* const a = A + rw[r] * PA + tl * QA * 16 + f;
              a[0] = mean; a[Fi] = omx; a[2 * Fi] = omn; a[3 * Fi] = sd;
            }
          }
        }
      }
    }
    TF_MARK(3);
    __syncthreads();
    TF_MARK(4);
    // ---- tower contraction: items (unit u = (tower, 16-column tile), K part kp of KS); item i = u*KS + kp runs on
    // wavefront i % 8 ----
    const int U = tg * NTo;
    int KS = 1;
    if (U <= kMaxItems) {                                 // the split that minimises rounds / KS (ties: the smaller split)
      int best = 840 * ((U + kWaves - 1) / kWaves);       // 840 * rounds / KS (840 = lcm(1..8): exact integer costs)
      for (int k = 2; k <= 8 && k <= QA && U * k <= kMaxItems; ++k) {
        const int cost = 840 * ((U * k + kWaves - 1) / kWaves) / k;
        if (cost < best) { best = cost; KS = k; }
      }
    }
    const int n_items = U * KS;
    auto unit_epilogue = [&](int tl, int nt, f4 zt) __attribute__((always_inline)) {
      const int n = nt * 16 + li;
      if (n < Fo) {
        const int col = (t0 + tl) * Fo + n;
        const float bn = CB[col], cs = CB[QM * 16 + col], ct = CB[2 * QM * 16 + col];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float z = zt[i] + bn;
          if (g.row_post) z = z * SC[3 * kRows + 4 * lg + i];
          if (g.col_scale) z = z * cs + ct;
          ZC[(4 * lg + i) * PM + col] = z;
        }
      }
    };
    if (wave < n_items) {
      constexpr int R = kRing / (S + 1);                  // super-steps in flight
      StepCursor<S> cl, cc;                               // loading / consuming
      cl.KS = KS; cl.NTo = NTo; cl.QA = QA; cl.t0 = t0; cl.upt_b = upt * 16; cl.n_items = n_items;
      cl.set_item(wave);
      cc = cl;
      int total = 0;                                      // super-steps of this wavefront
      for (int it = wave; it < n_items; it += kWaves) total += (QA - it % KS + KS - 1) / KS;
      f4 b[R][S + 1];
      f4 ah = (f4){0.f, 0.f, 0.f, 0.f}, as[S];
#pragma unroll
      for (int s = 0; s < S; ++s) as[s] = ah;
      for (int g0 = -R; g0 < total; g0 += R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int gi = g0 + r;
          if (gi >= 0) {
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"((R - 1) * (S + 1)) : "memory");   // the younger super-steps' loads may fly
#pragma unroll
            for (int s = 0; s <= S; ++s) asm volatile("" : "+v"(b[r][s]));
            if (gi < total) {
              const f4 a = *reinterpret_cast<const f4*>(A + li * PA + cc.tl * QA * 16 + 4 * lg + cc.q * 16);
#pragma unroll
              for (int s = 0; s < S; ++s) quad_fma(as[s], a, b[r][s]);
              if (cc.q < QH) {
                const f4 hq = *reinterpret_cast<const f4*>(HL + ((g.h_stride ? t0 + cc.tl : 0) * kRows + li) * PH + 4 * lg + cc.q * 16);
                quad_fma(ah, hq, b[r][S]);
              }
              if (cc.last()) {                            // item complete: scale, then park the partial tile or finish the unit
                f4 zt;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  float z = ah[i];
#pragma unroll
                  for (int s = 0; s < S; ++s) z = z + SC[s * kRows + 4 * lg + i] * as[s][i];
                  zt[i] = z;
                }
                if (KS > 1) *reinterpret_cast<f4*>(RED + cc.it * 256 + lane * 4) = zt;
                else unit_epilogue(cc.tl, cc.nt, zt);
                ah = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < S; ++s) as[s] = ah;
              }
              cc.advance();
            }
          }
          const unsigned off = cl.voff(lane);
#pragma unroll
          for (int s = 0; s <= S; ++s) aload128(b[r][s], g.post_img, off + (unsigned)s * 1024u);
          cl.advance();
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" : : : "memory");  // the clamped tail loads still target b[]: drain before it dies
    }
    TF_MARK(5);
    if (KS > 1) {                                         // the K parts of a unit, summed in part order (deterministic)
      __syncthreads();
      TF_MARK(6);
      for (int u = wave; u < U; u += kWaves) {
        const int tl = u / NTo, nt = u - tl * NTo;
        f4 zt = *reinterpret_cast<const f4*>(RED + (u * KS) * 256 + lane * 4);
        for (int k = 1; k < KS; ++k) zt = zt + *reinterpret_cast<const f4*>(RED + (u * KS + k) * 256 + lane * 4);
        unit_epilogue(tl, nt, zt);
      }
    }
    TF_MARK(7);
    if (t0 + g.TG < T) __syncthreads();                   // A / RED are rewritten by the next group
  }

  // ---- mixing network (or plain store of the concatenation) ----
  if (g.mix_img == nullptr) {
    __syncthreads();
    for (int i = tid; i < nrows * TFo; i += kThreads) {
      const int r = i / TFo, n = i - r * TFo;
      g.y[(size_t)(r0 + r) * g.ldy + n] = ZC[r * PM + n];
    }
    return;
  }
  const Stream st(g.mix_img, QM, NTm, wave, lane);
  f4 res = (f4){0.f, 0.f, 0.f, 0.f};
  const bool res_early = g.residual && NTm <= kWaves && wave < NTm && wave * 16 + li < g.No;   // one tile per wavefront: its residual too
  if (res_early) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (r0 + 4 * lg + i < g.V) res[i] = g.residual[(size_t)(r0 + 4 * lg + i) * g.ld_res + wave * 16 + li];
  }
  __syncthreads();                                        // ZC complete
  TF_MARK(8);
  stream_tiles<8>(st, ZC + li * PM + 4 * lg, [&](int nt, const f4 acc) __attribute__((always_inline)) {
    const int n = nt * 16 + li;
    if (n < g.No) {
      const float bn = MB[n];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = r0 + 4 * lg + i;
        if (row < g.V) {
          float v = activate(acc[i] + bn, g.mix_act, g.mix_slope);
          if (g.residual) v = (res_early ? res[i] : g.residual[(size_t)row * g.ld_res + n]) + v;
          g.y[(size_t)row * g.ldy + n] = v;
        }
      }
    }
  });
  TF_MARK(9);
}

size_t tower_lds_bytes(int T, int TG, int Fi, int Fo, int No, bool divided) {
  const int QA = quads(4 * Fi), QH = quads(Fi), QM = quads(T * Fo);
  size_t f = (size_t)kRows * pitch_of(TG * QA) + (size_t)(divided ? T : 1) * kRows * pitch_of(QH) + (size_t)kRows * pitch_of(QM) +
             (size_t)kMaxItems * 256 + 3 * (size_t)QM * 16 + (size_t)((No + 15) / 16) * 16 + 4 * kRows;
  return f * sizeof(float);
}

}  // namespace

extern "C" int64_t pna_small_packed_floats(int32_t N, int32_t K) {
  if (N <= 0 || K <= 0) return 0;
  return (int64_t)((N + 15) / 16) * quads(K) * 256;
}

extern "C" int pna_small_pack_f32(const float* w_ref, int64_t ldw_ref, int32_t N, int32_t K, float* img, pna_stream_t stream) {
  if (!w_ref || !img || N <= 0 || K <= 0 || ldw_ref < K) return pna_set_error(PNA_E_INVALID, "pna_small_pack_f32: bad argument");
  if (launch_pack(w_ref, (long)ldw_ref, N, K, quads(K), 1, quads(K), img, (hipStream_t)stream)) return pna_set_error(PNA_E_LAUNCH, "pna_small_pack_f32: launch failed");
  return PNA_OK;
}

extern "C" int pna_small_linear_f32(const pna_small_linear_args* p, pna_stream_t stream) {
  if (!p) return pna_set_error(PNA_E_INVALID, "pna_small_linear_f32: null args");
  if (int rc_ss = pna_check_struct_size("pna_small_linear_f32", p->struct_size, sizeof(*p))) return rc_ss;
  if (!p || !p->x || !p->img || !p->y || p->M < 0 || p->K <= 0 || p->N <= 0 || p->act < 0 || p->act > 2)
    return pna_set_error(PNA_E_INVALID, "pna_small_linear_f32: bad argument");
  if (p->M == 0) return PNA_OK;
  const size_t lds = ((size_t)kRows * pitch_of(quads(p->K)) + (size_t)((p->N + 15) / 16) * 16) * sizeof(float);
  if (lds > (size_t)kMaxLds) return pna_set_error(PNA_E_INVALID, "pna_small_linear_f32: K (and N) too large for one LDS tile (16*K + N <= ~40000)");
  LArgs g;
  g.x = p->x; g.ldx = (long)p->ldx; g.M = p->M; g.K = p->K; g.N = p->N; g.img = p->img; g.bias = p->bias;
  g.act = p->act; g.slope = p->act_slope; g.residual = p->residual; g.ld_res = (long)p->ld_res; g.y = p->y; g.ldy = (long)p->ldy;
#ifdef PNA_AMD_EXPERIMENTS
  g.dbg = nullptr;
  if (const char* e = getenv("PNA_TF_DBG_LINEAR")) g.dbg = (unsigned long long*)strtoull(e, nullptr, 0);   // 16 counters per wavefront
#endif
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)k_small_linear, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return pna_set_error(PNA_E_LAUNCH, "pna_small_linear_f32: LDS attribute refused");
  hipLaunchKernelGGL(k_small_linear, dim3((unsigned)((p->M + kRows - 1) / kRows)), dim3(kThreads), lds, (hipStream_t)stream, g);
  if (hipGetLastError() != hipSuccess) return pna_set_error(PNA_E_LAUNCH, "pna_small_linear_f32: launch failed");
  return PNA_OK;
}

extern "C" int64_t pna_tower_post_packed_floats(int32_t Fi, int32_t Fo, int32_t n_scaler) {
  if (Fi <= 0 || Fo <= 0 || n_scaler <= 0) return 0;
  return (int64_t)((Fo + 15) / 16) * quads(4 * Fi) * (n_scaler + 1) * 256;
}

extern "C" int pna_tower_post_pack_f32(const float* w_ref, int64_t ldw_ref, int32_t Fi, int32_t Fo, int32_t n_scaler, float* img,
                                       pna_stream_t stream) {
  if (!w_ref || !img || Fi <= 0 || Fo <= 0 || n_scaler <= 0 || n_scaler > 3 || ldw_ref < (int64_t)Fi * (1 + 4 * n_scaler))
    return pna_set_error(PNA_E_INVALID, "pna_tower_post_pack_f32: bad argument");
  // unit (16-column tile) image: for every aggregate quad q the fragments of scaler 0 .. S-1, then the own-features block's
  // fragment of the same quad (all zero past its ceil(Fi / 16) quads)
  const int QA = quads(4 * Fi), S1 = n_scaler + 1, per_nt = QA * S1;
  int rc = 0;
  for (int s = 0; s < n_scaler && !rc; ++s)
    rc = launch_pack(w_ref + Fi + (size_t)s * 4 * Fi, (long)ldw_ref, Fo, 4 * Fi, QA, S1, per_nt, img + (size_t)s * 256, (hipStream_t)stream);
  if (!rc) rc = launch_pack(w_ref, (long)ldw_ref, Fo, Fi, QA, S1, per_nt, img + (size_t)n_scaler * 256, (hipStream_t)stream);
  if (rc) return pna_set_error(PNA_E_LAUNCH, "pna_tower_post_pack_f32: launch failed");
  return PNA_OK;
}

extern "C" int pna_tower_layer_f32(const pna_tower_layer_args* p, pna_stream_t stream) {
  if (!p) return pna_set_error(PNA_E_INVALID, "pna_tower_layer_f32: null args");
  if (int rc_ss = pna_check_struct_size("pna_tower_layer_f32", p->struct_size, sizeof(*p))) return rc_ss;
  if (!p || !p->rowptr || !p->h || !p->x_cat || !p->proj_img || !p->post_img || !p->y)      // (col may be NULL: a graph without edges)
    return pna_set_error(PNA_E_INVALID, "pna_tower_layer_f32: null argument");
  const int T = p->n_tower, Fi = p->Fi, Fo = p->Fo, S = p->n_scaler;
  if (p->V < 0 || T <= 0 || Fi <= 0 || Fo <= 0 || S < 1 || S > 3 || p->mix_act < 0 || p->mix_act > 2)
    return pna_set_error(PNA_E_INVALID, "pna_tower_layer_f32: bad shape (n_tower, Fi, Fo >= 1; 1 <= n_scaler <= 3)");
  const int Fin = p->divide_input ? T * Fi : Fi;
  if (p->ldh < Fin || p->ldx < 2 * (int64_t)T * Fi || p->ldy < (p->mix_img ? p->No : T * Fo))
    return pna_set_error(PNA_E_INVALID, "pna_tower_layer_f32: leading dimension too small");
  if (p->mix_img && p->No <= 0) return pna_set_error(PNA_E_INVALID, "pna_tower_layer_f32: mixing network without an output width");
  if (p->col_scale && !p->col_shift) return pna_set_error(PNA_E_INVALID, "pna_tower_layer_f32: col_scale without col_shift");
  if (p->edge_type && (!p->edge_table || p->n_edge_types < 1 || p->n_edge_types > 4 || p->ld_edge_table < (int64_t)T * Fi))
    return pna_set_error(PNA_E_INVALID, "pna_tower_layer_f32: edge_type needs edge_table (n_edge_types in 1..4 rows of >= n_tower * Fi floats)");
  if (p->V == 0) return PNA_OK;
  const int No = p->mix_img ? p->No : T * Fo;
  int TG = T;
  while (TG > 1 && tower_lds_bytes(T, TG, Fi, Fo, No, p->divide_input != 0) > (size_t)kMaxLds) --TG;
  const size_t lds2 = tower_lds_bytes(T, TG, Fi, Fo, No, p->divide_input != 0);
  if (lds2 > (size_t)kMaxLds) return pna_set_error(PNA_E_INVALID, "pna_tower_layer_f32: one tower's tile does not fit the LDS (use the large-graph path)");
  hipStream_t st = (hipStream_t)stream;
  // launch 1: x_cat = h [W_a ; W_b]^T + [0 ; b] for every tower
  pna_small_linear_args l;
  memset(&l, 0, sizeof(l));
  l.struct_size = (uint32_t)sizeof(l);
  l.x = p->h; l.ldx = p->ldh; l.M = p->V; l.K = Fin; l.N = 2 * T * Fi; l.img = p->proj_img; l.bias = p->proj_bias;
  l.y = p->x_cat; l.ldy = p->ldx;
  const int rc = pna_small_linear_f32(&l, stream);
  if (rc != PNA_OK) return rc;
  // launch 2
  TArgs g;
  g.rowptr = p->rowptr; g.col = p->col; g.V = p->V; g.T = T; g.Fi = Fi; g.Fo = Fo; g.TG = TG;
  g.xcat = p->x_cat; g.ldx = (long)p->ldx; g.h = p->h; g.ldh = (long)p->ldh; g.h_stride = p->divide_input ? Fi : 0;
  for (int s = 0; s < 3; ++s) g.scale[s] = s < S ? p->row_scale[s] : nullptr;
  g.post_img = p->post_img; g.post_bias = p->post_bias; g.row_post = p->row_post; g.col_scale = p->col_scale; g.col_shift = p->col_shift;
  g.mix_img = p->mix_img; g.mix_bias = p->mix_bias; g.No = p->mix_img ? p->No : T * Fo; g.mix_act = p->mix_act; g.mix_slope = p->mix_slope;
  g.residual = p->residual; g.ld_res = (long)p->ld_res; g.y = p->y; g.ldy = (long)p->ldy;
  g.no_self = p->no_self_panel != 0;
  g.etype = p->edge_type; g.etab = p->edge_table; g.ldet = (long)p->ld_edge_table; g.n_types = p->n_edge_types;
#ifdef PNA_AMD_EXPERIMENTS
  g.dbg = nullptr;
  if (const char* e = getenv("PNA_TF_DBG_ROWS")) g.dbg = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
  const void* fn = S == 1 ? (const void*)k_tower_rows<1> : S == 2 ? (const void*)k_tower_rows<2> : (const void*)k_tower_rows<3>;
  if (lds2 > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)
    return pna_set_error(PNA_E_LAUNCH, "pna_tower_layer_f32: LDS attribute refused");
  const dim3 grid((unsigned)((p->V + kRows - 1) / kRows)), block(kThreads);
  if (S == 1) hipLaunchKernelGGL(k_tower_rows<1>, grid, block, lds2, st, g);
  else if (S == 2) hipLaunchKernelGGL(k_tower_rows<2>, grid, block, lds2, st, g);
  else hipLaunchKernelGGL(k_tower_rows<3>, grid, block, lds2, st, g);
  if (hipGetLastError() != hipSuccess) return pna_set_error(PNA_E_LAUNCH, "pna_tower_layer_f32: launch failed");
  return PNA_OK;
}
