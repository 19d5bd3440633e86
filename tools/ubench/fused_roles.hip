// ARCHIVED in round 5 (ABI 20): no longer part of libpna_amd.so.  Build as a standalone library with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Iinclude -Ipna_amd/csrc -Itools/ubench tools/ubench/fused_roles.hip pna_amd/csrc/pna_common.hip -o tools/ubench/libfused_roles.so
// Result: parity-green, 2.1-2.7x slower than pna_fused_degree_f32 (profiles/r04_roles_*).
// pna_fused_roles.hip -- PNASimpleLayer forward (models/dgl/pna_layer.py:186-216) as ONE kernel of SPECIALISED wavefronts on
// degree-ordered rows (round 4; the successor of pna_fused_degree.hip for its main shapes).  Implements
// pna_fused_roles_f32; the weight images are pna_fused_degree_pack_f32's.
//
//   y[perm[v]] = epilogue( bias + W_D . [mean | max | min | std](x[src] over the in-edges of perm[v]) ),  W_D = sum_s scale_s(D) W_s
//
// Why roles.  In pna_fused_degree.hip every wavefront alternates a gather phase (latency-bound: its loads stop while it
// multiplies) and a multiply phase (issue-bound); VMEM returns in order per wavefront, so nothing of the next tile can be
// requested under the multiply phase without stalling the weight copies behind it.  Here a workgroup is 8 wavefronts on one CU,
// two per SIMD:
//   * G wavefronts (0..3) never leave the gather: a register ring of RING edge packets (one packet = the 16-byte strips of 16
//     source rows + the id the slot needs next) is kept full ACROSS tile boundaries -- the wavefront's tiles are a contiguous
//     range, so its id records are one linear stream -- and the running sum / sum of squares / max / min are folded by the same
//     single VALU instructions as pna_segreduce.hip (same bits).  When a tile's last packet is folded the statistics are
//     finished ([mean | max | min | std], pna_rowstats.h arithmetic) and handed to the SIMD's M wavefront through LDS.
//   * M wavefronts (4..7) never touch the source table: they stream the degree group's combined weight image through LDS
//     (global_load_lds, 5 buffers, counted waits), split the 8 statistics of a chunk into three bf16 terms (pna_x3_split.h),
//     multiply (v_mfma_f32_16x16x32_bf16, six partial products, fp32 accumulate) and run the epilogue.
//   G -> M: `full` / `empty` words in LDS per SIMD pair (monotonic tile counters).  M <-> M: the four M wavefronts share the
//   weight buffers; s_barrier would join the G wavefronts too, so they keep step through an arrive word each and a poll.
//   Every spin is bounded (a wavefront that gives up sets err[0] and leaves; its partners then give up in turn).
// The multiply computes the TRANSPOSED product (weights as the MFMA A operand, statistics as B): a lane then holds 4 consecutive
// output COLUMNS of ONE row -- 16 contiguous bytes of y -- and the epilogue needs no transpose (pna_fused_degree.hip: two DPP
// butterflies per column tile).
// Source rows are addressed with 64-bit lane addresses: tables >= 4 GiB and >= 2^24 rows (BASELINE configs[4] at 8 ranks: the
// [local | halo] table of a shard) take this path too.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "fused_roles_abi.h"   // (includes pna_amd.h)
#include "pna_internal.h"
#include "pna_rowstats.h"
#include "pna_x3_split.h"

namespace {

using namespace pna_x3;
using pna_dev::div_rn;

typedef int i4 __attribute__((ext_vector_type(4)));

struct FRArgs {
  const i4* wdesc;             // per 64-row workgroup tile: {first id record, in-degree, weight image, 0}
  const int* ids;              // 4 arrays (one per 16-row block of a tile) of 16-id records, ids_stride bytes apart
  const int* wg_range;         // [grid][2]: workgroup b owns the tiles [wg_range[2 b], wg_range[2 b + 1])
  const char* x;
  const int* perm; const unsigned char* w_img; long img_stride;
  const float* bias; const float* col_scale; const float* col_shift; const float* residual; float* y;
  float* agg_out; long ld_agg; // optional (verification): the statistics as the contraction sees them
  unsigned long long* dbg;     // experiments build only: per-wavefront timers
  int* err;                    // [1]: set when a spin gave up
  long ids_stride;             // bytes between the four id arrays
  unsigned ldb;                // row pitch of x in bytes
  unsigned ldyb, ldrb;         // row pitch of y / residual in bytes
  int F, N, relu;
  float slope;
  int prio_g, prio_m;          // experiments build only: s_setprio of the two roles (default 0 / 2)
  int abl;                     // experiments build only: parts skipped for timing (bit 0 MFMAs, 1 fragment maths, 2 the fold, 3 the y
};                             // stores, 4 the weight reads, 5 the whole M side but the hand-over): results are then meaningless

constexpr int kNW = 80, kG = 4, kM = 4, kThreads = 64 * (kG + kM);
constexpr int kNT = kNW / 16;                             // column tiles
__host__ __device__ constexpr int ahead_of(int nc) { return nc < 5 ? nc : 5; }   // a chunk image is requested at the start of the step this many steps before it is read
constexpr int kChunkV = 3 * 4 * kNW;                      // 16-byte pieces of one chunk image: [term][lane group][80 cols][8 k] bf16
constexpr int kNI = (kChunkV + 64 * kM - 1) / (64 * kM);  // global_load_lds instructions per M wavefront per chunk
constexpr int kSpin = 1 << 20;                            // polls before a wavefront gives up

__host__ __device__ constexpr int shape_full(int F) { return (F % 32 == 0 || F % 32 > 16) ? (F + 31) / 32 : F / 32; }
__host__ __device__ constexpr bool shape_half(int F) { return F % 32 != 0 && F % 32 <= 16; }
__host__ __device__ constexpr int shape_chunks(int F) { return 4 * shape_full(F) + (shape_half(F) ? 2 : 0); }

// ---- inline-asm memory operations (hipcc neither counts nor waits for them: every wait below is ours) -----------------------
template <int OFF>
__device__ __forceinline__ void ldx16(f4& dst, const char* p) {     // 64-bit lane address + immediate
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF) : "memory");
}
__device__ __forceinline__ void ld16s(f4& dst, const void* base, unsigned voff) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void ld4(int& dst, const void* base, unsigned voff) {
  asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
// (s_nop 4: "VALU writes SGPR -> VMEM reads that SGPR" needs five wait states; hipcc does not look inside the statement and may
// reload a spilled base pointer with v_readlane_b32 directly in front of it -- pna_fused_degree.hip, tools/isa_audit.py)
__device__ __forceinline__ void sld16(i4& dst, const void* base, unsigned soff) {
  asm volatile("s_load_dwordx4 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(dst) : "s"(base), "s"(soff) : "memory");
}
// (the lane read happens INSIDE the statement: hoisted out of a conditional block by hipcc, a v_readfirstlane_b32 of the statement's
// VGPR result reads an arbitrary register on the path that skipped the block -- harmless, but tools/isa_audit.py would flag it)
__device__ __forceinline__ int lds_peek(unsigned addr) {
  int v, r;
  asm volatile("ds_read_b32 %1, %2\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 0\n\tv_readfirstlane_b32 %0, %1" : "=s"(r), "=&v"(v) : "v"(addr) : "memory");
  return r;
}
__device__ __forceinline__ int lds_peek_min4(unsigned addr) {      // the smallest of four consecutive words
  int a, b, c, d, r;
  asm volatile("ds_read_b32 %1, %5\n\tds_read_b32 %2, %5 offset:4\n\tds_read_b32 %3, %5 offset:8\n\tds_read_b32 %4, %5 offset:12\n\t"
               "s_waitcnt lgkmcnt(0)\n\tv_min_i32 %1, %1, %2\n\tv_min_i32 %3, %3, %4\n\tv_min_i32 %1, %1, %3\n\ts_nop 0\n\tv_readfirstlane_b32 %0, %1"
               : "=s"(r), "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(addr) : "memory");
  return r;
}
__device__ __forceinline__ void lds_poke(unsigned addr, int val) {   // (every lane writes the same word: one lane is enough)
  asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(val) : "memory");
}
template <int N, int NL>
__device__ __forceinline__ void wait_slot(f4 (&s)[NL], int& id) {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit count");
  static_assert(NL >= 2 && NL <= 5, "2..5 row loads per edge");
  if constexpr (NL == 2)
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(s[0]), "+v"(s[1]), "+v"(id) : "n"(N) : "memory");
  else if constexpr (NL == 3)
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(id) : "n"(N) : "memory");
  else if constexpr (NL == 4)
    asm volatile("s_waitcnt vmcnt(%5)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(id) : "n"(N) : "memory");
  else
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(id) : "n"(N) : "memory");
}
// one message into the running statistics of one feature: the production fold as single VALU instructions (pna_segreduce.hip;
// NOT packed: a packed-fp32 op whose low lane reads src1's high half drops results beside MFMA wavefronts)
__device__ __forceinline__ void fold1(float& S, float& Q, float& MX, float& MN, float m) {
  float s1, p1, q1, x1, n1;
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(S), "v"(m));
  asm volatile("v_mul_f32 %0, %1, %1" : "=v"(p1) : "v"(m));
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(q1) : "v"(Q), "v"(p1));
  asm volatile("v_max_f32 %0, %1, %2" : "=v"(x1) : "v"(MX), "v"(m));
  asm volatile("v_min_f32 %0, %1, %2" : "=v"(n1) : "v"(MN), "v"(m));
  S = s1; Q = q1; MX = x1; MN = n1;
}

#ifdef PNA_AMD_EXPERIMENTS
#define FR_ABL(bit) ((g.abl >> (bit)) & 1)
__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }
#else
#define FR_ABL(bit) 0
__device__ __forceinline__ unsigned long long now() { return 0; }
#endif

// LDS map (bytes): [kNBuf weight buffers][4 hand-over buffers][3 x 80 column constants][flags]
template <int NC, int NP>
struct Lds {
  static constexpr int NBUF = ahead_of(NC) + 2;            // weight buffers (F = 75: seven)
  static constexpr int CHB = kChunkV * 16;                 // one chunk image (15 360 bytes)
  static constexpr int SB = 2 * NP * 1024;                 // one hand-over PHASE of one tile: two quantities x NP pieces x [lane][4 floats]
  static constexpr int stats = NBUF * CHB;
  static constexpr int colc = stats + kG * SB;
  static constexpr int flags = colc + 3 * kNW * 4;         // full[4] | empty[4] | (unused)[4] | arrive[4]
  static constexpr int total = flags + 64;
};

template <int NFBF, bool HALF, bool DUMP, int RING>
__global__ __launch_bounds__(kThreads, 2) void k_fused_roles(const FRArgs g) {
  constexpr int NB = NFBF + (HALF ? 1 : 0);               // feature blocks
  constexpr int NC = 4 * NFBF + (HALF ? 2 : 0);           // chunks of 32 k values = steps per tile
  constexpr int NT = kNT;                                 // column tiles
  constexpr int NL = 2 * NFBF + (HALF ? 1 : 0);           // 16-byte row loads per edge
  constexpr int NP = NL;                                  // 16-byte pieces of one quantity of the statistics
  constexpr int LB = NL + 1;                              // loads of one edge packet (the strips + the slot's next id)
  static_assert(NC >= 4, "every shape has at least 4 steps");
  static_assert((RING - 1) * LB < 64, "vmcnt is a 6-bit count");
  using L = Lds<NC, NP>;
  constexpr int kAhead = ahead_of(NC), kNBuf = L::NBUF;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;

  // ---- once per workgroup: column constants and flags, then the roles part for good -------------------------------------
  {
    float* const colc = reinterpret_cast<float*>(lds + L::colc);
    for (int i = tid; i < kNW; i += kThreads) {
      colc[i] = (g.bias && i < g.N) ? g.bias[i] : 0.f;
      colc[kNW + i] = (g.col_scale && i < g.N) ? g.col_scale[i] : 1.f;
      colc[2 * kNW + i] = (g.col_shift && i < g.N) ? g.col_shift[i] : 0.f;
    }
    if (tid < 16) reinterpret_cast<int*>(lds + L::flags)[tid] = 0;
  }
  __syncthreads();
  const int T0 = g.wg_range[2 * blockIdx.x], T1 = g.wg_range[2 * blockIdx.x + 1];
  if (T0 >= T1) return;
  const unsigned long long t00 = now();
  unsigned long long tw0 = 0, tw1 = 0;                     // cycles spent waiting for the partner / for the other M wavefronts
  int spins = 0;

  if (wave < kG) {
    // =========================================== G: gather, fold, finish, hand over ========================================
    const int w = wave;
    const unsigned a_full = lds0 + L::flags + w * 4, a_empty = lds0 + L::flags + 16 + w * 4;
    const unsigned a_stats = lds0 + L::stats + w * L::SB + lane * 16;
    float S_[NB][8], Q_[NB][8], MX[NB][8], MN[NB][8];     // (a half block uses [0..3])
    auto reset = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int fb = 0; fb < NB; ++fb)
#pragma unroll
        for (int j = 0; j < 8; ++j) { S_[fb][j] = 0.f; Q_[fb][j] = 0.f; MX[fb][j] = -INFINITY; MN[fb][j] = INFINITY; }
    };
    reset();
    // lane addresses of the row strips.  A: the lane's strip of block 0 (block fb of the same lane lies 128 fb bytes further: an
    // immediate); L: the LAST block's strip -- the half block, or the last full block of a shape whose F ends inside it -- which
    // is the only one that can lie past F: such a lane re-reads the row's last strip (its values are masked in stat()).
    constexpr int NFA = HALF ? NFBF : NFBF - 1;            // full blocks addressed through A
    // (a window that would reach past F slides back to end at F: no read leaves the row; the pack kernel zeroes the weights of the
    // slots that then repeat a lower lane group's feature -- pna_fused_degree.hip, feat0 / is_dup)
    const int fl = min(HALF ? NFBF * 32 + lg * 4 : (NFBF - 1) * 32 + lg * 8, g.F - (HALF ? 4 : 8));
    const char* const xA = g.x + (size_t)lg * 32;
    const char* const xL = g.x + (size_t)fl * 4;
    const unsigned ldb = g.ldb;
    const char* const idsw = reinterpret_cast<const char*>(g.ids) + (size_t)w * g.ids_stride;
    const unsigned lib = (unsigned)li * 4u;

    i4 td;
    sld16(td, g.wdesc, (unsigned)T0 * 16u);
    int T = T0, rem = max(td.y, 1), k = 0, dnext = 0;
    unsigned nid = (unsigned)td.x * 64u + lib;             // byte offset (in this block's id array) of the next id to request
    int idr[RING];
    f4 sl[RING][NL];
    // rows of the edge whose id sits in idr[j] -> ring slot j, then the id slot j gathers next -> idr[j]
    auto issue = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      const size_t ro = (size_t)(unsigned)idr[j] * ldb;
      if constexpr (NFA > 0) {
        const char* const pa = xA + ro;
        ldx16<0>(sl[j][0], pa); ldx16<16>(sl[j][1], pa);
        if constexpr (NFA > 1) { ldx16<128>(sl[j][NFA > 1 ? 2 : 0], pa); ldx16<144>(sl[j][NFA > 1 ? 3 : 0], pa); }
        static_assert(NFA <= 2, "full blocks addressed through A");
      }
      {
        const char* const pl = xL + ro;
        ldx16<0>(sl[j][2 * NFA], pl);
        if constexpr (!HALF) ldx16<16>(sl[j][HALF ? 0 : 2 * NFA + 1], pl);
      }
      ld4(idr[j], idsw, nid);
      nid += 64u;
    };
    auto fold = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      if (FR_ABL(2)) { asm volatile("" : "+v"(sl[j][0])); return; }
#pragma unroll
      for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int fb = l >> 1, c = (l & 1) * 4 + q;
          fold1(S_[fb][c], Q_[fb][c], MX[fb][c], MN[fb][c], sl[j][l][q]);
        }
    };
    // The tile's RAW statistics to the M wavefront in TWO phases through one buffer ([quantity][16-byte piece][lane][4 floats]; a
    // full block is two pieces, the half block one): sum | sum of squares, then max | min -- seven weight buffers (the copies'
    // latency under the gather's load needs that many in flight) leave 10 KB per SIMD pair.  full / empty count phases.  At the end of the stream (or when a spin gave up) T = T1 and no
    // further boundary comes.
    auto boundary = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        // the buffer is free once the M wavefront has taken the phase before
        const unsigned long long tp = now();
        int n = 0;
        while (lds_peek(a_empty) < 2 * k + ph) {
          if (++n > kSpin) { g.err[0] = 1; T = T1; rem = 0x7fffffff; return; }
          __builtin_amdgcn_s_sleep(1);
        }
        tw0 += now() - tp; spins += n;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int pc = 0; pc < NP; ++pc) {
            const int fb = pc >> 1, h0 = (pc & 1) * 4;
            float (&src)[NB][8] = ph == 0 ? (q == 0 ? S_ : Q_) : (q == 0 ? MX : MN);
            const f4 v = (f4){src[fb][h0], src[fb][h0 + 1], src[fb][h0 + 2], src[fb][h0 + 3]};
            asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(a_stats), "v"(v), "n"((q * NP + pc) * 1024) : "memory");
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) lds_poke(a_full, 2 * k + ph + 1);
      }
      ++k;
      reset();
      if (++T == T1) { rem = 0x7fffffff; return; }    // (the stream's end: the trip's remaining slots fold packets nobody hands over)
      // (the next tile's in-degree; the result is waited for inside the statement: pna_fused_degree.hip on in-flight SGPRs)
      asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(dnext) : "s"(g.wdesc), "s"((unsigned)T * 16u + 4u) : "memory");
      rem = max(dnext, 1);
    };

    // the ids of the stream's first RING packets, then their rows
#pragma unroll
    for (int j = 0; j < RING; ++j) ld4(idr[j], idsw, nid + (unsigned)j * 64u);
    nid += (unsigned)RING * 64u;
    static_assert(RING >= 2 && RING <= 7, "ring slots");
    if constexpr (RING == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(idr[0]), "+v"(idr[1]) : : "memory");
    if constexpr (RING == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(idr[0]), "+v"(idr[1]), "+v"(idr[RING > 2 ? 2 : 0]) : : "memory");
    if constexpr (RING == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(idr[0]), "+v"(idr[1]), "+v"(idr[RING > 2 ? 2 : 0]), "+v"(idr[RING > 3 ? 3 : 0]) : : "memory");
    if constexpr (RING == 5) asm volatile("s_waitcnt vmcnt(0)" : "+v"(idr[0]), "+v"(idr[1]), "+v"(idr[RING > 2 ? 2 : 0]), "+v"(idr[RING > 3 ? 3 : 0]), "+v"(idr[RING > 4 ? 4 : 0]) : : "memory");
    if constexpr (RING == 6) asm volatile("s_waitcnt vmcnt(0)" : "+v"(idr[0]), "+v"(idr[1]), "+v"(idr[RING > 2 ? 2 : 0]), "+v"(idr[RING > 3 ? 3 : 0]), "+v"(idr[RING > 4 ? 4 : 0]), "+v"(idr[RING > 5 ? 5 : 0]) : : "memory");
    if constexpr (RING == 7) asm volatile("s_waitcnt vmcnt(0)" : "+v"(idr[0]), "+v"(idr[1]), "+v"(idr[RING > 2 ? 2 : 0]), "+v"(idr[RING > 3 ? 3 : 0]), "+v"(idr[RING > 4 ? 4 : 0]), "+v"(idr[RING > 5 ? 5 : 0]), "+v"(idr[RING > 6 ? 6 : 0]) : : "memory");
    using J0 = std::integral_constant<int, 0>; using J1 = std::integral_constant<int, 1>; using J2 = std::integral_constant<int, (RING > 2 ? 2 : 0)>;
    using J3 = std::integral_constant<int, (RING > 3 ? 3 : 0)>; using J4 = std::integral_constant<int, (RING > 4 ? 4 : 0)>;
    using J5 = std::integral_constant<int, (RING > 5 ? 5 : 0)>; using J6 = std::integral_constant<int, (RING > 6 ? 6 : 0)>;
    issue(J0{}); issue(J1{});
    if constexpr (RING > 2) issue(J2{});
    if constexpr (RING > 3) issue(J3{});
    if constexpr (RING > 4) issue(J4{});
    if constexpr (RING > 5) issue(J5{});
    if constexpr (RING > 6) issue(J6{});
    // one trip = RING packets: wait for a slot, fold it, refill it with the packet RING places ahead (always a real record: the
    // next wavefront's stream, or the array's padding, follows this one's last tile); a tile may end behind any slot.
#ifdef PNA_AMD_EXPERIMENTS
    if (g.prio_g == 0) __builtin_amdgcn_s_setprio(0); else if (g.prio_g == 1) __builtin_amdgcn_s_setprio(1); else if (g.prio_g == 3) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2);
#else
    __builtin_amdgcn_s_setprio(0);
#endif
    // (no exit in the middle of a trip: hipcc fills the merge points of such a loop with lane reads of undefined registers)
#define FR_SLOT(J) { wait_slot<(RING - 1) * LB, NL>(sl[J::value], idr[J::value]); fold(J{}); issue(J{}); if (--rem == 0) boundary(); }
    while (T != T1) {
      FR_SLOT(J0) FR_SLOT(J1)
      if constexpr (RING > 2) FR_SLOT(J2)
      if constexpr (RING > 3) FR_SLOT(J3)
      if constexpr (RING > 4) FR_SLOT(J4)
      if constexpr (RING > 5) FR_SLOT(J5)
      if constexpr (RING > 6) FR_SLOT(J6)
    }
#undef FR_SLOT
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the packets requested past the stream's end
  } else {
    // =========================================== M: finish, split, multiply, epilogue ====================================
    const int m = wave - kG;
    const unsigned a_full = lds0 + L::flags + m * 4, a_empty = lds0 + L::flags + 16 + m * 4;
    const unsigned a_arr = lds0 + L::flags + 48, a_my = a_arr + m * 4;
    const unsigned a_stats = lds0 + L::stats + m * L::SB + lane * 16;
    const unsigned a_colc = lds0 + L::colc + lg * 16;
    f4 acc[NT], res[NT];
    f4 raw[4][NP];                                         // the tile's raw statistics: sum | sum of squares | max | min
    int deg = 0;                                           // the tile's in-degree
    int pr = -1, pr_next = -1;                             // node of tile row li (-1: padding)
    const void* const resb = g.residual ? (const void*)g.residual : (const void*)g.y;
    const bool has_res = g.residual != nullptr;
    int na = 0;                                            // arrivals of this wavefront so far
    bool dead = false;
    unsigned long long tph[7] = {0, 0, 0, 0, 0, 0, 0}, tq = now();   // experiments build: cycles per phase of the tile loop
    unsigned long long tst[5] = {0, 0, 0, 0, 0}, tsq = 0;           // ... and inside a step: wait for the others | image request | first half | own pieces | rest
    auto slap = [&](int i) __attribute__((always_inline)) { const unsigned long long t = now(); tst[i] += t - tsq; tsq = t; };
    auto lap = [&](int i) __attribute__((always_inline)) { const unsigned long long t = now(); tph[i] += t - tq; tq = t; };
    // weight chunks: global -> LDS, asynchronously (every M wavefront issues exactly kNI copies per chunk)
    auto stage = [&](int c, int buf, long ib) __attribute__((always_inline)) {
      const unsigned char* src = g.w_img + ib + (size_t)c * kChunkV * 16;
      unsigned char* dst = lds + (size_t)buf * kChunkV * 16;
#pragma unroll
      for (int i = 0; i < kNI; ++i) {
        int w0 = (i * kM + m) * 64;
        if (w0 >= kChunkV) w0 = w0 % kChunkV;             // a slot past the image re-copies an earlier piece (same bytes, same address)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(w0 + lane) * 16),
                                         (__attribute__((address_space(3))) void*)(dst + (size_t)w0 * 16), 16, 0, 0);
      }
    };
    auto arrive = [&]() __attribute__((always_inline)) { ++na; if (lane == 0) lds_poke(a_my, na); };
    auto waitall = [&]() __attribute__((always_inline)) {
      const unsigned long long tp = now();
      int n = 0;
      while (lds_peek_min4(a_arr) < na) {
        if (++n > kSpin) { g.err[0] = 2; dead = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      tw1 += now() - tp; spins += n;
    };
    // chunk c: the lane's eight A values, finished from the raw statistics (pna_rowstats.h arithmetic: the bits of
    // pna_segreduce_fwd_f32) and split into three bf16 terms.  Full block fb, chunk 4 fb + a: aggregator a (0 mean, 1 max, 2 min,
    // 3 std) of the lane's 8 features; half block, chunk 4 NFBF + h: aggregators 2h | 2h + 1 of its 4 features.
    auto rawv = [&](int q, int fb, int j) __attribute__((always_inline)) -> float { return raw[q][2 * fb + (j >> 2)][j & 3]; };
    // (single VALU instructions through inline asm, like the fold: written as plain C++ hipcc packs two features' chains into
    // v_pk_fma_f32 / v_pk_mul_f32 with op_sel swizzles -- the form that drops results beside MFMA wavefronts)
    auto mul1 = [](float a, float b) __attribute__((always_inline)) -> float { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
    auto fma1 = [](float a, float b, float c) __attribute__((always_inline)) -> float { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; };
    auto fnma1 = [](float a, float b, float c) __attribute__((always_inline)) -> float { float r; asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; };
    auto sub1 = [](float a, float b) __attribute__((always_inline)) -> float { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
    auto add1 = [](float a, float b) __attribute__((always_inline)) -> float { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
    auto div_rn1 = [&](float a, float D_, float invD_) __attribute__((always_inline)) -> float {   // pna_rowstats.h div_rn, op by op
      const float q0 = mul1(a, invD_), q = fma1(fnma1(D_, q0, a), invD_, q0);
      return (q == q && __builtin_fabsf(q) != INFINITY) ? q : q0;
    };
    auto stat = [&](int fb, int j, int a, int f) __attribute__((always_inline)) -> float {
      const float Df = (float)deg, invD = 1.0f / Df;
      const float sv = rawv(0, fb, j), q = rawv(1, fb, j);
      float r;
      if (a == 0) {
        r = div_rn1(sv, Df, invD);
      } else if (a == 3) {
        const float mean = div_rn1(sv, Df, invD), msq = div_rn1(q, Df, invD);
        float var = sub1(msq, mul1(mean, mean));
        var = var < 0.f ? 0.f : var;
        r = sqrtf(add1(var, 1e-5f));
      } else {
        const float e = a == 1 ? rawv(2, fb, j) : rawv(3, fb, j);
        r = q != q ? q : e;                                 // v_max / v_min drop NaN; q is NaN iff a message is (pna_rowstats.h)
      }
      if (deg <= 0) r = 0.f;                                // rows without in-edges: DGL leaves them at zero
      return r;                                             // (every slot holds a feature < F: no padding)
    };
    int Tm = 0;                                            // (the tile, for the verification output)
    constexpr int WL = HALF ? 4 : 8;                       // the last block's window (pna_fused_degree.hip: feat0 / is_dup)
    const int fl_nom = (NB - 1) * 32 + lg * WL, fl_abs = min(fl_nom, g.F - WL);
    auto feat0 = [&](int fb) __attribute__((always_inline)) -> int { return fb == NB - 1 ? fl_abs : fb * 32 + lg * 8; };
    // FAST tiles (every raw statistic finite, in-degree > 0: all but pathological inputs; decided per tile from the sums of squares):
    // the same arithmetic without the special-value selects -- div_rn's NaN / Inf fall-back, the NaN test of max / min, sqrtf's
    // denormal scaling and class test (var + 1e-5 is a normal number), the Inf test of the split -- and with the mean kept from
    // the block's first chunk: ~900 VALU instructions per tile instead of ~1500.  The bits are the slow path's.
      auto sqrt_rn = [&](float x) __attribute__((always_inline)) -> float {     // correctly rounded for normal x (hipcc's own sequence behind
      const float r = __builtin_amdgcn_sqrtf(x);                               // v_sqrt_f32, less the denormal scaling and the class test)
      const float rm = bfloat(fbits(r) - 1u), rp2 = bfloat(fbits(r) + 1u);
      const float e1 = fnma1(rm, r, x), e2 = fnma1(rp2, r, x);
      float o = e1 <= 0.f ? rm : r;
      o = e2 > 0.f ? rp2 : o;
      return o;
    };
    auto div_fast = [&](float a, float D_, float invD_) __attribute__((always_inline)) -> float {   // div_rn without its NaN / Inf fall-back
      const float q0 = mul1(a, invD_);
      return fma1(fnma1(D_, q0, a), invD_, q0);
    };
    auto stat_fast = [&](int fb, int j, int a) __attribute__((always_inline)) -> float {
      const float Df = (float)deg, invD = 1.0f / Df;
      if (a == 1) return rawv(2, fb, j);
      if (a == 2) return rawv(3, fb, j);
      if (a == 0) return div_fast(rawv(0, fb, j), Df, invD);
      const float mean = div_fast(rawv(0, fb, j), Df, invD), msq = div_fast(rawv(1, fb, j), Df, invD);
      float var = sub1(msq, mul1(mean, mean));
      var = pna_dev::vmax(var, 0.f);
      return sqrt_rn(add1(var, 1e-5f));
    };
    u4 pc[3], pn[3];                                       // the A fragments (three bf16 terms) of the step being multiplied / of the next
    // half `part` (values 4 part .. + 4) of chunk c -> pn
    auto fragpart = [&](auto c_c, auto part_c, auto fast_c) __attribute__((always_inline)) {
      constexpr int c = decltype(c_c)::value, part = decltype(part_c)::value;
      constexpr bool FAST = decltype(fast_c)::value;
      if (FR_ABL(1)) { asm volatile("" : "+v"(pn[0]), "+v"(pn[1]), "+v"(pn[2])); return; }
      float v[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = 4 * part + jj;
        int fb, sj, a, f;
        if constexpr (c < 4 * NFBF) { fb = c / 4; sj = j; a = c % 4; f = feat0(fb) + j; }
        else { fb = NFBF; sj = j & 3; a = 2 * (c - 4 * NFBF) + (j >> 2); f = feat0(fb) + (j & 3); }
        v[jj] = FAST ? stat_fast(fb, sj, a) : stat(fb, sj, a, f);
        if constexpr (DUMP) {
          if (!(fb == NB - 1 && f < fl_nom)) g.agg_out[(size_t)(Tm * 64 + m * 16 + li) * g.ld_agg + a * g.F + f] = v[jj];
        }
      }
      const bool inf = !FAST && __builtin_amdgcn_ballot_w64(pna_dev::vmax(pna_dev::vmax(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])),
                                                                           pna_dev::vmax(__builtin_fabsf(v[2]), __builtin_fabsf(v[3]))) == INFINITY) != 0;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {                      // (pna_x3_split.h: split8 / split8_inf, one pair of values at a time)
        const float xe = v[2 * h2], xo = v[2 * h2 + 1];
        const bool ie = inf && __builtin_fabsf(xe) == INFINITY, io = inf && __builtin_fabsf(xo) == INFINITY;
        const float fe = ie ? 0.f : xe, fo = io ? 0.f : xo;
        const float re = fe - top16(fe), ro = fo - top16(fo);
        const float se = re - top16(re), so = ro - top16(ro);
        pn[0][2 * part + h2] = pack_hi(fe, fo);
        pn[1][2 * part + h2] = pack_hi(re, ro);
        pn[2][2 * part + h2] = pack_hi(ie ? xe : se, io ? xo : so);
      }
    };

    i4 td_cur, td_nxt;
    auto desc_off = [&](int tt) -> unsigned { return (unsigned)min(tt, T1 - 1) * 16u; };
    sld16(td_cur, g.wdesc, desc_off(T0));
    sld16(td_nxt, g.wdesc, desc_off(T0 + 1));
    long ib_cur = (long)td_cur.z * g.img_stride, ib_next = (long)td_nxt.z * g.img_stride;
    if (FR_ABL(6)) { ib_cur = 0; ib_next = 0; }            // (experiment: every tile multiplies the FIRST group's image: one image live in L2)
    int T = T0, k = 0, buf = 0;
#ifdef PNA_AMD_EXPERIMENTS
    if (g.prio_m == 0) __builtin_amdgcn_s_setprio(0); else if (g.prio_m == 1) __builtin_amdgcn_s_setprio(1); else if (g.prio_m == 3) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2);
#else
    __builtin_amdgcn_s_setprio(2);                         // (its rare memory instructions go ahead of the gather's: measured)
#endif
    ld4(pr_next, g.perm, (unsigned)(T * 64 + m * 16 + li) * 4u);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pr_next) : : "memory");
    // The (tile, chunk) pipeline (F = 75: kNBuf = 7 buffers, kAhead = 5).  Step k reads LDS buffer k % 7.  At the START of step k every M wavefront has arrived (their
    // pieces of step k's image have landed; each is past the middle of step k - 1, i.e. done with step k - 2): buffer (k - 2) % 7 is
    // free and step k + 5's image is requested into it.  In the MIDDLE of step k a wavefront waits for ITS pieces of step k + 1
    // (requested at the start of step k - 4: four younger images stay in flight) and arrives.  The copies queue behind the gather
    // wavefronts' requests in the CU's memory pipeline (4 us under load, measured: with 2.5 steps of cover the M side ran at
    // 1.6 us per step): hence seven buffers, and a gather ring no deeper than the bandwidth needs.
#pragma unroll
    for (int c = 0; c < kAhead; ++c) {
      if (c < NC) stage(c, c, ib_cur); else stage(c - NC, c, ib_next);
    }
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"((kAhead - 1) * kNI) : "memory");       // this wavefront's pieces of step 0
    arrive();

    while (true) {
      lap(6);
      // ---- the tile's raw statistics from the G wavefront, two phases ----------------------------------------------------------
      deg = td_cur.y; Tm = T;
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        const unsigned long long tp = now();
        int n = 0;
        while (lds_peek(a_full) < 2 * k + ph + 1) {
          if (++n > kSpin) { g.err[0] = 3; dead = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        tw0 += now() - tp; spins += n;
        if (dead) break;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int pc = 0; pc < NP; ++pc)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(raw[2 * ph + q][pc]) : "v"(a_stats), "n"((q * NP + pc) * 1024) : "memory");
        static_assert(NP >= 2 && NP <= 5, "operand lists of the waits below");
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          f4 (&r)[NP] = raw[2 * ph + q];
          if constexpr (NP == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]) : : "memory");
          if constexpr (NP == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[NP > 2 ? 2 : 0]) : : "memory");
          if constexpr (NP == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[NP > 2 ? 2 : 0]), "+v"(r[NP > 3 ? 3 : 0]) : : "memory");
          if constexpr (NP == 5) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[NP > 2 ? 2 : 0]), "+v"(r[NP > 3 ? 3 : 0]), "+v"(r[NP > 4 ? 4 : 0]) : : "memory");
        }
        if (lane == 0) lds_poke(a_empty, 2 * k + ph + 1);
      }
      if (dead) break;
      ++k;
      lap(0);
      // ---- the tile's rows: residual now (lands under the first steps), the next tile's nodes ------------------------------
      pr = pr_next;
      {
        const unsigned rrow = (unsigned)max(pr, 0) * g.ldrb;
#pragma unroll
        for (int n = 0; n < NT; ++n) ld16s(res[n], resb, rrow + (unsigned)max(0, min(n * 16 + 4 * lg, g.N - 4)) * 4u);
        ld4(pr_next, g.perm, (unsigned)(min(T + 1, T1 - 1) * 64 + m * 16 + li) * 4u);
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = (f4){0.f, 0.f, 0.f, 0.f};
      // a tile is FAST when every sum of squares of the lane's features is finite (then every message was) and it has in-edges
      float qs = 0.f;
#pragma unroll
      for (int pc_ = 0; pc_ < NP; ++pc_)
#pragma unroll
        for (int e = 0; e < 4; ++e) qs = qs + raw[1][pc_][e];
      const bool fast_tile = deg > 0 && __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(qs) < INFINITY)) == 0;
      lap(1);

      // One step = one chunk of 32 k values: 30 MFMAs (column tiles in pairs (0,1) (2,3) (4), alternating accumulators; the
      // TRANSPOSED product: the weight fragment is the MFMA's A operand, the statistics' the B operand) with the NEXT chunk's
      // statistics finished and split in two halves behind the first two MFMA groups (VALU beside the matrix pipe), the next
      // pair's weight fragments requested behind each group, and at its end: every M wavefront has arrived for the next step,
      // the image kAhead steps ahead is requested, the next step's first fragments are requested.
      bf8 B[2][3];
      constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
#define FR_READ_B(slot, n)                                                                                                                  \
      _Pragma("unroll") for (int tm_ = 0; tm_ < 3; ++tm_)                                                                                   \
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(B[slot][tm_]) : "v"(ba0), "n"(tm_ * 4 * kNW * 16 + (n) * 256) : "memory")
#define FR_WAIT_B() asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(B[0][0]), "+v"(B[0][1]), "+v"(B[0][2]), "+v"(B[1][0]), "+v"(B[1][1]), "+v"(B[1][2]) : : "memory")
#define FR_MM(n, slot) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[slot][TB[pp]], __builtin_bit_cast(bf8, pc[TA[pp]]), acc[n], 0, 0, 0)
      unsigned ba0 = 0;
      // the step's opening: all arrived, the image kAhead steps ahead requested, the step's first fragments requested
      auto open_step = [&](auto s_c) __attribute__((always_inline)) {
        constexpr int s = decltype(s_c)::value;             // the step being opened (s == NC: the next tile's step 0 is opened by that tile)
        tsq = now();
        waitall();
        slap(0);
        {
          const int bufn = buf + kAhead >= kNBuf ? buf + kAhead - kNBuf : buf + kAhead;
          if (s + kAhead < NC) stage(s + kAhead, bufn, ib_cur);
          else stage(s + kAhead - NC, bufn, ib_next);
        }
        static_assert(kAhead <= NC, "a step's image lies in this tile or the next");
        slap(1);
        ba0 = lds0 + (unsigned)(buf * kChunkV + lg * kNW + li) * 16u;
        if (!FR_ABL(4)) { FR_READ_B(0, 0); FR_READ_B(1, 1); }
      };
      auto step = [&](auto s_c, auto fast_c) __attribute__((always_inline)) {
        constexpr int s = decltype(s_c)::value;
        using CN = std::integral_constant<int, (s + 1 < NC ? s + 1 : 0)>;
        tsq = now();
        FR_WAIT_B();
        if (!FR_ABL(0))
#pragma unroll
        for (int pp = 0; pp < 6; ++pp) { FR_MM(0, 0); FR_MM(1, 1); }
        if (!FR_ABL(4)) { FR_READ_B(0, 2); FR_READ_B(1, 3); }
        if constexpr (s + 1 < NC) fragpart(CN{}, std::integral_constant<int, 0>{}, fast_c);
        FR_WAIT_B();
        // this wavefront's pieces of step s + 1; younger: the images of the next kAhead - 1 steps and, in a tile's first kAhead - 1
        // steps, its residual rows and the next tile's nodes (requested at the tile's start, before step 0's image request)
        constexpr int young = s < kAhead - 1 ? NT + 1 : 0;
        slap(2);
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"((kAhead - 1) * kNI + young) : "memory");
        slap(3);
        arrive();
        if (!FR_ABL(0))
#pragma unroll
        for (int pp = 0; pp < 6; ++pp) { FR_MM(2, 0); FR_MM(3, 1); }
        if (!FR_ABL(4)) { FR_READ_B(0, 4); }
        if constexpr (s + 1 < NC) fragpart(CN{}, std::integral_constant<int, 1>{}, fast_c);
        FR_WAIT_B();
        if (!FR_ABL(0))
#pragma unroll
        for (int pp = 0; pp < 6; ++pp) FR_MM(4, 0);
        slap(4);
        buf = buf == kNBuf - 1 ? 0 : buf + 1;
        if constexpr (s + 1 < NC) {
          open_step(CN{});
          pc[0] = pn[0]; pc[1] = pn[1]; pc[2] = pn[2];
        }
      };
      auto steps = [&](auto fast_c) __attribute__((always_inline)) {
        fragpart(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, fast_c);
        fragpart(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, fast_c);
        pc[0] = pn[0]; pc[1] = pn[1]; pc[2] = pn[2];
        open_step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 0>{}, fast_c); step(std::integral_constant<int, 1>{}, fast_c);
        step(std::integral_constant<int, 2>{}, fast_c); step(std::integral_constant<int, 3>{}, fast_c);
        if constexpr (NC > 4) { step(std::integral_constant<int, (NC > 4 ? 4 : 0)>{}, fast_c); step(std::integral_constant<int, (NC > 4 ? 5 : 0)>{}, fast_c); }
        if constexpr (NC > 6) { step(std::integral_constant<int, (NC > 6 ? 6 : 0)>{}, fast_c); step(std::integral_constant<int, (NC > 6 ? 7 : 0)>{}, fast_c); }
        if constexpr (NC > 8) { step(std::integral_constant<int, (NC > 8 ? 8 : 0)>{}, fast_c); step(std::integral_constant<int, (NC > 8 ? 9 : 0)>{}, fast_c); }
      };
      static_assert(NT == 5, "the column-tile pairing of a step");
      if (!FR_ABL(5)) {
        if (fast_tile) steps(std::true_type{});
        else steps(std::false_type{});
      }
#undef FR_READ_B
#undef FR_WAIT_B
#undef FR_MM
      if (dead) break;
      lap(2);
      // the residual rows and the next tile's nodes are older than the last kAhead - 1 image requests (every shape has >= 4 steps)
      if (!FR_ABL(5)) asm volatile("s_waitcnt vmcnt(%6)" : "+v"(res[0]), "+v"(res[1]), "+v"(res[2]), "+v"(res[3]), "+v"(res[4]), "+v"(pr_next) : "n"((kAhead - 1) * kNI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" : "+v"(res[0]), "+v"(res[1]), "+v"(res[2]), "+v"(res[3]), "+v"(res[4]), "+v"(pr_next) : : "memory");
      lap(3);

      // ---- epilogue: lane (li, lg) holds columns 16 n + 4 lg .. + 4 of row li: bias, BatchNorm scale / shift, ReLU, residual ----
      {
        const float lo = g.relu ? 0.f : -INFINITY;
        const bool leaky = g.relu == 2;
        const int row = pr;
        char* const yrow = reinterpret_cast<char*>(g.y) + (size_t)(unsigned)max(row, 0) * g.ldyb;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          f4 cb, cs, ct;
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(cb) : "v"(a_colc), "n"(n * 64) : "memory");
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(cs) : "v"(a_colc), "n"(kNW * 4 + n * 64) : "memory");
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ct) : "v"(a_colc), "n"(2 * kNW * 4 + n * 64) : "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cb), "+v"(cs), "+v"(ct) : : "memory");
          const int c0 = n * 16 + 4 * lg;
          float z[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float t = acc[n][r] + cb[r];
            t = __builtin_fmaf(t, cs[r], ct[r]);
            z[r] = t < lo ? (leaky ? t * g.slope : 0.f) : t; // ReLU / LeakyReLU / none (lo = -inf); NaN < lo is false: NaN is kept
          }
          if (has_res) {
            const f4 rr = fix4(c0, g.N, res[n]);
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] = rr[j] + z[j];
          }
          if (row >= 0 && c0 < g.N && !FR_ABL(3)) {
            float* const o = reinterpret_cast<float*>(yrow + (unsigned)c0 * 4u);
            if (c0 + 4 <= g.N) {
              f4u wv; wv.v = (f4){z[0], z[1], z[2], z[3]};
              *reinterpret_cast<f4u*>(o) = wv;
            } else {                                          // the row's last, partial window
              o[0] = z[0];
              if (c0 + 1 < g.N) o[1] = z[1];
              if (c0 + 2 < g.N) o[2] = z[2];
            }
          }
        }
      }
      lap(4);
      if (++T == T1) break;
      td_cur = td_nxt;
      sld16(td_nxt, g.wdesc, desc_off(T + 1));
      ib_cur = ib_next;
      ib_next = FR_ABL(6) ? 0 : (long)td_nxt.z * g.img_stride;
      lap(5);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the copies issued for steps that do not exist
#ifdef PNA_AMD_EXPERIMENTS
    if (g.dbg && lane == 0) {
      unsigned long long* d = g.dbg + ((size_t)gridDim.x * (kG + kM) * 4) + ((size_t)blockIdx.x * kM + m) * 8;
      for (int i = 0; i < 7; ++i) d[i] = tph[i];
      unsigned long long* d2 = g.dbg + ((size_t)gridDim.x * (kG + kM) * 4) + (size_t)gridDim.x * kM * 8 + ((size_t)blockIdx.x * kM + m) * 8;
      for (int i = 0; i < 5; ++i) d2[i] = tst[i];
    }
#endif
  }
#ifdef PNA_AMD_EXPERIMENTS
  if (g.dbg && lane == 0) {
    unsigned long long* d = g.dbg + ((size_t)blockIdx.x * (kG + kM) + wave) * 4;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    d[0] = tw0; d[1] = tw1; d[2] = ((unsigned long long)spins << 32) | hw; d[3] = now() - t00;
  }
#endif
}

template <int NFBF, bool HALF, bool DUMP, int RING>
int launch(const FRArgs& g, int wgs, hipStream_t st) {
  constexpr int NC = 4 * NFBF + (HALF ? 2 : 0), NP = 2 * NFBF + (HALF ? 1 : 0);
  const size_t lds = Lds<NC, NP>::total;
  auto* fn = k_fused_roles<NFBF, HALF, DUMP, RING>;
  if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
  hipLaunchKernelGGL(fn, dim3((unsigned)wgs), dim3(kThreads), lds, st, g);
  return 0;
}
// ring depth: the gather's throughput does not depend on it between 3 and 6 packets (measured at C3); everything of the gather
// wavefronts that is in flight stands in front of the M wavefronts' weight copies in the CU's memory pipeline, so: as few as
// the bandwidth needs
template <bool DUMP>
int launch_shape(const FRArgs& g, int wgs, int ring, hipStream_t st) {
  const int nf = shape_full(g.F);
  const bool half = shape_half(g.F);
#ifdef PNA_AMD_EXPERIMENTS
  if constexpr (!DUMP) {
    if (nf == 2 && half && ring == 2) return launch<2, true, false, 2>(g, wgs, st);
    if (nf == 2 && half && ring == 4) return launch<2, true, false, 4>(g, wgs, st);
    if (nf == 2 && half && ring == 5) return launch<2, true, false, 5>(g, wgs, st);
  }
#endif
  (void)ring;
  if (nf == 1 && !half) return launch<1, false, DUMP, 3>(g, wgs, st);
  if (nf == 1 && half) return launch<1, true, DUMP, 3>(g, wgs, st);
  if (nf == 2 && !half) return launch<2, false, DUMP, 3>(g, wgs, st);
  if (nf == 2 && half) return launch<2, true, DUMP, 3>(g, wgs, st);
  return -2;
}

}  // namespace

extern "C" int32_t pna_fused_roles_supported(int32_t F, int32_t N) {
  return F >= 17 && F <= 80 && N >= 4 && N <= kNW ? 1 : 0;
}

extern "C" int64_t pna_fused_roles_image_bytes(int32_t F, int32_t N) {   // (the images are pna_fused_degree_pack_f32's)
  return pna_fused_roles_supported(F, N) ? pna_fused_degree_image_bytes(F, N) : 0;
}

extern "C" int32_t pna_fused_roles_grid(int32_t spare_units) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 0;
  int wgs = cus - (spare_units > 0 ? spare_units : 0);
  return wgs < 1 ? 1 : wgs;
}

extern "C" int pna_fused_roles_f32(const pna_fused_roles_args* p, pna_stream_t stream) {
  if (!p) return pna_set_error(PNA_E_INVALID, "pna_fused_roles_f32: null args");
  if (p->struct_size < sizeof(pna_fused_roles_args)) return pna_set_error(PNA_E_INVALID, "pna_fused_roles_f32: args.struct_size is smaller than this library's pna_fused_roles_args");
  if (p->n_tiles == 0) return PNA_OK;
  if (!p->tile_desc || !p->tile_ids || !p->wg_range || !p->x || !p->row_perm || !p->w_img || !p->y || !p->err)
    return pna_set_error(PNA_E_INVALID, "pna_fused_roles_f32: tile_desc / tile_ids / wg_range / x / row_perm / w_img / y / err must be non-null");
  if (!pna_fused_roles_supported(p->F, p->N)) return pna_set_error(PNA_E_INVALID, "pna_fused_roles_f32: F in 17..80, N in 4..80");
  if (p->ldx < p->F || ((uintptr_t)p->x & 3) != 0 || (int64_t)p->ldx * 4 >= (1ll << 31))
    return pna_set_error(PNA_E_INVALID, "pna_fused_roles_f32: ldx >= F, x 4-byte aligned (and readable up to the last row's rounded-up 16-byte strip)");
  if (p->x_rows < 1 || p->x_rows >= (1ll << 32)) return pna_set_error(PNA_E_INVALID, "pna_fused_roles_f32: 1 <= x_rows < 2^32");
  if (p->n_tiles < 0 || p->n_tiles * 64 >= (1ll << 31)) return pna_set_error(PNA_E_INVALID, "pna_fused_roles_f32: bad n_tiles");
  if (p->n_records < 1 || p->ids_stride < (p->n_records + 24) * 64 || (p->n_records + 24) * 64 >= (1ll << 32))
    return pna_set_error(PNA_E_INVALID, "pna_fused_roles_f32: each id array holds n_records + 24 (padding) records of 64 bytes, < 4 GiB, ids_stride bytes apart");
  if (p->n_nodes < 1 || p->ldy < p->N || p->n_nodes * p->ldy * 4 >= (1ll << 32) ||
      (p->residual && (p->ld_res < p->N || p->n_nodes * p->ld_res * 4 >= (1ll << 32))) || p->image_stride < pna_fused_roles_image_bytes(p->F, p->N))
    return pna_set_error(PNA_E_INVALID, "pna_fused_roles_f32: bad n_nodes / ldy / ld_res / image_stride (y and residual must be < 4 GiB)");
  if (p->relu < 0 || p->relu > 2) return pna_set_error(PNA_E_INVALID, "pna_fused_roles_f32: relu must be 0, 1 or 2");
  if (p->n_workgroups < 1) return pna_set_error(PNA_E_INVALID, "pna_fused_roles_f32: n_workgroups (= pairs in wg_range) must be >= 1");
  if ((p->col_scale == nullptr) != (p->col_shift == nullptr)) return pna_set_error(PNA_E_INVALID, "pna_fused_roles_f32: col_scale and col_shift come together");
  if (p->agg_out && p->ld_agg < 4 * (int64_t)p->F) return pna_set_error(PNA_E_INVALID, "pna_fused_roles_f32: ld_agg < 4 F");
  FRArgs g;
  memset(&g, 0, sizeof(g));
  g.wdesc = reinterpret_cast<const i4*>(p->tile_desc); g.ids = p->tile_ids; g.ids_stride = p->ids_stride; g.wg_range = p->wg_range;
  g.x = reinterpret_cast<const char*>(p->x); g.ldb = (unsigned)(p->ldx * 4); g.F = p->F;
  g.perm = p->row_perm; g.w_img = (const unsigned char*)p->w_img; g.img_stride = p->image_stride;
  g.bias = p->bias; g.col_scale = p->col_scale; g.col_shift = p->col_shift; g.residual = p->residual; g.y = p->y;
  g.ldyb = (unsigned)(p->ldy * 4); g.ldrb = p->residual ? (unsigned)(p->ld_res * 4) : 0u;
  g.N = p->N; g.relu = p->relu; g.slope = p->relu == 2 ? p->act_slope : 0.f;
  g.agg_out = p->agg_out; g.ld_agg = p->ld_agg; g.err = p->err;
  int ring = 3;
#ifdef PNA_AMD_EXPERIMENTS
  if (const char* e = getenv("PNA_FR_DBG_PTR")) g.dbg = (unsigned long long*)strtoull(e, nullptr, 0);   // device buffer: 4 counters per wavefront
  if (const char* e = getenv("PNA_FR_ABL")) g.abl = atoi(e);
  if (const char* e = getenv("PNA_FR_RING")) ring = atoi(e);
  g.prio_g = 0; g.prio_m = 2;
  if (const char* e = getenv("PNA_FR_PRIO_G")) g.prio_g = atoi(e);
  if (const char* e = getenv("PNA_FR_PRIO_M")) g.prio_m = atoi(e);
#endif
  hipStream_t st = (hipStream_t)stream;
  const int rc = p->agg_out ? launch_shape<true>(g, p->n_workgroups, ring, st) : launch_shape<false>(g, p->n_workgroups, ring, st);
  if (rc != 0) return pna_set_error(PNA_E_LAUNCH, rc == -2 ? "pna_fused_roles_f32: no instantiation for this F" : "pna_fused_roles_f32: hipFuncSetAttribute failed");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}
