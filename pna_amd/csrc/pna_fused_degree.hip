// pna_fused_degree.hip -- PNASimpleLayer forward (models/dgl/pna_layer.py:186-216) as ONE kernel on degree-ordered rows: the
// gather, the four aggregators, the degree scalers and the posttrans contraction; the (V, 4F) aggregate never reaches HBM
// (the reference materialises (V, 12F): pna_layer.py:189-194, :206).  Implements pna_fused_degree_{image_bytes,pack_f32,f32}.
//
//   y[perm[v]] = epilogue( bias + W_D . [mean | max | min | std](x[src] over the in-edges of perm[v]) ),  W_D = sum_s scale_s(D) W_s
//
// Every PNA scaler is a function of the destination's in-degree alone (models/dgl/scalers.py:7-19), so rows of one in-degree D
// share ONE combined weight W_D.  The host orders the rows by in-degree (pna_amd/degree_groups.py): a wavefront
// owns 16 rows of one degree, so its gather loop is uniform -- D iterations for every lane, no divergence, no per-row tail.
//
// Gather.  Lane (li = lane & 15, lg = lane >> 4) keeps sum / sum of squares / max / min of features fb * 32 + lg * 8 .. + 8 of row
// li for every feature block fb: the four lanes of a row read one 128-byte strip of a source row per block, all blocks of a row
// back to back (the row's DRAM page is touched once).  The source ids come from a TILE-MAJOR edge list the plan builds once per
// graph -- record (tile, e) = the e-th source of each of the tile's 16 rows, 64 contiguous bytes -- so no id depends on a rowptr
// lookup.  One edge = one "packet" of 2 NFB + 1 loads (the strips, then the id this ring slot needs next); FOUR packets ride in a
// register ring, three always in flight behind the one being folded: every load is inline asm with a counted s_waitcnt (VMEM
// returns in order; the count names only this kernel's own younger loads, so compiler-issued stores / copies in between can only
// make a wait stricter).  The fold is the production gather's (pna_segreduce.hip: s += m, q += m * m, v_max / v_min, edge
// order; mean = s / D correctly rounded) as single VALU instructions: the statistics are the same bits as the two-kernel path's.
//
// Contraction.  The statistics ARE the MFMA A operand: with K ordered (feature block, aggregator) -- the weight image is packed
// to match -- chunk c = 4 fb + a multiplies the lane's eight values of aggregator a, split into TWO fp16 terms after a power-of-two
// row scale (round 5, pna_x3_split.h: three partial products per multiply, fp32 accumulation, the accuracy of the bf16 x 3 form of
// rounds 3-4 -- six products).  One combined
// image per degree group, streamed through five LDS buffers by global_load_lds (four steps ahead, counted waits), one barrier per
// chunk in the middle of the chunk's MFMA stream.  Workgroups of 4 wavefronts (64 rows), two per CU: while one multiplies, the
// other gathers.  Tower mode (PNALayer with one tower) and the wide shapes (F or N up to 128) are template parameters below.
// Rows that no degree group holds (rare degrees, hub rows) stay on the two-kernel path over their compact list (host) -- on a
// second stream BESIDE this kernel when the caller asks it to leave some workgroups out (spare_workgroups, ABI 17: a persistent
// kernel that books every register of every CU starves whatever another stream launches).
//
// The running sums are folded by single v_add_f32 / v_mul_f32 instructions ON PURPOSE: written as plain C++, hipcc packs them into
// v_pk_add_f32 / v_pk_mul_f32 with op_sel swizzles, and a packed-fp32 instruction whose low lane reads src1's HIGH half drops its
// low-half result in lanes 48-63 while the SIMD's other wavefront issues MFMAs (tools/ubench/pk_opsel_mfma_repro.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "pna_amd.h"
#include "pna_internal.h"
#include "pna_rowstats.h"
#include "pna_x3_split.h"

namespace {

using namespace pna_x3;
using pna_dev::div_rn;

typedef int i4 __attribute__((ext_vector_type(4)));
typedef f4 f4a4 __attribute__((aligned(4)));

struct FDArgs {
  const i4* tdesc;             // per 16-row wavefront tile: {first id record, in-degree, weight image, 0}
  const int* ids;              // records of 16 source ids (tile-major edge list)
  const float* x;
  const int* perm; const unsigned char* w_img; long img_stride;
  const float* bias; const float* col_scale; const float* col_shift; const float* residual; float* y;
  float* agg_out; long ld_agg; // optional: the statistics as the contraction sees them, [mean | max | min | std] x F per virtual row
  const float* xd; const float* xh; const float* row_post;   // tower layers: the rows' own projections / features, the per-row factor
  unsigned lddb, ldhb;         // row pitch of xd / xh in bytes
  int* counter;                // dynamic tile schedule (round 5): two device int32, zero at launch (claims | finished workgroups); nullptr: tile b, b + G, b + 2 G, ...
  // the fp16 x 2 guard (round 6).  guard = {tiles handed over | workgroups of the consuming launch that have left | tiles handed over since
  // the caller last cleared it | guarded calls}: a workgroup tile whose outputs the floor-error bound does not certify is appended to the
  // hand-over list -- copies of its descriptors, its rows of perm (and of row_post) -- and the bf16 x 3 instantiation, launched behind this
  // one with the list as its tile tables, computes those tiles again (k_fused_degree; nullptr: no guard).
  int* guard; i4* g_desc; int* g_perm; float* g_post;
  const float* pre_add; unsigned ldpb;   // tower mode (round 6): rows added to the biased accumulator in front of the row factor (a layer of several
                                         // towers over the whole input: one launch per tower, the partial sums carried from launch to launch)
  int* list_count;             // the consuming launch (ARITH 1 behind a guarded one): its tile count lives on the device (= guard of the producer)
  unsigned long long* dbg;     // experiments build only: per-wavefront phase timers
  unsigned ldb;                // row pitch of x in bytes
  unsigned ldyb, ldrb;         // row pitch of y / residual in bytes (ldrb = 0 without a residual: every load reads y's first row)
  int F, M, N, relu;
  int ncw;                     // columns of a row of y the kernel may write (>= N; zeros behind N)
  float slope;
  int abl;                     // experiments build only: parts skipped for timing (bit 0 MFMAs, 1 fragment maths after the first chunk,
};                             // 2 the fold, 3 the y stores, 4 the B-fragment reads, 7 the weight copies, 8 gather from row 0): results are then meaningless

constexpr int kNW = 80, kNT = 5;
#ifndef FD_WIDE_WAVES
#define FD_WIDE_WAVES 4
#endif
// Wavefronts per workgroup (template parameter WAVES of the kernel; a workgroup tile = 16 WAVES rows of ONE degree = one weight
// image): 4, two workgroups per CU, for the shapes of one gather pass and one panel -- while one workgroup multiplies the other
// gathers, and the launch is bound by the memory system.  The WIDE shapes are not: their time is a per-workgroup
// chain of latencies (4.8.13), and the weight image is streamed once per tile -- FD_WIDE_WAVES wavefronts per workgroup there.
constexpr int waves_for(int gp, int npan) { return (gp == 2 || npan == 2) ? FD_WIDE_WAVES : 4; }
constexpr int kWavesMax = 8;
// LDS weight buffers (template parameter NBUF of the kernel): a step's image is requested NBUF - 1 steps before it is read.  With the
// bf16 x 3 images of rounds 3-4 five buffers of 15 KB (one panel of 80 columns) were what two workgroups per CU could hold, and six of
// the wide shapes' 12 KB panels of 64 columns; the fp16 x 2 images are 10 / 8 KB and more would fit -- 6 / 8 and 7 / 9 buffers measured
// the same as 5 / 6 on both benchmark shapes (-DFD_NBUF_REG / -DFD_NBUF_WIDE; profiles/README.md), so the counts stay.  Why six for the
// wide shapes (round 5, measured on the bf16 x 3 kernel): their multiply phase is paced by the COPY LATENCY, not by the
// matrix pipe (phase timers at BASELINE configs[4]'s shape: 73 % of a wavefront's time in the multiply phase against 21 % of
// matrix-pipe time; skipping every MFMA, fragment and B read leaves 2.55 of 3.26 ms), and a copy that has four steps to land instead
// of three shortens every paced step by a quarter.
#ifndef FD_NBUF_REG
#define FD_NBUF_REG 5
#endif
#ifndef FD_NBUF_WIDE
#define FD_NBUF_WIDE 6
#endif
// (nc: chunks of one gather pass -- a tile must have at least NBUF - 1 steps; tower mode, and the bf16 x 3 images of 15 / 12 KB: five / six)
constexpr int buffers_for(int gp, int npan, int nc = 4, bool tower = true, bool h2 = true) {
  return waves_for(gp, npan) == 8 ? 10 : tower ? 5 : !h2 ? (npan == 2 ? 6 : 5) : npan == 2 ? FD_NBUF_WIDE : (nc * gp >= FD_NBUF_REG - 1 + 2) ? FD_NBUF_REG : 5;
}
constexpr int kChunkV = 3 * 4 * kNW;                      // 16-byte pieces of one chunk image, tower mode: [term][lane group][80 cols][8 k] bf16 x 3
constexpr int kChunkVH = 2 * 4 * kNW;                     // ... of the layer proper (round 5): two fp16 terms (pna_x3_split.h)
constexpr int kTailBytes = 1024;                          // behind every fp16 image: 128 floats, 2^-s_n of the columns' power-of-two scales, and 128
                                                          // floats of the guard's per-column threshold (kFloorWeightScale x the column's floor error; -inf: nothing to certify)
constexpr int kRing = 4;                                  // edge packets in the register ring
constexpr int kNRes = kNT;                                // residual loads per lane (16 bytes each: 4 consecutive columns of one row)

// Feature blocks of a row: NFBF full blocks of 32 features (a lane owns 8: two 16-byte loads, four chunks -- one per aggregator)
// and, when the remainder is <= 16 features, a HALF block (a lane owns 4: one load, two chunks -- (mean | max), (min | std)).
// F = 75: 2 full + half = 5 loads per edge, 80 running statistics, 10 chunks (three full blocks: 6 loads, 96, 12).
// (round 6: 97 <= F <= 112 count as FOUR full blocks, two gather passes of two like 113..128 -- the last block's window slides back to end
// at F as everywhere, so only its first lane groups hold new features and the rest repeat earlier ones against zero weights: a fourth
// block of mostly idle multiply-adds, but the layer stays ONE kernel at the hidden sizes 100 and 110 of the reference's README.)
__host__ __device__ constexpr int shape_full(int F) { return F > 96 ? 4 : (F % 32 == 0 || F % 32 > 16) ? (F + 31) / 32 : F / 32; }
__host__ __device__ constexpr bool shape_half(int F) { return F <= 96 && F % 32 != 0 && F % 32 <= 16; }
__host__ __device__ constexpr int shape_chunks(int F) { return 4 * shape_full(F) + (shape_half(F) ? 2 : 0); }

// ---- inline-asm loads (hipcc neither counts nor waits for them: every wait below is ours) -----------------------------------
__device__ __forceinline__ void ld16(f4& dst, const void* base, unsigned voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void ld16_hi(f4& dst, const void* base, unsigned voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
// 64-bit lane address + immediate (the gather of the source table: no 4 GiB / 2^24-row limit, rows of any 4-byte aligned pitch)
template <int OFF>
__device__ __forceinline__ void ldx16(f4& dst, const char* p) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF) : "memory");
}
__device__ __forceinline__ void ld16i(i4& dst, const void* base, unsigned voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void ld4(int& dst, const void* base, unsigned voff) {
  asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void ld4f(float& dst, const void* base, unsigned voff) {
  asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
// The same loads behind five wait states.  "VALU writes SGPR -> VMEM reads that SGPR" needs them on gfx9 / CDNA; hipcc inserts the
// s_nop for its own memory instructions but does not look inside inline asm, and under SGPR pressure it reloads a spilled base
// pointer with v_readlane_b32 directly in front of the statement: the load then goes out with the stale base (the two-full-block
// tower instantiation faulted at address 0x1000 that way).  Used for the once-per-tile loads whose base pointers are not live in
// the gather loop; tools/isa_audit.py::sgpr_hazards checks EVERY inline-asm memory instruction of the compiled kernels for this.
__device__ __forceinline__ void ld16_ws(f4& dst, const void* base, unsigned voff) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void ld16_hi_ws(f4& dst, const void* base, unsigned voff) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:16" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void ld4_ws(int& dst, const void* base, unsigned voff) {
  asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void ld16i_ws(i4& dst, const void* base, unsigned voff) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
// Scalar load of a tile descriptor, waited for INSIDE the statement: an SGPR result that is still in flight when the statement
// ends may be SPILLED by hipcc (v_writelane of the stale value) -- the 2 full + half instantiation of the verification kernel did
// exactly that with a descriptor requested at the top of the gather and waited for at its end: wild record offsets, a memory
// fault.  (Nothing else of this wavefront is on the LGKM counter when these are issued.)
__device__ __forceinline__ void sld16(i4& dst, const void* base, unsigned soff) {
  asm volatile("s_load_dwordx4 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(dst) : "s"(base), "s"(soff) : "memory");
}
// wait until at most N of this wavefront's loads are outstanding; the slot's registers and its id become readable here
template <int N, int NL>
__device__ __forceinline__ void wait_slot(f4 (&s)[NL], int& id) {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit count");
  static_assert(NL >= 2 && NL <= 6, "2..6 row loads per edge");
  if constexpr (NL == 2)
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(s[0]), "+v"(s[1]), "+v"(id) : "n"(N) : "memory");
  else if constexpr (NL == 3)
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(id) : "n"(N) : "memory");
  else if constexpr (NL == 4)
    asm volatile("s_waitcnt vmcnt(%5)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(id) : "n"(N) : "memory");
  else if constexpr (NL == 5)
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(id) : "n"(N) : "memory");
  else
    asm volatile("s_waitcnt vmcnt(%7)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(id) : "n"(N) : "memory");
}
// the four messages of one 16-byte load into the running statistics of four features: the production fold as single VALU instructions.
// ONE asm statement per 16-byte load (round 5, last): between two inline-asm statements of which the second reads a register the first wrote, hipcc for
// gfx940+ inserts an `s_nop 0` -- it must assume the first ended in a dst_sel write (the forwarding hazard of SDWA / op_sel
// destinations), and it does not count other inline-asm statements in between as wait states.  Statement by statement this fold
// paid one wasted issue slot per message and feature (v_mul -> v_add), the operand split three per pair of values, the means' division
// two per value: ~700 of the ~6 000 instructions a wavefront issues per tile, in a kernel whose wide shapes are bound by instruction
// issue.  Inside one statement the instructions are ours: plain VALU results forward without wait states.
__device__ __forceinline__ void fold4(float* S, float* Q, float* MX, float* MN, const f4 m, const f4 ms) {
  float p0, p1, p2, p3;
  asm volatile("v_mul_f32 %16, %24, %24\n\tv_mul_f32 %17, %25, %25\n\tv_mul_f32 %18, %26, %26\n\tv_mul_f32 %19, %27, %27\n\t"
               "v_add_f32 %0, %0, %24\n\tv_add_f32 %1, %1, %25\n\tv_add_f32 %2, %2, %26\n\tv_add_f32 %3, %3, %27\n\t"
               "v_add_f32 %4, %4, %16\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %18\n\tv_add_f32 %7, %7, %19\n\t"
               "v_max_f32 %8, %8, %20\n\tv_max_f32 %9, %9, %21\n\tv_max_f32 %10, %10, %22\n\tv_max_f32 %11, %11, %23\n\t"
               "v_min_f32 %12, %12, %20\n\tv_min_f32 %13, %13, %21\n\tv_min_f32 %14, %14, %22\n\tv_min_f32 %15, %15, %23"
               : "+v"(S[0]), "+v"(S[1]), "+v"(S[2]), "+v"(S[3]), "+v"(Q[0]), "+v"(Q[1]), "+v"(Q[2]), "+v"(Q[3]),
                 "+v"(MX[0]), "+v"(MX[1]), "+v"(MX[2]), "+v"(MX[3]), "+v"(MN[0]), "+v"(MN[1]), "+v"(MN[2]), "+v"(MN[3]),
                 "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3)
               : "v"(m.x), "v"(m.y), "v"(m.z), "v"(m.w), "v"(ms.x), "v"(ms.y), "v"(ms.z), "v"(ms.w));
}

#ifdef PNA_AMD_EXPERIMENTS
#define FD_ABL(bit) ((g.abl >> (bit)) & 1)
#else
#define FD_ABL(bit) 0
#endif
__device__ __forceinline__ unsigned long long now() {
#ifdef PNA_AMD_EXPERIMENTS
  return __builtin_readcyclecounter();
#else
  return 0;
#endif
}

// RESPF: the tile's residual rows are requested during the gather's drain and ride through the multiply phase in 20 registers
// (not in the verification instantiation DUMP, whose stores need registers of their own).
//
// TOWER (PNALayer with one tower, models/dgl/pna_layer.py:33-76 + :130-145).  The message of edge (u, v) is the pretrans Linear of
// [h_u | h_v] = x_src[u] + x_dst[v] (two node-level projections): the gather runs over x_src alone and the destination's term
// enters through the contraction -- mean / max / min of (a_u + b) are mean / max / min of a_u plus b (rounding is monotone: the
// max / min are the same bits), the std does not see a shift -- as an extra K panel [x_dst[v]] against W_mean + W_max + W_min (not for
// rows without in-edges: DGL leaves their aggregate at zero).  A second extra panel carries the row's own features h[v] against the
// collapsed self weight (functional._tower_collapsed_weights).  Both are read like one more edge packet each -- the same strips,
// of the row's own node -- but late: after step 3, into the registers the first feature block's statistics have left, three
// steps before their first use; the residual rows follow four steps before the epilogue.  A per-row factor (graph norm) multiplies
// the biased accumulator in the epilogue.
//
// WIDE shapes (BASELINE configs[4]: F = 128, out_dim = 128).  GP = 2: the features in TWO gather passes of NFBF full blocks each
// (pass p: features p * 32 NFBF ..): the running statistics and the ring are re-used, the accumulators carry over -- gather 0,
// its chunks, gather 1 (the ids of the tile's edges a second time, its source rows' second half), its chunks, epilogue.  NPAN = 2:
// 81..128 output columns as two PANELS of 64 (4 column tiles each): a step multiplies one chunk's A fragment against one panel's
// image (12 KB: the five-buffer pipeline still fits two workgroups per CU), the fragment is formed once per chunk.
template <int NFBF, bool HALF, bool DUMP, bool TOWER = false, bool H2A = true, bool RESPF = !DUMP && !TOWER, int GP = 1, int NPAN = 1,
          int WAVES = waves_for(GP, NPAN), int NBUF = buffers_for(GP, NPAN, 4 * NFBF + (HALF ? 2 : 0), TOWER, H2A)>
__device__ __forceinline__ void fd_body(const FDArgs& g, const int t_first, const int t_stride, const int ntiles) {
  constexpr int kNBuf = NBUF, kAhead = NBUF - 1, kWaves = WAVES, kThreads = 64 * WAVES;
  static_assert(!TOWER || (NFBF == 2 && !DUMP), "tower mode: two full feature blocks (49 <= F <= 80), production only");
  static_assert((GP == 1 || (GP == 2 && !HALF && !TOWER)) && (NPAN == 1 || (NPAN == 2 && !TOWER)), "wide shapes: full blocks, no tower mode");
  constexpr int NB = NFBF + (HALF ? 1 : 0);               // feature blocks of one gather pass
  constexpr int NC = 4 * NFBF + (HALF ? 2 : 0);           // chunks of 32 k values of the statistics per gather pass
  constexpr int NPC = TOWER ? 2 * NFBF + (HALF ? 1 : 0) : 0;   // chunks of the two node panels (the half blocks of both share one)
  constexpr int NWP = NPAN == 1 ? kNW : 64;               // columns of one panel
  constexpr int NT = NWP / 16, NTA = NT * NPAN;           // column tiles per panel / accumulator tiles
  constexpr int NWA = NWP * NPAN;                         // all (padded) output columns
  // ARITHMETIC (round 5).  H2: statistics and weights as TWO fp16 terms each, three partial products per multiply (pna_x3_split.h):
  // half the MFMAs, two thirds of the weight stream and of the B-fragment reads of the bf16 x 3 form (six products), the same accuracy
  // class -- the row of statistics is scaled by a power of two so that its bound sits in [2^13, 2^14) (rscl below), the weights'
  // columns likewise at pack time (their 2^-s in the image's tail), the accumulator is scaled back in the epilogue.  Tower mode too
  // (H2A = false: bf16 x 3 as in rounds 3-4): its node panels enter the contraction as they come from memory, mid-tile, and the
  // row's scale is lowered to cover them when they land (panel_rescale below).
  // H2A = false (round 6): the bf16 x 3 arithmetic of rounds 3-4 -- componentwise fp32-accurate for operands of ANY dynamic range --, the
  // instantiation that re-computes the tiles the fp16 x 2 guard hands over (and the whole launch under PNA_FD_ARITH_X3).
  constexpr bool H2 = H2A;
  constexpr int NTERM = H2 ? 2 : 3, NPROD = H2 ? 3 : 6, NCC = H2 ? 5 : 3;
  using frag_t = std::conditional_t<H2, h8, bf8>;
  constexpr int CHV = NTERM * 4 * NWP;                    // 16-byte pieces of one step's image: [term][lane group][NWP cols][8 k]
  constexpr int NI = (CHV + kThreads - 1) / kThreads;     // global_load_lds instructions per wavefront per step
  constexpr int NSP = NC * NPAN;                          // steps of one gather pass
  constexpr int NCT = NSP * GP + NPC;                     // steps per tile
  static_assert(NT == 5 || NT == 4, "the column-tile pairing below");
  constexpr int PSTEP = 3;                                // the panels are requested at the end of this step ...
  constexpr int PWAIT = PSTEP + kAhead;                   // ... and have landed by this step's counted wait
  constexpr int RSTEP = NCT - kAhead;                     // tower mode: the residual rows are requested at the end of this step
  static_assert(!TOWER || (PWAIT < NC && RSTEP > PWAIT), "panel / residual windows");
  constexpr int NL = 2 * NFBF + (HALF ? 1 : 0);           // 16-byte row loads per edge
  constexpr int LB = NL + 1;                              // loads of one edge packet (the strips + the slot's next id)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int G = t_stride;                                  // the workgroup walks tiles t_first, t_first + G, .. below ntiles (a launch: blockIdx.x, its grid)
  const bool guard_on = H2 && g.guard != nullptr;

  float* const colc = reinterpret_cast<float*>(lds + (size_t)kNBuf * CHV * 16);           // [NCC][NWA]: bias | scale | shift | (H2) 2^-s of the column | (H2) guard threshold
  for (int i = tid; i < NWA; i += kThreads) {
    colc[i] = (g.bias && i < g.N) ? g.bias[i] : 0.f;
    colc[NWA + i] = (g.col_scale && i < g.N) ? g.col_scale[i] : 1.f;
    colc[2 * NWA + i] = (g.col_shift && i < g.N) ? g.col_shift[i] : 0.f;
    if constexpr (H2) {                                                                                        // (the first image's tail: the same in all)
      colc[3 * NWA + i] = reinterpret_cast<const float*>(g.w_img + (size_t)NCT * CHV * 16)[i];
      colc[4 * NWA + i] = i < g.N ? reinterpret_cast<const float*>(g.w_img + (size_t)NCT * CHV * 16)[128 + i] : -INFINITY;
    }
  }
  // GUARD: bit n of cmask = column tile n holds a column whose WEIGHTS have a floor error of their own (kernel-uniform; every wavefront
  // forms it from the tail: two loads, two ballots): only those tiles' outputs need certifying for rows without a floor error
  [[maybe_unused]] unsigned cmask = 0u;
  if constexpr (H2) {
    if (guard_on) {
      const float* const cvt = reinterpret_cast<const float*>(g.w_img + (size_t)NCT * CHV * 16) + 128;
      const unsigned long long b0 = __builtin_amdgcn_ballot_w64(lane < g.N && cvt[lane] > 0.f);
      const unsigned long long b1 = __builtin_amdgcn_ballot_w64(lane + 64 < g.N && cvt[lane + 64] > 0.f);
#pragma unroll
      for (int n = 0; n < 4; ++n) cmask |= (((b0 >> (16 * n)) & 0xffffull) ? 1u : 0u) << n | (((b1 >> (16 * n)) & 0xffffull) ? 1u : 0u) << (4 + n);
    }
  }
  if (tid == 0) {                                          // the guard's two hand-over words (behind the dynamic schedule's two)
    unsigned* const gw = reinterpret_cast<unsigned*>(lds + (size_t)kNBuf * CHV * 16 + (size_t)NCC * NWA * 4 + 8);
    gw[0] = 0u; gw[1] = 0u;
  }

  f4 acc[NTA];

  // ---- weight chunks: global -> LDS, asynchronously (every wavefront issues exactly kNI copies per chunk) ----------------
  auto stage = [&](int c, int buf, long ib) __attribute__((always_inline)) {
    if (FD_ABL(7)) return;                               // (experiments: no weight copies at all -- what the L2 -> LDS stream costs)
    const unsigned char* src = g.w_img + ib + (size_t)c * CHV * 16;
    unsigned char* dst = lds + (size_t)buf * CHV * 16;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int w0 = (i * kWaves + wave) * 64;
      if (w0 >= CHV) w0 = w0 % CHV;                      // a slot past the image re-copies an earlier piece (same bytes, same address)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(w0 + lane) * 16),
                                       (__attribute__((address_space(3))) void*)(dst + (size_t)w0 * 16), 16, 0, 0);
    }
  };

  // when a workgroup is through (also one that finds no tile): the dynamic schedule leaves its two words as it found them -- the LAST
  // workgroup to get here (every claim of the launch has been made by then) zeroes them for the next launch on the stream, no memset
  // node in front of every launch
  auto finish = [&]() __attribute__((always_inline)) {
    if (wave == 0 && lane == 0 && g.counter != nullptr &&
        __hip_atomic_fetch_add(g.counter + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
      __hip_atomic_store(g.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(g.counter + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  int t = t_first;                                        // the workgroup's current tile
  if (t >= ntiles) { finish(); return; }

  // ---- loop-carried across tiles: descriptors of this tile, of this wavefront's next tile and (in flight) of the one after;
  //      the ids of the tile's first four edges (the previous tile's last packets fetched them); the tile's rows of y ----------
  auto desc_off = [&](int tt) -> unsigned { return (unsigned)(min(tt, ntiles - 1) * kWaves + wave) * 16u; };
  i4 td_cur, td_nxt, td_n2;
  sld16(td_cur, g.tdesc, desc_off(t));
  sld16(td_nxt, g.tdesc, desc_off(t + G));
  td_n2 = td_nxt;
  // DYNAMIC TILE SCHEDULE (round 5, g.counter != nullptr).  Statically a workgroup walks tiles b, b + G, b + 2 G, ...; the phase timers
  // show the slowest workgroup 9 % behind the mean even when the tile list is cost-balanced (the memory system does not serve every CU
  // alike).  Dynamically the first FOUR tiles are the static ones and every later tile index is 4 G + a claim from a device-wide counter
  // (zero at launch, zeroed again by the last workgroup to finish): wavefront 0 claims an index at the start of a multiply phase (one atomic, never waited for: it is older than every load
  // the next gather waits on), hands it to the workgroup through one of two LDS words in that gather's drain, and everybody reads the
  // word behind the next multiply phase's barriers -- a claim is used four tiles after it was made, which is what the descriptor
  // prefetch (two tiles ahead) and the id prefetch (the previous tile's last packets) need.  Indices grow monotonically, so tiles
  // are still STARTED in list order device-wide: the list's degree order is the time order, one weight image live per L2.
  const bool dyn = g.counter != nullptr;
  int t_n1 = t + G, t_n2 = t + 2 * G;                     // the workgroup's next tile and the one after
  int claimed = t + 3 * G;                                // (wavefront 0, lane 0) the index claimed last: the tile after t_n2
  int par = 0;                                            // which LDS word the current tile's hand-over uses
  bool first_claim = true;                                // (wavefront 0) `claimed` still holds the static index t + 3 G
  const unsigned slot_b = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + (unsigned)(kNBuf * CHV * 16) + (unsigned)(NCC * NWA * 4);
  const unsigned lib = (unsigned)li * 4u;
  int idr[kRing];
  int pr = -1;                                            // the node (row of y / residual) of tile row li (-1: padding)
  int pn = 0;                                             // tower mode: the same, requested before the gather (the panels need it early)
  float rp = 1.f;                                         // tower mode: the per-row factor of tile row li
  f4 res[NTA];                                            // residual: row li, columns 16 n + 4 lg .. + 4

  // ---- the gather: running statistics of the wavefront's 16 rows ----------------------------------------------------------
  float S_[NB][8], Q_[NB][8], MX[NB][8], MN[NB][8];       // (a half block uses [0..3])
  int deg = 0;                                            // in-degree of the tile's rows (wave-uniform)
  bool fast_tile = false;                                 // the tile's statistics are all finite and it has in-edges (set by gather())
  int sA = 0;                                             // H2: the power of two row li's statistics are multiplied by (set by gather()) ...
  float rscl = 1.f, runs = 1.f;                           // ... 2^sA and 2^-sA
  // the fp16 x 2 GUARD (round 6; pna_x3_split.h): fex = the lane's FLOOR error so far -- the sum over its statistics of the
  // part of the two-term split's error that is absolute (a non-zero operand below 2^-3 in the row's units) instead of relative
  float fex = 0.f;
  int t_done = -1, gpar = 0;                              // the tile whose epilogue ran last, and which of the two hand-over words it used
  // The lane's strip of feature block fb starts at feature 32 fb + 8 lg (a half block: + 4 lg) -- except in the row's LAST block,
  // where a window that would reach past F slides back to end at F (round 4): no read ever leaves the row (rows of any pitch >= F,
  // the table's last row included), no statistic is ever made of padding.  A feature the slide covers twice counts once: the
  // pack kernel zeroes the weights of a lane group's slots below its nominal start (k_pack_fused_degree, same rule).
  constexpr int WL = HALF ? 4 : 8;                         // features per lane in the last block
  const int fl_nom = ((GP - 1) * NFBF + NB - 1) * 32 + lg * WL;             // (absolute feature: wide shapes gather in two passes)
  const int fl_abs = min(fl_nom, g.F - WL);
  unsigned f0b[NB];                                       // byte offset of the lane's strip of feature block fb inside a row (one pass)
#pragma unroll
  for (int fb = 0; fb < NB; ++fb) f0b[fb] = (unsigned)(fb * 32 + (HALF && fb == NFBF ? lg * 4 : lg * 8)) * 4u;
  if constexpr (GP == 1) f0b[NB - 1] = (unsigned)fl_abs * 4u;
  // first feature of the lane's slots of block fb in gather pass P, and whether slot j repeats a lower lane group's feature
  auto feat0 = [&](int P_, int fb) __attribute__((always_inline)) -> int {
    return (P_ == GP - 1 && fb == NB - 1) ? fl_abs : (P_ * NFBF + fb) * 32 + (HALF && fb == NFBF ? lg * 4 : lg * 8);
  };
  auto is_dup = [&](int P_, int fb, int f) __attribute__((always_inline)) -> bool { return P_ == GP - 1 && fb == NB - 1 && f < fl_nom; };
  const unsigned ldb = g.ldb;
  // lane addresses of the source strips (round 4: 64-bit, so that tables >= 4 GiB / >= 2^24 rows and contiguous (V, F) rows take this
  // kernel).  A: the lane's strip of block 0 of a row (block fb lies 128 fb bytes further: an immediate); L: the strip of the LAST
  // block of the last gather pass (slid back to end at F: fl_abs above).
  const char* const xA = reinterpret_cast<const char*>(g.x) + (size_t)lg * 32;
  const char* const xL = reinterpret_cast<const char*>(g.x) + (size_t)fl_abs * 4;
  const void* const resb = g.residual ? (const void*)g.residual : (const void*)g.y;
  const bool has_res = g.residual != nullptr;

  // The multiply computes the TRANSPOSED product (round 4): the weight fragment is the MFMA's A operand, the statistics' the B
  // operand, so lane (li, lg) holds columns 16 n + 4 lg .. + 4 of ROW li of column tile n -- 16 contiguous bytes of y: the residual
  // is 5 loads and y 5 stores of 16 bytes per lane with no transpose (round 3 transposed 4 x 4 blocks inside every quad of lanes:
  // two DPP butterflies, 32 VALU instructions per column tile).
  auto prow = [&]() __attribute__((always_inline)) -> int { return TOWER ? pn : pr; };
  // byte offset of the 16-byte window of column tile n inside a row; a window past N slides back to [N - 4, N) (realigned by fix4)
  auto res_col = [&](int n) __attribute__((always_inline)) -> unsigned { return (unsigned)max(0, min(n * 16 + 4 * lg, g.N - 4)) * 4u; };

  // P: the gather pass (features P * 32 NFBF ..); after a pass that is not the last, the ring's ids are refilled with the SAME
  // tile's first edges
  auto gather = [&](int t, auto p_c) __attribute__((always_inline)) {
    constexpr int P = decltype(p_c)::value;
    const int D = td_cur.y;
    const unsigned rb = (unsigned)td_cur.x * 64u;                                        // byte offset of this tile's records ...
    const unsigned rbn = P + 1 < GP ? rb : (unsigned)td_nxt.x * 64u;                     // ... and of the records the next gather starts with
    deg = D;
    if constexpr (P == 0) fex = 0.f;
    // (tower mode: the rows of y and their factors are needed from step RSTEP on only: requested with the panels)
    if constexpr (TOWER) ld4_ws(pn, g.perm, (unsigned)((t * kWaves + wave) * 16 + li) * 4u);
    else if constexpr (P == 0) ld4_ws(pr, g.perm, (unsigned)((t * kWaves + wave) * 16 + li) * 4u);
#pragma unroll
    for (int fb = 0; fb < NB; ++fb)
#pragma unroll
      for (int j = 0; j < 8; ++j) { S_[fb][j] = 0.f; Q_[fb][j] = 0.f; MX[fb][j] = -INFINITY; MN[fb][j] = INFINITY; }
    const int ng = max((D + 3) >> 2, 1);                  // groups of 4 edges; the plan pads the tile's records to 4 ng (repeats of the
                                                          // last edge: idempotent for max / min, masked out of the sums; a tile without
                                                          // in-edges has 4 records of row 0, all masked)
    f4 sl[kRing][NL];
    // rows of the edge whose id sits in idr[j] -> ring slot j, then the id slot j gathers next (record at byte `nrec`) -> idr[j]
    auto issue = [&](auto jc, unsigned nrec) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      constexpr bool LASTP = P == GP - 1;                 // the pass that holds the row's last block
      constexpr int NFA = LASTP ? NB - 1 : NB;             // blocks addressed through A (never past F)
      constexpr int PO = P * NFBF * 128;                    // byte offset of the pass's features inside a row
      const size_t ro = FD_ABL(8) ? (size_t)0 : (size_t)(unsigned)idr[j] * ldb;   // (experiments, bit 8: every packet reads row 0)
      if constexpr (NFA > 0) {
        const char* const pa = xA + ro;
        ldx16<PO>(sl[j][0], pa); ldx16<PO + 16>(sl[j][1], pa);
        if constexpr (NFA > 1) { ldx16<PO + 128>(sl[j][NFA > 1 ? 2 : 0], pa); ldx16<PO + 144>(sl[j][NFA > 1 ? 3 : 0], pa); }
        static_assert(NFA <= 2, "blocks addressed through A");
      }
      if constexpr (LASTP) {
        const char* const pl = xL + ro;
        ldx16<0>(sl[j][2 * (NB - 1)], pl);
        if constexpr (!HALF) ldx16<16>(sl[j][HALF ? 0 : 2 * (NB - 1) + 1], pl);
      }
      ld4_ws(idr[j], g.ids, nrec + (unsigned)j * 64u + lib);      // (_ws: under SGPR pressure the base of the ids is reloaded right in front)
    };
    auto fold = [&](auto jc, bool on) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      if (FD_ABL(2)) { asm volatile("" : "+v"(sl[j][0])); return; }
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const int fb = l >> 1, c0 = (l & 1) * 4;
        const f4 m = sl[j][l];
        const f4 ms = on ? m : (f4){0.f, 0.f, 0.f, 0.f};
        fold4(&S_[fb][c0], &Q_[fb][c0], &MX[fb][c0], &MN[fb][c0], m, ms);
      }
    };
    using J0 = std::integral_constant<int, 0>; using J1 = std::integral_constant<int, 1>;
    using J2 = std::integral_constant<int, 2>; using J3 = std::integral_constant<int, 3>;
    // the packets of edges 0..3; each refills its id with the next group's, or -- after the last group -- the NEXT tile's edge j
    const unsigned n0 = 1 < ng ? rb + 4u * 64u : rbn;
    issue(J0{}, n0); issue(J1{}, n0); issue(J2{}, n0); issue(J3{}, n0);
    if constexpr (P == 0) sld16(td_n2, g.tdesc, desc_off(t_n2));        // (the descriptor after next: its latency sits behind the packets just issued)
    // steady state: slot j holds edge 4 gi + j; behind it in flight: the three younger packets.  Every edge here is a real one.
    for (int gi = 0; gi + 1 < ng; ++gi) {
      const unsigned nr = gi + 2 < ng ? rb + (unsigned)(4 * gi + 8) * 64u : rbn;
      wait_slot<3 * LB, NL>(sl[0], idr[0]); fold(J0{}, true); issue(J0{}, nr);
      wait_slot<3 * LB, NL>(sl[1], idr[1]); fold(J1{}, true); issue(J1{}, nr);
      wait_slot<3 * LB, NL>(sl[2], idr[2]); fold(J2{}, true); issue(J2{}, nr);
      wait_slot<3 * LB, NL>(sl[3], idr[3]); fold(J3{}, true); issue(J3{}, nr);
    }
    // drain: the last group (its slots past D hold copies of edge D - 1).  The residual rows of the tile are requested behind
    // slot 0 -- perm (requested before every packet) has landed by then -- and stay in flight into the multiply phase.
    const int e0 = 4 * (ng - 1);
    wait_slot<3 * LB, NL>(sl[0], idr[0]);
    if constexpr (TOWER) asm volatile("" : "+v"(pn));
    else if constexpr (P == 0) asm volatile("" : "+v"(pr));
    fold(J0{}, e0 < D);
    constexpr bool RES_NOW = RESPF && P == GP - 1;        // (the residual rows ride behind the LAST pass's drain)
    constexpr int NR = RES_NOW ? NTA : 0;
    if constexpr (RES_NOW) {
      const unsigned rrow = (unsigned)max(prow(), 0) * g.ldrb;
#pragma unroll
      for (int n = 0; n < NTA; ++n) ld16_ws(res[n], resb, rrow + res_col(n));   // (_ws: under SGPR pressure the base is reloaded right in front)
    }
    wait_slot<2 * LB + NR, NL>(sl[1], idr[1]); fold(J1{}, e0 + 1 < D);
    wait_slot<1 * LB + NR, NL>(sl[2], idr[2]); fold(J2{}, e0 + 2 < D);
    wait_slot<NR, NL>(sl[3], idr[3]);          fold(J3{}, e0 + 3 < D);
    if constexpr (P == 0) {
      if (dyn && wave == 0) {                             // the claim made a multiply phase ago has landed (older than every load waited for above)
        asm volatile("" : "+v"(claimed));
        if (__builtin_amdgcn_readfirstlane(claimed) < 0)  // (belt and braces: the register was set to -1 when the claim was issued;
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(claimed) : : "memory");       //  never seen -- returns are in order)
        // (a claim returns the counter's value, 0, 1, 2, ..: tile index = 4 G + that; the first write hands over the static fourth tile)
        const int idx = first_claim ? claimed : claimed + 4 * G;
        if (lane == 0) asm volatile("ds_write_b32 %0, %1" : : "v"(slot_b + (unsigned)par * 4u), "v"(idx) : "memory");
        first_claim = false;
      }
    }
    // FAST tiles (round 4): every sum of squares of the lane's features finite (then every message was: the terms are >= 0) and
    // in-edges present -- all but pathological inputs.  Their statistics are finished without the special-value selects (frag()).
    {
      float qs = 0.f;
#pragma unroll
      for (int fb = 0; fb < NB; ++fb)
#pragma unroll
        for (int j = 0; j < ((HALF && fb == NFBF) ? 4 : 8); ++j) qs = qs + Q_[fb][j];
      // (not in the tower instantiation with a half block: 256 registers are taken there)
      fast_tile = !(TOWER && HALF) && D > 0 && __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(qs) < INFINITY)) == 0;
    }
    // H2: the row's power-of-two scale.  Every statistic of a feature is bounded by its largest message magnitude m (mean and min
    // inside [min, max]; std <= sqrt(mean of squares + 1e-5) <= m + 0.0032): bound = 2 max(m, 0.0032) over the row's features -- the
    // lane's own, then the row's other three lanes (li + 16 k: two ds_bpermute, no LDS memory touched) --, clamped to FLT_MAX (a row
    // that holds an infinite message is non-finite in every output column anyway).  2^sA puts the bound into [2^13, 2^14).
    if constexpr (H2) {
      float m = 0.f;
#pragma unroll
      for (int fb = 0; fb < NB; ++fb)
#pragma unroll
        for (int j = 0; j < ((HALF && fb == NFBF) ? 4 : 8); j += 4)                     // (one statement per four features: see fold4)
          asm("v_max3_f32 %0, %0, |%1|, |%2|\n\tv_max3_f32 %0, %0, |%3|, |%4|\n\tv_max3_f32 %0, %0, |%5|, |%6|\n\tv_max3_f32 %0, %0, |%7|, |%8|"
              : "+v"(m) : "v"(MX[fb][j]), "v"(MN[fb][j]), "v"(MX[fb][j + 1]), "v"(MN[fb][j + 1]), "v"(MX[fb][j + 2]), "v"(MN[fb][j + 2]),
                "v"(MX[fb][j + 3]), "v"(MN[fb][j + 3]));
      float o;
      asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(o) : "v"((unsigned)((lane ^ 16) * 4)), "v"(m) : "memory");
      asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(o));
      asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(o) : "v"((unsigned)((lane ^ 32) * 4)), "v"(m) : "memory");
      asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(o));
      const float bound = __builtin_fminf(__builtin_fmaxf(m + m, 0.0064f), 3.4028234663852886e38f);
      const int s1 = h2_row_scale_exp(bound);
      if constexpr (P == 0) {
        sA = s1;
      } else {
        // the second gather pass of a wide shape: ONE scale per row for both passes' products -- the smaller of the two; the accumulator
        // (in units of the first pass's scale) follows it down: a power of two <= 1, exact
        const int sN = min(sA, s1);
        const float fac = __builtin_ldexpf(1.0f, max(sN - sA, -126));
#pragma unroll
        for (int n = 0; n < NTA; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) asm("v_mul_f32 %0, %1, %2" : "=v"(acc[n][r]) : "v"(acc[n][r]), "v"(fac));
        sA = sN;
      }
      rscl = __builtin_ldexpf(1.0f, sA);
      runs = __builtin_ldexpf(1.0f, -sA);
    }
  };

  // ---- chunk c: the lane's eight A values, split into three bf16 terms.  Full block fb, chunk 4 fb + a: aggregator a (0 mean,
  //      1 max, 2 min, 3 std) of the lane's 8 features; half block, chunk 4 NFBF + h: aggregators 2h | 2h + 1 of its 4 features ----
  frag_t A[NTERM];
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
  auto stat = [&](int fb, int j, int a, int f, int deg) __attribute__((always_inline)) -> float {   // (deg: the tile's in-degree, see split_h2_guarded)
    const float D = (float)deg, invD = 1.0f / D;
    const float s = S_[fb][j], q = Q_[fb][j];
    float r;
    if (a == 0) {
      r = div_rn1(s, D, invD);
    } else if (a == 3) {
      const float mean = div_rn1(s, D, invD), msq = div_rn1(q, D, invD);
      float var = sub1(msq, mul1(mean, mean));
      var = var < 0.f ? 0.f : var;
      r = sqrtf(add1(var, 1e-5f));
    } else {
      const float e = a == 1 ? MX[fb][j] : MN[fb][j];
      r = q != q ? q : e;                                 // v_max / v_min drop NaN; q is NaN iff a message is (pna_rowstats.h)
    }
    if (deg <= 0) r = 0.f;                                // rows without in-edges: DGL leaves them at zero
    return r;                                             // (f: every slot holds a feature < F since round 4: no padding)
  };
  // The same arithmetic for a FAST tile, without div_rn's NaN / Inf fall-back, the NaN test of max / min, sqrtf's denormal scaling
  // and class test (var + 1e-5 is a normal number) and the padding / empty-row selects: the bits of stat().  (The mean is computed
  // again for the std: eight registers to keep it are not there at F = 75.)
  auto sqrt_rn = [&](float x) __attribute__((always_inline)) -> float {     // correctly rounded for normal x (hipcc's own sequence behind
    const float r = __builtin_amdgcn_sqrtf(x);                               // v_sqrt_f32, less the denormal scaling and the class test)
    const float rm = bfloat(fbits(r) - 1u), rp2 = bfloat(fbits(r) + 1u);
    const float e1 = fnma1(rm, r, x), e2 = fnma1(rp2, r, x);
    float o = e1 <= 0.f ? rm : r;
    o = e2 > 0.f ? rp2 : o;
    return o;
  };
  // div_rn without its NaN / Inf fall-back (q0 = a / D rounded, one Newton step), and var + 1e-5 of the std: mean = s / D, msq = q / D,
  // max(msq - mean * mean, 0) + 1e-5 -- stat()'s instructions for a FAST tile, several values per asm statement (see fold4):
  // four means / two variances per statement (consecutive one-value statements share scratch registers: a nop each, see fold4)
  auto div_fast4 = [&](const float* a, float D_, float invD_, float* r) __attribute__((always_inline)) {
    float q0, q1, q2, q3, t0, t1, t2, t3;
    asm("v_mul_f32 %4, %12, %17\n\tv_mul_f32 %5, %13, %17\n\tv_mul_f32 %6, %14, %17\n\tv_mul_f32 %7, %15, %17\n\t"
        "v_fma_f32 %8, -%16, %4, %12\n\tv_fma_f32 %9, -%16, %5, %13\n\tv_fma_f32 %10, -%16, %6, %14\n\tv_fma_f32 %11, -%16, %7, %15\n\t"
        "v_fma_f32 %0, %8, %17, %4\n\tv_fma_f32 %1, %9, %17, %5\n\tv_fma_f32 %2, %10, %17, %6\n\tv_fma_f32 %3, %11, %17, %7"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(D_), "v"(invD_));
  };
  auto var_eps_fast2 = [&](const float* s_, const float* q_, float D_, float invD_, float* r) __attribute__((always_inline)) {
    float a0, a1, m0, b0, b1, q0, c0, c1, m1, d0, d1, q1;
    asm("v_mul_f32 %2, %14, %19\n\tv_mul_f32 %5, %16, %19\n\tv_mul_f32 %8, %15, %19\n\tv_mul_f32 %11, %17, %19\n\t"
        "v_fma_f32 %3, -%18, %2, %14\n\tv_fma_f32 %6, -%18, %5, %16\n\tv_fma_f32 %9, -%18, %8, %15\n\tv_fma_f32 %12, -%18, %11, %17\n\t"
        "v_fma_f32 %4, %3, %19, %2\n\tv_fma_f32 %7, %6, %19, %5\n\tv_fma_f32 %10, %9, %19, %8\n\tv_fma_f32 %13, %12, %19, %11\n\t"
        "v_mul_f32 %0, %4, %4\n\tv_mul_f32 %1, %10, %10\n\t"
        "v_sub_f32 %0, %7, %0\n\tv_sub_f32 %1, %13, %1\n\t"
        "v_max_f32 %0, %0, 0\n\tv_max_f32 %1, %1, 0\n\t"
        "v_add_f32 %0, %0, %20\n\tv_add_f32 %1, %1, %20"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(a0), "=&v"(a1), "=&v"(m0), "=&v"(b0), "=&v"(b1), "=&v"(q0), "=&v"(c0), "=&v"(c1), "=&v"(m1), "=&v"(d0),
          "=&v"(d1), "=&v"(q1)
        : "v"(s_[0]), "v"(s_[1]), "v"(q_[0]), "v"(q_[1]), "v"(D_), "v"(invD_), "v"(1e-5f));
  };
  // the eight statistics of chunk c of a FAST tile: stat_fast value by value, in wider statements
  auto stats_fast8 = [&](auto c_c, float* v, int deg) __attribute__((always_inline)) {
    constexpr int c = decltype(c_c)::value;
    const float D = (float)deg, invD = 1.0f / D;
    auto stds = [&](int fb, int j0, int n, float* o) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < n; j += 2) {
        float x[2];
        var_eps_fast2(&S_[fb][j0 + j], &Q_[fb][j0 + j], D, invD, x);
        o[j] = sqrt_rn(x[0]); o[j + 1] = sqrt_rn(x[1]);
      }
    };
    if constexpr (c < 4 * NFBF) {
      constexpr int fb = c / 4, a = c % 4;
      if constexpr (a == 0) { div_fast4(&S_[fb][0], D, invD, v); div_fast4(&S_[fb][4], D, invD, v + 4); }
      else if constexpr (a == 3) stds(fb, 0, 8, v);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = a == 1 ? MX[fb][j] : MN[fb][j];
      }
    } else {                                              // the half block: (mean | max) then (min | std) of its four features
      constexpr int fb = NFBF, hh = c - 4 * NFBF;
      if constexpr (hh == 0) {
        div_fast4(&S_[fb][0], D, invD, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 + j] = MX[fb][j];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = MN[fb][j];
        stds(fb, 0, 4, v + 4);
      }
    }
  };
  f4 pk[2][NL];                                           // tower mode: strips of x_dst (0) and h (1) of the row's own node
  constexpr int NPL = 2 * NL + 1;                         // loads of the panel request: the strips, the rows' factors
  auto issue_panels = [&]() __attribute__((always_inline)) {
    const unsigned r0 = (unsigned)max(pn, 0);
    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(rp) : "v"((unsigned)((t * kWaves + wave) * 16 + li) * 4u), "s"(g.row_post) : "memory");
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const void* const base = p == 0 ? (const void*)g.xd : (const void*)g.xh;
      const unsigned rowb = __umul24(r0, p == 0 ? g.lddb : g.ldhb);
#pragma unroll
      for (int fb = 0; fb < NB; ++fb) {
        ld16_ws(pk[p][2 * fb], base, rowb + f0b[fb]);
        if (!(HALF && fb == NFBF)) ld16_hi_ws(pk[p][2 * fb + 1], base, rowb + f0b[fb]);
      }
    }
  };
  // Tower mode on fp16 x 2: the node panels (the row's own x_dst and h strips) land mid-tile; once they have (end of step PWAIT) the
  // row's scale is lowered to cover them too -- the smaller exponent of the two -- and the accumulator follows it down (a power of two
  // <= 1: exact); the chunks from PWAIT + 1 on, statistics and panels alike, are split in the new units.
  auto panel_rescale = [&]() __attribute__((always_inline)) {
    float m = 0.f;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(m) : "v"(m), "v"(pk[p][l].x), "v"(pk[p][l].y));
        asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(m) : "v"(m), "v"(pk[p][l].z), "v"(pk[p][l].w));
      }
    float o;
    asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(o) : "v"((unsigned)((lane ^ 16) * 4)), "v"(m) : "memory");
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(o));
    asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(o) : "v"((unsigned)((lane ^ 32) * 4)), "v"(m) : "memory");
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(o));
    const int sN = min(sA, h2_row_scale_exp(__builtin_fminf(__builtin_fmaxf(m, 1e-30f), 3.4028234663852886e38f)));
    const float fac = __builtin_ldexpf(1.0f, max(sN - sA, -126));
#pragma unroll
    for (int n = 0; n < NTA; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) asm("v_mul_f32 %0, %1, %2" : "=v"(acc[n][r]) : "v"(acc[n][r]), "v"(fac));
    sA = sN;
    rscl = __builtin_ldexpf(1.0f, sA);
    runs = __builtin_ldexpf(1.0f, -sA);
  };
  // the two-term split of a chunk's eight values with the guard's bookkeeping: one more small chunk when any of them is non-zero and
  // below 2^kFloorExp in the row's units (a fragment that holds +-Inf: the row is non-finite in every column it reaches -- nothing to certify)
  // regen(x): the eight values once more, for the rare branch -- recomputed there from the running statistics (behind an opaque copy of the
  // degree, so that hipcc does not keep the first copies alive across the split instead: eight registers at the kernel's pressure peak)
  auto split_h2_guarded = [&](const f4 lo4, const f4 hi4, bool maybe_inf, auto regen) __attribute__((always_inline)) {
    if constexpr (H2) {
      if (maybe_inf && __builtin_amdgcn_ballot_w64(absmax8(lo4, hi4) == INFINITY) != 0) { split8_h2_inf(lo4, hi4, rscl, A[0], A[1]); return; }
      int fe = 127;
      split8_h2(lo4, hi4, rscl, A[0], A[1], fe);
      // one of the 8 x 64 values is small (rare: ~1 % of the rows of Gaussian features): its lanes add up the floor errors exactly
      if (__builtin_amdgcn_ballot_w64(fe <= kFloorExp) != 0) {
        float x[8];
        regen(x);
#pragma unroll
        for (int j = 0; j < 8; ++j) fex += h2_floor_error(mul_1(x[j], rscl));
      }
    }
  };
  auto frag = [&](auto c_c, auto p_c) __attribute__((always_inline)) {
    constexpr int c = decltype(c_c)::value;               // chunk within the gather pass P
    constexpr int P = decltype(p_c)::value;
    if ((c > 0 || P > 0) && FD_ABL(1)) { asm volatile("" : "+v"(A[0])); return; }
    if constexpr (TOWER) if constexpr (c >= NC) {         // a node panel (tower mode): the strips are the A operand as they come
      constexpr int pc = c - NC;
      constexpr bool halfc = pc >= 2 * NFBF;
      constexpr int p = halfc ? 0 : pc / NFBF, fb = halfc ? NFBF : pc % NFBF;
      f4 lo4 = pk[p][2 * fb], hi4 = halfc ? pk[1][2 * fb] : pk[p][2 * fb + 1];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if ((halfc || p == 0) && deg <= 0) lo4[j] = 0.f;     // (the x_dst panel: not for rows without in-edges)
        if (!halfc && p == 0 && deg <= 0) hi4[j] = 0.f;
      }
      if constexpr (H2) {                                 // (in units of the row's scale, which panel_rescale() made cover these strips)
        split_h2_guarded(lo4, hi4, true, [&](float* x) __attribute__((always_inline)) {
          x[0] = lo4.x; x[1] = lo4.y; x[2] = lo4.z; x[3] = lo4.w; x[4] = hi4.x; x[5] = hi4.y; x[6] = hi4.z; x[7] = hi4.w;
        });
      } else {
        if (__builtin_amdgcn_ballot_w64(absmax8(lo4, hi4) == INFINITY) != 0) split8_inf(lo4, hi4, A[0], A[1], A[2]);
        else split8(lo4, hi4, A[0], A[1], A[2]);
      }
      return;
    }
    float v[8];
    if (!(TOWER && HALF) && fast_tile) {
      if constexpr (c < NC) stats_fast8(c_c, v, deg);      // (c >= NC: tower panel chunks, handled above)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int fb, sj, a, f;
        if constexpr (c < 4 * NFBF) { fb = c / 4; sj = j; a = c % 4; f = feat0(P, fb) + j; }
        else { fb = NFBF; sj = j & 3; a = 2 * (c - 4 * NFBF) + (j >> 2); f = feat0(P, fb) + (j & 3); }
        (void)sj;
        if constexpr (DUMP) {
          if (!is_dup(P, fb, f)) g.agg_out[(size_t)((t * kWaves + wave) * 16 + li) * g.ld_agg + a * g.F + f] = v[j];
        }
      }
      if constexpr (H2) split_h2_guarded((f4){v[0], v[1], v[2], v[3]}, (f4){v[4], v[5], v[6], v[7]}, false, [&](float* x) __attribute__((always_inline)) {
        int dg = deg;
        asm volatile("" : "+s"(dg));
        if constexpr (c < NC) stats_fast8(c_c, x, dg);
      });
      else split8((f4){v[0], v[1], v[2], v[3]}, (f4){v[4], v[5], v[6], v[7]}, A[0], A[1], A[2]);
      return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int fb, sj, a, f;
      if constexpr (c < 4 * NFBF) { fb = c / 4; sj = j; a = c % 4; f = feat0(P, fb) + j; }
      else { fb = NFBF; sj = j & 3; a = 2 * (c - 4 * NFBF) + (j >> 2); f = feat0(P, fb) + (j & 3); }
      v[j] = stat(fb, sj, a, f, deg);
      if constexpr (DUMP) {
        if (!is_dup(P, fb, f)) g.agg_out[(size_t)((t * kWaves + wave) * 16 + li) * g.ld_agg + a * g.F + f] = v[j];
      }
    }
    const f4 lo4 = (f4){v[0], v[1], v[2], v[3]}, hi4 = (f4){v[4], v[5], v[6], v[7]};
    if constexpr (H2) {
      split_h2_guarded(lo4, hi4, true, [&](float* x) __attribute__((always_inline)) {
        int dg = deg;
        asm volatile("" : "+s"(dg));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          int fb, sj, a, f;
          if constexpr (c < 4 * NFBF) { fb = c / 4; sj = j; a = c % 4; f = feat0(P, fb) + j; }
          else { fb = NFBF; sj = j & 3; a = 2 * (c - 4 * NFBF) + (j >> 2); f = feat0(P, fb) + (j & 3); }
          x[j] = stat(fb, sj, a, f, dg);
        }
      });
    } else {
      if (__builtin_amdgcn_ballot_w64(absmax8(lo4, hi4) == INFINITY) != 0) split8_inf(lo4, hi4, A[0], A[1], A[2]);
      else split8(lo4, hi4, A[0], A[1], A[2]);
    }
  };

  // ---- epilogue: BatchNorm / ReLU / residual, rows scattered to node order through perm -----------------------------------
  const unsigned colc_b = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + (unsigned)(kNBuf * CHV * 16) + (unsigned)lg * 16u;
  // ---- the guard's hand-over (round 6) --------------------------------------------------------------------------------------------
  // Every wavefront whose epilogue finds an output it cannot certify sets the tile's LDS word; the word is read one barrier later -- behind
  // the first barrier of the workgroup's NEXT multiply phase, or behind the loop -- by wavefront 0, which appends the tile to the hand-over
  // list: its position from the device-wide count, then copies of the tile's descriptors, its rows of perm and (tower mode) of row_post.
  const unsigned gflag_b = slot_b + 8u;
  auto collect = [&]() __attribute__((always_inline)) {
    if (wave != 0 || t_done < 0) return;
    unsigned f;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(gflag_b + (unsigned)(gpar ^ 1) * 4u) : "memory");
    if (__builtin_amdgcn_readfirstlane(f) == 0u) return;
    if (lane == 0) asm volatile("ds_write_b32 %0, %1" : : "v"(gflag_b + (unsigned)(gpar ^ 1) * 4u), "v"(0u) : "memory");
    int pos = 0;
    if (lane == 0) pos = __hip_atomic_fetch_add(g.guard, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pos = __builtin_amdgcn_readfirstlane(pos);
    int ln = lane;                                         // (opaque: or hipcc forms the lane addresses below ahead of the tile loop and keeps
    asm volatile("" : "+v"(ln));                           //  them -- spilled -- through every tile for a branch that is almost never taken)
    if (ln < kWaves) g.g_desc[(size_t)pos * kWaves + ln] = g.tdesc[(size_t)t_done * kWaves + ln];
#pragma unroll
    for (int i = 0; i < kWaves / 4; ++i) {
      const size_t o = (size_t)i * 64 + ln;
      g.g_perm[(size_t)pos * (16 * kWaves) + o] = g.perm[(size_t)t_done * (16 * kWaves) + o];
      if (g.row_post != nullptr) g.g_post[(size_t)pos * (16 * kWaves) + o] = g.row_post[(size_t)t_done * (16 * kWaves) + o];
    }
  };
  auto epilogue = [&]() __attribute__((always_inline)) {
    const float lo = g.relu ? 0.f : -INFINITY;
    const bool leaky = g.relu == 2;
    const int row = prow();
    // GUARD: the row's floor-error threshold from its lanes' small chunks (the row's four lanes: li + 16 k, two ds_bpermute like the bound's)
    [[maybe_unused]] float thr_row = 0.f;
    [[maybe_unused]] bool bad = false, thr_any = false;
    if constexpr (H2) {
      // (rows with a floor error are rare -- ~1 % on Gaussian features: the row sums only when the wavefront holds one)
      if (guard_on && __builtin_amdgcn_ballot_w64(fex > 0.f) != 0) {
        float c = fex, o;
        asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(o) : "v"((unsigned)((lane ^ 16) * 4)), "v"(c) : "memory");
        c = add1(c, o);
        asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(o) : "v"((unsigned)((lane ^ 32) * 4)), "v"(c) : "memory");
        c = add1(c, o);
        thr_row = mul1(c, kFloorStatScale);
        thr_any = true;
      }
    }
    if constexpr (TOWER) {
      // the residual rows were requested at the end of step RSTEP; younger than them: the weight copies of the three steps since
      asm volatile("s_waitcnt vmcnt(%5)" : "+v"(res[0]), "+v"(res[1]), "+v"(res[2]), "+v"(res[3]), "+v"(res[4]) : "n"((kAhead - 1) * NI) : "memory");
    } else if constexpr (!RESPF) {
      const char* rbase = reinterpret_cast<const char*>(resb) + (size_t)((unsigned)max(row, 0) * g.ldrb);
#pragma unroll
      for (int n = 0; n < NTA; ++n) res[n] = reinterpret_cast<const f4u*>(rbase + res_col(n))->v;
    }
    char* const yrow = reinterpret_cast<char*>(g.y) + (size_t)(unsigned)max(row, 0) * g.ldyb;
    // pre_add (round 6; production instantiations): the rows a launch adds to its biased accumulator in front of the row factor -- the partial
    // sums of a layer evaluated in several launches.  Kernel-uniform; read here, not prefetched: hipcc waits for the loads with vmcnt(0),
    // i.e. also for the weight copies in flight -- a stall only the launches that carry partial sums pay.  The layer proper (not tower mode)
    // takes its optional row factor here too (tower mode: requested with the node panels).
    [[maybe_unused]] const char* parow = nullptr;
    [[maybe_unused]] float rpe = 1.f;
    if constexpr (!DUMP) {
      if (g.pre_add != nullptr) parow = reinterpret_cast<const char*>(g.pre_add) + (size_t)(unsigned)max(row, 0) * g.ldpb;
      if constexpr (!TOWER) {
        if (g.row_post != nullptr) rpe = g.row_post[(size_t)(t * kWaves + wave) * 16 + li];
      }
    }
#pragma unroll
    for (int n = 0; n < NTA; ++n) {
      // the column constants of the lane's four columns, read through inline asm: an LDS read hipcc can see while a weight copy
      // is in flight makes it drain the copies (vmcnt(0)) first (DESIGN.md 4.2c point 1)
      f4 cb, cs, ct;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(cb) : "v"(colc_b), "n"(n * 64) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(cs) : "v"(colc_b), "n"(NWA * 4 + n * 64) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ct) : "v"(colc_b), "n"(2 * NWA * 4 + n * 64) : "memory");
      [[maybe_unused]] f4 cu;
      if constexpr (H2) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(cu) : "v"(colc_b), "n"(3 * NWA * 4 + n * 64) : "memory");
        // GUARD: an output is certified when |acc| >= the row's share + the column's (pna_x3_split.h); -inf columns (padding, all-zero
        // weights) certify anything; a NaN / Inf accumulator compares false: non-finite rows are not the guard's.  Only for the column
        // tiles that hold a column with a floor error of its own (cmask, kernel-uniform) -- or all of them when the wavefront holds a row with one.
        if (guard_on && (thr_any || ((cmask >> n) & 1u))) {
          f4 cv;
          asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(cv) : "v"(colc_b), "n"(4 * NWA * 4 + n * 64) : "memory");
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float tcv;
            asm("v_add_f32 %0, %1, %2" : "=v"(tcv) : "v"(thr_row), "v"(cv[r]));
            bad = bad || (__builtin_fabsf(acc[n][r]) < tcv);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cb), "+v"(cs), "+v"(ct), "+v"(cu) : : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cb), "+v"(cs), "+v"(ct) : : "memory");
      }
      float z[4];
      [[maybe_unused]] f4 pv = (f4){0.f, 0.f, 0.f, 0.f};
      if constexpr (!DUMP) {
        if (parow != nullptr) pv = fix4(n * 16 + 4 * lg, g.N, reinterpret_cast<const f4u*>(parow + res_col(n))->v);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // (H2: the accumulator is in units of 2^(sA + s_n): back by the row's and the column's powers of two, exact, in the bias' fma)
        float v;
        if constexpr (H2) v = __builtin_fmaf(acc[n][r], runs * cu[r], cb[r]);
        else v = acc[n][r] + cb[r];
        if constexpr (TOWER) v = (v + pv[r]) * rp;
        else if constexpr (!DUMP) v = (v + pv[r]) * rpe;
        v = __builtin_fmaf(v, cs[r], ct[r]);
        z[r] = v < lo ? (leaky ? v * g.slope : 0.f) : v;  // ReLU / LeakyReLU / none (lo = -inf); NaN < lo is false: NaN is kept
      }
      // z[j] = column 16 n + 4 lg + j of row li
      const int c0 = n * 16 + 4 * lg;
      if (has_res) {
        const f4 rr = fix4(c0, g.N, res[n]);
#pragma unroll
        for (int j = 0; j < 4; ++j) z[j] = rr[j] + z[j];
      }
      // (ncw >= N: the columns of a row the caller lets the kernel write -- y_cols_writable; the ones behind N receive zeros, so that a row
      //  of N = 75 floats is ten whole 32-byte sectors instead of nine and a half, and its last window one 16-byte store instead of three scalar ones)
      if (row >= 0 && c0 < g.ncw && !FD_ABL(3)) {
        float* const o = reinterpret_cast<float*>(yrow + (unsigned)c0 * 4u);
        if (c0 + 4 <= g.ncw) {
          f4u w; w.v = (f4){z[0], c0 + 1 < g.N ? z[1] : 0.f, c0 + 2 < g.N ? z[2] : 0.f, c0 + 3 < g.N ? z[3] : 0.f};
          if (c0 >= g.N) w.v.x = 0.f;
          if (FD_ABL(5)) __builtin_nontemporal_store(w.v, reinterpret_cast<f4a4*>(o));
          else *reinterpret_cast<f4u*>(o) = w;
        } else if (c0 < g.N) {                            // the row's last, partial window
          o[0] = z[0];
          if (c0 + 1 < g.N) o[1] = z[1];
          if (c0 + 2 < g.N) o[2] = z[2];
        }
      }
    }
    if constexpr (H2) {
      if (guard_on && __builtin_amdgcn_ballot_w64(bad && row >= 0) != 0 && lane == 0)
        asm volatile("ds_or_b32 %0, %1" : : "v"(gflag_b + (unsigned)gpar * 4u), "v"(1u) : "memory");
    }
  };

  // ---- the (tile, chunk) pipeline: step k reads LDS buffer k % 5; barrier B_k sits in the middle of step k; after B_k every
  //      wavefront has finished step k-1, so buffer (k-1) % 5 = (k+4) % 5 is free: step k+4's image is copied then.  Before B_k a
  //      wavefront waits for ITS pieces of step k+1's image with a counted vmcnt that leaves the two younger images (steps k+2,
  //      k+3) in flight: a copy has three steps to land.  (Round 3, first version: 3 buffers and vmcnt(0) -- one step of cover
  //      against ~2 us of copy latency under the gather's load: the phase timers showed 45 % of a wavefront's time in the
  //      multiply phase, 5x its matrix-pipe time.)  The same waits retire the residual rows requested at the end of the gather:
  //      they are older than every copy issued during the steps. ------------------------------------------------------------------
  unsigned long long tg = 0, tm = 0, te = 0, t00 = now();
  unsigned long long tvm = 0, tb0 = 0, tbar = 0;           // experiments build: inside the multiply phase -- copy waits | first barrier of a pass | other barriers
  long ib_cur = (long)td_cur.z * g.img_stride, ib_next = (long)td_nxt.z * g.img_stride;
  int buf = 0;
#pragma unroll
  for (int c = 0; c < kAhead; ++c) stage(c, c, ib_cur);   // (every shape has at least kAhead steps)
  // the ids of the first tile's edges 0..3 (later tiles: fetched by the previous tile's last packets)
#pragma unroll
  for (int j = 0; j < kRing; ++j) ld4_ws(idr[j], g.ids, (unsigned)td_cur.x * 64u + (unsigned)j * 64u + lib);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(idr[0]), "+v"(idr[1]), "+v"(idr[2]), "+v"(idr[3]) : : "memory");
  gather(t, std::integral_constant<int, 0>{});
  frag(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  tg += now() - t00;

  // step s of the tile (s_c), in gather pass P: chunk c = (s - P NSP) / NPAN of the pass against column panel (s - P NSP) % NPAN
  auto step = [&](auto s_c, auto p_c) __attribute__((always_inline)) {
    constexpr int sg = decltype(s_c)::value, P = decltype(p_c)::value;
    constexpr int sp = sg - P * NSP;                      // step within the pass (tower mode: the panel chunks follow, P = 0)
    constexpr int c = sp / NPAN, pan = sp % NPAN, a0 = pan * NT;
    const unsigned ba0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + (unsigned)(buf * CHV + lg * NWP + li) * 16u;
    const int buf2 = buf == 0 ? kNBuf - 1 : buf - 1;     // (k + kAhead) % kNBuf
    frag_t B[2][NTERM];                                  // B fragments of column tile n (slot n & 1): one ds_read_b128 per term
    // partial products, smallest first: (statistics' term TA, weights' term TB)
    constexpr int TA[6] = {H2 ? 1 : 2, H2 ? 0 : 1, 0, 1, 0, 0}, TB[6] = {0, 1, H2 ? 0 : 2, 0, 1, 0};
    auto mma = [](const frag_t& w_, const frag_t& a_, f4 c_) __attribute__((always_inline)) -> f4 {
      if constexpr (H2) return __builtin_amdgcn_mfma_f32_16x16x32_f16(w_, a_, c_, 0, 0, 0);
      else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_, a_, c_, 0, 0, 0);
    };
    if constexpr (sg == 0) {
#pragma unroll
      for (int n = 0; n < NTA; ++n) acc[n] = (f4){0.f, 0.f, 0.f, 0.f};
    }
    // Column tiles in PAIRS (0,1) (2,3) (4): both tiles' B fragments are read (6 x ds_read_b128, ONE wait), then their 12 MFMAs are
    // issued alternating between the two accumulators -- every accumulator sees an MFMA every other issue slot instead of six
    // dependent ones back to back (round 3, first version: tile by tile, lgkmcnt(0) before every tile's six dependent MFMAs -- the
    // phase timers put 44 % of a wavefront's time into the multiply phase, 2400 cycles per step against 480 of matrix-pipe time).
    // The next pair's reads go into the registers the MFMAs just issued have read (the pattern of pna_posttrans_x3.hip).  One
    // address register per step, the (term, column tile) displacement in the instruction's offset field: as "v" operands the
    // 45 distinct addresses were hoisted out of the tile loop and spilled.
#define FD_READ_B(slot, n)                                                                                                                  \
    _Pragma("unroll") for (int tm_ = 0; tm_ < NTERM; ++tm_)                                                                                 \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(B[slot][tm_]) : "v"(ba0), "n"(tm_ * 4 * NWP * 16 + (n) * 256) : "memory")
#ifdef FD_SCHED_FENCE
#define FD_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define FD_FENCE()
#endif
#define FD_WAIT_B()                                                                                                                         \
    if constexpr (H2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(B[0][0]), "+v"(B[0][1]), "+v"(B[1][0]), "+v"(B[1][1]) : : "memory");          \
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(B[0][0]), "+v"(B[0][1]), "+v"(B[0][NTERM - 1]), "+v"(B[1][0]), "+v"(B[1][1]), "+v"(B[1][NTERM - 1]) : : "memory"); \
    FD_FENCE()
    if (!FD_ABL(4)) { FD_READ_B(0, 0); FD_READ_B(1, 1); }
    FD_WAIT_B();
    if (!FD_ABL(0))
#pragma unroll
    for (int pp = 0; pp < NPROD; ++pp) {
      acc[a0] = mma(B[0][TB[pp]], A[TA[pp]], acc[a0]);
      acc[a0 + 1] = mma(B[1][TB[pp]], A[TA[pp]], acc[a0 + 1]);
    }
    FD_FENCE();
    if (!FD_ABL(4)) { FD_READ_B(0, 2); FD_READ_B(1, 3); }
    FD_WAIT_B();
    // (the images of a tile's first kAhead - 1 steps were requested before its gather, whose waits retired them; waiting here
    // would only wait for the residual rows requested at the end of the gather -- a full memory round trip per tile)
    // tower mode: the panel strips (requested at the end of step PSTEP) may stay in flight for three steps, the residual rows
    // (end of step RSTEP) into the epilogue: they are younger than the image this wait is for
    // (wide shapes: the same holds for the first steps behind the second gather)
    constexpr int young = !TOWER ? 0 : (c > PSTEP && c < PWAIT) ? NPL : c > RSTEP ? NTA : 0;
#ifdef FD_FINE_TIMERS
    {   // (-DPNA_AMD_EXPERIMENTS -DFD_FINE_TIMERS) the same wait and barrier, timed apart: copy wait | barrier wait (the first step of a pass: the other wavefronts' gathers)
      const unsigned long long w0 = now();
      if constexpr (sp >= kAhead - 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"((kAhead - 2) * NI + young) : "memory");
      const unsigned long long w1 = now();
      asm volatile("s_barrier" ::: "memory");
      const unsigned long long w2 = now();
      tvm += w1 - w0;
      if constexpr (sp == 0) tb0 += w2 - w1; else tbar += w2 - w1;
    }
#else
    if constexpr (sp >= kAhead - 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" : : "n"((kAhead - 2) * NI + young) : "memory");
    else asm volatile("s_barrier" ::: "memory");
#endif
    if constexpr (H2 && sg == 0) {                        // (every wavefront's epilogue of the previous tile lies behind this barrier)
      if (guard_on) collect();
    }
    if constexpr (TOWER && c == PWAIT) {                  // (the wait above left only the two youngest images in flight)
      static_assert(NL == 4 || NL == 5, "tower shapes");
      asm volatile("" : "+v"(rp));
      if constexpr (NL == 4)
        asm volatile("" : "+v"(pk[0][0]), "+v"(pk[0][1]), "+v"(pk[0][2]), "+v"(pk[0][3]), "+v"(pk[1][0]), "+v"(pk[1][1]), "+v"(pk[1][2]), "+v"(pk[1][3]));
      else
        asm volatile("" : "+v"(pk[0][0]), "+v"(pk[0][1]), "+v"(pk[0][2]), "+v"(pk[0][3]), "+v"(pk[0][NL - 1]), "+v"(pk[1][0]), "+v"(pk[1][1]), "+v"(pk[1][2]),
                     "+v"(pk[1][3]), "+v"(pk[1][NL - 1]));
    }
    if (sg + kAhead < NCT) stage(sg + kAhead, buf2, ib_cur);
    else stage(sg + kAhead - NCT, buf2, ib_next);
    if (!FD_ABL(0))
#pragma unroll
    for (int pp = 0; pp < NPROD; ++pp) {
      acc[a0 + 2] = mma(B[0][TB[pp]], A[TA[pp]], acc[a0 + 2]);
      acc[a0 + 3] = mma(B[1][TB[pp]], A[TA[pp]], acc[a0 + 3]);
    }
    FD_FENCE();
    if constexpr (NT == 5) {
      if (!FD_ABL(4)) { FD_READ_B(0, 4); }
      FD_WAIT_B();
      if (!FD_ABL(0))
#pragma unroll
      for (int pp = 0; pp < NPROD; ++pp) acc[a0 + 4] = mma(B[0][TB[pp]], A[TA[pp]], acc[a0 + 4]);
      FD_FENCE();
    }
#undef FD_READ_B
#undef FD_WAIT_B
#undef FD_FENCE
    buf = buf == kNBuf - 1 ? 0 : buf + 1;
    if constexpr (TOWER && H2 && c == PWAIT) panel_rescale();
    // the next chunk's fragment, once the chunk's last panel is issued (tower mode: the panel chunks continue the numbering)
    if constexpr (pan == NPAN - 1 && c + 1 < NC + NPC) frag(std::integral_constant<int, c + 1>{}, p_c);
    if constexpr (TOWER && c == PSTEP) issue_panels();
    if constexpr (TOWER && c == RSTEP) {
      const unsigned rrow = (unsigned)max(prow(), 0) * g.ldrb;
#pragma unroll
      for (int n = 0; n < NTA; ++n) ld16_ws(res[n], resb, rrow + res_col(n));
    }
  };
  // the steps [P (NSP + NPC) + b, .. + 4) of gather pass P
  auto steps4 = [&](auto b_c, auto p_c) __attribute__((always_inline)) {
    constexpr int b = decltype(b_c)::value, P = decltype(p_c)::value, N1 = NSP + NPC, o = P * NSP;
    if constexpr (b < N1) step(std::integral_constant<int, (b < N1) ? o + b : 0>{}, p_c);
    if constexpr (b + 1 < N1) step(std::integral_constant<int, (b + 1 < N1) ? o + b + 1 : 0>{}, p_c);
    if constexpr (b + 2 < N1) step(std::integral_constant<int, (b + 2 < N1) ? o + b + 2 : 0>{}, p_c);
    if constexpr (b + 3 < N1) step(std::integral_constant<int, (b + 3 < N1) ? o + b + 3 : 0>{}, p_c);
  };
  auto pass_steps = [&](auto p_c) __attribute__((always_inline)) {
    steps4(std::integral_constant<int, 0>{}, p_c);
    steps4(std::integral_constant<int, 4>{}, p_c);
    steps4(std::integral_constant<int, 8>{}, p_c);
    steps4(std::integral_constant<int, 12>{}, p_c);
    steps4(std::integral_constant<int, 16>{}, p_c);
    static_assert(NSP + NPC <= 20, "steps4 calls");
  };
  while (true) {
    const unsigned long long t0 = now();
    // The multiplying wavefront goes ahead of the gathering one on its SIMD (the other workgroup's): its phase is issue-bound
    // (~2500 instructions per tile at one per ~4 cycles) and while it lasts none of its loads are in flight, whereas the gathering
    // wavefront mostly waits for memory and its fold hides behind that (skipping the fold entirely changes nothing).  Measured in
    // the experiments build on the C3 layer: 0.873 -> 0.790 ms.
    if (!FD_ABL(6)) __builtin_amdgcn_s_setprio(1);
    if (dyn && wave == 0 && lane == 0) {                  // claim the tile three after the next one (result harvested in the next gather's drain)
      const int one = 1;
      asm volatile("v_mov_b32 %0, -1\n\ts_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=&v"(claimed) : "v"(0u), "v"(one), "s"(g.counter) : "memory");
    }
    pass_steps(std::integral_constant<int, 0>{});
    if constexpr (GP == 2) {                              // the second half of the features: gather again, multiply on
      if (!FD_ABL(6)) __builtin_amdgcn_s_setprio(0);
      gather(t, std::integral_constant<int, GP - 1>{});
      frag(std::integral_constant<int, 0>{}, std::integral_constant<int, GP - 1>{});
      if (!FD_ABL(6)) __builtin_amdgcn_s_setprio(1);
      pass_steps(std::integral_constant<int, GP - 1>{});
    }
    // the residual rows landed by the last step's counted wait (they are older than all but the first copies of the tile, and
    // every shape has at least 4 steps); from here on the compiler may read them
    if constexpr (RESPF) {
      asm volatile("" : "+v"(res[0]), "+v"(res[1]), "+v"(res[2]), "+v"(res[3]), "+v"(res[4]));
      if constexpr (NTA == 8) asm volatile("" : "+v"(res[5]), "+v"(res[6]), "+v"(res[NTA - 1]));
    }
    const unsigned long long t1 = now();
    epilogue();
    if (!FD_ABL(6)) __builtin_amdgcn_s_setprio(0);
    const unsigned long long t2 = now();
    tm += t1 - t0; te += t2 - t1;
    {
      int t_n3 = t_n2 + G;
      if (dyn) {                                          // the word wavefront 0 wrote in this tile's first gather: read behind the multiply phase's barriers
        int v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(slot_b + (unsigned)par * 4u) : "memory");
        t_n3 = __builtin_amdgcn_readfirstlane(v);
        par ^= 1;
      }
      t_done = t; gpar ^= 1;
      t = t_n1; t_n1 = t_n2; t_n2 = t_n3;
    }
    if (t >= ntiles) break;                               // (wave-uniform)
    td_cur = td_nxt; td_nxt = td_n2;
    ib_cur = ib_next;
    ib_next = (long)td_nxt.z * g.img_stride;
    gather(t, std::integral_constant<int, 0>{});
    frag(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    tg += now() - t2;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the copies issued for steps that do not exist
  if constexpr (H2) {
    if (guard_on) {                                        // the last tile's hand-over word: every wavefront's epilogue has written by the barrier
      __syncthreads();
      collect();
    }
  }
  finish();
#ifdef PNA_AMD_EXPERIMENTS
  if (g.dbg && lane == 0) {
    unsigned long long* d = g.dbg + ((size_t)blockIdx.x * kWaves + wave) * 8;
    d[0] = tg; d[1] = tm; d[2] = te; d[3] = now() - t00; d[4] = tvm; d[5] = tb0; d[6] = tbar; d[7] = 0;
  }
#endif
}

// The launches (round 6).  ARITH 0 / 2: the layer in fp16 x 2 (2: the verification instantiations) -- GUARDED when the call carries a guard
// workspace: a tile whose outputs the floor-error bound does not certify is appended to the device-wide hand-over list.  ARITH 1: the layer
// in bf16 x 3 -- over its own tile tables, or (g.list_count set) as the CONSUMING launch behind a guarded one: the tile tables are the list's
// copies, the tile count lives on the device, a workgroup that finds nothing to do leaves at once (one atomic load), and the launch zeroes
// the count behind itself.  Measured alternatives (profiles/r06_arith_ab*.log): both phases in ONE launch behind a grid barrier -- the
// combined kernel's hot loop ran 1.9 % slower (register allocation over two bodies) and 472 workgroups polling one word slowed the last
// gathers by another 6 %; a persistent kernel's grid barrier is also only deadlock-free while every workgroup is resident, which two such
// launches on one device (two streams, two ranks) do not guarantee.
template <int NFBF, bool HALF, bool DUMP, bool TOWER, int ARITH, int GP, int NPAN>
__global__ __launch_bounds__(64 * waves_for(GP, NPAN), (DUMP || waves_for(GP, NPAN) == 8) ? 1 : 2) void k_fused_degree(const FDArgs g) {
  const int n1 = g.M / (waves_for(GP, NPAN) * 16);
  if constexpr (ARITH == 1) {
    int* const gw = g.list_count;                          // {tiles handed over | workgroups that have left | handed over since cleared | guarded calls}
    int n = n1;
    if (gw != nullptr) {
      n = __builtin_amdgcn_readfirstlane(__hip_atomic_load(gw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if (n == 0) {                                        // (almost always: nothing was handed over -- nobody has anything to do or to reset)
        if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(gw + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
    }
    if ((int)blockIdx.x < n) fd_body<NFBF, HALF, DUMP, TOWER, false, !DUMP && !TOWER, GP, NPAN>(g, (int)blockIdx.x, (int)gridDim.x, n);
    // the last workgroup to leave zeroes the count for the next call (every workgroup has read it by then) and keeps the statistics
    if (gw != nullptr && threadIdx.x == 0 && __hip_atomic_fetch_add(gw + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
      __hip_atomic_fetch_add(gw + 2, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(gw + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(gw, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(gw + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    fd_body<NFBF, HALF, DUMP, TOWER, true, !DUMP && !TOWER, GP, NPAN>(g, (int)blockIdx.x, (int)gridDim.x, n1);
  }
}

// ---- weight images: W_D = sum_s scale[i][s] W_s in fp32 (scaler order), K reordered into the kernel's chunks, cut into three
//      bf16 terms, laid out as the LDS image of every chunk: [chunk][term][lane group][80 cols][8 k] ---------------------------
//      Tower images (tower != 0): scaler blocks of K = 5 F columns [4 F aggregators | F self panel (block 0 only)], followed by
//      the chunks of the two node panels: x_dst against W_D,mean + W_D,max + W_D,min, h against the self panel.
//      Wide shapes (npan = 2): [chunk][panel][term][lane group][64 cols][8 k], panel p = output columns 64 p .. 64 p + 64.
//      fp16 x 2 images (round 5): TWO fp16 terms of W_D[n][.] * 2^s_n, s_n the power of two that puts the largest |W_D[n][k]| over
//      k and over ALL images into [2^13, 2^14) (colmax: the maxima's bit patterns, k_fused_colmax; pna_x3_split.h); images `stride`
//      elements apart, each followed by the tail of the columns' 2^-s_n and of the guard's per-column thresholds (k_fused_wsmall, k_fused_tails).
__device__ __forceinline__ float colmax_bound(unsigned bits) {   // (NaN / Inf / 0 maxima: any finite positive bound will do)
  return __builtin_fminf(__builtin_fmaxf(__builtin_bit_cast(float, bits), 1e-30f), 3.4028234663852886e38f);
}
// W_D[n][col] = sum_s scale_s(D) W_s[n][col], scaler order (the pack kernel's, the maxima's and the small-weight count's: op for op)
__device__ __forceinline__ float fd_combined(const float* w_ref, long ldw, int n, int col, int im, int S, int K, const float* scale) {
  const float* row = w_ref + (long)n * ldw + col;
  float w = scale ? scale[(long)im * S] * row[0] : row[0];
  for (int s = 1; s < S; ++s) w = w + (scale ? scale[(long)im * S + s] * row[(long)s * K] : row[(long)s * K]);
  return w;
}
// weight `col` of output column n of image im as the images hold it: col < 5 F a combined weight, beyond the x_dst panel's sum (tower)
__device__ __forceinline__ float fd_image_weight(const float* w_ref, long ldw, int n, int col, int im, int S, int F, int tower, const float* scale) {
  const int K = tower ? 5 * F : 4 * F;
  if (col < 5 * F) return fd_combined(w_ref, ldw, n, col, im, S, K, scale);
  const int f = col - 5 * F;
  return (fd_combined(w_ref, ldw, n, f, im, S, K, scale) + fd_combined(w_ref, ldw, n, F + f, im, S, K, scale)) + fd_combined(w_ref, ldw, n, 2 * F + f, im, S, K, scale);
}
// One block per (image, column): the block's maximum first, ONE atomic per block (round 6: an atomic per weight -- 1.4 M of them onto 75
// addresses for the benchmark layer's 63 images -- took 5.6 ms of the 7.3 ms image pack).
__global__ void k_fused_colmax(const float* w_ref, long ldw, int N, int F, int S, const float* scale, int n_img, unsigned* colmax, int tower) {
  const int KC = tower ? 6 * F : 4 * F;                   // weights the images hold per output column: + the x_dst panel's sums (tower)
  const int n = blockIdx.x % N, im = blockIdx.x / N;
  __shared__ unsigned part[128];
  unsigned m = 0u;
  for (int col = threadIdx.x; col < KC; col += blockDim.x) {
    const unsigned b = __builtin_bit_cast(unsigned, __builtin_fabsf(fd_image_weight(w_ref, ldw, n, col, im, S, F, tower, scale)));
    m = b > m ? b : m;                                     // (|w| as bits: ordered like the floats; NaN above all)
  }
  part[threadIdx.x] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)blockDim.x; ++i) m = part[i] > m ? part[i] : m;
    if (m != 0u) atomicMax(colmax + n, m);
  }
}
// GUARD (round 6): wex[n] = the largest, over the images, floor error of output column n's weights -- the sum over k of the part of the
// two-term split's error of W_D[n][k] 2^s_n that is absolute (a non-zero weight below 2^-3 in the column's units) instead of relative
// (pna_x3_split.h::h2_floor_error), as bits (non-negative floats order like their bit patterns).  One block per (image, column).
__global__ void k_fused_wsmall(const float* w_ref, long ldw, int N, int F, int S, const float* scale, const unsigned* colmax, unsigned* wex, int tower) {
  const int KC = tower ? 6 * F : 4 * F;
  const int n = blockIdx.x % N, im = blockIdx.x / N;
  const float sc = __builtin_ldexpf(1.0f, h2_scale_exp(colmax_bound(colmax[n])));
  __shared__ float part[128];
  float c = 0.f;
  for (int col = threadIdx.x; col < KC; col += blockDim.x) c += h2_floor_error(fd_image_weight(w_ref, ldw, n, col, im, S, F, tower, scale) * sc);
  part[threadIdx.x] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < (int)blockDim.x; ++i) t += part[i];            // (fixed order: the same bits every time)
    if (t > 0.f) atomicMax(wex + n, __builtin_bit_cast(unsigned, t));
  }
}
__global__ void k_fused_tails(unsigned char* img, long payload, long stride_bytes, int n_img, int N) {
  const int n = threadIdx.x;                              // 128 threads
  const unsigned bits = reinterpret_cast<const unsigned*>(img + payload)[n];
  const float wex = __builtin_bit_cast(float, reinterpret_cast<const unsigned*>(img + payload)[128 + n]);
  const float u = n < N ? __builtin_ldexpf(1.0f, -h2_scale_exp(colmax_bound(bits))) : 1.0f;
  // (a column without a non-zero weight, and the padding: -inf -- an exact zero certifies itself)
  const float cv = (n < N && bits != 0u) ? wex * kFloorWeightScale : -INFINITY;
  __syncthreads();
  for (int im = 0; im < n_img; ++im) {
    reinterpret_cast<float*>(img + (long)im * stride_bytes + payload)[n] = u;
    reinterpret_cast<float*>(img + (long)im * stride_bytes + payload)[128 + n] = cv;
  }
}
__global__ void k_pack_fused_degree(const float* w_ref, long ldw, int N, int F, int S, const float* scale, int n_img, unsigned short* img, int tower,
                                    int nwp, int npan, int nterm, const unsigned* colmax, long stride) {
  const int nfull = shape_full(F), NCS = shape_chunks(F);
  const int NC = NCS + (tower ? 2 * nfull + (shape_half(F) ? 1 : 0) : 0);
  const long per = (long)NC * npan * nterm * 4 * nwp * 8;
  const long total = per * n_img;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i;
    const int e = r % 8; r /= 8;
    const int nn = r % nwp; r /= nwp;
    const int lgp = r % 4; r /= 4;
    const int term = r % nterm; r /= nterm;
    const int pan = r % npan; r /= npan;
    const int c = r % NC; r /= NC;
    const int im = (int)r;
    const int n = pan * nwp + nn;                         // output column
    int a, f, fb, sj;                                     // a: aggregator 0..3; 4: the self panel; 5: the x_dst panel; fb: feature block
    if (c < 4 * nfull) { a = c % 4; fb = c / 4; sj = e; }
    else if (c < NCS) { a = 2 * (c - 4 * nfull) + (e >> 2); fb = nfull; sj = e & 3; }
    else if (c - NCS < 2 * nfull) { a = (c - NCS) / nfull == 0 ? 5 : 4; fb = (c - NCS) % nfull; sj = e; }
    else { a = (e >> 2) == 0 ? 5 : 4; fb = nfull; sj = e & 3; }
    // the lane group's window of block fb: its nominal start, slid back in the row's LAST block to end at F; a slot below the
    // nominal start repeats a feature a lower lane group multiplies: weight 0 (the kernel's feat0 / is_dup)
    const bool halfb = fb == nfull, lastb = fb == nfull + (shape_half(F) ? 1 : 0) - 1;
    const int wl = halfb ? 4 : 8, nom = fb * 32 + lgp * wl, st = lastb ? (nom < F - wl ? nom : F - wl) : nom;
    f = st + sj;
    const bool dup = f < nom;
    float w = 0.f;
    if (n < N && f < F && !dup) w = fd_image_weight(w_ref, ldw, n, a * F + f, im, S, F, tower, scale);     // (a = 5: column 5 F + f, the x_dst panel's sum)
    if (nterm == 3) img[(long)im * stride + i % per] = weight_term(w, term);
    else img[(long)im * stride + i % per] = weight_term_h2(w * __builtin_ldexpf(1.0f, h2_scale_exp(colmax_bound(colmax[n < N ? n : 0]))), term);
  }
}

// wide shapes (4 full feature blocks in two gather passes, 81..128 output columns in two panels of 64)
__host__ __device__ constexpr bool shape_wide_f(int F) { return F > 96 && F <= 128 && shape_full(F) == 4 && !shape_half(F); }
__host__ __device__ constexpr bool shape_wide_n(int N) { return N > kNW && N <= 128; }

// ARITH: 0 fp16 x 2, guarded when g.guard is set | 1 bf16 x 3 (g.list_count: the consuming launch) | 2 fp16 x 2, the verification instantiations (DUMP)
template <int NFBF, bool HALF, bool DUMP, bool TOWER, int GP, int NPAN, int ARITH>
int launch(const FDArgs& g, int wgs, hipStream_t st) {
  constexpr int NWP = NPAN == 1 ? kNW : 64, NC = 4 * NFBF + (HALF ? 2 : 0), WAVES = waves_for(GP, NPAN);
  constexpr size_t lds_h2 = (size_t)buffers_for(GP, NPAN, NC, TOWER, true) * (2 * 4 * NWP) * 16 + (size_t)(5 * NWP * NPAN) * sizeof(float) + 16;
  constexpr size_t lds_x3 = (size_t)buffers_for(GP, NPAN, NC, TOWER, false) * (3 * 4 * NWP) * 16 + (size_t)(3 * NWP * NPAN) * sizeof(float) + 16;
  // (+ 16: the hand-over words of the dynamic schedule and of the guard)
  const size_t lds = ARITH == 1 ? lds_x3 : lds_h2;
  auto* fn = k_fused_degree<NFBF, HALF, DUMP, TOWER, ARITH, GP, NPAN>;
  if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
  if (WAVES == 8) wgs = (wgs + 1) / 2;                    // (the caller counts 4-wavefront workgroups, two per CU)
  hipLaunchKernelGGL(fn, dim3((unsigned)wgs), dim3(64 * WAVES), lds, st, g);
  return 0;
}
template <bool DUMP, int ARITH>
int launch_shape(const FDArgs& g, int wgs, hipStream_t st) {
  const int nf = shape_full(g.F);
  const bool half = shape_half(g.F);
  if (shape_wide_f(g.F) || shape_wide_n(g.N)) {
    if (g.xd) return -2;
    if (shape_wide_f(g.F)) return shape_wide_n(g.N) ? launch<2, false, DUMP, false, 2, 2, ARITH>(g, wgs, st) : launch<2, false, DUMP, false, 2, 1, ARITH>(g, wgs, st);
    if (nf == 2 && !half) return launch<2, false, DUMP, false, 1, 2, ARITH>(g, wgs, st);
    return -2;
  }
  if constexpr (!DUMP) {
    if (g.xd) return nf != 2 ? -2 : half ? launch<2, true, false, true, 1, 1, ARITH>(g, wgs, st) : launch<2, false, false, true, 1, 1, ARITH>(g, wgs, st);
  }
  if (nf == 1 && !half) return launch<1, false, DUMP, false, 1, 1, ARITH>(g, wgs, st);
  if (nf == 1 && half) return launch<1, true, DUMP, false, 1, 1, ARITH>(g, wgs, st);
  if (nf == 2 && !half) return launch<2, false, DUMP, false, 1, 1, ARITH>(g, wgs, st);
  if (nf == 2 && half) return launch<2, true, DUMP, false, 1, 1, ARITH>(g, wgs, st);
  return -2;
}

bool shape_ok(int F, int N) {
  // F: 17..80 (one gather pass) or 97..128 (two passes of two full blocks; 81..96 would need unequal passes: not built);
  // N: 4..80 (one panel of 80 columns) or 81..128 (two panels of 64), the latter with exactly two full feature blocks per pass
  // (49 <= F <= 64 or 113..128: with a half block on top, 80 statistics + the ring + 32 accumulators + the residual spill)
  const bool f_ok = (F >= 17 && F <= 80) || shape_wide_f(F), n_ok = (N >= 4 && N <= kNW) || shape_wide_n(N);
  return f_ok && n_ok && !(shape_wide_n(N) && !shape_wide_f(F) && !(shape_full(F) == 2 && !shape_half(F)));
}
bool tower_shape_ok(int F, int N) { return F <= 80 && N <= kNW && shape_ok(F, N) && shape_full(F) == 2; }
// bytes of one image (stride between images): x3 = 0 two fp16 terms + the tail, 1 three bf16 terms
int64_t image_bytes(int F, int N, int tower, int x3) {
  if (tower ? !tower_shape_ok(F, N) : !shape_ok(F, N)) return 0;
  const int chunks = shape_chunks(F) + (tower ? 2 * shape_full(F) + (shape_half(F) ? 1 : 0) : 0);
  const int nterm = x3 ? 3 : 2;
  const int64_t step = shape_wide_n(N) ? 2 * (nterm * 4 * 64) : nterm * 4 * kNW;
  return (int64_t)chunks * step * 16 + (x3 ? 0 : kTailBytes);
}

}  // namespace

extern "C" int32_t pna_fused_degree_tile_rows(int32_t F, int32_t N) {
  if (!shape_ok(F, N)) return 0;
  return 16 * waves_for(shape_wide_f(F) ? 2 : 1, shape_wide_n(N) ? 2 : 1);
}

extern "C" int64_t pna_fused_image_bytes(int32_t F, int32_t N, int32_t tower, int32_t x3) { return image_bytes(F, N, tower != 0, x3 != 0); }
extern "C" int64_t pna_fused_degree_image_bytes(int32_t F, int32_t N) { return image_bytes(F, N, 0, 0); }
extern "C" int64_t pna_fused_tower_image_bytes(int32_t F, int32_t N) { return image_bytes(F, N, 1, 0); }

extern "C" int64_t pna_fused_degree_guard_bytes(int64_t M) {
  // {count | finished | handed over since cleared | consuming launches | pad to 64 bytes} + descriptors (16 bytes per 16 rows) + perm + row_post
  return M < 0 ? 0 : 64 + M + 4 * M + 4 * M;
}

// column maxima -> small-weight counts -> fp16 x 2 images -> tails (stream-ordered; the first image's tail is the scratch of both until k_fused_tails)
static int pack_h2(const char* who, const float* w_ref, int64_t ldw, int N, int F, int S, const float* scale, int n_img, void* img, int64_t stride,
                   int tower, int nwp, int npan, hipStream_t st) {
  const int64_t payload = stride - kTailBytes;
  const int64_t elems = payload / 2 * n_img;
  const int blocks = (int)((elems + 255) / 256 > 8192 ? 8192 : (elems + 255) / 256);
  unsigned* const colmax = reinterpret_cast<unsigned*>((unsigned char*)img + payload);
  if (hipMemsetAsync(colmax, 0, kTailBytes, st) != hipSuccess) return pna_set_error(PNA_E_LAUNCH, who);
  hipLaunchKernelGGL(k_fused_colmax, dim3((unsigned)(n_img * N)), dim3(128), 0, st, w_ref, (long)ldw, N, F, S, scale, n_img, colmax, tower);
  hipLaunchKernelGGL(k_fused_wsmall, dim3((unsigned)(n_img * N)), dim3(128), 0, st, w_ref, (long)ldw, N, F, S, scale, (const unsigned*)colmax, colmax + 128, tower);
  hipLaunchKernelGGL(k_pack_fused_degree, dim3(blocks), dim3(256), 0, st, w_ref, (long)ldw, N, F, S, scale, n_img,
                     (unsigned short*)img, tower, nwp, npan, 2, (const unsigned*)colmax, (long)(stride / 2));
  hipLaunchKernelGGL(k_fused_tails, dim3(1), dim3(128), 0, st, (unsigned char*)img, (long)payload, (long)stride, n_img, N);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}

extern "C" int pna_fused_pack_f32(const float* w_ref, int64_t ldw, int32_t N, int32_t F, int32_t n_scaler, const float* scale,
                                  int32_t n_img, void* img, int32_t tower, int32_t x3, pna_stream_t stream) {
  const int64_t stride = image_bytes(F, N, tower != 0, x3 != 0);
  if (!w_ref || !img || n_img < 1 || n_scaler < 1 || n_scaler > PNA_MAX_SCALER || stride == 0 ||
      ldw < (int64_t)n_scaler * (tower ? 5 : 4) * F || (n_scaler > 1 && !scale))
    return pna_set_error(PNA_E_INVALID, "pna_fused_pack_f32: bad arguments (pna_fused_image_bytes(F, N, tower, x3) > 0, ldw >= n_scaler * (tower ? 5 : 4) F, scale required for n_scaler > 1)");
  const bool wide = shape_wide_n(N);
  if (!x3) return pack_h2("pna_fused_pack_f32: hipMemsetAsync failed", w_ref, ldw, N, F, n_scaler, scale, n_img, img, stride, tower ? 1 : 0, wide ? 64 : kNW,
                          wide ? 2 : 1, (hipStream_t)stream);
  const int64_t elems = stride / 2 * n_img;
  const int blocks = (int)((elems + 255) / 256 > 8192 ? 8192 : (elems + 255) / 256);
  hipLaunchKernelGGL(k_pack_fused_degree, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_ref, (long)ldw, N, F, n_scaler, scale, n_img,
                     (unsigned short*)img, tower ? 1 : 0, wide ? 64 : kNW, wide ? 2 : 1, 3, (const unsigned*)nullptr, (long)(stride / 2));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}
extern "C" int pna_fused_tower_pack_f32(const float* w_ref, int64_t ldw, int32_t N, int32_t F, int32_t n_scaler, const float* scale,
                                        int32_t n_img, void* img, pna_stream_t stream) {
  return pna_fused_pack_f32(w_ref, ldw, N, F, n_scaler, scale, n_img, img, 1, 0, stream);
}
extern "C" int pna_fused_degree_pack_f32(const float* w_ref, int64_t ldw, int32_t N, int32_t F, int32_t n_scaler, const float* scale,
                                         int32_t n_img, void* img, pna_stream_t stream) {
  return pna_fused_pack_f32(w_ref, ldw, N, F, n_scaler, scale, n_img, img, 0, 0, stream);
}

extern "C" int pna_fused_degree_f32(const pna_fused_degree_args* p, pna_stream_t stream) {
  if (!p) return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: null args");
  if (int rc_ss = pna_check_struct_size("pna_fused_degree_f32", p->struct_size, sizeof(*p))) return rc_ss;
  if (p->M == 0) return PNA_OK;
  const int arith = p->arith;
  if (arith != PNA_FD_ARITH_GUARDED && arith != PNA_FD_ARITH_X3 && arith != PNA_FD_ARITH_H2)
    return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: arith must be PNA_FD_ARITH_GUARDED, PNA_FD_ARITH_X3 or PNA_FD_ARITH_H2");
  const bool need_h2 = arith != PNA_FD_ARITH_X3, need_x3 = arith != PNA_FD_ARITH_H2;
  if (!p->tile_desc || !p->tile_ids || !p->x || !p->row_perm || !p->y || (need_h2 && !p->w_img) || (need_x3 && !p->w_img_x3))
    return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: tile_desc / tile_ids / x / row_perm / y and the images of the arithmetic (w_img: fp16 x 2, w_img_x3: bf16 x 3) must be non-null");
  if (!shape_ok(p->F, p->N))
    return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: F in 17..80 or 97..128, N in 4..128 (N > 80 needs F in 49..64 or 97..128)");
  // (round 4: the source rows are read through 64-bit lane addresses -- any 4-byte aligned pitch >= F, no 4 GiB / 2^24-row limit; the
  // strips of a row's last block are 16-byte reads that end at the row's F-th float: no read leaves a row)
  if (p->ldx < p->F || ((uintptr_t)p->x & 3) != 0 || (int64_t)p->ldx * 4 >= (1ll << 31))
    return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: x must be 4-byte aligned with a row pitch >= F (< 2^29 floats)");
  if (p->x_rows < 1 || p->x_rows >= (1ll << 32))
    return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: the source table must have 1 <= x_rows < 2^32");
  const int tile_rows = 16 * waves_for(shape_wide_f(p->F) ? 2 : 1, shape_wide_n(p->N) ? 2 : 1);
  if (p->M < 0 || p->M % tile_rows != 0 || p->M >= (1ll << 31))
    return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: M must be a multiple of pna_fused_degree_tile_rows(F, N) (64, or 128 for the wide shapes)");
  if (p->n_records < 4 || p->n_records * 64 >= (1ll << 32))
    return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: tile_ids must hold 4 <= n_records < 2^26 records");
  if (p->n_nodes < 1 || p->ldy < p->N || p->n_nodes * p->ldy * 4 >= (1ll << 32) ||
      (p->residual && (p->ld_res < p->N || p->n_nodes * p->ld_res * 4 >= (1ll << 32))))
    return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: bad n_nodes / ldy / ld_res (y and residual must be < 4 GiB)");
  if (p->relu < 0 || p->relu > 2) return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: relu must be 0, 1 or 2");
  if (p->y_cols_writable != 0 && (p->y_cols_writable < p->N || p->y_cols_writable > p->ldy || p->y_cols_writable > (p->N + 15) / 16 * 16))
    return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: y_cols_writable must be 0 or in [N, min(ldy, 16 ceil(N / 16))]");
  if (p->spare_workgroups < 0) return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: spare_workgroups must be >= 0");
  if ((p->col_scale == nullptr) != (p->col_shift == nullptr))
    return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: col_scale and col_shift come together");
  if (p->agg_out && p->ld_agg < 4 * (int64_t)p->F) return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: ld_agg < 4 F");
  if (p->agg_out && arith != PNA_FD_ARITH_H2)
    return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: agg_out (the verification instantiation) takes arith = PNA_FD_ARITH_H2");
  const bool tower = p->x_dst || p->h_self;                // (row_post alone: the layer proper with a row factor -- round 6)
  if (p->pre_add && p->agg_out) return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: pre_add is not for the verification instantiation (agg_out)");
  if (p->pre_add && (p->ld_pre_add < p->N || ((uintptr_t)p->pre_add & 3) != 0 || p->n_nodes * p->ld_pre_add * 4 >= (1ll << 32)))
    return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: pre_add must be a 4-byte aligned (n_nodes, >= N) table below 4 GiB");
  if (p->row_post && !tower && p->agg_out) return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: row_post is not for the verification instantiation (agg_out)");
  if (tower) {
    if (!p->x_dst || !p->h_self || !p->row_post || p->agg_out || !tower_shape_ok(p->F, p->N))
      return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: tower mode takes x_dst, h_self and row_post together, 49 <= F <= 80, no agg_out");
    if (p->ld_xdst < p->F || ((uintptr_t)p->x_dst & 3) != 0 || p->ld_h < p->F || ((uintptr_t)p->h_self & 3) != 0 ||
        p->n_nodes >= (1 << 24) || p->n_nodes * p->ld_xdst * 4 >= (1ll << 32) || p->n_nodes * p->ld_h * 4 >= (1ll << 32))
      return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: x_dst / h_self must be 4-byte aligned (n_nodes, >= F) tables below 4 GiB and 2^24 rows");
  }
  // the images lie exactly pna_fused_image_bytes apart: the pack functions' layout (ABI 20; a padded stride would misplace the tails)
  if ((need_h2 && p->image_stride != image_bytes(p->F, p->N, tower, 0)) || (need_x3 && p->image_stride_x3 != image_bytes(p->F, p->N, tower, 1)))
    return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: image_stride / image_stride_x3 must equal pna_fused_image_bytes(F, N, tower, 0 / 1)");
  if (arith == PNA_FD_ARITH_GUARDED && (!p->guard_ws || p->guard_ws_bytes < pna_fused_degree_guard_bytes(p->M) || ((uintptr_t)p->guard_ws & 15) != 0))
    return pna_set_error(PNA_E_INVALID, "pna_fused_degree_f32: PNA_FD_ARITH_GUARDED needs a 16-byte aligned guard_ws of pna_fused_degree_guard_bytes(M) bytes (zero before its first use)");
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    return pna_set_error(PNA_E_NODEVICE, "pna_fused_degree_f32: no device");
  FDArgs g;
  memset(&g, 0, sizeof(g));
  g.tdesc = reinterpret_cast<const i4*>(p->tile_desc); g.ids = p->tile_ids; g.x = p->x; g.ldb = (unsigned)(p->ldx * 4); g.F = p->F;
  g.perm = p->row_perm; g.w_img = (const unsigned char*)p->w_img; g.img_stride = p->image_stride;
  g.bias = p->bias; g.col_scale = p->col_scale; g.col_shift = p->col_shift; g.residual = p->residual; g.y = p->y;
  g.ldyb = (unsigned)(p->ldy * 4); g.ldrb = p->residual ? (unsigned)(p->ld_res * 4) : 0u;
  g.M = (int)p->M; g.N = p->N; g.relu = p->relu; g.slope = p->relu == 2 ? p->act_slope : 0.f;
  g.ncw = p->y_cols_writable > p->N ? (int)p->y_cols_writable : p->N;
  g.agg_out = p->agg_out; g.ld_agg = p->ld_agg;
  g.xd = p->x_dst; g.xh = p->h_self; g.row_post = p->row_post; g.lddb = (unsigned)(p->ld_xdst * 4); g.ldhb = (unsigned)(p->ld_h * 4);
  g.counter = p->tile_counter;
  g.pre_add = p->pre_add; g.ldpb = (unsigned)(p->ld_pre_add * 4);
#ifdef PNA_AMD_EXPERIMENTS
  if (const char* e = getenv("PNA_FD_DBG_PTR")) g.dbg = (unsigned long long*)strtoull(e, nullptr, 0);   // device buffer: 8 counters per wavefront
  if (const char* e = getenv("PNA_FD_ABL")) g.abl = atoi(e);
#endif
  const int ntiles = (int)(p->M / 64);                     // in 64-row units: launch() halves the grid for 8-wavefront workgroups
  int per_cu = 2;
#ifdef PNA_AMD_EXPERIMENTS
  if (const char* e = getenv("PNA_FD_WGS")) per_cu = atoi(e) > 0 ? atoi(e) : 2;
#endif
  if (p->agg_out) per_cu = 1;                             // (the verification instantiation is built for one workgroup per CU)
  int wgs = per_cu * cus;
  if (p->spare_workgroups > 0) wgs = wgs - p->spare_workgroups > cus ? wgs - p->spare_workgroups : cus;   // (never below one per CU)
  if (ntiles < wgs) wgs = ntiles;
  hipStream_t st = (hipStream_t)stream;
  int rc = 0;
  if (arith == PNA_FD_ARITH_X3) {
    g.w_img = (const unsigned char*)p->w_img_x3; g.img_stride = p->image_stride_x3;
    rc = launch_shape<false, 1>(g, wgs, st);
  } else if (p->agg_out) {
    rc = launch_shape<true, 2>(g, wgs, st);
  } else {
    if (arith == PNA_FD_ARITH_GUARDED) {
      unsigned char* ws = (unsigned char*)p->guard_ws;
      g.guard = reinterpret_cast<int*>(ws);
      g.g_desc = reinterpret_cast<i4*>(ws + 64);
      g.g_perm = reinterpret_cast<int*>(ws + 64 + p->M);
      g.g_post = reinterpret_cast<float*>(ws + 64 + 5 * p->M);
    }
    rc = launch_shape<false, 0>(g, wgs, st);
    if (rc == 0 && arith == PNA_FD_ARITH_GUARDED) {
      // the consuming launch: the SAME layer in bf16 x 3 over the tiles handed over (their count lives on the device: a grid of
      // workgroups that almost always find nothing and leave at once)
      FDArgs c = g;
      c.w_img = (const unsigned char*)p->w_img_x3; c.img_stride = p->image_stride_x3;
      c.tdesc = g.g_desc; c.perm = g.g_perm; c.row_post = p->row_post ? g.g_post : nullptr;
      c.counter = nullptr; c.guard = nullptr; c.list_count = g.guard;
      rc = launch_shape<false, 1>(c, wgs, st);
    }
  }
  if (rc != 0) return pna_set_error(PNA_E_LAUNCH, rc == -2 ? "pna_fused_degree_f32: no instantiation for this F" : "pna_fused_degree_f32: hipFuncSetAttribute failed");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}
