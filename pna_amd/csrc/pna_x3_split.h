// pna_x3_split.h -- the exact fp32 -> 3 x bf16 operand split shared by the bf16x3 contraction kernels
// (pna_posttrans_x3.hip, pna_fused_degree.hip's tower mode), and the fp32 -> 2 x fp16 split of the one-kernel layer (below).
//   x = x0 + x1 + x2,   x0 = top 16 bits of x,  x1 = top 16 bits of (x - x0),  x2 = top 16 bits of (x - x0 - x1)
// by truncation, so every term is exact and finite inputs never overflow.  See include/pna_amd.h for the non-finite rules.
#ifndef PNA_X3_SPLIT_H
#define PNA_X3_SPLIT_H
#include <hip/hip_runtime.h>

namespace pna_x3 {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef short bf8 __attribute__((ext_vector_type(8)));       // 8 bf16 = one MFMA A/B fragment
typedef unsigned u4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) f4u { f4 v; };

__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bfloat(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ float top16(float x) { return bfloat(fbits(x) & 0xFFFF0000u); }
// upper halves of (even, odd) -> one dword {odd.hi16, even.hi16}
__device__ __forceinline__ unsigned pack_hi(float even, float odd) { return __builtin_amdgcn_perm(fbits(odd), fbits(even), 0x07060302u); }

// 8 floats -> the three bf16 fragments
__device__ __forceinline__ void split8(const f4 lo, const f4 hi, bf8& t0, bf8& t1, bf8& t2) {
  const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  u4 p0, p1, p2;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float xe = x[2 * j], xo = x[2 * j + 1];
    const float re = xe - top16(xe), ro = xo - top16(xo);
    const float se = re - top16(re), so = ro - top16(ro);
    p0[j] = pack_hi(xe, xo);
    p1[j] = pack_hi(re, ro);
    p2[j] = pack_hi(se, so);
  }
  t0 = __builtin_bit_cast(bf8, p0); t1 = __builtin_bit_cast(bf8, p1); t2 = __builtin_bit_cast(bf8, p2);
}

// The same for a fragment that holds +-Inf: an infinite element is carried by its LOWEST term alone (t0 = t1 = 0,
// t2 = +-Inf): of the six partial products only a2*b0 sees it, and b0 = 0 only where the fp32 product Inf * w is NaN too.
__device__ __forceinline__ void split8_inf(const f4 lo, const f4 hi, bf8& t0, bf8& t1, bf8& t2) {
  const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  u4 p0, p1, p2;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float xe = x[2 * j], xo = x[2 * j + 1];
    const bool ie = __builtin_fabsf(xe) == INFINITY, io = __builtin_fabsf(xo) == INFINITY;
    const float fe = ie ? 0.f : xe, fo = io ? 0.f : xo;
    const float re = fe - top16(fe), ro = fo - top16(fo);
    const float se = re - top16(re), so = ro - top16(ro);
    p0[j] = pack_hi(fe, fo);
    p1[j] = pack_hi(re, ro);
    p2[j] = pack_hi(ie ? xe : se, io ? xo : so);
  }
  t0 = __builtin_bit_cast(bf8, p0); t1 = __builtin_bit_cast(bf8, p1); t2 = __builtin_bit_cast(bf8, p2);
}

// largest magnitude of the 8 floats (NaN operands are ignored by v_max3: they need no special path)
__device__ __forceinline__ float absmax8(const f4 lo, const f4 hi) {
  float m;
  asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(m) : "v"(lo.x), "v"(lo.y), "v"(lo.z));
  asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(m) : "v"(m), "v"(lo.w), "v"(hi.x));
  asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(m) : "v"(m), "v"(hi.y), "v"(hi.z));
  asm("v_max_f32 %0, %1, |%2|" : "=v"(m) : "v"(m), "v"(hi.w));
  return m;
}

// A 16-byte window [kk, kk+4) with kk = max(0, min(k, kmax-4)) was loaded instead of [k, k+4): realign it to the elements
// [k, k+4) and zero those at or beyond kmax.
__device__ __forceinline__ f4 fix4(int k, int kmax, f4 t) {
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
}

// one weight -> its bf16 term `term` (pack kernels); an infinite weight is carried by its lowest term alone
__device__ __forceinline__ unsigned short weight_term(float w, int term) {
  const bool winf = __builtin_fabsf(w) == INFINITY;
  const float wf = winf ? 0.f : w;
  const float r1 = wf - top16(wf), r2 = winf ? w : r1 - top16(r1);
  const float t = term == 0 ? wf : term == 1 ? r1 : r2;
  return (unsigned short)(fbits(t) >> 16);
}


// ---- fp16 x 2 (round 5; guarded in round 6: the one-kernel layer's contraction, pna_fused_degree.hip) --------------------------
// x = h0 + h1 + r,  h0 = fp16(x), h1 = fp16(x - h0) (round to nearest, both): two terms carry 22-23 of fp32's 24 significand bits, and
// THREE partial products (h1 w0, h0 w1, h0 w0; the dropped h1 w1 is 2^-22 of the product) do what bf16 x 3 needs six for -- provided
// the operand sits high in fp16's narrow range: the caller scales a row of statistics / a column of weights by a power of two (exact)
// so that its bound lands in [2^14, 2^15) (statistics: twice the row's largest message magnitude) / [2^13, 2^14) (weights: the column's
// largest magnitude), and scales the accumulator back.  In these units
//     |r| <= max(2^-22 |x|, 2^-25):   an operand below 2^-3 sits on fp16's SUBNORMAL grid with its second term -- an absolute error of
// up to 2^-25, i.e. a relative one of 2^-25 / |x| ("normwise-, not componentwise-accurate").  That FLOOR error is what the
// guard of pna_fused_degree.hip bounds per output and, where it could matter, sends to the bf16 x 3 arithmetic (see `floor_exp` below).
// Measured against float64 on the benchmark layer's shapes: 1.25x the bf16 x 3 error, a fifth of an fp32 GEMM's.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));     // 8 fp16 = one MFMA A/B fragment

__device__ __forceinline__ unsigned cvt_pk_h(float lo, float hi) { unsigned r; asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r; }
__device__ __forceinline__ float h_lo(unsigned p) { float r; asm("v_cvt_f32_f16 %0, %1" : "=v"(r) : "v"(p)); return r; }
__device__ __forceinline__ float h_hi(unsigned p) {
  float r; asm("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r) : "v"(p)); return r;
}
__device__ __forceinline__ float sub_1(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float mul_1(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// 8 floats x scl (a power of two: exact) -> the two fp16 fragments.  Single VALU instructions through inline asm, like split8's
// callers: hipcc would pack the chains into v_pk_*_f32 with op_sel swizzles.
// floor_exp (round 6): min(floor_exp, frexp exponents of the eight SCALED values) -- v_frexp_exp_i32_f32 gives e with |u| in [2^(e-1), 2^e)
// and 0 for u = 0, so "floor_exp <= -3" says: one of the values is non-zero and below 2^-3, where the two-term split has an absolute
// (not a relative) error bound.  Three instructions per pair of values.
__device__ __forceinline__ void split8_h2(const f4 lo, const f4 hi, float scl, h8& t0, h8& t1, int& floor_exp) {
  const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  // ONE statement for the eight values (see fold4 in pna_fused_degree.hip: statement by statement hipcc puts an `s_nop 0` behind every
  // instruction whose result -- or whose scratch register -- the next statement touches: five per pair of values here)
  u4 p0, p1;
  float a, b, c, d;
#define PNA_H2_PAIR(P0, P1, XE, XO)                                                                                   \
  "v_mul_f32 %8, " XE ", %21\n\tv_mul_f32 %9, " XO ", %21\n\t"                                                       \
  "v_frexp_exp_i32_f32 %10, %8\n\tv_frexp_exp_i32_f32 %11, %9\n\tv_min3_i32 %12, %12, %10, %11\n\t"                  \
  "v_cvt_pk_f16_f32 " P0 ", %8, %9\n\t"                                                                              \
  "v_cvt_f32_f16 %10, " P0 "\n\tv_cvt_f32_f16_sdwa %11, " P0 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t" \
  "v_sub_f32 %8, %8, %10\n\tv_sub_f32 %9, %9, %11\n\tv_cvt_pk_f16_f32 " P1 ", %8, %9"
  asm(PNA_H2_PAIR("%0", "%4", "%13", "%14") "\n\t" PNA_H2_PAIR("%1", "%5", "%15", "%16") "\n\t"
      PNA_H2_PAIR("%2", "%6", "%17", "%18") "\n\t" PNA_H2_PAIR("%3", "%7", "%19", "%20")
      : "=&v"(p0[0]), "=&v"(p0[1]), "=&v"(p0[2]), "=&v"(p0[3]), "=&v"(p1[0]), "=&v"(p1[1]), "=&v"(p1[2]), "=&v"(p1[3]),
        "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "+v"(floor_exp)
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(scl));
#undef PNA_H2_PAIR
  t0 = __builtin_bit_cast(h8, p0); t1 = __builtin_bit_cast(h8, p1);
}
// The same for a fragment that holds +-Inf: an infinite element is carried by its LOWER term alone (t0 = 0, t1 = +-Inf): of the three
// partial products only h1 w0 sees it, and w0 = 0 only where the weight is zero (fp32: Inf * 0 = NaN too) or 2^-39 below its column's largest.
__device__ __forceinline__ void split8_h2_inf(const f4 lo, const f4 hi, float scl, h8& t0, h8& t1) {
  const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  u4 p0, p1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float xe = mul_1(x[2 * j], scl), xo = mul_1(x[2 * j + 1], scl);
    const bool ie = __builtin_fabsf(xe) == INFINITY, io = __builtin_fabsf(xo) == INFINITY;
    const float fe = ie ? 0.f : xe, fo = io ? 0.f : xo;
    p0[j] = cvt_pk_h(fe, fo);
    const float re = sub_1(fe, h_lo(p0[j])), ro = sub_1(fo, h_hi(p0[j]));
    p1[j] = cvt_pk_h(ie ? xe : re, io ? xo : ro);
  }
  t0 = __builtin_bit_cast(h8, p0); t1 = __builtin_bit_cast(h8, p1);
}

// power-of-two scale that puts a magnitude bound into [2^13, 2^14) (a column of weights) / [2^14, 2^15) (a row of statistics: every
// statistic is below its row's bound, and 2^15 is below fp16's largest number): the exponent s (bound * 2^s); bound in (0, FLT_MAX]
__device__ __forceinline__ int h2_scale_exp(float bound) { return 14 - __builtin_amdgcn_frexp_expf(bound); }
__device__ __forceinline__ int h2_row_scale_exp(float bound) { return 15 - __builtin_amdgcn_frexp_expf(bound); }
// The guard's arithmetic, in the scaled units.  u = h0 + h1 + e with |e| <= max(2^-22 |u|, 2^-25): the part of |e| above
// 2^-22 |u| is the operand's FLOOR error -- zero unless 0 < |u| < 2^kFloorExp.  With fs = the sum of a row's statistics' floor errors and fw = the
// sum of a column's weights' the floor error of an accumulator is at most  fs max|w| + fw max|u| <= fs 2^14 + fw 2^15,  and an output is
// CERTIFIED when that is below 2^-20 of its magnitude:  |acc| >= fs 2^34 + fw 2^35.  Everything else of the split's error is relative to
// |u| |w| term by term -- componentwise, like fp32's own rounding.
constexpr int kFloorExp = -3;
constexpr float kFloorStatScale = 17179869184.f /* 2^34 */, kFloorWeightScale = 34359738368.f /* 2^35 */;
// floor error of one scaled operand (exact: h0, h1 as the split rounds them, every difference representable)
__device__ __forceinline__ float h2_floor_error(float u) {
  const _Float16 h0 = (_Float16)u;
  const float r = u - (float)h0;
  const _Float16 h1 = (_Float16)r;
  const float e = __builtin_fabsf(r - (float)h1) - 0x1p-22f * __builtin_fabsf(u);
  return e > 0.f ? e : 0.f;                                  // (NaN / Inf operands: 0 -- non-finite rows are not the guard's)
}
// one weight (already multiplied by its column's power of two) -> its fp16 term `term`; an infinite weight is carried by its lower term alone
__device__ __forceinline__ unsigned short weight_term_h2(float w, int term) {
  const bool winf = __builtin_fabsf(w) == INFINITY;
  const float wf = winf ? 0.f : w;
  const _Float16 t0 = (_Float16)wf;
  const _Float16 t1 = winf ? (_Float16)w : (_Float16)(wf - (float)t0);
  return __builtin_bit_cast(unsigned short, term == 0 ? t0 : t1);
}

}  // namespace pna_x3
#endif
