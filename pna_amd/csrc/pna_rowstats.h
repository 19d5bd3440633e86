// Device helpers shared by the kernels that reduce a destination row's messages (pna_segreduce.hip, pna_fused.hip,
// pna_tower_fused.hip): the single-instruction max / min and the correctly rounded per-row division.  Not part of the C ABI.
#ifndef PNA_ROWSTATS_H
#define PNA_ROWSTATS_H
#include <hip/hip_runtime.h>

namespace pna_dev {

// a / b, correctly rounded, from the correctly rounded reciprocal r = RN(1 / b) (ONE IEEE division per row): Markstein's
// sequence q0 = RN(a r), e = RN(a - b q0) (exact in an fma), q = RN(q0 + e r) equals RN(a / b) for every b whose significand
// is not all ones (an in-degree of 2^24 - 1 does not occur: rows < 2^24).  This is the reference's `sum / D`
// (models/dgl/aggregators.py:6-7,:22-26) bit for bit at a third of the cost of a per-feature division.  It matters beyond
// the last ulp: with equal neighbours the reference's var = E[x^2] - E[x]^2 is EXACTLY 0 (std = sqrt(1e-5)), and a mean
// that is 1 ulp off turns that into ~1e-7 x^2 -- 5 % of the std for |x| ~ 3 (seen on the multitask GNN's later iterations).
// Non-finite or zero b falls through to the plain product (the result is then NaN / Inf either way).
__device__ __forceinline__ float div_rn(float a, float b, float r) {
  const float q0 = a * r;
  const float e = __builtin_fmaf(-b, q0, a);
  const float q = __builtin_fmaf(e, r, q0);
  return (q == q && __builtin_fabsf(q) != INFINITY) ? q : q0;
}

// Single-instruction max/min (no canonicalisation prologue; a quiet-NaN operand is ignored).
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmin(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// mean | max | min | std of one feature of one destination row from its running sums (models/dgl/aggregators.py:6-26):
// s = sum m, q = sum m*m (each product rounded, -ffp-contract=off), mx / mn through vmax / vmin (which drop NaN: q is NaN
// iff some message is NaN -- its terms are >= 0, so infinities never cancel -- and then replaces max and min, as torch's
// max / min propagate NaN).  deg <= 0: all four are 0 (DGL leaves rows without in-edges at their zero initial value).
__device__ __forceinline__ void row_stats(float s, float q, float mx, float mn, int deg, float& mean, float& omx, float& omn, float& sd) {
  if (deg <= 0) { mean = omx = omn = sd = 0.f; return; }
  const float D = (float)deg, invD = 1.0f / D;
  mean = div_rn(s, D, invD);
  const float msq = div_rn(q, D, invD);
  float var = msq - mean * mean;
  var = var < 0.f ? 0.f : var;
  omx = q != q ? q : mx;
  omn = q != q ? q : mn;
  sd = sqrtf(var + 1e-5f);
}

}  // namespace pna_dev
#endif
