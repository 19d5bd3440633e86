// pk_sanity2.hip -- which context makes v_pk_add_f32 return hi == lo?  (follow-up of pk_mfma_repro.hip: there EVERY packed sum's hi half
// equals its lo half, in every mode, while pk_sanity.hip's single instructions are right.)  One wavefront; sources from a per-lane
// global load; variants: 0 plain; 1 s_nop 7 between the wait and the packed op; 2 sources copied through v_mov first; 3 the loop form of
// the reproducer (accumulators loop-carried, two loads per iteration); 4 as 3 with single v_add_f32 (control).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, const float* x, int iters) {
  const int lane = threadIdx.x;
  const f4 v = *reinterpret_cast<const f4*>(x + lane * 4);
  f2 m = (f2){v[0], v[1]}, p0 = (f2){0.f, 0.f}, r0, r1, r2;
  asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r0) : "v"(p0), "v"(m));
  asm volatile("s_nop 7\n\tv_pk_add_f32 %0, %1, %2" : "=v"(r1) : "v"(p0), "v"(m));
  f2 mc; mc[0] = v[0] * 1.0f; mc[1] = v[1] * 1.0f;
  asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r2) : "v"(p0), "v"(mc));
  f2 P = (f2){0.f, 0.f}; float s0 = 0.f, s1 = 0.f;
  unsigned row = lane;
  for (int it = 0; it < iters; ++it) {
    row = row * 1664525u + 1013904223u;
    const f4 w = *reinterpret_cast<const f4*>(x + (row % 64u) * 4);
    const f2 mm = (f2){w[0], w[1]};
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(P) : "v"(P), "v"(mm));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(s0) : "v"(s0), "v"(w[0]));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(s1), "v"(w[1]));
  }
  float* o = out + lane * 12;
  o[0] = r0[0]; o[1] = r0[1]; o[2] = r1[0]; o[3] = r1[1]; o[4] = r2[0]; o[5] = r2[1]; o[6] = P[0]; o[7] = P[1]; o[8] = s0; o[9] = s1; o[10] = v[0]; o[11] = v[1];
}
int main() {
  float h[256], o[64 * 12], *dx, *dout;
  for (int i = 0; i < 256; ++i) h[i] = (float)(i + 1);
  hipMalloc(&dx, sizeof(h)); hipMalloc(&dout, sizeof(o)); hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, dx, 4); hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  for (int lane = 0; lane < 64; lane += 21) {
    const float* q = o + lane * 12;
    printf("lane %2d: m = (%g, %g); 0 + m: plain (%g, %g), after s_nop 7 (%g, %g), from v_mul copies (%g, %g); loop x4 packed (%g, %g) single (%g, %g)\n",
           lane, q[10], q[11], q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9]);
  }
  return 0;
}
