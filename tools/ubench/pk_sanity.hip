// pk_sanity.hip -- semantics check of the packed-fp32 forms used by tools/ubench/pk_mfma_repro.hip (one wavefront, no co-residence)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out, const float* in) {
  f2 p = (f2){in[0], in[1]}, m = (f2){in[2], in[3]};
  f2 r1, r2, r3, r4;
  asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r1) : "v"(p), "v"(m));
  asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r2) : "v"(p), "v"(m));
  asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]" : "=v"(r3) : "v"(m));
  r4 = p + m;
  f2 acc = p;                                   // in place: vdst == src0, four times
  for (int i = 0; i < 4; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(m));
  f2 acc2 = p;                                  // separate "=v" / "v" operands like the reproducer
  for (int i = 0; i < 4; ++i) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc2) : "v"(acc2), "v"(m));
  out[0] = r1[0]; out[1] = r1[1]; out[2] = r2[0]; out[3] = r2[1]; out[4] = r3[0]; out[5] = r3[1]; out[6] = r4[0]; out[7] = r4[1];
  out[8] = acc[0]; out[9] = acc[1]; out[10] = acc2[0]; out[11] = acc2[1];
}
int main() {
  float h[4] = {1, 2, 10, 20}, o[12], *di, *dout;
  hipMalloc(&di, 16); hipMalloc(&dout, 48); hipMemcpy(di, h, 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, di); hipMemcpy(o, dout, 48, hipMemcpyDeviceToHost);
  printf("pk_add (1,2)+(10,20) = (%g, %g) expect (11, 22)\npk_add op_sel:[0,1] op_sel_hi:[1,0] = (%g, %g) expect (21, 12)\npk_mov op_sel:[1,0] = (%g, %g) expect (20, 10)\nC++ p + m = (%g, %g)\n", o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]);
  printf("in place x4: (%g, %g) expect (41, 82); separate operands x4: (%g, %g) expect (41, 82)\n", o[8], o[9], o[10], o[11]);
  return 0;
}
