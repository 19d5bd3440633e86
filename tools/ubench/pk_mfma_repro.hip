// pk_mfma_repro.hip -- minimal reproducer attempt for the packed-fp32 wrong-sum observation.
// Two wavefronts per SIMD (one 512-thread workgroup per CU):
//   role M (wavefronts 0-3): a chain of v_mfma_f32_16x16x32_bf16 on four accumulators, nothing else;
//   role F (wavefronts 4-7): the fused kernel's fold -- two 16-byte loads per lane and iteration, then s += m, q += m * m for the 8
//     values with the packed instructions hipcc chose in the failing kernel (v_pk_add_f32 / v_pk_mul_f32, plain and op_sel, behind
//     v_pk_mov_b32 swaps) AND again from the SAME registers with single v_add_f32 / v_mul_f32: any differing bit is counted.
// mode 0: M + F (the failing co-residence); mode 1: F on all eight wavefronts; mode 2: F alone, one wavefront per SIMD (M idle).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ubench/pk_mfma_repro.hip -o tools/ubench/pk_mfma_repro
//   tools/ubench/pk_mfma_repro [iterations per launch] [launches]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;                       // a 64-bit VGPR pair: (lo, hi) floats of a packed operand
typedef short bf8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512, 2) void k_repro(const float* __restrict__ x, int rows, int iters, int mode, float* sink, unsigned* bad) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool fold_role = mode == 1 || wave >= 4;
  if (mode == 2 && wave < 4) return;
  if (!fold_role) {
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + lane + j); b[j] = (short)(0x3f00 + 2 * lane + j); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[n], 0, 0, 0);
      asm volatile("" : "+v"(a), "+v"(b));
    }
    sink[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    return;
  }
  // packed accumulators: pair k = values (2k, 2k+1); pairs 0,1 plain packed ops, pairs 2,3 the failing kernel's op_sel swizzles behind a v_pk_mov_b32 swap
  // (u64 operands, not float2: hipcc read element 0 of a float2 asm output in an array for BOTH halves -- DESIGN.md 4.8.5)
  u64 P[4], PQ[4];
  float Sr[8], Qr[8];
  auto pack = [](float lo, float hi) -> u64 { return (u64)__builtin_bit_cast(unsigned, lo) | ((u64)__builtin_bit_cast(unsigned, hi) << 32); };
  auto half = [](u64 v, int h) -> unsigned { return (unsigned)(v >> (32 * h)); };
#pragma unroll
  for (int k = 0; k < 4; ++k) { P[k] = 0; PQ[k] = 0; }
#pragma unroll
  for (int j = 0; j < 8; ++j) Sr[j] = Qr[j] = 0.f;
  unsigned row = (blockIdx.x * 8 + wave) * 64 + lane;
  for (int it = 0; it < iters; it += 2) {
    f4 v[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      row = (row * 1664525u + 1013904223u);
      const float* p = x + (size_t)(row % (unsigned)rows) * 8;
      v[u][0] = *reinterpret_cast<const f4*>(p);
      v[u][1] = *reinterpret_cast<const f4*>(p + 4);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u64 m = pack(v[u][k >> 1][2 * (k & 1)], v[u][k >> 1][2 * (k & 1) + 1]);
        u64 t;
        if (k < 2) {
          asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(P[k]) : "v"(P[k]), "v"(m));
          asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(t) : "v"(m));
          asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(PQ[k]) : "v"(PQ[k]), "v"(t));
        } else {
          u64 sw;
          asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(P[k]) : "v"(P[k]), "v"(m));       // lo += m.hi, hi += m.lo
          asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]" : "=v"(sw) : "v"(m));                                      // (m.hi, m.lo)
          asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(t) : "v"(sw));
          asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(PQ[k]) : "v"(PQ[k]), "v"(t));
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {                       // the same from the same registers, one VALU instruction each
        const float m = v[u][j >> 2][j & 3];
        float s1, p1, q1;
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(Sr[j]), "v"(m));
        asm volatile("v_mul_f32 %0, %1, %1" : "=v"(p1) : "v"(m));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(q1) : "v"(Qr[j]), "v"(p1));
        Sr[j] = s1; Qr[j] = q1;
      }
    }
  }
  // bad[0..3]: differing bits in plain sums | plain sums of squares | swizzled sums | swizzled sums of squares; bad[4..]: one example
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j0 = k < 2 ? 2 * k : 2 * k + 1, j1 = k < 2 ? 2 * k + 1 : 2 * k;      // swizzled pairs hold (value 2k+1, value 2k)
    const unsigned ds = (half(P[k], 0) != __builtin_bit_cast(unsigned, Sr[j0])) + (half(P[k], 1) != __builtin_bit_cast(unsigned, Sr[j1]));
    const unsigned dq = (half(PQ[k], 0) != __builtin_bit_cast(unsigned, Qr[j0])) + (half(PQ[k], 1) != __builtin_bit_cast(unsigned, Qr[j1]));
    if (ds) atomicAdd(bad + (k < 2 ? 0 : 2), ds);
    if (dq) atomicAdd(bad + (k < 2 ? 1 : 3), dq);
    if ((ds || dq) && atomicAdd(bad + 4, 1u) == 0) {
      bad[5] = blockIdx.x; bad[6] = threadIdx.x; bad[7] = k;
      bad[8] = half(P[k], 0); bad[9] = half(P[k], 1);
      bad[10] = __builtin_bit_cast(unsigned, Sr[j0]); bad[11] = __builtin_bit_cast(unsigned, Sr[j1]);
      bad[12] = half(PQ[k], 0); bad[13] = half(PQ[k], 1);
      bad[14] = __builtin_bit_cast(unsigned, Qr[j0]); bad[15] = __builtin_bit_cast(unsigned, Qr[j1]);
    }
  }
  sink[blockIdx.x * 512 + threadIdx.x] = __builtin_bit_cast(float, half(P[0], 0)) + __builtin_bit_cast(float, half(PQ[3], 1));
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4096, launches = argc > 2 ? atoi(argv[2]) : 50;
  const int rows = 1 << 22;                                // 128 MB table: the loads miss L2 like the gather's
  int dev = 0, cus = 0;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  float *x, *sink;
  unsigned* bad;
  hipMalloc(&x, (size_t)rows * 8 * 4);
  hipMalloc(&sink, (size_t)cus * 512 * 4);
  hipMalloc(&bad, 64);
  std::vector<float> hx((size_t)rows * 8);
  unsigned s = 12345;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 22)); }
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 3; ++mode) {
    hipMemset(bad, 0, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(k_repro, dim3(cus), dim3(512), 0, 0, x, rows, iters, mode, sink, bad);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned hb[16];
    hipMemcpy(hb, bad, 64, hipMemcpyDeviceToHost);
    printf("mode %d (%s): %d launches x %d CUs x %d iterations: differing sums plain %u / %u (s / q), swizzled %u / %u, %.3f ms per launch\n", mode,
           mode == 0 ? "MFMA wavefront + fold wavefront per SIMD" : mode == 1 ? "two fold wavefronts per SIMD" : "one fold wavefront per SIMD",
           launches, cus, iters, hb[0], hb[1], hb[2], hb[3], ms / launches);
    if (hb[4]) {
      printf("   first: block %u thread %u pair %u: packed s = %08x %08x, single s = %08x %08x; packed q = %08x %08x, single q = %08x %08x\n",
             hb[5], hb[6], hb[7], hb[8], hb[9], hb[10], hb[11], hb[12], hb[13], hb[14], hb[15]);
      if (iters <= 2) {                                   // the host's own sums for that lane (two rows)
        unsigned row = (hb[5] * 8 + (hb[6] >> 6)) * 64 + (hb[6] & 63);
        float e0 = 0, e1 = 0;
        for (int u = 0; u < 2; ++u) {
          row = row * 1664525u + 1013904223u;
          const float* pr = hx.data() + (size_t)(row % (unsigned)rows) * 8 + 2 * hb[7];
          e0 += pr[0]; e1 += pr[1];
        }
        union { float f; unsigned u; } a, b; a.f = e0; b.f = e1;
        printf("   host: sum of element 2k = %08x, of element 2k+1 = %08x\n", a.u, b.u);
      }
    }
  }
  return 0;
}
