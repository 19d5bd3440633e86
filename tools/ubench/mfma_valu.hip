// Do VALU instructions overlap with MFMAs on the same SIMD?  Each wavefront runs ITERS x (6 x NACC MFMAs 32x32x16 bf16 with
// NV independent v_fma_f32 behind each MFMA).  mode 0: every wavefront does both; mode 1: even wavefronts only MFMA, odd
// wavefronts only VALU (the same total instruction counts per SIMD pair of wavefronts).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8m __attribute__((ext_vector_type(8)));

template <int NV, int MODE>
__global__ void k(float* out, int iters) {
  bf8m a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i - 3); }
  f16v acc[4];
  for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  const int wave = threadIdx.x >> 6;
  const bool do_m = MODE == 0 || (wave & 4) == 0, do_v = MODE == 0 || (wave & 4) != 0;   // waves w and w+4 share a SIMD
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 6; ++rep)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if (do_m) {
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[n]) : "v"(a), "v"(b));
        }
        if (do_v) {
#pragma unroll
          for (int q = 0; q < (MODE == 0 ? NV : 2 * NV); ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[q & 7]) : "v"(v[(q + 3) & 7]));
        }
      }
  }
  float s = 0.f;
  for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int MODE>
void run() {
  float* out;
  const int blocks = 256, threads = 512, iters = 2000;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipLaunchKernelGGL((k<NV, MODE>), dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, MODE>), dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n_mfma = (double)iters * 24 * (MODE == 0 ? 8 : 4) * blocks;   // whole chip
  printf("mode %d NV=%d: %.3f ms, MFMA %.0f TF/s, %.1f ns per (MFMA + its VALU) per SIMD\n", MODE, NV, ms, n_mfma * 32768 / (ms * 1e-3) / 1e12,
         ms * 1e6 / ((double)iters * 24 * 2));
  hipFree(out);
}

int main() {
  run<0, 0>(); run<1, 0>(); run<2, 0>(); run<4, 0>(); run<6, 0>(); run<8, 0>(); run<12, 0>();
  run<1, 1>(); run<2, 1>(); run<4, 1>(); run<6, 1>(); run<8, 1>();
  return 0;
}
