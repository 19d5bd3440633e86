// MFMA issue-rate micro-benchmark for v_mfma_f32_32x32x16_bf16 / 16x16x32: cycles per instruction as a function of how many
// independent accumulators a wavefront rotates through and how many wavefronts share a SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_dep.hip -o tools/ubench/mfma_dep && tools/ubench/mfma_dep
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8m __attribute__((ext_vector_type(8)));

template <int NACC, bool BIG>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  bf8m a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i - 3); }
  f16v acc[NACC];
  f4v acs[NACC];
  for (int n = 0; n < NACC; ++n) {
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    acs[n] = (f4v){0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 6; ++rep)
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
        if constexpr (BIG) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
        else acs[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acs[n], 0, 0, 0);
      }
  }
  const unsigned long long t1 = clock64();
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) {
    if constexpr (BIG) for (int i = 0; i < 16; ++i) s += acc[n][i];
    else for (int i = 0; i < 4; ++i) s += acs[n][i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, bool BIG>
void run(int waves_per_simd) {
  float* out; unsigned long long* cyc;
  const int blocks = 256, threads = 256 * waves_per_simd, iters = 2000;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  hipLaunchKernelGGL((k<NACC, BIG>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, BIG>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double c = 0; for (int i = 0; i < blocks; ++i) c += (double)h[i]; c /= blocks;
  const double n_mfma = (double)iters * 6 * NACC;               // per wavefront
  const double flops = n_mfma * (BIG ? 32768.0 : 16384.0) * blocks * (threads / 64);
  printf("%s NACC=%d waves/SIMD=%d: %.1f cycles per MFMA per wave, %.1f per SIMD-slot; %.0f TF/s\n", BIG ? "32x32x16" : "16x16x32", NACC,
         waves_per_simd, c / n_mfma, c / n_mfma / waves_per_simd, flops / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 3; ++w) {
    run<1, true>(w); run<2, true>(w); run<3, true>(w); run<4, true>(w); run<7, true>(w);
    run<1, false>(w); run<2, false>(w); run<4, false>(w); run<15, false>(w);
  }
  return 0;
}
