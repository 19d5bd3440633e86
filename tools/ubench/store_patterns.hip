// How fast can a wave-tile of 16 rows x 1600 B (pitch 1600 B, 1.6 GB in all) be WRITTEN, by the shape of one store instruction?
//   0: 1 KB contiguous per instruction (the tile as a flat 25 600 B run)         1: 16 rows x 64 B per instruction (MFMA 16x16 transposed output)
//   2: as 1 with `nt`            3: 4 rows x 256 B per instruction               4: 2 rows x 512 B       5: as 1, five instructions of one row span back to back, sc1
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/store_patterns.hip -o tools/ubench/store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int P>
__global__ __launch_bounds__(512) void k(float* y, long M) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f4 v = {1.f, 2.f, 3.f, (float)lane};
  for (long r0 = ((long)blockIdx.x * 8 + wave) * 16; r0 + 16 <= M; r0 += (long)gridDim.x * 128) {
    float* const t = y + r0 * 400;
    if (P == 0) {
#pragma unroll
      for (int i = 0; i < 25; ++i) *(f4*)(t + i * 256 + lane * 4) = v;
    } else if (P == 1 || P == 2 || P == 5) {
      float* const o = t + (lane & 15) * 400 + (lane >> 4) * 4;
#pragma unroll
      for (int i = 0; i < 25; ++i) {
        if (P == 1) *(f4*)(o + i * 16) = v;
        else if (P == 2) __builtin_nontemporal_store(v, (f4*)(o + i * 16));
        else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(o + i * 16), "v"(v) : "memory");
      }
    } else if (P == 3) {
      float* const o = t + (lane >> 4) * 400 + (lane & 15) * 4;          // 4 rows x 64 floats
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 6; ++i) *(f4*)(o + q * 1600 + i * 64) = v;   // (384 of the 400 columns: the shape, not the exact bytes)
    } else {
      float* const o = t + (lane >> 5) * 400 + (lane & 31) * 4;          // 2 rows x 128 floats
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int i = 0; i < 3; ++i) *(f4*)(o + q * 800 + i * 128) = v;
    }
  }
}
template <int P> void run(float* y, long M) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<P>, dim3(256), dim3(512), 0, 0, y, M);
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<P>, dim3(256), dim3(512), 0, 0, y, M);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
  const double bytes = (P == 3 ? 384.0 : P == 4 ? 384.0 : 400.0) * 4 * M;
  printf("pattern %d: %.4f ms  %.0f GB/s\n", P, ms, bytes / ms * 1e-6);
}
int main() {
  const long M = 1000000; float* y; hipMalloc(&y, M * 1600);
  run<0>(y, M); run<1>(y, M); run<2>(y, M); run<3>(y, M); run<4>(y, M); run<5>(y, M); run<0>(y, M);
  return 0;
}
