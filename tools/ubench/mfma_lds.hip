// Do LDS reads (the B fragments) slow down the MFMA stream?  8 wavefronts per CU (2 per SIMD), each: ITERS x 42 MFMAs 32x32x16
// with NR ds_read_b128 (1 KB per wavefront each) spread between them, the reads feeding the MFMAs' B operand (DEP = 1) or not.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short bf8 __attribute__((ext_vector_type(8)));

template <int NR, int DEP>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<float*>(lds)[i] = 1.0f;
  __syncthreads();
  bf8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80); b[i] = (short)0x3f80; }
  f16v acc[7];
  for (int n = 0; n < 7; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + (threadIdx.x & 63) * 16;
  bf8 B[3];
  B[0] = B[1] = B[2] = b;
  for (int it = 0; it < iters; ++it) {
    const unsigned ba = base + (it & 3) * 16384;
#pragma unroll
    for (int n = 0; n < 7; ++n) {
#pragma unroll
      for (int tm = 0; tm < 3; ++tm)
        if (n * 3 + tm < NR) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(B[tm]) : "v"(ba), "n"((n * 3 + tm) * 1024 % 16384));
      if (NR > 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(B[0]), "+v"(B[1]), "+v"(B[2]));
#pragma unroll
      for (int pp = 0; pp < 6; ++pp)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[n]) : "v"(a), "v"(DEP ? B[pp % 3] : b));
    }
  }
  float s = 0.f;
  for (int n = 0; n < 7; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NR, int DEP>
void run() {
  float* out;
  const int blocks = 256, iters = 500;
  hipMalloc(&out, sizeof(float) * blocks * 512);
  hipFuncSetAttribute((const void*)k<NR, DEP>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL((k<NR, DEP>), dim3(blocks), dim3(512), 65536, 0, out, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NR, DEP>), dim3(blocks), dim3(512), 65536, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n_mfma = (double)iters * 42 * 8 * blocks;
  printf("NR=%2d DEP=%d: %.3f ms, MFMA %.0f TF/s, %.0f ns per step (2 waves x 42 MFMAs per SIMD)\n", NR, DEP, ms, n_mfma * 32768 / (ms * 1e-3) / 1e12, ms * 1e6 / iters);
  hipFree(out);
}

int main() {
  run<0, 0>(); run<7, 0>(); run<21, 0>(); run<21, 1>();
  return 0;
}
