// Times pna_project_f32 built with different -DPNA_PROJECT_* (tools/ubench/project_variants.sh builds one binary per variant).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "pna_amd.h"
int main(int argc, char** argv) {
  const long M = argc > 1 ? atol(argv[1]) : 1000000; const int K = 75, N = 400;
  float *x, *w, *y;
  hipMalloc(&x, M * K * 4); hipMalloc(&w, N * K * 4); hipMalloc(&y, M * N * 4);
  std::vector<float> hx(M * K), hw(N * K);
  for (auto& v : hx) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : hw) v = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(x, hx.data(), M * K * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), N * K * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) if (pna_project_f32(x, K, M, K, w, K, N, y, N, nullptr)) { printf("error %s\n", pna_last_error()); return 1; }
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) pna_project_f32(x, K, M, K, w, K, N, y, N, nullptr);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<float> hy(8 * N); hipMemcpy(hy.data(), y, 8 * N * 4, hipMemcpyDeviceToHost);
  double ref = 0, got = hy[3 * N + 77]; for (int k = 0; k < K; ++k) ref += (double)hx[3 * K + k] * hw[77 * K + k];
  printf("%s: %.4f ms  (y[3][77] %.6f ref %.6f)\n", argv[0], ms / 20, got, ref);
  return 0;
}
