// gather_ceiling.hip -- what the memory system gives a RANDOM row gather shaped like the one-kernel layer's (development, not shipped).
//
// A wavefront gathers for 16 "rows" at a time: lane (li = lane & 15, lg = lane >> 4) reads 32 bytes (two dwordx4) of every 128-byte
// block of the source row its row li names -- NB blocks per source row, back to back, like pna_fused_degree.hip's edge packet -- with
// R packets in flight (register ring, counted waits by the compiler: plain C++ loads, unrolled).  The source ids come from a hash
// (no id stream), the loaded values are folded into one add per dword (no statistics, no multiply): the time is the memory system's.
//
//   gather_ceiling <table MiB> <row pitch bytes> <blocks per read: 2|3|4> <packets in flight: 4|6|8> <edges per row-slot> <waves per CU: 4|8|16>
//
// prints: bytes requested (lines touched x 128) / time.  Used for DESIGN.md 4.8.13 (C5: a 1 GiB table is served by HBM, not by the
// 256 MiB Infinity Cache: what is the ceiling of 256-byte vs 512-byte random reads?).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/gather_ceiling.hip -o tools/ubench/gather_ceiling
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int NB, int R>
__global__ __launch_bounds__(256, 2) void k_gather(const char* __restrict__ x, unsigned n_rows, unsigned pitch, int edges, int tiles_per_wave,
                                                float* __restrict__ out, unsigned off2) {
  extern __shared__ float dummy[];                        // occupancy cap only
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
  const int li = lane & 15, lg = lane >> 4;
  float acc = 0.f;
  for (int t = 0; t < tiles_per_wave; ++t) {
    const unsigned seed = (unsigned)(wave * tiles_per_wave + t) * 2654435761u + (unsigned)li * 40503u;
    f4 v[R][NB][2];
    auto issue = [&](int slot, int e) __attribute__((always_inline)) {
      const unsigned id = __umulhi(mix(seed + (unsigned)e * 0x9e3779b9u), n_rows);
      const char* p = x + (size_t)id * pitch + lg * 32 + off2;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        v[slot][b][0] = *reinterpret_cast<const f4*>(p + b * 128);
        v[slot][b][1] = *reinterpret_cast<const f4*>(p + b * 128 + 16);
      }
    };
#pragma unroll
    for (int s = 0; s < R; ++s) issue(s, s);
    for (int e = 0; e < edges; e += R) {
#pragma unroll
      for (int s = 0; s < R; ++s) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          acc += v[s][b][0][0] + v[s][b][0][1] + v[s][b][0][2] + v[s][b][0][3];
          acc += v[s][b][1][0] + v[s][b][1][1] + v[s][b][1][2] + v[s][b][1][3];
        }
        asm volatile("" ::: "memory");                     // (keeps hipcc from hoisting the next packets' loads above this fold)
        issue(s, e + R + s);                              // keeps R packets in flight; the tail re-reads R extra rows
        asm volatile("" ::: "memory");
      }
    }
#pragma unroll
    for (int s = 0; s < R; ++s)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc += v[s][b][0][0] + v[s][b][1][0];
  }
  out[(size_t)wave * 64 + lane] = acc;
}

template <int NB, int R>
static float run(const char* x, unsigned n_rows, unsigned pitch, int edges, int tpw, int wgs, size_t lds, float* out, unsigned off2) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipFuncSetAttribute((const void*)k_gather<NB, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_gather<NB, R>), dim3(wgs), dim3(256), lds, 0, x, n_rows, pitch, edges, tpw, out, off2);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a, 0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_gather<NB, R>), dim3(wgs), dim3(256), lds, 0, x, n_rows, pitch, edges, tpw, out, off2);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    if (ms / 5 < best) best = ms / 5;
  }
  return best;
}

int main(int argc, char** argv) {
  if (argc < 7) { fprintf(stderr, "usage: %s <table MiB> <pitch B> <blocks 2|3|4> <in flight 4|8> <edges> <waves per CU>\n", argv[0]); return 2; }
  const size_t mib = strtoull(argv[1], nullptr, 0);
  const unsigned pitch = (unsigned)atoi(argv[2]);
  const int nb = atoi(argv[3]), r = atoi(argv[4]), edges = atoi(argv[5]), wpc = atoi(argv[6]);
  const unsigned off2 = argc > 7 ? (unsigned)atoi(argv[7]) : 0u;     // byte offset of the read inside the row (a second pass's half)
  const size_t bytes = mib << 20;
  const unsigned n_rows = (unsigned)(bytes / pitch) - 1;
  char* x = nullptr; float* out = nullptr;
  if (hipMalloc(&x, bytes + 4096) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  hipMemset(x, 0, bytes + 4096);
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int wg_per_cu = wpc / 4;
  const int wgs = cus * wg_per_cu;
  const size_t lds = (size_t)(160 * 1024 / wg_per_cu - 1024) / 4 * 4;   // cap the occupancy at wg_per_cu
  const int tpw = 24;
  hipMalloc(&out, (size_t)wgs * 256 * 4);
  float ms = -1.f;
  if (nb == 2 && r == 4) ms = run<2, 4>(x, n_rows, pitch, edges, tpw, wgs, lds, out, off2);
  else if (nb == 2 && r == 6) ms = run<2, 6>(x, n_rows, pitch, edges, tpw, wgs, lds, out, off2);
  else if (nb == 3 && r == 4) ms = run<3, 4>(x, n_rows, pitch, edges, tpw, wgs, lds, out, off2);
  else if (nb == 4 && r == 4) ms = run<4, 4>(x, n_rows, pitch, edges, tpw, wgs, lds, out, off2);
  else if (nb == 4 && r == 5) ms = run<4, 5>(x, n_rows, pitch, edges, tpw, wgs, lds, out, off2);
  else if (nb == 1 && r == 8) ms = run<1, 8>(x, n_rows, pitch, edges, tpw, wgs, lds, out, off2);
  else { fprintf(stderr, "no instantiation\n"); return 2; }
  const double reads = (double)wgs * 4 * tpw * 16 * (edges + r);
  const double gb = reads * nb * 128 / 1e9;
  printf("{\"table_MiB\": %zu, \"pitch\": %u, \"blocks\": %d, \"granule_B\": %d, \"in_flight\": %d, \"edges\": %d, \"waves_per_cu\": %d, \"offset\": %u, "
         "\"row_reads_M\": %.2f, \"ms\": %.4f, \"TB_per_s\": %.3f}\n",
         mib, pitch, nb, nb * 128, r, edges, wpc, off2, reads / 1e6, ms, gb / ms);
  hipFree(x); hipFree(out);
  return 0;
}
