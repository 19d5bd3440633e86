// Prototype (development, not shipped): the gather of a fused degree-grouped layer in the MFMA A layout (DESIGN.md 4.7 point 7).
// A wavefront owns 16 rows of ONE in-degree D (plan order); lane (li = lane & 15, lg = lane >> 4) accumulates sum / sum of
// squares / max / min of 8 features of row li -- features fb * 32 + lg * 8 .. + 8 of feature block fb -- over the D source rows:
// the four lanes of a row read one 128-byte strip of the source row (two 16-byte loads each).  Three feature blocks per tile
// (F <= 96), so every source row is visited three times as strips instead of once as 300 bytes.  The question: does this access
// pattern hold the bandwidth of the row-wise production gather?  mode 0: results reduced to one float per lane (no output
// stream); mode 1: the aggregate rows are stored ([mean | max | min | std] x F, plan order) -- for checking, and to see the store cost.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/ubench/gather_strip.hip -o tools/ubench/libgather_strip.so
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void k_gather_strip(const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ x,
                                                      long ldx, int F, const int* __restrict__ perm, int ntiles, float* __restrict__ out,
                                                      float* __restrict__ agg, long ld_agg) {
  extern __shared__ float dummy[];                       // occupancy cap only
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= ntiles) return;
  const int li = lane & 15, lg = lane >> 4;
  const int first = perm[tile * 16];
  if (first < 0) return;                                 // a tile of padding rows (wave-uniform)
  int node = perm[tile * 16 + li];
  const bool real = node >= 0;
  if (!real) node = first;
  const int beg = rowptr[node];
  const int D = __builtin_amdgcn_readfirstlane(rowptr[node + 1] - beg);
  const float invD = D > 0 ? 1.0f / (float)D : 0.f;
  float check = 0.f;
  const int nfb = (F + 31) / 32;
  for (int fb = 0; fb < nfb; ++fb) {
    int f0 = fb * 32 + lg * 8;
    const bool live = f0 < F;                            // (a strip past the row: re-read the row's last strip, results dropped)
    if (!live) f0 = (F - 1) / 8 * 8;
    const float* xb = x + f0;
    float s[8], q[8], mx[8], mn[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; mx[j] = -INFINITY; mn[j] = INFINITY; }
    for (int e = 0; e < D; e += U) {
      int idx[U];
#pragma unroll
      for (int u = 0; u < U; ++u) idx[u] = col[beg + min(e + u, D - 1)];
      f4 v[U][2];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float* p = xb + (size_t)idx[u] * ldx;
        v[u][0] = *reinterpret_cast<const f4*>(p);
        v[u][1] = *reinterpret_cast<const f4*>(p + 4);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool on = e + u < D;                       // wave-uniform
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = v[u][j >> 2][j & 3];
          if (on) { s[j] += t; q[j] += t * t; mx[j] = fmaxf(mx[j], t); mn[j] = fminf(mn[j], t); }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float mean = s[j] * invD;
      const float var = fmaxf(q[j] * invD - mean * mean, 0.f);
      const float sd = sqrtf(var + 1e-5f);
      if (D == 0) { mx[j] = 0.f; mn[j] = 0.f; }
      check += mean + sd + mx[j] + mn[j];
      if (agg && real && live && f0 + j < F) {
        float* o = agg + (size_t)(tile * 16 + li) * ld_agg + f0 + j;
        o[0] = mean; o[F] = mx[j]; o[2 * F] = mn[j]; o[3 * F] = sd;
      }
    }
  }
  out[(size_t)tile * 64 + lane] = check;
}

// Variant: ONE pass over the edges, all three strips of a source row loaded back to back (the row's 320 bytes are touched within
// a few instructions instead of in three passes a whole tile apart): 96 running statistics per lane instead of 32.
template <int U>
__global__ __launch_bounds__(256) void k_gather_strip_all(const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ x,
                                                          long ldx, int F, const int* __restrict__ perm, int ntiles, float* __restrict__ out) {
  extern __shared__ float dummy[];
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= ntiles) return;
  const int li = lane & 15, lg = lane >> 4;
  const int first = perm[tile * 16];
  if (first < 0) return;
  int node = perm[tile * 16 + li];
  if (node < 0) node = first;
  const int beg = rowptr[node];
  const int D = __builtin_amdgcn_readfirstlane(rowptr[node + 1] - beg);
  const float invD = D > 0 ? 1.0f / (float)D : 0.f;
  int f0[3];
#pragma unroll
  for (int fb = 0; fb < 3; ++fb) { f0[fb] = fb * 32 + lg * 8; if (f0[fb] >= F) f0[fb] = (F - 1) / 8 * 8; }
  float s[3][8], q[3][8], mx[3][8], mn[3][8];
#pragma unroll
  for (int fb = 0; fb < 3; ++fb)
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[fb][j] = 0.f; q[fb][j] = 0.f; mx[fb][j] = -INFINITY; mn[fb][j] = INFINITY; }
  for (int e = 0; e < D; e += U) {
    int idx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) idx[u] = col[beg + min(e + u, D - 1)];
    f4 v[U][3][2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float* p = x + (size_t)idx[u] * ldx;
#pragma unroll
      for (int fb = 0; fb < 3; ++fb) {
        v[u][fb][0] = *reinterpret_cast<const f4*>(p + f0[fb]);
        v[u][fb][1] = *reinterpret_cast<const f4*>(p + f0[fb] + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool on = e + u < D;
#pragma unroll
      for (int fb = 0; fb < 3; ++fb)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = v[u][fb][j >> 2][j & 3];
          if (on) { s[fb][j] += t; q[fb][j] += t * t; mx[fb][j] = fmaxf(mx[fb][j], t); mn[fb][j] = fminf(mn[fb][j], t); }
        }
    }
  }
  float check = 0.f;
#pragma unroll
  for (int fb = 0; fb < 3; ++fb)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float mean = s[fb][j] * invD;
      check += mean + sqrtf(fmaxf(q[fb][j] * invD - mean * mean, 0.f) + 1e-5f) + mx[fb][j] + mn[fb][j];
    }
  out[(size_t)tile * 64 + lane] = check;
}

extern "C" int gather_strip_all(const int* rowptr, const int* col, const float* x, long ldx, int F, const int* perm, int ntiles, float* out,
                                int unroll, int lds_bytes, void* stream) {
  const dim3 grid((ntiles + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (F > 96) return -3;
  if (unroll == 2) hipLaunchKernelGGL(k_gather_strip_all<2>, grid, block, lds_bytes, st, rowptr, col, x, ldx, F, perm, ntiles, out);
  else if (unroll == 4) hipLaunchKernelGGL(k_gather_strip_all<4>, grid, block, lds_bytes, st, rowptr, col, x, ldx, F, perm, ntiles, out);
  else return -2;
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int gather_strip(const int* rowptr, const int* col, const float* x, long ldx, int F, const int* perm, int ntiles, float* out,
                            float* agg, long ld_agg, int unroll, int lds_bytes, void* stream) {
  const dim3 grid((ntiles + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (unroll == 4) hipLaunchKernelGGL(k_gather_strip<4>, grid, block, lds_bytes, st, rowptr, col, x, ldx, F, perm, ntiles, out, agg, ld_agg);
  else if (unroll == 8) hipLaunchKernelGGL(k_gather_strip<8>, grid, block, lds_bytes, st, rowptr, col, x, ldx, F, perm, ntiles, out, agg, ld_agg);
  else if (unroll == 12) hipLaunchKernelGGL(k_gather_strip<12>, grid, block, lds_bytes, st, rowptr, col, x, ldx, F, perm, ntiles, out, agg, ld_agg);
  else return -2;
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
