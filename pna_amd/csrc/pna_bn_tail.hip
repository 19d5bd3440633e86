// pna_bn_tail.hip -- the tail of PNASimpleLayer's TRAINING forward and its backward as four streaming kernels:
//   out = residual + relu(BatchNorm1d(y))      with BATCH statistics       (models/dgl/pna_layer.py:207-213: batchnorm_h, F.relu,
//                                                                            the residual; nn.BatchNorm1d in training mode)
// Implements pna_bn_tail_{workspace_bytes, fwd_f32, bwd_f32} of include/pna_amd.h.
//
// The library route is five passes forward (statistics, normalise, clamp, add, ...) and five backward over (M, N) tensors of
// 300 MB at the benchmark size (profiles/r03_train_kernel_stats.csv: 0.7 + 0.9 ms of ~25 launches); here: statistics (one read
// of y) + apply (y, residual -> out) forward; reduce (y, grad_out) + apply (y, grad_out -> grad_y) backward.  HBM-bound, no
// reuse: rows are walked in 128-byte column strips by 32 lanes, eight rows per workgroup pass, nothing staged.
//
// Numerics.  Column sums are taken of d = y - K_c with K_c = y[0, c] (one shift per column for the whole tensor: the partial
// sums of all workgroups simply add up), in fp32 over a workgroup's 512 rows and in float64 across workgroups: var = E[d^2] -
// E[d]^2 is then free of the cancellation that the raw moments would have for a column whose mean is large against its spread.
// The ReLU mask of the backward is recomputed from y with the forward's own expression (bn_affine below), not stored.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "pna_amd.h"
#include "pna_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kColLanes = 32;                             // lanes across a row: 128 bytes
constexpr int kRowLanes = kThreads / kColLanes;           // rows in flight per workgroup pass
constexpr int kRowsPerWg = 512;
constexpr int kMaxJ = 4;                                  // column strips per lane: N <= 128

struct TArgs {
  const float* y; const float* go; const float* res; float* out; float* gy;
  long ldy, ld_go, ld_res, ld_out, ld_gy;
  long M; int N; int relu; int nwg;
  const float* gamma; const float* beta;
  float* mean; float* invstd;                             // [N]
  float* rmean; float* rvar; float momentum; float eps;
  float* part;                                            // [2][N][nwg] partial column sums
  float* colc;                                            // [2][N]: backward: mean of g', mean of g' xhat
  float* ggamma; float* gbeta;
};

// z = (y - mean) (gamma invstd) + beta: the centred form (no cancellation between y a and mean a for columns of large mean)
__device__ __forceinline__ float bn_affine(float y, float mean, float a, float b) { return __builtin_fmaf(y - mean, a, b); }

// ---- column sums of (d, d^2) [forward] or (g', g' xhat) [backward] over this workgroup's rows ---------------------------------
// NJ = ceil(N / 32) strips per lane.  Full blocks: the loads of kGroup row steps are issued together (clamped columns: every load
// is unconditional, a lane without a column in the last strip re-reads the row's last element and drops it), then folded in row order.
constexpr int kGroup = 4;

template <bool BWD, int NJ>
__global__ __launch_bounds__(kThreads) void k_bn_colsums(const TArgs a) {
  __shared__ float red[2][kRowLanes][kColLanes * NJ];
  const int tx = threadIdx.x % kColLanes, ty = threadIdx.x / kColLanes;
  const long r0 = (long)blockIdx.x * kRowsPerWg, r1 = min(r0 + kRowsPerWg, a.M);
  float s0[NJ], s1[NJ], ca[NJ], cb[NJ], cm[NJ], ci[NJ];
  int cc[NJ];
  bool ok[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = tx + kColLanes * j;
    cc[j] = min(c, a.N - 1); ok[j] = c < a.N;
    s0[j] = 0.f; s1[j] = 0.f; ca[j] = 0.f; cb[j] = 0.f;
    if constexpr (BWD) {
      cm[j] = a.mean[cc[j]]; ci[j] = a.invstd[cc[j]];
      ca[j] = (a.gamma ? a.gamma[cc[j]] : 1.f) * ci[j];
      cb[j] = a.beta ? a.beta[cc[j]] : 0.f;
    } else {
      cm[j] = a.y[cc[j]]; ci[j] = 0.f;                    // the column's shift K_c = y[0, c]
    }
  }
  auto fold = [&](int j, float v, float g) __attribute__((always_inline)) {
    if constexpr (BWD) {
      if ((a.relu && !(bn_affine(v, cm[j], ca[j], cb[j]) > 0.f)) || !ok[j]) g = 0.f;
      s0[j] = s0[j] + g;
      s1[j] = s1[j] + g * ((v - cm[j]) * ci[j]);
    } else {
      const float d = ok[j] ? v - cm[j] : 0.f;
      s0[j] = s0[j] + d;
      s1[j] = s1[j] + d * d;
    }
  };
  if (r1 - r0 == kRowsPerWg) {
    for (int i = 0; i < kRowsPerWg / kRowLanes; i += kGroup) {
      float v[kGroup][NJ], g[kGroup][NJ];
#pragma unroll
      for (int u = 0; u < kGroup; ++u) {
        const long r = r0 + ty + (long)(i + u) * kRowLanes;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          v[u][j] = a.y[r * a.ldy + cc[j]];
          g[u][j] = BWD ? a.go[r * a.ld_go + cc[j]] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < kGroup; ++u)
#pragma unroll
        for (int j = 0; j < NJ; ++j) fold(j, v[u][j], g[u][j]);
    }
  } else {
    for (long r = r0 + ty; r < r1; r += kRowLanes)
#pragma unroll
      for (int j = 0; j < NJ; ++j) fold(j, a.y[r * a.ldy + cc[j]], BWD ? a.go[r * a.ld_go + cc[j]] : 0.f);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) { red[0][ty][tx + kColLanes * j] = s0[j]; red[1][ty][tx + kColLanes * j] = s1[j]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * a.N; i += kThreads) {
    const int w = i / a.N, c = i - w * a.N;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < kRowLanes; ++k) t = t + red[w][k][c];     // (a fixed order: results do not depend on the launch)
    a.part[((long)w * a.N + c) * a.nwg + blockIdx.x] = t;
  }
}

// ---- one workgroup per column: the partial sums in float64, then the column's constants -----------------------------------------
template <bool BWD>
__global__ __launch_bounds__(kThreads) void k_bn_finalize(const TArgs a) {
  __shared__ double red[2][kThreads];
  const int c = blockIdx.x;
  double t0 = 0.0, t1 = 0.0;
  for (int b = threadIdx.x; b < a.nwg; b += kThreads) {
    t0 += (double)a.part[((long)0 * a.N + c) * a.nwg + b];
    t1 += (double)a.part[((long)1 * a.N + c) * a.nwg + b];
  }
  red[0][threadIdx.x] = t0; red[1][threadIdx.x] = t1;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const double S = red[0][0], Q = red[1][0], M = (double)a.M;
  if constexpr (BWD) {
    if (a.gbeta) a.gbeta[c] = (float)S;
    if (a.ggamma) a.ggamma[c] = (float)Q;
    a.colc[c] = (float)(S / M);
    a.colc[a.N + c] = (float)(Q / M);
  } else {
    const double md = S / M;
    double var = Q / M - md * md;                          // biased, like nn.BatchNorm1d's normalisation
    if (var < 0.0) var = 0.0;
    const double mean = (double)a.y[c] + md;
    a.mean[c] = (float)mean;
    a.invstd[c] = (float)(1.0 / sqrt(var + (double)a.eps));
    if (a.rmean && a.momentum >= 0.f) {                    // running statistics: the UNBIASED variance (torch.nn.functional.batch_norm)
      const double m = (double)a.momentum;
      a.rmean[c] = (float)((1.0 - m) * (double)a.rmean[c] + m * mean);
      a.rvar[c] = (float)((1.0 - m) * (double)a.rvar[c] + m * var * (M / (M - 1.0)));
    }
  }
}

// ---- forward: out = residual + act((y - mean) a + beta);  backward: grad_y = a (g' - mean(g') - xhat mean(g' xhat)) ----------
template <bool BWD, int NJ>
__global__ __launch_bounds__(kThreads) void k_bn_apply(const TArgs a) {
  const int tx = threadIdx.x % kColLanes, ty = threadIdx.x / kColLanes;
  const long r0 = (long)blockIdx.x * kRowsPerWg, r1 = min(r0 + kRowsPerWg, a.M);
  float ca[NJ], cb[NJ], cm[NJ], ci[NJ], c1[NJ], c2[NJ];
  int cc[NJ];
  bool ok[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = tx + kColLanes * j;
    cc[j] = min(c, a.N - 1); ok[j] = c < a.N;
    cm[j] = a.mean[cc[j]]; ci[j] = a.invstd[cc[j]];
    ca[j] = (a.gamma ? a.gamma[cc[j]] : 1.f) * ci[j];
    cb[j] = a.beta ? a.beta[cc[j]] : 0.f;
    c1[j] = BWD ? a.colc[cc[j]] : 0.f; c2[j] = BWD ? a.colc[a.N + cc[j]] : 0.f;
  }
  const bool has_res = a.res != nullptr;
  // w: the residual (forward) or grad_out (backward) at the element
  auto finish = [&](long r, int j, float v, float w) __attribute__((always_inline)) {
    const float z = bn_affine(v, cm[j], ca[j], cb[j]);
    if constexpr (BWD) {
      const float g = (a.relu && !(z > 0.f)) ? 0.f : w;
      if (ok[j]) a.gy[r * a.ld_gy + cc[j]] = ca[j] * ((g - c1[j]) - ((v - cm[j]) * ci[j]) * c2[j]);
    } else {
      float o = a.relu ? (z > 0.f ? z : (z != z ? z : 0.f)) : z;      // (NaN stays NaN, like F.relu)
      if (has_res) o = w + o;
      if (ok[j]) a.out[r * a.ld_out + cc[j]] = o;
    }
  };
  const float* const wsrc = BWD ? a.go : (has_res ? a.res : a.y);      // (no residual: a harmless second read of y)
  const long ldw = BWD ? a.ld_go : (has_res ? a.ld_res : a.ldy);
  if (r1 - r0 == kRowsPerWg) {
    for (int i = 0; i < kRowsPerWg / kRowLanes; i += kGroup) {
      float v[kGroup][NJ], w[kGroup][NJ];
#pragma unroll
      for (int u = 0; u < kGroup; ++u) {
        const long r = r0 + ty + (long)(i + u) * kRowLanes;
#pragma unroll
        for (int j = 0; j < NJ; ++j) { v[u][j] = a.y[r * a.ldy + cc[j]]; w[u][j] = wsrc[r * ldw + cc[j]]; }
      }
#pragma unroll
      for (int u = 0; u < kGroup; ++u)
#pragma unroll
        for (int j = 0; j < NJ; ++j) finish(r0 + ty + (long)(i + u) * kRowLanes, j, v[u][j], w[u][j]);
    }
  } else {
    for (long r = r0 + ty; r < r1; r += kRowLanes)
#pragma unroll
      for (int j = 0; j < NJ; ++j) finish(r, j, a.y[r * a.ldy + cc[j]], wsrc[r * ldw + cc[j]]);
  }
}

int fill(const pna_bn_tail_args* p, TArgs& a, bool bwd, const char* who) {
  memset(&a, 0, sizeof(a));
  if (!p) return pna_set_error(PNA_E_INVALID, who);
  if (int rc_ss = pna_check_struct_size(bwd ? "pna_bn_tail_bwd_f32" : "pna_bn_tail_fwd_f32", p->struct_size, sizeof(*p))) return rc_ss;
  if (p->M < 2 || p->N <= 0 || p->N > kColLanes * kMaxJ || !p->y || p->ldy < p->N || !p->save_mean || !p->save_invstd || !p->workspace)
    return pna_set_error(PNA_E_INVALID, who);
  if (p->workspace_bytes < pna_bn_tail_workspace_bytes(p->M, p->N)) return pna_set_error(PNA_E_INVALID, who);
  if ((p->running_mean == nullptr) != (p->running_var == nullptr)) return pna_set_error(PNA_E_INVALID, who);
  if (!bwd && (!p->out || p->ld_out < p->N || (p->residual && p->ld_res < p->N))) return pna_set_error(PNA_E_INVALID, who);
  if (bwd && (!p->grad_out || p->ld_go < p->N || !p->grad_y || p->ld_gy < p->N)) return pna_set_error(PNA_E_INVALID, who);
  const long nwg = (p->M + kRowsPerWg - 1) / kRowsPerWg;
  if (nwg > 0x7fffffffL) return pna_set_error(PNA_E_INVALID, who);
  a.y = p->y; a.ldy = p->ldy; a.M = p->M; a.N = p->N; a.relu = p->relu != 0; a.nwg = (int)nwg;
  a.gamma = p->gamma; a.beta = p->beta; a.mean = p->save_mean; a.invstd = p->save_invstd;
  a.rmean = p->running_mean; a.rvar = p->running_var; a.momentum = p->momentum; a.eps = p->eps;
  a.res = p->residual; a.ld_res = p->ld_res; a.out = p->out; a.ld_out = p->ld_out;
  a.go = p->grad_out; a.ld_go = p->ld_go; a.gy = p->grad_y; a.ld_gy = p->ld_gy; a.ggamma = p->grad_gamma; a.gbeta = p->grad_beta;
  a.part = (float*)p->workspace;
  a.colc = a.part + 2 * (long)p->N * nwg;
  return PNA_OK;
}

template <bool BWD, int NJ>
void launch3(const TArgs& a, hipStream_t st) {
  hipLaunchKernelGGL((k_bn_colsums<BWD, NJ>), dim3((unsigned)a.nwg), dim3(kThreads), 0, st, a);
  hipLaunchKernelGGL((k_bn_finalize<BWD>), dim3((unsigned)a.N), dim3(kThreads), 0, st, a);
  hipLaunchKernelGGL((k_bn_apply<BWD, NJ>), dim3((unsigned)a.nwg), dim3(kThreads), 0, st, a);
}

template <bool BWD>
int run(const TArgs& a, hipStream_t st) {
  switch ((a.N + kColLanes - 1) / kColLanes) {
    case 1: launch3<BWD, 1>(a, st); break;
    case 2: launch3<BWD, 2>(a, st); break;
    case 3: launch3<BWD, 3>(a, st); break;
    default: launch3<BWD, 4>(a, st); break;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}

}  // namespace

extern "C" int64_t pna_bn_tail_workspace_bytes(int64_t M, int32_t N) {
  if (M < 1 || N <= 0) return 0;
  const int64_t nwg = (M + kRowsPerWg - 1) / kRowsPerWg;
  return (2 * (int64_t)N * nwg + 2 * (int64_t)N) * (int64_t)sizeof(float);
}

extern "C" int pna_bn_tail_fwd_f32(const pna_bn_tail_args* p, pna_stream_t stream) {
  TArgs a;
  const int rc = fill(p, a, false, "pna_bn_tail_fwd_f32: needs M >= 2 rows, 1 <= N <= 128, y / out (ld >= N), save_mean / save_invstd [N], running_mean and "
                                   "running_var together, a workspace of pna_bn_tail_workspace_bytes(M, N)");
  if (rc != PNA_OK) return rc;
  return run<false>(a, (hipStream_t)stream);
}

extern "C" int pna_bn_tail_bwd_f32(const pna_bn_tail_args* p, pna_stream_t stream) {
  TArgs a;
  const int rc = fill(p, a, true, "pna_bn_tail_bwd_f32: needs M >= 2 rows, 1 <= N <= 128, y / grad_out / grad_y (ld >= N), the forward's save_mean / "
                                  "save_invstd [N], a workspace of pna_bn_tail_workspace_bytes(M, N)");
  if (rc != PNA_OK) return rc;
  return run<true>(a, (hipStream_t)stream);
}
