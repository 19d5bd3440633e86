/*
 * pna_oracle.c -- TEST INFRASTRUCTURE ONLY (checker and reported CPU baseline; never shipped,
 * never linked or called by pna_amd/).
 *
 * Plain-C restatement of the reference's aggregation step for sizes where running the reference's
 * Python degree-bucket loop is too slow (the 10 M-edge roofline graph).  It follows, per
 * destination node v with in-edge messages m_1..m_D (CSR order = DGL mailbox order):
 *   models/dgl/aggregators.py:6-7    mean = sum / D
 *   models/dgl/aggregators.py:10-15  max / min over the mailbox (NaN-propagating like torch)
 *   models/dgl/aggregators.py:18-26  var = relu(mean(m*m) - mean(m)^2),  std = sqrt(var + 1e-5)
 *   models/dgl/aggregators.py:50-51  sum
 *   models/dgl/scalers.py:12-19      amplification / attenuation factors of the bucket degree D
 *   models/dgl/pna_layer.py:48-49    output order: scaler-major, aggregator-minor
 * and, with edge weights, the dense variant's models/pytorch/pna/aggregators.py:17-73 (adjacency as
 * weight for mean/sum/std/var, as mask for max/min).
 * It is pinned against oracle/torch_oracle.py (itself pinned bit-for-bit to golden vectors produced
 * by the reference's source) in tests/test_oracle_c.py.
 *
 * acc_double = 0: fp32 accumulation in edge order;  1: float64 accumulation ("ground truth" used to
 * attribute error between implementations, SURVEY.md section 7 "hard parts").
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { AGG_MEAN = 0, AGG_SUM = 1, AGG_MAX = 2, AGG_MIN = 3, AGG_STD = 4, AGG_VAR = 5 };

int pna_oracle_segreduce(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col, int32_t V,
                         int32_t F, const float* dst_term, int64_t ld_dst, const float* edge_term, int64_t ld_edge,
                         const float* edge_weight, int32_t n_aggr, const int32_t* aggr, int32_t n_scaler,
                         const float* const* row_scale, float* out, int64_t ldo, int32_t block_stride,
                         int32_t acc_double) {
  int bad = 0;
  for (int32_t a = 0; a < n_aggr; ++a)
    if (aggr[a] < AGG_MEAN || aggr[a] > AGG_VAR) return -1;
#pragma omp parallel
  {
    /* per-thread accumulators, one slot per feature; edges are folded in CSR order (k outer, f inner) so
     * every feature sees exactly the sequential sum s = ((m_1 + m_2) + m_3) + ... */
    float* s32 = (float*)malloc(sizeof(float) * F * 4);
    double* s64 = (double*)malloc(sizeof(double) * F * 2);
    float *q32 = s32 + F, *mx = s32 + 2 * F, *mn = s32 + 3 * F;
    double* q64 = s64 + F;
#pragma omp for schedule(dynamic, 64)
    for (int32_t v = 0; v < V; ++v) {
      const int32_t beg = rowptr[v], end = rowptr[v + 1];
      float w32 = 0.f;
      double w64 = 0.0;
      for (int32_t f = 0; f < F; ++f) { s32[f] = 0.f; q32[f] = 0.f; mx[f] = -INFINITY; mn[f] = INFINITY; s64[f] = 0.0; q64[f] = 0.0; }
      for (int32_t k = beg; k < end; ++k) {
        const int64_t r = col ? col[k] : k;
        const float* xr = x + r * ldx;
        const float* dt = dst_term ? dst_term + (int64_t)v * ld_dst : 0;
        const float* et = edge_term ? edge_term + (int64_t)k * ld_edge : 0;
        const float w = edge_weight ? edge_weight[k] : 1.f;
        if (acc_double) w64 += w; else w32 = w32 + w;
        if (!acc_double && !edge_weight && !dt && !et) {
          /* common case (PNASimpleLayer, pna_layer.py:202: message = raw source features), branch-free */
          for (int32_t f = 0; f < F; ++f) {
            const float m = xr[f];
            s32[f] = s32[f] + m;
            q32[f] = q32[f] + m * m;
            mx[f] = (m > mx[f] || m != m) ? m : mx[f];
            mn[f] = (m < mn[f] || m != m) ? m : mn[f];
          }
          continue;
        }
        for (int32_t f = 0; f < F; ++f) {
          float m = xr[f];
          if (dt) m = m + dt[f];
          if (et) m = m + et[f];
          if (acc_double) { s64[f] += (double)m * w; q64[f] += (double)m * m * w; }
          else if (edge_weight) { s32[f] = s32[f] + m * w; q32[f] = q32[f] + (m * m) * w; }
          else { s32[f] = s32[f] + m; q32[f] = q32[f] + m * m; }
          if (w > 0.f) {
            if (m > mx[f] || (m != m && mx[f] == mx[f])) mx[f] = m;
            if (m < mn[f] || (m != m && mn[f] == mn[f])) mn[f] = m;
          }
        }
      }
      for (int32_t f = 0; f < F; ++f) {
        float mean, var, sum;
        if (acc_double) {
          const double me = s64[f] / w64, t = q64[f] / w64 - me * me;
          mean = (float)me; var = (float)(t < 0.0 ? 0.0 : t); sum = (float)s64[f];
        } else {
          mean = s32[f] / w32;
          const float msq = q32[f] / w32;
          const float t = msq - mean * mean;
          var = t < 0.f ? 0.f : t; sum = s32[f];
        }
        const float sd = acc_double ? (float)sqrt((double)var + 1e-5) : sqrtf(var + 1e-5f);
        for (int32_t sc_i = 0; sc_i < n_scaler; ++sc_i) {
          const float sc = row_scale[sc_i] ? row_scale[sc_i][v] : 1.f;
          for (int32_t a = 0; a < n_aggr; ++a) {
            float val;
            switch (aggr[a]) {
              case AGG_MEAN: val = mean; break;
              case AGG_SUM: val = sum; break;
              case AGG_MAX: val = mx[f]; break;
              case AGG_MIN: val = mn[f]; break;
              case AGG_STD: val = sd; break;
              default: val = var; break;
            }
            out[(int64_t)v * ldo + (int64_t)(sc_i * n_aggr + a) * block_stride + f] = (end > beg) ? val * sc : 0.f;
          }
        }
      }
    }
    free(s32);
    free(s64);
  }
  return bad ? -1 : 0;
}

/* models/dgl/scalers.py:12-19 with the reference's rounding sequence (np.log in float64 -> fp32;
 * amplification = avg.reciprocal() * scalar, attenuation = avg / scalar, both in fp32). */
void pna_oracle_degree_scalers(const int32_t* rowptr, int32_t V, float avg_log, float* amp, float* att) {
  const float inv = 1.0f / avg_log;
  for (int32_t v = 0; v < V; ++v) {
    const int32_t d = rowptr[v + 1] - rowptr[v];
    if (d <= 0) { amp[v] = 0.f; att[v] = 0.f; continue; }
    const float lg = (float)log((double)d + 1.0);
    amp[v] = inv * lg;
    att[v] = avg_log / lg;
  }
}

void pna_oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int pna_oracle_num_threads(void) {
  int n = 1;
#ifdef _OPENMP
#pragma omp parallel
  {
#pragma omp master
    n = omp_get_num_threads();
  }
#endif
  return n;
}
