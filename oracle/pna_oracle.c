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
#pragma omp parallel for schedule(dynamic, 64)
  for (int32_t v = 0; v < V; ++v) {
    const int32_t beg = rowptr[v], end = rowptr[v + 1];
    for (int32_t f = 0; f < F; ++f) {
      float s32 = 0.f, q32 = 0.f, w32 = 0.f;
      double s64 = 0.0, q64 = 0.0, w64 = 0.0;
      float mx = -INFINITY, mn = INFINITY;
      for (int32_t k = beg; k < end; ++k) {
        const int64_t r = col ? col[k] : k;
        float m = x[r * ldx + f];
        if (dst_term) m = m + dst_term[(int64_t)v * ld_dst + f];
        if (edge_term) m = m + edge_term[(int64_t)k * ld_edge + f];
        const float w = edge_weight ? edge_weight[k] : 1.f;
        if (acc_double) {
          s64 += (double)m * w; q64 += (double)m * m * w; w64 += w;
        } else if (edge_weight) {
          s32 = s32 + m * w; q32 = q32 + (m * m) * w; w32 = w32 + w;
        } else {
          s32 = s32 + m; q32 = q32 + m * m; w32 = w32 + 1.f;
        }
        if (w > 0.f) {
          if (m > mx || (m != m && mx == mx)) mx = m;
          if (m < mn || (m != m && mn == mn)) mn = m;
        }
      }
      float mean, var, sum;
      if (acc_double) {
        const double me = s64 / w64, t = q64 / w64 - me * me;
        mean = (float)me; var = (float)(t < 0.0 ? 0.0 : t); sum = (float)s64;
      } else {
        mean = s32 / w32;
        const float msq = q32 / w32;
        const float t = msq - mean * mean;
        var = t < 0.f ? 0.f : t; sum = s32;
      }
      const float sd = acc_double ? (float)sqrt((double)var + 1e-5) : sqrtf(var + 1e-5f);
      for (int32_t s = 0; s < n_scaler; ++s) {
        const float sc = row_scale[s] ? row_scale[s][v] : 1.f;
        for (int32_t a = 0; a < n_aggr; ++a) {
          float val;
          switch (aggr[a]) {
            case AGG_MEAN: val = mean; break;
            case AGG_SUM: val = sum; break;
            case AGG_MAX: val = mx; break;
            case AGG_MIN: val = mn; break;
            case AGG_STD: val = sd; break;
            case AGG_VAR: val = var; break;
            default: val = 0.f; bad = 1; break;
          }
          out[(int64_t)v * ldo + (int64_t)(s * n_aggr + a) * block_stride + f] = (end > beg) ? val * sc : 0.f;
        }
      }
    }
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
