/* hip_pvq_search.c - pvq_search_rdo_double (src/pvq_encoder.c:121-225) for the searches
 * that stay on the host (with-reference bands: their input is the Householder reflection
 * about a predictor built from the serial reconstruction), four coefficients at a time.
 *
 * Same idea as the device's register-resident search (daala_amd/csrc/pvq_kernels.hpp):
 * the reference's result is defined by a SEQUENTIAL scan whose greedy-phase comparison
 *     (xy + x_j)^2 * best_yy  >  best_xy * (yy + 2 y_j + 1)
 * is done on rounded products and is therefore not guaranteed transitive, so a parallel
 * maximum is not automatically the scan's answer.  Per pulse:
 *   1. every lane scans its residue class with the reference's own comparison, the four
 *      lane winners are merged in index order -> candidate w;
 *   2. w is VERIFIED: it strictly beats every earlier element and no later element
 *      strictly beats it.  That is sufficient for w to be what the sequential scan
 *      returns (whatever the incumbent is when the scan reaches w, w replaces it, and
 *      nothing after w replaces w);
 *   3. if the check fails (possible only at near-ties) the pulse is redone with the literal
 *      scan.
 * The RDO phase compares plain doubles with '>' (a total order), so lane-wise maxima merged
 * by (value, lowest index) are exact.  Every per-element value is produced by the same
 * IEEE operations in the same order as in the reference; sums that the reference
 * accumulates sequentially (xx, the L1 norm, the projection's xy) stay sequential.
 * -ffp-contract=off: no multiply-add is fused. */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "encint.h"
#include "pvq.h"

#include "hip_glue_int.h"

/* od_rsqrt_table (src/pvq_encoder.c:83-91): 6-digit literals up to 16, 1/sqrt(i) above */
static const double RSQRT16[16] = {
  1.000000, 0.707107, 0.577350, 0.500000,
  0.447214, 0.408248, 0.377964, 0.353553,
  0.333333, 0.316228, 0.301511, 0.288675,
  0.277350, 0.267261, 0.258199, 0.250000};

static inline double rsqrt_int(int i) {
  if (i <= 16) return RSQRT16[i - 1];
  return 1./sqrt(i);
}

#define NPAD (MAXN + 8)

double od_hip_pvq_search_lanes(const double *xcoeff, int n, int k, od_coeff *ypulse,
 double g2);

/* Below this many coefficients the reference's scalar scan is faster than four lanes plus
   the verification pass (measured: 7 and 14 coefficients 103/292 ns scalar vs 142/370 ns
   here; 31: 1.07 vs 0.99 us; 127: 9.8 vs 6.4 us). */
#define OD_HIP_SEARCH_MIN_N (24)

double od_hip_pvq_search_host(const double *xcoeff, int n, int k, od_coeff *ypulse,
 double g2) {
  if (n < OD_HIP_SEARCH_MIN_N) return od_ref_pvq_search_rdo_double_cpu(xcoeff, n, k, ypulse, g2);
  return od_hip_pvq_search_lanes(xcoeff, n, k, ypulse, g2);
}

double od_hip_pvq_search_lanes(const double *xcoeff, int n, int k, od_coeff *ypulse,
 double g2) {
  double x[NPAD] __attribute__((aligned(32)));
  double yd[NPAD] __attribute__((aligned(32)));      /* pulses as doubles (exact integers) */
  double a[NPAD] __attribute__((aligned(32)));
  double b[NPAD] __attribute__((aligned(32)));
  double lj[NPAD] __attribute__((aligned(32)));
  int32_t yi[NPAD] __attribute__((aligned(32)));
  double xx;
  double xy;
  double yy;
  double lambda;
  double norm_1;
  double delta_rate;
  int rdo_pulses;
  int i;
  int j;
  int nv;
  xx = xy = yy = 0;
  for (j = 0; j < n; j++) {
    x[j] = fabs(xcoeff[j]);
    xx += x[j]*x[j];
  }
  nv = (n + 3) & ~3;
  for (j = n; j < nv; j++) x[j] = 0;
  norm_1 = 1./sqrt(1e-30 + xx);
  lambda = OD_PVQ_LAMBDA/(1e-30 + g2);
  i = 0;
  if (k > 2) {
    double l1_norm;
    double l1_inv;
    l1_norm = 0;
    for (j = 0; j < n; j++) l1_norm += x[j];
    l1_inv = 1./OD_MAXF(l1_norm, 1e-100);
    for (j = 0; j < n; j++) {
      yi[j] = OD_MAXI(0, (int)floor(k*x[j]*l1_inv));
      xy += x[j]*yi[j];
      yy += yi[j]*yi[j];
      i += yi[j];
    }
  }
  else {
    for (j = 0; j < n; j++) yi[j] = 0;
  }
  for (j = n; j < nv; j++) yi[j] = 0;
  for (j = 0; j < nv; j++) yd[j] = yi[j];
  rdo_pulses = 1 + k/4;
  delta_rate = 3./n;
  /* ---- greedy phase (:166-188) ---- */
  for (; i < k - rdo_pulses; i++) {
    const __m256d vxy = _mm256_set1_pd(xy);
    const __m256d vyy = _mm256_set1_pd(yy);
    const __m256d one = _mm256_set1_pd(1.);
    __m256d ba;
    __m256d bb;
    __m256d bi;
    __m256d idx;
    double la[4] __attribute__((aligned(32)));
    double lb[4] __attribute__((aligned(32)));
    double li[4] __attribute__((aligned(32)));
    double wa;
    double wb;
    int pos;
    int ok;
    int v;
    /* tmp_xy = (xy + x_j)^2, tmp_yy = yy + 2 y_j + 1 for every j; lane-wise scan */
    idx = _mm256_set_pd(3., 2., 1., 0.);
    {
      __m256d t;
      t = _mm256_add_pd(vxy, _mm256_load_pd(x));
      ba = _mm256_mul_pd(t, t);
      bb = _mm256_add_pd(_mm256_add_pd(vyy, _mm256_add_pd(_mm256_load_pd(yd), _mm256_load_pd(yd))), one);
      _mm256_store_pd(a, ba);
      _mm256_store_pd(b, bb);
      bi = idx;
    }
    for (v = 4; v < nv; v += 4) {
      __m256d t;
      __m256d ta;
      __m256d tb;
      __m256d m;
      idx = _mm256_add_pd(idx, _mm256_set1_pd(4.));
      t = _mm256_add_pd(vxy, _mm256_load_pd(x + v));
      ta = _mm256_mul_pd(t, t);
      tb = _mm256_add_pd(_mm256_add_pd(vyy, _mm256_add_pd(_mm256_load_pd(yd + v), _mm256_load_pd(yd + v))), one);
      _mm256_store_pd(a + v, ta);
      _mm256_store_pd(b + v, tb);
      /* tmp_xy*best_yy > best_xy*tmp_yy */
      m = _mm256_cmp_pd(_mm256_mul_pd(ta, bb), _mm256_mul_pd(ba, tb), _CMP_GT_OQ);
      if (v + 4 > n) {
        /* padding lanes never win */
        m = _mm256_and_pd(m, _mm256_cmp_pd(idx, _mm256_set1_pd((double)n), _CMP_LT_OQ));
      }
      ba = _mm256_blendv_pd(ba, ta, m);
      bb = _mm256_blendv_pd(bb, tb, m);
      bi = _mm256_blendv_pd(bi, idx, m);
    }
    _mm256_store_pd(la, ba);
    _mm256_store_pd(lb, bb);
    _mm256_store_pd(li, bi);
    /* merge the lane winners in index order with the reference's comparison */
    {
      int o[4] = {0, 1, 2, 3};
      int s;
      int t;
      for (s = 1; s < 4; s++) {
        for (t = s; t > 0 && li[o[t]] < li[o[t - 1]]; t--) {
          int tmp;
          tmp = o[t];
          o[t] = o[t - 1];
          o[t - 1] = tmp;
        }
      }
      s = 0;
      while (s < 4 && !(li[o[s]] < n)) s++;
      wa = la[o[s]];
      wb = lb[o[s]];
      pos = (int)li[o[s]];
      for (s++; s < 4; s++) {
        if (li[o[s]] < n && la[o[s]]*wb > wa*lb[o[s]]) {
          wa = la[o[s]];
          wb = lb[o[s]];
          pos = (int)li[o[s]];
        }
      }
    }
    /* verification: w beats every earlier element strictly, no later element beats w */
    {
      const __m256d vwa = _mm256_set1_pd(wa);
      const __m256d vwb = _mm256_set1_pd(wb);
      const __m256d vpos = _mm256_set1_pd((double)pos);
      const __m256d vn = _mm256_set1_pd((double)n);
      __m256d bad;
      bad = _mm256_setzero_pd();
      idx = _mm256_set_pd(3., 2., 1., 0.);
      for (v = 0; v < nv; v += 4) {
        __m256d p;
        __m256d q;
        __m256d before;
        __m256d after;
        p = _mm256_mul_pd(_mm256_load_pd(a + v), vwb);      /* tmp_xy_j * yy_w */
        q = _mm256_mul_pd(vwa, _mm256_load_pd(b + v));      /* xy_w * tmp_yy_j */
        before = _mm256_cmp_pd(idx, vpos, _CMP_LT_OQ);
        after = _mm256_and_pd(_mm256_cmp_pd(idx, vpos, _CMP_GT_OQ), _mm256_cmp_pd(idx, vn, _CMP_LT_OQ));
        /* earlier j: need q > p; later j: need !(p > q) */
        bad = _mm256_or_pd(bad, _mm256_andnot_pd(_mm256_cmp_pd(q, p, _CMP_GT_OQ), before));
        bad = _mm256_or_pd(bad, _mm256_and_pd(_mm256_cmp_pd(p, q, _CMP_GT_OQ), after));
        idx = _mm256_add_pd(idx, _mm256_set1_pd(4.));
      }
      ok = _mm256_movemask_pd(bad) == 0;
    }
    if (!ok) {
      double best_xy;
      double best_yy;
      pos = 0;
      best_xy = -10;
      best_yy = 1;
      for (j = 0; j < n; j++) {
        if (j == 0 || a[j]*best_yy > best_xy*b[j]) {
          best_xy = a[j];
          best_yy = b[j];
          pos = j;
        }
      }
    }
    xy = xy + x[pos];
    yy = yy + 2*yi[pos] + 1;
    yi[pos]++;
    yd[pos] = yi[pos];
  }
  /* ---- RDO phase (:193-220) ---- */
  if (i < k) {
    for (j = 0; j < nv; j++) lj[j] = lambda*j*delta_rate;
  }
  for (; i < k; i++) {
    double tb[4];
    const __m256d vxy = _mm256_set1_pd(xy);
    const __m256d vn1 = _mm256_set1_pd(norm_1);
    const __m256d two = _mm256_set1_pd(2.);
    __m256d bc;
    __m256d bi;
    __m256d idx;
    double lc[4] __attribute__((aligned(32)));
    double li[4] __attribute__((aligned(32)));
    double best;
    int pos;
    int v;
    int l;
    for (l = 0; l < 4; l++) tb[l] = rsqrt_int((int)(yy + 2*l + 1));
    bc = _mm256_set1_pd(-INFINITY);
    bi = _mm256_setzero_pd();
    idx = _mm256_set_pd(3., 2., 1., 0.);
    for (v = 0; v < nv; v += 4) {
      double r[4] __attribute__((aligned(32)));
      __m256d c;
      __m256d m;
      for (l = 0; l < 4; l++) {
        int yv;
        yv = yi[v + l];
        r[l] = yv < 4 ? tb[yv] : rsqrt_int((int)(yy + 2*yv + 1));
      }
      /* 2*tmp_xy*norm_1*tmp_yy - lambda*j*delta_rate */
      c = _mm256_mul_pd(two, _mm256_add_pd(vxy, _mm256_load_pd(x + v)));
      c = _mm256_mul_pd(_mm256_mul_pd(c, vn1), _mm256_load_pd(r));
      c = _mm256_sub_pd(c, _mm256_load_pd(lj + v));
      m = _mm256_cmp_pd(c, bc, _CMP_GT_OQ);
      if (v + 4 > n) m = _mm256_and_pd(m, _mm256_cmp_pd(idx, _mm256_set1_pd((double)n), _CMP_LT_OQ));
      bc = _mm256_blendv_pd(bc, c, m);
      bi = _mm256_blendv_pd(bi, idx, m);
      idx = _mm256_add_pd(idx, _mm256_set1_pd(4.));
    }
    _mm256_store_pd(lc, bc);
    _mm256_store_pd(li, bi);
    /* the scan's answer: the largest cost, lowest index among equals (j == 0 always
       enters first, so a lane that saw nothing better than its first element holds it) */
    best = lc[0];
    pos = (int)li[0];
    for (l = 1; l < 4; l++) {
      if (lc[l] > best || (lc[l] == best && li[l] < pos)) {
        best = lc[l];
        pos = (int)li[l];
      }
    }
    xy = xy + x[pos];
    yy = yy + 2*yi[pos] + 1;
    yi[pos]++;
  }
  for (j = 0; j < n; j++) ypulse[j] = xcoeff[j] < 0 ? -yi[j] : yi[j];
  return xy/(1e-100 + sqrt(xx*yy));
}
