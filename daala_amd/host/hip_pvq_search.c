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

/* Below this many coefficients the reference's scalar scan is faster than four lanes plus
   the verification pass (measured: 7 and 14 coefficients 103/292 ns scalar vs 142/370 ns
   with lanes; 31: 1.07 vs 0.99 us; 127: 9.8 vs 6.4 us). */
#define OD_HIP_SEARCH_MIN_N (24)
#ifndef OD_HIP_RDO_LANES_MIN_N
#define OD_HIP_RDO_LANES_MIN_N (24)
#endif

/* One vector, many searches.  pvq_theta (src/pvq_encoder.c:399-481) searches the SAME input
   once per (gain, theta) candidate, changing only K and g2.  Of what pvq_search_rdo_double
   computes, |x|, xx, norm_1 and the L1 norm do not depend on either; the projection and the
   greedy pulses (:147-188) depend on K alone; only the last 1 + K/4 pulses (:193-220) see
   g2.  The context keeps the first group per vector and the second per distinct K (two in
   five of a frame's with-reference searches repeat a K of their band), so a candidate costs
   its RDO pulses only - every value still comes out of the reference's operations in the
   reference's order. */
void od_hip_search_begin(od_hip_search *S, const double *xcoeff, int n) {
  /* bit 0: lane scans in the greedy phase (pays from 24 coefficients: the verification pass
     is overhead), bit 1: in the RDO phase (a plain maximum: pays from OD_HIP_RDO_LANES_MIN_N) */
  od_hip_search_begin_ex(S, xcoeff, n, (n >= OD_HIP_SEARCH_MIN_N ? 1 : 0) | (n >= OD_HIP_RDO_LANES_MIN_N ? 2 : 0));
}

void od_hip_search_begin_ex(od_hip_search *S, const double *xcoeff, int n, int lanes) {
  double xx;
  int j;
  S->xcoeff = xcoeff;
  S->n = n;
  S->nv = (n + 3) & ~3;
  xx = 0;
  for (j = 0; j < n; j++) {
    S->x[j] = fabs(xcoeff[j]);
    xx += S->x[j]*S->x[j];
  }
  for (j = n; j < S->nv; j++) S->x[j] = 0;
  S->xx = xx;
  S->norm_1 = 1./sqrt(1e-30 + xx);
  S->have_l1 = 0;
  S->nk = 0;
  S->lanes = lanes;
}

/* ---- greedy phase (:166-188): pulses i .. to-1 ---- */
static void greedy_scalar(const od_hip_search *S, int32_t *yi, double *pxy, double *pyy,
 int i, int to) {
  const double *x;
  double xy;
  double yy;
  int n;
  int j;
  x = S->x;
  n = S->n;
  xy = *pxy;
  yy = *pyy;
  for (; i < to; i++) {
    int pos;
    double best_xy;
    double best_yy;
    pos = 0;
    best_xy = -10;
    best_yy = 1;
    for (j = 0; j < n; j++) {
      double tmp_xy;
      double tmp_yy;
      tmp_xy = xy + x[j];
      tmp_yy = yy + 2*yi[j] + 1;
      tmp_xy *= tmp_xy;
      if (j == 0 || tmp_xy*best_yy > best_xy*tmp_yy) {
        best_xy = tmp_xy;
        best_yy = tmp_yy;
        pos = j;
      }
    }
    xy = xy + x[pos];
    yy = yy + 2*yi[pos] + 1;
    yi[pos]++;
  }
  *pxy = xy;
  *pyy = yy;
}

static void greedy_lanes(const od_hip_search *S, int32_t *yi, double *pxy, double *pyy,
 int i, int to) {
  double yd[NPAD] __attribute__((aligned(32)));      /* pulses as doubles (exact integers) */
  double a[NPAD] __attribute__((aligned(32)));
  double b[NPAD] __attribute__((aligned(32)));
  const double *x;
  double xy;
  double yy;
  int n;
  int nv;
  int j;
  x = S->x;
  n = S->n;
  nv = S->nv;
  xy = *pxy;
  yy = *pyy;
  for (j = 0; j < nv; j++) yd[j] = yi[j];
  for (; i < to; i++) {
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
  *pxy = xy;
  *pyy = yy;
}

/* ---- RDO phase (:193-220): pulses i .. k-1 ---- */
static void rdo_scalar(const od_hip_search *S, int32_t *yi, double *pxy, double *pyy,
 int i, int k, double lambda) {
  const double *x;
  double xy;
  double yy;
  double norm_1;
  double delta_rate;
  int n;
  int j;
  x = S->x;
  n = S->n;
  xy = *pxy;
  yy = *pyy;
  norm_1 = S->norm_1;
  delta_rate = 3./n;
  for (; i < k; i++) {
    double tb[4];
    int pos;
    double best_cost;
    int l;
    pos = 0;
    best_cost = -1e5;
    for (l = 0; l < 4; l++) tb[l] = rsqrt_int((int)(yy + 2*l + 1));
    for (j = 0; j < n; j++) {
      double tmp_xy;
      double tmp_yy;
      tmp_xy = xy + x[j];
      tmp_yy = yi[j] < 4 ? tb[yi[j]] : rsqrt_int((int)(yy + 2*yi[j] + 1));
      tmp_xy = 2*tmp_xy*norm_1*tmp_yy - lambda*j*delta_rate;
      if (j == 0 || tmp_xy > best_cost) {
        best_cost = tmp_xy;
        pos = j;
      }
    }
    xy = xy + x[pos];
    yy = yy + 2*yi[pos] + 1;
    yi[pos]++;
  }
  *pxy = xy;
  *pyy = yy;
}

static void rdo_lanes(const od_hip_search *S, int32_t *yi, double *pxy, double *pyy,
 int i, int k, double lambda) {
  double lj[NPAD] __attribute__((aligned(32)));
  const double *x;
  double xy;
  double yy;
  double norm_1;
  double delta_rate;
  int n;
  int nv;
  int j;
  x = S->x;
  n = S->n;
  nv = S->nv;
  xy = *pxy;
  yy = *pyy;
  norm_1 = S->norm_1;
  delta_rate = 3./n;
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
  *pxy = xy;
  *pyy = yy;
}

double od_hip_search_run(od_hip_search *S, int k, od_coeff *ypulse, double g2) {
  int32_t yi[NPAD] __attribute__((aligned(32)));
  const double *x;
  double xy;
  double yy;
  double lambda;
  int rdo_pulses;
  int n;
  int nv;
  int i;
  int j;
  int e;
  x = S->x;
  n = S->n;
  nv = S->nv;
  lambda = OD_PVQ_LAMBDA/(1e-30 + g2);
  rdo_pulses = 1 + k/4;
  for (e = 0; e < S->nk && S->ka[e].k != k; e++);
  if (e < S->nk) {
    memcpy(yi, S->ky[e], sizeof(int32_t)*nv);
    xy = S->ka[e].xy;
    yy = S->ka[e].yy;
    i = S->ka[e].placed;
  }
  else {
    xy = yy = 0;
    i = 0;
    if (k > 2) {
      double l1_inv;
      if (!S->have_l1) {
        double l1_norm;
        l1_norm = 0;
        for (j = 0; j < n; j++) l1_norm += x[j];
        S->l1_inv = 1./OD_MAXF(l1_norm, 1e-100);
        S->have_l1 = 1;
      }
      l1_inv = S->l1_inv;
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
    if (i < k - rdo_pulses) {
      if (!(S->lanes & 1)) greedy_scalar(S, yi, &xy, &yy, i, k - rdo_pulses);
      else greedy_lanes(S, yi, &xy, &yy, i, k - rdo_pulses);
      i = k - rdo_pulses;
    }
    if (S->nk < OD_HIP_SEARCH_KCACHE) {
      e = S->nk++;
      S->ka[e].k = k;
      S->ka[e].placed = i;
      S->ka[e].xy = xy;
      S->ka[e].yy = yy;
      memcpy(S->ky[e], yi, sizeof(int32_t)*nv);
    }
  }
  if (i < k) {
    if (!(S->lanes & 2)) rdo_scalar(S, yi, &xy, &yy, i, k, lambda);
    else rdo_lanes(S, yi, &xy, &yy, i, k, lambda);
  }
  for (j = 0; j < n; j++) ypulse[j] = S->xcoeff[j] < 0 ? -yi[j] : yi[j];
  return xy/(1e-100 + sqrt(S->xx*yy));
}

/* one search of one vector */
double od_hip_pvq_search_host(const double *xcoeff, int n, int k, od_coeff *ypulse,
 double g2) {
  od_hip_search S;
  od_hip_search_begin(&S, xcoeff, n);
  return od_hip_search_run(&S, k, ypulse, g2);
}

/* test entry: ncand searches of one vector through one context, scalar or lane scans */
void od_hip_pvq_search_multi(const double *xcoeff, int n, int lanes, int ncand, const int *k,
 const double *g2, od_coeff *y, double *cos_dist) {
  od_hip_search S;
  int c;
  od_hip_search_begin_ex(&S, xcoeff, n, lanes);
  for (c = 0; c < ncand; c++) cos_dist[c] = od_hip_search_run(&S, k[c], y + (size_t)c*n, g2[c]);
}
