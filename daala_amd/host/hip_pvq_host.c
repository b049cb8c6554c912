/* hip_pvq_host.c - host side of SURVEY rows A16/A19: the band decision pvq_theta() as a
 * CONSUMER of the device feeds - keyframe luma: the no-reference candidates (section 4b of
 * include/daala_hip.h); inter frames: the complete candidate list of every band (4d) - plus a
 * rate-only form of od_pvq_rate() and one search context per band for what stays here.
 * od_pvq_encode() itself is the reference's function, compiled from its source by the build
 * recipe with its pvq_theta / checkpoint call sites bound here (Makefile: pvq_encoder.o).
 *
 * What the reference does per band (src/pvq_encoder.c:311-511) and what happens here:
 *   gain of x (:360, n multiply-adds + pow)   -> read from the feed (g exact from the
 *                                                 device; cg companded from it by this
 *                                                 process's libm between the device passes)
 *   no-reference search (:452-481)            -> candidates (qg, k, pulses, cos_dist) read
 *                                                 in place from the feed, no copy
 *   od_pvq_rate (:248-284): trial range-coding into a freshly malloc'ed encoder with a
 *   copy of the codeword context              -> hip_rc below: the same integer recurrence
 *                                                 on (rng, bit count) only - the number
 *                                                 od_ec_enc_tell_frac() returns depends on
 *                                                 nothing else (src/entcode.c:65-91)
 *   with-reference theta search (:399-448)    -> same arithmetic on the host (on keyframes its
 *                                                 input depends on the serial reconstruction);
 *                                                 what the candidates of one vector share is
 *                                                 computed once (hip_pvq_search.c), equal
 *                                                 codewords are priced once
 * and per block (src/pvq_encoder.c:645-815): the 19.7 KB od_encode_checkpoint() of the
 * whole adaptation context becomes a copy of the ~2 KB this function can modify.
 *
 * Every floating-point expression keeps the reference's operand order (gcc does not
 * re-associate without -ffast-math, contraction is off), libm calls are the process's own
 * glibc: results are bit-identical by construction, and `check` mode runs the reference's
 * od_pvq_encode on a copy of the state after every block and compares everything
 * (coefficients, return value, range-coder state and bytes, adaptation context). */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "encint.h"
#include "pvq.h"
#include "pvq_encoder.h"
#include "partition.h"
#include "entcode.h"
#include "generic_code.h"

#include "hip_glue_int.h"

#define T od_hipenc_tls

int od_vector_is_null(const od_coeff *x, int len);     /* src/pvq_encoder.c:242 */

/* ------------------------------------------------------------------------ */
/* Rate-only Laplace coder: the reference's own src/laplace_encoder.c, compiled a second time
   by the build recipe (Makefile: build/obj/laplace_rate.o, -include laplace_rate_head.h) with
   its range-coder calls bound to the recurrence of hip_rc.h - od_ec_encode_cdf_unscaled /
   od_ec_encode_cdf_q15 update (rng, bit count) only, od_ec_enc_bits adds its bit count - so
   that od_ec_enc_tell_frac() of a trial encoder is reproduced exactly without a buffer.  No
   text of that file lives here (rounds 2-3 carried an edited restatement of it). */
#include "hip_rc.h"
void od_hip_rate_laplace_vector(od_ec_enc *enc, const od_coeff *y, int n, int k, int32_t *curr,
 const int32_t *means);

/* The codeword's share of od_pvq_rate (src/pvq_encoder.c:257-276): trial coding of y into
   a fresh range coder.  It depends on (y, k, n, noref, bs) and the adaptation state only, so
   within one band - where the state does not move - equal codewords have equal rates. */
static double pvq_codeword_rate_untimed(const od_adapt_ctx *adapt, const od_coeff *y0, int k, int n,
 int noref, int bs);

/* What of a codeword's pricing does not depend on the adaptation state (HIPENC_TIME=1 runs it
   beside every pricing and times both: the measured answer to "could the device pre-digest the
   candidates' rate").  The symbols a codeword turns into - magnitudes, the running pulse
   budget, where the delta coder takes over - are state free; every symbol's alphabet, shift
   and CDF row (ex = f(exp_q8, kn, i), decay = f(ex)) and the rng recurrence are not. */
static volatile int rate_skeleton_sink;
static void rate_skeleton(const od_coeff *y, int n, int k) {
  int kn;
  int i;
  int acc;
  kn = k;
  acc = 0;
  for (i = 0; i < n && kn > 0; i++) {
    int x;
    x = abs(y[i]);
    acc += (x != 0) + (kn <= 1);
    kn -= x;
  }
  rate_skeleton_sink = acc;
}

static double pvq_codeword_rate(const od_adapt_ctx *adapt, const od_coeff *y0, int k, int n,
 int noref, int bs) {
  double t0;
  double t1;
  double r;
  if (!T.time_cpu) return pvq_codeword_rate_untimed(adapt, y0, k, n, noref, bs);
  T.st.rate_calls++;
  /* one call in OD_HIPENC_TIME_SAMPLE is timed and stands for all of them: a timer read costs
     about as much as a pricing on these guests, timing every call triples the step */
  if (T.st.rate_calls % OD_HIPENC_TIME_SAMPLE) return pvq_codeword_rate_untimed(adapt, y0, k, n, noref, bs);
  t0 = od_hipenc_fine_now();
  r = pvq_codeword_rate_untimed(adapt, y0, k, n, noref, bs);
  t1 = od_hipenc_fine_now();
  rate_skeleton(y0, n - !noref, k);
  T.st.rate_s += OD_HIPENC_TIME_SAMPLE*(t1 - t0);
  T.st.rate_state_free_s += OD_HIPENC_TIME_SAMPLE*(od_hipenc_fine_now() - t1);
  return r;
}

static double pvq_codeword_rate_untimed(const od_adapt_ctx *adapt, const od_coeff *y0, int k, int n,
 int noref, int bs) {
  const od_pvq_codeword_ctx *cd;
  hip_rc c;
  c.rng = 0x8000;
  c.nbits = 1;
  cd = &adapt->pvq.pvq_codeword_ctx;
  /* od_encode_pvq_codeword (src/pvq_encoder.c:41-80) */
  if (k == 1 && n < 16) {
    int cdf_id;
    int i;
    int pos;
    int nn;
    cdf_id = 2*(n == 15) + !noref;
    nn = n - !noref;
    pos = 32;
    for (i = 0; i < nn; i++) {
      if (y0[i]) {
        pos = i;
        break;
      }
    }
    rc_cdf_unscaled(&c, pos, cd->pvq_k1_cdf[cdf_id], 0, nn);
    c.nbits += 1;
  }
  else {
    int32_t curr[OD_NSB_ADAPT_CTXS];       /* the adaptation outputs of the trial coding: not kept */
    od_hip_rate_laplace_vector((od_ec_enc *)&c, y0, n - !noref, k, curr, cd->pvq_adapt + 4*(2*bs + noref));
  }
  /* (od_ec_enc_tell_frac(&ec) - tell)/8. with tell = od_ec_tell_frac(1, 0x8000) = 8 */
  return (uint32_t)(od_ec_tell_frac(c.nbits, c.rng) - 8)/8.;
}

static double pvq_rate_with_codeword(double rate, int qg, int icgr, int theta, int ts,
 int is_keyframe, int pli);

/* od_pvq_rate (src/pvq_encoder.c:248-284) */
double od_hip_pvq_rate(int qg, int icgr, int theta, int ts, const od_adapt_ctx *adapt,
 const od_coeff *y0, int k, int n, int is_keyframe, int pli, int bs) {
  double rate;
  if (k > 0) rate = pvq_codeword_rate(adapt, y0, k, n, theta == -1, bs);
  else rate = 0;
  return pvq_rate_with_codeword(rate, qg, icgr, theta, ts, is_keyframe, pli);
}

/* the gain/theta terms (:277-283) added to a codeword rate */
static double pvq_rate_with_codeword(double rate, int qg, int icgr, int theta, int ts,
 int is_keyframe, int pli) {
  if (qg > 0 && theta >= 0) {
    /* .9*OD_LOG2(ts): ts is a small integer, the libm value is cached per thread (same
       call, same bits) */
    static __thread double log2_ts[64];
    static __thread unsigned char have_ts[64];
    double l2;
    if (ts >= 0 && ts < 64) {
      if (!have_ts[ts]) {
        log2_ts[ts] = OD_LOG2(ts);
        have_ts[ts] = 1;
      }
      l2 = log2_ts[ts];
    }
    else l2 = OD_LOG2(ts);
    rate += .9*l2;
    if (is_keyframe && pli == 0) rate += 6;
    if (qg == icgr) rate -= .5;
  }
  return rate;
}

/* ------------------------------------------------------------------------ */
/* The band decision.  One band of one block = a set of CANDIDATES - the null / skip case, the
   with-reference (gain i, angle j) pairs of src/pvq_encoder.c:406-417, the no-reference gains
   of :457 - each with a distortion and a rate; the cheapest wins (:435 '<', :469 '<=').  What
   differs between the sources of a band is only where a candidate's codeword and cosine
   distance come from:
     keyframe feed (4b)   no-reference candidates of keyframe luma
     P-frame feed (4d)    every candidate of every band of an inter frame
     this file            a search on the host (hip_pvq_search.c) for whatever no feed covers
   so the function below is written around a small candidate record and one pricing step,
   not around the reference's control flow.  Arithmetic that decides anything keeps the
   reference's operand order (cited line by line); libm calls are this process's. */

typedef struct band_src {
  const od_hip_feed_level *L;     /* keyframe feed of the block's level, or NULL */
  const od_hip_pfeed_level *P;    /* P-frame feed of the block's (plane, level), or NULL */
  size_t rec;                     /* band*nblk + block */
  size_t nrec;
  int band;
  int blk;
} band_src;

typedef struct band_best {
  double cost;
  double dist;
  double qtheta;
  int qg;
  int k;
  int itheta;
  int max_theta;
  int noref;
  const od_coeff *y;              /* winning codeword, or NULL: none (gain 0) */
} band_best;

/* widen a run of the feeds' 16-bit pulses */
static inline void widen(od_coeff *dst, const int16_t *src, int n) {
  int j;
  for (j = 0; j < n; j++) dst[j] = src[j];
}

/* src/pvq_encoder.c:236-240 */
static inline int interleave_gain(int x, int ref) {
  if (x < ref) return -2*(x - ref) - 1;
  if (x < 2*ref) return 2*(x - ref);
  return x - 1;
}

/* the codeword cache of one band: equal codewords have equal rates while the adaptation
   state stands still (it does within a band) */
typedef struct cw_cache {
  od_coeff y[OD_HIP_SEARCH_KCACHE][128];
  double rate[OD_HIP_SEARCH_KCACHE];
  int k[OD_HIP_SEARCH_KCACHE];
  int n;
} cw_cache;

static double cached_codeword_rate(cw_cache *C, const od_adapt_ctx *adapt, const od_coeff *y, int k,
 int n, int bs) {
  double cw;
  int e;
  if (k <= 0) return 0;
  for (e = 0; e < C->n; e++) {
    if (C->k[e] == k && memcmp(C->y[e], y, sizeof(od_coeff)*(n - 1)) == 0) return C->rate[e];
  }
  cw = pvq_codeword_rate(adapt, y, k, n, 0, bs);
  if (C->n < OD_HIP_SEARCH_KCACHE && n - 1 <= 128) {
    C->k[C->n] = k;
    C->rate[C->n] = cw;
    memcpy(C->y[C->n], y, sizeof(od_coeff)*(n - 1));
    C->n++;
  }
  return cw;
}

/* x = the input reflected by the reference's Householder vector with the reference's axis
   dropped (src/pvq_encoder.c:402-404), r = that vector: what the with-reference searches and
   the synthesis work on */
static void reflect(double *x, double *r, const od_coeff *x0, const od_coeff *r0, const int16_t *qm,
 int n, double gr, int *m, int *s) {
  int i;
  for (i = 0; i < n; i++) {
    x[i] = x0[i]*qm[i]*OD_QM_SCALE_1;
    r[i] = r0[i]*qm[i]*OD_QM_SCALE_1;
  }
  *m = od_compute_householder(r, n, gr, s);
  od_apply_householder(x, r, n);
  for (i = *m; i < n - 1; i++) x[i] = x[i + 1];
}

/* timing of the host searches (HIPENC_TIME=1) */
static inline double timed_search(od_hip_search *sc, int k, od_coeff *y, double g2, int cls) {
  double t0;
  double r;
  if (!T.time_cpu || ++T.search_tick % OD_HIPENC_TIME_SAMPLE) return od_hip_search_run(sc, k, y, g2);
  t0 = od_hipenc_fine_now();
  r = od_hip_search_run(sc, k, y, g2);
  t0 = OD_HIPENC_TIME_SAMPLE*(od_hipenc_fine_now() - t0);
  T.st.search_cpu_s += t0;
  T.st.search_class_s[cls] += t0;
  return r;
}

/* pvq_theta (src/pvq_encoder.c:311-511), same interface; the feed context of the block being
   coded comes from the calling thread's state (set by od_pvq_encode below). */
int od_ref_pvq_theta(od_coeff *out, od_coeff *x0, od_coeff *r0, int n, int q0, od_coeff *y,
 int *itheta, int *max_theta, int *vk, double beta, double *skip_diff, int robust,
 int is_keyframe, int pli, const od_adapt_ctx *adapt, int bs, const int16_t *qm,
 const int16_t *qm_inv);

int pvq_theta(od_coeff *out, od_coeff *x0, od_coeff *r0, int n, int q0, od_coeff *y,
 int *itheta, int *max_theta, int *vk, double beta, double *skip_diff, int robust,
 int is_keyframe, int pli, const od_adapt_ctx *adapt, int bs, const int16_t *qm,
 const int16_t *qm_inv) {
  const double lambda = OD_PVQ_LAMBDA;
  const double gain_weight = 1.4;
  band_src S;
  band_best B;
  od_coeff y_tmp[MAXN];
  od_coeff y_keep[MAXN];          /* the incumbent's codeword when it came from y_tmp */
  double x[MAXN];
  double r[MAXN];
  double g;
  double gr;
  double cg;
  double cgr;
  double corr;
  double theta;
  double gain_offset;
  double skip_dist;
  int icgr;
  int theta_search;
  int noref_search;
  int have_xr;                    /* x[] / r[] hold the scaled input and reference */
  int r_null;
  int cfl;
  int nodesync;
  int i;
  int m;
  int s;
  if (!T.host_pvq) {
    return od_ref_pvq_theta(out, x0, r0, n, q0, y, itheta, max_theta, vk, beta, skip_diff, robust,
     is_keyframe, pli, adapt, bs, qm, qm_inv);
  }
  /* ---- which feed covers this band */
  memset(&S, 0, sizeof(S));
  S.band = T.cur_band++;
  S.blk = T.cur_blk;
  if (T.cur_L != NULL) {
    S.L = T.cur_L;
    S.rec = (size_t)S.band*S.L->nblk + S.blk;
    S.nrec = (size_t)S.L->nbands*S.L->nblk;
  }
  else if (T.cur_P != NULL) {
    S.P = T.cur_P;
    S.rec = (size_t)S.band*S.P->nblk + S.blk;
    S.nrec = (size_t)S.P->nbands*S.P->nblk;
  }
  nodesync = robust || is_keyframe;
  cfl = is_keyframe && pli != 0 && !OD_DISABLE_CFL;
  have_xr = 0;
  m = 0;
  s = 1;
  theta = 0;
  /* ---- gains, correlation (:353-381) */
  if (S.P != NULL) {
    /* all of it from the feed: exact sums from the device, pow / acos by this process's libm
       in the feed's host stage */
    g = S.P->g[S.rec];
    gr = S.P->gr[S.rec];
    cg = S.P->cg[S.rec];
    cgr = S.P->cgr[S.rec];
    corr = S.P->corr[S.rec];
    r_null = S.P->isnull[S.rec];
    theta_search = S.P->flags[S.rec] & 1;
    noref_search = (S.P->flags[S.rec] >> 1) & 1;
    if (theta_search) theta = S.P->theta[S.rec];
    if (T.check) {
      double gc;
      double grc;
      if (od_pvq_compute_gain(x0, n, q0, &gc, beta, qm) != cg || gc != g
       || od_pvq_compute_gain(r0, n, q0, &grc, beta, qm) != cgr || grc != gr) {
        T.st.g2_mismatch++;
        T.st.check_fail++;
      }
    }
  }
  else {
    r_null = od_vector_is_null(r0, n);
    if (S.L != NULL) {
      /* g: the device's sqrt of the exact sum; cg: od_gain_compand of that g by THIS process's
         libm between the two device passes (od_hip_enc_feed_compand) */
      g = S.L->g[S.rec];
      cg = S.L->cg[S.rec];
      if (T.check) {
        double gc;
        if (od_pvq_compute_gain(x0, n, q0, &gc, beta, qm) != cg || gc != g) {
          T.st.g2_mismatch++;
          T.st.check_fail++;
        }
      }
    }
    else cg = od_pvq_compute_gain(x0, n, q0, &g, beta, qm);
    corr = 0;
    gr = 0;
    cgr = 0;
    if (!r_null) {
      for (i = 0; i < n; i++) {
        x[i] = x0[i]*qm[i]*OD_QM_SCALE_1;
        r[i] = r0[i]*qm[i]*OD_QM_SCALE_1;
        corr += x[i]*r[i];
      }
      have_xr = 1;
      cgr = od_pvq_compute_gain(r0, n, q0, &gr, beta, qm);
    }
    /* a null reference: corr is a sum of zeros, gr = sqrt(0) and od_gain_compand(0) = 0 */
    corr = corr/(1e-100 + g*gr);
    corr = OD_MAXF(OD_MINF(corr, 1.), -1.);
    theta_search = n <= OD_MAX_PVQ_SIZE && !r_null && corr > 0;
    noref_search = -1;            /* decided below, once cg is final */
  }
  if (cfl) cgr = 1;
  icgr = (int)floor(.5 + cgr);
  gain_offset = cgr - icgr;
  if (noref_search < 0) {
    noref_search = n <= OD_MAX_PVQ_SIZE && ((is_keyframe && pli == 0) || corr < .5 || cg < 2.);
  }
  /* ---- the incumbent every candidate has to beat: gain 0, no pulse, rate 0 (:368-398) */
  B.qg = 0;
  B.k = 0;
  B.qtheta = 0;
  B.y = NULL;
  B.dist = gain_weight*cg*cg;
  B.cost = B.dist + lambda*0.;
  if (is_keyframe) {
    B.noref = 1;
    B.itheta = -1;
    B.max_theta = 0;
    skip_dist = gain_weight*cg*cg;
  }
  else {
    /* an inter band cannot be "no reference, gain 0"; it can be skipped */
    double scgr;
    skip_dist = gain_weight*(cg - cgr)*(cg - cgr) + cgr*cg*(2 - 2*corr);
    scgr = OD_MAXF(0, gain_offset);
    if (icgr == 0) B.dist = gain_weight*(cg - scgr)*(cg - scgr) + scgr*cg*(2 - 2*corr);
    B.cost = B.dist + lambda*0.;
    B.noref = 0;
    B.itheta = 0;
    B.max_theta = 0;
  }
  /* ---- with-reference candidates (:399-448) */
  if (theta_search) {
    od_hip_search sc;
    cw_cache cache;
    int slot;
    int searching;                /* host searches: the reflected vector is set up */
    const int16_t *py;
    int ns;
    cache.n = 0;
    slot = 0;
    searching = 0;
    py = NULL;
    ns = (n + 1) & ~1;
    if (S.P != NULL) py = S.P->y + (size_t)S.P->nslots*S.P->nblk*(S.band == 0 ? 0 : S.P->off[S.band]);
    else {
      theta = acos(corr);
      m = od_compute_householder(r, n, gr, &s);
      od_apply_householder(x, r, n);
      for (i = m; i < n - 1; i++) x[i] = x[i + 1];
      od_hip_search_begin(&sc, x, n - 1);
      searching = 1;
      have_xr = 2;                /* r[] is now the Householder vector, x[] is consumed */
    }
    for (i = OD_MAXI(1, (int)floor(cg - gain_offset) - 1); i <= (int)ceil(cg - gain_offset); i++) {
      double qcg;
      int ts;
      int j;
      qcg = i + gain_offset;
      ts = od_pvq_compute_max_theta(qcg, beta);
      for (j = OD_MAXI(0, (int)floor(.5 + theta*2/M_PI*ts) - 2);
       j <= OD_MINI(ts - 1, (int)ceil(theta*2/M_PI*ts)); j++, slot++) {
        double qtheta;
        double cos_dist;
        double dist;
        double cost;
        const od_coeff *yc;
        int k;
        qtheta = od_pvq_compute_theta(j, ts);
        k = od_pvq_compute_k(qcg, j, qtheta, 0, n, beta, nodesync);
        if (py != NULL && slot < S.P->nref_slots && S.P->k[(size_t)slot*S.nrec + S.rec] == k) {
          /* the feed's slot for this (i, j): K re-derived here must be the feed's */
          cos_dist = S.P->cos_dist[(size_t)slot*S.nrec + S.rec];
          widen(y_tmp, py + ((size_t)slot*S.P->nblk + S.blk)*ns, n - 1);
          T.st.dev_hits++;
          if (T.check || (T.sample_every > 0 && ++T.sample_ctr >= T.sample_every)) {
            /* the sampled re-search that keeps a silently wrong feed from going unnoticed */
            od_coeff yv[MAXN];
            double cv;
            T.sample_ctr = 0;
            T.st.resampled++;
            if (!searching) {
              reflect(x, r, x0, r0, qm, n, gr, &m, &s);
              od_hip_search_begin(&sc, x, n - 1);
              searching = 1;
              have_xr = 2;
            }
            cv = od_hip_search_run(&sc, k, yv, qcg*cg*sin(theta)*sin(qtheta));
            if (cv != cos_dist || memcmp(yv, y_tmp, sizeof(od_coeff)*(n - 1)) != 0) T.st.check_fail++;
          }
        }
        else {
          if (py != NULL) {
            /* the feed does not hold this candidate (resolution beyond its table, a slot it
               does not reach): searched here */
            T.st.lost_sync++;
            if (!searching) {
              reflect(x, r, x0, r0, qm, n, gr, &m, &s);
              od_hip_search_begin(&sc, x, n - 1);
              searching = 1;
              have_xr = 2;
            }
          }
          cos_dist = timed_search(&sc, k, y_tmp, qcg*cg*sin(theta)*sin(qtheta), (pli != 0)*2 + 1);
          T.st.cpu_other++;
        }
        /* :428-431 */
        {
          double dist_theta;
          dist_theta = 2 - 2*cos(theta - qtheta) + sin(theta)*sin(qtheta)*(2 - 2*cos_dist);
          dist = gain_weight*(qcg - cg)*(qcg - cg) + qcg*cg*dist_theta;
        }
        /* The codeword's bits are >= 0 and + and * are monotonic, so the cost is at least the
           cost with the codeword bits left out (only the theta / gain terms of od_pvq_rate, which
           can be negative).  A candidate that cannot beat the incumbent even so is not priced. */
        if (!(dist + lambda*pvq_rate_with_codeword(0, i, icgr, j, ts, is_keyframe, pli) < B.cost)) continue;
        yc = y_tmp;
        cost = dist + lambda*pvq_rate_with_codeword(cached_codeword_rate(&cache, adapt, yc, k, n, bs), i, icgr,
         j, ts, is_keyframe, pli);
        if (cost < B.cost) {
          B.cost = cost;
          B.dist = dist;
          B.qg = i;
          B.k = k;
          B.qtheta = qtheta;
          B.itheta = j;
          B.max_theta = ts;
          B.noref = 0;
          OD_COPY(y_keep, yc, n - 1);
          B.y = y_keep;
        }
      }
    }
  }
  /* ---- no-reference candidates (:452-481) */
  if (noref_search) {
    od_hip_search sc;
    double xs[MAXN];              /* the vector sc searches: lives as long as sc (it keeps a pointer) */
    const int16_t *py;
    int ns;
    int base_slot;
    int c;
    int fed;                      /* candidates come from a feed */
    int searching;
    ns = (n + 1) & ~1;
    py = NULL;
    base_slot = 0;
    fed = 0;
    searching = 0;
    if (S.L != NULL) {
      /* the candidates the device enumerated must be the ones this loop visits: gains
         max(1, floor(cg)) .. ceil(cg), K recomputed here (the independent value that exposes
         a feed that is corrupt or out of step) */
      int nc;
      nc = 0;
      for (i = OD_MAXI(1, (int)floor(cg)); i <= ceil(cg); i++) nc++;
      fed = nc == S.L->ncand[S.rec] && nc <= 2;
      py = S.L->y + (size_t)2*S.L->nblk*(S.band == 0 ? 0 : S.L->off[S.band]);
      for (c = 0; fed && c < nc; c++) {
        const int16_t *yc;
        double cd;
        int sum;
        int j;
        i = OD_MAXI(1, (int)floor(cg)) + c;
        if (S.L->qg[c*S.nrec + S.rec] != i
         || S.L->k[c*S.nrec + S.rec] != od_pvq_compute_k(i, -1, -1, 1, n, beta, nodesync)) fed = 0;
        /* a codeword of the search has exactly K pulses and a cosine in [0, 1]: cheap
           integrity checks of the two fields that are taken on trust */
        yc = py + ((size_t)c*S.L->nblk + S.blk)*ns;
        sum = 0;
        for (j = 0; j < n; j++) sum += abs(yc[j]);
        cd = S.L->cos_dist[c*S.nrec + S.rec];
        if (sum != S.L->k[c*S.nrec + S.rec] || !(cd >= 0 && cd <= 1.0000001)) fed = 0;
      }
      if (!fed) T.st.lost_sync++;
    }
    else if (S.P != NULL) {
      fed = 1;
      base_slot = S.P->nref_slots;
      py = S.P->y + (size_t)S.P->nslots*S.P->nblk*(S.band == 0 ? 0 : S.P->off[S.band]);
    }
    c = 0;
    for (i = OD_MAXI(1, (int)floor(cg)); i <= ceil(cg); i++, c++) {
      double qcg;
      double cos_dist;
      double dist;
      double cost;
      int k;
      int from_feed;
      qcg = i;
      k = od_pvq_compute_k(qcg, -1, -1, 1, n, beta, nodesync);
      from_feed = 0;
      if (fed && S.L != NULL) {
        from_feed = 1;
        cos_dist = S.L->cos_dist[c*S.nrec + S.rec];
        widen(y_tmp, py + ((size_t)c*S.L->nblk + S.blk)*ns, n);
      }
      else if (fed && c < 2 && S.P->k[(size_t)(base_slot + c)*S.nrec + S.rec] == k) {
        from_feed = 1;
        cos_dist = S.P->cos_dist[(size_t)(base_slot + c)*S.nrec + S.rec];
        widen(y_tmp, py + ((size_t)(base_slot + c)*S.P->nblk + S.blk)*ns, n);
      }
      if (from_feed) {
        T.st.dev_hits++;
        if (T.check || (T.sample_every > 0 && ++T.sample_ctr >= T.sample_every)) {
          /* OD_CHECKASM for the candidate: always in check mode, and on every sample_every-th
             candidate otherwise (HIPENC_SAMPLE, default 1 in 256) */
          double x1[MAXN];
          od_coeff yv[MAXN];
          double rc;
          int j;
          T.sample_ctr = 0;
          T.st.resampled++;
          for (j = 0; j < n; j++) x1[j] = x0[j]*qm[j]*OD_QM_SCALE_1;
          rc = od_ref_pvq_search_rdo_double_cpu(x1, n, k, yv, qcg*cg);
          if (rc != cos_dist || memcmp(yv, y_tmp, sizeof(od_coeff)*n) != 0) T.st.check_fail++;
        }
      }
      else {
        if (!searching) {
          int j;
          for (j = 0; j < n; j++) xs[j] = x0[j]*qm[j]*OD_QM_SCALE_1;
          od_hip_search_begin(&sc, xs, n);
          searching = 1;
          if (fed) T.st.lost_sync++;
        }
        cos_dist = timed_search(&sc, k, y_tmp, qcg*cg, (pli != 0)*2);
        if (S.L != NULL) T.st.cpu_noref_luma++;
        else T.st.cpu_other++;
      }
      dist = gain_weight*(qcg - cg)*(qcg - cg) + qcg*cg*(2 - 2*cos_dist);
      if (!(dist <= B.cost)) continue;            /* no-reference rate = codeword bits >= 0 */
      cost = dist + lambda*od_hip_pvq_rate(i, 0, -1, 0, adapt, y_tmp, k, n, is_keyframe, pli, bs);
      if (cost <= B.cost) {
        B.cost = cost;
        B.dist = dist;
        B.qg = i;
        B.noref = 1;
        B.k = k;
        B.itheta = -1;
        B.max_theta = 0;
        OD_COPY(y_keep, y_tmp, n);
        B.y = y_keep;
      }
    }
  }
  /* ---- hand the winner over and synthesise like the decoder would (:483-503) */
  if (B.y != NULL) OD_COPY(y, B.y, B.noref ? n : n - 1);
  else OD_CLEAR(y, n);
  *itheta = B.itheta;
  *max_theta = B.max_theta;
  *vk = B.k;
  {
    int skip;
    skip = 0;
    if (B.noref) {
      if (B.qg == 0) skip = OD_PVQ_SKIP_ZERO;
    }
    else {
      if (!is_keyframe && B.qg == 0) skip = icgr ? OD_PVQ_SKIP_ZERO : OD_PVQ_SKIP_COPY;
      if (B.qg == icgr && B.itheta == 0 && !cfl) skip = OD_PVQ_SKIP_COPY;
    }
    if (skip == OD_PVQ_SKIP_COPY) OD_COPY(out, r0, n);
    else if (skip) OD_CLEAR(out, n);
    else {
      if (!B.noref && have_xr != 2) {
        /* the Householder vector of the reference (the searches ran on the device) */
        for (i = 0; i < n; i++) r[i] = r0[i]*qm[i]*OD_QM_SCALE_1;
        m = od_compute_householder(r, n, gr, &s);
      }
      g = od_gain_expand(B.qg + (B.noref ? 0 : gain_offset), q0, beta);
      od_pvq_synthesis_partial(out, y, r, n, B.noref, g, B.qtheta, m, s, qm_inv);
    }
  }
  *skip_diff += skip_dist - B.dist;
  if (is_keyframe) return B.noref ? B.qg : interleave_gain(B.qg, icgr);
  return B.noref ? B.qg - 1 : interleave_gain(B.qg + 1, icgr + 1);
}

/* ------------------------------------------------------------------------ */
/* od_encode_checkpoint / od_encode_rollback as od_pvq_encode calls them (src/pvq_encoder.c:718,
   :796; bound by the build recipe for that one source): between the two the function can only
   modify the range coder and a handful of members of the adaptation context - 2 KB instead of
   the 19.7 KB of the whole context.  The subset lives in the caller's od_rollback_buffer. */
typedef struct pvq_ckpt {
  od_ec_enc ec;
  od_pvq_codeword_ctx cw;
  generic_encoder model[3];
  int ext[PVQ_MAX_PARTITIONS];
  int exg[PVQ_MAX_PARTITIONS];
  uint16_t gaintheta[PVQ_MAX_PARTITIONS][16];
  uint16_t skip_dir[7];
  uint16_t skip_cdf[5];
  uint16_t q_cdf[4*4][4];
} pvq_ckpt;

void od_hip_pvq_checkpoint(const daala_enc_ctx *enc, od_rollback_buffer *rbuf) {
  const od_adapt_ctx *a;
  pvq_ckpt *c;
  int pli;
  int bs;
  int nb;
  int gt0;
  _Static_assert(sizeof(pvq_ckpt) <= sizeof(od_rollback_buffer), "the subset fits the caller's buffer");
  if (!T.host_pvq || T.in_pure) {
    od_encode_checkpoint(enc, rbuf);
    return;
  }
  a = &enc->state.adapt;
  c = (pvq_ckpt *)rbuf;
  pli = T.pli;
  bs = T.cur_bs;
  nb = OD_BAND_OFFSETS[bs][0];
  gt0 = (pli != 0)*OD_NBSIZES*PVQ_MAX_PARTITIONS + bs*PVQ_MAX_PARTITIONS;
  od_ec_enc_checkpoint(&c->ec, &enc->ec);
  c->cw = a->pvq.pvq_codeword_ctx;
  memcpy(c->model, a->pvq.pvq_param_model, sizeof(c->model));
  memcpy(c->ext, a->pvq.pvq_ext + bs*PVQ_MAX_PARTITIONS, sizeof(int)*nb);
  memcpy(c->exg, a->pvq.pvq_exg[pli][bs], sizeof(int)*nb);
  memcpy(c->gaintheta, a->pvq.pvq_gaintheta_cdf[gt0], sizeof(c->gaintheta[0])*nb);
  if (bs > 0) memcpy(c->skip_dir, a->pvq.pvq_skip_dir_cdf[(pli != 0) + 2*(bs - 1)], sizeof(c->skip_dir));
  memcpy(c->skip_cdf, a->skip_cdf[2*bs + (pli != 0)], sizeof(c->skip_cdf));
  if (bs == OD_NBSIZES - 1 && pli == 0) memcpy(c->q_cdf, a->q_cdf, sizeof(c->q_cdf));
}

void od_hip_pvq_rollback(daala_enc_ctx *enc, const od_rollback_buffer *rbuf) {
  od_adapt_ctx *a;
  const pvq_ckpt *c;
  int pli;
  int bs;
  int nb;
  int gt0;
  if (!T.host_pvq || T.in_pure) {
    od_encode_rollback(enc, rbuf);
    return;
  }
  a = &enc->state.adapt;
  c = (const pvq_ckpt *)rbuf;
  pli = T.pli;
  bs = T.cur_bs;
  nb = OD_BAND_OFFSETS[bs][0];
  gt0 = (pli != 0)*OD_NBSIZES*PVQ_MAX_PARTITIONS + bs*PVQ_MAX_PARTITIONS;
  od_ec_enc_rollback(&enc->ec, &c->ec);
  a->pvq.pvq_codeword_ctx = c->cw;
  memcpy(a->pvq.pvq_param_model, c->model, sizeof(c->model));
  memcpy(a->pvq.pvq_ext + bs*PVQ_MAX_PARTITIONS, c->ext, sizeof(int)*nb);
  memcpy(a->pvq.pvq_exg[pli][bs], c->exg, sizeof(int)*nb);
  memcpy(a->pvq.pvq_gaintheta_cdf[gt0], c->gaintheta, sizeof(c->gaintheta[0])*nb);
  if (bs > 0) memcpy(a->pvq.pvq_skip_dir_cdf[(pli != 0) + 2*(bs - 1)], c->skip_dir, sizeof(c->skip_dir));
  memcpy(a->skip_cdf[2*bs + (pli != 0)], c->skip_cdf, sizeof(c->skip_cdf));
  if (bs == OD_NBSIZES - 1 && pli == 0) memcpy(a->q_cdf, c->q_cdf, sizeof(c->q_cdf));
}

/* ------------------------------------------------------------------------ */
/* check mode: the reference's od_pvq_encode, untouched (od_pvq_encode_pure: a second compile
   of src/pvq_encoder.c with nothing bound), on the same inputs and state */
typedef struct ec_sig {
  od_ec_window low;
  uint16_t rng;
  int16_t cnt;
  uint32_t offs;
  uint32_t end_offs;
  od_ec_window end_window;
  int nend_bits;
} ec_sig;

static void ec_signature(ec_sig *s, const od_ec_enc *ec) {
  memset(s, 0, sizeof(*s));
  s->low = ec->low;
  s->rng = ec->rng;
  s->cnt = ec->cnt;
  s->offs = ec->offs;
  s->end_offs = ec->end_offs;
  s->end_window = ec->end_window;
  s->nend_bits = ec->nend_bits;
}

int od_pvq_encode_pure(daala_enc_ctx *enc, od_coeff *ref, od_coeff *in, od_coeff *out, int q0, int pli,
 int bs, const double *beta, int robust, int is_keyframe, int q_scaling, int bx, int by,
 const int16_t *qm, const int16_t *qm_inv);

/* the feed context of one block: which records its bands read */
static void set_block_context(daala_enc_ctx *enc, int pli, int bs, int is_keyframe, int bx, int by) {
  T.cur_band = 0;
  T.cur_bs = bs;
  T.cur_L = NULL;
  T.cur_P = NULL;
  T.cur_blk = 0;
  if (is_keyframe && T.lev != NULL && (pli == 0 || (pli > 0 && pli < 3 && T.levc[pli - 1] != NULL && bs <= 2))) {
    /* keyframes: luma always, a chroma plane when it is in the feed (its no-reference candidates
       are used whenever pvq_theta's condition for that search holds) */
    const od_hip_feed_level *L;
    int blk;
    L = pli == 0 ? &T.lev[3 - bs] : &T.levc[pli - 1][2 - bs];
    blk = (by >> bs)*L->nbx + (bx >> bs);
    if (L->g != NULL && blk >= 0 && blk < L->nblk && L->nbands == OD_BAND_OFFSETS[bs][0]) {
      T.cur_L = L;
      T.cur_blk = blk;
    }
  }
  else if (!is_keyframe && T.pf_valid && pli >= 0 && pli < 3) {
    const od_hip_pfeed_level *P;
    int level;
    int dec;
    int blk;
    dec = pli > 0;
    level = (dec ? 2 : 3) - bs;
    if (level >= 0 && level < (dec ? 3 : 4)) {
      int sbx;
      int sby;
      P = &T.pfv[pli][level];
      blk = (by >> bs)*P->nbx + (bx >> bs);
      /* superblocks that contain padding are not in the feed: there the encoder's input is
         the prediction's, not the picture's (src/encode.c:2443-2457) */
      sbx = (bx << 2) >> (5 - dec);
      sby = (by << 2) >> (5 - dec);
      if (P->g != NULL && blk >= 0 && blk < P->nblk && P->nbands == OD_BAND_OFFSETS[bs][0]
       && (sbx + 1)*32 <= enc->state.info.pic_width && (sby + 1)*32 <= enc->state.info.pic_height) {
        T.cur_P = P;
        T.cur_blk = blk;
      }
    }
  }
}

int od_pvq_encode_cpu(daala_enc_ctx *enc, od_coeff *ref, od_coeff *in, od_coeff *out, int q0, int pli,
 int bs, const double *beta, int robust, int is_keyframe, int q_scaling, int bx, int by,
 const int16_t *qm, const int16_t *qm_inv);

int od_hip_pvq_encode_host(daala_enc_ctx *enc, od_coeff *ref, od_coeff *in,
 od_coeff *out, int q0, int pli, int bs, const double *beta, int robust,
 int is_keyframe, int q_scaling, int bx, int by, const int16_t *qm,
 const int16_t *qm_inv) {
  int n2;
  int ret;
  n2 = 1 << (2*bs + 4);
  T.pli = pli;
  if (T.check) {
    /* OD_CHECKASM for the block: ours first, then the untouched reference function from the same
       state; the reference's result is the one that stays */
    od_rollback_buffer *rb;
    od_adapt_ctx *mine;
    od_coeff *ref0;
    od_coeff *out1;
    od_coeff *ref1;
    uint16_t *pre1;
    ec_sig s1;
    ec_sig s2;
    uint32_t offs0;
    unsigned char sbq;
    int sbi;
    int ret2;
    rb = (od_rollback_buffer *)malloc(sizeof(*rb));
    mine = (od_adapt_ctx *)malloc(sizeof(*mine));
    ref0 = (od_coeff *)malloc(sizeof(od_coeff)*n2*3);
    if (rb == NULL || mine == NULL || ref0 == NULL) {
      /* no memory for the comparison: the block is coded unchecked and counted as a failure */
      free(ref0);
      free(mine);
      free(rb);
      T.st.pvq_check_fail++;
      set_block_context(enc, pli, bs, is_keyframe, bx, by);
      return od_pvq_encode_cpu(enc, ref, in, out, q0, pli, bs, beta, robust, is_keyframe, q_scaling, bx, by,
       qm, qm_inv);
    }
    out1 = ref0 + n2;
    ref1 = out1 + n2;
    od_encode_checkpoint_cpu(enc, rb);
    memcpy(ref0, ref, sizeof(od_coeff)*n2);
    offs0 = enc->ec.offs;
    sbi = (by >> (OD_NBSIZES - 1))*enc->state.nhsb + (bx >> (OD_NBSIZES - 1));
    sbq = enc->state.sb_q_scaling[sbi];
    set_block_context(enc, pli, bs, is_keyframe, bx, by);
    ret = od_pvq_encode_cpu(enc, ref, in, out, q0, pli, bs, beta, robust, is_keyframe, q_scaling, bx, by,
     qm, qm_inv);
    memcpy(out1, out, sizeof(od_coeff)*n2);
    memcpy(ref1, ref, sizeof(od_coeff)*n2);
    *mine = enc->state.adapt;
    ec_signature(&s1, &enc->ec);
    pre1 = (uint16_t *)malloc(sizeof(uint16_t)*(enc->ec.offs - offs0 + 1));
    if (pre1 != NULL) memcpy(pre1, enc->ec.precarry_buf + offs0, sizeof(uint16_t)*(enc->ec.offs - offs0));
    od_encode_rollback_cpu(enc, rb);
    memcpy(ref, ref0, sizeof(od_coeff)*n2);
    enc->state.sb_q_scaling[sbi] = sbq;
    T.in_pure = 1;
    ret2 = od_pvq_encode_pure(enc, ref, in, out, q0, pli, bs, beta, robust, is_keyframe,
     q_scaling, bx, by, qm, qm_inv);
    T.in_pure = 0;
    ec_signature(&s2, &enc->ec);
    if (ret != ret2 || memcmp(out1 + 1, out + 1, sizeof(od_coeff)*(n2 - 1)) != 0
     || out1[0] != out[0] || memcmp(ref1, ref, sizeof(od_coeff)*n2) != 0
     || memcmp(mine, &enc->state.adapt, sizeof(*mine)) != 0
     || memcmp(&s1, &s2, sizeof(s1)) != 0
     || pre1 == NULL
     || memcmp(pre1, enc->ec.precarry_buf + offs0, sizeof(uint16_t)*(s1.offs - offs0)) != 0) {
      T.st.pvq_check_fail++;
    }
    free(pre1);
    free(ref0);
    free(mine);
    free(rb);
    return ret2;
  }
  set_block_context(enc, pli, bs, is_keyframe, bx, by);
  ret = od_pvq_encode_cpu(enc, ref, in, out, q0, pli, bs, beta, robust, is_keyframe, q_scaling, bx, by,
   qm, qm_inv);
  return ret;
}
