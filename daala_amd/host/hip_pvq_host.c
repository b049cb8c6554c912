/* hip_pvq_host.c - host side of SURVEY rows A16/A19: od_pvq_encode() and pvq_theta()
 * restated (keyframes and inter frames) so that they CONSUME the device feed where there is
 * one (keyframe luma) instead of redoing the device's work, plus a rate-only form of
 * od_pvq_rate() and one search context per band for the searches that stay here.
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
/* Rate-only range coder.  od_ec_encode() (src/entenc.c:173-215) updates rng from
   (fl, fh, ft, rng) alone and od_ec_enc_normalize() (:62-119) adds d = 16 - ilog(rng) to
   cnt + 8*offs; od_ec_enc_tell() is that sum + 10 + the raw bits (:655-659).  So the
   pair below reproduces od_ec_enc_tell_frac() of a trial encoder exactly. */
typedef struct hip_rc {
  unsigned rng;
  int nbits;       /* od_ec_enc_tell(): 1 after od_ec_enc_reset (cnt = -9) */
} hip_rc;

static inline void rc_renorm(hip_rc *c, unsigned r) {
  int d;
  d = 16 - OD_ILOG_NZ(r);
  c->nbits += d;
  c->rng = r << d;
}

/* ft in [16384, 32768] (od_ec_encode) */
static inline void rc_encode(hip_rc *c, unsigned fl, unsigned fh, unsigned ft) {
  unsigned r;
  unsigned d;
  unsigned e;
  unsigned u;
  unsigned v;
  int s;
  r = c->rng;
  s = r - ft >= ft;
  ft <<= s;
  fl <<= s;
  fh <<= s;
  d = r - ft;
  e = OD_SUBSATU(2*d, ft);
  u = fl + OD_MINI(fl, e) + OD_MINI(OD_SUBSATU(fl, e) >> 1, d);
  v = fh + OD_MINI(fh, e) + OD_MINI(OD_SUBSATU(fh, e) >> 1, d);
  rc_renorm(c, v - u);
}

/* od_ec_encode_q15 (src/entenc.c:222-252): ft == 32768 */
static inline void rc_encode_q15(hip_rc *c, unsigned fl, unsigned fh) {
  unsigned d;
  unsigned e;
  unsigned u;
  unsigned v;
  d = c->rng - 32768U;
  e = OD_SUBSATU(2*d, 32768U);
  u = fl + OD_MINI(fl, e) + OD_MINI(OD_SUBSATU(fl, e) >> 1, d);
  v = fh + OD_MINI(fh, e) + OD_MINI(OD_SUBSATU(fh, e) >> 1, d);
  rc_renorm(c, v - u);
}

/* od_ec_encode_cdf_unscaled (src/entenc.c:386-391) with the table given as row - offset */
static inline void rc_cdf_unscaled(hip_rc *c, int s, const uint16_t *cdf, unsigned offset,
 int nsyms) {
  unsigned fl;
  unsigned fh;
  unsigned ft;
  int sh;
  fl = s > 0 ? (uint16_t)(cdf[s - 1] - offset) : 0;
  fh = (uint16_t)(cdf[s] - offset);
  ft = (uint16_t)(cdf[nsyms - 1] - offset);
  sh = 15 - OD_ILOG_NZ(ft - 1);
  rc_encode(c, fl << sh, fh << sh, ft << sh);
}

/* laplace_encode_special (src/laplace_encoder.c:48-92) */
static void rc_laplace_special(hip_rc *c, int x, unsigned decay, int max) {
  int shift;
  int xs;
  int ms;
  int sym;
  const uint16_t *cdf;
  shift = 0;
  if (max == 0) return;
  while (((max >> shift) >= 15 || max == -1) && decay > 235) {
    decay = (decay*decay + 128) >> 8;
    shift++;
  }
  decay = OD_MINI(decay, 254);
  decay = OD_MAXI(decay, 2);
  xs = x >> shift;
  ms = max >> shift;
  cdf = EXP_CDF_TABLE[(decay + 1) >> 1];
  do {
    sym = OD_MINI(xs, 15);
    if (ms > 0 && ms < 15) rc_cdf_unscaled(c, sym, cdf, 0, ms + 1);
    else rc_encode_q15(c, sym > 0 ? cdf[sym - 1] : 0, cdf[sym]);
    xs -= 15;
    ms -= 15;
  }
  while (sym >= 15 && ms != 0);
  if (shift) c->nbits += shift;
}

/* min(254, 256*ex/(ex + 256)) for ex < 4096 (what ex is after laplace_encode's shift),
   filled when the library is loaded: no first-use race between workers */
static uint8_t decay_tab[4096];
static void __attribute__((constructor)) decay_tab_fill(void) {
  int e;
  for (e = 0; e < 4096; e++) decay_tab[e] = (uint8_t)OD_MINI(254, 256*e/(e + 256));
}

/* laplace_encode (src/laplace_encoder.c:101-138) */
static inline void rc_laplace(hip_rc *c, int x, int ex_q8, int k) {
  int shift;
  int xs;
  int sym;
  int decay;
  shift = OD_ILOG(ex_q8) - 11;
  if (shift < 0) shift = 0;
  ex_q8 = (ex_q8 + (1 << shift >> 1)) >> shift;
  k = (k + (1 << shift >> 1)) >> shift;
  xs = (x + (1 << shift >> 1)) >> shift;
  /* decay = min(254, 256*ex/(ex + 256)): ex < 4096 after the shift above, tabulated once */
  if (ex_q8 < 4096) decay = decay_tab[ex_q8];
  else decay = OD_MINI(254, 256*ex_q8/(ex_q8 + 256));
  sym = xs;
  if (sym > 15) sym = 15;
  if (k != 0) {
    rc_cdf_unscaled(c, sym, EXP_CDF_TABLE[(decay + 1) >> 1], LAPLACE_OFFSET[(decay + 1) >> 1],
     OD_MINI(k + 1, 16));
  }
  if (shift) {
    int special;
    special = xs == 0;
    if (shift - special > 0) c->nbits += shift - special;
  }
  if (xs >= 15) rc_laplace_special(c, xs - 15, decay, k - 15);
}

/* laplace_encode_vector_delta (src/laplace_encoder.c:140-200), bits only */
static void rc_laplace_vector_delta(hip_rc *c, const od_coeff *y, int n, int k,
 const int32_t *means) {
  int i;
  int prev;
  int first;
  int k_left;
  int coef;
  prev = 0;
  first = 1;
  k_left = k;
  coef = 256*means[OD_ADAPT_COUNT_Q8]/(1 + means[OD_ADAPT_COUNT_EX_Q8]);
  coef = OD_MAXI(coef, 1);
  for (i = 0; i < n; i++) {
    if (y[i] != 0) {
      int j;
      int count;
      int mag;
      mag = abs(y[i]);
      count = i - prev;
      if (first) {
        int decay;
        int ex;
        ex = coef*(n - prev)/k_left;
        if (ex > 65280) decay = 255;
        else {
          decay = OD_MINI(255,
           (int)((256*ex/(ex + 256) + (ex >> 5)*ex/((n + 1)*(n - 1)*(n - 1)))));
        }
        rc_laplace_special(c, count, decay, n - 1);
        first = 0;
      }
      else rc_laplace(c, count, coef*(n - prev)/k_left, n - prev - 1);
      c->nbits += 1;
      for (j = 0; j < mag - 1; j++) {
        rc_laplace(c, 0, coef*(n - i)/(k_left - 1 - j), n - i - 1);
      }
      k_left -= mag;
      prev = i;
      if (k_left == 0) break;
    }
  }
}

/* laplace_encode_vector (src/laplace_encoder.c:212-260), bits only */
static void rc_laplace_vector(hip_rc *c, const od_coeff *y, int n, int k,
 const int32_t *means) {
  int i;
  int kn;
  int exp_q8;
  int mean_k_q8;
  int mean_sum_ex_q8;
  if (k <= 1) {
    rc_laplace_vector_delta(c, y, n, k, means);
    return;
  }
  kn = k;
  mean_k_q8 = means[OD_ADAPT_K_Q8];
  mean_sum_ex_q8 = means[OD_ADAPT_SUM_EX_Q8];
  if (mean_k_q8 < 1 << 23) exp_q8 = 256*mean_k_q8/(1 + mean_sum_ex_q8);
  else exp_q8 = mean_k_q8/(1 + (mean_sum_ex_q8 >> 8));
  for (i = 0; i < n; i++) {
    int ex;
    int x;
    if (kn == 0) break;
    if (kn <= 1 && i != n - 1) {
      rc_laplace_vector_delta(c, y + i, n - i, kn, means);
      break;
    }
    x = abs(y[i]);
    ex = (2*exp_q8*kn + (n - i))/(2*(n - i));
    if (ex > kn*256) ex = kn*256;
    if (i != n - 1) rc_laplace(c, x, ex, kn);
    if (x != 0) c->nbits += 1;
    kn -= x;
  }
}

/* The codeword's share of od_pvq_rate (src/pvq_encoder.c:257-276): trial coding of y into
   a fresh range coder.  It depends on (y, k, n, noref, bs) and the adaptation state only, so
   within one band - where the state does not move - equal codewords have equal rates. */
static double pvq_codeword_rate(const od_adapt_ctx *adapt, const od_coeff *y0, int k, int n,
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
    rc_laplace_vector(&c, y0, n - !noref, k, cd->pvq_adapt + 4*(2*bs + noref));
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
static int neg_interleave(int x, int ref) {       /* src/pvq_encoder.c:236-240 */
  if (x < ref) return -2*(x - ref) - 1;
  else if (x < 2*ref) return 2*(x - ref);
  else return x - 1;
}

/* pvq_theta (src/pvq_encoder.c:311-511).  L/band/blk: the feed records of this band
   (keyframe luma with a device feed), or L == NULL. */
static int hip_pvq_theta(od_coeff *out, const od_coeff *x0, const od_coeff *r0, int n,
 int q0, od_coeff *y, int *itheta, int *max_theta, int *vk, double beta,
 double *skip_diff, int robust, int is_keyframe, int pli, const od_adapt_ctx *adapt,
 int bs, const int16_t *qm, const int16_t *qm_inv, const od_hip_feed_level *L, int band,
 int blk) {
  double g;
  double gr;
  double x[MAXN];
  double r[MAXN];
  od_coeff y_tmp[MAXN];
  od_coeff feed_y[2][MAXN];       /* the feed's 16-bit pulses of the two candidates, widened */
  const od_coeff *y_best;
  int i;
  int k;
  double cg;
  double cgr;
  int icgr;
  int qg;
  double best_cost;
  double best_dist;
  double dist;
  int s;
  int m;
  double theta;
  double corr;
  int best_k;
  double best_qtheta;
  double gain_offset;
  int noref;
  double lambda;
  double skip_dist;
  int cfl_enabled;
  int skip;
  double gain_weight;
  int r_null;
  int feed_ok;
  size_t rec;
  size_t nrec;
  lambda = OD_PVQ_LAMBDA;
  gain_weight = 1.4;
  rec = 0;
  nrec = 0;
  feed_ok = 0;
  r_null = od_vector_is_null(r0, n);
  if (L != NULL) {
    rec = (size_t)band*L->nblk + blk;
    nrec = (size_t)L->nbands*L->nblk;
    /* :360 - g: the device's sqrt of the exact sum; cg: od_gain_compand of that g, computed
       by THIS process's libm between the two device passes (od_hip_enc_feed_compand), i.e.
       the value od_pvq_compute_gain returns here */
    g = L->g[rec];
    cg = L->cg[rec];
    feed_ok = 1;
    if (T.check) {
      double gc;
      double cgc;
      cgc = od_pvq_compute_gain((od_coeff *)x0, n, q0, &gc, beta, qm);
      if (gc != g || cgc != cg) {
        T.st.g2_mismatch++;
        T.st.check_fail++;
      }
    }
  }
  else cg = od_pvq_compute_gain((od_coeff *)x0, n, q0, &g, beta, qm);
  corr = 0;
  if (!r_null) {
    /* :353-361 */
    for (i = 0; i < n; i++) {
      x[i] = x0[i]*qm[i]*OD_QM_SCALE_1;
      r[i] = r0[i]*qm[i]*OD_QM_SCALE_1;
      corr += x[i]*r[i];
    }
    cgr = od_pvq_compute_gain((od_coeff *)r0, n, q0, &gr, beta, qm);
  }
  else {
    /* a null reference: corr is a sum of zeros, gr = sqrt(0) and od_gain_compand(0) = 0 */
    gr = 0;
    cgr = 0;
  }
  cfl_enabled = is_keyframe && pli != 0 && !OD_DISABLE_CFL;
  if (cfl_enabled) cgr = 1;
  icgr = (int)floor(.5 + cgr);
  gain_offset = cgr - icgr;
  /* null case: gain 0, no pulse; its rate is 0 (:368-372) */
  qg = 0;
  dist = gain_weight*cg*cg;
  best_dist = dist;
  best_cost = dist + lambda*0.;
  noref = 1;
  best_k = 0;
  *itheta = -1;
  *max_theta = 0;
  OD_CLEAR(y, n);
  y_best = NULL;
  best_qtheta = 0;
  m = 0;
  s = 1;
  corr = corr/(1e-100 + g*gr);
  corr = OD_MAXF(OD_MINF(corr, 1.), -1.);
  if (is_keyframe) skip_dist = gain_weight*cg*cg;
  else skip_dist = gain_weight*(cg - cgr)*(cg - cgr) + cgr*cg*(2 - 2*corr);
  if (!is_keyframe) {
    /* noref with gain 0 is not allowed on inter frames, skip is (:385-398); the rate of
       (qg = 0, theta = 0, no codeword) is 0 */
    double scgr;
    scgr = OD_MAXF(0, gain_offset);
    if (icgr == 0) {
      best_dist = gain_weight*(cg - scgr)*(cg - scgr) + scgr*cg*(2 - 2*corr);
    }
    best_cost = best_dist + lambda*0.;
    best_qtheta = 0;
    *itheta = 0;
    *max_theta = 0;
    noref = 0;
  }
  if (n <= OD_MAX_PVQ_SIZE && !r_null && corr > 0) {
    /* :399-448, the reference's arithmetic: its input depends on the reconstruction of the
       neighbours (or of luma), so there is nothing the device could have prepared */
    od_hip_search sc;
    /* codewords already priced in this band: (k, y) -> codeword rate */
    od_coeff seen_y[OD_HIP_SEARCH_KCACHE][128];
    double seen_rate[OD_HIP_SEARCH_KCACHE];
    int seen_k[OD_HIP_SEARCH_KCACHE];
    int nseen;
    nseen = 0;
    theta = acos(corr);
    m = od_compute_householder(r, n, gr, &s);
    od_apply_householder(x, r, n);
    for (i = m; i < n - 1; i++) x[i] = x[i + 1];
    od_hip_search_begin(&sc, x, n - 1);
    for (i = OD_MAXI(1, (int)floor(cg - gain_offset) - 1);
     i <= (int)ceil(cg - gain_offset); i++) {
      int j;
      double qcg;
      int ts;
      qcg = i + gain_offset;
      ts = od_pvq_compute_max_theta(qcg, beta);
      for (j = OD_MAXI(0, (int)floor(.5 + theta*2/M_PI*ts) - 2);
       j <= OD_MINI(ts - 1, (int)ceil(theta*2/M_PI*ts)); j++) {
        double cos_dist;
        double cost;
        double dist_theta;
        double qtheta;
        double t0;
        qtheta = od_pvq_compute_theta(j, ts);
        k = od_pvq_compute_k(qcg, j, qtheta, 0, n, beta, robust || is_keyframe);
        t0 = T.time_cpu ? od_hipenc_now() : 0;
        cos_dist = od_hip_search_run(&sc, k, y_tmp, qcg*cg*sin(theta)*sin(qtheta));
        if (T.time_cpu) {
          double dt;
          dt = od_hipenc_now() - t0;
          T.st.search_cpu_s += dt;
          T.st.search_class_s[(pli != 0)*2 + 1] += dt;
        }
        T.st.cpu_other++;
        dist_theta = 2 - 2*cos(theta - qtheta)
         + sin(theta)*sin(qtheta)*(2 - 2*cos_dist);
        dist = gain_weight*(qcg - cg)*(qcg - cg) + qcg*cg*dist_theta;
        /* The codeword's bits are >= 0 and floating-point + and * are monotonic, so the
           cost is at least the cost with the codeword bits left out (k = 0 below: only the
           theta/gain terms of od_pvq_rate, which can be negative).  A candidate that cannot
           beat the incumbent even so is not priced. */
        if (!(dist + lambda*od_hip_pvq_rate(i, icgr, j, ts, adapt, NULL, 0, n, is_keyframe, pli, bs)
         < best_cost)) continue;
        {
          double cw;
          int e;
          cw = 0;
          if (k > 0) {
            for (e = 0; e < nseen; e++) {
              if (seen_k[e] == k && memcmp(seen_y[e], y_tmp, sizeof(od_coeff)*(n - 1)) == 0) break;
            }
            if (e < nseen) cw = seen_rate[e];
            else {
              cw = pvq_codeword_rate(adapt, y_tmp, k, n, 0, bs);
              if (nseen < OD_HIP_SEARCH_KCACHE && n - 1 <= 128) {
                seen_k[nseen] = k;
                seen_rate[nseen] = cw;
                memcpy(seen_y[nseen], y_tmp, sizeof(od_coeff)*(n - 1));
                nseen++;
              }
            }
          }
          cost = dist + lambda*pvq_rate_with_codeword(cw, i, icgr, j, ts, is_keyframe, pli);
        }
        if (cost < best_cost) {
          best_cost = cost;
          best_dist = dist;
          qg = i;
          best_k = k;
          best_qtheta = qtheta;
          *itheta = j;
          *max_theta = ts;
          noref = 0;
          OD_COPY(y, y_tmp, n - 1);
        }
      }
    }
  }
  if (n <= OD_MAX_PVQ_SIZE && ((is_keyframe && pli == 0) || corr < .5 || cg < 2.)) {
    /* :452-481 */
    int c;
    int from_feed;
    from_feed = 0;
    if (feed_ok) {
      /* the candidates the device enumerated must be the ones this loop visits: gains
         max(1, floor(cg)) .. ceil(cg), K recomputed here (the independent value that
         exposes a feed that is corrupt or out of step) */
      int nc;
      nc = 0;
      for (i = OD_MAXI(1, (int)floor(cg)); i <= ceil(cg); i++) nc++;
      from_feed = nc == L->ncand[rec] && nc <= 2;
      for (c = 0; from_feed && c < nc; c++) {
        const int16_t *yc;
        double cd;
        int sum;
        int j;
        i = OD_MAXI(1, (int)floor(cg)) + c;
        if (L->qg[c*nrec + rec] != i
         || L->k[c*nrec + rec] != od_pvq_compute_k(i, -1, -1, 1, n, beta, robust || is_keyframe)) from_feed = 0;
        /* a codeword of the search has exactly K pulses and a cosine in [0, 1]: cheap
           integrity checks of the two fields that are taken on trust.  16-bit pulses, band b
           at 2*nblk*yo[b] in runs of ns[b] (include/daala_hip.h section 4b). */
        yc = L->y + (size_t)2*L->nblk*(band == 0 ? 0 : L->off[band]) + ((size_t)c*L->nblk + blk)*((n + 1) & ~1);
        sum = 0;
        for (j = 0; j < n; j++) sum += abs(yc[j]);
        cd = L->cos_dist[c*nrec + rec];
        if (sum != L->k[c*nrec + rec] || !(cd >= 0 && cd <= 1.0000001)) from_feed = 0;
      }
      if (!from_feed) T.st.lost_sync++;
    }
    if (from_feed) {
      const int16_t *yb;
      int ns;
      ns = (n + 1) & ~1;
      yb = L->y + (size_t)2*L->nblk*(band == 0 ? 0 : L->off[band]) + (size_t)blk*ns;
      c = 0;
      for (i = OD_MAXI(1, (int)floor(cg)); i <= ceil(cg); i++, c++) {
        double cos_dist;
        double cost;
        double qcg;
        od_coeff *yc;
        int j;
        qcg = i;
        k = L->k[c*nrec + rec];
        cos_dist = L->cos_dist[c*nrec + rec];
        yc = feed_y[c];
        for (j = 0; j < n; j++) yc[j] = yb[(size_t)c*L->nblk*ns + j];
        if (T.check || (T.sample_every > 0 && ++T.sample_ctr >= T.sample_every)) {
          /* OD_CHECKASM for the candidate: always in check mode, and on every
             sample_every-th candidate otherwise (the sampled re-search that keeps a silently
             wrong feed from going unnoticed; HIPENC_SAMPLE, default 1 in 256) */
          double x1[MAXN];
          double rc;
          int j;
          T.sample_ctr = 0;
          T.st.resampled++;
          for (j = 0; j < n; j++) x1[j] = x0[j]*qm[j]*OD_QM_SCALE_1;
          rc = od_ref_pvq_search_rdo_double_cpu(x1, n, k, y_tmp, qcg*cg);
          if (rc != cos_dist || memcmp(y_tmp, yc, sizeof(od_coeff)*n) != 0) T.st.check_fail++;
        }
        T.st.dev_hits++;
        dist = gain_weight*(qcg - cg)*(qcg - cg) + qcg*cg*(2 - 2*cos_dist);
        if (!(dist <= best_cost)) continue;         /* no-reference rate = codeword bits >= 0 */
        cost = dist + lambda*od_hip_pvq_rate(i, 0, -1, 0, adapt, yc, k, n, is_keyframe, pli, bs);
        if (cost <= best_cost) {
          best_cost = cost;
          best_dist = dist;
          qg = i;
          noref = 1;
          best_k = k;
          *itheta = -1;
          *max_theta = 0;
          y_best = yc;
        }
      }
    }
    else {
      double x1[MAXN];
      od_hip_search sc;
      for (i = 0; i < n; i++) x1[i] = x0[i]*qm[i]*OD_QM_SCALE_1;
      od_hip_search_begin(&sc, x1, n);
      for (i = OD_MAXI(1, (int)floor(cg)); i <= ceil(cg); i++) {
        double cos_dist;
        double cost;
        double qcg;
        double t0;
        qcg = i;
        k = od_pvq_compute_k(qcg, -1, -1, 1, n, beta, robust || is_keyframe);
        t0 = T.time_cpu ? od_hipenc_now() : 0;
        cos_dist = od_hip_search_run(&sc, k, y_tmp, qcg*cg);
        if (T.time_cpu) {
          double dt;
          dt = od_hipenc_now() - t0;
          T.st.search_cpu_s += dt;
          T.st.search_class_s[(pli != 0)*2] += dt;
        }
        if (L != NULL) T.st.cpu_noref_luma++;
        else T.st.cpu_other++;
        dist = gain_weight*(qcg - cg)*(qcg - cg) + qcg*cg*(2 - 2*cos_dist);
        if (!(dist <= best_cost)) continue;
        cost = dist + lambda*od_hip_pvq_rate(i, 0, -1, 0, adapt, y_tmp, k, n, is_keyframe, pli, bs);
        if (cost <= best_cost) {
          best_cost = cost;
          best_dist = dist;
          qg = i;
          noref = 1;
          best_k = k;
          *itheta = -1;
          *max_theta = 0;
          y_best = NULL;
          OD_COPY(y, y_tmp, n);
        }
      }
    }
  }
  if (y_best != NULL) OD_COPY(y, y_best, n);
  k = best_k;
  theta = best_qtheta;
  skip = 0;
  if (noref) {
    if (qg == 0) skip = OD_PVQ_SKIP_ZERO;
  }
  else {
    if (!is_keyframe && qg == 0) skip = (icgr ? OD_PVQ_SKIP_ZERO : OD_PVQ_SKIP_COPY);
    if (qg == icgr && *itheta == 0 && !cfl_enabled) skip = OD_PVQ_SKIP_COPY;
  }
  /* Synthesize like the decoder would (:493-503). */
  if (skip) {
    if (skip == OD_PVQ_SKIP_COPY) OD_COPY(out, r0, n);
    else OD_CLEAR(out, n);
  }
  else {
    if (noref) gain_offset = 0;
    g = od_gain_expand(qg + gain_offset, q0, beta);
    od_pvq_synthesis_partial(out, y, r, n, noref, g, theta, m, s, qm_inv);
  }
  *vk = k;
  *skip_diff += skip_dist - best_dist;
  if (is_keyframe) return noref ? qg : neg_interleave(qg, icgr);
  return noref ? qg - 1 : neg_interleave(qg + 1, icgr + 1);
}

/* ------------------------------------------------------------------------ */
/* What od_pvq_encode's coding section (src/pvq_encoder.c:718-776) can modify: the
   range coder and these members of the adaptation context. */
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

static void pvq_save(pvq_ckpt *c, daala_enc_ctx *enc, int pli, int bs, int nb, int gt0) {
  od_adapt_ctx *a;
  a = &enc->state.adapt;
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

static void pvq_restore(const pvq_ckpt *c, daala_enc_ctx *enc, int pli, int bs, int nb, int gt0) {
  od_adapt_ctx *a;
  a = &enc->state.adapt;
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

/* od_pvq_encode (src/pvq_encoder.c:645-815) for keyframes */
static int pvq_encode_block(daala_enc_ctx *enc, od_coeff *ref, od_coeff *in,
 od_coeff *out, int q0, int pli, int bs, const double *beta, int robust, int is_keyframe,
 int q_scaling, int bx, int by, const int16_t *qm, const int16_t *qm_inv) {
  int theta[PVQ_MAX_PARTITIONS];
  int max_theta[PVQ_MAX_PARTITIONS];
  int qg[PVQ_MAX_PARTITIONS];
  int k[PVQ_MAX_PARTITIONS];
  od_coeff y[OD_BSIZE_MAX*OD_BSIZE_MAX];
  int size[PVQ_MAX_PARTITIONS];
  int *exg;
  int *ext;
  int nb_bands;
  int nb0;
  int i;
  const int *off;
  generic_encoder *model;
  double skip_diff;
  int tell;
  uint16_t *skip_cdf;
  pvq_ckpt buf;
  int flip;
  int cfl_encoded;
  int skip_rest;
  int skip_dir;
  int gt0;
  int dc_quant;
  int skip_theta_value;
  double dc_rate;
  const unsigned char *pvq_qm;
  const od_hip_feed_level *L;
  int blk;
  od_adapt_ctx *adapt;
  adapt = &enc->state.adapt;
  pvq_qm = &enc->state.pvq_qm_q4[pli][0];
  exg = &adapt->pvq.pvq_exg[pli][bs][0];
  ext = adapt->pvq.pvq_ext + bs*PVQ_MAX_PARTITIONS;
  skip_cdf = adapt->skip_cdf[2*bs + (pli != 0)];
  model = adapt->pvq.pvq_param_model;
  nb_bands = OD_BAND_OFFSETS[bs][0];
  nb0 = nb_bands;
  off = &OD_BAND_OFFSETS[bs][1];
  gt0 = (pli != 0)*OD_NBSIZES*PVQ_MAX_PARTITIONS + bs*PVQ_MAX_PARTITIONS;
  dc_quant = OD_MAXI(1, q0*pvq_qm[od_qm_get_index(bs, 0)] >> 4);
  for (i = 0; i < nb_bands; i++) size[i] = off[i + 1] - off[i];
  skip_diff = 0;
  flip = 0;
  /* chroma of a keyframe is predicted from luma: negate the reference when the first
     band points away from it (:697-709) */
  if (pli != 0 && is_keyframe) {
    double xy;
    xy = 0;
    for (i = off[0]; i < off[1]; i++) {
      xy += ref[i]*qm[i]*OD_QM_SCALE_1*(double)in[i]*qm[i]*OD_QM_SCALE_1;
    }
    if (xy < 0) {
      flip = 1;
      for (i = off[0]; i < off[nb_bands]; i++) ref[i] = -ref[i];
    }
  }
  /* the feed records of this block: keyframe luma, level 3 - bs, bx/by in 4x4 units */
  L = NULL;
  blk = 0;
  if (T.lev != NULL && pli == 0 && is_keyframe) {
    L = &T.lev[3 - bs];
    blk = (by >> bs)*L->nbx + (bx >> bs);
    if (L->g == NULL || blk < 0 || blk >= L->nblk || L->nbands != nb_bands) L = NULL;
    else {
      /* the records of one block sit nblk entries apart (band-major, the device's
         coalesced layout): start pulling them in before the first band needs them */
      size_t nrec;
      nrec = (size_t)L->nbands*L->nblk;
      for (i = 0; i < nb_bands; i++) {
        size_t rec;
        rec = (size_t)i*L->nblk + blk;
        __builtin_prefetch(L->g + rec);
        __builtin_prefetch(L->cg + rec);
        __builtin_prefetch(L->ncand + rec);
        __builtin_prefetch(L->k + rec);
        __builtin_prefetch(L->qg + rec);
        __builtin_prefetch(L->cos_dist + rec);
        __builtin_prefetch(L->k + nrec + rec);
        __builtin_prefetch(L->qg + nrec + rec);
        __builtin_prefetch(L->cos_dist + nrec + rec);
      }
    }
  }
  for (i = 0; i < nb_bands; i++) {
    int q;
    q = OD_MAXI(1, q0*pvq_qm[od_qm_get_index(bs, i + 1)] >> 4);
    qg[i] = hip_pvq_theta(out + off[i], in + off[i], ref + off[i], size[i], q, y + off[i],
     &theta[i], &max_theta[i], &k[i], beta[i], &skip_diff, robust, is_keyframe, pli, adapt,
     bs, qm + off[i], qm_inv + off[i], L, i, blk);
  }
  pvq_save(&buf, enc, pli, bs, nb0, gt0);
  if (is_keyframe) out[0] = 0;
  else {
    dc_rate = -OD_LOG2((double)(skip_cdf[1] - skip_cdf[0])/(double)skip_cdf[0]);
    out[0] = od_rdo_quant(in[0] - ref[0], dc_quant, dc_rate);
  }
  tell = od_ec_enc_tell_frac(&enc->ec);
  /* Code as if we're not skipping. */
  od_encode_cdf_adapt(&enc->ec, out[0] != 0, skip_cdf, 4 + (pli == 0 && bs > 0),
   adapt->skip_increment);
#if OD_SIGNAL_Q_SCALING
  if (bs == OD_NBSIZES - 1 && pli == 0) {
    od_encode_quantizer_scaling(enc, q_scaling, bx >> (OD_NBSIZES - 1), by >> (OD_NBSIZES - 1), 0);
  }
#endif
  cfl_encoded = 0;
  skip_rest = 1;
  skip_theta_value = is_keyframe ? -1 : 0;
  for (i = 1; i < nb_bands; i++) {
    if (theta[i] != skip_theta_value || qg[i]) skip_rest = 0;
  }
  skip_dir = 0;
  if (nb_bands > 1) {
    for (i = 0; i < 3; i++) {
      int j;
      int tmp;
      tmp = 1;
      for (j = i + 1; j < nb_bands; j += 3) {
        if (theta[j] != skip_theta_value || qg[j]) tmp = 0;
      }
      skip_dir |= tmp << i;
    }
  }
  if (theta[0] == skip_theta_value && qg[0] == 0 && skip_rest) nb_bands = 0;
  for (i = 0; i < nb_bands; i++) {
    if (i == 0 || (!skip_rest && !(skip_dir & (1 << ((i - 1)%3))))) {
      od_ref_pvq_encode_partition(&enc->ec, qg[i], theta[i], max_theta[i], y + off[i], size[i],
       k[i], model, adapt, exg + i, ext + i, robust || is_keyframe, gt0 + i, is_keyframe,
       i == 0 && (i < nb_bands - 1), skip_rest, bs);
    }
    if (i == 0 && !skip_rest && bs > 0) {
      od_encode_cdf_adapt(&enc->ec, skip_dir,
       &adapt->pvq.pvq_skip_dir_cdf[(pli != 0) + 2*(bs - 1)][0], 7,
       adapt->pvq.pvq_skip_dir_increment);
    }
    if (pli != 0 && is_keyframe && theta[i] != -1 && !cfl_encoded) {
      od_ec_enc_bits(&enc->ec, flip, 1);
      cfl_encoded = 1;
    }
  }
  tell = od_ec_enc_tell_frac(&enc->ec) - tell;
  /* the rate of skipping the AC instead (:778-787) */
  {
    double skip_rate;
    int skip_flag;
    skip_flag = 2 + (out[0] != 0);
    skip_rate = -OD_LOG2((skip_cdf[skip_flag] - skip_cdf[skip_flag - 1])/
     (double)skip_cdf[3 + (pli == 0 && bs > 0)]);
    tell -= (int)floor(.5 + 8*skip_rate);
  }
  if (nb_bands == 0 || skip_diff <= OD_PVQ_LAMBDA/8*tell) {
    /* skip: everything back as it was (:788-813) */
    if (is_keyframe) out[0] = 0;
    else {
      dc_rate = -OD_LOG2((double)(skip_cdf[3] - skip_cdf[2])/(double)(skip_cdf[2] - skip_cdf[1]));
      out[0] = od_rdo_quant(in[0] - ref[0], dc_quant, dc_rate);
    }
    pvq_restore(&buf, enc, pli, bs, nb0, gt0);
    od_encode_cdf_adapt(&enc->ec, 2 + (out[0] != 0), skip_cdf, 4 + (pli == 0 && bs > 0),
     adapt->skip_increment);
#if OD_SIGNAL_Q_SCALING
    if (bs == OD_NBSIZES - 1 && pli == 0) {
      int skip;
      skip = out[0] == 0;
      if (skip) q_scaling = 0;
      od_encode_quantizer_scaling(enc, q_scaling, bx >> (OD_NBSIZES - 1), by >> (OD_NBSIZES - 1),
       skip);
    }
#endif
    if (is_keyframe) for (i = 1; i < 1 << (2*bs + 4); i++) out[i] = 0;
    else for (i = 1; i < 1 << (2*bs + 4); i++) out[i] = ref[i];
    if (out[0] == 0) return 1;
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* check mode: the reference's od_pvq_encode on the same inputs and state */
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

int od_hip_pvq_encode_host(daala_enc_ctx *enc, od_coeff *ref, od_coeff *in,
 od_coeff *out, int q0, int pli, int bs, const double *beta, int robust,
 int is_keyframe, int q_scaling, int bx, int by, const int16_t *qm,
 const int16_t *qm_inv) {
  int n2;
  int ret;
  n2 = 1 << (2*bs + 4);
  if (T.check) {
    /* OD_CHECKASM for the block: ours first, then the reference's own od_pvq_encode from
       the same state; the reference's result is the one that stays */
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
    const od_hip_feed_level *lev;
    rb = (od_rollback_buffer *)malloc(sizeof(*rb));
    mine = (od_adapt_ctx *)malloc(sizeof(*mine));
    ref0 = (od_coeff *)malloc(sizeof(od_coeff)*n2*3);
    out1 = ref0 + n2;
    ref1 = out1 + n2;
    od_encode_checkpoint_cpu(enc, rb);
    memcpy(ref0, ref, sizeof(od_coeff)*n2);
    offs0 = enc->ec.offs;
    sbi = (by >> (OD_NBSIZES - 1))*enc->state.nhsb + (bx >> (OD_NBSIZES - 1));
    sbq = enc->state.sb_q_scaling[sbi];
    ret = pvq_encode_block(enc, ref, in, out, q0, pli, bs, beta, robust, is_keyframe, q_scaling, bx, by,
     qm, qm_inv);
    memcpy(out1, out, sizeof(od_coeff)*n2);
    memcpy(ref1, ref, sizeof(od_coeff)*n2);
    *mine = enc->state.adapt;
    ec_signature(&s1, &enc->ec);
    pre1 = (uint16_t *)malloc(sizeof(uint16_t)*(enc->ec.offs - offs0 + 1));
    memcpy(pre1, enc->ec.precarry_buf + offs0, sizeof(uint16_t)*(enc->ec.offs - offs0));
    od_encode_rollback_cpu(enc, rb);
    memcpy(ref, ref0, sizeof(od_coeff)*n2);
    enc->state.sb_q_scaling[sbi] = sbq;
    lev = T.lev;
    T.lev = NULL;                 /* the reference runs its own C search */
    ret2 = od_pvq_encode_cpu(enc, ref, in, out, q0, pli, bs, beta, robust, is_keyframe,
     q_scaling, bx, by, qm, qm_inv);
    T.lev = lev;
    ec_signature(&s2, &enc->ec);
    if (ret != ret2 || memcmp(out1 + 1, out + 1, sizeof(od_coeff)*(n2 - 1)) != 0
     || out1[0] != out[0] || memcmp(ref1, ref, sizeof(od_coeff)*n2) != 0
     || memcmp(mine, &enc->state.adapt, sizeof(*mine)) != 0
     || memcmp(&s1, &s2, sizeof(s1)) != 0
     || memcmp(pre1, enc->ec.precarry_buf + offs0, sizeof(uint16_t)*(s1.offs - offs0)) != 0) {
      T.st.pvq_check_fail++;
    }
    free(pre1);
    free(ref0);
    free(mine);
    free(rb);
    return ret2;
  }
  ret = pvq_encode_block(enc, ref, in, out, q0, pli, bs, beta, robust, is_keyframe, q_scaling, bx, by,
   qm, qm_inv);
  return ret;
}
