/* hip_enc_glue.c - reference-side binding of the batched frame seam
 * (INTEGRATION.md seam 2; header hip_enc_glue.h).
 *
 * Compiled into a build of the reference encoder (the integration build, INTEGRATION.md
 * "Seam 2, live": libdaala_hipenc.so): it includes the reference's internal headers and
 * provides the two symbols that build leaves open,
 *
 *   pvq_search_rdo_double()   the call sites of src/pvq_encoder.c:426,463 - the
 *                             reference's own definition (:121) is kept under the
 *                             name pvq_search_rdo_double_cpu by the build recipe;
 *   od_pvq_encode()           called by od_block_encode (src/encode.c:1173) - the
 *                             reference's definition (src/pvq_encoder.c:645) is
 *                             kept as od_pvq_encode_cpu;
 *
 * so that every keyframe-luma no-reference search (src/pvq_encoder.c:452-481, the
 * state-free half of pvq_theta) is answered from the device feed
 * (include/daala_hip.h section 4b) while with-reference and chroma searches, whose
 * inputs depend on the serial reconstruction, stay the reference's C code.
 * No reference text lives in this file. */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "encint.h"
#include "pvq.h"
#include "pvq_encoder.h"
#include "partition.h"

#include "hip_glue_int.h"

__thread glue_tls od_hipenc_tls;
#define T od_hipenc_tls

void od_hipenc_copy_pad(daala_enc_ctx *enc, od_img *img);

double od_hipenc_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9*ts.tv_nsec;
}
#define now_s od_hipenc_now

/* The fine-grained timers of HIPENC_TIME=1 (tens of millions of intervals of about a
   microsecond per step) read the time stamp counter: a clock_gettime that is a system call on
   this kind of guest costs as much as what it brackets.  Seconds per tick are calibrated once
   per process against the monotonic clock. */
#if defined(__x86_64__)
static double tsc_period;
static pthread_once_t tsc_once = PTHREAD_ONCE_INIT;
static void tsc_calibrate(void) {
  double t0;
  double t1;
  unsigned long long c0;
  unsigned long long c1;
  t0 = od_hipenc_now();
  c0 = __builtin_ia32_rdtsc();
  do t1 = od_hipenc_now(); while (t1 - t0 < 0.005);
  c1 = __builtin_ia32_rdtsc();
  tsc_period = (t1 - t0)/(double)(c1 - c0);
}
void od_hipenc_fine_timer_init(void) {
  pthread_once(&tsc_once, tsc_calibrate);
}
double od_hipenc_fine_now(void) {
  return (double)__builtin_ia32_rdtsc()*tsc_period;
}
#else
/* no time stamp counter: the sampled timers of HIPENC_TIME=1 use the monotonic clock */
void od_hipenc_fine_timer_init(void) {
}
double od_hipenc_fine_now(void) {
  return od_hipenc_now();
}
#endif

/* pvq_search_rdo_double as the reference's own pvq_theta calls it (inter frames, and the
   reference's od_pvq_encode in check mode).  Check mode keeps the reference's C search (it is
   the checker there); otherwise vectors of 24 coefficients and more take the lane-wise search
   of hip_pvq_search.c (bit-identical, tests/test_hipenc_cpu.py), shorter ones the C search. */
double pvq_search_rdo_double(const double *xcoeff, int n, int k, od_coeff *ypulse,
 double g2) {
  double t0;
  double r;
  int timed;
  T.st.cpu_other++;
  timed = T.time_cpu && ++T.search_tick % OD_HIPENC_TIME_SAMPLE == 0;
  t0 = timed ? od_hipenc_fine_now() : 0;
  if (n >= 24 && !T.check) {
    /* the reference's pvq_theta searches one vector once per (gain, theta) candidate: while
       the input stays the same (compared by value) the search context is kept, so the
       candidate-independent sums and the greedy pulses of a repeated K are not redone */
    static __thread od_hip_search sc;
    static __thread double sc_x[MAXN];
    static __thread int sc_n;
    if (sc_n != n || memcmp(sc_x, xcoeff, sizeof(double)*n) != 0) {
      memcpy(sc_x, xcoeff, sizeof(double)*n);
      sc_n = n;
      od_hip_search_begin(&sc, sc_x, n);
    }
    r = od_hip_search_run(&sc, k, ypulse, g2);
  }
  else r = od_ref_pvq_search_rdo_double_cpu(xcoeff, n, k, ypulse, g2);
  if (timed) {
    double dt;
    dt = OD_HIPENC_TIME_SAMPLE*(od_hipenc_fine_now() - t0);
    T.st.search_cpu_s += dt;
    T.st.search_class_s[(T.pli != 0)*2 + !(n == 15 || n == 8 || n == 32 || n == 128)] += dt;
  }
  return r;
}

/* for the bindings in hip_dec_glue.c that serve encoder threads too (od_dering,
   od_state_mc_predict): is this thread a worker of a device session / in check mode */
int od_hipenc_device_thread(void) {
  return T.dr != NULL;
}

int od_hipenc_check_mode(void) {
  return T.check;
}

/* od_dering (src/filter.c:1835) as the encoder's filter on/off loop calls it
   (src/encode.c:2597, :2662), superblock by superblock and plane by plane: the first
   call of a frame runs od_hip_dering_run once on the three unfiltered planes
   (state.etmp, complete before that loop starts, :2541-2549) - deringing reads only
   those, so every superblock is independent - and every call copies its block out of
   the result.  Called from the od_dering binding in hip_dec_glue.c; returns 1 when it
   served the call. */
void od_dering_cpu(od_state *state, int16_t *y, int ystride, int16_t *x, int xstride,
 int ln, int sbx, int sby, int nhsb, int nvsb, int q, int xdec,
 int dir[OD_DERING_NBLOCKS][OD_DERING_NBLOCKS], int pli, unsigned char *bskip,
 int skip_stride);

int od_hipenc_dering_hook(od_state *state, int16_t *y, int ystride, int16_t *x,
 int xstride, int ln, int sbx, int sby, int nhsb, int nvsb, int q, int xdec,
 int dir[OD_DERING_NBLOCKS][OD_DERING_NBLOCKS], int pli, unsigned char *bskip,
 int skip_stride) {
  const int16_t *src;
  int n;
  int w;
  int i;
  if (T.dr == NULL || T.enc == NULL || state != &T.enc->state || T.dr_error
   || (state->frame_type != OD_I_FRAME && state->frame_type != OD_P_FRAME) || pli < 0 || pli > 2) {
    return 0;
  }
  if (!T.dr_valid) {
    const int16_t *in[3];
    const unsigned char *sk[3];
    int32_t thr[3];
    int32_t quant[3];
    int p;
    for (p = 0; p < 3; p++) {
      in[p] = state->etmp[p];
      sk[p] = state->bskip[p];
      quant[p] = state->quantizer[p];
      thr[p] = (int32_t)(1.0*pow(state->quantizer[p], 0.84182));     /* src/filter.c:1876 */
    }
    /* The two od_compute_dist calls per luma superblock of the same loop (:2634-2635) can
       come from the same device pass when their operands are what the device sees: the HVS
       matrix (the other branch is a plain SSD), and an unfiltered reconstruction that fits
       the int16 planes od_dering works on (the distortion reads the int32 ctmp plane). */
    T.dist_valid = 0;
    T.dist_sb = -1;
    if (T.dist[0] != NULL && T.enc->qm == OD_HVS_QM) {
      const od_coeff *c0;
      size_t np;
      size_t i;
      int fits;
      c0 = state->ctmp[0];
      np = (size_t)state->frame_width*state->frame_height;
      fits = 1;
      for (i = 0; i < np; i++) fits &= c0[i] == (int16_t)c0[i];
      if (fits) {
        double mag2[64];
        const od_img_plane *ip;
        int a;
        int b;
        /* src/encode.c:1018-1025, bs = 3 */
        for (a = 0; a < 8; a++) {
          for (b = 0; b < 8; b++) {
            double mag;
            mag = 16./OD_QM8_Q4_HVS[a*8 + b];
            mag *= OD_BASIS_MAG[0][3][a << 2]*OD_BASIS_MAG[0][3][b << 2];
            mag *= mag;
            mag2[a*8 + b] = mag;
          }
        }
        ip = &T.enc->input_img[T.enc->curr_frame].planes[0];
        if (ip->xstride == 1 && od_hip_dering_run_dist(T.dr, in, sk, state->skip_stride, thr, quant,
         T.dr_out, ip->data, ip->ystride, mag2, T.enc->use_activity_masking, T.dist[0], T.dist[1],
         T.dist[2]) == 0) {
          T.dist_valid = 1;
        }
        else {
          T.dr_error = 1;
          return 0;
        }
      }
    }
    if (!T.dist_valid && od_hip_dering_run(T.dr, in, sk, state->skip_stride, thr, quant, T.dr_out) != 0) {
      T.dr_error = 1;      /* reported as a failed frame by encode_frame: no silent C path */
      return 0;
    }
    T.dr_valid = 1;
  }
  n = 1 << ln;
  w = state->frame_width >> (pli > 0);
  src = T.dr_out[pli] + (size_t)(sby << ln)*w + (sbx << ln);
  if (T.check) {
    int16_t ref[32*32];
    od_dering_cpu(state, ref, n, x, xstride, ln, sbx, sby, nhsb, nvsb, q, xdec, dir, pli, bskip,
     skip_stride);
    for (i = 0; i < n; i++) {
      if (memcmp(ref + i*n, src + (size_t)i*w, sizeof(int16_t)*n) != 0) {
        T.st.dering_check_fail++;
        break;
      }
    }
  }
  for (i = 0; i < n; i++) memcpy(y + (size_t)i*ystride, src + (size_t)i*w, sizeof(int16_t)*n);
  T.st.dering_dev_sbs++;
  if (pli == 0 && T.dist_valid) {
    /* the loop computes the two distortions of this superblock next (:2634-2635) */
    T.dist_sb = sby*nhsb + sbx;
    T.dist_calls = 0;
  }
  return 1;
}

/* od_compute_dist (src/encode.c:1032) for the two calls that follow a luma od_dering call of
   the on/off loop: operand 0 = the unfiltered superblock, operand 1 = the deringed one.  The
   device delivered, per 8x8 sub-block, the pow argument and the weighted energy; the
   activity power is this process's libm, the sum runs in the reference's raster order. */
int od_hipenc_dist_hook(daala_enc_ctx *enc, const od_coeff *x, const od_coeff *y, int n, int bs,
 double *dist, double (*cpu)(daala_enc_ctx *, od_coeff *, od_coeff *, int, int)) {
  const double *arg;
  const double *en;
  double calibration;
  double sum;
  int k;
  if (T.dist_sb < 0 || enc != T.enc || n != 32 || bs != 3) return 0;
  arg = T.dist[0] + (size_t)T.dist_sb*16;
  en = T.dist[1 + T.dist_calls] + (size_t)T.dist_sb*16;
  calibration = enc->use_activity_masking ? 1.95 : 1.62;      /* :997-1004 */
  sum = 0;
  for (k = 0; k < 16; k++) {
    double activity;
    activity = calibration*pow(arg[k], -1./6);                 /* :1007 */
    sum += activity*activity*en[k];                            /* :1029, :1048 */
  }
  sum *= 1.7;                                                   /* :1055 */
  if (T.check) {
    if (cpu(enc, (od_coeff *)x, (od_coeff *)y, n, bs) != sum) T.st.dist_check_fail++;
  }
  T.st.dist_dev++;
  if (++T.dist_calls == 2) T.dist_sb = -1;
  *dist = sum;
  return 1;
}

/* ------------------------------------------------------------------------ */
/* P-frame feed (include/daala_hip.h section 4d).  Called by the od_state_mc_predict binding
   (hip_dec_glue.c) on an encoder thread right after the frame's prediction exists: from here
   on both inputs of every pvq_theta call of the frame - the input's and the prediction's
   transforms at every block size - are fixed, so the device enumerates every candidate of
   every band while this thread (and a few helpers, for the libm stage) waits; the block-size
   RDO pass and the final pass then only price. */
typedef struct pf_stage_job {
  od_hip_pfeed *pf;
  int part;
  int nparts;
  int rc;
} pf_stage_job;

static void *pf_stage_thread(void *arg) {
  pf_stage_job *j;
  int pli;
  int level;
  j = (pf_stage_job *)arg;
  j->rc = 0;
  for (pli = 0; pli < 3; pli++) {
    for (level = 0; level < (pli ? 3 : 4); level++) {
      long n;
      long a;
      long b;
      n = od_hip_pfeed_nrec(j->pf, pli, level);
      if (n < 0) {
        j->rc = (int)n;
        return NULL;
      }
      a = n*j->part/j->nparts;
      b = n*(j->part + 1)/j->nparts;
      if (b > a && od_hip_pfeed_host_stage(j->pf, pli, level, a, b) != 0) j->rc = -1;
    }
  }
  return NULL;
}

static int pf_helpers = 6;        /* HIPENC_PF_THREADS: threads of the feed's libm stage */

int od_hipenc_pframe_feed(od_state *state, od_img *pred) {
  daala_enc_ctx *enc;
  const unsigned char *pin[3];
  const unsigned char *ppr[3];
  int sin_[3];
  int spr[3];
  int pli;
  int level;
  double t0;
  enc = T.enc;
  T.pf_valid = 0;
  if (T.time_cpu && enc != NULL && state == &enc->state) T.st.pre_mc_s += now_s() - T.t_frame0;
  if (T.pf == NULL || enc == NULL || state != &enc->state || !T.host_pvq
   || state->frame_type != OD_P_FRAME || state->info.nplanes != 3
   || enc->use_haar_wavelet || enc->quality[0] == 0) {
    return 0;
  }
  t0 = now_s();
  for (pli = 0; pli < 3; pli++) {
    const od_img_plane *ip;
    int dec;
    ip = &enc->input_img[enc->curr_frame].planes[pli];
    dec = pli > 0;
    if (ip->xstride != 1 || pred->planes[pli].xstride != 1) return 0;
    pin[pli] = ip->data;
    sin_[pli] = ip->ystride;
    ppr[pli] = pred->planes[pli].data;
    spr[pli] = pred->planes[pli].ystride;
    for (level = 0; level < (dec ? 3 : 4); level++) {
      int32_t q[11];
      double beta[11];
      int bs;
      int nb;
      int b;
      bs = (dec ? 2 : 3) - level;
      nb = OD_BAND_OFFSETS[bs][0];
      for (b = 0; b < 11; b++) {
        q[b] = 1;
        beta[b] = 1;
      }
      for (b = 0; b < nb; b++) {
        /* src/pvq_encoder.c:712, src/encode.c:1187 */
        q[b] = OD_MAXI(1, OD_MAXI(1, state->quantizer[pli])*state->pvq_qm_q4[pli][od_qm_get_index(bs, b + 1)] >> 4);
        beta[b] = OD_PVQ_BETA[enc->use_activity_masking][pli][bs][b];
      }
      if (od_hip_pfeed_set_level(T.pf, pli, level, state->qm + od_qm_offset(bs, dec), q, beta) != 0) return -1;
    }
  }
  if (od_hip_pfeed_gains(T.pf, pin, sin_, ppr, spr) != 0) return -1;
  {
    pthread_t th[16];
    int started[16];
    pf_stage_job jobs[16];
    int n;
    int i;
    int bad;
    n = pf_helpers < 1 ? 1 : pf_helpers > 16 ? 16 : pf_helpers;
    for (i = 0; i < n; i++) {
      jobs[i].pf = T.pf;
      jobs[i].part = i;
      jobs[i].nparts = n;
      jobs[i].rc = 0;
    }
    for (i = 1; i < n; i++) {
      started[i] = pthread_create(&th[i], NULL, pf_stage_thread, &jobs[i]) == 0;
      /* no thread: this one does that part too */
      if (!started[i]) pf_stage_thread(&jobs[i]);
    }
    pf_stage_thread(&jobs[0]);
    bad = jobs[0].rc;
    for (i = 1; i < n; i++) {
      if (started[i]) pthread_join(th[i], NULL);
      bad |= jobs[i].rc;
    }
    if (bad) return -1;
  }
  if (od_hip_pfeed_search(T.pf) != 0) return -1;
  for (pli = 0; pli < 3; pli++) {
    for (level = 0; level < (pli ? 3 : 4); level++) {
      if (od_hip_pfeed_view(T.pf, pli, level, &T.pfv[pli][level]) != 0) return -1;
    }
  }
  T.pf_valid = 1;
  T.st.pfeed_frames++;
  T.st.t_pfeed_s += now_s() - t0;
  return 1;
}

/* The batch stage of od_mv_est (mcenc_tail.c: od_mv_est_calc_sads): every block SAD of the
   fixed vector grid in one device call on this worker's prediction object (hip_dec_glue.c).
   1: sad[] written; 0: no device on this thread, the reference's loop runs; < 0: failed frame. */
int od_hipdec_mc_sad_items(od_state *state, const od_img *input, int nplanes,
 const od_hip_mc_sad_item *items, int nitems, int32_t *sad);
static int mv_dev_sads = 1;       /* HIPENC_MV_SADS=0: leave od_mv_est_calc_sads on the host */

int od_hipenc_mv_sad_items(daala_enc_ctx *enc, int nplanes, const od_hip_mc_sad_item *items,
 int nitems, int32_t *sad) {
  double t0;
  int rc;
  if (!mv_dev_sads || enc == NULL || enc != T.enc) return 0;
  t0 = now_s();
  rc = od_hipdec_mc_sad_items(&enc->state, enc->input_img + enc->curr_frame, nplanes, items, nitems, sad);
  if (rc > 0) {
    T.st.mv_dev_calls++;
    T.st.mv_dev_sads += nitems;
    T.st.mv_dev_wait_s += now_s() - t0;
  }
  return rc;
}

/* The EPZS initialisation's block-matching windows (mcenc_tail.c: od_mv_est_init_mvs) on this
   worker's prediction object.  nrec == 0 asks whether a device serves this thread at all. */
int od_hipdec_mc_bma_windows(od_state *state, const od_img *input, int nplanes,
 const od_hip_mc_bma_rec *recs, int nrec, int radius, int32_t *out);
static int mv_dev_epzs = 1;       /* HIPENC_MV_EPZS=0: the reference's od_mv_est_init_mvs, host SADs;
                                     2: the level-by-level walk even without a device (host SADs): the CPU
                                     tests pin the walk's equivalence to the reference's block-by-block one */

int od_hipenc_mv_bma_windows(daala_enc_ctx *enc, int nplanes, const od_hip_mc_bma_rec *recs, int nrec,
 int radius, int32_t *out) {
  double t0;
  int rc;
  if (!mv_dev_epzs || enc == NULL || enc != T.enc) return 0;
  if (nrec == 0) return od_hipenc_device_thread() || mv_dev_epzs == 2;
  t0 = now_s();
  rc = od_hipdec_mc_bma_windows(&enc->state, enc->input_img + enc->curr_frame, nplanes, recs, nrec, radius, out);
  if (rc > 0) {
    T.st.mv_bma_calls++;
    T.st.mv_bma_windows += nrec;
    T.st.mv_dev_wait_s += now_s() - t0;
  }
  return rc;
}

void od_hipenc_mv_bma_stats(long hits, long misses) {
  T.st.mv_level_walks++;
  T.st.mv_bma_hits += hits;
  T.st.mv_bma_misses += misses;
}

/* check mode: the reference's loop ran beside the device call and its sad_cache differs */
void od_hipenc_mv_check_fail(long n) {
  T.st.mv_check_fail += n;
}

/* stage timers of od_mv_est (mcenc_tail.c) */
void od_hipenc_mv_stage(int stage, double seconds) {
  if (stage >= 0 && stage < 8) T.st.mv_stage_s[stage] += seconds;
}

#define FDCT_MIN_BS_DEFAULT (1)     /* 4x4 blocks: four cache misses cost more than the transform */

/* fdct_2d entries of the worker's vtable (struct od_state_opt_vtbl, src/state.h:106).
   For keyframe luma every forward transform the encoder asks for - od_block_encode in
   the block-size RDO pass (src/encode.c:1139) and od_compute_dcts (:1308) - has the
   lapped picture + the split filters of its ancestors as input and writes into the
   dtmp[0] plane at the block's position: exactly one block of the device's forward
   pyramid, which the feed carries.  Anything else (chroma, od_compute_dist's 8x8
   error transform into a stack buffer) runs the context's C transform.  P frames are NOT
   served: their input plane is not the picture's transform near the padding - the encoder
   overwrites the padded samples of ctmp with the prediction's (src/encode.c:2443-2457). */
/* smallest block size served from the feed (HIPENC_FDCT_MIN_BS, read by od_hipenc_open
   before any worker exists) */
static int fdct_min_bs = FDCT_MIN_BS_DEFAULT;

static void fdct_from_feed(int bs, od_coeff *y, int ystride, const od_coeff *x,
 int xstride) {
  if (bs >= fdct_min_bs && T.lev != NULL && T.enc != NULL
   && T.enc->state.frame_type == OD_I_FRAME) {
    const od_state *st;
    od_coeff *d0;
    const od_hip_feed_level *L;
    int w;
    int h;
    int pli;
    st = &T.enc->state;
    /* which plane's dtmp the block is written into: luma, or a chroma plane of the feed */
    L = NULL;
    d0 = NULL;
    w = h = 0;
    for (pli = 0; pli < 3; pli++) {
      w = st->frame_width >> (pli > 0);
      h = st->frame_height >> (pli > 0);
      d0 = st->dtmp[pli];
      if (d0 != NULL && y >= d0 && y < d0 + (size_t)w*h) {
        if (pli == 0) L = &T.lev[3 - bs];
        else if (T.levc[pli - 1] != NULL && bs <= 2) L = &T.levc[pli - 1][2 - bs];
        break;
      }
    }
    if (L != NULL && ystride == w) {
      size_t off;
      int n;
      int yy;
      int xx;
      off = (size_t)(y - d0);
      yy = (int)(off/w);
      xx = (int)(off%w);
      n = 4 << bs;
      if (L->lev != NULL && L->n == n && (yy & (n - 1)) == 0 && (xx & (n - 1)) == 0) {
        const od_coeff *src;
        int i;
        src = L->lev + (size_t)yy*L->lev_stride + xx;
        if (T.check) {
          od_coeff tmp[32*32];
          (*T.fdct_cpu[bs])(tmp, n, x, xstride);
          for (i = 0; i < n; i++) {
            if (memcmp(tmp + i*n, src + (size_t)i*L->lev_stride, sizeof(od_coeff)*n) != 0) {
              T.st.fdct_check_fail++;
              break;
            }
          }
        }
        for (i = 0; i < n; i++) {
          memcpy(y + (size_t)i*ystride, src + (size_t)i*L->lev_stride, sizeof(od_coeff)*n);
        }
        T.st.fdct_hits++;
        return;
      }
    }
  }
  (*T.fdct_cpu[bs])(y, ystride, x, xstride);
}

#define FDCT_HOOK(name, bs) \
  static void name(od_coeff *y, int ystride, const od_coeff *x, int xstride) { \
    fdct_from_feed(bs, y, ystride, x, xstride); \
  }
FDCT_HOOK(hook_fdct4, 0)
FDCT_HOOK(hook_fdct8, 1)
FDCT_HOOK(hook_fdct16, 2)
FDCT_HOOK(hook_fdct32, 3)

/* od_encode_checkpoint / od_encode_rollback (src/encode.c:600-608) as the block-size RDO
   recursion calls them (several times per non-leaf block, :1124, :1571-1642): 12 KB of the
   19.7 KB adaptation context are the Haar-wavelet CDFs, which only lossless frames touch
   (od_wavelet_quantize, :798-898; mbctx.use_haar_wavelet, :3002) - on lossy frames neither
   copy can differ there, so they are left out. */
#define ADAPT_HEAD (offsetof(od_adapt_ctx, haar_coeff_cdf))
#define ADAPT_TAIL (offsetof(od_adapt_ctx, clpf_cdf))
static int lossy_frame(const daala_enc_ctx *enc) {
  return !enc->use_haar_wavelet && enc->quality[0] != 0;
}

void od_encode_checkpoint(const daala_enc_ctx *enc, od_rollback_buffer *rbuf) {
  od_ec_enc_checkpoint(&rbuf->ec, &enc->ec);
  if (lossy_frame(enc)) {
    memcpy(&rbuf->adapt, &enc->state.adapt, ADAPT_HEAD);
    memcpy((char *)&rbuf->adapt + ADAPT_TAIL, (const char *)&enc->state.adapt + ADAPT_TAIL,
     sizeof(od_adapt_ctx) - ADAPT_TAIL);
  }
  else OD_COPY(&rbuf->adapt, &enc->state.adapt, 1);
}

void od_encode_rollback(daala_enc_ctx *enc, const od_rollback_buffer *rbuf) {
  od_ec_enc_rollback(&enc->ec, &rbuf->ec);
  if (lossy_frame(enc)) {
    memcpy(&enc->state.adapt, &rbuf->adapt, ADAPT_HEAD);
    memcpy((char *)&enc->state.adapt + ADAPT_TAIL, (const char *)&rbuf->adapt + ADAPT_TAIL,
     sizeof(od_adapt_ctx) - ADAPT_TAIL);
  }
  else OD_COPY(&enc->state.adapt, &rbuf->adapt, 1);
}

/* od_haar (src/dct.c:1960) as the encoder calls it for lossless frames (quantizer 0): every
   whole superblock of ctmp[pli] into dtmp[pli] (od_compute_dcts, src/encode.c:1305; the
   block-size RDO pass and inter frames also come here, :1129).  With a device the three
   Haar planes of the frame are already on the host (od_hip_enc_feed_run_lossless): a call
   that transforms a whole superblock of the current frame into its dtmp position is a
   block copy.  The build renames the reference's definition to od_haar_cpu. */
void od_haar_cpu(od_coeff *y, int ystride, const od_coeff *x, int xstride, int ln);
void od_hipdec_haar_notify(void);        /* hip_dec_glue.c */
int od_hipdec_take_failure(void);
void od_hipdec_thread_cleanup(void);
void od_hipenc_mv_thread_cleanup(void);   /* mcenc_tail.c */
void od_hipdec_set_device(int device);

void od_haar(od_coeff *y, int ystride, const od_coeff *x, int xstride, int ln) {
  /* decoder threads: a Haar frame with a quantizer > 0 takes the host path (hip_dec_glue.c) */
  if (T.enc == NULL) od_hipdec_haar_notify();
  if (T.haar[0] != NULL && T.enc != NULL && T.enc->state.frame_type == OD_I_FRAME) {
    const od_state *st;
    int pli;
    st = &T.enc->state;
    for (pli = 0; pli < 3; pli++) {
      const od_coeff *d0;
      int w;
      int h;
      int dec;
      dec = pli > 0;
      d0 = st->dtmp[pli];
      w = st->frame_width >> dec;
      h = st->frame_height >> dec;
      if (ystride == w && y >= d0 && y < d0 + (size_t)w*h && ln == 5 - dec
       && T.haar_stride[pli] == w) {
        size_t off;
        int n;
        int yy;
        int xx;
        int i;
        off = (size_t)(y - d0);
        yy = (int)(off/w);
        xx = (int)(off%w);
        n = 1 << ln;
        if ((yy & (n - 1)) == 0 && (xx & (n - 1)) == 0) {
          const od_coeff *src;
          src = T.haar[pli] + (size_t)yy*w + xx;
          if (T.check) {
            od_coeff tmp[32*32];
            od_haar_cpu(tmp, n, x, xstride, ln);
            for (i = 0; i < n; i++) {
              if (memcmp(tmp + i*n, src + (size_t)i*w, sizeof(od_coeff)*n) != 0) {
                T.st.fdct_check_fail++;
                break;
              }
            }
          }
          for (i = 0; i < n; i++) memcpy(y + (size_t)i*ystride, src + (size_t)i*w, sizeof(od_coeff)*n);
          T.st.haar_hits++;
          return;
        }
      }
    }
  }
  od_haar_cpu(y, ystride, x, xstride, ln);
}

/* od_pvq_encode as od_block_encode calls it (src/encode.c:1187): keyframes go through
   hip_pvq_host.c (the feed consumer), inter frames through the reference's definition. */
int od_pvq_encode(daala_enc_ctx *enc, od_coeff *ref, od_coeff *in, od_coeff *out,
 int q0, int pli, int bs, const double *beta, int robust, int is_keyframe,
 int q_scaling, int bx, int by, const int16_t *qm, const int16_t *qm_inv) {
  T.pli = pli;
  if (T.host_pvq) {
    return od_hip_pvq_encode_host(enc, ref, in, out, q0, pli, bs, beta, robust, is_keyframe,
     q_scaling, bx, by, qm, qm_inv);
  }
  return od_pvq_encode_cpu(enc, ref, in, out, q0, pli, bs, beta, robust, is_keyframe,
   q_scaling, bx, by, qm, qm_inv);
}

/* ------------------------------------------------------------------------ */
static daala_enc_ctx *make_encoder(const od_hipenc_params *p, int w, int h) {
  daala_info di;
  daala_enc_ctx *enc;
  int v;
  daala_info_init(&di);
  di.pic_width = w;
  di.pic_height = h;
  di.nplanes = 3;
  di.plane_info[0].xdec = di.plane_info[0].ydec = 0;
  di.plane_info[1].xdec = di.plane_info[1].ydec = 1;
  di.plane_info[2].xdec = di.plane_info[2].ydec = 1;
  di.timebase_numerator = 30;
  di.timebase_denominator = 1;
  di.frame_duration = 1;
  di.pixel_aspect_numerator = di.pixel_aspect_denominator = 1;
  di.bitdepth_mode = OD_BITDEPTH_MODE_8;
  di.keyframe_rate = p->keyframe_rate > 1 ? p->keyframe_rate : 1;
  enc = daala_encode_create(&di);
  if (enc == NULL) return NULL;
  v = p->quant;
  daala_encode_ctl(enc, OD_SET_QUANT, &v, sizeof(v));
  v = p->complexity;
  daala_encode_ctl(enc, OD_SET_COMPLEXITY, &v, sizeof(v));
  v = p->masking;
  daala_encode_ctl(enc, OD_SET_ACTIVITY_MASKING, &v, sizeof(v));
  /* use_dering has no default inside the library (only src/encode.c:527 writes
     it); the reference CLI always sets 1 (examples/encoder_example.c:675,900). */
  v = 1;
  daala_encode_ctl(enc, OD_SET_DERING, &v, sizeof(v));
  /* Test hook: the stream syntax allows Haar-wavelet frames with a quantizer > 0 (the flag of
     src/encode.c:3022), which the reference encoder only writes when built with
     OD_USE_HAAR_WAVELET; the decoder glue's handling of such frames needs one. */
  if (getenv("HIPENC_TEST_HAAR") != NULL) enc->use_haar_wavelet = 1;
  return enc;
}

static void fill_img(od_img *img, const unsigned char *base, int w, int h) {
  int cw;
  int ch;
  int pli;
  cw = (w + 1) >> 1;
  ch = (h + 1) >> 1;
  memset(img, 0, sizeof(*img));
  img->nplanes = 3;
  img->width = w;
  img->height = h;
  img->planes[0].data = (unsigned char *)base;
  img->planes[1].data = (unsigned char *)base + (size_t)w*h;
  img->planes[2].data = (unsigned char *)base + (size_t)w*h + (size_t)cw*ch;
  for (pli = 0; pli < 3; pli++) {
    img->planes[pli].xdec = img->planes[pli].ydec = pli > 0;
    img->planes[pli].xstride = 1;
    img->planes[pli].ystride = pli ? cw : w;
    img->planes[pli].bitdepth = 8;
  }
}

int od_hipenc_level_params_plane(const od_hipenc_params *p, int pli, int16_t qm[4][1024],
 int32_t q[4][11], double beta[4][11]);
int od_hipenc_level_params(const od_hipenc_params *p, int16_t qm[4][1024],
 int32_t q[4][11], double beta[4][11]) {
  return od_hipenc_level_params_plane(p, 0, qm, q, beta);
}

/* The same for plane pli of a 4:2:0 stream: level l of a chroma plane holds its (16 >> l)-sized
   blocks (three levels; the fourth entry is left zero). */
int od_hipenc_level_params_plane(const od_hipenc_params *p, int pli, int16_t qm[4][1024],
 int32_t q[4][11], double beta[4][11]) {
  unsigned char frame[64*64 + 2*32*32];
  daala_enc_ctx *enc;
  daala_packet dp;
  od_img img;
  int left;
  int l;
  if (p == NULL) return OD_HIP_EFAULT;
  if (pli < 0 || pli > 2) return OD_HIP_EINVAL;
  /* state.qm / pvq_qm_q4 / quantizer are filled while the first frame is coded
     (src/encode.c:3025-3050): code one flat 64x64 frame and read them back. */
  memset(frame, 128, sizeof(frame));
  enc = make_encoder(p, 64, 64);
  if (enc == NULL) return OD_HIP_EINVAL;
  fill_img(&img, frame, 64, 64);
  if (daala_encode_img_in(enc, &img, 0, 0, &left) < 0) {
    daala_encode_free(enc);
    return OD_HIP_EINVAL;
  }
  while (daala_encode_packet_out(enc, 0, &dp) > 0);
  for (l = 0; l < 4; l++) {
    int bs;
    int n;
    int nb;
    int b;
    bs = (pli > 0 ? 2 : 3) - l;
    memset(qm[l], 0, sizeof(qm[l]));
    for (b = 0; b < 11; b++) {
      q[l][b] = 1;
      beta[l][b] = 1;
    }
    if (bs < 0) continue;
    n = 4 << bs;
    nb = OD_BAND_OFFSETS[bs][0];
    memcpy(qm[l], enc->state.qm + od_qm_offset(bs, pli > 0), sizeof(int16_t)*n*n);
    for (b = 0; b < nb; b++) {
      /* src/pvq_encoder.c:712 */
      q[l][b] = OD_MAXI(1, enc->state.quantizer[pli]
       *enc->state.pvq_qm_q4[pli][od_qm_get_index(bs, b + 1)] >> 4);
      beta[l][b] = OD_PVQ_BETA[enc->use_activity_masking][pli][bs][b];
    }
  }
  daala_encode_free(enc);
  return 0;
}

/* ------------------------------------------------------------------------ */
/* The pipelined multi-frame driver.  A session (od_hipenc_open) owns N host workers - one
   reference encoder context, one device deringing object each - and, with a device, one
   od_hip_ctx + encoder feed with `nslots` frame slots.  A job (od_hipenc_encode) is one
   stream of independent keyframes: frame f lives in slot f % nslots; a stream that does
   not fit the slots is cut into two half-buffers of nslots/2 frames, batch k + 1 is
   uploaded, computed and copied while batch k is being coded. */
typedef struct job {
  int nframes;
  long frame0;          /* stream index of frames[0] (golden-frame flag, see encode_frame) */
  const unsigned char *frames;
  const od_hip_feed_level *views;
  int batch;
  int nslots;
  int uploaded;         /* frames uploaded in the current batch */
  int batch0;           /* first frame of the current device batch */
  int batch_n;
  int next_upload;
  int next_compand;     /* frames of the current batch whose gains are on the host: claimable for companding */
  int compand_upto;
  int companded;
  int next_encode;
  int encoded;          /* frames completely coded */
  unsigned char *done;  /* per frame: completely coded */
  int done_prefix;      /* frames [0, done_prefix) are all coded */
  int launched_upto;    /* frames [0, launched_upto) have a feed run enqueued */
  int go;               /* 1: claimable */
  int failed;
  int workers_done;
  /* outputs */
  unsigned char **pkt;
  long *pkt_len;
  od_hipenc_stats st;
} job;

struct od_hipenc {
  od_hipenc_params p;
  int use_device;
  int device;
  int nw;
  int nslots;
  size_t frame_bytes;
  pthread_t *th;
  od_hip_ctx *ctx;
  od_hip_enc_feed *feed;
  pthread_mutex_t mu;
  pthread_cond_t cv;
  int encoders_ready;
  int setup_failed;
  int quit;
  int next_worker_id;
  int host_pvq;         /* keyframe od_pvq_encode: 1 hip_pvq_host.c (default), 0 the reference's */
  int time_cpu;         /* per-call timers around the C searches (HIPENC_TIME=1) */
  int chroma_feed;      /* the chroma planes are in the device feed too (keyframes) */
  int sample_every;     /* HIPENC_SAMPLE: re-search every n-th feed candidate (default 256, 0 off) */
  int pfeed_on;         /* HIPENC_PFEED (default 1): inter frames take the P-frame feed */
  long job_seq;         /* number of jobs submitted so far */
  job *J;               /* the job being worked on, or NULL */
  double t_setup_s;
};

static void add_stats(od_hipenc_stats *a, const od_hipenc_stats *b) {
  a->dev_hits += b->dev_hits;
  a->cpu_noref_luma += b->cpu_noref_luma;
  a->cpu_other += b->cpu_other;
  a->g2_mismatch += b->g2_mismatch;
  a->lost_sync += b->lost_sync;
  a->check_fail += b->check_fail;
  a->pvq_check_fail += b->pvq_check_fail;
  a->resampled += b->resampled;
  a->search_cpu_s += b->search_cpu_s;
  a->fdct_hits += b->fdct_hits;
  a->haar_hits += b->haar_hits;
  a->fdct_check_fail += b->fdct_check_fail;
  a->dering_dev_sbs += b->dering_dev_sbs;
  a->dering_check_fail += b->dering_check_fail;
  a->dist_dev += b->dist_dev;
  a->dist_check_fail += b->dist_check_fail;
  a->pfeed_frames += b->pfeed_frames;
  a->t_pfeed_s += b->t_pfeed_s;
  a->rate_s += b->rate_s;
  a->rate_state_free_s += b->rate_state_free_s;
  a->rate_calls += b->rate_calls;
  a->frame_cpu_s += b->frame_cpu_s;
  a->pre_mc_s += b->pre_mc_s;
  {
    int i;
    for (i = 0; i < 8; i++) a->mv_stage_s[i] += b->mv_stage_s[i];
    a->mv_dev_calls += b->mv_dev_calls;
    a->mv_dev_sads += b->mv_dev_sads;
    a->mv_dev_wait_s += b->mv_dev_wait_s;
    a->mv_check_fail += b->mv_check_fail;
    a->mv_bma_calls += b->mv_bma_calls;
    a->mv_bma_windows += b->mv_bma_windows;
    a->mv_bma_hits += b->mv_bma_hits;
    a->mv_bma_misses += b->mv_bma_misses;
    a->mv_level_walks += b->mv_level_walks;
  }
  for (int i = 0; i < 4; i++) a->search_class_s[i] += b->search_class_s[i];
}

static int upload_frame(od_hipenc *S, job *J, daala_enc_ctx *enc, int f) {
  od_img img;
  od_img *pad;
  const unsigned char *planes[3];
  int ystride[3];
  int pli;
  fill_img(&img, J->frames + S->frame_bytes*f, S->p.pic_width, S->p.pic_height);
  /* the reference's own padding (od_img_copy_pad, src/encode.c:1728) */
  od_hipenc_copy_pad(enc, &img);
  pad = &enc->input_img[0];
  for (pli = 0; pli < 3; pli++) {
    planes[pli] = pad->planes[pli].data;
    ystride[pli] = pad->planes[pli].ystride;
  }
  return od_hip_upload_planes(S->ctx, f % J->nslots, planes, ystride);
}

static int encode_frame(od_hipenc *S, job *J, daala_enc_ctx *enc, int f) {
  od_img img;
  daala_packet dp;
  int left;
  od_hip_feed_level lev[4];
  od_hip_feed_level levc[2][4];
  T.lev = NULL;
  T.levc[0] = T.levc[1] = NULL;
  T.haar[0] = T.haar[1] = T.haar[2] = NULL;
  if (J->views != NULL) T.lev = J->views + 4*(size_t)f;
  else if (S->feed != NULL && S->p.quant == 0) {
    if (od_hip_enc_feed_haar_view(S->feed, f % J->nslots, T.haar, T.haar_stride) != 0) return -1;
  }
  else if (S->feed != NULL) {
    if (od_hip_enc_feed_view(S->feed, f % J->nslots, lev) != 0) return -1;
    T.lev = lev;
    if (S->chroma_feed) {
      if (od_hip_enc_feed_view_plane(S->feed, f % J->nslots, 1, levc[0]) != 0
       || od_hip_enc_feed_view_plane(S->feed, f % J->nslots, 2, levc[1]) != 0) {
        return -1;
      }
      T.levc[0] = levc[0];
      T.levc[1] = levc[1];
    }
  }
  /* Frame f of the stream on a context that did not code frames 0..f-1: the
     only history a keyframe packet carries is the golden-frame flag
     (ip_frame_count % OD_GOLDEN_FRAME_INTERVAL, forced on while no golden
     reference exists; src/encode.c:2958-2963, :3023). */
  if (S->p.keyframe_rate <= 1) {
    enc->ip_frame_count = (int)(J->frame0 + f);
    if (J->frame0 + f > 0 && enc->state.ref_imgi[OD_FRAME_GOLD] < 0) {
      enc->state.ref_imgi[OD_FRAME_GOLD] = 0;
      enc->state.ref_imgi[OD_FRAME_PREV] = 0;
    }
  }
  fill_img(&img, J->frames + S->frame_bytes*f, S->p.pic_width, S->p.pic_height);
  T.enc = enc;
  T.dr_valid = 0;
  T.dist_valid = 0;
  T.dist_sb = -1;
  T.pf_valid = 0;
  od_hipenc_mc_cache_flush();          /* reference frames change between frames */
  (void)od_hipdec_take_failure();
  T.t_frame0 = now_s();
  if (daala_encode_img_in(enc, &img, 0, 0, &left) < 0) return -2;
  if (T.dr_error) return -4;
  /* od_state_mc_predict's device pass (hip_dec_glue.c) failed for this frame */
  if (od_hipdec_take_failure()) return -5;
  J->pkt_len[f] = 0;
  while (daala_encode_packet_out(enc, 0, &dp) > 0) {
    unsigned char *q;
    q = (unsigned char *)realloc(J->pkt[f], J->pkt_len[f] + 4 + dp.bytes);
    if (q == NULL) return -3;
    J->pkt[f] = q;
    q += J->pkt_len[f];
    q[0] = dp.bytes & 255;
    q[1] = (dp.bytes >> 8) & 255;
    q[2] = (dp.bytes >> 16) & 255;
    q[3] = (dp.bytes >> 24) & 255;
    memcpy(q + 4, dp.packet, dp.bytes);
    J->pkt_len[f] += 4 + dp.bytes;
  }
  if (T.time_cpu) T.st.frame_cpu_s += now_s() - T.t_frame0;
  T.lev = NULL;
  T.levc[0] = T.levc[1] = NULL;
  T.haar[0] = T.haar[1] = T.haar[2] = NULL;
  T.enc = NULL;
  T.pf_valid = 0;
  return 0;
}

/* Optional placement of the workers: HIPENC_PIN=<stride> pins worker i to the
   (i*stride)-th CPU of the process's affinity mask, so that workers do not end up as
   SMT siblings of one core or crowd one L3 slice when the scheduler is free to put
   them anywhere on a 256-thread host.  Unset: the scheduler decides. */
static void pin_worker(int idx) {
  const char *e;
  int stride;
  cpu_set_t all;
  cpu_set_t one;
  int c;
  int seen;
  e = getenv("HIPENC_PIN");
  if (e == NULL) return;
  stride = atoi(e);
  if (stride < 1) return;
  if (sched_getaffinity(0, sizeof(all), &all) != 0) return;
  seen = 0;
  for (c = 0; c < CPU_SETSIZE; c++) {
    if (!CPU_ISSET(c, &all)) continue;
    if (seen == idx*stride) {
      CPU_ZERO(&one);
      CPU_SET(c, &one);
      pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
      return;
    }
    seen++;
  }
}

/* One reference encoder context with the worker's vtable entries installed. */
static daala_enc_ctx *worker_encoder(const od_hipenc *S) {
  static const od_dct_func_2d hooks[OD_NBSIZES] = {hook_fdct4, hook_fdct8, hook_fdct16,
   hook_fdct32};
  static const od_dct_func_2d vfdct[OD_NBSIZES] = {od_hipenc_fdct4x4, od_hipenc_fdct8x8,
   od_hipenc_fdct16x16, od_hipenc_fdct32x32};
  static const od_dct_func_2d vidct[OD_NBSIZES] = {od_hipenc_idct4x4, od_hipenc_idct8x8,
   od_hipenc_idct16x16, od_hipenc_idct32x32};
  daala_enc_ctx *enc;
  int i;
  enc = make_encoder(&S->p, S->p.pic_width, S->p.pic_height);
  if (enc == NULL) return NULL;
  for (i = 0; i < OD_NBSIZES; i++) {
    /* the transforms that stay on the host run on its vector unit (hip_dct_host.c); check
       mode keeps the reference's C functions: they are the checker there */
    T.fdct_cpu[i] = T.check ? enc->state.opt_vtbl.fdct_2d[i] : vfdct[i];
    enc->state.opt_vtbl.fdct_2d[i] = hooks[i];
    if (!T.check) enc->state.opt_vtbl.idct_2d[i] = vidct[i];
  }
  /* the motion search's per-block leaves (hip_mc_host.c) */
  enc->state.opt_vtbl.mc_blend_full = od_hipenc_mc_blend_full8;
  enc->state.opt_vtbl.mc_blend_full_split = od_hipenc_mc_blend_full_split8;
  enc->state.opt_vtbl.mc_predict1fmv = od_hipenc_mc_predict1fmv8;
  return enc;
}

static void *worker(void *arg) {
  od_hipenc *S;
  daala_enc_ctx *enc;
  long seen_seq;
  S = (od_hipenc *)arg;
  memset(&T, 0, sizeof(T));
  pthread_mutex_lock(&S->mu);
  pin_worker(S->next_worker_id++);
  pthread_mutex_unlock(&S->mu);
  od_hipdec_set_device(S->device);
  T.check = S->p.check;
  T.time_cpu = S->time_cpu;
  T.host_pvq = S->host_pvq;
  T.sample_every = S->sample_every;
  enc = worker_encoder(S);
  if (enc != NULL && S->ctx != NULL && S->p.keyframe_rate > 1 && S->pfeed_on) {
    od_hip_geometry g;
    memset(&g, 0, sizeof(g));
    g.pic_width = S->p.pic_width;
    g.pic_height = S->p.pic_height;
    g.frame_width = enc->state.frame_width;
    g.frame_height = enc->state.frame_height;
    g.nplanes = 3;
    g.xdec[1] = g.xdec[2] = 1;
    g.nslots = 2;
    T.pf = od_hip_pfeed_create(S->device, &g);      /* NULL: inter frames are searched on the host */
  }
  if (enc != NULL && S->ctx != NULL) {
    int xdec[3] = {0, 1, 1};
    int pli;
    T.dr = od_hip_dering_create(S->device, enc->state.frame_width, enc->state.frame_height, 3, xdec);
    {
      size_t nsub;
      int k;
      nsub = (size_t)(enc->state.frame_width/32)*(enc->state.frame_height/32)*16;
      for (k = 0; k < 3; k++) T.dist[k] = (double *)malloc(sizeof(double)*nsub);
      if (T.dist[0] == NULL || T.dist[1] == NULL || T.dist[2] == NULL) {
        free(T.dist[0]);
        T.dist[0] = NULL;        /* no distortions from the device: the reference's code runs */
      }
    }
    for (pli = 0; pli < 3; pli++) {
      T.dr_out[pli] = (int16_t *)malloc(sizeof(int16_t)*(size_t)(enc->state.frame_width >> (pli > 0))
       *(enc->state.frame_height >> (pli > 0)));
      if (T.dr_out[pli] == NULL && T.dr != NULL) {
        od_hip_dering_destroy(T.dr);
        T.dr = NULL;
      }
    }
  }
  seen_seq = 0;
  pthread_mutex_lock(&S->mu);
  if (enc == NULL || (S->ctx != NULL && T.dr == NULL)) S->setup_failed = 1;
  S->encoders_ready++;
  pthread_cond_broadcast(&S->cv);
  for (;;) {
    job *J;
    while (!S->quit && (S->J == NULL || S->job_seq == seen_seq)) pthread_cond_wait(&S->cv, &S->mu);
    if (S->quit) break;
    J = S->J;
    /* One stream per call (hip_enc_glue.h).  With inter frames the encoder context carries
       the GOP position and the reference frames of the stream it coded last: a new job gets
       a fresh context, so its first frame is a keyframe and nothing refers to another
       stream.  Keyframe-only sessions keep theirs (encode_frame sets the only history a
       keyframe packet carries). */
    if (S->p.keyframe_rate > 1 && seen_seq != 0 && enc != NULL) {
      daala_enc_ctx *fresh;
      pthread_mutex_unlock(&S->mu);
      fresh = worker_encoder(S);
      pthread_mutex_lock(&S->mu);
      if (fresh != NULL) {
        daala_encode_free(enc);
        enc = fresh;
      }
      else J->failed = 1;
    }
    seen_seq = S->job_seq;
    memset(&T.st, 0, sizeof(T.st));
    T.dr_error = 0;            /* a failed device pass fails ITS job, not every later one */
    for (;;) {
      int f;
      if (J->failed || S->setup_failed) break;
      if (J->go >= 1 && S->ctx != NULL && J->next_upload < J->batch0 + J->batch_n) {
        int rc;
        f = J->next_upload++;
        pthread_mutex_unlock(&S->mu);
        rc = upload_frame(S, J, enc, f);
        pthread_mutex_lock(&S->mu);
        if (rc != 0) J->failed = 1;
        J->uploaded++;
        pthread_cond_broadcast(&S->cv);
        continue;
      }
      if (J->go >= 1 && S->ctx != NULL && J->next_compand < J->compand_upto) {
        int rc;
        /* cg = od_gain_compand(g) of every band of this frame with this process's libm
           (the one transcendental of the feed; include/daala_hip.h section 4) */
        f = J->next_compand++;
        pthread_mutex_unlock(&S->mu);
        rc = od_hip_enc_feed_compand(S->feed, f % J->nslots);
        pthread_mutex_lock(&S->mu);
        if (rc != 0) J->failed = 1;
        J->companded++;
        pthread_cond_broadcast(&S->cv);
        continue;
      }
      if (J->go >= 1 && J->next_encode < J->launched_upto) {
        int rc;
        f = J->next_encode++;
        pthread_mutex_unlock(&S->mu);
        rc = encode_frame(S, J, enc, f);
        pthread_mutex_lock(&S->mu);
        if (rc != 0) J->failed = 1;
        J->encoded++;
        J->done[f] = 1;
        while (J->done_prefix < J->nframes && J->done[J->done_prefix]) J->done_prefix++;
        pthread_cond_broadcast(&S->cv);
        continue;
      }
      if (J->go >= 1 && J->next_encode >= J->nframes) break;
      pthread_cond_wait(&S->cv, &S->mu);
    }
    add_stats(&J->st, &T.st);
    J->workers_done++;
    pthread_cond_broadcast(&S->cv);
  }
  pthread_mutex_unlock(&S->mu);
  if (T.pf != NULL) od_hip_pfeed_destroy(T.pf);
  T.pf = NULL;
  if (T.dr != NULL) od_hip_dering_destroy(T.dr);
  free(T.dr_out[0]);
  free(T.dr_out[1]);
  free(T.dr_out[2]);
  free(T.dist[0]);
  free(T.dist[1]);
  free(T.dist[2]);
  if (enc != NULL) daala_encode_free(enc);
  od_hipenc_mc_cache_free();             /* this thread's prediction cache (hip_mc_host.c) */
  od_hipdec_thread_cleanup();            /* its resident motion-compensation object (hip_dec_glue.c) */
  od_hipenc_mv_thread_cleanup();         /* the item list of the motion search's batch stage */
  return NULL;
}

void od_hipenc_close(od_hipenc *S) {
  int i;
  if (S == NULL) return;
  pthread_mutex_lock(&S->mu);
  S->quit = 1;
  pthread_cond_broadcast(&S->cv);
  pthread_mutex_unlock(&S->mu);
  for (i = 0; i < S->nw; i++) pthread_join(S->th[i], NULL);
  if (S->feed != NULL) od_hip_enc_feed_destroy(S->feed);
  if (S->ctx != NULL) od_hip_ctx_destroy(S->ctx);
  free(S->th);
  pthread_mutex_destroy(&S->mu);
  pthread_cond_destroy(&S->cv);
  free(S);
}

od_hipenc *od_hipenc_open(const od_hipenc_params *p, int use_device, int device, int *err) {
  od_hipenc *S;
  int nw;
  int i;
  int cw;
  int ch;
  int rc;
  double t0;
  rc = 0;
  if (err != NULL) *err = 0;
  if (p == NULL) {
    if (err != NULL) *err = OD_HIP_EFAULT;
    return NULL;
  }
  S = (od_hipenc *)calloc(1, sizeof(*S));
  if (S == NULL) {
    if (err != NULL) *err = OD_HIP_EFAULT;
    return NULL;
  }
  t0 = now_s();
  S->p = *p;
  {
    const char *e;
    e = getenv("HIPENC_HOST_PVQ");
    S->host_pvq = e == NULL || atoi(e) != 0;
    e = getenv("HIPENC_SAMPLE");
    S->sample_every = e != NULL ? atoi(e) : 256;
    e = getenv("HIPENC_TIME");
    S->time_cpu = e != NULL && atoi(e) != 0;
    if (S->time_cpu) od_hipenc_fine_timer_init();
    e = getenv("HIPENC_FDCT_MIN_BS");
    if (e != NULL) fdct_min_bs = atoi(e);
    e = getenv("HIPENC_PFEED");
    S->pfeed_on = e == NULL || atoi(e) != 0;
    e = getenv("HIPENC_PF_THREADS");
    if (e != NULL) pf_helpers = atoi(e);
    e = getenv("HIPENC_MV_SADS");
    mv_dev_sads = e == NULL || atoi(e) != 0;
    e = getenv("HIPENC_MV_EPZS");
    mv_dev_epzs = e == NULL ? 1 : atoi(e);
  }
  S->use_device = use_device;
  S->device = device;
  nw = p->nworkers < 1 ? 1 : p->nworkers;
  if (p->keyframe_rate > 1) nw = 1;       /* inter frames depend on their predecessors */
  S->nw = nw;
  cw = (p->pic_width + 1) >> 1;
  ch = (p->pic_height + 1) >> 1;
  S->frame_bytes = (size_t)p->pic_width*p->pic_height + 2*(size_t)cw*ch;
  /* Device slots.  batch > 0: that many; default: as many as keep the pinned host mirror
     of the feed (about 49 bytes per padded luma sample and frame: records + pyramid)
     under ~6 GB, at least two per worker. */
  if (p->batch > 0) S->nslots = p->batch;
  else {
    double per_frame;
    per_frame = 49.*((p->pic_width + 63) & ~63)*((p->pic_height + 63) & ~63);
    S->nslots = (int)(6e9/per_frame);
    if (S->nslots < 2*nw) S->nslots = 2*nw;
  }
  if (S->nslots < 2) S->nslots = 2;
  pthread_mutex_init(&S->mu, NULL);
  pthread_cond_init(&S->cv, NULL);
  S->th = (pthread_t *)calloc(nw, sizeof(*S->th));
  if (S->th == NULL) rc = OD_HIP_EFAULT;
  if (rc == 0 && use_device) {
    /* No device, no encode: the product path does not fall back to the C search. */
    od_hip_geometry g;
    int16_t qm[4][1024];
    int32_t q[4][11];
    double beta[4][11];
    int l;
    memset(&g, 0, sizeof(g));
    g.pic_width = p->pic_width;
    g.pic_height = p->pic_height;
    /* frame size as od_state_init pads it (src/state.c:372-375) */
    g.frame_width = (p->pic_width + (2*OD_BSIZE_MAX - 1)) & ~(2*OD_BSIZE_MAX - 1);
    g.frame_height = (p->pic_height + (2*OD_BSIZE_MAX - 1)) & ~(2*OD_BSIZE_MAX - 1);
    g.nplanes = 3;
    g.xdec[1] = g.xdec[2] = 1;
    g.nslots = S->nslots;
    S->ctx = od_hip_ctx_create(device, &g);
    if (S->ctx != NULL) S->feed = od_hip_enc_feed_create(S->ctx);
    if (S->ctx == NULL || S->feed == NULL) rc = OD_HIP_ENODEV;
    else if (od_hipenc_level_params(p, qm, q, beta) != 0) rc = OD_HIP_EINVAL;
    else for (l = 0; l < 4; l++) od_hip_enc_feed_set_level(S->feed, l, qm[l], q[l], beta[l]);
    /* the chroma planes' transforms and no-reference candidates travel with luma's
       (HIPENC_CHROMA_FEED=0: luma only, as before round 3) */
    {
      const char *e;
      int pli;
      e = getenv("HIPENC_CHROMA_FEED");
      S->chroma_feed = rc == 0 && p->quant != 0 && !(e != NULL && atoi(e) == 0);
      for (pli = 1; S->chroma_feed && pli < 3 && rc == 0; pli++) {
        if (od_hipenc_level_params_plane(p, pli, qm, q, beta) != 0) rc = OD_HIP_EINVAL;
        for (l = 0; l < 3 && rc == 0; l++) {
          if (od_hip_enc_feed_set_level_plane(S->feed, pli, l, qm[l], q[l], beta[l]) != 0) rc = OD_HIP_ENODEV;
        }
      }
    }
  }
  if (rc != 0) {
    S->nw = 0;
    od_hipenc_close(S);
    if (err != NULL) *err = rc;
    return NULL;
  }
  for (i = 0; i < nw; i++) pthread_create(&S->th[i], NULL, worker, S);
  pthread_mutex_lock(&S->mu);
  while (S->encoders_ready < nw) pthread_cond_wait(&S->cv, &S->mu);
  rc = S->setup_failed;
  pthread_mutex_unlock(&S->mu);
  S->t_setup_s = now_s() - t0;
  if (rc) {
    od_hipenc_close(S);
    if (err != NULL) *err = use_device ? OD_HIP_ENODEV : OD_HIP_EINVAL;
    return NULL;
  }
  return S;
}

long od_hipenc_encode(od_hipenc *S, int nframes, long frame0, const unsigned char *frames,
 const od_hip_feed_level *views, unsigned char *pkt_out, long pkt_cap,
 od_hipenc_stats *stats) {
  job J;
  int i;
  long total;
  long used;
  int nospace;
  double t0;
  if (S == NULL || frames == NULL || nframes < 1) return OD_HIP_EFAULT;
  if (views != NULL && S->ctx != NULL) return OD_HIP_EINVAL;
  nospace = 0;
  memset(&J, 0, sizeof(J));
  J.nframes = nframes;
  J.frame0 = frame0;
  J.frames = frames;
  J.views = views;
  /* frame f lives in slot f % nslots.  A stream that does not fit the slots uses two
     half-buffers; batch k (frames [k*batch, (k+1)*batch)) may be uploaded and launched
     as soon as batch k-2, the previous user of its half, is completely coded. */
  if (nframes <= S->nslots) {
    J.batch = nframes;
    J.nslots = nframes;
  }
  else {
    J.batch = S->nslots/2;
    J.nslots = 2*J.batch;
  }
  J.done = (unsigned char *)calloc(nframes, 1);
  J.pkt = (unsigned char **)calloc(nframes, sizeof(*J.pkt));
  J.pkt_len = (long *)calloc(nframes, sizeof(*J.pkt_len));
  if (J.done == NULL || J.pkt == NULL || J.pkt_len == NULL) {
    free(J.done);
    free(J.pkt);
    free(J.pkt_len);
    return OD_HIP_EFAULT;
  }
  J.st.t_setup_s = S->t_setup_s;
  pthread_mutex_lock(&S->mu);
  t0 = now_s();
  S->J = &J;
  S->job_seq++;
  if (S->ctx == NULL) {
    /* host-only modes: everything is encodable at once */
    J.launched_upto = nframes;
    J.go = 1;
    pthread_cond_broadcast(&S->cv);
  }
  else {
    int b0;
    for (b0 = 0; b0 < nframes && !J.failed; b0 += J.batch) {
      double ta;
      double tb;
      int rc;
      /* this batch's half-buffer (device slots + pinned host mirror) was last used
         by batch k-2: wait until all of that batch's frames are completely coded */
      while (!J.failed && J.done_prefix < b0 - J.batch) pthread_cond_wait(&S->cv, &S->mu);
      J.batch0 = b0;
      J.batch_n = nframes - b0 < J.batch ? nframes - b0 : J.batch;
      J.uploaded = 0;
      J.next_upload = b0;
      J.go = 1;
      ta = now_s();
      pthread_cond_broadcast(&S->cv);
      while (!J.failed && J.uploaded < J.batch_n) pthread_cond_wait(&S->cv, &S->mu);
      tb = now_s();
      J.st.t_upload_s += tb - ta;
      if (J.failed) break;
      /* ~40 launches and event waits: workers coding the previous batch must be able to
         take the lock meanwhile (frames [b0, b0 + batch_n) are not claimable before
         launched_upto moves) */
      if (S->p.quant == 0) {
        /* lossless stream: Haar planes instead of the PVQ feed, nothing to compand */
        pthread_mutex_unlock(&S->mu);
        rc = od_hip_enc_feed_run_lossless(S->feed, b0 % J.nslots, J.batch_n);
        pthread_mutex_lock(&S->mu);
        if (rc != 0) {
          J.failed = 1;
          break;
        }
        J.st.t_launch_s += now_s() - tb;
        J.launched_upto = b0 + J.batch_n;
        pthread_cond_broadcast(&S->cv);
        continue;
      }
      pthread_mutex_unlock(&S->mu);
      rc = od_hip_enc_feed_gains(S->feed, b0 % J.nslots, J.batch_n);
      pthread_mutex_lock(&S->mu);
      if (rc != 0) {
        J.failed = 1;
        break;
      }
      /* the workers compand the batch, frame by frame */
      J.companded = 0;
      J.next_compand = b0;
      J.compand_upto = b0 + J.batch_n;
      pthread_cond_broadcast(&S->cv);
      while (!J.failed && J.companded < J.batch_n) pthread_cond_wait(&S->cv, &S->mu);
      if (J.failed) break;
      J.st.t_compand_s += now_s() - tb;
      pthread_mutex_unlock(&S->mu);
      rc = od_hip_enc_feed_search(S->feed, b0 % J.nslots, J.batch_n);
      pthread_mutex_lock(&S->mu);
      if (rc != 0) {
        J.failed = 1;
        break;
      }
      J.st.t_launch_s += now_s() - tb;
      J.launched_upto = b0 + J.batch_n;
      pthread_cond_broadcast(&S->cv);
    }
    if (J.failed) J.go = 1;
    pthread_cond_broadcast(&S->cv);
  }
  while (J.workers_done < S->nw) pthread_cond_wait(&S->cv, &S->mu);
  S->J = NULL;
  pthread_mutex_unlock(&S->mu);
  J.st.t_total_s = now_s() - t0;
  /* All packets or none: a stream with a frame missing (or out of order) is corrupt, so
     the space needed is summed first and a buffer that is too small fails the call. */
  total = 0;
  used = 0;
  for (i = 0; i < nframes; i++) {
    if (J.pkt[i] != NULL) {
      total += J.pkt_len[i] - 4;
      used += J.pkt_len[i];
    }
    else J.failed = 1;
  }
  J.st.pkt_bytes_needed = used;
  if (pkt_out != NULL && used > pkt_cap) nospace = 1;
  used = 0;
  for (i = 0; i < nframes; i++) {
    if (J.pkt[i] != NULL) {
      if (pkt_out != NULL && !nospace && !J.failed) {
        memcpy(pkt_out + used, J.pkt[i], J.pkt_len[i]);
        used += J.pkt_len[i];
      }
      free(J.pkt[i]);
    }
  }
  free(J.pkt);
  free(J.pkt_len);
  free(J.done);
  if (stats != NULL) *stats = J.st;
  if (J.failed) return OD_HIP_EINVAL;
  return nospace ? OD_HIP_ENOSPC : total;
}

long od_hipenc_encode_frames(const od_hipenc_params *p, int nframes,
 const unsigned char *frames, const od_hip_feed_level *views, int use_device,
 int device, unsigned char *pkt_out, long pkt_cap, od_hipenc_stats *stats) {
  od_hipenc_params q;
  od_hipenc *S;
  long n;
  int err;
  if (p == NULL || frames == NULL || nframes < 1) return OD_HIP_EFAULT;
  q = *p;
  if (q.nworkers > nframes) q.nworkers = nframes;
  /* one-shot: size the slots for this stream (bounded like the session default) */
  if (q.batch <= 0) {
    double per_frame;
    int cap;
    per_frame = 49.*((p->pic_width + 63) & ~63)*((p->pic_height + 63) & ~63);
    cap = (int)(6e9/per_frame);
    if (cap < 2*q.nworkers) cap = 2*q.nworkers;
    q.batch = nframes <= cap ? nframes : cap;
  }
  else if (q.batch < nframes) q.batch *= 2;      /* `batch` frames per half-buffer, as before */
  S = od_hipenc_open(&q, use_device && views == NULL, device, &err);
  if (S == NULL) return err;
  n = od_hipenc_encode(S, nframes, 0, frames, views, pkt_out, pkt_cap, stats);
  od_hipenc_close(S);
  return n;
}

/* Test hook: an adaptation context for the rate-coder parity test.  seed 0: the state
   od_adapt_ctx_reset leaves for a keyframe; otherwise that state with the PVQ codeword
   adaptation (pvq_adapt, pvq_k1_cdf) driven away from it the way coding does. */
const od_adapt_ctx *od_hipenc_test_adapt(unsigned seed) {
  static __thread od_state st;
  int i;
  int j;
  memset(&st.adapt, 0, sizeof(st.adapt));
  st.info.nplanes = 3;
  od_adapt_ctx_reset(&st.adapt, 1);
  if (seed != 0) {
    od_pvq_codeword_ctx *cw;
    cw = &st.adapt.pvq.pvq_codeword_ctx;
    for (i = 0; i < 2*OD_NBSIZES*OD_NSB_ADAPT_CTXS; i++) {
      seed = seed*1664525u + 1013904223u;
      cw->pvq_adapt[i] = 1 + (int)((seed >> 8)%(i & 1 ? 200000u : 60000u));
    }
    for (i = 0; i < 4; i++) {
      for (j = 0; j < 16; j++) {
        seed = seed*1664525u + 1013904223u;
        cw->pvq_k1_cdf[i][j] = (uint16_t)((j ? cw->pvq_k1_cdf[i][j - 1] : 0) + 1 + (seed >> 20)%900u);
      }
    }
  }
  return &st.adapt;
}

/* The padded input planes daala_encode_img_in() codes for one frame (the
   reference's od_img_copy_pad through a throw-away context): dense
   frame_width x frame_height luma and half-size chroma.  For tests and tools that
   need the device's input outside od_hipenc_encode_frames. */
int od_hipenc_pad_frame(const od_hipenc_params *p, const unsigned char *frame,
 unsigned char *const planes[3], int *frame_width, int *frame_height) {
  daala_enc_ctx *enc;
  od_img img;
  int pli;
  if (p == NULL || frame == NULL) return OD_HIP_EFAULT;
  enc = make_encoder(p, p->pic_width, p->pic_height);
  if (enc == NULL) return OD_HIP_EINVAL;
  if (frame_width != NULL) *frame_width = enc->state.frame_width;
  if (frame_height != NULL) *frame_height = enc->state.frame_height;
  if (planes != NULL) {
    fill_img(&img, frame, p->pic_width, p->pic_height);
    od_hipenc_copy_pad(enc, &img);
    for (pli = 0; pli < 3; pli++) {
      int w;
      int h;
      int y;
      w = enc->state.frame_width >> (pli > 0);
      h = enc->state.frame_height >> (pli > 0);
      for (y = 0; y < h; y++) {
        memcpy(planes[pli] + (size_t)y*w, enc->input_img[0].planes[pli].data
         + (size_t)y*enc->input_img[0].planes[pli].ystride, w);
      }
    }
  }
  daala_encode_free(enc);
  return 0;
}

/* The stream's three header packets (daala_encode_flush_header), 4-byte length
   prefixed like the video packets: what a decoder needs before od_hipdec_decode_frames. */
long od_hipenc_headers(const od_hipenc_params *p, unsigned char *out, long cap) {
  daala_enc_ctx *enc;
  daala_comment dc;
  daala_packet dp;
  long used;
  if (p == NULL || out == NULL) return OD_HIP_EFAULT;
  enc = make_encoder(p, p->pic_width, p->pic_height);
  if (enc == NULL) return OD_HIP_EINVAL;
  daala_comment_init(&dc);
  used = 0;
  while (daala_encode_flush_header(enc, &dc, &dp) > 0) {
    if (used + 4 + dp.bytes > cap) {
      used = OD_HIP_EINVAL;
      break;
    }
    out[used] = dp.bytes & 255;
    out[used + 1] = (dp.bytes >> 8) & 255;
    out[used + 2] = (dp.bytes >> 16) & 255;
    out[used + 3] = (dp.bytes >> 24) & 255;
    memcpy(out + used + 4, dp.packet, dp.bytes);
    used += 4 + dp.bytes;
  }
  daala_comment_clear(&dc);
  daala_encode_free(enc);
  return used;
}
