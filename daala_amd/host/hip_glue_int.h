/* hip_glue_int.h - state shared by the files of the reference-side glue
 * (hip_enc_glue.c, hip_pvq_host.c).  Not part of the C-ABI. */
#ifndef HIP_GLUE_INT_H
#define HIP_GLUE_INT_H

#include "encint.h"
#include "hip_enc_glue.h"

/* Per-thread state: which frame's feed this worker consumes. */
typedef struct glue_tls {
  const od_hip_feed_level *lev;   /* 4 views, or NULL: plain reference */
  const od_hip_feed_level *levc[2];  /* keyframes: the chroma planes' views (3 levels each), or NULL */
  const od_coeff *haar[3];        /* lossless frames: the device's Haar planes of this frame, or NULL */
  int haar_stride[3];
  int check;
  int time_cpu;
  int pli;                        /* plane of the block being coded */
  int sample_every;               /* re-search every n-th feed candidate on the host (0: off) */
  int sample_ctr;
  int host_pvq;                   /* 1: od_pvq_encode is hip_pvq_host.c's, 0: the reference's */
  daala_enc_ctx *enc;             /* encoder of the frame being coded by this thread */
  od_dct_func_2d fdct_cpu[OD_NBSIZES];   /* the context's own fdct_2d entries */
  od_hip_dering *dr;              /* this worker's device deringing object, or NULL */
  int16_t *dr_out[3];             /* deringed planes of the frame being coded */
  int dr_valid;                   /* dr_out holds the current frame */
  double *dist[3];                /* per 8x8 sub-block of every luma superblock: pow argument, energy vs the
                                     unfiltered and vs the deringed reconstruction (od_hip_dering_run_dist) */
  int dist_valid;                 /* dist[] holds the current frame */
  int dist_sb;                    /* superblock whose two od_compute_dist calls come next, or -1 */
  int dist_calls;
  int dr_error;
  /* the block being coded by od_pvq_encode: which feed records its bands read (hip_pvq_host.c) */
  const od_hip_feed_level *cur_L;   /* keyframe feed of the block's level, or NULL */
  const od_hip_pfeed_level *cur_P;  /* P-frame feed of the block's (plane, level), or NULL */
  int cur_blk;
  int cur_band;                   /* next band pvq_theta is called for */
  int cur_bs;
  int in_pure;                    /* inside the untouched reference od_pvq_encode (check mode) */
  /* P-frame feed of the inter frame being coded */
  od_hip_pfeed *pf;
  int pf_valid;
  double t_frame0;                /* start of the frame being coded (timers) */
  unsigned search_tick;           /* sampling counter of the search timers */
  od_hip_pfeed_level pfv[3][4];
  od_hipenc_stats st;
} glue_tls;

extern __thread glue_tls od_hipenc_tls;

/* kept reference definitions (the build recipe renames them, see Makefile) */
double od_ref_pvq_search_rdo_double_cpu(const double *xcoeff, int n, int k,
 od_coeff *ypulse, double g2);
int od_pvq_encode_cpu(daala_enc_ctx *enc, od_coeff *ref, od_coeff *in,
 od_coeff *out, int q0, int pli, int bs, const double *beta, int robust,
 int is_keyframe, int q_scaling, int bx, int by, const int16_t *qm,
 const int16_t *qm_inv);
void od_ref_pvq_encode_partition(od_ec_enc *ec, int qg, int theta, int max_theta,
 const od_coeff *in, int n, int k, generic_encoder model[3], od_adapt_ctx *adapt,
 int *exg, int *ext, int nodesync, int cdf_ctx, int is_keyframe, int code_skip,
 int skip_rest, int bs);
double od_ref_pvq_rate(int qg, int icgr, int theta, int ts, const od_adapt_ctx *adapt,
 const od_coeff *y0, int k, int n, int is_keyframe, int pli, int bs);
void od_encode_checkpoint_cpu(const daala_enc_ctx *enc, od_rollback_buffer *rbuf);
void od_encode_rollback_cpu(daala_enc_ctx *enc, const od_rollback_buffer *rbuf);

double od_hipenc_now(void);
double od_hipenc_fine_now(void);   /* time stamp counter, seconds; HIPENC_TIME=1 only */
void od_hipenc_fine_timer_init(void);
#define OD_HIPENC_TIME_SAMPLE 61   /* the per-call timers time one call in this many */
/* src/pvq_encoder.c:589 (no prototype in the reference's headers) */
int od_rdo_quant(od_coeff x, int q, double delta0);

/* hip_pvq_search.c: pvq_search_rdo_double, bit-identical, for the many searches pvq_theta
   makes of one vector (what does not depend on the candidate is computed once, the greedy
   pulses once per distinct K) */
#define OD_HIP_SEARCH_KCACHE (12)
typedef struct od_hip_search {
  double x[MAXN + 8] __attribute__((aligned(32)));   /* |x|, zero padded to a multiple of 4 */
  const double *xcoeff;
  double xx;
  double norm_1;
  double l1_inv;
  int n;
  int nv;
  int have_l1;
  int lanes;                   /* AVX2 lane scans (n >= 24) or the scalar scan */
  int nk;
  struct {
    int k;
    int placed;
    double xy;
    double yy;
  } ka[OD_HIP_SEARCH_KCACHE];
  int32_t ky[OD_HIP_SEARCH_KCACHE][MAXN + 8] __attribute__((aligned(32)));
} od_hip_search;
void od_hip_search_begin(od_hip_search *S, const double *xcoeff, int n);
void od_hip_search_begin_ex(od_hip_search *S, const double *xcoeff, int n, int lanes);
double od_hip_search_run(od_hip_search *S, int k, od_coeff *ypulse, double g2);
double od_hip_pvq_search_host(const double *xcoeff, int n, int k, od_coeff *ypulse,
 double g2);

/* hip_dct_host.c: the 2-D lifting DCTs on the host vector unit (od_dct_func_2d signatures) */
void od_hipenc_fdct4x4(od_coeff *y, int ystride, const od_coeff *x, int xstride);
void od_hipenc_fdct8x8(od_coeff *y, int ystride, const od_coeff *x, int xstride);
void od_hipenc_fdct16x16(od_coeff *y, int ystride, const od_coeff *x, int xstride);
void od_hipenc_fdct32x32(od_coeff *y, int ystride, const od_coeff *x, int xstride);
void od_hipenc_idct4x4(od_coeff *x, int xstride, const od_coeff *y, int ystride);
void od_hipenc_idct8x8(od_coeff *x, int xstride, const od_coeff *y, int ystride);
void od_hipenc_idct16x16(od_coeff *x, int xstride, const od_coeff *y, int ystride);
void od_hipenc_idct32x32(od_coeff *x, int xstride, const od_coeff *y, int ystride);

/* hip_mc_host.c: od_state_opt_vtbl leaves of the motion search, host vector unit */
void od_hipenc_mc_blend_full8(unsigned char *dst, int dystride, const unsigned char *src[4],
 int log_xblk_sz, int log_yblk_sz);
void od_hipenc_mc_blend_full_split8(unsigned char *dst, int dystride, const unsigned char *src[4],
 int oc, int s, int log_xblk_sz, int log_yblk_sz);
void od_hipenc_mc_predict1fmv8(od_state *state, unsigned char *dst, const unsigned char *src,
 int systride, int32_t mvx, int32_t mvy, int log_xblk_sz, int log_yblk_sz);
void od_hipenc_mc_cache_flush(void);
void od_hipenc_mc_cache_free(void);
void od_hipenc_mc_cache_stats(int64_t *hits, int64_t *misses);

/* hip_pvq_host.c */
int od_hip_pvq_encode_host(daala_enc_ctx *enc, od_coeff *ref, od_coeff *in,
 od_coeff *out, int q0, int pli, int bs, const double *beta, int robust,
 int is_keyframe, int q_scaling, int bx, int by, const int16_t *qm,
 const int16_t *qm_inv);
double od_hip_pvq_rate(int qg, int icgr, int theta, int ts, const od_adapt_ctx *adapt,
 const od_coeff *y0, int k, int n, int is_keyframe, int pli, int bs);

#endif
