/* hip_dec_glue.c - reference-side binding of the batched frame seam, decoder
 * (INTEGRATION.md seam 2; header hip_enc_glue.h).
 *
 * Compiled into the integration library libdaala_hipenc.so next to the reference decoder.  The
 * decoder's symbol parse (serial range decoder + PVQ synthesis into dtmp) stays the
 * reference's C code; its whole pixel-domain stage of a keyframe -
 *   idct_2d per block           src/decode.c:637  (vtable, src/state.h:106)
 *   od_postfilter_split         src/decode.c:864  (src/filter.c:1512)
 *   od_apply_postfilter_frame_sbs src/decode.c:1037 (src/filter.c:1588)
 *   od_dering per flagged SB    src/decode.c:1117 (src/filter.c:1835)
 *   od_smooth_recursive         src/decode.c:1146 (src/filter.c:2010)
 *   od_coeff_to_ref_plane       src/decode.c:1154 (src/state.c:1320)
 * - is replaced by ONE device pass per frame (od_hip_decode_tail) issued when the
 * decoder reaches od_coeff_to_ref_plane for plane 0: by then dtmp holds every
 * dequantised coefficient, bsize/bskip/dering_flags are final.  The five extern
 * functions above are bound here (the build renames the reference's definitions to
 * *_cpu with objcopy --redefine-sym, no source is touched); outside a device decode
 * (encoder threads, inter/lossless frames) they forward to the reference's code.
 * No reference text lives in this file. */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "decint.h"
#include "filter.h"
#include "partition.h"
#include "pvq.h"
#include "state.h"

#include "hip_enc_glue.h"

/* hip_dct_host.c */
void od_hipenc_fdct4x4(od_coeff *y, int ystride, const od_coeff *x, int xstride);
void od_hipenc_fdct8x8(od_coeff *y, int ystride, const od_coeff *x, int xstride);
void od_hipenc_fdct16x16(od_coeff *y, int ystride, const od_coeff *x, int xstride);
void od_hipenc_fdct32x32(od_coeff *y, int ystride, const od_coeff *x, int xstride);
void od_hipenc_idct4x4(od_coeff *x, int xstride, const od_coeff *y, int ystride);
void od_hipenc_idct8x8(od_coeff *x, int xstride, const od_coeff *y, int ystride);
void od_hipenc_idct16x16(od_coeff *x, int xstride, const od_coeff *y, int ystride);
void od_hipenc_idct32x32(od_coeff *x, int xstride, const od_coeff *y, int ystride);

/* the reference's definitions, renamed by the build (integration build recipe) */
void od_postfilter_split_cpu(od_coeff *c0, int stride, int bs, int f, int q,
 unsigned char *skip, int skip_stride, int hfilter, int vfilter);
void od_apply_postfilter_frame_sbs_cpu(od_coeff *c0, int stride, int nhsb, int nvsb,
 int xdec, int ydec, int q, unsigned char *skip, int skip_stride);
void od_dering_cpu(od_state *state, int16_t *y, int ystride, int16_t *x, int xstride,
 int ln, int sbx, int sby, int nhsb, int nvsb, int q, int xdec,
 int dir[OD_DERING_NBLOCKS][OD_DERING_NBLOCKS], int pli, unsigned char *bskip,
 int skip_stride);
void od_smooth_recursive_cpu(od_coeff *c, unsigned char *bsize, int bstride, int bx,
 int by, int bsi, int w, int xdec, int ydec, int min_bs, int quantizer, int pli);
void od_coeff_to_ref_plane_cpu(od_state *state, od_img *dst, int pli, od_coeff *src,
 int lossless_p);
void od_state_mc_predict_cpu(od_state *state, od_img *img_dst);
/* hip_enc_glue.c */
int od_hipenc_device_thread(void);
int od_hipenc_check_mode(void);
int od_hipenc_pframe_feed(od_state *state, od_img *pred);
void od_hipdec_thread_cleanup(void);
int od_hipenc_dering_hook(od_state *state, int16_t *y, int ystride, int16_t *x,
 int xstride, int ln, int sbx, int sby, int nhsb, int nvsb, int q, int xdec,
 int dir[OD_DERING_NBLOCKS][OD_DERING_NBLOCKS], int pli, unsigned char *bskip,
 int skip_stride);

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9*ts.tv_nsec;
}

typedef struct dec_tls {
  od_dec_ctx *dec;          /* decoder whose packet this thread is decoding */
  od_hip_ctx *ctx;          /* one-slot device context of this worker, or NULL */
  od_dct_func_2d idct_cpu[OD_NBSIZES];
  unsigned char *rec[3];    /* device output of the current frame */
  int pinned[6];            /* rec[0..2], stage[0..2] page-locked */
  od_coeff *stage[3];       /* page-locked staging copies of the coefficient planes */
  od_dct_func_2d fdct_cpu[OD_NBSIZES];
  od_coeff *md[3][4];       /* P frames: forward pyramid of the prediction (page-locked), per plane/level */
  int md_valid;             /* md[][] holds this frame's prediction */
  long md_hits;
  long md_check_fail;
  long idct_skipped;
  long haar_skipped;
  int device;               /* HIP device of this thread's objects */
  od_hip_mc *mc;            /* this thread's resident motion-compensation object */
  const od_state *mc_state; /* the codec state its reference copies belong to */
  unsigned char mc_dirty[OD_FRAME_MAX + 1];   /* reference image k changed since its upload */
  const od_state *mc_src_state;     /* the frame whose source planes the prediction object holds: */
  int64_t mc_src_time;              /* ... its state and state->cur_time */
  const void *poisoned;             /* decoder whose last frame failed: packets fail until its next keyframe */
  int mc_resident;                  /* image index the frame being reconstructed was written to on the device
                                       (od_hip_mc_set_ref_ctx), or -1 */
  int failed;               /* a device stage of the current packet / frame failed: surfaced as an error code */
  int haar_frame;           /* the frame being decoded uses the Haar wavelet with a quantizer > 0: host path */
  int check;
  double t_device;
  /* P frames: PVQ synthesis on the device (below) */
  int pred_on_device;       /* this P frame's prediction went straight into the context's picture planes */
  od_hip_dsynth *ds;
  int ds_off;               /* HIPDEC_SYNTH=0 */
  int ds_on;                /* this frame's blocks are recorded for the device */
  int ds_overflow;
  od_hip_dsynth_block *ds_blocks;
  od_hip_dsynth_band *ds_bands;
  int16_t *ds_pulses;
  long ds_max_blocks, ds_max_bands, ds_max_pulses;
  long ds_nblocks, ds_nbands, ds_npulses;
  const double *ds_gr[3][4];
  long ds_cur;              /* record of the block being parsed, -1: none */
  unsigned ds_cur_mask;     /* its bands that went through pvq_synthesis */
  int16_t *ds_qm_seen;      /* the tables the device holds */
} dec_tls;

static __thread dec_tls D;

static __thread long mc_dev_frames;
static __thread long tail_dev_frames;     /* frames whose pixel-domain stage ran on the device */
static long g_tail_dev_frames;
static long g_md_hits;            /* prediction-side transforms served from the device pyramid */
static long g_md_check_fail;
static __thread long mc_check_fail;
static __thread long ds_frames;          /* P frames whose PVQ synthesis ran on the device */
static __thread long ds_check_fail;
static __thread long ds_wide_bands;      /* bands whose pulses needed more than 16 bits */
static long g_ds_frames;
static long g_ref_resident_frames;   /* frames of the last decode that became a reference on the device */
static long g_ds_check_fail;
static long g_ds_wide_bands;
static long g_mc_dev_frames;      /* totals of the last od_hipdec_decode_frames call */
static long g_mc_check_fail;

void od_hipdec_md_stats(long out[2]) {
  out[0] = g_md_hits;
  out[1] = g_md_check_fail;
}

long od_hipdec_tail_frames(void) {
  return g_tail_dev_frames;
}

/* After od_hipdec_decode_frames: out[0] = P frames whose PVQ synthesis ran on the device,
   out[1] = check-mode mismatches (reference gains, coefficient planes), out[2] = bands whose
   pulses went up as two 16-bit entries each. */
void od_hipdec_synth_stats(long out[3]) {
  out[0] = g_ds_frames;
  out[1] = g_ds_check_fail;
  out[2] = g_ds_wide_bands;
}

/* Frames of the last decode whose reconstruction became a reference image on the device
   (od_hip_mc_set_ref_ctx): no reference upload for the frame that follows. */
long od_hipdec_ref_resident_frames(void) {
  return g_ref_resident_frames;
}

void od_hipdec_mc_stats(long out[2]) {
  out[0] = g_mc_dev_frames;
  out[1] = g_mc_check_fail;
}

/* Device decode applies to DCT frames, I and P: the pixel-domain stage after the block
   loop (src/decode.c:1032-1155) is the same for both except that only keyframes are
   smoothed (:1140); every block of a P frame - skipped ones too - leaves its coefficients
   in dtmp and goes through idct_2d (:637).  B frames are left to the reference.  Frame type
   and quantizers are known before the first block is parsed (src/decode.c:1195, :989-993). */
static int on_device(void) {
  return D.ctx != NULL && D.dec != NULL && !D.haar_frame && !D.failed
   && (D.dec->state.frame_type == OD_I_FRAME || D.dec->state.frame_type == OD_P_FRAME)
   && D.dec->state.quantizer[0] > 0;
}

void od_ref_plane_to_coeff_cpu(od_state *state, od_coeff *dst, int lossless_p, od_img *src,
 int pli);

/* The Haar-wavelet flag of a frame is a decoded bit (src/decode.c:1206) that the quantizer
   says nothing about: a stream may carry Haar frames with a quantizer > 0 (the reference's
   encoder writes them when built with OD_USE_HAAR_WAVELET).  Such a frame never reaches the
   DCT hooks; the decoder's first od_haar / od_haar_inv call reveals it before anything of
   the frame's pixel stage has been skipped, and the frame then takes the reference's host
   path entirely.  On a P frame the prediction's coefficient planes were going to come from
   the device pyramid (od_ref_plane_to_coeff skipped): they are rebuilt here, without the
   lapping filter a Haar frame does not apply (src/decode.c:1004). */
static void haar_frame_seen(void) {
  od_state *st;
  int pli;
  if (D.dec == NULL || D.ctx == NULL || D.haar_frame || D.dec->state.quantizer[0] == 0) return;
  D.haar_frame = 1;
  D.ds_on = 0;
  st = &D.dec->state;
  if (D.md_valid && !D.check) {
    od_img *rec;
    rec = st->ref_imgs + st->ref_imgi[OD_FRAME_SELF];
    /* the prediction itself may never have come to the host: the reference's own function
       computes the same picture */
    if (D.pred_on_device) od_state_mc_predict_cpu(st, rec);
    D.pred_on_device = 0;
    for (pli = 0; pli < st->info.nplanes; pli++) {
      od_ref_plane_to_coeff_cpu(st, st->mctmp[pli], st->quantizer[pli] == 0, rec, pli);
    }
  }
  D.md_valid = 0;
}

/* called by the od_haar binding (hip_enc_glue.c) */
void od_hipdec_haar_notify(void) {
  haar_frame_seen();
}

/* Lossless keyframes (quantizer 0: Haar wavelet of every whole superblock instead of
   lapping + DCT, src/decode.c:785, :1036): od_haar_inv per superblock (:621) and the shift-0
   od_coeff_to_ref_plane are one od_hip_inverse_haar pass per frame. */
static int on_device_lossless(void) {
  return D.ctx != NULL && D.dec != NULL && !D.failed && D.dec->state.frame_type == OD_I_FRAME
   && D.dec->state.quantizer[0] == 0 && D.dec->state.quantizer[1] == 0
   && D.dec->state.quantizer[2] == 0 && D.dec->state.info.nplanes == 3;
}

void od_haar_inv_cpu(od_coeff *x, int xstride, const od_coeff *y, int ystride, int ln);
void od_haar_inv(od_coeff *x, int xstride, const od_coeff *y, int ystride, int ln) {
  if (on_device_lossless()) {
    D.haar_skipped++;
    return;
  }
  haar_frame_seen();
  od_haar_inv_cpu(x, xstride, y, ystride, ln);
}

/* P frames, prediction side.  The decoder turns the motion-compensated picture into the
   PVQ reference of every block: od_ref_plane_to_coeff + od_apply_prefilter_frame_sbs over the
   plane (src/decode.c:997-1008), od_prefilter_split down the recursion (:848) and one
   forward transform per coded block (:565) - the same lapped multi-size transform the encoder
   applies to its input, i.e. one block of the device's forward PYRAMID at the block's level.
   md_pyramid() runs it right after the device prediction; the four entry points then have
   nothing left to do for the mctmp / mdtmp planes.  (The block sizes are parsed along with
   the coefficients, so every level is needed, as in the encoder's RDO.)  Check mode keeps the
   host path and compares block by block. */
static int md_plane_of(const od_state *st, const od_coeff *p, od_coeff *const planes[]) {
  int pli;
  for (pli = 0; pli < st->info.nplanes && pli < 3; pli++) {
    size_t n;
    n = (size_t)(st->frame_width >> (pli > 0))*(st->frame_height >> (pli > 0));
    if (planes[pli] != NULL && p >= planes[pli] && p < planes[pli] + n) return pli;
  }
  return -1;
}

static int md_served(void) {
  return D.md_valid && !D.check && D.dec != NULL && D.dec->state.quantizer[0] > 0;
}

void od_ref_plane_to_coeff(od_state *state, od_coeff *dst, int lossless_p, od_img *src,
 int pli) {
  if (md_served() && state == &D.dec->state && pli >= 0 && pli < 3 && dst == state->mctmp[pli]) return;
  od_ref_plane_to_coeff_cpu(state, dst, lossless_p, src, pli);
}

void od_apply_prefilter_frame_sbs_cpu(od_coeff *c0, int stride, int nhsb, int nvsb, int xdec,
 int ydec);
void od_apply_prefilter_frame_sbs(od_coeff *c0, int stride, int nhsb, int nvsb, int xdec,
 int ydec) {
  if (md_served() && md_plane_of(&D.dec->state, c0, D.dec->state.mctmp) >= 0) return;
  od_apply_prefilter_frame_sbs_cpu(c0, stride, nhsb, nvsb, xdec, ydec);
}

void od_prefilter_split_cpu(od_coeff *c0, int stride, int bs, int f, int hfilter, int vfilter);
void od_prefilter_split(od_coeff *c0, int stride, int bs, int f, int hfilter, int vfilter) {
  if (md_served() && md_plane_of(&D.dec->state, c0, D.dec->state.mctmp) >= 0) return;
  od_prefilter_split_cpu(c0, stride, bs, f, hfilter, vfilter);
}

/* ------------------------------------------------------------------------ */
/* P frames: PVQ synthesis on the device (include/daala_hip.h section 4e).
   The reference of every band of a P frame is the prediction's transform, which the device
   holds (md_pyramid): the decoder thread only PARSES.  The reference's od_block_decode /
   od_pvq_decode (src/decode.c:536-640, src/pvq_decoder.c:312-388) stay the code that runs;
   four of their leaf calls are bound by the build (Makefile):
     fdct_2d of the prediction      md_fdct(): names the block (plane, size, origin)
     od_pvq_compute_gain(ref, ..)   od_hipdec_pvq_compute_gain(): gr from the device's gain pass
     pvq_synthesis(..)              records (gain, theta -> host sin / cos, pulses)
     od_coding_order_to_raster(..)  od_hipdec_coding_order_to_raster(): closes the block - DC,
                                    which bands were cleared (OD_PVQ_SKIP_ZERO)
   Nothing else of the parse reads the reference VALUES, so the host never sees them: three
   more bindings take the per-coefficient work the device now does out of the block decoder -
     od_decode_compute_pred         (static; decode_tail.c) no copy of the prediction's transform
     od_init_skipped_coeffs         no host coefficient plane is written
     od_raster_to_coding_order      the reference vector handed to od_pvq_decode is a SENTINEL
   - a band that holds the sentinel after od_pvq_decode was copied from the reference
   (OD_PVQ_SKIP_COPY, skipped blocks) and needs no record: the device starts every block from
   the prediction's transform.  The 33 MB of level planes down and the 12.5 MB of coefficients
   up per 1080p frame become 4 MB of gains down and the records up, and the decoder thread no
   longer touches a coefficient plane.  Check mode keeps the reference's host path with real
   planes beside the records and compares the gains and the finished coefficient planes. */
#define DS_SENTINEL (1 << 28)

static int ds_lvl(int pli, int bs) {
  return (pli > 0 ? 2 : 3) - bs;
}

static const int DS_OFF[] = {1, 16, 24, 32, 64, 96, 128, 256, 384, 512};
static const int DS_NB[4] = {1, 4, 7, 9};

static void ds_begin_block(int pli, int bs, long org) {
  od_hip_dsynth_block *b;
  D.ds_cur = -1;
  if (D.ds_nblocks >= D.ds_max_blocks) {
    D.ds_overflow = 1;
    return;
  }
  b = D.ds_blocks + D.ds_nblocks;
  b->org = (int32_t)org;
  b->dc = 0;
  b->pli = (uint8_t)pli;
  b->bs = (uint8_t)bs;
  b->pad[0] = b->pad[1] = 0;
  D.ds_cur = D.ds_nblocks++;
  D.ds_cur_mask = 0;
}

/* which band of the current block a qm pointer of od_pvq_decode belongs to */
static int ds_band_of(const int16_t *qm) {
  const od_state *st;
  const od_hip_dsynth_block *b;
  long o;
  int i;
  st = &D.dec->state;
  b = D.ds_blocks + D.ds_cur;
  o = qm - (st->qm + od_qm_offset(b->bs, b->pli > 0));
  for (i = 0; i < DS_NB[b->bs]; i++) if (DS_OFF[i] == o) return i;
  return -1;
}

double od_hipdec_pvq_compute_gain(od_coeff *x, int n, int q0, double *g, double beta,
 const int16_t *qm) {
  if (D.ds_on && D.ds_cur >= 0 && D.dec != NULL) {
    const od_hip_dsynth_block *b;
    int band;
    b = D.ds_blocks + D.ds_cur;
    band = ds_band_of(qm);
    if (band >= 0 && DS_OFF[band + 1] - DS_OFF[band] == n) {
      int w;
      int bn;
      long blk;
      long nblk;
      double gr;
      double cg2;
      w = D.dec->state.frame_width >> (b->pli > 0);
      bn = 4 << b->bs;
      blk = (b->org/w/bn)*(w/bn) + (b->org%w)/bn;
      nblk = (long)(w/bn)*((D.dec->state.frame_height >> (b->pli > 0))/bn);
      gr = D.ds_gr[b->pli][ds_lvl(b->pli, b->bs)][band*nblk + blk];
      cg2 = 0;
      if (D.check) {
        double g2;
        cg2 = od_pvq_compute_gain(x, n, q0, &g2, beta, qm);
        if (g2 != gr) ds_check_fail++;
      }
      *g = gr;
      {
        /* od_gain_compand (static, src/pvq.c:422-425) as the C-ABI restates it on the host:
           this process's pow; check mode compares with the reference's own return value */
        double cgr;
        if (od_hip_pvq_compand(1, &gr, q0, beta, &cgr) != 0) D.ds_overflow = 1;
        if (D.check && cgr != cg2) ds_check_fail++;
        return cgr;
      }
    }
    D.ds_overflow = 1;
  }
  return od_pvq_compute_gain(x, n, q0, g, beta, qm);
}

void pvq_synthesis_cpu(od_coeff *xcoeff, od_coeff *ypulse, od_coeff *ref, int n, double gr,
 int noref, double g, double theta, const int16_t *qm, const int16_t *qm_inv);
void pvq_synthesis(od_coeff *xcoeff, od_coeff *ypulse, od_coeff *ref, int n, double gr,
 int noref, double g, double theta, const int16_t *qm, const int16_t *qm_inv) {
  if (D.ds_on && D.ds_cur >= 0 && D.dec != NULL) {
    int band;
    int nn;
    band = ds_band_of(qm);
    nn = n - !noref;
    int wide;
    int i;
    wide = 0;
    for (i = 0; i < nn; i++) wide |= ypulse[i] > 32767 || ypulse[i] < -32768;
    if (band >= 0 && DS_OFF[band + 1] - DS_OFF[band] == n && D.ds_nbands < D.ds_max_bands
     && D.ds_npulses + (nn << wide) <= D.ds_max_pulses) {
      od_hip_dsynth_band *r;
      int16_t *y;
      r = D.ds_bands + D.ds_nbands++;
      r->block = (uint32_t)D.ds_cur;
      r->band = (uint8_t)band;
      r->mode = (noref ? OD_HIP_DSYNTH_NOREF : OD_HIP_DSYNTH_REF) | (wide ? OD_HIP_DSYNTH_WIDE : 0);
      r->pad[0] = r->pad[1] = 0;
      r->yoff = (uint32_t)D.ds_npulses;
      r->pad2 = 0;
      r->g = g;
      /* od_pvq_synthesis_partial's sin(theta), cos(theta) (src/pvq.c:574-577): this process's libm */
      r->sin_theta = noref ? 0 : sin(theta);
      r->cos_theta = noref ? 0 : cos(theta);
      y = D.ds_pulses + D.ds_npulses;
      if (!wide) for (i = 0; i < nn; i++) y[i] = (int16_t)ypulse[i];
      else {
        /* quantizer 1-2 on noise-like content: K beyond 16 bits - two entries per pulse */
        for (i = 0; i < nn; i++) {
          y[2*i] = (int16_t)(uint16_t)((uint32_t)ypulse[i] & 0xffff);
          y[2*i + 1] = (int16_t)(uint16_t)((uint32_t)ypulse[i] >> 16);
        }
      }
      D.ds_npulses += nn << wide;
      ds_wide_bands += wide;
      D.ds_cur_mask |= 1u << band;
    }
    else D.ds_overflow = 1;
    if (!D.check) {
      /* the band is synthesised on the device; the host buffer is never read for it, but it is
         left defined */
      memset(xcoeff, 0, sizeof(*xcoeff)*n);
      return;
    }
  }
  pvq_synthesis_cpu(xcoeff, ypulse, ref, n, gr, noref, g, theta, qm, qm_inv);
}

/* the block being parsed is synthesised on the device and no host copy of its data is kept */
static int ds_device_only(void) {
  return D.ds_on && D.ds_cur >= 0 && !D.check;
}

/* decode_tail.c: od_decode_compute_pred asks before it copies the prediction's transform */
int od_hipdec_pred_from_device(void) {
  return ds_device_only();
}

void od_hipdec_init_skipped_coeffs(od_coeff *d, od_coeff *pred, int is_keyframe, int bo, int n,
 int w) {
  if (ds_device_only()) return;
  od_init_skipped_coeffs(d, pred, is_keyframe, bo, n, w);
}

void od_hipdec_raster_to_coding_order(od_coeff *dst, int n, const od_coeff *src, int stride) {
  if (ds_device_only()) {
    int i;
    int len;
    len = n*n < 512 ? n*n : 512;        /* the coded positions (OD_BAND_OFFSETS, src/partition.c:77-83) */
    dst[0] = 0;                         /* index 0 is the DC: no band, and the block decoder adds the DC to it */
    for (i = 1; i < len; i++) dst[i] = DS_SENTINEL;
    for (; i < n*n; i++) dst[i] = 0;    /* never coded (32x32): defined, never read by the device path */
    return;
  }
  od_raster_to_coding_order(dst, n, src, stride);
}

void od_hipdec_coding_order_to_raster(od_coeff *dst, int stride, const od_coeff *src, int n) {
  if (D.ds_on && D.ds_cur >= 0 && D.dec != NULL) {
    od_hip_dsynth_block *b;
    const od_state *st;
    int i;
    st = &D.dec->state;
    b = D.ds_blocks + D.ds_cur;
    if (dst == st->dtmp[b->pli] + b->org && n == 4 << b->bs) {
      b->dc = src[0] - (D.check ? st->mdtmp[b->pli][b->org] : 0);
      for (i = 0; i < DS_NB[b->bs]; i++) {
        int j;
        int len;
        if (D.ds_cur_mask >> i & 1) continue;
        len = DS_OFF[i + 1] - DS_OFF[i];
        for (j = 0; j < len && src[DS_OFF[i] + j] == 0; j++);
        if (j < len) continue;          /* still the reference (sentinel): copied, no record */
        if (D.ds_nbands >= D.ds_max_bands) {
          D.ds_overflow = 1;
          break;
        }
        memset(D.ds_bands + D.ds_nbands, 0, sizeof(*D.ds_bands));
        D.ds_bands[D.ds_nbands].block = (uint32_t)D.ds_cur;
        D.ds_bands[D.ds_nbands].band = (uint8_t)i;
        D.ds_bands[D.ds_nbands].mode = OD_HIP_DSYNTH_ZERO;
        D.ds_nbands++;
      }
    }
    else D.ds_overflow = 1;
    D.ds_cur = -1;
    if (!D.check) return;               /* the host's dtmp plane is not what the device reads */
  }
  od_coding_order_to_raster(dst, stride, src, n);
}

/* per P frame, after the prediction's pyramid: tables (when they changed), gains, a clean
   record list */
static int ds_frame_begin(od_state *state) {
  int pli;
  int bs;
  size_t qn;
  D.ds_on = 0;
  if (D.ds == NULL) return 0;
  qn = OD_QM_BUFFER_SIZE*sizeof(state->qm[0]);
  if (D.ds_qm_seen == NULL) {
    D.ds_qm_seen = (int16_t *)malloc(2*qn);
    if (D.ds_qm_seen == NULL) return -1;
    memset(D.ds_qm_seen, 0xff, 2*qn);
  }
  if (memcmp(D.ds_qm_seen, state->qm, qn) != 0 || memcmp((char *)D.ds_qm_seen + qn, state->qm_inv, qn) != 0) {
    for (pli = 0; pli < 3; pli++) {
      for (bs = 0; bs < (pli > 0 ? 3 : 4); bs++) {
        int off;
        off = od_qm_offset(bs, pli > 0);
        if (od_hip_dsynth_set_level(D.ds, pli, ds_lvl(pli, bs), state->qm + off, state->qm_inv + off) != 0) return -2;
      }
    }
    memcpy(D.ds_qm_seen, state->qm, qn);
    memcpy((char *)D.ds_qm_seen + qn, state->qm_inv, qn);
  }
  if (od_hip_dsynth_ref_gains(D.ds, D.ds_gr) != 0) return -3;
  D.ds_nblocks = D.ds_nbands = D.ds_npulses = 0;
  D.ds_cur = -1;
  D.ds_overflow = 0;
  D.ds_on = 1;
  return 0;
}

static void md_fdct(int bs, od_coeff *y, int ystride, const od_coeff *x, int xstride) {
  if (D.md_valid && D.dec != NULL && D.dec->state.quantizer[0] > 0) {
    const od_state *st;
    int pli;
    st = &D.dec->state;
    pli = md_plane_of(st, y, st->mdtmp);
    if (pli >= 0) {
      int w;
      int n;
      int lvl;
      size_t off;
      w = st->frame_width >> (pli > 0);
      n = 4 << bs;
      lvl = (pli > 0 ? 2 : 3) - bs;
      off = (size_t)(y - st->mdtmp[pli]);
      if (ystride == w && lvl >= 0 && (D.md[pli][lvl] != NULL || (D.ds_on && !D.check)) && (off/w & (n - 1)) == 0
       && (off%w & (n - 1)) == 0) {
        const od_coeff *src;
        int i;
        if (D.ds_on) ds_begin_block(pli, bs, (long)off);
        if (D.ds_on && !D.check) {
          /* the prediction's transform stays on the device */
          D.md_hits++;
          return;
        }
        src = D.md[pli][lvl] + off;
        if (D.check) {
          od_coeff tmp[32*32];
          (*D.fdct_cpu[bs])(tmp, n, x, xstride);
          for (i = 0; i < n; i++) {
            if (memcmp(tmp + i*n, src + (size_t)i*w, sizeof(od_coeff)*n) != 0) {
              D.md_check_fail++;
              break;
            }
          }
        }
        for (i = 0; i < n; i++) memcpy(y + (size_t)i*ystride, src + (size_t)i*w, sizeof(od_coeff)*n);
        D.md_hits++;
        return;
      }
    }
  }
  if (D.ds_on) {
    /* a block of a frame recorded for the device took the host transform: the records no
       longer cover the frame */
    D.ds_overflow = 1;
  }
  (*D.fdct_cpu[bs])(y, ystride, x, xstride);
}

#define MD_FDCT_HOOK(name, bs) \
  static void name(od_coeff *y, int ystride, const od_coeff *x, int xstride) { \
    md_fdct(bs, y, ystride, x, xstride); \
  }
MD_FDCT_HOOK(hook_md_fdct4, 0)
MD_FDCT_HOOK(hook_md_fdct8, 1)
MD_FDCT_HOOK(hook_md_fdct16, 2)
MD_FDCT_HOOK(hook_md_fdct32, 3)

/* prediction planes -> device -> forward pyramid -> page-locked level planes */
static int md_pyramid(od_state *state, od_img *pred) {
  const unsigned char *planes[3];
  int strides[3];
  int pli;
  int lvl;
  double t0;
  t0 = now_s();
  if (D.ds == NULL && !D.ds_off) {
    /* HIPDEC_SYNTH=0: the PVQ synthesis of P frames stays on the host (A/B, tests) */
    const char *e;
    e = getenv("HIPDEC_SYNTH");
    D.ds_off = e != NULL && atoi(e) == 0;
    if (!D.ds_off) {
      D.ds = od_hip_dsynth_create(D.ctx);
      if (D.ds == NULL || od_hip_dsynth_buffers(D.ds, &D.ds_blocks, &D.ds_max_blocks, &D.ds_bands,
       &D.ds_max_bands, &D.ds_pulses, &D.ds_max_pulses) != 0) {
        return -6;
      }
    }
  }
  for (pli = 0; pli < 3; pli++) {
    int nl;
    size_t np;
    nl = pli > 0 ? 3 : 4;
    np = (size_t)(state->frame_width >> (pli > 0))*(state->frame_height >> (pli > 0))*sizeof(od_coeff);
    np = (np + 4095) & ~(size_t)4095;
    for (lvl = 0; lvl < nl; lvl++) {
      if (D.md[pli][lvl] == NULL && (D.ds == NULL || D.check)) {
        void *mem;
        mem = NULL;
        if (posix_memalign(&mem, 4096, np) != 0) return -1;
        (void)od_hip_host_register(mem, np);        /* failure to lock only costs speed */
        D.md[pli][lvl] = (od_coeff *)mem;
      }
    }
    planes[pli] = pred->planes[pli].data;
    strides[pli] = pred->planes[pli].ystride;
  }
  if (!D.pred_on_device && od_hip_upload_planes(D.ctx, 0, planes, strides) != 0) return -2;
  if (od_hip_forward_pyramid(D.ctx, 0, 1) != 0) return -3;
  if (D.ds != NULL && ds_frame_begin(state) != 0) return -5;
  if (!D.ds_on || D.check) {
    for (pli = 0; pli < 3; pli++) {
      for (lvl = 0; lvl < (pli > 0 ? 3 : 4); lvl++) {
        if (od_hip_download_level(D.ctx, 0, pli, lvl, D.md[pli][lvl]) != 0) return -4;
      }
    }
  }
  D.t_device += now_s() - t0;
  return 0;
}

#define IDCT_HOOK(name, bs) \
  static void name(od_coeff *x, int xstride, const od_coeff *y, int ystride) { \
    if (on_device()) { \
      D.idct_skipped++; \
      return; \
    } \
    (*D.idct_cpu[bs])(x, xstride, y, ystride); \
  }
IDCT_HOOK(hook_idct4, 0)
IDCT_HOOK(hook_idct8, 1)
IDCT_HOOK(hook_idct16, 2)
IDCT_HOOK(hook_idct32, 3)

void od_postfilter_split(od_coeff *c0, int stride, int bs, int f, int q,
 unsigned char *skip, int skip_stride, int hfilter, int vfilter) {
  if (on_device()) return;
  od_postfilter_split_cpu(c0, stride, bs, f, q, skip, skip_stride, hfilter, vfilter);
}

void od_apply_postfilter_frame_sbs(od_coeff *c0, int stride, int nhsb, int nvsb,
 int xdec, int ydec, int q, unsigned char *skip, int skip_stride) {
  if (on_device()) return;
  od_apply_postfilter_frame_sbs_cpu(c0, stride, nhsb, nvsb, xdec, ydec, q, skip,
   skip_stride);
}

void od_dering(od_state *state, int16_t *y, int ystride, int16_t *x, int xstride,
 int ln, int sbx, int sby, int nhsb, int nvsb, int q, int xdec,
 int dir[OD_DERING_NBLOCKS][OD_DERING_NBLOCKS], int pli, unsigned char *bskip,
 int skip_stride) {
  if (on_device()) return;
  /* encoder threads: answered from the worker's per-frame device pass when it has one */
  if (od_hipenc_dering_hook(state, y, ystride, x, xstride, ln, sbx, sby, nhsb, nvsb, q, xdec,
   dir, pli, bskip, skip_stride)) {
    return;
  }
  od_dering_cpu(state, y, ystride, x, xstride, ln, sbx, sby, nhsb, nvsb, q, xdec, dir,
   pli, bskip, skip_stride);
}

void od_smooth_recursive(od_coeff *c, unsigned char *bsize, int bstride, int bx,
 int by, int bsi, int w, int xdec, int ydec, int min_bs, int quantizer, int pli) {
  if (on_device()) return;
  od_smooth_recursive_cpu(c, bsize, bstride, bx, by, bsi, w, xdec, ydec, min_bs,
   quantizer, pli);
}

static int injected_failure(void);

/* Which device stage failed and with what code (the frame itself is reported through
   daala_decode_packet_in's OD_EFAULT): one line on stderr when HIPDEC_DEBUG is set. */
static __thread double tm_mc, tm_pyr, tm_tail, tm_mark, tm_parse;   /* HIPDEC_DEBUG=2: where a frame's time goes */
static void stage_failed(const char *stage, int rc) {
  if (getenv("HIPDEC_DEBUG") != NULL) {
    fprintf(stderr, "daala_hipdec: %s failed (step %d): %s\n", stage, rc, od_hip_last_error());
  }
}

/* HIPDEC_REF_RESIDENT=0: references always uploaded from the host image (rounds 2-3) */
static int ref_resident_on = 1;
static __thread long ref_resident_frames;   /* frames that became a reference on the device */

static int device_frame(od_state *state) {
  const unsigned char *bskip[3];
  int32_t thr[3];
  int32_t quant[3];
  int pli;
  int nplanes;
  int synth;
  double t0;
  if (injected_failure()) return -9;
  t0 = now_s();
  nplanes = state->info.nplanes;
  if (od_hip_set_bsize(D.ctx, 0, state->bsize, state->bstride) != 0) return -1;
  synth = D.ds_on && state->frame_type == OD_P_FRAME;
  if (synth) {
    if (D.ds_overflow || D.ds_cur >= 0) return -6;
    if (od_hip_dsynth_run(D.ds, D.ds_nblocks, D.ds_nbands, D.ds_npulses) != 0) return -7;
    ds_frames++;
    if (D.check) {
      /* the reference's host path filled dtmp from real planes: the device's must be equal */
      for (pli = 0; pli < nplanes; pli++) {
        size_t np;
        od_coeff *got;
        np = (size_t)(state->frame_width >> (pli > 0))*(state->frame_height >> (pli > 0));
        got = (od_coeff *)malloc(np*sizeof(od_coeff));
        if (got == NULL || od_hip_download_coeffs(D.ctx, 0, pli, got) != 0
         || memcmp(got, state->dtmp[pli], np*sizeof(od_coeff)) != 0) {
          ds_check_fail++;
        }
        free(got);
      }
    }
  }
  D.ds_on = 0;
  for (pli = 0; pli < nplanes; pli++) {
    const od_coeff *src;
    bskip[pli] = state->bskip[pli];
    quant[pli] = state->quantizer[pli];
    /* od_dering's threshold (src/filter.c:1876), host libm as in the reference */
    thr[pli] = (int32_t)(1.0*pow(state->quantizer[pli], 0.84182));
    if (synth) continue;
    src = state->dtmp[pli];
    if (D.stage[pli] != NULL) {
      memcpy(D.stage[pli], src, sizeof(od_coeff)*(size_t)(state->frame_width >> (pli > 0))
       *(state->frame_height >> (pli > 0)));
      src = D.stage[pli];
    }
    if (od_hip_upload_coeffs(D.ctx, 0, pli, src) != 0) return -2;
  }
  if (od_hip_set_decode_info(D.ctx, 0, state->dering_flags, bskip,
   state->skip_stride) != 0) return -3;
  if (od_hip_decode_tail(D.ctx, 0, 1, thr, quant, state->frame_type == OD_I_FRAME) != 0) return -4;
  /* The frame just reconstructed is the next frame's reference (src/decode.c:1267-1270:
     od_img_edge_ext of ref_imgs[SELF]): when this thread's prediction object exists - the
     stream has P frames - the reference is taken from the context's reconstruction planes on
     the device, padding included; the host copy below is only the picture the caller gets. */
  D.mc_resident = -1;
  if (D.mc != NULL && D.mc_state == state && ref_resident_on && nplanes <= 3 && !state->full_precision_references) {
    int k;
    int ok;
    k = state->ref_imgi[OD_FRAME_SELF];
    ok = k >= 0 && k <= OD_FRAME_MAX;
    for (pli = 0; ok && pli < nplanes; pli++) {
      const od_img_plane *rp;
      rp = state->ref_imgs[k].planes + pli;
      ok = rp->xstride == 1 && od_hip_mc_set_ref_ctx(D.mc, pli, k, D.ctx, 0, rp->ystride,
       (state->frame_height + 2*OD_BUFFER_PADDING) >> rp->ydec, OD_BUFFER_PADDING >> rp->xdec,
       OD_BUFFER_PADDING >> rp->ydec) == 0;
    }
    if (ok) {
      D.mc_resident = k;
      ref_resident_frames++;
    }
    else if (k >= 0 && k <= OD_FRAME_MAX) D.mc_dirty[k] = 1;      /* falls back to the upload of the host image */
  }
  for (pli = 0; pli < nplanes; pli++) {
    if (od_hip_download_recon(D.ctx, 0, pli, D.rec[pli]) != 0) return -5;
  }
  D.t_device += now_s() - t0;
  tail_dev_frames++;
  return 0;
}

static int device_frame_lossless(od_state *state) {
  int pli;
  double t0;
  t0 = now_s();
  for (pli = 0; pli < 3; pli++) {
    const od_coeff *src;
    src = state->dtmp[pli];
    if (D.stage[pli] != NULL) {
      memcpy(D.stage[pli], src, sizeof(od_coeff)*(size_t)(state->frame_width >> (pli > 0))
       *(state->frame_height >> (pli > 0)));
      src = D.stage[pli];
    }
    if (od_hip_upload_coeffs(D.ctx, 0, pli, src) != 0) return -2;
  }
  if (od_hip_inverse_haar(D.ctx, 0, 1) != 0) return -4;
  for (pli = 0; pli < 3; pli++) {
    if (od_hip_download_recon(D.ctx, 0, pli, D.rec[pli]) != 0) return -5;
  }
  D.t_device += now_s() - t0;
  return 0;
}

void od_coeff_to_ref_plane(od_state *state, od_img *dst, int pli, od_coeff *src,
 int lossless_p) {
  od_img_plane *ip;
  int w;
  int h;
  int y;
  /* every frame's reconstruction lands in state->ref_imgs[SELF] through this function: the
     device copy of that reference image (mc_predict_device) is stale from here on */
  if (state == D.mc_state && dst >= state->ref_imgs && dst <= state->ref_imgs + OD_FRAME_MAX) {
    D.mc_dirty[dst - state->ref_imgs] = 1;
  }
  if (D.failed && D.dec != NULL && state == &D.dec->state) {
    od_coeff_to_ref_plane_cpu(state, dst, pli, src, lossless_p);
    return;
  }
  if (lossless_p && on_device_lossless()) {
    if (pli == 0) {
      if (D.haar_skipped == 0 || device_frame_lossless(state) != 0) {
        /* The inverse transforms of this frame were skipped for the device pass that has now
           failed: there is no picture.  The frame is reported as failed (daala_decode_packet_in
           returns OD_EFAULT, see the binding below); the plane is written from what is there so
           that no caller reads uninitialised memory. */
        D.failed = 1;
      }
      D.haar_skipped = 0;
    }
    if (D.failed) {
      od_coeff_to_ref_plane_cpu(state, dst, pli, src, lossless_p);
      return;
    }
    ip = dst->planes + pli;
    w = state->frame_width >> ip->xdec;
    h = state->frame_height >> ip->ydec;
    for (y = 0; y < h; y++) {
      memcpy(ip->data + (size_t)y*ip->ystride, D.rec[pli] + (size_t)y*w, w);
    }
    return;
  }
  if (!on_device() || lossless_p) {
    od_coeff_to_ref_plane_cpu(state, dst, pli, src, lossless_p);
    return;
  }
  if (pli == 0) {
    int rc;
    double t_a;
    t_a = now_s();
    if (tm_mark > 0) tm_parse += t_a - tm_mark;
    tm_mark = 0;
    rc = D.idct_skipped == 0 ? -100 : device_frame(state);
    tm_tail += now_s() - t_a;
    if (rc != 0) {
      stage_failed("pixel-domain stage", rc);
      /* The device pass for a frame whose inverse transforms were skipped has failed (or,
         idct_skipped == 0, a frame reached this point without a single block: malformed).
         No picture exists: the frame is reported as failed - daala_decode_packet_in returns
         OD_EFAULT - and the plane is written from what is there. */
      D.failed = 1;
    }
    D.idct_skipped = 0;
  }
  if (D.failed) {
    od_coeff_to_ref_plane_cpu(state, dst, pli, src, lossless_p);
    return;
  }
  ip = dst->planes + pli;
  w = state->frame_width >> ip->xdec;
  h = state->frame_height >> ip->ydec;
  for (y = 0; y < h; y++) {
    memcpy(ip->data + (size_t)y*ip->ystride, D.rec[pli] + (size_t)y*w, w);
  }
  /* the device copy of this reference image was written by device_frame() from the same
     reconstruction: it is current */
  if (D.mc_resident >= 0 && dst == state->ref_imgs + D.mc_resident) D.mc_dirty[D.mc_resident] = 0;
}

/* daala_decode_packet_in (src/decode.c:1159), the decoder's entry point, keeps its signature
   and its error contract: the build renames the reference's definition to *_cpu and this
   binding runs it.  On a thread that decodes through the device a failed device stage -
   prediction, prediction pyramid, pixel-domain tail - makes the call return OD_EFAULT
   instead of a picture that was not computed; nothing in the library aborts. */
int daala_decode_packet_in_cpu(daala_dec_ctx *dec, const daala_packet *op);
int daala_decode_packet_in(daala_dec_ctx *dec, const daala_packet *op) {
  int rc;
  if (D.ctx == NULL || dec == NULL) return daala_decode_packet_in_cpu(dec, op);
  /* after a failed frame the decoder's reference images are not what the stream expects: every
     packet keeps failing until a keyframe resynchronises it */
  if (D.poisoned == dec && op != NULL && !daala_packet_iskeyframe((daala_packet *)op)) return OD_EFAULT;
  D.dec = (od_dec_ctx *)dec;
  D.md_valid = 0;
  D.haar_frame = 0;
  D.failed = 0;
  D.idct_skipped = 0;
  D.haar_skipped = 0;
  D.ds_on = 0;
  {
    static int dbg = -1;
    double t_a;
    if (dbg < 0) dbg = getenv("HIPDEC_DEBUG") != NULL ? atoi(getenv("HIPDEC_DEBUG")) : 0;
    tm_mc = tm_pyr = tm_tail = tm_parse = tm_mark = 0;
    t_a = now_s();
    rc = daala_decode_packet_in_cpu(dec, op);
    if (dbg >= 2) {
      fprintf(stderr, "daala_hipdec: packet %.2f ms: prediction %.2f, pyramid+gains %.2f, parse %.2f, synthesis+tail %.2f\n",
       1e3*(now_s() - t_a), 1e3*tm_mc, 1e3*tm_pyr, 1e3*tm_parse, 1e3*tm_tail);
    }
  }
  D.dec = NULL;
  D.md_valid = 0;
  D.ds_on = 0;
  if (rc >= 0 && D.failed) {
    rc = OD_EFAULT;
    D.poisoned = dec;
  }
  else if (rc >= 0 && D.poisoned == dec) D.poisoned = NULL;
  return rc;
}

/* Encoder threads: od_state_mc_predict below serves them too.  Returns and clears this
   thread's failure flag (hip_enc_glue.c fails the frame). */
int od_hipdec_take_failure(void) {
  int f;
  f = D.failed;
  D.failed = 0;
  return f;
}

/* Test hook: the next device pass of this process's decoder threads fails as if the
   runtime had returned an error (n > 0: the n-th from now). */
static int g_fail_after;
void od_hipdec_test_fail_after(int n) {
  g_fail_after = n;
}
static int injected_failure(void) {
  if (g_fail_after > 0 && __sync_sub_and_fetch(&g_fail_after, 1) == 0) return 1;
  return 0;
}

/* ------------------------------------------------------------------------ */
typedef struct djob {
  const od_hipenc_params *p;
  int nframes;
  const unsigned char *hdr;
  long hdr_bytes;
  const unsigned char **pkt;     /* per frame: pointer to the packet bytes */
  long *pkt_len;
  unsigned char *out;
  size_t frame_bytes;
  int use_device;
  int device;
  pthread_mutex_t mu;
  int next;
  int failed;
  double t_device;
  double t0;
  double t_end;
  int ready;
  pthread_cond_t cv;
  int go;
} djob;

static daala_dec_ctx *make_decoder(const djob *J) {
  daala_info di;
  daala_comment dc;
  daala_setup_info *dsi;
  daala_dec_ctx *dec;
  long o;
  dsi = NULL;
  daala_info_init(&di);
  daala_comment_init(&dc);
  o = 0;
  while (o + 4 <= J->hdr_bytes) {
    daala_packet dp;
    long n;
    n = J->hdr[o] | J->hdr[o + 1] << 8 | J->hdr[o + 2] << 16 | (long)J->hdr[o + 3] << 24;
    /* a length that runs past the caller's buffer is a corrupt container, not a packet */
    if (n < 0 || n > J->hdr_bytes - o - 4) {
      daala_setup_free(dsi);
      daala_comment_clear(&dc);
      return NULL;
    }
    memset(&dp, 0, sizeof(dp));
    dp.packet = (unsigned char *)J->hdr + o + 4;
    dp.bytes = n;
    dp.b_o_s = o == 0;
    if (daala_decode_header_in(&di, &dc, &dsi, &dp) < 0) {
      daala_setup_free(dsi);
      daala_comment_clear(&dc);
      return NULL;
    }
    o += 4 + n;
  }
  /* the pictures are copied out with the caller's dimensions (dworker): they must be the
     stream's, or the copy would run past the decoder's planes */
  if (di.pic_width != J->p->pic_width || di.pic_height != J->p->pic_height) {
    daala_setup_free(dsi);
    daala_comment_clear(&dc);
    return NULL;
  }
  dec = daala_decode_create(&di, dsi);
  daala_setup_free(dsi);
  daala_comment_clear(&dc);
  return dec;
}

static void *dworker(void *arg) {
  djob *J;
  daala_dec_ctx *dec;
  od_state *st;
  int pli;
  int i;
  static const od_dct_func_2d hooks[OD_NBSIZES] = {hook_idct4, hook_idct8, hook_idct16,
   hook_idct32};
  J = (djob *)arg;
  memset(&D, 0, sizeof(D));
  D.check = J->p->check;
  D.device = J->device;
  dec = make_decoder(J);
  st = dec != NULL ? &((od_dec_ctx *)dec)->state : NULL;
  if (dec != NULL && J->use_device) {
    od_hip_geometry g;
    memset(&g, 0, sizeof(g));
    g.pic_width = st->info.pic_width;
    g.pic_height = st->info.pic_height;
    g.frame_width = st->frame_width;
    g.frame_height = st->frame_height;
    g.nplanes = 3;
    g.xdec[1] = g.xdec[2] = 1;
    g.nslots = 1;
    D.ctx = od_hip_ctx_create(J->device, &g);
    for (pli = 0; pli < 3; pli++) {
      size_t np;
      void *mem;
      np = (size_t)(st->frame_width >> (pli > 0))*(st->frame_height >> (pli > 0));
      /* The picture buffers cross PCIe every frame: page-lock them.  They are
         page-aligned and padded to whole pages so that the locked range never
         shares a page with anybody else's memory (a registered range that covers
         part of a neighbouring heap block makes the runtime treat that block as
         pinned too).  The reference's own dtmp planes are left alone for that
         reason.  Failure to lock only costs speed. */
      np = (np + 4095) & ~(size_t)4095;
      mem = NULL;
      if (posix_memalign(&mem, 4096, np) != 0) mem = NULL;
      D.rec[pli] = (unsigned char *)mem;
      D.pinned[pli] = mem != NULL && od_hip_host_register(mem, np) == 0;
      /* coefficient upload: dtmp -> own page-locked staging plane -> DMA (a copy
         straight from the reference's pageable plane goes through the runtime's
         staging buffers in small synchronous chunks) */
      np = (np*sizeof(od_coeff) + 4095) & ~(size_t)4095;
      mem = NULL;
      if (posix_memalign(&mem, 4096, np) != 0) mem = NULL;
      D.stage[pli] = (od_coeff *)mem;
      D.pinned[3 + pli] = mem != NULL && od_hip_host_register(mem, np) == 0;
      if (!D.pinned[3 + pli]) {
        free(mem);
        D.stage[pli] = NULL;
      }
    }
    for (i = 0; i < OD_NBSIZES; i++) {
      static const od_dct_func_2d fhooks[OD_NBSIZES] = {hook_md_fdct4, hook_md_fdct8, hook_md_fdct16,
       hook_md_fdct32};
      static const od_dct_func_2d vfdct[OD_NBSIZES] = {od_hipenc_fdct4x4, od_hipenc_fdct8x8,
       od_hipenc_fdct16x16, od_hipenc_fdct32x32};
      static const od_dct_func_2d vidct[OD_NBSIZES] = {od_hipenc_idct4x4, od_hipenc_idct8x8,
       od_hipenc_idct16x16, od_hipenc_idct32x32};
      /* what the device does not take runs on the host vector unit (hip_dct_host.c), except
         in check mode where the reference's C transforms are the checker */
      D.idct_cpu[i] = D.check ? st->opt_vtbl.idct_2d[i] : vidct[i];
      st->opt_vtbl.idct_2d[i] = hooks[i];
      D.fdct_cpu[i] = D.check ? st->opt_vtbl.fdct_2d[i] : vfdct[i];
      st->opt_vtbl.fdct_2d[i] = fhooks[i];
    }
  }
  pthread_mutex_lock(&J->mu);
  if (dec == NULL || (J->use_device && D.ctx == NULL)) J->failed = 1;
  J->ready++;
  pthread_cond_broadcast(&J->cv);
  while (!J->go) pthread_cond_wait(&J->cv, &J->mu);
  while (!J->failed && J->next < J->nframes) {
    daala_packet dp;
    od_img img;
    od_img *o;
    int f;
    int rc;
    f = J->next++;
    pthread_mutex_unlock(&J->mu);
    memset(&dp, 0, sizeof(dp));
    dp.packet = (unsigned char *)J->pkt[f];
    dp.bytes = J->pkt_len[f];
    rc = daala_decode_packet_in(dec, &dp);
    if (rc >= 0) {
      /* This frame's picture (od_img_copy into output_img, src/decode.c:1268).
         daala_decode_img_out is called once per packet, as the reference's player
         does: it only advances the two-entry reorder queue (it may hand back the
         previous picture, and repeats its last answer when called again). */
      o = ((od_dec_ctx *)dec)->output_img + ((od_dec_ctx *)dec)->curr_dec_frame;
      unsigned char *dst = J->out + J->frame_bytes*f;
      for (pli = 0; pli < 3; pli++) {
        int pw;
        int ph;
        int y;
        pw = (J->p->pic_width + (pli > 0)) >> (pli > 0);
        ph = (J->p->pic_height + (pli > 0)) >> (pli > 0);
        for (y = 0; y < ph; y++) {
          memcpy(dst, o->planes[pli].data + (size_t)y*o->planes[pli].ystride, pw);
          dst += pw;
        }
      }
      (void)daala_decode_img_out(dec, &img);
    }
    pthread_mutex_lock(&J->mu);
    if (rc < 0) J->failed = 1;
  }
  J->t_device += D.t_device;
  g_mc_dev_frames += mc_dev_frames;
  g_tail_dev_frames += tail_dev_frames;
  tail_dev_frames = 0;
  g_md_hits += D.md_hits;
  g_ds_frames += ds_frames;
  g_ref_resident_frames += ref_resident_frames;
  ref_resident_frames = 0;
  g_ds_check_fail += ds_check_fail;
  g_ds_wide_bands += ds_wide_bands;
  g_md_check_fail += D.md_check_fail;
  g_mc_check_fail += mc_check_fail;
  mc_dev_frames = mc_check_fail = 0;
  {
    double t;
    t = now_s();
    if (t > J->t_end) J->t_end = t;
  }
  pthread_mutex_unlock(&J->mu);
  for (pli = 0; pli < 3; pli++) {
    int lvl;
    for (lvl = 0; lvl < 4; lvl++) {
      if (D.md[pli][lvl] != NULL) {
        (void)od_hip_host_unregister(D.md[pli][lvl]);
        free(D.md[pli][lvl]);
      }
    }
    if (D.pinned[pli]) od_hip_host_unregister(D.rec[pli]);
    if (D.pinned[3 + pli]) od_hip_host_unregister(D.stage[pli]);
    free(D.stage[pli]);
  }
  od_hipdec_thread_cleanup();
  if (D.ds != NULL) od_hip_dsynth_destroy(D.ds);
  free(D.ds_qm_seen);
  if (D.ctx != NULL) od_hip_ctx_destroy(D.ctx);
  for (pli = 0; pli < 3; pli++) free(D.rec[pli]);
  if (dec != NULL) daala_decode_free(dec);
  return NULL;
}

/* ------------------------------------------------------------------------ */
/* od_state_mc_predict (src/state.c:993): the motion-compensated prediction of a whole
   inter frame.  It is a pure function of the motion vector grid and the reference frames,
   both final when the decoder (src/decode.c:1248) or the encoder (src/encode.c:2219) asks
   for it, so every leaf of the grid's quadtree (od_state_pred_block, :735-786) becomes one
   entry of a block list and od_hip_mc_predict predicts each plane in one launch. */
typedef struct mc_list {
  od_hip_mc_block *b;
  int n;
  int cap;
} mc_list;

static int mc_push(mc_list *L, const od_hip_mc_block *m) {
  if (L->n == L->cap) {
    od_hip_mc_block *q;
    L->cap = L->cap ? 2*L->cap : 1024;
    q = (od_hip_mc_block *)realloc(L->b, sizeof(*q)*L->cap);
    if (q == NULL) return -1;
    L->b = q;
  }
  L->b[L->n++] = *m;
  return 0;
}

/* od_state_pred_block + od_state_pred_block_from_setup (src/state.c:689-786), the leaf
   recorded instead of predicted */
static int mc_collect(od_state *state, mc_list *L, int pli, int xdec, int ydec, int vx,
 int vy, int log_mvb_sz) {
  int half;
  half = 1 << log_mvb_sz >> 1;
  if (log_mvb_sz > 0 && state->mv_grid[vy + half][vx + half].valid) {
    if (mc_collect(state, L, pli, xdec, ydec, vx, vy, log_mvb_sz - 1)) return -1;
    if (mc_collect(state, L, pli, xdec, ydec, vx + half, vy, log_mvb_sz - 1)) return -1;
    if (mc_collect(state, L, pli, xdec, ydec, vx, vy + half, log_mvb_sz - 1)) return -1;
    return mc_collect(state, L, pli, xdec, ydec, vx + half, vy + half, log_mvb_sz - 1);
  }
  else {
    od_hip_mc_block m;
    const int *dxp;
    const int *dyp;
    int oc;
    int sp;
    int k;
    if (log_mvb_sz < OD_LOG_MVB_DELTA0) {
      int mask;
      mask = (1 << (log_mvb_sz + 1)) - 1;
      oc = !!(vx & mask);
      if (vy & mask) oc = 3 - oc;
      sp = state->mv_grid[vy + (OD_VERT_DY[(oc + 1) & 3] << log_mvb_sz)]
       [vx + (OD_VERT_DX[(oc + 1) & 3] << log_mvb_sz)].valid
       | state->mv_grid[vy + (OD_VERT_DY[(oc + 3) & 3] << log_mvb_sz)]
       [vx + (OD_VERT_DX[(oc + 3) & 3] << log_mvb_sz)].valid << 1;
    }
    else {
      oc = 0;
      sp = 3;
    }
    dxp = OD_VERT_SETUP_DX[oc][sp];
    dyp = OD_VERT_SETUP_DY[oc][sp];
    memset(&m, 0, sizeof(m));
    m.x = vx << (OD_LOG_MVBSIZE_MIN - xdec);
    m.y = vy << (OD_LOG_MVBSIZE_MIN - ydec);
    m.log_xblk_sz = log_mvb_sz + OD_LOG_MVBSIZE_MIN - xdec;
    m.log_yblk_sz = log_mvb_sz + OD_LOG_MVBSIZE_MIN - ydec;
    m.oc = oc;
    m.s = sp;
    for (k = 0; k < 4; k++) {
      const od_mv_grid_pt *g;
      int mvx;
      int mvy;
      g = state->mv_grid[vy + dyp[k]*(1 << log_mvb_sz)] + vx + dxp[k]*(1 << log_mvb_sz);
      if (g->ref == OD_FRAME_NEXT) {
        mvx = g->mv1[0];
        mvy = g->mv1[1];
      }
      else {
        mvx = g->mv[0];
        mvy = g->mv[1];
      }
      m.mvx[k] = (int32_t)OD_DIV_POW2_RE(mvx, xdec);
      m.mvy[k] = (int32_t)OD_DIV_POW2_RE(mvy, ydec);
      m.ref[k] = state->ref_imgi[g->ref];        /* index into state->ref_imgs */
      if (m.ref[k] < 0 || m.ref[k] > OD_FRAME_MAX) return -1;
    }
    return mc_push(L, &m);
  }
}

/* The thread's prediction object with every reference image the vector grid can name resident
   and current: an image is uploaded when it changed since its last upload (src/state.c:236-300:
   OD_BUFFER_PADDING >> dec samples of padding on every side of the frame).  img_dst: the image
   about to be predicted into (this frame's SELF: rewritten, never read as a reference), or NULL. */
static int mc_refresh_refs(od_state *state, od_img *img_dst, int nplanes) {
  int rc;
  int k;
  int pli;
  rc = 0;
  if (D.mc == NULL) {
    D.mc = od_hip_mc_create(D.device, OD_FRAME_MAX + 1);
    if (D.mc == NULL) return -1;
    D.mc_state = NULL;
  }
  if (D.mc_state != state) {
    D.mc_state = state;
    memset(D.mc_dirty, 1, sizeof(D.mc_dirty));
  }
  if (img_dst != NULL && img_dst >= state->ref_imgs && img_dst <= state->ref_imgs + OD_FRAME_MAX) {
    D.mc_dirty[img_dst - state->ref_imgs] = 1;
  }
  for (k = 0; k <= OD_FRAME_MAX && rc == 0; k++) {
    int used;
    int t;
    used = 0;
    for (t = 0; t < OD_FRAME_MAX + 1; t++) used |= state->ref_imgi[t] == k && t != OD_FRAME_SELF;
    if (!used || !D.mc_dirty[k] || state->ref_imgs + k == img_dst) continue;
    for (pli = 0; pli < nplanes && rc == 0; pli++) {
      const od_img_plane *rp;
      int px;
      int py;
      rp = state->ref_imgs[k].planes + pli;
      px = OD_BUFFER_PADDING >> rp->xdec;
      py = OD_BUFFER_PADDING >> rp->ydec;
      rc = od_hip_mc_set_ref(D.mc, pli, k, rp->data - (ptrdiff_t)py*rp->ystride - px, rp->ystride,
       (state->frame_height + 2*OD_BUFFER_PADDING) >> rp->ydec, px, py);
    }
    if (rc == 0) D.mc_dirty[k] = 0;
  }
  return rc;
}

/* The batch stages of the encoder's motion search (mcenc_tail.c: od_mv_est_calc_sads): the SAD
   of the OBMC prediction of every item against the frame being coded, on this thread's
   prediction object.  Returns 1 when sad[] was written by the device, 0 when this thread has no
   device (the caller runs the reference's loop), < 0 when a device stage failed (the frame
   fails, like a failed od_state_mc_predict). */
int od_hipdec_mc_sad_items(od_state *state, const od_img *input, int nplanes,
 const od_hip_mc_sad_item *items, int nitems, int32_t *sad) {
  int pli;
  int rc;
  if (!od_hipenc_device_thread() || D.failed) return 0;
  if (state->full_precision_references || nplanes < 1 || nplanes > 3 || input->nplanes < nplanes) return 0;
  for (pli = 0; pli < nplanes; pli++) {
    if (input->planes[pli].xstride != 1 || input->planes[pli].xdec > 1 || input->planes[pli].ydec > 1) return 0;
  }
  if (injected_failure()) rc = -1;
  else rc = mc_refresh_refs(state, NULL, nplanes);
  /* the frame being coded: uploaded once per frame (the EPZS windows may have done it already) */
  if (rc == 0 && (D.mc_src_state != state || D.mc_src_time != state->cur_time)) {
    for (pli = 0; pli < nplanes && rc == 0; pli++) {
      const od_img_plane *ip;
      ip = input->planes + pli;
      rc = od_hip_mc_set_src(D.mc, pli, ip->data, ip->ystride, state->frame_width >> ip->xdec,
       state->frame_height >> ip->ydec, ip->xdec, ip->ydec);
    }
    if (rc == 0) {
      D.mc_src_state = state;
      D.mc_src_time = state->cur_time;
    }
  }
  if (rc == 0) {
    rc = od_hip_mc_sad_items(D.mc, nplanes, state->info.pic_width, state->info.pic_height, items, nitems, sad);
  }
  if (rc != 0) {
    D.failed = 1;
    stage_failed("motion search SADs", rc);
    return -1;
  }
  return 1;
}

/* The block-matching windows of the EPZS initialisation (mcenc_tail.c) on this thread's
   prediction object: references current, the frame being coded resident (od_mv_est_init_mvs runs
   before od_mv_est_calc_sads: this is the frame's first use of the source planes).  Return codes
   as od_hipdec_mc_sad_items. */
int od_hipdec_mc_bma_windows(od_state *state, const od_img *input, int nplanes,
 const od_hip_mc_bma_rec *recs, int nrec, int radius, int32_t *out) {
  int pli;
  int rc;
  if (!od_hipenc_device_thread() || D.failed) return 0;
  if (state->full_precision_references || nplanes < 1 || nplanes > 3 || input->nplanes < nplanes) return 0;
  for (pli = 0; pli < nplanes; pli++) {
    if (input->planes[pli].xstride != 1 || input->planes[pli].xdec > 1 || input->planes[pli].ydec > 1) return 0;
  }
  if (injected_failure()) rc = -1;
  else rc = mc_refresh_refs(state, NULL, nplanes);
  /* the source planes: once per frame (D.mc_src_frame: the encoder's frame counter) */
  if (rc == 0 && (D.mc_src_state != state || D.mc_src_time != state->cur_time)) {
    for (pli = 0; pli < nplanes && rc == 0; pli++) {
      const od_img_plane *ip;
      ip = input->planes + pli;
      rc = od_hip_mc_set_src(D.mc, pli, ip->data, ip->ystride, state->frame_width >> ip->xdec,
       state->frame_height >> ip->ydec, ip->xdec, ip->ydec);
    }
    if (rc == 0) {
      D.mc_src_state = state;
      D.mc_src_time = state->cur_time;
    }
  }
  if (rc == 0) {
    rc = od_hip_mc_bma_windows(D.mc, nplanes, state->info.pic_width, state->info.pic_height, recs, nrec,
     radius, out);
  }
  if (rc != 0) {
    D.failed = 1;
    stage_failed("motion search block-matching windows", rc);
    return -1;
  }
  return 1;
}

static int mc_predict_device(od_state *state, od_img *img_dst) {
  mc_list L;
  int pli;
  int rc;
  int to_ctx;
  rc = 0;
  memset(&L, 0, sizeof(L));
  if (state->full_precision_references) return -1;
  /* A decoder thread's prediction has one consumer, the forward pyramid of the thread's
     context (md_pyramid): it is predicted straight into the context's picture planes and never
     visits the host - unless somebody wants to look at it (check mode, user_mc_img). */
  to_ctx = D.ctx != NULL && D.dec != NULL && state == &D.dec->state && !D.check
   && D.dec->user_mc_img == NULL && img_dst->nplanes == 3 && state->frame_type == OD_P_FRAME
   && state->quantizer[0] > 0;
  D.pred_on_device = 0;
  rc = mc_refresh_refs(state, img_dst, img_dst->nplanes);
  for (pli = 0; pli < img_dst->nplanes && rc == 0; pli++) {
    od_img_plane *dp;
    int xdec;
    int ydec;
    int vx;
    int vy;
    dp = img_dst->planes + pli;
    xdec = dp->xdec;
    ydec = dp->ydec;
    if (dp->xstride != 1) {
      rc = -1;
      break;
    }
    L.n = 0;
    for (vy = 0; vy < state->nvmvbs && rc == 0; vy += OD_MVB_DELTA0) {
      for (vx = 0; vx < state->nhmvbs && rc == 0; vx += OD_MVB_DELTA0) {
        rc = mc_collect(state, &L, pli, xdec, ydec, vx, vy, OD_LOG_MVB_DELTA0);
      }
    }
    /* a block may name any image the grid refers to: all of those were refreshed above */
    if (rc == 0) {
      if (to_ctx) rc = od_hip_mc_predict_ctx(D.mc, pli, L.b, L.n, D.ctx, 0);
      else {
        rc = od_hip_mc_predict(D.mc, pli, L.b, L.n, dp->data, dp->ystride, state->frame_width >> xdec,
         state->frame_height >> ydec);
      }
    }
  }
  free(L.b);
  if (rc == 0 && to_ctx) D.pred_on_device = 1;
  return rc;
}

/* encoder workers: the device their prediction object lives on */
void od_hipdec_set_device(int device) {
  D.device = device;
}

/* per-thread clean-up (worker exit) */
void od_hipdec_thread_cleanup(void) {
  if (D.mc != NULL) od_hip_mc_destroy(D.mc);
  D.mc = NULL;
  D.mc_state = NULL;
}

void od_state_mc_predict(od_state *state, od_img *img_dst) {
  if ((D.ctx != NULL || od_hipenc_device_thread()) && !D.failed) {
    double t_a;
    t_a = now_s();
    if (!injected_failure() && mc_predict_device(state, img_dst) == 0) {
      tm_mc += now_s() - t_a;
      mc_dev_frames++;
      if (D.ctx != NULL && D.dec != NULL && state == &D.dec->state
       && state->info.nplanes == 3 && state->frame_type == OD_P_FRAME) {
        int rc;
        t_a = now_s();
        rc = md_pyramid(state, img_dst);
        tm_pyr += now_s() - t_a;
        tm_mark = now_s();
        D.md_valid = rc == 0;
        if (!D.md_valid) {
          D.failed = 1;      /* surfaced by daala_decode_packet_in, no silent host path */
          stage_failed("prediction pyramid", rc);
        }
      }
      /* encoder threads: the frame's prediction exists - the P-frame feed can run */
      if (D.ctx == NULL && od_hipenc_device_thread() && od_hipenc_pframe_feed(state, img_dst) < 0) D.failed = 1;
      if (D.check || od_hipenc_check_mode()) {
        /* OD_CHECKASM: the reference's own prediction of the same frame; its result stays */
        unsigned char *keep[3];
        int pli;
        for (pli = 0; pli < img_dst->nplanes && pli < 3; pli++) {
          od_img_plane *dp;
          size_t bytes;
          dp = img_dst->planes + pli;
          bytes = (size_t)dp->ystride*(state->frame_height >> dp->ydec);
          keep[pli] = (unsigned char *)malloc(bytes);
          if (keep[pli] != NULL) memcpy(keep[pli], dp->data, bytes);
        }
        od_state_mc_predict_cpu(state, img_dst);
        for (pli = 0; pli < img_dst->nplanes && pli < 3; pli++) {
          od_img_plane *dp;
          int y;
          dp = img_dst->planes + pli;
          for (y = 0; keep[pli] != NULL && y < state->frame_height >> dp->ydec; y++) {
            if (memcmp(keep[pli] + (size_t)y*dp->ystride, dp->data + (size_t)y*dp->ystride,
             state->frame_width >> dp->xdec) != 0) {
              mc_check_fail++;
              break;
            }
          }
          free(keep[pli]);
        }
      }
      return;
    }
    /* the frame fails: daala_decode_packet_in returns OD_EFAULT on decoder threads,
       encode_frame (hip_enc_glue.c) fails the job on encoder threads; the reference's
       prediction below only keeps the memory defined */
    D.failed = 1;
  }
  od_state_mc_predict_cpu(state, img_dst);
}

long od_hipdec_decode_frames(const od_hipenc_params *p, const unsigned char *hdr,
 long hdr_bytes, int nframes, const unsigned char *pkts, long pkt_bytes,
 int use_device, int device, unsigned char *frames_out, double *seconds,
 double *device_seconds) {
  djob J;
  pthread_t *th;
  int nw;
  int i;
  long o;
  if (p == NULL || hdr == NULL || pkts == NULL || frames_out == NULL || nframes < 1) {
    return OD_HIP_EFAULT;
  }
  if (use_device && od_hip_device_count() <= device) return OD_HIP_ENODEV;
  g_mc_dev_frames = g_mc_check_fail = 0;
  g_tail_dev_frames = 0;
  g_md_hits = g_md_check_fail = 0;
  g_ds_frames = g_ds_check_fail = g_ds_wide_bands = 0;
  g_ref_resident_frames = 0;
  {
    const char *e;
    e = getenv("HIPDEC_REF_RESIDENT");
    ref_resident_on = e == NULL || atoi(e) != 0;
  }
  memset(&J, 0, sizeof(J));
  J.p = p;
  J.nframes = nframes;
  J.hdr = hdr;
  J.hdr_bytes = hdr_bytes;
  J.out = frames_out;
  J.frame_bytes = (size_t)p->pic_width*p->pic_height
   + 2*(size_t)((p->pic_width + 1) >> 1)*((p->pic_height + 1) >> 1);
  J.use_device = use_device;
  J.device = device;
  J.pkt = (const unsigned char **)calloc(nframes, sizeof(*J.pkt));
  J.pkt_len = (long *)calloc(nframes, sizeof(*J.pkt_len));
  if (J.pkt == NULL || J.pkt_len == NULL) {
    free(J.pkt);
    free(J.pkt_len);
    return OD_HIP_EFAULT;
  }
  o = 0;
  for (i = 0; i < nframes; i++) {
    long n;
    if (o + 4 > pkt_bytes) {
      free(J.pkt);
      free(J.pkt_len);
      return OD_HIP_EINVAL;
    }
    n = pkts[o] | pkts[o + 1] << 8 | pkts[o + 2] << 16 | (long)pkts[o + 3] << 24;
    if (n < 0 || n > pkt_bytes - o - 4) {       /* truncated or corrupt packet blob */
      free(J.pkt);
      free(J.pkt_len);
      return OD_HIP_EINVAL;
    }
    J.pkt[i] = pkts + o + 4;
    J.pkt_len[i] = n;
    o += 4 + n;
  }
  nw = p->nworkers < 1 ? 1 : p->nworkers;
  if (nw > nframes) nw = nframes;
  /* workers take frames independently, which only keyframes allow: a stream with a packet
     that is not one (daala_packet_iskeyframe, src/internal.c:654) is decoded in order by
     one worker */
  for (i = 0; i < nframes; i++) {
    if (J.pkt_len[i] > 0 && !(J.pkt[i][0] & 0x40)) nw = 1;
  }
  th = (pthread_t *)calloc(nw, sizeof(*th));
  if (th == NULL) {
    free(J.pkt);
    free(J.pkt_len);
    return OD_HIP_EFAULT;
  }
  pthread_mutex_init(&J.mu, NULL);
  pthread_cond_init(&J.cv, NULL);
  for (i = 0; i < nw; i++) pthread_create(&th[i], NULL, dworker, &J);
  pthread_mutex_lock(&J.mu);
  while (J.ready < nw) pthread_cond_wait(&J.cv, &J.mu);
  J.t0 = now_s();
  J.go = 1;
  pthread_cond_broadcast(&J.cv);
  pthread_mutex_unlock(&J.mu);
  for (i = 0; i < nw; i++) pthread_join(th[i], NULL);
  if (seconds != NULL) *seconds = J.t_end - J.t0;
  if (device_seconds != NULL) *device_seconds = J.t_device;
  free(J.pkt);
  free(J.pkt_len);
  free(th);
  pthread_mutex_destroy(&J.mu);
  pthread_cond_destroy(&J.cv);
  return J.failed ? OD_HIP_EINVAL : nframes;
}
