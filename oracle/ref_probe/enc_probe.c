/* Probe translation unit (TEST INFRASTRUCTURE, dev container only).
 * #includes the reference's src/encode.c (through -I/root/reference/src, the
 * technique of the reference's own src/tests/test_coef_coder.c:25-34) to reach
 * the static od_compute_dcts, and exports a wrapper that runs the reference's
 * forward path of one plane - od_ref_buf_to_coeff, od_apply_prefilter_frame_sbs,
 * od_compute_dcts per superblock (src/encode.c:2425-2428, :2492) - on caller
 * supplied padded pixels and a caller supplied block-size map. */
#include "encode.c"

/* pix: padded plane of (frame_width>>dec) x (frame_height>>dec) bytes.
   bsize8: (nvsb*4) x (nhsb*4) luma block-size map, values 0..3.
   Returns 0 on success; writes the lapped plane c and the coefficients d. */
int probe_forward_plane(int pic_w, int pic_h, int pli, int keyframe,
 const unsigned char *pix, const unsigned char *bsize8, od_coeff *c_out,
 od_coeff *d_out) {
  daala_info di;
  daala_enc_ctx *enc;
  od_mb_enc_ctx mbctx;
  od_state *state;
  int xdec, ydec, w, h, sbx, sby, i, j;
  daala_info_init(&di);
  di.pic_width = pic_w;
  di.pic_height = pic_h;
  di.nplanes = 3;
  di.plane_info[0].xdec = di.plane_info[0].ydec = 0;
  di.plane_info[1].xdec = di.plane_info[1].ydec = 1;
  di.plane_info[2].xdec = di.plane_info[2].ydec = 1;
  di.timebase_numerator = 30;
  di.timebase_denominator = 1;
  di.frame_duration = 1;
  di.pixel_aspect_numerator = di.pixel_aspect_denominator = 1;
  di.keyframe_rate = 1;
  enc = daala_encode_create(&di);
  if (enc == NULL) return -1;
  state = &enc->state;
  xdec = di.plane_info[pli].xdec;
  ydec = di.plane_info[pli].ydec;
  w = state->frame_width >> xdec;
  h = state->frame_height >> ydec;
  for (i = 0; i < state->nvsb*4; i++)
    for (j = 0; j < state->nhsb*4; j++)
      state->bsize[i*state->bstride + j] = bsize8[i*state->nhsb*4 + j];
  memset(&mbctx, 0, sizeof(mbctx));
  mbctx.c = state->ctmp[pli];
  mbctx.d = state->dtmp;
  mbctx.is_keyframe = keyframe;
  od_ref_buf_to_coeff(state, state->ctmp[pli], w, 0, (unsigned char *)pix, 1, w,
   w, h);
  od_apply_prefilter_frame_sbs(state->ctmp[pli], w, state->nhsb, state->nvsb,
   xdec, ydec);
  for (sby = 0; sby < state->nvsb; sby++)
    for (sbx = 0; sbx < state->nhsb; sbx++)
      od_compute_dcts(enc, &mbctx, pli, sbx, sby, OD_NBSIZES - 1, xdec, ydec, 0);
  memcpy(c_out, state->ctmp[pli], sizeof(od_coeff)*w*h);
  memcpy(d_out, state->dtmp[pli], sizeof(od_coeff)*w*h);
  daala_encode_free(enc);
  return 0;
}

/* ------------------------------------------------------------------------ */
/* Whole-encoder drivers (the sequence of SURVEY.md Appendix B): used for the
   end-to-end anchors in tests/golden and as the CPU baseline (kind "reference")
   of bench.py.  Input: nframes frames of planar 4:2:0, Y then U then V, each
   plane dense (pic size, chroma (w+1)/2 x (h+1)/2). */
#include <time.h>

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9*ts.tv_nsec;
}

static daala_enc_ctx *make_encoder(int w, int h, int quant, int complexity,
 int masking, int keyrate) {
  daala_info di;
  daala_enc_ctx *enc;
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
  di.keyframe_rate = keyrate;
  enc = daala_encode_create(&di);
  if (enc == NULL) return NULL;
  daala_encode_ctl(enc, OD_SET_QUANT, &quant, sizeof(quant));
  daala_encode_ctl(enc, OD_SET_COMPLEXITY, &complexity, sizeof(complexity));
  daala_encode_ctl(enc, OD_SET_ACTIVITY_MASKING, &masking, sizeof(masking));
  {
    /* enc->use_dering is never initialised by daala_encode_create (it is only
       written by this ctl, src/encode.c:527); the reference CLI always sets it,
       default 1 (examples/encoder_example.c:675,900).  Do the same, or the
       deringing decisions depend on heap garbage. */
    int use_dering = 1;
    daala_encode_ctl(enc, OD_SET_DERING, &use_dering, sizeof(use_dering));
  }
  return enc;
}

/* Encodes nframes; returns total video packet bytes (headers excluded), or <0.
   fnv: FNV-1a-32 over all video packet bytes.  seconds: wall time spent inside
   daala_encode_img_in + daala_encode_packet_out only.  If pkt_out != NULL the
   packets are appended there (up to pkt_cap bytes, each prefixed by a 4-byte
   little-endian length). */
/* fdct_2d / idct_2d: optional replacement kernels installed into the context's
   od_state_opt_vtbl after creation (entries 0..3) - the drop-in seam of
   src/state.c:347-353 / src/x86/x86state.c:39-96 exercised at run time, without
   touching the reference sources.  NULL keeps the pure-C tables. */
/* Progress callback (bench.py's reference child): called after every frame with the frame
   index and the encode seconds so far. */
static void (*g_after_frame)(int f, double seconds);
void probe_set_frame_callback(void (*cb)(int f, double seconds)) {
  g_after_frame = cb;
}

long probe_encode_frames_vtbl(int w, int h, int nframes, int quant, int complexity,
 int masking, int keyrate, const unsigned char *frames, unsigned *fnv,
 double *seconds, unsigned char *pkt_out, long pkt_cap,
 const od_dct_func_2d *fdct_2d, const od_dct_func_2d *idct_2d,
 unsigned char *recon_out) {
  daala_enc_ctx *enc;
  od_img img;
  daala_packet dp;
  long total = 0, used = 0;
  unsigned hsh = 2166136261u;
  double t = 0;
  int f, cw = (w + 1) >> 1, ch = (h + 1) >> 1;
  size_t fsz = (size_t)w*h + 2*(size_t)cw*ch;
  enc = make_encoder(w, h, quant, complexity, masking, keyrate);
  if (enc == NULL) return -1;
  if (fdct_2d != NULL && idct_2d != NULL) {
    int i;
    for (i = 0; i < OD_NBSIZES; i++) {
      enc->state.opt_vtbl.fdct_2d[i] = fdct_2d[i];
      enc->state.opt_vtbl.idct_2d[i] = idct_2d[i];
    }
  }
  memset(&img, 0, sizeof(img));
  img.nplanes = 3;
  img.width = w;
  img.height = h;
  for (f = 0; f < nframes; f++) {
    unsigned char *base = (unsigned char *)frames + fsz*f;
    double t0;
    int left;
    img.planes[0].data = base;
    img.planes[0].xdec = img.planes[0].ydec = 0;
    img.planes[0].xstride = 1;
    img.planes[0].ystride = w;
    img.planes[0].bitdepth = 8;
    img.planes[1].data = base + (size_t)w*h;
    img.planes[2].data = base + (size_t)w*h + (size_t)cw*ch;
    img.planes[1].xdec = img.planes[1].ydec = 1;
    img.planes[2].xdec = img.planes[2].ydec = 1;
    img.planes[1].xstride = img.planes[2].xstride = 1;
    img.planes[1].ystride = img.planes[2].ystride = cw;
    img.planes[1].bitdepth = img.planes[2].bitdepth = 8;
    t0 = now_s();
    if (daala_encode_img_in(enc, &img, 0, 0, &left) < 0) {
      daala_encode_free(enc);
      return -2;
    }
    while (daala_encode_packet_out(enc, 0, &dp) > 0) {
      long i;
      t += now_s() - t0;
      for (i = 0; i < dp.bytes; i++) hsh = (hsh ^ dp.packet[i])*16777619u;
      total += dp.bytes;
      if (pkt_out != NULL && used + 4 + dp.bytes <= pkt_cap) {
        pkt_out[used] = dp.bytes & 255;
        pkt_out[used + 1] = (dp.bytes >> 8) & 255;
        pkt_out[used + 2] = (dp.bytes >> 16) & 255;
        pkt_out[used + 3] = (dp.bytes >> 24) & 255;
        memcpy(pkt_out + used + 4, dp.packet, dp.bytes);
        used += 4 + dp.bytes;
      }
      t0 = now_s();
    }
    t += now_s() - t0;
    if (g_after_frame != NULL) (*g_after_frame)(f, t);
  }
  if (recon_out != NULL) {
    /* reconstruction of the last frame, visible area, Y U V dense */
    od_img *rec = enc->state.ref_imgs + enc->state.ref_imgi[OD_FRAME_SELF];
    unsigned char *o = recon_out;
    int pli, y;
    for (pli = 0; pli < 3; pli++) {
      int pw = pli ? cw : w, ph = pli ? ch : h;
      for (y = 0; y < ph; y++) {
        memcpy(o, rec->planes[pli].data + (size_t)y*rec->planes[pli].ystride, pw);
        o += pw;
      }
    }
  }
  daala_encode_free(enc);
  if (fnv != NULL) *fnv = hsh;
  if (seconds != NULL) *seconds = t;
  return total;
}

long probe_encode_frames(int w, int h, int nframes, int quant, int complexity,
 int masking, int keyrate, const unsigned char *frames, unsigned *fnv,
 double *seconds, unsigned char *pkt_out, long pkt_cap) {
  return probe_encode_frames_vtbl(w, h, nframes, quant, complexity, masking,
   keyrate, frames, fnv, seconds, pkt_out, pkt_cap, NULL, NULL, NULL);
}

/* Per-frame encoder constants the device PVQ stage needs, read back after one
   keyframe: quantizer[3], pvq_qm_q4[3][OD_QM_SIZE], qm / qm_inv
   (OD_QM_BUFFER_SIZE int16 each; src/encode.c:3025-3050, :240). */
int probe_encoder_params(int quant, int masking, int *quantizer,
 unsigned char *pvq_qm_q4, int16_t *qm, int16_t *qm_inv) {
  unsigned char frame[64*64 + 2*32*32];
  unsigned fnv;
  daala_enc_ctx *enc;
  od_img img;
  daala_packet dp;
  int pli;
  int left;
  (void)fnv;
  memset(frame, 128, sizeof(frame));
  enc = make_encoder(64, 64, quant, 7, masking, 1);
  if (enc == NULL) return -1;
  memset(&img, 0, sizeof(img));
  img.nplanes = 3;
  img.width = img.height = 64;
  for (pli = 0; pli < 3; pli++) {
    img.planes[pli].data = frame + (pli == 0 ? 0 : pli == 1 ? 4096 : 4096 + 1024);
    img.planes[pli].xdec = img.planes[pli].ydec = pli > 0;
    img.planes[pli].xstride = 1;
    img.planes[pli].ystride = pli ? 32 : 64;
    img.planes[pli].bitdepth = 8;
  }
  if (daala_encode_img_in(enc, &img, 0, 0, &left) < 0) return -2;
  while (daala_encode_packet_out(enc, 0, &dp) > 0);
  for (pli = 0; pli < 3; pli++) {
    quantizer[pli] = enc->state.quantizer[pli];
    memcpy(pvq_qm_q4 + pli*OD_QM_SIZE, enc->state.pvq_qm_q4[pli], OD_QM_SIZE);
  }
  memcpy(qm, enc->state.qm, sizeof(int16_t)*OD_QM_BUFFER_SIZE);
  memcpy(qm_inv, enc->state.qm_inv, sizeof(int16_t)*OD_QM_BUFFER_SIZE);
  daala_encode_free(enc);
  return 0;
}

/* ------------------------------------------------------------------------ */
/* od_compute_dist (static, src/encode.c:1032-1058) on caller data, plus the
   per-coefficient weight table it builds from OD_QM8_Q4_HVS and OD_BASIS_MAG
   (src/encode.c:1018-1027) so that fixtures can carry it as data. */
double probe_compute_dist(int masking, const od_coeff *x, const od_coeff *y, int n,
 int bs) {
  daala_enc_ctx *enc;
  double d;
  enc = make_encoder(64, 64, 20, 7, masking, 1);
  if (enc == NULL) return -1;
  d = od_compute_dist(enc, (od_coeff *)x, (od_coeff *)y, n, bs);
  daala_encode_free(enc);
  return d;
}

void probe_dist_weights(int bs, double *mag2) {
  int i, j;
  for (i = 0; i < 8; i++) {
    for (j = 0; j < 8; j++) {
      double mag;
      mag = 16./OD_QM8_Q4_HVS[i*8 + j];
      mag *= OD_BASIS_MAG[0][bs][i << (bs - 1)]*OD_BASIS_MAG[0][bs][j << (bs - 1)];
      mag *= mag;
      mag2[i*8 + j] = mag;
    }
  }
}

/* od_dering (src/filter.c:1835) needs an od_state for its vtable: borrow one
   from a throw-away encoder context.  threshold is derived from q inside. */
int probe_dering(int16_t *y, int ystride, int16_t *x, int xstride, int ln, int sbx,
 int sby, int nhsb, int nvsb, int q, int xdec, int *dir, int pli,
 unsigned char *bskip, int skip_stride) {
  daala_enc_ctx *enc;
  enc = make_encoder(64, 64, 20, 7, 1, 1);
  if (enc == NULL) return -1;
  od_dering(&enc->state, y, ystride, x, xstride, ln, sbx, sby, nhsb, nvsb, q, xdec,
   (int (*)[OD_DERING_NBLOCKS])dir, pli, bskip, skip_stride);
  daala_encode_free(enc);
  return 0;
}

/* ------------------------------------------------------------------------ */
/* od_mc_predict (src/mc.c:2006) of one block through a live context's vtable (C entries:
   od_mc_predict1fmv8_c, od_mc_blend_full8_c, od_mc_blend_full_split8_c).  src[k]: pointer
   to the block's position in the k-th corner's reference plane. */
int probe_mc_predict(unsigned char *dst, int dystride, const unsigned char *src0,
 const unsigned char *src1, const unsigned char *src2, const unsigned char *src3,
 int systride, const int32_t *mvx, const int32_t *mvy, int oc, int s, int log_xblk_sz,
 int log_yblk_sz) {
  daala_enc_ctx *enc;
  const unsigned char *src[4];
  enc = make_encoder(64, 64, 20, 7, 1, 1);
  if (enc == NULL) return -1;
  src[0] = src0; src[1] = src1; src[2] = src2; src[3] = src3;
  od_mc_predict(&enc->state, dst, dystride, src, systride, mvx, mvy, oc, s, log_xblk_sz,
   log_yblk_sz);
  daala_encode_free(enc);
  return 0;
}
