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
