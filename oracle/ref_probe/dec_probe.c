/* Probe translation unit (TEST INFRASTRUCTURE, dev container only): drives the
 * reference encoder + decoder through their public API on one frame and dumps
 * what the decoder's pixel-domain tail consumed and produced, so that the device
 * tail (iDCT -> post-filters -> dering -> smoothing -> clamp) can be checked
 * against a REAL decode.  Uses only reference headers (struct layouts). */
#include <string.h>
#include "../include/daala/daalaenc.h"
#include "../include/daala/daaladec.h"
#include "decint.h"

/* Encodes ONE keyframe of the given 4:2:0 picture with the reference encoder
   (quantizer `quant`, complexity 7, masking as given), decodes it with the
   reference decoder and returns, for the padded frame (fw x fh, multiples of
   64): the dequantised coefficient planes dtmp[pli] the iDCTs read, the luma
   block-size map (1 byte / 8x8, dense), the skip maps (1 byte / 4x4 per plane,
   dense rows of fw/4 bytes), the dering flags (1 byte / 32x32), the quantizers,
   and the decoder's visible output planes.  Returns 0 on success. */
int probe_decode_dump(int w, int h, int quant, int masking, const unsigned char *frame,
 int *fw_out, int *fh_out, od_coeff *d0, od_coeff *d1, od_coeff *d2,
 unsigned char *bsize_out, unsigned char *bskip0, unsigned char *bskip1,
 unsigned char *bskip2, unsigned char *dering_flags, int *quantizer,
 unsigned char *out_y, unsigned char *out_u, unsigned char *out_v) {
  daala_info di;
  daala_comment dc;
  daala_enc_ctx *enc;
  daala_dec_ctx *dec;
  daala_setup_info *dsi = NULL;
  daala_info di2;
  daala_comment dc2;
  daala_packet dp;
  od_img img, out;
  od_state *st;
  int cw = (w + 1) >> 1, ch = (h + 1) >> 1, pli, i, j, left, rc, complexity = 7;
  od_coeff *dd[3];
  unsigned char *bs[3];
  unsigned char *oo[3];
  dd[0] = d0; dd[1] = d1; dd[2] = d2;
  bs[0] = bskip0; bs[1] = bskip1; bs[2] = bskip2;
  oo[0] = out_y; oo[1] = out_u; oo[2] = out_v;
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
  di.keyframe_rate = 1;
  enc = daala_encode_create(&di);
  if (enc == NULL) return -1;
  daala_encode_ctl(enc, OD_SET_QUANT, &quant, sizeof(quant));
  daala_encode_ctl(enc, OD_SET_COMPLEXITY, &complexity, sizeof(complexity));
  daala_encode_ctl(enc, OD_SET_ACTIVITY_MASKING, &masking, sizeof(masking));
  {
    /* never initialised by daala_encode_create; the reference CLI sets it to 1
       (examples/encoder_example.c:675,900) */
    int use_dering = 1;
    daala_encode_ctl(enc, OD_SET_DERING, &use_dering, sizeof(use_dering));
  }
  daala_comment_init(&dc);
  daala_info_init(&di2);
  daala_comment_init(&dc2);
  while (daala_encode_flush_header(enc, &dc, &dp) > 0) {
    rc = daala_decode_header_in(&di2, &dc2, &dsi, &dp);
    if (rc < 0) return -2;
  }
  dec = daala_decode_create(&di2, dsi);
  if (dec == NULL) return -3;
  memset(&img, 0, sizeof(img));
  img.nplanes = 3;
  img.width = w;
  img.height = h;
  for (pli = 0; pli < 3; pli++) {
    img.planes[pli].data = (unsigned char *)frame
     + (pli == 0 ? 0 : pli == 1 ? (size_t)w*h : (size_t)w*h + (size_t)cw*ch);
    img.planes[pli].xdec = img.planes[pli].ydec = pli > 0;
    img.planes[pli].xstride = 1;
    img.planes[pli].ystride = pli ? cw : w;
    img.planes[pli].bitdepth = 8;
  }
  if (daala_encode_img_in(enc, &img, 0, 0, &left) < 0) return -4;
  rc = -5;
  while (daala_encode_packet_out(enc, 0, &dp) > 0) {
    if (daala_decode_packet_in(dec, &dp) < 0) return -6;
    if (daala_decode_img_out(dec, &out) < 0) return -7;
    rc = 0;
  }
  if (rc) return rc;
  st = &((od_dec_ctx *)dec)->state;
  *fw_out = st->frame_width;
  *fh_out = st->frame_height;
  for (pli = 0; pli < 3; pli++) {
    int xd = pli > 0, pw = st->frame_width >> xd, ph = st->frame_height >> xd;
    memcpy(dd[pli], st->dtmp[pli], sizeof(od_coeff)*pw*ph);
    for (i = 0; i < ph/4; i++)
      memcpy(bs[pli] + (size_t)i*(st->frame_width/4), st->bskip[pli] + (size_t)i*st->skip_stride,
       pw/4);
    quantizer[pli] = st->quantizer[pli];
    for (i = 0; i < (pli ? ch : h); i++)
      memcpy(oo[pli] + (size_t)i*(pli ? cw : w),
       out.planes[pli].data + (size_t)i*out.planes[pli].ystride, pli ? cw : w);
  }
  for (i = 0; i < st->nvsb*4; i++)
    for (j = 0; j < st->nhsb*4; j++)
      bsize_out[i*st->nhsb*4 + j] = st->bsize[i*st->bstride + j];
  memcpy(dering_flags, st->dering_flags, (size_t)st->nhsb*st->nvsb);
  daala_decode_free(dec);
  daala_encode_free(enc);
  return 0;
}
