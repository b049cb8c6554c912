/* Probe translation unit (TEST INFRASTRUCTURE, dev container only).
 * #includes the reference's src/mcenc.c (through -I/root/reference/src, the technique of the
 * reference's own src/tests/test_coef_coder.c:25-34) to reach the static
 * od_mv_est_calc_sads (:3761).  probe_mvest_open() codes a short inter stream with the
 * reference library, then runs the reference's od_mv_est_calc_sads on the encoder's own
 * motion estimation context as the last frame left it: the final vector grid (1/8-sample
 * vectors, both references), the encoder's reference images and its input frame.  The
 * getters hand out everything that call read (grid, images) and wrote (sad_cache), which
 * tools/gen_golden.py stores as tests/golden/mvest_sads.npz. */
#include "mcenc.c"

static daala_enc_ctx *g_enc;

void probe_mvest_close(void) {
  if (g_enc != NULL) daala_encode_free(g_enc);
  g_enc = NULL;
}

/* frames: nframes dense 4:2:0 frames (Y, U, V at picture size).  Returns 0, or < 0. */
int probe_mvest_open(int w, int h, int nframes, int quant, int keyrate,
 const unsigned char *frames) {
  daala_info di;
  daala_enc_ctx *enc;
  int f;
  int use_dering;
  int cw;
  int ch;
  probe_mvest_close();
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
  if (enc == NULL) return -1;
  daala_encode_ctl(enc, OD_SET_QUANT, &quant, sizeof(quant));
  use_dering = 1;
  daala_encode_ctl(enc, OD_SET_DERING, &use_dering, sizeof(use_dering));
  cw = (w + 1) >> 1;
  ch = (h + 1) >> 1;
  for (f = 0; f < nframes; f++) {
    od_img img;
    daala_packet dp;
    const unsigned char *base;
    int pli;
    int left;
    base = frames + (size_t)f*(w*h + 2*cw*ch);
    memset(&img, 0, sizeof(img));
    img.nplanes = 3;
    img.width = w;
    img.height = h;
    for (pli = 0; pli < 3; pli++) {
      img.planes[pli].data = (unsigned char *)base;
      img.planes[pli].xdec = img.planes[pli].ydec = pli > 0;
      img.planes[pli].xstride = 1;
      img.planes[pli].ystride = pli > 0 ? cw : w;
      img.planes[pli].bitdepth = 8;
      base += pli > 0 ? cw*ch : w*h;
    }
    if (daala_encode_img_in(enc, &img, 0, 0, &left) < 0) {
      daala_encode_free(enc);
      return -2;
    }
    while (daala_encode_packet_out(enc, 0, &dp) > 0);
  }
  g_enc = enc;
  /* the reference's function, on the state the last frame left */
  od_mv_est_calc_sads(enc->mvest);
  return 0;
}

/* dims: nhmvbs, nvmvbs, level_min, level_max, number of images, frame_width, frame_height,
   then per plane: reference stride, reference rows, origin x, origin y. */
int probe_mvest_dims(int32_t *dims) {
  od_state *state;
  int pli;
  if (g_enc == NULL) return -1;
  state = &g_enc->state;
  dims[0] = state->nhmvbs;
  dims[1] = state->nvmvbs;
  dims[2] = g_enc->mvest->level_min;
  dims[3] = g_enc->mvest->level_max;
  dims[4] = OD_FRAME_MAX + 1;
  dims[5] = state->frame_width;
  dims[6] = state->frame_height;
  for (pli = 0; pli < 3; pli++) {
    od_img_plane *rp;
    rp = state->ref_imgs[0].planes + pli;
    dims[7 + 4*pli] = rp->ystride;
    dims[8 + 4*pli] = (state->frame_height + 2*OD_BUFFER_PADDING) >> rp->ydec;
    dims[9 + 4*pli] = OD_BUFFER_PADDING >> rp->xdec;
    dims[10 + 4*pli] = OD_BUFFER_PADDING >> rp->ydec;
  }
  return 0;
}

/* gmvx, gmvy, gref: (nvmvbs + 1) x (nhmvbs + 1), the vector a prediction reads and the INDEX
   of the image it points into (state->ref_imgi[ref], -1: none); refs[pli]: every image of
   plane pli with its padding, back to back; src[pli]: the encoder's input plane, frame size
   (dense); sad[l]: sad_cache[l] as (nvmvbs >> l) x (nhmvbs >> l) x 4. */
int probe_mvest_get(int32_t *gmvx, int32_t *gmvy, int32_t *gref, unsigned char *const refs[3],
 unsigned char *const src[3], int32_t *const sad[OD_LOG_MVB_DELTA0]) {
  od_state *state;
  int vx;
  int vy;
  int pli;
  int k;
  int l;
  if (g_enc == NULL) return -1;
  state = &g_enc->state;
  for (vy = 0; vy <= state->nvmvbs; vy++) {
    for (vx = 0; vx <= state->nhmvbs; vx++) {
      od_mv_grid_pt *g;
      int o;
      g = state->mv_grid[vy] + vx;
      o = vy*(state->nhmvbs + 1) + vx;
      if (g->ref == OD_FRAME_NEXT) {
        gmvx[o] = g->mv1[0];
        gmvy[o] = g->mv1[1];
      }
      else {
        gmvx[o] = g->mv[0];
        gmvy[o] = g->mv[1];
      }
      gref[o] = state->ref_imgi[g->ref];
    }
  }
  for (pli = 0; pli < 3; pli++) {
    od_img_plane *ip;
    int y;
    for (k = 0; k <= OD_FRAME_MAX; k++) {
      od_img_plane *rp;
      size_t bytes;
      rp = state->ref_imgs[k].planes + pli;
      bytes = (size_t)rp->ystride*((state->frame_height + 2*OD_BUFFER_PADDING) >> rp->ydec);
      memcpy(refs[pli] + bytes*k, rp->data - (ptrdiff_t)(OD_BUFFER_PADDING >> rp->ydec)*rp->ystride
       - (OD_BUFFER_PADDING >> rp->xdec), bytes);
    }
    ip = g_enc->input_img[g_enc->curr_frame].planes + pli;
    for (y = 0; y < state->frame_height >> ip->ydec; y++) {
      memcpy(src[pli] + (size_t)y*(state->frame_width >> ip->xdec), ip->data + (size_t)y*ip->ystride,
       state->frame_width >> ip->xdec);
    }
  }
  for (l = 0; l < OD_LOG_MVB_DELTA0; l++) {
    int nh;
    int nv;
    nh = state->nhmvbs >> l;
    nv = state->nvmvbs >> l;
    if (sad[l] == NULL) continue;
    for (vy = 0; vy < nv; vy++) {
      for (vx = 0; vx < nh; vx++) {
        for (k = 0; k < 4; k++) sad[l][(vy*nh + vx)*4 + k] = g_enc->mvest->sad_cache[l][vy][vx][k];
      }
    }
  }
  return 0;
}

/* od_mv_est_bma_sad (static, src/mcenc.c:2228) on the same state: the SAD of the single-vector
   prediction of the block at luma (bx, by) - the caller passes the corner, as od_mv_est_init_mv
   does after centring the block on its vertex - with the half-sample vector (mvx, mvy) from
   reference frame type `ref` (OD_FRAME_GOLD = 0, OD_FRAME_PREV = 1).  *img receives the image index
   that frame type maps to (state->ref_imgi[ref]). */
int32_t probe_mvest_bma_sad(int ref, int bx, int by, int mvx, int mvy, int log_mvb_sz, int *img) {
  if (g_enc == NULL) return -1;
  if (img != NULL) *img = g_enc->state.ref_imgi[ref];
  if (g_enc->state.ref_imgi[ref] < 0) return -1;
  return od_mv_est_bma_sad(g_enc->mvest, ref, bx, by, mvx, mvy, log_mvb_sz);
}
