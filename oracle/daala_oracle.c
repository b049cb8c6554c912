/* daala_oracle.c - CPU restatement of the Daala transform + PVQ hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (daala_amd/, include/) may
 * include, link or call this file; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, as the checker for the HIP path.
 *
 * Parity status: PINNED.  Every function below is checked bit-exactly against
 * the real reference compiled from /root/reference (oracle/_ref, built by
 * oracle/Makefile) in tests/test_oracle_vs_ref.py, against golden vectors
 * generated from that build (tests/golden/, tools/gen_golden.py) and - for the
 * transforms - against the reference's own self-test `dcttest`
 * (src/dct.c:2192-3951; stdout md5 eaace07761b6fd646b83285dfbdbc5c2).
 *
 * Exception (documented in DESIGN.md): orc_gain_compand with beta != 1 calls
 * the host libm pow(), exactly like the reference does; its value is therefore
 * "the reference on this libm", not an independent restatement of pow.
 *
 * Plain C99, no FMA contraction (-ffp-contract=off in oracle/Makefile) so double
 * arithmetic evaluates left to right like the gcc -O2 reference build.
 * od_coeff == int32_t (reference src/filter.h:31).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "gen_lift_oracle.h"
#include "coding_order_tables.h"
#include "filter_params.h"

typedef int32_t coeff;

/* ------------------------------------------------------------------------ */
/* A6/A7: 1-D and separable 2-D reversible DCT (reference src/dct.c:137-149,
   :335-347, :774-788, :2028-2042).  Forward: N column transforms into the rows
   of a scratch, then N column transforms of the scratch into rows of y.
   Inverse: rows of y into columns of the scratch, rows of scratch into columns
   of x.  The order is observable because of rounding. */
typedef void (*fdct1d_fn)(int32_t *y, const int32_t *x, int xs);
typedef void (*idct1d_fn)(int32_t *x, int xs, const int32_t *y);

static fdct1d_fn fdct_1d_of(int n) {
  return n == 4 ? orc_fdct4_1d : n == 8 ? orc_fdct8_1d
       : n == 16 ? orc_fdct16_1d : orc_fdct32_1d;
}
static idct1d_fn idct_1d_of(int n) {
  return n == 4 ? orc_idct4_1d : n == 8 ? orc_idct8_1d
       : n == 16 ? orc_idct16_1d : orc_idct32_1d;
}

void orc_fdct_1d(int n, coeff *y, const coeff *x, int xstride) {
  fdct_1d_of(n)(y, x, xstride);
}
void orc_idct_1d(int n, coeff *x, int xstride, const coeff *y) {
  idct_1d_of(n)(x, xstride, y);
}

void orc_fdct_2d(int n, coeff *y, int ystride, const coeff *x, int xstride) {
  coeff z[32*32];
  fdct1d_fn f = fdct_1d_of(n);
  int i;
  for (i = 0; i < n; i++) f(z + n*i, x + i, xstride);
  for (i = 0; i < n; i++) f(y + ystride*i, z + i, n);
}

void orc_idct_2d(int n, coeff *x, int xstride, const coeff *y, int ystride) {
  coeff z[32*32];
  idct1d_fn f = idct_1d_of(n);
  int i;
  for (i = 0; i < n; i++) f(z + i, n, y + ystride*i);
  for (i = 0; i < n; i++) f(x + i, xstride, z + n*i);
}

/* ------------------------------------------------------------------------ */
/* A8: 2x2 Haar kernel (reference src/tf.h:34-45) and the multi-level 2-D Haar
   used for lossless frames (src/dct.c:1960-2026). */
static void haar_kernel(coeff *ll, coeff *lh, coeff *hl, coeff *hh) {
  coeff a = *ll, b = *lh, c = *hl, d = *hh, m;
  a += c;
  d -= b;
  m = (a - d) >> 1;
  b = m - b;
  c = m - c;
  a -= b;
  d += c;
  *ll = a; *lh = b; *hl = c; *hh = d;
}

void orc_haar(coeff *y, int ystride, const coeff *x, int xstride, int ln) {
  coeff tmp[32*32];
  int n = 1 << ln, i, j, level;
  for (i = 0; i < n; i++)
    for (j = 0; j < n; j++) tmp[i*n + j] = x[i*xstride + j];
  for (level = 0; level < ln; level++) {
    int np = n >> level >> 1;
    for (i = 0; i < np; i++) {
      for (j = 0; j < np; j++) {
        coeff a = tmp[2*i*n + 2*j], b = tmp[(2*i + 1)*n + 2*j];
        coeff c = tmp[2*i*n + 2*j + 1], d = tmp[(2*i + 1)*n + 2*j + 1];
        haar_kernel(&a, &b, &c, &d);
        tmp[i*n + j] = a;
        y[i*ystride + j + np] = b;
        y[(i + np)*ystride + j] = c;
        y[(i + np)*ystride + j + np] = d;
      }
    }
  }
  y[0] = tmp[0];
}

void orc_haar_inv(coeff *x, int xstride, const coeff *y, int ystride, int ln) {
  int i, j, level;
  x[0] = y[0];
  for (level = ln - 1; level >= 0; level--) {
    int np = 1 << (ln - 1 - level);
    for (i = np - 1; i >= 0; i--) {
      for (j = np - 1; j >= 0; j--) {
        coeff a = x[i*xstride + j], b = y[i*ystride + j + np];
        coeff c = y[(i + np)*ystride + j], d = y[(i + np)*ystride + j + np];
        haar_kernel(&a, &b, &c, &d);
        x[2*i*xstride + 2*j] = a;
        x[(2*i + 1)*xstride + 2*j] = b;
        x[2*i*xstride + 2*j + 1] = c;
        x[(2*i + 1)*xstride + 2*j + 1] = d;
      }
    }
  }
}

/* ------------------------------------------------------------------------ */
/* A3: 4-point lapping pre/post filter (reference src/filter.c:174-249, params
   85,75,-15,33 at :164-167).  In/out may alias.  The forward scale uses a floor
   shift plus "+1 if positive"; the inverse undoes it with C truncating
   division. */
void orc_pre_filter4(coeff *y, const coeff *x) {
  int d3 = x[0] - x[3];
  int d2 = x[1] - x[2];
  int s1 = x[1] - (d2 >> 1);
  int s0 = x[0] - (d3 >> 1);
  d2 = d2*85 >> 6;
  if (d2 > 0) d2++;
  d3 = d3*75 >> 6;
  if (d3 > 0) d3++;
  d3 += (d2*-15 + 32) >> 6;
  d2 += (d3*33 + 32) >> 6;
  s0 += d3 >> 1;
  s1 += d2 >> 1;
  y[0] = s0;
  y[1] = s1;
  y[2] = s1 - d2;
  y[3] = s0 - d3;
}

void orc_post_filter4(coeff *x, const coeff *y) {
  int d3 = y[0] - y[3];
  int d2 = y[1] - y[2];
  int s1 = y[1] - (d2 >> 1);
  int s0 = y[0] - (d3 >> 1);
  d2 -= (d3*33 + 32) >> 6;
  d3 -= (d2*-15 + 32) >> 6;
  d3 = d3*64/75;
  d2 = d2*64/85;
  s0 += d3 >> 1;
  s1 += d2 >> 1;
  x[0] = s0;
  x[1] = s1;
  x[2] = s1 - d2;
  x[3] = s0 - d3;
}

static void filt4_col(coeff *c, int stride, int inv) {
  coeff t[4];
  int k;
  for (k = 0; k < 4; k++) t[k] = c[k*stride];
  if (inv) orc_post_filter4(t, t); else orc_pre_filter4(t, t);
  for (k = 0; k < 4; k++) c[k*stride] = t[k];
}

/* A4: lapping across every superblock boundary of a plane (reference
   src/filter.c:1556-1586 pre, :1588-1646 post).  Pre: all horizontal boundaries
   (vertical taps) first, then all vertical boundaries; post: the reverse.
   sb = superblock size in this plane (32 >> dec). */
void orc_prefilter_frame_sbs(coeff *c, int stride, int nhsb, int nvsb, int dec) {
  int sb = 32 >> dec, w = nhsb*sb, h = nvsb*sb, s, i;
  for (s = 1; s < nvsb; s++)
    for (i = 0; i < w; i++) filt4_col(c + (s*sb - 2)*stride + i, stride, 0);
  for (s = 1; s < nhsb; s++)
    for (i = 0; i < h; i++) {
      coeff *p = c + i*stride + s*sb - 2;
      orc_pre_filter4(p, p);
    }
}

void orc_postfilter_frame_sbs(coeff *c, int stride, int nhsb, int nvsb, int dec) {
  int sb = 32 >> dec, w = nhsb*sb, h = nvsb*sb, s, i;
  for (s = 1; s < nhsb; s++)
    for (i = 0; i < h; i++) {
      coeff *p = c + i*stride + s*sb - 2;
      orc_post_filter4(p, p);
    }
  for (s = 1; s < nvsb; s++)
    for (i = 0; i < w; i++) filt4_col(c + (s*sb - 2)*stride + i, stride, 1);
}

/* A5: lapping on the internal cross of an n x n block about to be split
   (reference src/filter.c:1486-1510 pre, :1512-1554 post; always the 4-point
   filter since OD_FILT_SIZE()==0, src/filter.h:99).  `hfilter` gates the taps
   across the horizontal centre line, `vfilter` those across the vertical one. */
void orc_prefilter_split(coeff *c, int stride, int n, int hfilter, int vfilter) {
  int i;
  if (hfilter)
    for (i = 0; i < n; i++) filt4_col(c + (n/2 - 2)*stride + i, stride, 0);
  if (vfilter)
    for (i = 0; i < n; i++) {
      coeff *p = c + i*stride + n/2 - 2;
      orc_pre_filter4(p, p);
    }
}

void orc_postfilter_split(coeff *c, int stride, int n, int hfilter, int vfilter) {
  int i;
  if (vfilter)
    for (i = 0; i < n; i++) {
      coeff *p = c + i*stride + n/2 - 2;
      orc_post_filter4(p, p);
    }
  if (hfilter)
    for (i = 0; i < n; i++) filt4_col(c + (n/2 - 2)*stride + i, stride, 1);
}

/* ------------------------------------------------------------------------ */
/* A1/A2: pixel <-> coefficient domain (reference src/state.c:1209-1232 and
   :1274-1300, 8-bit references: xstride 1). */
void orc_ref_buf_to_coeff(coeff *dst, int dstride, const uint8_t *src,
 int sstride, int w, int h, int coeff_shift) {
  int x, y;
  for (y = 0; y < h; y++)
    for (x = 0; x < w; x++)
      dst[y*dstride + x] = (src[y*sstride + x] - 128)*(1 << coeff_shift);
}

void orc_coeff_to_ref_buf(uint8_t *dst, int dstride, const coeff *src,
 int sstride, int w, int h, int coeff_shift) {
  int x, y;
  for (y = 0; y < h; y++)
    for (x = 0; x < w; x++) {
      int v = ((src[y*sstride + x] + (1 << coeff_shift >> 1)) >> coeff_shift) + 128;
      dst[y*dstride + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

/* ------------------------------------------------------------------------ */
/* A11: raster <-> coding (band) order (reference src/partition.c:144-194).
   dst has n*n entries; for n==32 only entries 0..511 are written/read. */
static const uint16_t *coding_table(int n, int *ncoded) {
  switch (n) {
    case 4: *ncoded = CODING_NCODED_4; return CODING_TO_RASTER_4;
    case 8: *ncoded = CODING_NCODED_8; return CODING_TO_RASTER_8;
    case 16: *ncoded = CODING_NCODED_16; return CODING_TO_RASTER_16;
    default: *ncoded = CODING_NCODED_32; return CODING_TO_RASTER_32;
  }
}

void orc_raster_to_coding_order(coeff *dst, int n, const coeff *src, int stride) {
  int nc, i;
  const uint16_t *t = coding_table(n, &nc);
  for (i = 0; i < nc; i++) dst[i] = src[(t[i]/n)*stride + t[i]%n];
}

void orc_coding_order_to_raster(coeff *dst, int stride, const coeff *src, int n) {
  int nc, i;
  const uint16_t *t = coding_table(n, &nc);
  for (i = 0; i < nc; i++) dst[(t[i]/n)*stride + t[i]%n] = src[i];
}

/* Band boundaries in coding order (reference src/partition.c:77-91): returns the
   number of bands of an n x n block and fills off[0..nb]. */
int orc_band_offsets(int n, int *off) {
  static const int all[] = {1, 16, 24, 32, 64, 96, 128, 256, 384, 512};
  int nb = n == 4 ? 1 : n == 8 ? 4 : n == 16 ? 7 : 9, i;
  for (i = 0; i <= nb; i++) off[i] = all[i];
  return nb;
}

/* ------------------------------------------------------------------------ */
/* A9: TF resampling (reference src/tf.c:38-108, :112-170) and the CfL luma
   resample built on it (src/intra.c:72-109). */
void orc_tf_up_h_lp(coeff *dst, int dstride, const coeff *src, int sstride,
 int dx, int n) {
  int x, y;
  for (y = 0; y < n; y++)
    for (x = 0; x < n >> 1; x++) {
      coeff ll = src[y*sstride + x], lh = src[y*sstride + x + dx];
      int sw = x & 1;
      lh = ll - lh;
      ll -= LIFT_HALF(lh);
      dst[y*dstride + 2*x + sw] = ll;
      dst[y*dstride + 2*x + 1 - sw] = lh;
    }
}

void orc_tf_up_v_lp(coeff *dst, int dstride, const coeff *src, int sstride,
 int dy, int n) {
  int x, y;
  for (y = 0; y < n >> 1; y++) {
    int sw = y & 1;
    for (x = 0; x < n; x++) {
      coeff ll = src[y*sstride + x], hl = src[(y + dy)*sstride + x];
      hl = ll - hl;
      ll -= LIFT_HALF(hl);
      dst[(2*y + sw)*dstride + x] = ll;
      dst[(2*y + 1 - sw)*dstride + x] = hl;
    }
  }
}

/* Shared body of od_tf_up_hv_lp (count = n/2, offsets dx,dy) and od_tf_up_hv
   (count = n, offsets n,n): note the lh/hl swap into the Haar kernel. */
static void tf_up_hv_core(coeff *dst, int dstride, const coeff *src, int sstride,
 int dx, int dy, int cnt) {
  int x, y;
  for (y = 0; y < cnt; y++) {
    int vs = y & 1;
    for (x = 0; x < cnt; x++) {
      coeff ll = src[y*sstride + x], lh = src[y*sstride + x + dx];
      coeff hl = src[(y + dy)*sstride + x], hh = src[(y + dy)*sstride + x + dx];
      int hs = x & 1;
      haar_kernel(&ll, &hl, &lh, &hh);
      dst[(2*y + vs)*dstride + 2*x + hs] = ll;
      dst[(2*y + vs)*dstride + 2*x + 1 - hs] = lh;
      dst[(2*y + 1 - vs)*dstride + 2*x + hs] = hl;
      dst[(2*y + 1 - vs)*dstride + 2*x + 1 - hs] = hh;
    }
  }
}

void orc_tf_up_hv_lp(coeff *dst, int dstride, const coeff *src, int sstride,
 int dx, int dy, int n) {
  tf_up_hv_core(dst, dstride, src, sstride, dx, dy, n >> 1);
}

void orc_tf_up_hv(coeff *dst, int dstride, const coeff *src, int sstride, int n) {
  tf_up_hv_core(dst, dstride, src, sstride, n, n, n);
}

void orc_tf_down_hv(coeff *dst, int dstride, const coeff *src, int sstride, int n) {
  int x, y;
  n >>= 1;
  for (y = 0; y < n; y++) {
    int vs = y & 1;
    for (x = 0; x < n; x++) {
      int hs = x & 1;
      coeff ll = src[(2*y + vs)*sstride + 2*x + hs];
      coeff lh = src[(2*y + vs)*sstride + 2*x + 1 - hs];
      coeff hl = src[(2*y + 1 - vs)*sstride + 2*x + hs];
      coeff hh = src[(2*y + 1 - vs)*sstride + 2*x + 1 - hs];
      haar_kernel(&ll, &lh, &hl, &hh);
      dst[y*dstride + x] = ll;
      dst[y*dstride + x + n] = lh;
      dst[(y + n)*dstride + x] = hl;
      dst[(y + n)*dstride + x + n] = hh;
    }
  }
}

/* CfL scaling of the 4x4 predictor, indexed [column][row] by the reference
   (src/intra.c:65-70, :85-86); the table is symmetric so it reads the same. */
static const int16_t CFL_SCALE4[4][4] = {
  {128, 128, 100, 36}, {128, 80, 71, 35}, {100, 71, 35, 31}, {36, 35, 31, 18}};

void orc_resample_luma_coeffs(coeff *pred, int pstride, const coeff *luma,
 int lstride, int xdec, int ydec, int bs, int chroma_bs) {
  int n = 4 << bs, x, y;
  if (chroma_bs == 0 && (xdec || ydec)) {
    if (xdec && ydec) {
      orc_tf_up_hv_lp(pred, pstride, luma, lstride, n, n, n);
      for (y = 0; y < 4; y++)
        for (x = 0; x < 4; x++)
          pred[y*pstride + x] = (CFL_SCALE4[x][y]*pred[y*pstride + x] + 64) >> 7;
    }
    else if (xdec) orc_tf_up_h_lp(pred, pstride, luma, lstride, n, n);
    else orc_tf_up_v_lp(pred, pstride, luma, lstride, n, n);
  }
  else {
    for (y = 0; y < n; y++)
      for (x = 0; x < n; x++) pred[y*pstride + x] = luma[y*lstride + x];
  }
}

/* ------------------------------------------------------------------------ */
/* A10: keyframe luma H/V predictor (reference src/intra.c:37-61).  bsize is the
   per-8x8... per-4x4-addressed block-size map accessor of the reference
   (OD_BLOCK_SIZE4x4: bsize[(by>>1)*bstride + (bx>>1)], src/block_size.h). */
void orc_hv_intra_pred(coeff *pred, const coeff *d, int w, int bx, int by,
 const unsigned char *bsize, int bstride, int bs) {
  int n = 4 << bs, i;
  int top = by > 0 && bsize[((by - 1) >> 1)*bstride + (bx >> 1)] == bs;
  int left = bx > 0 && bsize[(by >> 1)*bstride + ((bx - 1) >> 1)] == bs;
  const coeff *t = d + (by << 2)*w + (bx << 2);
  double g1 = 0, g2 = 0;
  if (top) for (i = 1; i < 4; i++) g1 += t[-n*w + i]*(double)t[-n*w + i];
  if (left) for (i = 1; i < 4; i++) g2 += t[-n + i*w]*(double)t[-n + i*w];
  if (top) for (i = 4; i < n; i++) pred[i] = t[-n*w + i];
  if (left) for (i = 4; i < n; i++) pred[i*n] = t[-n + i*w];
  if (g1 > g2) {
    if (top) for (i = 1; i < 4; i++) pred[i] = t[-n*w + i];
  }
  else if (left) for (i = 1; i < 4; i++) pred[i*n] = t[-n + i*w];
}

/* ------------------------------------------------------------------------ */
/* Forward path of one plane with KNOWN block sizes: pixel->coeff (A1), frame
   lapping (A4), then per superblock the recursion of od_compute_dcts
   (reference src/encode.c:1286-1343): split lapping (A5) down to the coded
   block size, fDCT (A6), and on keyframes the Haar merge of the four child DCs
   at every split.  bsize: luma block-size map, 1 byte per 8x8 luma area,
   values 0..3 (src/state.h:207-224), addressed without border here.
   c: w x h work plane (ends up lapped), d: w x h coefficient plane. */
static void compute_dcts_rec(coeff *c, coeff *d, int w, const unsigned char *bsize,
 int bstride, int bx, int by, int bsi, int dec, int pic_w, int pic_h,
 int keyframe) {
  int obs = bsize[((by << bsi) >> 1)*bstride + ((bx << bsi) >> 1)];
  int bs = obs > dec ? obs : dec;
  if (bs == bsi) {
    int n, bo;
    bs -= dec;
    n = 4 << bs;
    bo = (by*n)*w + bx*n;
    orc_fdct_2d(n, d + bo, w, c + bo, w);
  }
  else {
    int n, bo, hf, vf, ln;
    coeff x0, x1, x2, x3;
    bs = bsi - dec;
    n = 4 << bs;
    bo = (by*n)*w + bx*n;
    /* Quirk kept literally: plane-sample extents against the LUMA picture size,
       and the x-extent test gates the horizontal-centre-line filter
       (reference src/encode.c:1318-1320). */
    hf = (bx + 1)*n <= pic_w;
    vf = (by + 1)*n <= pic_h;
    orc_prefilter_split(c + bo, w, n, hf, vf);
    bsi--;
    bx <<= 1;
    by <<= 1;
    compute_dcts_rec(c, d, w, bsize, bstride, bx, by, bsi, dec, pic_w, pic_h, keyframe);
    compute_dcts_rec(c, d, w, bsize, bstride, bx + 1, by, bsi, dec, pic_w, pic_h, keyframe);
    compute_dcts_rec(c, d, w, bsize, bstride, bx, by + 1, bsi, dec, pic_w, pic_h, keyframe);
    compute_dcts_rec(c, d, w, bsize, bstride, bx + 1, by + 1, bsi, dec, pic_w, pic_h, keyframe);
    if (keyframe) {
      ln = bsi - dec + 2;
      x0 = d[(by << ln)*w + (bx << ln)];
      x1 = d[(by << ln)*w + ((bx + 1) << ln)];
      x2 = d[((by + 1) << ln)*w + (bx << ln)];
      x3 = d[((by + 1) << ln)*w + ((bx + 1) << ln)];
      haar_kernel(&x0, &x2, &x1, &x3);
      d[(by << ln)*w + (bx << ln)] = x0;
      d[(by << ln)*w + ((bx + 1) << ln)] = x1;
      d[((by + 1) << ln)*w + (bx << ln)] = x2;
      d[((by + 1) << ln)*w + ((bx + 1) << ln)] = x3;
    }
  }
}

void orc_forward_plane(coeff *c, coeff *d, const uint8_t *pix, int pstride,
 int nhsb, int nvsb, int dec, const unsigned char *bsize, int bstride,
 int pic_w, int pic_h, int keyframe) {
  int w = (nhsb*32) >> dec, h = (nvsb*32) >> dec, sbx, sby;
  orc_ref_buf_to_coeff(c, w, pix, pstride, w, h, 4);
  orc_prefilter_frame_sbs(c, w, nhsb, nvsb, dec);
  for (sby = 0; sby < nvsb; sby++)
    for (sbx = 0; sbx < nhsb; sbx++)
      compute_dcts_rec(c, d, w, bsize, bstride, sbx, sby, 3, dec, pic_w, pic_h,
       keyframe);
}

/* Forward PYRAMID of one plane for block-size RDO: level L (0..3 <-> block size
   4<<L in this plane... see DESIGN.md) holds the fDCT of every block of that
   size, computed from the frame-lapped plane plus the split lapping of all its
   ancestors - exactly the input the reference's RDO recursion hands to
   od_block_encode at that level (src/encode.c:1554-1592; c is restored from
   c_orig between the no-split trial and the split).  lev[k] are w x h planes,
   k = 0 for the superblock-sized transform, k = 1 its four children, ...
   nlev = 4 for luma (32,16,8,4), 3 for 4:2:0 chroma (16,8,4). */
static void pyramid_rec(coeff *c, coeff **lev, int w, int x0, int y0, int n,
 int k, int nlev, int bx, int by, int pic_w, int pic_h) {
  int bo = y0*w + x0;
  orc_fdct_2d(n, lev[k] + bo, w, c + bo, w);
  if (k + 1 < nlev) {
    int hf = (bx + 1)*n <= pic_w, vf = (by + 1)*n <= pic_h, h = n/2;
    orc_prefilter_split(c + bo, w, n, hf, vf);
    pyramid_rec(c, lev, w, x0, y0, h, k + 1, nlev, 2*bx, 2*by, pic_w, pic_h);
    pyramid_rec(c, lev, w, x0 + h, y0, h, k + 1, nlev, 2*bx + 1, 2*by, pic_w, pic_h);
    pyramid_rec(c, lev, w, x0, y0 + h, h, k + 1, nlev, 2*bx, 2*by + 1, pic_w, pic_h);
    pyramid_rec(c, lev, w, x0 + h, y0 + h, h, k + 1, nlev, 2*bx + 1, 2*by + 1, pic_w, pic_h);
  }
}

void orc_forward_pyramid_plane(coeff *c, coeff **lev, int nlev,
 const uint8_t *pix, int pstride, int nhsb, int nvsb, int dec, int pic_w,
 int pic_h) {
  int sb = 32 >> dec, w = nhsb*sb, h = nvsb*sb, sbx, sby;
  orc_ref_buf_to_coeff(c, w, pix, pstride, w, h, 4);
  orc_prefilter_frame_sbs(c, w, nhsb, nvsb, dec);
  for (sby = 0; sby < nvsb; sby++)
    for (sbx = 0; sbx < nhsb; sbx++)
      pyramid_rec(c, lev, w, sbx*sb, sby*sb, sb, 0, nlev, sbx, sby, pic_w, pic_h);
}

/* Inverse path of one plane with known block sizes (reference decoder order,
   src/decode.c:767-868 + :1037 + :1154): per superblock, iDCT (A7) of every
   coded block into c then the split post-filters on the way back up; after all
   superblocks the frame post-filter (A4) and the clamp to 8 bit (A2).
   Does NOT undo the keyframe Haar-of-DCs: that belongs to the DC decoding
   (od_decode_haar_dc_*), i.e. d must hold final DCT coefficients. */
static void inverse_rec(coeff *c, const coeff *d, int w, const unsigned char *bsize,
 int bstride, int bx, int by, int bsi, int dec, int pic_w, int pic_h) {
  int obs = bsize[((by << bsi) >> 1)*bstride + ((bx << bsi) >> 1)];
  int bs = obs > dec ? obs : dec;
  if (bs == bsi) {
    int n = 4 << (bs - dec), bo = (by*n)*w + bx*n;
    orc_idct_2d(n, c + bo, w, d + bo, w);
  }
  else {
    int n = 4 << (bsi - dec), bo = (by*n)*w + bx*n;
    int hf = (bx + 1)*n <= pic_w, vf = (by + 1)*n <= pic_h;
    inverse_rec(c, d, w, bsize, bstride, 2*bx, 2*by, bsi - 1, dec, pic_w, pic_h);
    inverse_rec(c, d, w, bsize, bstride, 2*bx + 1, 2*by, bsi - 1, dec, pic_w, pic_h);
    inverse_rec(c, d, w, bsize, bstride, 2*bx, 2*by + 1, bsi - 1, dec, pic_w, pic_h);
    inverse_rec(c, d, w, bsize, bstride, 2*bx + 1, 2*by + 1, bsi - 1, dec, pic_w, pic_h);
    orc_postfilter_split(c + bo, w, n, hf, vf);
  }
}

void orc_inverse_plane(uint8_t *pix, int pstride, coeff *c, const coeff *d,
 int nhsb, int nvsb, int dec, const unsigned char *bsize, int bstride,
 int pic_w, int pic_h) {
  int w = (nhsb*32) >> dec, h = (nvsb*32) >> dec, sbx, sby;
  for (sby = 0; sby < nvsb; sby++)
    for (sbx = 0; sbx < nhsb; sbx++)
      inverse_rec(c, d, w, bsize, bstride, sbx, sby, 3, dec, pic_w, pic_h);
  orc_postfilter_frame_sbs(c, w, nhsb, nvsb, dec);
  orc_coeff_to_ref_buf(pix, pstride, c, w, w, h, 4);
}

/* ------------------------------------------------------------------------ */
/* PVQ math (all double, sequential evaluation order kept). */
#define QM_SCALE_1 (1./32767)          /* OD_QM_SCALE_1, reference src/pvq.h:57-59 */
#define QM_INV_SCALE_1 (1./4096)       /* OD_QM_INV_SCALE_1 */
#define COMPAND_SCALE 4096.            /* OD_COMPAND_SCALE = 256 << 4, src/pvq.h:68 */
#define PVQ_LAMBDA .147                /* OD_PVQ_LAMBDA, src/pvq.h:49 */

/* A13: reference src/pvq.c:422-425 and :456-468. */
double orc_gain_compand(double g, int q0, double beta) {
  if (beta == 1) return g/q0;
  return COMPAND_SCALE*pow(g*(1./COMPAND_SCALE), 1./beta)/q0;
}

double orc_pvq_compute_gain(const coeff *x, int n, int q0, double *g, double beta,
 const int16_t *qm) {
  double acc = 0;
  int i;
  for (i = 0; i < n; i++)
    acc += x[i]*(double)x[i]*qm[i]*QM_SCALE_1*qm[i]*QM_SCALE_1;
  *g = sqrt(acc);
  return orc_gain_compand(*g, q0, beta);
}

/* A17 helper: reference src/pvq.c:434-443. */
double orc_gain_expand(double cg, int q0, double beta) {
  if (beta == 1) return cg*q0;
  if (beta == 1.5) {
    cg *= q0*(1./COMPAND_SCALE);
    return COMPAND_SCALE*cg*sqrt(cg);
  }
  return COMPAND_SCALE*pow(cg*q0*(1./COMPAND_SCALE), beta);
}

/* A18: reference src/pvq.c:476-535. */
int orc_pvq_compute_max_theta(double qcg, double beta) {
  int ts = (int)floor(.5 + qcg*M_PI/(2*beta));
  if (qcg < 1.4) ts = 1;
  return ts;
}

double orc_pvq_compute_theta(int t, int max_theta) {
  if (max_theta != 0) return (t < max_theta - 1 ? t : max_theta - 1)*.5*M_PI/max_theta;
  return 0;
}

int orc_pvq_compute_k(double qcg, int itheta, double theta, int noref, int n,
 double beta, int nodesync) {
  int k;
  if (noref) {
    if (qcg == 0) return 0;
    if (n == 15 && qcg == 1 && beta > 1.25) return 1;
    k = (int)floor(.5 + (qcg - .2)*sqrt((n + 3)/2)/beta);   /* integer (n+3)/2 */
    return k > 1 ? k : 1;
  }
  if (itheta == 0) return 0;
  if (nodesync) k = (int)floor(.5 + (itheta - .2)*sqrt((n + 2)/2));
  else k = (int)floor(.5 + (qcg*sin(theta) - .2)*sqrt((n + 2)/2)/beta);
  return k > 1 ? k : 1;
}

/* A14: reference src/pvq.c:364-413. */
int orc_compute_householder(double *r, int n, double gr, int *sign) {
  int m = 0, i, s;
  double maxr = 0;
  for (i = 0; i < n; i++)
    if (fabs(r[i]) > maxr) { maxr = fabs(r[i]); m = i; }
  s = r[m] > 0 ? 1 : -1;
  r[m] += gr*s;
  *sign = s;
  return m;
}

void orc_apply_householder(double *x, const double *r, int n) {
  double l2r = 0, proj = 0, proj_1;
  int i;
  for (i = 0; i < n; i++) l2r += r[i]*r[i];
  for (i = 0; i < n; i++) proj += r[i]*x[i];
  proj_1 = proj*2./(1e-100 + l2r);
  for (i = 0; i < n; i++) x[i] -= r[i]*proj_1;
}

/* A15: PVQ codeword search (reference src/pvq_encoder.c:83-225).  The first 16
   reciprocal square roots are the reference's 6-digit literals (:84-88). */
static double rsqrt_small(int i) {
  static const double t[16] = {
    1.000000, 0.707107, 0.577350, 0.500000, 0.447214, 0.408248, 0.377964,
    0.353553, 0.333333, 0.316228, 0.301511, 0.288675, 0.277350, 0.267261,
    0.258199, 0.250000};
  return i <= 16 ? t[i - 1] : 1./sqrt(i);
}

double orc_pvq_search_rdo_double(const double *xcoeff, int n, int k, coeff *yp,
 double g2) {
  double x[1024], xx = 0, xy = 0, yy = 0, norm_1, lambda, delta_rate;
  int i = 0, j, rdo_pulses;
  for (j = 0; j < n; j++) {
    x[j] = fabs(xcoeff[j]);
    xx += x[j]*x[j];
  }
  norm_1 = 1./sqrt(1e-30 + xx);
  lambda = PVQ_LAMBDA/(1e-30 + g2);
  if (k > 2) {
    double l1 = 0, l1_inv;
    for (j = 0; j < n; j++) l1 += x[j];
    l1_inv = 1./(l1 > 1e-100 ? l1 : 1e-100);
    for (j = 0; j < n; j++) {
      int p = (int)floor(k*x[j]*l1_inv);
      yp[j] = p > 0 ? p : 0;
      xy += x[j]*yp[j];
      yy += yp[j]*yp[j];
      i += yp[j];
    }
  }
  else for (j = 0; j < n; j++) yp[j] = 0;
  rdo_pulses = 1 + k/4;
  delta_rate = 3./n;
  for (; i < k - rdo_pulses; i++) {
    int pos = 0;
    double best_xy = -10, best_yy = 1;
    for (j = 0; j < n; j++) {
      double txy = xy + x[j], tyy = yy + 2*yp[j] + 1;
      txy *= txy;
      if (j == 0 || txy*best_yy > best_xy*tyy) {
        best_xy = txy;
        best_yy = tyy;
        pos = j;
      }
    }
    xy = xy + x[pos];
    yy = yy + 2*yp[pos] + 1;
    yp[pos]++;
  }
  for (; i < k; i++) {
    int pos = 0;
    double best_cost = -1e5;
    for (j = 0; j < n; j++) {
      double txy = xy + x[j];
      double rs = rsqrt_small((int)(yy + 2*yp[j] + 1));
      txy = 2*txy*norm_1*rs - lambda*j*delta_rate;
      if (j == 0 || txy > best_cost) {
        best_cost = txy;
        pos = j;
      }
    }
    xy = xy + x[pos];
    yy = yy + 2*yp[pos] + 1;
    yp[pos]++;
  }
  for (j = 0; j < n; j++) if (xcoeff[j] < 0) yp[j] = -yp[j];
  return xy/(1e-100 + sqrt(xx*yy));
}

/* A17: reference src/pvq.c:552-585. */
void orc_pvq_synthesis_partial(coeff *xcoeff, const coeff *yp, const double *r,
 int n, int noref, double g, double theta, int m, int s, const int16_t *qm_inv) {
  int i, yy = 0, nn = n - (!noref);
  double scale;
  for (i = 0; i < nn; i++) yy += yp[i]*(int32_t)yp[i];
  scale = yy == 0 ? 0 : g/sqrt(yy);
  if (noref) {
    for (i = 0; i < n; i++)
      xcoeff[i] = (coeff)floor(.5 + (yp[i]*scale)*(qm_inv[i]*QM_INV_SCALE_1));
  }
  else {
    double x[1024];
    scale *= sin(theta);
    for (i = 0; i < m; i++) x[i] = yp[i]*scale;
    x[m] = -s*g*cos(theta);
    for (i = m; i < nn; i++) x[i + 1] = yp[i]*scale;
    orc_apply_householder(x, r, n);
    for (i = 0; i < n; i++)
      xcoeff[i] = (coeff)floor(.5 + (x[i]*(qm_inv[i]*QM_INV_SCALE_1)));
  }
}

/* A16 (state-free part): the no-reference candidate loop of pvq_theta
   (reference src/pvq_encoder.c:452-481) WITHOUT the rate term: for every gain
   candidate i in [max(1,floor(cg)), ceil(cg)] the pulse count, codeword, cosine
   distance and distortion.  The host adds lambda*od_pvq_rate() (adaptive
   entropy state) and takes the argmin.  At most 2 candidates exist.
   Returns the candidate count; y is [2][n]. */
int orc_pvq_noref_candidates(const coeff *x0, int n, int q0, double beta,
 const int16_t *qm, int nodesync, double *cg_out, double *g_out, int *qg, int *k,
 double *cos_dist, double *dist, coeff *y) {
  double x1[1024], g, cg;
  int i, nc = 0;
  cg = orc_pvq_compute_gain(x0, n, q0, &g, beta, qm);
  *cg_out = cg;
  *g_out = g;
  for (i = 0; i < n; i++) x1[i] = x0[i]*qm[i]*QM_SCALE_1;   /* int*int first */
  i = (int)floor(cg);
  if (i < 1) i = 1;
  for (; i <= ceil(cg) && nc < 2; i++, nc++) {
    double qcg = i;
    k[nc] = orc_pvq_compute_k(qcg, -1, -1, 1, n, beta, nodesync);
    cos_dist[nc] = orc_pvq_search_rdo_double(x1, n, k[nc], y + nc*n, qcg*cg);
    dist[nc] = 1.4*(qcg - cg)*(qcg - cg) + qcg*cg*(2 - 2*cos_dist[nc]);
    qg[nc] = i;
  }
  return nc;
}

/* ------------------------------------------------------------------------ */
/* CPU baseline of bench.py (kind "port"): the SAME work one bench step does on
   the device for one 4:2:0 frame, single thread: forward pyramid of all planes,
   no-reference PVQ candidates of every band of every block of every level,
   forward with known block sizes, inverse to 8-bit.  qm: the reference's
   OD_QM_BUFFER_SIZE table (state.qm), q0[pli], pvq_qm_q4[pli][20], masking.
   Returns a checksum so the work cannot be optimised away. */
long orc_bench_frame(const uint8_t *const pix[3], int fw, int fh, int pic_w,
 int pic_h, const unsigned char *bsize, const int16_t *qm, const int *q0,
 const unsigned char *pvq_qm_q4, int masking) {
  long sum = 0;
  int pli;
  for (pli = 0; pli < 3; pli++) {
    int dec = pli > 0, w = fw >> dec, h = fh >> dec, nlev = 4 - dec, k;
    int nhsb = fw/32, nvsb = fh/32;
    coeff *c = (coeff *)malloc(sizeof(coeff)*w*h);
    coeff *d = (coeff *)malloc(sizeof(coeff)*w*h);
    uint8_t *rec = (uint8_t *)malloc((size_t)w*h);
    coeff *lev[4];
    for (k = 0; k < nlev; k++) lev[k] = (coeff *)malloc(sizeof(coeff)*w*h);
    orc_forward_pyramid_plane(c, lev, nlev, pix[pli], w, nhsb, nvsb, dec, pic_w, pic_h);
    for (k = 0; k < nlev; k++) {
      int n = (32 >> dec) >> k, bs = n == 4 ? 0 : n == 8 ? 1 : n == 16 ? 2 : 3;
      int off[11], nb = orc_band_offsets(n, off), bx, by, b;
      const int16_t *qmb = qm + bs*2048 + dec*1024;
      for (by = 0; by < h/n; by++) {
        for (bx = 0; bx < w/n; bx++) {
          coeff co[1024];
          orc_raster_to_coding_order(co, n, lev[k] + (by*n)*w + bx*n, w);
          for (b = 0; b < nb; b++) {
            int nn = off[b + 1] - off[b], qg[2], kk[2], nc, q;
            double cg, g, cd[2], dist[2], beta;
            coeff y[2*128];
            q = q0[pli]*pvq_qm_q4[pli*20 + bs*(bs + 1) + (b + 1) - (b + 1)/3] >> 4;
            if (q < 1) q = 1;
            beta = (masking && pli == 0 && bs > 0) ? 1.5 : 1.0;
            nc = orc_pvq_noref_candidates(co + off[b], nn, q, beta, qmb + off[b], 1,
             &cg, &g, qg, kk, cd, dist, y);
            sum += nc ? kk[nc - 1] + y[0] : 0;
          }
        }
      }
    }
    orc_forward_plane(c, d, pix[pli], w, nhsb, nvsb, dec, bsize, nhsb*4, pic_w, pic_h, 1);
    orc_forward_plane(c, d, pix[pli], w, nhsb, nvsb, dec, bsize, nhsb*4, pic_w, pic_h, 0);
    orc_inverse_plane(rec, w, c, d, nhsb, nvsb, dec, bsize, nhsb*4, pic_w, pic_h);
    sum += rec[w*h/2] + d[w + 1];
    for (k = 0; k < nlev; k++) free(lev[k]);
    free(c); free(d); free(rec);
  }
  return sum;
}

/* ------------------------------------------------------------------------ */
/* A16, complete: every candidate pvq_theta (reference src/pvq_encoder.c:311-511)
   evaluates for one band, WITHOUT the rate term (od_pvq_rate needs the adaptive
   entropy state and stays on the host).  With-reference candidates (gain i,
   angle j) come first, in the reference's loop order, then the no-reference
   ones.  The caller reproduces the decision with
     cost = dist + lambda*rate,  '<' for with-ref, '<=' for no-ref candidates.
   Layout of the outputs (MAXC = 12 with-ref, 2 no-ref candidates):
     y_ref[c][n] (n-1 entries used), y_noref[c][n]. */
typedef struct orc_theta_out {
  double cg, cgr, g, gr, corr, theta, gain_offset, skip_dist, null_dist;
  int32_t icgr, m, s, nref, nnoref, theta_searched, noref_searched, pad;
  int32_t ref_qg[12], ref_itheta[12], ref_ts[12], ref_k[12];
  double ref_qtheta[12], ref_cos_dist[12], ref_dist[12];
  int32_t nr_qg[2], nr_k[2];
  double nr_cos_dist[2], nr_dist[2];
} orc_theta_out;

void orc_pvq_theta_candidates(const coeff *x0, const coeff *r0, int n, int q0,
 double beta, int robust, int is_keyframe, int pli, const int16_t *qm,
 orc_theta_out *o, coeff *y_ref, coeff *y_noref) {
  double x[1024], r[1024], g, gr, cg, cgr, corr = 0, gain_offset, theta = 0;
  const double gain_weight = 1.4;
  int i, icgr, m = 0, s = 1, cfl_enabled, nodesync = robust || is_keyframe;
  memset(o, 0, sizeof(*o));
  for (i = 0; i < n; i++) {
    x[i] = x0[i]*qm[i]*QM_SCALE_1;
    r[i] = r0[i]*qm[i]*QM_SCALE_1;
    corr += x[i]*r[i];
  }
  cfl_enabled = is_keyframe && pli != 0;
  cg = orc_pvq_compute_gain(x0, n, q0, &g, beta, qm);
  cgr = orc_pvq_compute_gain(r0, n, q0, &gr, beta, qm);
  if (cfl_enabled) cgr = 1;
  icgr = (int)floor(.5 + cgr);
  gain_offset = cgr - icgr;
  corr = corr/(1e-100 + g*gr);
  corr = corr < 1. ? corr : 1.;
  corr = corr > -1. ? corr : -1.;
  o->null_dist = gain_weight*cg*cg;
  if (is_keyframe) o->skip_dist = gain_weight*cg*cg;
  else o->skip_dist = gain_weight*(cg - cgr)*(cg - cgr) + cgr*cg*(2 - 2*corr);
  {
    int isnull = 1;
    for (i = 0; i < n; i++) if (r0[i]) isnull = 0;
    if (n <= 128 && !isnull && corr > 0) {
      o->theta_searched = 1;
      theta = acos(corr);
      m = orc_compute_householder(r, n, gr, &s);
      orc_apply_householder(x, r, n);
      for (i = m; i < n - 1; i++) x[i] = x[i + 1];
      i = (int)floor(cg - gain_offset) - 1;
      if (i < 1) i = 1;
      for (; i <= (int)ceil(cg - gain_offset); i++) {
        double qcg = i + gain_offset;
        int ts = orc_pvq_compute_max_theta(qcg, beta), j, jhi;
        j = (int)floor(.5 + theta*2/M_PI*ts) - 2;
        if (j < 0) j = 0;
        jhi = (int)ceil(theta*2/M_PI*ts);
        if (jhi > ts - 1) jhi = ts - 1;
        for (; j <= jhi; j++) {
          int c = o->nref;
          double qtheta = orc_pvq_compute_theta(j, ts), cos_dist, dist_theta;
          int k = orc_pvq_compute_k(qcg, j, qtheta, 0, n, beta, nodesync);
          if (c >= 12) break;
          cos_dist = orc_pvq_search_rdo_double(x, n - 1, k, y_ref + c*n,
           qcg*cg*sin(theta)*sin(qtheta));
          dist_theta = 2 - 2*cos(theta - qtheta) + sin(theta)*sin(qtheta)*(2 - 2*cos_dist);
          o->ref_qg[c] = i;
          o->ref_itheta[c] = j;
          o->ref_ts[c] = ts;
          o->ref_k[c] = k;
          o->ref_qtheta[c] = qtheta;
          o->ref_cos_dist[c] = cos_dist;
          o->ref_dist[c] = gain_weight*(qcg - cg)*(qcg - cg) + qcg*cg*dist_theta;
          o->nref++;
        }
      }
    }
  }
  if (n <= 128 && ((is_keyframe && pli == 0) || corr < .5 || cg < 2.)) {
    double x1[1024];
    o->noref_searched = 1;
    for (i = 0; i < n; i++) x1[i] = x0[i]*qm[i]*QM_SCALE_1;
    i = (int)floor(cg);
    if (i < 1) i = 1;
    for (; i <= ceil(cg) && o->nnoref < 2; i++) {
      int c = o->nnoref;
      double qcg = i;
      o->nr_k[c] = orc_pvq_compute_k(qcg, -1, -1, 1, n, beta, nodesync);
      o->nr_cos_dist[c] = orc_pvq_search_rdo_double(x1, n, o->nr_k[c], y_noref + c*n, qcg*cg);
      o->nr_dist[c] = gain_weight*(qcg - cg)*(qcg - cg) + qcg*cg*(2 - 2*o->nr_cos_dist[c]);
      o->nr_qg[c] = i;
      o->nnoref++;
    }
  }
  o->cg = cg; o->cgr = cgr; o->g = g; o->gr = gr; o->corr = corr; o->theta = theta;
  o->gain_offset = gain_offset; o->icgr = icgr; o->m = m; o->s = s;
}

/* Decoder-side synthesis of one band (reference pvq_synthesis,
   src/pvq_decoder.c:104-118): rebuilds the Householder reflection from the
   reference vector, then od_pvq_synthesis_partial. */
void orc_pvq_synthesis(coeff *xcoeff, const coeff *ypulse, const coeff *ref, int n,
 double gr, int noref, double g, double theta, const int16_t *qm,
 const int16_t *qm_inv) {
  double r[1024];
  int i, s = 0, m;
  if (!noref) for (i = 0; i < n; i++) r[i] = ref[i]*qm[i]*QM_SCALE_1;
  m = noref ? 0 : orc_compute_householder(r, n, gr, &s);
  orc_pvq_synthesis_partial(xcoeff, ypulse, r, n, noref, g, theta, m, s, qm_inv);
}

/* ------------------------------------------------------------------------ */
/* A22: perceptual block distortion of the block-size RDO (reference
   od_compute_var_4x4 / od_compute_dist_8x8 / od_compute_dist,
   src/encode.c:940-1058; HVS quantisation matrix case).  mag2[64]: the squared
   per-coefficient weights the reference derives from OD_QM8_Q4_HVS and
   OD_BASIS_MAG for this block size (data, passed in). */
static int var_4x4(const coeff *x, int stride) {
  int sum = 0, s2 = 0, i, j;
  for (i = 0; i < 4; i++)
    for (j = 0; j < 4; j++) {
      int t = x[i*stride + j] >> 2;
      sum += t;
      s2 += t*t;
    }
  return s2 - (sum*sum >> 4);
}

static double dist_8x8(const coeff *x, const coeff *y, int stride, const double *mag2,
 int masking) {
  coeff e[64], et[64];
  double sum = 0, mean_var = 0, vardist = 0, var_stat, activity, calibration;
  int min_var = 2147483647, i, j;
  for (i = 0; i < 3; i++)
    for (j = 0; j < 3; j++) {
      int varx = var_4x4(x + 2*i*stride + 2*j, stride);
      int vary = var_4x4(y + 2*i*stride + 2*j, stride);
      double diff;
      if (varx < min_var) min_var = varx;
      mean_var += 1./(1 + varx);
      diff = sqrt(varx) - sqrt(vary);
      vardist += diff*diff;
    }
  if (masking) {
    calibration = 1.95;
    var_stat = 9./mean_var;
  }
  else {
    calibration = 1.62;
    var_stat = min_var;
  }
  activity = calibration*pow(.25 + var_stat/(1 << 2*4), -1./6);
  for (i = 0; i < 8; i++)
    for (j = 0; j < 8; j++) e[8*i + j] = x[i*stride + j] - y[i*stride + j];
  orc_fdct_2d(8, et, 8, e, 8);
  for (i = 0; i < 64; i++) sum += et[i]*(double)et[i]*mag2[i];
  return activity*activity*(sum + vardist);
}

double orc_compute_dist(const coeff *x, const coeff *y, int n, const double *mag2,
 int masking) {
  double sum = 0;
  int i, j;
  for (i = 0; i < n; i += 8)
    for (j = 0; j < n; j += 8) sum += dist_8x8(x + i*n + j, y + i*n + j, n, mag2, masking);
  return sum*1.7;
}

/* ======================================================================== */
/* directional deringing, its threshold logic, and the
   keyframe bilinear smoothing - the pixel-domain stages between the frame
   post-filter and the 8-bit clamp (reference src/decode.c:1040-1155).         */

#define DER_BSTRIDE 38                 /* OD_FILT_BSTRIDE = 32 + 2*OD_FILT_BORDER */
#define DER_VERY_LARGE 30000           /* OD_DERING_VERY_LARGE, src/filter.c:1711 */

/* (dy, dx) of the three taps of each direction: direction_offsets_table,
   reference src/filter.c:132-141 (generated there as flat offsets). */
static const int8_t DER_DIR[8][3][2] = {
  {{-1, 1}, {-2, 2}, {-3, 3}}, {{0, 1}, {-1, 2}, {-1, 3}}, {{0, 1}, {0, 2}, {0, 3}},
  {{0, 1}, {1, 2}, {1, 3}}, {{1, 1}, {2, 2}, {3, 3}}, {{1, 0}, {2, 1}, {3, 1}},
  {{1, 0}, {2, 0}, {3, 0}}, {{1, 0}, {2, -1}, {3, -1}}};

/* od_dir_find8 (src/filter.c:1655-1708).  OD_DIVU_SMALL(x, d) is an exact
   unsigned division for the small divisors used here (src/internal.h:227). */
int orc_dir_find8(const int16_t *img, int stride, int32_t *var) {
  int cost[8] = {0}, partial[8][15], best_cost = 0, best_dir = 0, i, j;
  memset(partial, 0, sizeof(partial));
  for (i = 0; i < 8; i++)
    for (j = 0; j < 8; j++) {
      int x = img[i*stride + j] >> 4;
      partial[0][i + j] += x;
      partial[1][i + j/2] += x;
      partial[2][i] += x;
      partial[3][3 + i - j/2] += x;
      partial[4][7 + i - j] += x;
      partial[5][3 - i/2 + j] += x;
      partial[6][j] += x;
      partial[7][i/2 + j] += x;
    }
  for (i = 0; i < 8; i++) {
    cost[2] += partial[2][i]*partial[2][i] >> 3;
    cost[6] += partial[6][i]*partial[6][i] >> 3;
  }
  for (i = 0; i < 7; i++) {
    cost[0] += (int)((uint32_t)(partial[0][i]*partial[0][i])/(uint32_t)(i + 1))
     + (int)((uint32_t)(partial[0][14 - i]*partial[0][14 - i])/(uint32_t)(i + 1));
    cost[4] += (int)((uint32_t)(partial[4][i]*partial[4][i])/(uint32_t)(i + 1))
     + (int)((uint32_t)(partial[4][14 - i]*partial[4][14 - i])/(uint32_t)(i + 1));
  }
  cost[0] += partial[0][7]*partial[0][7] >> 3;
  cost[4] += partial[4][7]*partial[4][7] >> 3;
  for (i = 1; i < 8; i += 2) {
    for (j = 0; j < 5; j++) cost[i] += partial[i][3 + j]*partial[i][3 + j] >> 3;
    for (j = 0; j < 3; j++) {
      cost[i] += (int)((uint32_t)(partial[i][j]*partial[i][j])/(uint32_t)(2*j + 2))
       + (int)((uint32_t)(partial[i][10 - j]*partial[i][10 - j])/(uint32_t)(2*j + 2));
    }
  }
  for (i = 0; i < 8; i++) {
    if (cost[i] > best_cost) {
      best_cost = cost[i];
      best_dir = i;
    }
  }
  *var = best_cost - cost[(best_dir + 4) & 7];
  return best_dir;
}

/* od_filter_dering_direction_c / _orthogonal_c (src/filter.c:1714-1793); `in`
   points into a DER_BSTRIDE-strided buffer with a 3-sample border. */
void orc_dering_direction(int16_t *y, int ystride, const int16_t *in, int ln,
 int threshold, int dir) {
  static const int taps[3] = {3, 2, 2};
  int i, j, k, n = 1 << ln;
  for (i = 0; i < n; i++)
    for (j = 0; j < n; j++) {
      int xx = in[i*DER_BSTRIDE + j], sum = 0;
      for (k = 0; k < 3; k++) {
        int off = DER_DIR[dir][k][0]*DER_BSTRIDE + DER_DIR[dir][k][1];
        int p0 = in[i*DER_BSTRIDE + j + off] - xx;
        int p1 = in[i*DER_BSTRIDE + j - off] - xx;
        if (abs(p0) < threshold) sum += taps[k]*p0;
        if (abs(p1) < threshold) sum += taps[k]*p1;
      }
      y[i*ystride + j] = (int16_t)(xx + ((sum + 8) >> 4));
    }
}

void orc_dering_orthogonal(int16_t *y, int ystride, const int16_t *in,
 const int16_t *x, int xstride, int ln, int threshold, int dir) {
  int i, j, n = 1 << ln, offset = dir <= 4 ? DER_BSTRIDE : 1;
  for (i = 0; i < n; i++)
    for (j = 0; j < n; j++) {
      int yy = in[i*DER_BSTRIDE + j], sum = 0, p, athresh;
      athresh = threshold/3 + abs(in[i*DER_BSTRIDE + j] - x[i*xstride + j]);
      if (threshold < athresh) athresh = threshold;
      p = in[i*DER_BSTRIDE + j + offset] - yy;
      if (abs(p) < athresh) sum += p;
      p = in[i*DER_BSTRIDE + j - offset] - yy;
      if (abs(p) < athresh) sum += p;
      p = in[i*DER_BSTRIDE + j + 2*offset] - yy;
      if (abs(p) < athresh) sum += p;
      p = in[i*DER_BSTRIDE + j - 2*offset] - yy;
      if (abs(p) < athresh) sum += p;
      y[i*ystride + j] = (int16_t)(yy + ((3*sum + 8) >> 4));
    }
}

static int orc_ilog(uint32_t v) {
  int r = 0;
  while (v) { r++; v >>= 1; }
  return r;
}

/* od_dering (src/filter.c:1835-1940) for one dering superblock of one plane.
   threshold: (int)pow(q, 0.84182) computed by the caller (:1878).  dir: written
   for luma, read for chroma.  bskip points at the superblock's first 4x4 unit. */
void orc_dering_sb(int16_t *y, int ystride, const int16_t *x, int xstride, int ln,
 int sbx, int sby, int nhsb, int nvsb, int threshold, int xdec, int dir[4][4],
 int pli, const unsigned char *bskip, int skip_stride) {
  static const int16_t thresh_q8[18] = {128, 134, 150, 168, 188, 210, 234, 262, 292,
    327, 365, 408, 455, 509, 569, 635, 710, 768};
  int16_t inbuf[DER_BSTRIDE*DER_BSTRIDE], *in = inbuf + 3*DER_BSTRIDE + 3;
  int n = 1 << ln, bsize = 3 - xdec, nb = n >> bsize, i, j, bx, by, varsum = 0;
  int32_t var[4][4];
  int thresh[4][4];
  for (i = 0; i < DER_BSTRIDE*DER_BSTRIDE; i++) inbuf[i] = DER_VERY_LARGE;
  for (i = -3*(sby != 0); i < n + 3*(sby != nvsb - 1); i++)
    for (j = -3*(sbx != 0); j < n + 3*(sbx != nhsb - 1); j++)
      in[i*DER_BSTRIDE + j] = x[i*xstride + j];
  if (pli == 0) {
    for (by = 0; by < nb; by++)
      for (bx = 0; bx < nb; bx++) {
        dir[by][bx] = orc_dir_find8(x + 8*by*xstride + 8*bx, xstride, &var[by][bx]);
        varsum += var[by][bx];
      }
    for (by = 0; by < nb; by++)
      for (bx = 0; bx < nb; bx++) {
        int v1 = var[by][bx] >> 6, v2 = varsum/1024, t;
        if (v1 > 32767) v1 = 32767;
        if (v2 > 32767) v2 = 32767;
        t = orc_ilog((uint32_t)(v1*v2)) - 9;
        t = t < 0 ? 0 : t > 17 ? 17 : t;
        thresh[by][bx] = threshold*thresh_q8[t] >> 8;
      }
  }
  else {
    for (by = 0; by < nb; by++) for (bx = 0; bx < nb; bx++) thresh[by][bx] = threshold;
  }
  for (by = 0; by < nb; by++)
    for (bx = 0; bx < nb; bx++) {
      int xstart = sbx == 0 ? 0 : -1, ystart = sby == 0 ? 0 : -1;
      int xend = (2 >> xdec) + (sbx != nhsb - 1), yend = (2 >> xdec) + (sby != nvsb - 1);
      int skip = 1;
      for (i = ystart; i < yend; i++)
        for (j = xstart; j < xend; j++)
          skip = skip && bskip[((by << 1 >> xdec) + i)*skip_stride + (bx << 1 >> xdec) + j];
      if (skip) thresh[by][bx] = 0;
    }
  for (by = 0; by < nb; by++)
    for (bx = 0; bx < nb; bx++)
      orc_dering_direction(y + (by*ystride << bsize) + (bx << bsize), ystride,
       in + (by*DER_BSTRIDE << bsize) + (bx << bsize), bsize, thresh[by][bx], dir[by][bx]);
  for (i = 0; i < n; i++) for (j = 0; j < n; j++) in[i*DER_BSTRIDE + j] = y[i*ystride + j];
  for (by = 0; by < nb; by++)
    for (bx = 0; bx < nb; bx++)
      orc_dering_orthogonal(y + (by*ystride << bsize) + (bx << bsize), ystride,
       in + (by*DER_BSTRIDE << bsize) + (bx << bsize),
       x + (by*xstride << bsize) + (bx << bsize), xstride, bsize, thresh[by][bx],
       dir[by][bx]);
}

/* od_bilinear_smooth (src/filter.c:1952-2008). */
void orc_bilinear_smooth(coeff *x, int ln, int stride, int q, int pli) {
  static const int strength[4] = {5, 20, 20, 5};
  coeff yb[32][32];
  int n = 1 << ln, i, j, shift, w;
  int32_t dist = 0;
  coeff x00 = x[0], x01 = x[n - 1], x10 = x[(n - 1)*stride];
  coeff x11 = x[(n - 1)*stride + (n - 1)];
  coeff a00 = x00, a01 = x01 - x00, a10 = x10 - x00, a11 = x11 + x00 - x10 - x01;
  a01 += (a01 + n/2) >> ln;
  a10 += (a10 + n/2) >> ln;
  a11 += (2*a10 + n/2) >> ln;
  shift = 2*4 + 2*ln - 16;
  if (shift < 0) shift = 0;
  for (i = 0; i < n; i++) {
    int32_t partial = 0;
    for (j = 0; j < n; j++) {
      yb[i][j] = a00 + ((j*a01 + i*a10 + (j*i*a11 >> ln) + n/2) >> ln);
      partial += (yb[i][j] - x[i*stride + j])*(yb[i][j] - x[i*stride + j]);
    }
    dist += partial >> shift;
  }
  dist += n/2;
  dist >>= 2*ln - shift;
  w = strength[pli]*q*q/(1 + 12*dist);
  if (w > 1024) w = 1024;
  w = w*w >> 12;
  for (i = 0; i < n; i++)
    for (j = 0; j < n; j++)
      x[i*stride + j] -= (w*(x[i*stride + j] - yb[i][j]) + 128) >> 8;
}

/* The decoder's pixel-domain tail for one frame (src/decode.c:1040-1155):
   c[pli]: planes after the frame post-filter (int32, modified in place),
   flags: one byte per 32x32 dering superblock (1 = filtered), bskip[pli]: skip
   maps (1 byte per 4x4 of that plane, stride skip_stride), bsize: luma
   block-size map (stride bstride), thr[pli] = (int)pow(quantizer[pli], 0.84182),
   q[pli] = quantizer.  Writes the 8-bit planes. */
void orc_decode_tail(coeff *const c[3], uint8_t *const rec[3], int nplanes, int fw,
 int fh, const int *xdec, const unsigned char *flags, const unsigned char *const bskip[3],
 int skip_stride, const unsigned char *bsize, int bstride, const int *thr,
 const int *q, int is_keyframe) {
  int nhdr = fw >> 5, nvdr = fh >> 5, pli, sbx, sby, i;
  int16_t *e[3];
  if (q[0] > 0) {
    for (pli = 0; pli < nplanes; pli++) {
      int sz = (fw >> xdec[pli])*(fh >> xdec[pli]);
      e[pli] = (int16_t *)malloc(sizeof(int16_t)*sz);
      for (i = 0; i < sz; i++) e[pli][i] = (int16_t)c[pli][i];
    }
    for (sby = 0; sby < nvdr; sby++)
      for (sbx = 0; sbx < nhdr; sbx++) {
        int dir[4][4];
        if (!flags[sby*nhdr + sbx]) continue;
        for (pli = 0; pli < nplanes; pli++) {
          int d = xdec[pli], w = fw >> d, ln = 5 - d, n = 1 << ln, y, x;
          int16_t buf[32*32];
          orc_dering_sb(buf, n, e[pli] + (sby << ln)*w + (sbx << ln), w, ln, sbx, sby,
           nhdr, nvdr, thr[pli], d, dir, pli,
           bskip[pli] + (sby << (3 - d))*skip_stride + (sbx << (3 - d)), skip_stride);
          for (y = 0; y < n; y++)
            for (x = 0; x < n; x++) c[pli][((sby << ln) + y)*w + (sbx << ln) + x] = buf[y*n + x];
        }
      }
    for (pli = 0; pli < nplanes; pli++) free(e[pli]);
  }
  for (pli = 0; pli < nplanes; pli++) {
    int d = xdec[pli], w = fw >> d, h = fh >> d;
    if (q[0] > 0 && is_keyframe) {
      for (sby = 0; sby < nvdr; sby++)
        for (sbx = 0; sbx < nhdr; sbx++) {
          /* od_smooth_recursive with min_bs = OD_BLOCK_32X32: only whole-superblock
             leaves are smoothed (src/filter.c:2010-2040) */
          if (bsize[(sby*4)*bstride + sbx*4] == 3) {
            int ln = 5 - d;
            orc_bilinear_smooth(c[pli] + (sby << ln)*w + (sbx << ln), ln, w, q[pli], pli);
          }
        }
    }
    orc_coeff_to_ref_buf(rec[pli], w, c[pli], w, w, h, 4);
  }
}

/* ------------------------------------------------------------------------ */
/* Encoder feed of one luma pyramid level in the layout of include/daala_hip.h
   section 4b (what od_hip_enc_feed_view hands to the host workers), computed
   with the oracle: the checker for the device feed and the stand-in producer
   for the CPU tests of the reference-side glue (tests/ only).
   lev: level plane (w x h, blocks of n x n); qm: the n*n QM of this block size;
   q, beta: per band.  Arrays sized as the header says. */
void orc_feed_level(const coeff *lev, int w, int h, int n, const int16_t *qm,
 const int *q, const double *beta, double *cg, double *gout, int32_t *ncand, int32_t *qg,
 int32_t *k, double *cos_dist, int32_t *y) {
  int off[11], nb = orc_band_offsets(n, off), nbx = w/n, nby = h/n, nblk = nbx*nby;
  int bx, by, b, c;
  size_t nrec = (size_t)nb*nblk;
  for (by = 0; by < nby; by++) {
    for (bx = 0; bx < nbx; bx++) {
      coeff co[1024];
      int blk = by*nbx + bx;
      orc_raster_to_coding_order(co, n, lev + (size_t)(by*n)*w + bx*n, w);
      for (b = 0; b < nb; b++) {
        int nn = off[b + 1] - off[b], cqg[2], ck[2], nc;
        double g, cd[2], dist[2];
        coeff yy[2*128];
        size_t r = (size_t)b*nblk + blk;
        nc = orc_pvq_noref_candidates(co + off[b], nn, q[b], beta[b], qm + off[b], 1,
         &cg[r], &g, cqg, ck, cd, dist, yy);
        ncand[r] = nc;
        gout[r] = g;
        for (c = 0; c < 2; c++) {
          int32_t *dst = y + (size_t)2*nblk*(off[b] - 1) + ((size_t)c*nblk + blk)*nn;
          qg[c*nrec + r] = c < nc ? cqg[c] : 0;
          k[c*nrec + r] = c < nc ? ck[c] : 0;
          cos_dist[c*nrec + r] = c < nc ? cd[c] : 0;
          if (c < nc) memcpy(dst, yy + c*nn, sizeof(coeff)*nn);
          else memset(dst, 0, sizeof(coeff)*nn);
        }
      }
    }
  }
}

/* ------------------------------------------------------------------------ */
/* Lossless frames (reference src/encode.c:3002,3090-3092; src/decode.c:785,1036):
   coefficient shift 0, no lapping, od_haar / od_haar_inv of every whole superblock
   (sb = 32 luma, 16 for 4:2:0 chroma). */
void orc_haar_forward_plane(coeff *d, const uint8_t *pix, int w, int h, int sb) {
  coeff *c = (coeff *)malloc(sizeof(coeff)*w*h);
  int ln = sb == 32 ? 5 : 4, x, y;
  orc_ref_buf_to_coeff(c, w, pix, w, w, h, 0);
  for (y = 0; y < h; y += sb)
    for (x = 0; x < w; x += sb) orc_haar(d + (size_t)y*w + x, w, c + (size_t)y*w + x, w, ln);
  free(c);
}

void orc_haar_inverse_plane(uint8_t *pix, const coeff *d, int w, int h, int sb) {
  coeff *c = (coeff *)malloc(sizeof(coeff)*w*h);
  int ln = sb == 32 ? 5 : 4, x, y;
  for (y = 0; y < h; y += sb)
    for (x = 0; x < w; x += sb) orc_haar_inv(c + (size_t)y*w + x, w, d + (size_t)y*w + x, w, ln);
  orc_coeff_to_ref_buf(pix, w, c, w, w, h, 0);
  free(c);
}

/* ------------------------------------------------------------------------ */
/* A3, all sizes: the n-point lapping pre/post filters od_pre_filter{4,8,16,32} /
   od_post_filter{4,8,16,32} (reference src/filter.c:174-249, :306-440, :546-808,
   :879-1380; TYPE3 rotation structure, parameters in filter_params.h).  Only the
   4-point pair is reachable in the codec (OD_FILT_SIZE() == 0, src/filter.h:99); the
   larger ones are exercised by the reference's dcttest/tools.  In/out may alias.
     pre : +-1 butterflies; scale t[h+k] = (t*P_k) >> 6, "+1 if positive" (skipped for
           P_k == 64); rotations j = n-2 .. h: t[j+1] += (t[j]*A_j+32)>>6,
           t[j] += (t[j+1]*B_j+32)>>6; closing butterflies.
     post: same butterflies; rotations undone j = h .. n-2 in reverse; scale undone
           with C truncating division (t << 6)/P_k; closing butterflies. */
static const int *lap_params(int n) {
  static const int p4[] = LAP_PARAMS4, p8[] = LAP_PARAMS8, p16[] = LAP_PARAMS16,
   p32[] = LAP_PARAMS32;
  return n == 4 ? p4 : n == 8 ? p8 : n == 16 ? p16 : p32;
}

void orc_pre_filter_n(int n, coeff *y, const coeff *x) {
  const int *P = lap_params(n);
  int h = n/2, i, j;
  coeff t[32];
  for (i = 0; i < h; i++) t[n - 1 - i] = x[i] - x[n - 1 - i];
  for (i = 0; i < h; i++) t[i] = x[i] - (t[n - 1 - i] >> 1);
  for (i = 0; i < h; i++) {
    if (P[i] != 64) {
      t[h + i] = (t[h + i]*P[i]) >> 6;
      t[h + i] += t[h + i] > 0;
    }
  }
  for (j = n - 2; j >= h; j--) {
    t[j + 1] += (t[j]*P[h + (j - h)] + 32) >> 6;
    t[j] += (t[j + 1]*P[2*h - 1 + (j - h)] + 32) >> 6;
  }
  for (i = 0; i < h; i++) t[i] += t[n - 1 - i] >> 1;
  for (i = 0; i < h; i++) y[i] = t[i];
  for (i = 0; i < h; i++) y[n - 1 - i] = t[i] - t[n - 1 - i];
}

void orc_post_filter_n(int n, coeff *x, const coeff *y) {
  const int *P = lap_params(n);
  int h = n/2, i, j;
  coeff t[32];
  for (i = 0; i < h; i++) t[n - 1 - i] = y[i] - y[n - 1 - i];
  for (i = 0; i < h; i++) t[i] = y[i] - (t[n - 1 - i] >> 1);
  for (j = h; j <= n - 2; j++) {
    t[j] -= (t[j + 1]*P[2*h - 1 + (j - h)] + 32) >> 6;
    t[j + 1] -= (t[j]*P[h + (j - h)] + 32) >> 6;
  }
  for (i = 0; i < h; i++) {
    if (P[i] != 64) t[h + i] = (t[h + i]*64)/P[i];
  }
  for (i = 0; i < h; i++) t[i] += t[n - 1 - i] >> 1;
  for (i = 0; i < h; i++) x[i] = t[i];
  for (i = 0; i < h; i++) x[n - 1 - i] = t[i] - t[n - 1 - i];
}

/* ------------------------------------------------------------------------ */
/* F3: overlapped block motion compensation of one prediction block,
   8-bit references.  Restates od_mc_predict1fmv8_c (reference src/mc.c:94-203),
   od_mc_blend_full8_c (:352-377), od_mc_setup_s_split + od_mc_blend_full_split8_c
   (:1056-1151) and od_mc_predict / od_mc_blend (:1938-2034; the multiresolution branch is
   compiled out there by `0 &&`). */
static const int16_t ORC_SUBPEL[8][6] = {      /* OD_SUBPEL_FILTER_SET, src/mc.c:66-77 */
  {0, 0, 128, 0, 0, 0}, {1, -9, 122, 18, -5, 1}, {3, -15, 112, 37, -11, 2},
  {3, -18, 97, 58, -15, 3}, {4, -20, 80, 80, -20, 4}, {3, -15, 58, 97, -18, 3},
  {2, -11, 37, 112, -15, 3}, {1, -5, 18, 122, -9, 1}};

static int orc_clamp255(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

/* dst: xblk x yblk, dense (stride xblk) */
void orc_mc_predict1fmv8(uint8_t *dst, const uint8_t *src, int systride, int32_t mvx,
 int32_t mvy, int log_xblk_sz, int log_yblk_sz) {
  int xblk = 1 << log_xblk_sz, yblk = 1 << log_yblk_sz, i, j, k;
  int mvxf = mvx & 7, mvyf = mvy & 7;
  const int16_t *fx = ORC_SUBPEL[mvxf], *fy = ORC_SUBPEL[mvyf];
  int16_t buff[(64 + 5)*64];
  const uint8_t *sp = src + (mvx >> 3) + (mvy >> 3)*systride;
  if (mvxf || mvyf) {
    int16_t *bp = buff;
    sp -= systride*2;
    for (j = -2; j < yblk + 3; j++) {
      for (i = 0; i < xblk; i++) {
        if (mvxf) {
          int32_t sum = 0;
          for (k = 0; k < 6; k++) sum += sp[i + k - 2]*fx[k];
          bp[i] = (int16_t)(sum - (128 << 7));
        }
        else bp[i] = (int16_t)((sp[i] << 7) - (128 << 7));
      }
      sp += systride;
      bp += xblk;
    }
    bp = buff + xblk*2;
    for (j = 0; j < yblk; j++) {
      for (i = 0; i < xblk; i++) {
        if (mvyf) {
          int32_t sum = 0;
          for (k = 0; k < 6; k++) sum += bp[i + (k - 2)*xblk]*fy[k];
          dst[i] = (uint8_t)orc_clamp255((sum + (1 << 13) + (128 << 14)) >> 14);
        }
        else dst[i] = (uint8_t)orc_clamp255((bp[i] + (1 << 6) + (128 << 7)) >> 7);
      }
      bp += xblk;
      dst += xblk;
    }
  }
  else {
    for (j = 0; j < yblk; j++) {
      for (i = 0; i < xblk; i++) dst[j*xblk + i] = sp[j*systride + i];
    }
  }
}

void orc_mc_blend_full8(uint8_t *dst, int dystride, const uint8_t *const src[4],
 int log_xblk_sz, int log_yblk_sz) {
  int xblk = 1 << log_xblk_sz, yblk = 1 << log_yblk_sz, l2 = log_xblk_sz + log_yblk_sz;
  int round = 1 << (l2 - 1), i, j;
  for (j = 0; j < yblk; j++) {
    for (i = 0; i < xblk; i++) {
      int32_t a = src[0][j*xblk + i], b = src[3][j*xblk + i];
      a = (a << log_xblk_sz) + (src[1][j*xblk + i] - a)*i;
      b = (b << log_xblk_sz) + (src[2][j*xblk + i] - b)*i;
      dst[j*dystride + i] = (uint8_t)(((a << log_yblk_sz) + (b - a)*j + round) >> l2);
    }
  }
}

void orc_mc_blend_full_split8(uint8_t *dst, int dystride, const uint8_t *const src[4], int oc,
 int s, int log_xblk_sz, int log_yblk_sz) {
  int xblk = 1 << log_xblk_sz, yblk = 1 << log_yblk_sz, l2 = log_xblk_sz + log_yblk_sz;
  int s0[4], dsdi[4], dsdj[4], dd[4], sw[4], i, j, k, round = 1 << l2;
  s0[0] = 2 << l2; s0[1] = s0[2] = s0[3] = 0;
  dsdi[0] = -(2 << log_xblk_sz); dsdi[1] = 2 << log_xblk_sz; dsdi[2] = dsdi[3] = 0;
  dsdj[0] = -(2 << log_yblk_sz); dsdj[1] = dsdj[2] = 0; dsdj[3] = 2 << log_yblk_sz;
  dd[0] = dd[2] = 2; dd[1] = dd[3] = -2;
  for (k = 0; k < 2; k++) {
    /* an unsplit edge hands half of the neighbouring corner's weight to the outside corner */
    int on = k == 0 ? !(s & 1) : !(s & 2), c = k == 0 ? (oc + 1) & 3 : (oc + 3) & 3;
    if (on) {
      s0[c] >>= 1; s0[oc] += s0[c];
      dsdi[c] >>= 1; dsdi[oc] += dsdi[c];
      dsdj[c] >>= 1; dsdj[oc] += dsdj[c];
      dd[c] >>= 1; dd[oc] += dd[c];
    }
  }
  for (k = 0; k < 4; k++) sw[k] = s0[k];
  for (j = 0; j < yblk; j++) {
    for (i = 0; i < xblk; i++) {
      int32_t a = src[0][j*xblk + i];
      int32_t b = (src[1][j*xblk + i] - a)*sw[1];
      int32_t c = (src[2][j*xblk + i] - a)*sw[2];
      int32_t d = (src[3][j*xblk + i] - a)*sw[3];
      dst[j*dystride + i] = (uint8_t)(((a << (l2 + 1)) + b + c + d + round) >> (l2 + 1));
      for (k = 0; k < 4; k++) sw[k] += dsdi[k];
    }
    for (k = 0; k < 4; k++) {
      s0[k] += dsdj[k];
      sw[k] = s0[k];
      dsdi[k] += dd[k];
    }
  }
}

/* od_mc_predict: four single-vector predictions (one per corner of the block, each from its
   own reference) blended by position. */
void orc_mc_predict(uint8_t *dst, int dystride, const uint8_t *const src[4], int systride,
 const int32_t mvx[4], const int32_t mvy[4], int oc, int s, int log_xblk_sz,
 int log_yblk_sz) {
  static __thread uint8_t pred[4][64*64];
  const uint8_t *p[4];
  int k;
  for (k = 0; k < 4; k++) {
    orc_mc_predict1fmv8(pred[k], src[k], systride, mvx[k], mvy[k], log_xblk_sz, log_yblk_sz);
    p[k] = pred[k];
  }
  if (s == 3) orc_mc_blend_full8(dst, dystride, p, log_xblk_sz, log_yblk_sz);
  else orc_mc_blend_full_split8(dst, dystride, p, oc, s, log_xblk_sz, log_yblk_sz);
}

/* ---------------------------------------------------------------------------------
 * F3: the batch stages of the motion search.  od_mv_est_sad (src/mcenc.c:2271-2300) of one
 * (block, exterior corner, split state): od_state_pred_block_from_setup (src/state.c:689-734)
 * of every plane - corner vectors scaled with OD_DIV_POW2_RE (src/odintrin.h:142) - and
 * od_enc_sad (src/mcenc.c:1615-1681: the block clipped against the picture), chroma sums
 * >> OD_MC_CHROMA_SCALE (:53).  An item carries what the reference reads from the vector grid.
 * Layout == od_hip_mc_sad_item (include/daala_hip.h). */
typedef struct orc_sad_item {
  int32_t x, y, log_blk_sz, oc, s;
  int32_t ref[4];
  int32_t mvx[4], mvy[4];
  int32_t reserved;
} orc_sad_item;

static int orc_div_pow2_re(int x, int shift) {
  return (x + (((1 << shift) + ((x >> shift) & 1) - 1) >> 1)) >> shift;
}

/* refs[pli]: nref reference planes of ref_h[pli] x ref_stride[pli] bytes, back to back, the
   picture origin at (org_x[pli], org_y[pli]); src[pli]: the frame being coded, src_stride[pli]
   bytes per row. */
void orc_mv_est_sad_items(const orc_sad_item *items, int nitems, int nplanes,
 const uint8_t *const refs[3], const int32_t *ref_stride, const int32_t *ref_h,
 const int32_t *org_x, const int32_t *org_y, const uint8_t *const src[3],
 const int32_t *src_stride, const int32_t *xdec, const int32_t *ydec, int pic_w, int pic_h,
 int32_t *sad) {
  static __thread uint8_t pred[64*64];
  int n;
  for (n = 0; n < nitems; n++) {
    const orc_sad_item *it = items + n;
    int32_t total = 0;
    int pli;
    for (pli = 0; pli < nplanes; pli++) {
      const uint8_t *at[4];
      int32_t mvx[4], mvy[4];
      int lx = it->log_blk_sz - xdec[pli], ly = it->log_blk_sz - ydec[pli];
      int bx = it->x >> xdec[pli], by = it->y >> ydec[pli];
      int w = 1 << lx, h = 1 << ly;
      int clipw = ((pic_w + (1 << xdec[pli]) - 1) >> xdec[pli]) - bx;
      int cliph = ((pic_h + (1 << ydec[pli]) - 1) >> ydec[pli]) - by;
      int32_t acc = 0;
      int i, j, k;
      for (k = 0; k < 4; k++) {
        mvx[k] = orc_div_pow2_re(it->mvx[k], xdec[pli]);
        mvy[k] = orc_div_pow2_re(it->mvy[k], ydec[pli]);
        at[k] = refs[pli] + (size_t)it->ref[k]*ref_stride[pli]*ref_h[pli]
         + (size_t)(org_y[pli] + by)*ref_stride[pli] + org_x[pli] + bx;
      }
      orc_mc_predict(pred, w, at, ref_stride[pli], mvx, mvy, it->oc, it->s, lx, ly);
      if (clipw < w) w = clipw;
      if (cliph < h) h = cliph;
      for (j = 0; j < h; j++) {
        for (i = 0; i < w; i++) {
          acc += abs((int)pred[(j << lx) + i] - (int)src[pli][(size_t)(by + j)*src_stride[pli] + bx + i]);
        }
      }
      total += pli > 0 ? acc >> 2 : acc;
    }
    sad[n] = total;
  }
}

/* od_mv_est_bma_sad (src/mcenc.c:2228-2268) for every half-sample vector of a window: the
   block-matching SAD the EPZS initialisation (od_mv_est_init_mv, :2511) judges its candidates by.
   ONE single-vector prediction per plane (od_mc_predict1fmv8_c with mvx*(1 << (2 - xdec)), the
   block's corner at (bx >> xdec, by >> ydec) of the reference - it may lie in the padding), then
   od_enc_sad (:1615-1681): the block clipped against the picture on all four sides, chroma >>
   OD_MC_CHROMA_SCALE.  Layout == od_hip_mc_bma_rec; out: [nrec][(2R + 1)^2], -1 outside the limits. */
typedef struct orc_bma_rec {
  int32_t bx, by, log_blk_sz, ref, cx, cy, xmin, xmax, ymin, ymax;
} orc_bma_rec;

void orc_mv_est_bma_windows(const orc_bma_rec *recs, int nrec, int radius, int nplanes,
 const uint8_t *const refs[3], const int32_t *ref_stride, const int32_t *ref_h,
 const int32_t *org_x, const int32_t *org_y, const uint8_t *const src[3],
 const int32_t *src_stride, const int32_t *xdec, const int32_t *ydec, int pic_w, int pic_h,
 int32_t *out) {
  static __thread uint8_t pred[64*64];
  int W = 2*radius + 1;
  int n, o;
  for (n = 0; n < nrec; n++) {
    const orc_bma_rec *r = recs + n;
    for (o = 0; o < W*W; o++) {
      int mvx = r->cx + o%W - radius, mvy = r->cy + o/W - radius;
      int32_t total = 0;
      int pli;
      if (mvx < r->xmin || mvx > r->xmax || mvy < r->ymin || mvy > r->ymax) {
        out[(size_t)n*W*W + o] = -1;
        continue;
      }
      for (pli = 0; pli < nplanes; pli++) {
        int lx = r->log_blk_sz - xdec[pli], ly = r->log_blk_sz - ydec[pli];
        int x = r->bx >> xdec[pli], y = r->by >> ydec[pli];
        int w = 1 << lx, h = 1 << ly;
        int clipw = (pic_w + (1 << xdec[pli]) - 1) >> xdec[pli];
        int cliph = (pic_h + (1 << ydec[pli]) - 1) >> ydec[pli];
        const uint8_t *at = refs[pli] + (size_t)r->ref*ref_stride[pli]*ref_h[pli]
         + (ptrdiff_t)(org_y[pli] + y)*ref_stride[pli] + org_x[pli] + x;
        int32_t acc = 0;
        int i, j;
        orc_mc_predict1fmv8(pred, at, ref_stride[pli], mvx*(1 << (2 - xdec[pli])),
         mvy*(1 << (2 - ydec[pli])), lx, ly);
        for (j = 0; j < h; j++) {
          for (i = 0; i < w; i++) {
            if (x + i >= 0 && x + i < clipw && y + j >= 0 && y + j < cliph) {
              acc += abs((int)pred[(j << lx) + i] - (int)src[pli][(size_t)(y + j)*src_stride[pli] + x + i]);
            }
          }
        }
        total += pli > 0 ? acc >> 2 : acc;
      }
      out[(size_t)n*W*W + o] = total;
    }
  }
}

/* OD_VERT_D / OD_VERT_SETUP_DX / OD_VERT_SETUP_DY (src/state.c:645-687): the grid offsets, in
   units of the block size, of the four vectors a block is predicted from, by exterior corner
   and split state. */
static const int ORC_VERT_D[22] = {0, 0, 1, 1, 0, 0, 1, 2, 0, 0, 2, 1, 0, -1, 1, 1, 0, -1, 0, 1, 1, -1};
static const int ORC_SETUP_DX[4][4] = {{9, 1, 9, 1}, {13, 13, 1, 1}, {18, 1, 18, 1}, {5, 5, 1, 1}};
static const int ORC_SETUP_DY[4][4] = {{4, 4, 0, 0}, {8, 0, 8, 0}, {12, 12, 0, 0}, {17, 0, 17, 0}};

/* One item from the grid: what od_state_pred_block_from_setup reads (P frames: mv, ref).
   gmvx/gmvy/gref: (nvmvbs + 1) x (nhmvbs + 1), gref already mapped to an image index. */
static void orc_mv_est_item(orc_sad_item *it, int nhmvbs, const int32_t *gmvx, const int32_t *gmvy,
 const int32_t *gref, int vx, int vy, int oc, int s, int log_mvb_sz) {
  const int *dxp = ORC_VERT_D + ORC_SETUP_DX[oc][s];
  const int *dyp = ORC_VERT_D + ORC_SETUP_DY[oc][s];
  int k;
  memset(it, 0, sizeof(*it));
  it->x = vx << 3;                  /* OD_LOG_MVBSIZE_MIN (src/internal.h:67) */
  it->y = vy << 3;
  it->log_blk_sz = log_mvb_sz + 3;
  it->oc = oc;
  it->s = s;
  for (k = 0; k < 4; k++) {
    int g = (vy + dyp[k]*(1 << log_mvb_sz))*(nhmvbs + 1) + vx + dxp[k]*(1 << log_mvb_sz);
    it->mvx[k] = gmvx[g];
    it->mvy[k] = gmvy[g];
    it->ref[k] = gref[g];
  }
}

/* od_mv_est_calc_sads (src/mcenc.c:3761-3823), the SAD part: for every block size the level
   limits admit, every block, split states 0 .. smax - 1, in the reference's loop order.  Writes
   the items; sizes[l] receives the number of items of log_mvb_sz l (0 when the size is not
   evaluated), smax_out[l] its smax.  Returns the item count (items may be NULL to count). */
int orc_mv_est_calc_sads_items(int nhmvbs, int nvmvbs, int level_min, int level_max,
 const int32_t *gmvx, const int32_t *gmvy, const int32_t *gref, orc_sad_item *items,
 int32_t *sizes, int32_t *smax_out) {
  int n = 0;
  int nh = nhmvbs, nv = nvmvbs;
  int l;
  for (l = 0; l < 3; l++) {         /* OD_LOG_MVB_DELTA0 (src/internal.h:87) */
    sizes[l] = 0;
    smax_out[l] = 0;
    if (level_max >= 6 - 1 - 2*l && level_min <= 6 - 2*l) {     /* OD_MC_LEVEL_MAX = 6 */
      int smax = level_max >= 6 - 2*l ? 4 : 1;
      int vx, vy, s;
      smax_out[l] = smax;
      for (vy = 0; vy < nv; vy++) {
        for (vx = 0; vx < nh; vx++) {
          int oc = (vx & 1) ^ ((vy & 1) << 1 | (vy & 1));
          for (s = 0; s < smax; s++) {
            if (items != NULL) orc_mv_est_item(items + n, nhmvbs, gmvx, gmvy, gref, vx << l, vy << l, oc, s, l);
            n++;
            sizes[l]++;
          }
        }
      }
    }
    nh >>= 1;
    nv >>= 1;
  }
  return n;
}

/* ---------------------------------------------------------------------------------
 * F3: SAD and SATD of a block pair (the C entries of od_enc_opt_vtbl, src/encint.h:61-82).
 * orc_mc_sad8: sum |ref - src| over the block (od_mc_compute_sad8_c, src/mcenc.c:1333-1347).
 * orc_mc_satd8: od_mc_compute_satd8 (src/mcenc.c:1464-1489): difference block, n-point
 * Hadamard of rows then of columns (od_mc_hadamard_1d :1415-1461, sums in the low half and
 * differences in the high half, recursing on both), sum of magnitudes, (sum + n/2) >> ln;
 * 4x4 is one such transform and larger blocks are the sum over their 8x8 sub-blocks
 * (od_mc_compute_sum_8x8_satd8 :1520-1539, entries :1562-1612). */
int32_t orc_mc_sad8(const uint8_t *src, int systride, const uint8_t *ref, int rystride, int ln) {
  int32_t sad = 0;
  int n = 1 << ln;
  int i, j;
  for (i = 0; i < n; i++) {
    for (j = 0; j < n; j++) sad += abs((int)ref[i*rystride + j] - (int)src[i*systride + j]);
  }
  return sad;
}

static void orc_hadamard_1d(int32_t *v, int step, int n) {
  /* in place on v[0], v[step], ..., v[(n-1)*step] */
  int32_t t[8];
  int h = n >> 1;
  int i;
  if (n < 2) return;
  for (i = 0; i < h; i++) {
    t[i] = v[(2*i)*step] + v[(2*i + 1)*step];
    t[h + i] = v[(2*i)*step] - v[(2*i + 1)*step];
  }
  for (i = 0; i < n; i++) v[i*step] = t[i];
  orc_hadamard_1d(v, step, h);
  orc_hadamard_1d(v + h*step, step, h);
}

static int32_t orc_satd_sub(const uint8_t *src, int systride, const uint8_t *ref, int rystride,
                            int ln) {
  int32_t w[64];
  int n = 1 << ln;
  int32_t sum = 0;
  int i, j;
  for (i = 0; i < n; i++) {
    for (j = 0; j < n; j++) w[i*n + j] = (int32_t)src[i*systride + j] - (int32_t)ref[i*rystride + j];
  }
  for (i = 0; i < n; i++) orc_hadamard_1d(w + i*n, 1, n);
  for (j = 0; j < n; j++) orc_hadamard_1d(w + j, n, n);
  for (i = 0; i < n*n; i++) sum += abs(w[i]);
  return (sum + (1 << ln >> 1)) >> ln;
}

int32_t orc_mc_satd8(const uint8_t *src, int systride, const uint8_t *ref, int rystride, int ln) {
  int n = 1 << ln;
  int32_t satd = 0;
  int i, j;
  if (ln == 2) return orc_satd_sub(src, systride, ref, rystride, 2);
  for (i = 0; i < n; i += 8) {
    for (j = 0; j < n; j += 8) {
      satd += orc_satd_sub(src + i*systride + j, systride, ref + i*rystride + j, rystride, 3);
    }
  }
  return satd;
}
