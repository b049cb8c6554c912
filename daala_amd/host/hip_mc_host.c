/* hip_mc_host.c - the per-block leaves of the motion search that stay on the host.
 *
 * od_mv_est (src/mcenc.c:6390) decides candidate by candidate: every candidate vector of every
 * grid point is judged by predicting the (at most four) blocks around it with OBMC and taking
 * their SAD, and the next candidate depends on the outcome.  One such prediction is a few
 * hundred pixels - far below what a kernel launch and a PCIe round trip cost - so inside the
 * search the two leaves stay host code; whole-frame predictions (od_state_mc_predict, the
 * frame the residual is taken from) go to the device (od_hip_mc_predict).
 *
 * A profile of two 1080p P frames on one worker puts 37 % of all host time in the blend and
 * 20 % in the sub-pel predictor as the reference's scalar C compiles them, so both are bound
 * through od_state_opt_vtbl (src/state.h:106-119) here with the host's vector unit, eight
 * pixels at a time, same integer arithmetic:
 *   od_hipenc_mc_blend_full8     replaces od_mc_blend_full8_c     (src/mc.c:352-377)
 *   od_hipenc_mc_predict1fmv8    replaces od_mc_predict1fmv8_c    (src/mc.c:94-203)
 */
#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mc.h"

#include "hip_glue_int.h"

/* dst[i] = ((a << ly) + (b - a)*j + round) >> (lx + ly) with
   a = (s0 << lx) + (s1 - s0)*i and b = (s3 << lx) + (s2 - s3)*i   (src/mc.c:364-376) */
void od_hipenc_mc_blend_full8(unsigned char *dst, int dystride, const unsigned char *src[4],
 int log_xblk_sz, int log_yblk_sz) {
  const unsigned char *s0;
  const unsigned char *s1;
  const unsigned char *s2;
  const unsigned char *s3;
  int xblk_sz;
  int yblk_sz;
  int log_blk_sz2;
  int round;
  int i;
  int j;
  xblk_sz = 1 << log_xblk_sz;
  yblk_sz = 1 << log_yblk_sz;
  log_blk_sz2 = log_xblk_sz + log_yblk_sz;
  round = 1 << (log_blk_sz2 - 1);
  s0 = src[0];
  s1 = src[1];
  s2 = src[2];
  s3 = src[3];
  if (xblk_sz >= 16) {
    /* sixteen pixels per step; the 32-bit halves come out of the unpacks per 128-bit lane and
       the saturating packs put them back in order */
    const __m256i vround = _mm256_set1_epi32(round);
    const __m128i cx = _mm_cvtsi32_si128(log_xblk_sz);
    const __m128i c2 = _mm_cvtsi32_si128(log_blk_sz2);
    for (j = 0; j < yblk_sz; j++) {
      const __m256i wj = _mm256_set1_epi32((1 << log_yblk_sz) | (j << 16));
      __m256i vi;
      vi = _mm256_setr_epi16(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
      for (i = 0; i < xblk_sz; i += 16) {
        __m256i a;
        __m256i b;
        __m256i p1;
        __m256i p2;
        __m256i lo;
        __m256i hi;
        a = _mm256_cvtepu8_epi16(_mm_loadu_si128((const __m128i *)(s0 + i)));
        p1 = _mm256_cvtepu8_epi16(_mm_loadu_si128((const __m128i *)(s1 + i)));
        p2 = _mm256_cvtepu8_epi16(_mm_loadu_si128((const __m128i *)(s2 + i)));
        b = _mm256_cvtepu8_epi16(_mm_loadu_si128((const __m128i *)(s3 + i)));
        a = _mm256_add_epi16(_mm256_sll_epi16(a, cx), _mm256_mullo_epi16(_mm256_sub_epi16(p1, a), vi));
        b = _mm256_add_epi16(_mm256_sll_epi16(b, cx), _mm256_mullo_epi16(_mm256_sub_epi16(p2, b), vi));
        b = _mm256_sub_epi16(b, a);
        lo = _mm256_madd_epi16(_mm256_unpacklo_epi16(a, b), wj);
        hi = _mm256_madd_epi16(_mm256_unpackhi_epi16(a, b), wj);
        lo = _mm256_sra_epi32(_mm256_add_epi32(lo, vround), c2);
        hi = _mm256_sra_epi32(_mm256_add_epi32(hi, vround), c2);
        lo = _mm256_packs_epi32(lo, hi);
        lo = _mm256_packus_epi16(lo, lo);
        _mm_storel_epi64((__m128i *)(dst + i), _mm256_castsi256_si128(lo));
        _mm_storel_epi64((__m128i *)(dst + i + 8), _mm256_extracti128_si256(lo, 1));
        vi = _mm256_add_epi16(vi, _mm256_set1_epi16(16));
      }
      s0 += xblk_sz;
      s1 += xblk_sz;
      s2 += xblk_sz;
      s3 += xblk_sz;
      dst += dystride;
    }
    return;
  }
  if (xblk_sz >= 8) {
    /* a and b fit 16 bits (<= 255 << 6); the second stage is one multiply-add of the pairs
       (a, b - a) with (1 << ly, j) into 32 bits */
    const __m128i vround = _mm_set1_epi32(round);
    const __m128i cx = _mm_cvtsi32_si128(log_xblk_sz);
    const __m128i c2 = _mm_cvtsi32_si128(log_blk_sz2);
    for (j = 0; j < yblk_sz; j++) {
      const __m128i wj = _mm_set1_epi32((1 << log_yblk_sz) | (j << 16));
      __m128i vi;
      vi = _mm_setr_epi16(0, 1, 2, 3, 4, 5, 6, 7);
      for (i = 0; i < xblk_sz; i += 8) {
        __m128i a;
        __m128i b;
        __m128i p1;
        __m128i p2;
        __m128i lo;
        __m128i hi;
        a = _mm_cvtepu8_epi16(_mm_loadl_epi64((const __m128i *)(s0 + i)));
        p1 = _mm_cvtepu8_epi16(_mm_loadl_epi64((const __m128i *)(s1 + i)));
        p2 = _mm_cvtepu8_epi16(_mm_loadl_epi64((const __m128i *)(s2 + i)));
        b = _mm_cvtepu8_epi16(_mm_loadl_epi64((const __m128i *)(s3 + i)));
        a = _mm_add_epi16(_mm_sll_epi16(a, cx), _mm_mullo_epi16(_mm_sub_epi16(p1, a), vi));
        b = _mm_add_epi16(_mm_sll_epi16(b, cx), _mm_mullo_epi16(_mm_sub_epi16(p2, b), vi));
        b = _mm_sub_epi16(b, a);
        lo = _mm_madd_epi16(_mm_unpacklo_epi16(a, b), wj);
        hi = _mm_madd_epi16(_mm_unpackhi_epi16(a, b), wj);
        lo = _mm_sra_epi32(_mm_add_epi32(lo, vround), c2);
        hi = _mm_sra_epi32(_mm_add_epi32(hi, vround), c2);
        /* the result is a convex combination of four bytes: no saturation happens */
        lo = _mm_packs_epi32(lo, hi);
        _mm_storel_epi64((__m128i *)(dst + i), _mm_packus_epi16(lo, lo));
        vi = _mm_add_epi16(vi, _mm_set1_epi16(8));
      }
      s0 += xblk_sz;
      s1 += xblk_sz;
      s2 += xblk_sz;
      s3 += xblk_sz;
      dst += dystride;
    }
    return;
  }
  for (j = 0; j < yblk_sz; j++) {
    for (i = 0; i < xblk_sz; i++) {
      int32_t a;
      int32_t b;
      a = s0[i];
      b = s3[i];
      a = (a << log_xblk_sz) + (s1[i] - a)*i;
      b = (b << log_xblk_sz) + (s2[i] - b)*i;
      dst[i] = (unsigned char)(((a << log_yblk_sz) + (b - a)*j + round) >> log_blk_sz2);
    }
    s0 += xblk_sz;
    s1 += xblk_sz;
    s2 += xblk_sz;
    s3 += xblk_sz;
    dst += dystride;
  }
}

/* od_mc_blend_full_split8_c (src/mc.c:1104-1150): the blend of a block with one or two unsplit
   edges.  The weight of source k at column i of row j is linear in i for a fixed j and the row's
   start and step are linear in j (od_mc_setup_s_split, src/mc.c:1056-1102, followed by the
   function's row updates): w_k(i, j) = s0_k + j dsdj_k + i (dsdi_k + j dd_k); the pixel is
   ((a << L) + (s1 - a) w1 + (s2 - a) w2 + (s3 - a) w3 + round) >> L with a = source 0 and
   L = log2(width) + log2(height) + 1.  Eight columns per vector in 32-bit lanes: a difference is
   at most 255 and a weight at most 2 << (L - 1), so every product and the sum of the four terms
   stay far inside 31 bits for the largest block (64x64: L = 13).  Blocks narrower than eight
   columns go to the reference's function. */
void od_hipenc_mc_blend_full_split8(unsigned char *dst, int dystride, const unsigned char *src[4],
 int oc, int s, int log_xblk_sz, int log_yblk_sz) {
  int w0[4];
  int di[4];
  int dj[4];
  int dd[4];
  int l2;
  int L;
  int xblk_sz;
  int yblk_sz;
  int j;
  int k;
  if (log_xblk_sz < 3) {
    od_mc_blend_full_split8_c(dst, dystride, src, oc, s, log_xblk_sz, log_yblk_sz);
    return;
  }
  l2 = log_xblk_sz + log_yblk_sz;
  L = l2 + 1;
  xblk_sz = 1 << log_xblk_sz;
  yblk_sz = 1 << log_yblk_sz;
  /* the bilinear weights of the four corners, doubled */
  for (k = 0; k < 4; k++) w0[k] = di[k] = dj[k] = 0;
  w0[0] = 2 << l2;
  di[0] = -(2 << log_xblk_sz);
  di[1] = 2 << log_xblk_sz;
  dj[0] = -(2 << log_yblk_sz);
  dj[3] = 2 << log_yblk_sz;
  dd[0] = dd[2] = 2;
  dd[1] = dd[3] = -2;
  /* an unsplit edge next to corner oc: half of that neighbour's weight moves to oc */
  for (k = 0; k < 2; k++) {
    int c;
    if (s & (1 << k)) continue;
    c = (oc + (k == 0 ? 1 : 3)) & 3;
    w0[c] >>= 1;
    di[c] >>= 1;
    dj[c] >>= 1;
    dd[c] >>= 1;
    w0[oc] += w0[c];
    di[oc] += di[c];
    dj[oc] += dj[c];
    dd[oc] += dd[c];
  }
  {
    const __m256i lane = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
    const __m256i vround = _mm256_set1_epi32(1 << (L - 1));
    const __m128i cL = _mm_cvtsi32_si128(L);
    const unsigned char *s0;
    const unsigned char *s1;
    const unsigned char *s2;
    const unsigned char *s3;
    s0 = src[0];
    s1 = src[1];
    s2 = src[2];
    s3 = src[3];
    for (j = 0; j < yblk_sz; j++) {
      __m256i w1;
      __m256i w2;
      __m256i w3;
      __m256i st1;
      __m256i st2;
      __m256i st3;
      int i;
      w1 = _mm256_add_epi32(_mm256_set1_epi32(w0[1] + j*dj[1]), _mm256_mullo_epi32(lane, _mm256_set1_epi32(di[1] + j*dd[1])));
      w2 = _mm256_add_epi32(_mm256_set1_epi32(w0[2] + j*dj[2]), _mm256_mullo_epi32(lane, _mm256_set1_epi32(di[2] + j*dd[2])));
      w3 = _mm256_add_epi32(_mm256_set1_epi32(w0[3] + j*dj[3]), _mm256_mullo_epi32(lane, _mm256_set1_epi32(di[3] + j*dd[3])));
      st1 = _mm256_set1_epi32(8*(di[1] + j*dd[1]));
      st2 = _mm256_set1_epi32(8*(di[2] + j*dd[2]));
      st3 = _mm256_set1_epi32(8*(di[3] + j*dd[3]));
      for (i = 0; i < xblk_sz; i += 8) {
        __m256i a;
        __m256i v;
        __m128i o;
        a = _mm256_cvtepu8_epi32(_mm_loadl_epi64((const __m128i *)(s0 + i)));
        v = _mm256_add_epi32(_mm256_sll_epi32(a, cL), vround);
        v = _mm256_add_epi32(v, _mm256_mullo_epi32(
         _mm256_sub_epi32(_mm256_cvtepu8_epi32(_mm_loadl_epi64((const __m128i *)(s1 + i))), a), w1));
        v = _mm256_add_epi32(v, _mm256_mullo_epi32(
         _mm256_sub_epi32(_mm256_cvtepu8_epi32(_mm_loadl_epi64((const __m128i *)(s2 + i))), a), w2));
        v = _mm256_add_epi32(v, _mm256_mullo_epi32(
         _mm256_sub_epi32(_mm256_cvtepu8_epi32(_mm_loadl_epi64((const __m128i *)(s3 + i))), a), w3));
        v = _mm256_sra_epi32(v, cL);
        /* the reference stores (unsigned char)value: the low byte, whatever the value */
        v = _mm256_and_si256(v, _mm256_set1_epi32(255));
        o = _mm_packs_epi32(_mm256_castsi256_si128(v), _mm256_extracti128_si256(v, 1));
        _mm_storel_epi64((__m128i *)(dst + i), _mm_packus_epi16(o, o));
        w1 = _mm256_add_epi32(w1, st1);
        w2 = _mm256_add_epi32(w2, st2);
        w3 = _mm256_add_epi32(w3, st3);
      }
      s0 += xblk_sz;
      s1 += xblk_sz;
      s2 += xblk_sz;
      s3 += xblk_sz;
      dst += dystride;
    }
  }
}

/* od_mc_predict1fmv8_c (src/mc.c:94-203): the 1/8-pel predictor, separable 6-tap filters
   (OD_SUBPEL_FILTER_SET, src/mc.c:69-80), horizontal pass into a 16-bit buffer with a two-row
   top and three-row bottom apron, then the vertical pass, one rounding at the end.  Eight
   columns at a time.  The horizontal sums are formed in 16-bit lanes: the value stored,
   sum - (128 << 7), always fits (its range is [-19954, 19826] for 8-bit input and these taps)
   and two's-complement additions are exact modulo 2^16, so a sum of pair sums that passes
   32767 on the way does no harm.  Blocks narrower than eight columns go to the reference's function. */
static void mc_predict1fmv8_compute(od_state *state, unsigned char *dst, const unsigned char *src,
 int systride, int32_t mvx, int32_t mvy, int log_xblk_sz, int log_yblk_sz) {
  int16_t buff[(OD_MVBSIZE_MAX + OD_SUBPEL_BUFF_APRON_SZ)*OD_MVBSIZE_MAX] __attribute__((aligned(32)));
  const unsigned char *src_p;
  const int16_t *fx;
  const int16_t *fy;
  int16_t *buff_p;
  int mvxf;
  int mvyf;
  int xblk_sz;
  int yblk_sz;
  int i;
  int j;
  mvxf = mvx & 0x07;
  mvyf = mvy & 0x07;
  if (log_xblk_sz < 3 || !(mvxf || mvyf)) {
    od_mc_predict1fmv8_c(state, dst, src, systride, mvx, mvy, log_xblk_sz, log_yblk_sz);
    return;
  }
  xblk_sz = 1 << log_xblk_sz;
  yblk_sz = 1 << log_yblk_sz;
  src_p = src + (mvx >> 3) + (mvy >> 3)*systride;
  fx = OD_SUBPEL_FILTER_SET[mvxf];
  fy = OD_SUBPEL_FILTER_SET[mvyf];
  buff_p = buff;
  src_p -= systride*OD_SUBPEL_TOP_APRON_SZ;
  if (mvxf) {
    /* taps paired (1, 2), (3, 4), (0, 5): in every phase each pair holds at most one large
       positive tap (<= 122), so a pair sum of byte x tap products stays inside 16 bits
       (<= 255*122, >= -255*20*2) and the saturating multiply-add never saturates */
    const __m128i c12 = _mm_set1_epi16((short)(uint16_t)((fx[1] & 255) | ((unsigned)(fx[2] & 255) << 8)));
    const __m128i c34 = _mm_set1_epi16((short)(uint16_t)((fx[3] & 255) | ((unsigned)(fx[4] & 255) << 8)));
    const __m128i c05 = _mm_set1_epi16((short)(uint16_t)((fx[0] & 255) | ((unsigned)(fx[5] & 255) << 8)));
    const __m128i norm = _mm_set1_epi16((short)OD_SUBPEL_COEFF_NORMALIZE);
    for (j = -OD_SUBPEL_TOP_APRON_SZ; j < yblk_sz + OD_SUBPEL_BOTTOM_APRON_SZ; j++) {
      for (i = 0; i < xblk_sz; i += 8) {
        const unsigned char *p;
        __m128i t0;
        __m128i t1;
        __m128i t2;
        __m128i t3;
        __m128i t4;
        __m128i t5;
        __m128i sum;
        p = src_p + i - OD_SUBPEL_TOP_APRON_SZ;
        t0 = _mm_loadl_epi64((const __m128i *)p);
        t1 = _mm_loadl_epi64((const __m128i *)(p + 1));
        t2 = _mm_loadl_epi64((const __m128i *)(p + 2));
        t3 = _mm_loadl_epi64((const __m128i *)(p + 3));
        t4 = _mm_loadl_epi64((const __m128i *)(p + 4));
        t5 = _mm_loadl_epi64((const __m128i *)(p + 5));
        sum = _mm_maddubs_epi16(_mm_unpacklo_epi8(t1, t2), c12);
        sum = _mm_add_epi16(sum, _mm_maddubs_epi16(_mm_unpacklo_epi8(t3, t4), c34));
        sum = _mm_add_epi16(sum, _mm_maddubs_epi16(_mm_unpacklo_epi8(t0, t5), c05));
        _mm_store_si128((__m128i *)(buff_p + i), _mm_sub_epi16(sum, norm));
      }
      src_p += systride;
      buff_p += xblk_sz;
    }
  }
  else {
    const __m128i norm = _mm_set1_epi16((short)OD_SUBPEL_COEFF_NORMALIZE);
    for (j = -OD_SUBPEL_TOP_APRON_SZ; j < yblk_sz + OD_SUBPEL_BOTTOM_APRON_SZ; j++) {
      for (i = 0; i < xblk_sz; i += 8) {
        __m128i v;
        v = _mm_cvtepu8_epi16(_mm_loadl_epi64((const __m128i *)(src_p + i)));
        _mm_store_si128((__m128i *)(buff_p + i),
         _mm_sub_epi16(_mm_slli_epi16(v, OD_SUBPEL_COEFF_SCALE), norm));
      }
      src_p += systride;
      buff_p += xblk_sz;
    }
  }
  buff_p = buff + xblk_sz*OD_SUBPEL_TOP_APRON_SZ;
  if (mvyf) {
    const __m128i c01 = _mm_set1_epi32((int)((uint32_t)(fy[0] & 0xffff) | ((uint32_t)(fy[1] & 0xffff) << 16)));
    const __m128i c23 = _mm_set1_epi32((int)((uint32_t)(fy[2] & 0xffff) | ((uint32_t)(fy[3] & 0xffff) << 16)));
    const __m128i c45 = _mm_set1_epi32((int)((uint32_t)(fy[4] & 0xffff) | ((uint32_t)(fy[5] & 0xffff) << 16)));
    const __m128i rnd = _mm_set1_epi32(OD_SUBPEL_RND_OFFSET3);
    for (j = 0; j < yblk_sz; j++) {
      for (i = 0; i < xblk_sz; i += 8) {
        const int16_t *p;
        __m128i r0;
        __m128i r1;
        __m128i lo;
        __m128i hi;
        p = buff_p + i - OD_SUBPEL_TOP_APRON_SZ*xblk_sz;
        r0 = _mm_load_si128((const __m128i *)p);
        r1 = _mm_load_si128((const __m128i *)(p + xblk_sz));
        lo = _mm_madd_epi16(_mm_unpacklo_epi16(r0, r1), c01);
        hi = _mm_madd_epi16(_mm_unpackhi_epi16(r0, r1), c01);
        r0 = _mm_load_si128((const __m128i *)(p + 2*xblk_sz));
        r1 = _mm_load_si128((const __m128i *)(p + 3*xblk_sz));
        lo = _mm_add_epi32(lo, _mm_madd_epi16(_mm_unpacklo_epi16(r0, r1), c23));
        hi = _mm_add_epi32(hi, _mm_madd_epi16(_mm_unpackhi_epi16(r0, r1), c23));
        r0 = _mm_load_si128((const __m128i *)(p + 4*xblk_sz));
        r1 = _mm_load_si128((const __m128i *)(p + 5*xblk_sz));
        lo = _mm_add_epi32(lo, _mm_madd_epi16(_mm_unpacklo_epi16(r0, r1), c45));
        hi = _mm_add_epi32(hi, _mm_madd_epi16(_mm_unpackhi_epi16(r0, r1), c45));
        lo = _mm_srai_epi32(_mm_add_epi32(lo, rnd), OD_SUBPEL_COEFF_SCALE2);
        hi = _mm_srai_epi32(_mm_add_epi32(hi, rnd), OD_SUBPEL_COEFF_SCALE2);
        /* OD_CLAMP255: saturating packs, 32 -> 16 (signed) -> 8 (unsigned) */
        lo = _mm_packs_epi32(lo, hi);
        _mm_storel_epi64((__m128i *)(dst + i), _mm_packus_epi16(lo, lo));
      }
      buff_p += xblk_sz;
      dst += xblk_sz;
    }
  }
  else {
    for (j = 0; j < yblk_sz; j++) {
      for (i = 0; i < xblk_sz; i += 8) {
        __m128i v;
        /* buff + RND_OFFSET4 <= 19826 + 16448 passes 32767: widen before adding */
        __m256i w;
        __m128i o;
        v = _mm_load_si128((const __m128i *)(buff_p + i));
        w = _mm256_add_epi32(_mm256_cvtepi16_epi32(v), _mm256_set1_epi32(OD_SUBPEL_RND_OFFSET4));
        w = _mm256_srai_epi32(w, OD_SUBPEL_COEFF_SCALE);
        o = _mm_packs_epi32(_mm256_castsi256_si128(w), _mm256_extracti128_si256(w, 1));
        _mm_storel_epi64((__m128i *)(dst + i), _mm_packus_epi16(o, o));
      }
      buff_p += xblk_sz;
      dst += xblk_sz;
    }
  }
}

/* The search asks for the same prediction again and again: a dynamic-programming step over
   two neighbouring grid points tries every pair of their candidate vectors, and each of the
   (up to four) blocks touched is rebuilt from its four corner vectors every time although
   only one or two of them changed (od_mv_dp_*, src/mcenc.c).  One corner's prediction is a
   pure function of (reference samples, vector, block shape), and reference frames are not
   written while a frame is being searched, so recent predictions are kept per thread -
   direct mapped, keyed by (source address, stride, vector, shape), emptied at every frame
   start (od_hipenc_mc_cache_flush) - and a repeat costs a copy instead of two filter passes. */
#define MC_CACHE_LOG (10)
#define MC_CACHE_N (1 << MC_CACHE_LOG)
#define MC_CACHE_BLK (OD_MVBSIZE_MAX*OD_MVBSIZE_MAX)
typedef struct mc_key {
  const unsigned char *src;
  int32_t mvx;
  int32_t mvy;
  int32_t shape;              /* systride << 8 | log_xblk_sz << 4 | log_yblk_sz */
  uint32_t gen;
} mc_key;
static __thread mc_key *mc_keys;
static __thread unsigned char *mc_data;
static __thread uint32_t mc_gen = 1;
static __thread int64_t mc_hits;
static __thread int64_t mc_misses;

void od_hipenc_mc_cache_flush(void) {
  mc_gen++;
  if (mc_gen == 0) {
    if (mc_keys != NULL) memset(mc_keys, 0, sizeof(mc_key)*MC_CACHE_N);
    mc_gen = 1;
  }
}

/* called by a worker thread before it exits */
void od_hipenc_mc_cache_free(void) {
  free(mc_keys);
  free(mc_data);
  mc_keys = NULL;
  mc_data = NULL;
}

void od_hipenc_mc_cache_stats(int64_t *hits, int64_t *misses) {
  *hits = mc_hits;
  *misses = mc_misses;
}

void od_hipenc_mc_predict1fmv8(od_state *state, unsigned char *dst, const unsigned char *src,
 int systride, int32_t mvx, int32_t mvy, int log_xblk_sz, int log_yblk_sz) {
  mc_key *e;
  unsigned char *d;
  uint64_t h;
  int32_t shape;
  size_t bytes;
  if (mc_keys == NULL) {
    mc_keys = (mc_key *)calloc(MC_CACHE_N, sizeof(mc_key));
    mc_data = (unsigned char *)malloc((size_t)MC_CACHE_N*MC_CACHE_BLK);
    if (mc_keys == NULL || mc_data == NULL) {
      free(mc_keys);
      free(mc_data);
      mc_keys = NULL;
      mc_data = NULL;
      mc_predict1fmv8_compute(state, dst, src, systride, mvx, mvy, log_xblk_sz, log_yblk_sz);
      return;
    }
  }
  shape = systride << 8 | log_xblk_sz << 4 | log_yblk_sz;
  h = (uint64_t)(uintptr_t)src*0x9E3779B97F4A7C15ull;
  h ^= ((uint64_t)(uint32_t)mvx*0x85EBCA6Bu) ^ ((uint64_t)(uint32_t)mvy*0xC2B2AE35u << 13) ^ (uint64_t)shape << 40;
  h ^= h >> 29;
  e = mc_keys + ((h*0x9E3779B97F4A7C15ull) >> (64 - MC_CACHE_LOG));
  d = mc_data + (size_t)(e - mc_keys)*MC_CACHE_BLK;
  bytes = (size_t)1 << (log_xblk_sz + log_yblk_sz);
  if (e->gen == mc_gen && e->src == src && e->mvx == mvx && e->mvy == mvy && e->shape == shape) {
    memcpy(dst, d, bytes);
    mc_hits++;
    return;
  }
  mc_predict1fmv8_compute(state, dst, src, systride, mvx, mvy, log_xblk_sz, log_yblk_sz);
  memcpy(d, dst, bytes);
  e->src = src;
  e->mvx = mvx;
  e->mvy = mvy;
  e->shape = shape;
  e->gen = mc_gen;
  mc_misses++;
}

/* test entries: the two leaves and the reference's, same arguments */
void od_hipenc_mc_leaves_test(int which, unsigned char *dst, int dystride, const unsigned char *s0,
 const unsigned char *s1, const unsigned char *s2, const unsigned char *s3, int systride,
 int32_t mvx, int32_t mvy, int log_xblk_sz, int log_yblk_sz) {
  const unsigned char *src[4];
  src[0] = s0;
  src[1] = s1;
  src[2] = s2;
  src[3] = s3;
  switch (which) {
    case 0: od_hipenc_mc_blend_full8(dst, dystride, src, log_xblk_sz, log_yblk_sz); break;
    case 1: od_mc_blend_full8_c(dst, dystride, src, log_xblk_sz, log_yblk_sz); break;
    case 2:
      /* a fresh plane may sit at the address of the last one: new frame, empty cache */
      od_hipenc_mc_cache_flush();
      od_hipenc_mc_predict1fmv8(NULL, dst, s0, systride, mvx, mvy, log_xblk_sz, log_yblk_sz);
      break;
    case 4: od_hipenc_mc_predict1fmv8(NULL, dst, s0, systride, mvx, mvy, log_xblk_sz, log_yblk_sz); break;
    case 5: od_hipenc_mc_blend_full_split8(dst, dystride, src, mvx, mvy, log_xblk_sz, log_yblk_sz); break;   /* mvx: oc, mvy: s */
    case 6: od_mc_blend_full_split8_c(dst, dystride, src, mvx, mvy, log_xblk_sz, log_yblk_sz); break;
    default: od_mc_predict1fmv8_c(NULL, dst, s0, systride, mvx, mvy, log_xblk_sz, log_yblk_sz); break;
  }
}
