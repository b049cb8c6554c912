/* hip_rc.h - rate-only range coder (own code; not part of the C-ABI).
 * Shared by hip_pvq_host.c and by the rate-only build of the reference's Laplace coder
 * (laplace_rate_head.h, Makefile rule build/obj/laplace_rate.o). */
#ifndef HIP_RC_H
#define HIP_RC_H

#include <stdint.h>
#include "odintrin.h"

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

/* od_ec_encode_cdf_q15 (src/entenc.c:368-375) */
static inline void rc_cdf_q15(hip_rc *c, int s, const uint16_t *cdf) {
  rc_encode_q15(c, s > 0 ? cdf[s - 1] : 0, cdf[s]);
}

#endif
