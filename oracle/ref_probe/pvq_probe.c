/* Probe translation unit (TEST INFRASTRUCTURE, dev container only).
 * Reaches the reference's *static* PVQ functions the same way the reference's
 * own src/tests/test_coef_coder.c:25-34 does: by #including the .c file (found
 * through -I/root/reference/src; no reference text lives in this repo) and
 * exporting thin wrappers.  Built into oracle/_ref/pvq_probe.so by
 * oracle/Makefile; used by tools/gen_golden.py and tests/test_oracle_vs_ref.py. */
#include "pvq_encoder.c"

double probe_pvq_search_rdo_double(const double *xcoeff, int n, int k,
 od_coeff *ypulse, double g2) {
  return pvq_search_rdo_double(xcoeff, n, k, ypulse, g2);
}

/* pvq_theta with a freshly reset adaptation context (keyframe state at the
   start of a frame, reference src/state.c:595-639). */
int probe_pvq_theta(od_coeff *out, od_coeff *x0, od_coeff *r0, int n, int q0,
 od_coeff *y, int *itheta, int *max_theta, int *vk, double beta,
 double *skip_diff, int robust, int is_keyframe, int pli, int bs,
 const int16_t *qm, const int16_t *qm_inv) {
  static od_adapt_ctx adapt;
  od_adapt_pvq_ctx_reset(&adapt.pvq, is_keyframe);
  return pvq_theta(out, x0, r0, n, q0, y, itheta, max_theta, vk, beta,
   skip_diff, robust, is_keyframe, pli, &adapt, bs, qm, qm_inv);
}

double probe_pvq_rate_reset(int qg, int icgr, int theta, int ts,
 const od_coeff *y0, int k, int n, int is_keyframe, int pli, int bs) {
  static od_adapt_ctx adapt;
  od_adapt_pvq_ctx_reset(&adapt.pvq, is_keyframe);
  return od_pvq_rate(qg, icgr, theta, ts, &adapt, y0, k, n, is_keyframe, pli, bs);
}
