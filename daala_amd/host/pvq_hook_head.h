/* Force-included (-include) ahead of the reference's src/pvq_encoder.c when it is
 * compiled for the HIP build (integration build recipe): the build recipe keeps the
 * reference's definition of pvq_search_rdo_double (:121) under the name
 * pvq_search_rdo_double_cpu, so its two call sites (:426, :463) bind to this
 * external symbol instead - implemented in hip_enc_glue.c. */
double pvq_search_rdo_double(const double *xcoeff, int n, int k, int *ypulse,
 double g2);

/* The same for pvq_theta (:311; kept as pvq_theta_cpu): od_pvq_encode's call site (:713) binds
 * to hip_pvq_host.c's band decision, which consumes the device feeds; and for the two
 * checkpoint calls of od_pvq_encode (:718, :796), bound through -D to versions that save
 * what the function can modify instead of the whole adaptation context. */
#include "encint.h"
int pvq_theta(od_coeff *out, od_coeff *x0, od_coeff *r0, int n, int q0, od_coeff *y,
 int *itheta, int *max_theta, int *vk, double beta, double *skip_diff, int robust,
 int is_keyframe, int pli, const od_adapt_ctx *adapt, int bs, const int16_t *qm,
 const int16_t *qm_inv);
void od_hip_pvq_checkpoint(const daala_enc_ctx *enc, od_rollback_buffer *rbuf);
void od_hip_pvq_rollback(daala_enc_ctx *enc, const od_rollback_buffer *rbuf);
