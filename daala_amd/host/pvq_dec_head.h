/* Prepended (-include) to the reference's pvq_decoder.c by the integration build: the two
   leaf calls of od_pvq_decode that the decoder glue binds (hip_dec_glue.c, "P frames: PVQ
   synthesis on the device"). */
#include "pvq.h"
void pvq_synthesis(od_coeff *xcoeff, od_coeff *ypulse, od_coeff *ref, int n, double gr,
 int noref, double g, double theta, const int16_t *qm, const int16_t *qm_inv);
double od_hipdec_pvq_compute_gain(od_coeff *x, int n, int q0, double *g, double beta,
 const int16_t *qm);
