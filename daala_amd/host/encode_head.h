/* Force-included (-include) ahead of the reference's src/encode.c in the HIP build: the build
 * recipe keeps the reference's static od_compute_dist (:1032) under the name
 * od_compute_dist_cpu, so its call sites (:1255, :1258, :1633-1634, :2634-2635) bind to the
 * definition appended by encode_tail.c.  Only a declaration: daala_enc_ctx and od_coeff are
 * declared by the headers encode.c includes first, hence the struct tag / int32_t spelling. */
#include <stdint.h>
struct daala_enc_ctx;
static double od_compute_dist(struct daala_enc_ctx *enc, int32_t *x, int32_t *y, int n, int bs);
