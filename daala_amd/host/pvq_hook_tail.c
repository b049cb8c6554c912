/* Appended to the reference's src/pvq_encoder.c in the HIP build (integration build recipe):
 * exports the kept C search so that hip_enc_glue.c can run it for the calls the
 * device cannot answer (with-reference and chroma bands) and in check mode. */
double od_ref_pvq_search_rdo_double_cpu(const double *xcoeff, int n, int k,
 od_coeff *ypulse, double g2) {
  return pvq_search_rdo_double_cpu(xcoeff, n, k, ypulse, g2);
}

/* hip_pvq_host.c codes the partitions with the reference's own (static) routine, and its
 * check mode prices candidates with the reference's od_pvq_rate. */
void od_ref_pvq_encode_partition(od_ec_enc *ec, int qg, int theta, int max_theta,
 const od_coeff *in, int n, int k, generic_encoder model[3], od_adapt_ctx *adapt,
 int *exg, int *ext, int nodesync, int cdf_ctx, int is_keyframe, int code_skip,
 int skip_rest, int bs) {
  pvq_encode_partition(ec, qg, theta, max_theta, in, n, k, model, adapt, exg, ext, nodesync,
   cdf_ctx, is_keyframe, code_skip, skip_rest, bs);
}

double od_ref_pvq_rate(int qg, int icgr, int theta, int ts, const od_adapt_ctx *adapt,
 const od_coeff *y0, int k, int n, int is_keyframe, int pli, int bs) {
  return od_pvq_rate(qg, icgr, theta, ts, adapt, y0, k, n, is_keyframe, pli, bs);
}

/* the reference's own band decision, for HIPENC_HOST_PVQ=0 */
int od_ref_pvq_theta(od_coeff *out, od_coeff *x0, od_coeff *r0, int n, int q0, od_coeff *y,
 int *itheta, int *max_theta, int *vk, double beta, double *skip_diff, int robust,
 int is_keyframe, int pli, const od_adapt_ctx *adapt, int bs, const int16_t *qm,
 const int16_t *qm_inv) {
  return pvq_theta_cpu(out, x0, r0, n, q0, y, itheta, max_theta, vk, beta, skip_diff, robust,
   is_keyframe, pli, adapt, bs, qm, qm_inv);
}
