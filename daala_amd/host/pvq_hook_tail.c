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
