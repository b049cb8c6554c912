/* Appended to the reference's src/pvq_encoder.c in the HIP build (integration build recipe):
 * exports the kept C search so that hip_enc_glue.c can run it for the calls the
 * device cannot answer (with-reference and chroma bands) and in check mode. */
double od_ref_pvq_search_rdo_double_cpu(const double *xcoeff, int n, int k,
 od_coeff *ypulse, double g2) {
  return pvq_search_rdo_double_cpu(xcoeff, n, k, ypulse, g2);
}
