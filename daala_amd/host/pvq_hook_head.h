/* Force-included (-include) ahead of the reference's src/pvq_encoder.c when it is
 * compiled for the HIP build (integration build recipe): the build recipe keeps the
 * reference's definition of pvq_search_rdo_double (:121) under the name
 * pvq_search_rdo_double_cpu, so its two call sites (:426, :463) bind to this
 * external symbol instead - implemented in hip_enc_glue.c. */
double pvq_search_rdo_double(const double *xcoeff, int n, int k, int *ypulse,
 double g2);
