/* Appended to the reference's decode.c by the integration build (Makefile).  On a P frame whose
   PVQ synthesis runs on the device (hip_dec_glue.c) nothing of the block decoder reads the
   prediction's coefficients: the copy out of the mdtmp plane is not made. */
int od_hipdec_pred_from_device(void);
static void od_decode_compute_pred(daala_dec_ctx *dec, od_mb_dec_ctx *ctx, od_coeff *pred,
 const od_coeff *d, int bs, int pli, int bx, int by) {
  if (od_hipdec_pred_from_device()) return;
  od_decode_compute_pred_cpu(dec, ctx, pred, d, bs, pli, bx, by);
}
