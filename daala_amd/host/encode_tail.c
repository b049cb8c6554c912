/* Appended to the reference's src/encode.c in the HIP build (Makefile: sed | cat | gcc):
 * one exported wrapper around the static od_img_copy_pad (src/encode.c:1728), so that the
 * glue can hand the device exactly the padded input planes daala_encode_img_in() codes. */
void od_hipenc_copy_pad(daala_enc_ctx *enc, od_img *img) {
  int keep;
  keep = enc->in_buff_ptr;
  enc->in_buff_ptr = 0;
  od_img_copy_pad(enc, img);
  enc->in_buff_ptr = keep;
}

/* od_compute_dist as the encoder calls it: the two calls of the deringing on/off decision
 * (src/encode.c:2634-2635) are answered from the device pass that deringed the frame
 * (od_hipenc_dist_hook, hip_enc_glue.c); every other call is the reference's own function. */
int od_hipenc_dist_hook(daala_enc_ctx *enc, const od_coeff *x, const od_coeff *y, int n, int bs,
 double *dist, double (*cpu)(daala_enc_ctx *, od_coeff *, od_coeff *, int, int));

static double od_compute_dist(daala_enc_ctx *enc, od_coeff *x, od_coeff *y, int n, int bs) {
  double d;
  if (od_hipenc_dist_hook(enc, x, y, n, bs, &d, od_compute_dist_cpu)) return d;
  return od_compute_dist_cpu(enc, x, y, n, bs);
}
