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
