/* Translation unit that stands in for the reference's src/encode.c in the HIP
 * build: the file itself (found through -I<reference>/src, the technique of the
 * reference's own src/tests/test_coef_coder.c:25-34) plus one exported wrapper
 * around the static od_img_copy_pad (src/encode.c:1728), so that the glue can
 * hand the device exactly the padded input planes daala_encode_img_in() codes. */
#include "encode.c"

void od_hipenc_copy_pad(daala_enc_ctx *enc, od_img *img) {
  int keep;
  keep = enc->in_buff_ptr;
  enc->in_buff_ptr = 0;
  od_img_copy_pad(enc, img);
  enc->in_buff_ptr = keep;
}
