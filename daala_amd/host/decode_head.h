/* Prepended (-include) to the reference's decode.c by the integration build: the block
   decoder's static od_decode_compute_pred keeps its definition as *_cpu (renamed in the piped
   source) and its call site binds to the definition in decode_tail.c. */
#include "decint.h"
typedef struct od_mb_dec_ctx od_mb_dec_ctx;
static void od_decode_compute_pred(daala_dec_ctx *dec, od_mb_dec_ctx *ctx, od_coeff *pred,
 const od_coeff *d, int bs, int pli, int bx, int by);
