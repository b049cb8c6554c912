/* Force-included (-include) ahead of the reference's src/laplace_encoder.c for its SECOND
 * compile in the HIP build (Makefile: build/obj/laplace_rate.o): the rate-only Laplace coder of
 * hip_pvq_host.c's od_pvq_rate.  The entropy coder's header is read first (its include guard
 * keeps the prototypes intact), then the three coder calls laplace_encoder.c makes
 * (:83, :86, :91, :127, :135, :178) are bound to the (rng, bit count) recurrence of hip_rc.h;
 * the od_ec_enc pointer the functions pass along is really a hip_rc.  The functions themselves
 * are renamed by -D in the recipe (laplace_encode_vector -> od_hip_rate_laplace_vector, ...) so
 * that they live beside the real coder of build/obj/laplace_encoder.o. */
#include "entenc.h"
#include "hip_rc.h"
#define od_ec_encode_cdf_unscaled(enc, s, cdf, nsyms) rc_cdf_unscaled((hip_rc *)(enc), s, cdf, 0, nsyms)
#define od_ec_encode_cdf_q15(enc, s, cdf, nsyms) rc_cdf_q15((hip_rc *)(enc), s, cdf)
#define od_ec_enc_bits(enc, fl, ftb) ((void)(fl), (void)(((hip_rc *)(enc))->nbits += (int)(ftb)))
