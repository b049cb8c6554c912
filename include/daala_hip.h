/* daala_hip.h - C ABI of the MI355X (gfx950) back-end for Daala's per-block
 * transform + quantisation hot path.
 *
 * Everything here is plain C: pointers, ints, no C++/torch types.  The entry
 * points are what the reference's own code would bind (INTEGRATION.md shows the
 * patch): each one cites the reference interface it replaces.
 *
 *   od_coeff          int32_t                           src/filter.h:31
 *   od_dct_func_2d    (out, out_stride, in, in_stride)  src/dct.h:61-62
 *
 * Error convention follows the reference (include/daala/codec.h:89-103):
 * 0 = success, negative = failure; od_hip_last_error() gives the text.
 * There is NO CPU fallback: without a usable HIP device every compute entry
 * point fails with OD_HIP_ENODEV.
 *
 * Threads: the entry points that take HOST pointers for their data (sections 1 and 2, the
 * od_hip_bin_* drop-ins od_hip_vtbl_fill installs, the *_vectors / *_blocks calls) share
 * one set of device scratch buffers and the null stream of the calling thread's current
 * device: they are single-threaded, one call at a time per process.  Context objects
 * (od_hip_ctx, od_hip_enc_feed, od_hip_dering, od_hip_comm) carry their own device id and
 * streams; one thread at a time per object, different objects concurrently.  The
 * exceptions, safe from several threads on one object: od_hip_upload_planes (different
 * slots), od_hip_enc_feed_compand (different slots), od_hip_enc_feed_view.
 *
 * EXPORT MAP - who binds what (kept current by tests/test_cabi_cpu.py::test_export_map):
 *   SEAM 1, the per-call vtable (installed into a live reference encoder by
 *     tests/test_gpu_vtable_seam.py): od_hip_bin_{f,i}dct{4x4,8x8,16x16,32x32}, od_hip_vtbl_fill.
 *   SEAM 2, the batched frame seam - bound by the integration library (daala_amd/host) behind
 *     daala_encode_img_in() / daala_decode_packet_in(): od_hip_ctx_*, od_hip_upload_planes /
 *     _coeffs / set_bsize / set_decode_info, od_hip_forward_pyramid, od_hip_forward_haar is the
 *     feed's, od_hip_inverse_haar, od_hip_decode_tail, od_hip_download_*, od_hip_enc_feed_*
 *     (4b), od_hip_dering_* (4c), od_hip_pfeed_* (4d), od_hip_dsynth_* (4e), od_hip_mc_create /
 *     destroy / set_ref / set_ref_ctx / set_src / predict / predict_ctx / sad_items / bma_windows, od_hip_pvq_compand,
 *     od_hip_host_register / _unregister, od_hip_last_error, od_hip_device_count.
 *   DRIVER - the device-only path of bench.py and the multi-GPU drivers: od_hip_forward_known,
 *     od_hip_inverse, od_hip_pvq_gains / _compand_level / _search / _noref_search / _nblocks /
 *     _download / _stats, od_hip_enc_feed_run / _refresh, od_hip_set_strip, od_hip_comm_*,
 *     od_hip_gather_strips, od_hip_strip_bytes / _export / _import, od_hip_sync, od_hip_device_sync,
 *     od_hip_timing_reset / _get, od_hip_calibrate_traffic, od_hip_version.
 *   VECTOR - dense batches on host memory, the form the reference's own tools and unit tests
 *     use (dcttest, filter.c -DTEST, test_coef_coder); they exist for PARITY TESTS and are bound
 *     by no seam: od_hip_fdct_blocks, od_hip_idct_blocks, od_hip_haar_blocks,
 *     od_hip_filter4_vectors, od_hip_filter_vectors, od_hip_resample_luma(_420),
 *     od_hip_hv_intra_pred_blocks, od_hip_coding_order_blocks, od_hip_compute_dist_blocks,
 *     od_hip_band_offsets, od_hip_pvq_search_vectors, od_hip_pvq_theta_vectors,
 *     od_hip_pvq_synthesis_noref, od_hip_pvq_synthesis_vectors, od_hip_mc_predict_blocks,
 *     od_hip_libm_probe.
 */
#ifndef DAALA_HIP_H
#define DAALA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t od_coeff;

#define OD_HIP_SUCCESS (0)
#define OD_HIP_EFAULT (-1)    /* bad pointer / argument (OD_EFAULT) */
#define OD_HIP_EINVAL (-10)   /* invalid geometry or state (OD_EINVAL) */
#define OD_HIP_ENODEV (-30)   /* no HIP device / HIP runtime error */
#define OD_HIP_ENOSPC (-11)   /* caller's output buffer is too small (nothing was written) */

#define OD_HIP_NPLANES_MAX (4)
#define OD_HIP_NBSIZES (4)    /* 4x4 .. 32x32, OD_NBSIZES src/internal.h:55 */

const char *od_hip_last_error(void);
int od_hip_device_count(void);
const char *od_hip_version(void);

/* ---------------------------------------------------------------------------
 * 1. Per-call drop-ins for the kernel vtable (struct od_state_opt_vtbl,
 *    src/state.h:106-126; C versions OD_FDCT_2D_C/OD_IDCT_2D_C src/dct.c:42-56).
 *    Host pointers, strides in elements, out may alias in.  Each call is a
 *    synchronous H2D + kernel + D2H round trip: correct, but only meant for
 *    OD_CHECKASM-style parity checking - the batched entry points below are the
 *    fast path.  They abort() on a HIP failure because the vtable signature
 *    returns void (the reference aborts on internal failures the same way,
 *    src/internal.h:128-158). */
typedef void (*od_dct_func_2d)(od_coeff *out, int out_stride,
 const od_coeff *in, int in_stride);

void od_hip_bin_fdct4x4(od_coeff *y, int ystride, const od_coeff *x, int xstride);
void od_hip_bin_fdct8x8(od_coeff *y, int ystride, const od_coeff *x, int xstride);
void od_hip_bin_fdct16x16(od_coeff *y, int ystride, const od_coeff *x, int xstride);
void od_hip_bin_fdct32x32(od_coeff *y, int ystride, const od_coeff *x, int xstride);
void od_hip_bin_idct4x4(od_coeff *x, int xstride, const od_coeff *y, int ystride);
void od_hip_bin_idct8x8(od_coeff *x, int xstride, const od_coeff *y, int ystride);
void od_hip_bin_idct16x16(od_coeff *x, int xstride, const od_coeff *y, int ystride);
void od_hip_bin_idct32x32(od_coeff *x, int xstride, const od_coeff *y, int ystride);

/* Overwrites entries 0..3 of the two vtable arrays (entry 4, the 64-point
 * float transform, is unreachable - src/encode.c:1098 - and left alone).
 * Replaces the body of od_state_opt_vtbl_init_x86 (src/x86/x86state.c:39-96)
 * for the fdct_2d / idct_2d members.  Returns 0 or OD_HIP_ENODEV. */
int od_hip_vtbl_fill(od_dct_func_2d fdct_2d[OD_HIP_NBSIZES + 1],
 od_dct_func_2d idct_2d[OD_HIP_NBSIZES + 1]);

/* ---------------------------------------------------------------------------
 * 2. Batched block transforms on host memory: nblocks dense n x n blocks
 *    (n = 4 << bs), block b at blocks[b*n*n].  Same arithmetic as
 *    od_bin_fdctNxN / od_bin_idctNxN (src/dct.c:137,144,335,342,774,782,2028,
 *    2036), and for `haar` od_haar / od_haar_inv (src/dct.c:1960-2026, lossless
 *    frames).  out may alias in. */
int od_hip_fdct_blocks(int bs, od_coeff *out, const od_coeff *in, int nblocks);
int od_hip_idct_blocks(int bs, od_coeff *out, const od_coeff *in, int nblocks);
int od_hip_haar_blocks(int bs, int inverse, od_coeff *out, const od_coeff *in,
 int nblocks);

/* nvec 4-sample vectors through od_pre_filter4 / od_post_filter4
 * (src/filter.c:174-249). */
int od_hip_filter4_vectors(int inverse, od_coeff *out, const od_coeff *in,
 int nvec);

/* The n-point variants od_pre_filter{4,8,16,32} / od_post_filter{4,8,16,32}
 * (src/filter.c:174-249, :306-440, :546-808, :879-1380), nvec vectors of n samples.
 * Only n = 4 is reachable in the codec (OD_FILT_SIZE() == 0, src/filter.h:99); the
 * larger ones are what the reference's dcttest / filter tools exercise. */
int od_hip_filter_vectors(int n, int inverse, od_coeff *out, const od_coeff *in,
 int nvec);

/* CfL luma resample, od_resample_luma_coeffs (src/intra.c:72-109) built on
 * od_tf_up_hv_lp (src/tf.c:82-108), 4:2:0 only: for each of nblk blocks reads
 * the luma coefficients at luma + luma_off[b] (stride lstride) and writes an
 * n x n dense predictor (n = 4 << bs) at pred + b*n*n. chroma_bs as in the
 * reference: 0 => TF-merge of the 2x2 group of 4x4 luma blocks. */
int od_hip_resample_luma_420(od_coeff *pred, const od_coeff *luma, size_t luma_len,
 int lstride, const int32_t *luma_off, int nblk, int bs, int chroma_bs);
/* The same for any chroma decimation (xdec, ydec in {0,1}): od_tf_up_hv_lp for 4:2:0,
 * od_tf_up_h_lp (src/tf.c:38-58) for horizontally decimated chroma (4:2:2),
 * od_tf_up_v_lp (:60-80) for vertically decimated chroma (4:4:0), a copy for 4:4:4. */
int od_hip_resample_luma(od_coeff *pred, const od_coeff *luma, size_t luma_len,
 int lstride, const int32_t *luma_off, int nblk, int bs, int chroma_bs, int xdec,
 int ydec);

/* ---------------------------------------------------------------------------
 * 3. Device-resident frame pipeline.  A context owns the HBM buffers of
 *    `nslots` frames of one geometry (the device mirror of what od_state_init
 *    allocates per context: ctmp/dtmp planes, src/state.c:395-455).
 *    Planes are dense, stride = padded plane width (frame padded to a multiple
 *    of 32 here; the reference pads to 64, src/state.c:372-375 - pass the
 *    padded frame size you use).  Only 8-bit input, 4:4:4 or 4:2:0. */
typedef struct od_hip_ctx od_hip_ctx;

typedef struct od_hip_geometry {
  int pic_width;            /* daala_info.pic_width  (luma, unpadded) */
  int pic_height;
  int frame_width;          /* padded luma size, multiple of 32 */
  int frame_height;
  int nplanes;              /* 1..3 */
  int xdec[OD_HIP_NPLANES_MAX];   /* 0 or 1; ydec == xdec (src/encode.c:1295) */
  int nslots;               /* frames resident at once */
} od_hip_geometry;

od_hip_ctx *od_hip_ctx_create(int device, const od_hip_geometry *geo);
void od_hip_ctx_destroy(od_hip_ctx *ctx);

/* Input: padded 8-bit planes (what od_img_copy_pad leaves in enc->input_img,
 * src/encode.c:2829-2841).  H2D copy into slot. */
int od_hip_upload_planes(od_hip_ctx *ctx, int slot,
 const unsigned char *const planes[], const int ystride[]);

/* Forward PYRAMID for block-size RDO, slots [slot0, slot0+nslots): fuses
 * od_ref_plane_to_coeff (src/state.c:1252), od_apply_prefilter_frame_sbs
 * (src/filter.c:1556), od_prefilter_split (src/filter.c:1486) of every ancestor
 * and od_bin_fdctNxN of every block of every size - the inputs
 * od_block_encode sees at each level of od_encode_recursive's RDO
 * (src/encode.c:1554-1592).  Level k of plane pli holds the transform of every
 * (SB >> k)-sized block, SB = 32 >> xdec[pli]; luma has 4 levels, 4:2:0 chroma
 * 3.  Results stay in HBM. */
int od_hip_forward_pyramid(od_hip_ctx *ctx, int slot0, int nslots);

/* Forward with KNOWN block sizes = od_compute_dcts over the frame
 * (src/encode.c:1286-1343, incl. the keyframe Haar merge of child DCs).
 * bsize: luma block-size map, 1 byte per 8x8 (values 0..3), (frame_height/8)
 * rows of `bstride` bytes, quadtree-consistent (src/state.h:207-224). */
int od_hip_set_bsize(od_hip_ctx *ctx, int slot, const unsigned char *bsize,
 int bstride);
int od_hip_forward_known(od_hip_ctx *ctx, int slot0, int nslots, int keyframe);

/* Inverse with known block sizes: od_bin_idctNxN of every coded block,
 * od_postfilter_split back up the quadtree (src/decode.c:843-866,
 * src/filter.c:1512), od_apply_postfilter_frame_sbs (src/filter.c:1588) and
 * od_coeff_to_ref_plane (src/state.c:1320).  Input: the slot's coefficient
 * planes (as written by od_hip_forward_known or od_hip_upload_coeffs);
 * output: 8-bit reconstruction planes in HBM. */
int od_hip_inverse(od_hip_ctx *ctx, int slot0, int nslots);

/* Lossless frames (quantizer 0; src/encode.c:3002,3090-3092, src/decode.c:785,1036):
 * no lapping, no DCT, coefficient shift 0 (od_ref_buf_to_coeff / od_coeff_to_ref_buf
 * with lossless_p, src/state.c:1209,1274) and od_haar / od_haar_inv
 * (src/dct.c:1960-2026) of every whole superblock - 32x32 luma, 16x16 4:2:0 chroma.
 * Forward: the slot's input planes -> coefficient planes; inverse: coefficient planes
 * -> 8-bit reconstruction planes.  inverse(forward(x)) == x. */
int od_hip_forward_haar(od_hip_ctx *ctx, int slot0, int nslots);
int od_hip_inverse_haar(od_hip_ctx *ctx, int slot0, int nslots);

/* Decoder reconstruction, the whole pixel-domain stage of od_decode_coefficients
 * after the symbol parse (src/decode.c:1010-1155): iDCT + split post-filters +
 * frame post-filter, then per 32x32 deringing superblock od_dering
 * (src/filter.c:1835) where the decoded flag says so, od_bilinear_smooth
 * (src/filter.c:1952) on keyframe 32x32 blocks, and the 8-bit clamp.
 * od_hip_set_decode_info uploads what the parse produced for the slot:
 * dering_flags (1 byte per superblock, raster, state.dering_flags) and the
 * skip maps state.bskip[pli] (1 byte per 4x4, rows of skip_stride bytes).
 * threshold[pli] = (int)pow(quantizer[pli], 0.84182) (src/filter.c:1878, computed
 * by the host's libm), quantizer[pli] = state.quantizer[pli]. */
int od_hip_set_decode_info(od_hip_ctx *ctx, int slot, const unsigned char *dering_flags,
 const unsigned char *const bskip[], int skip_stride);
int od_hip_decode_tail(od_hip_ctx *ctx, int slot0, int nslots, const int32_t *threshold,
 const int32_t *quantizer, int is_keyframe);

/* Host access to device-resident results (row-dense, stride = plane width). */
int od_hip_download_level(od_hip_ctx *ctx, int slot, int pli, int level,
 od_coeff *dst);
int od_hip_download_coeffs(od_hip_ctx *ctx, int slot, int pli, od_coeff *dst);
int od_hip_upload_coeffs(od_hip_ctx *ctx, int slot, int pli, const od_coeff *src);
int od_hip_download_recon(od_hip_ctx *ctx, int slot, int pli, unsigned char *dst);

/* ---------------------------------------------------------------------------
 * 4. PVQ.  State-free part of pvq_theta (src/pvq_encoder.c:311-511): the
 *    no-reference gain candidates of every band of every block of one pyramid
 *    level - od_pvq_compute_gain (src/pvq.c:456), od_pvq_compute_k (:508),
 *    pvq_search_rdo_double (src/pvq_encoder.c:121), distortion (:466).  The
 *    host adds lambda*od_pvq_rate (adaptive entropy state) and picks.
 *
 *    qm: the 16-bit QM of this block size / decimation in coding order, n*n
 *    entries (state.qm + od_qm_offset(bs, xdec), src/pvq.c:292,302).
 *    q[band]: per-band quantiser max(1, q0*pvq_qm_q4[idx] >> 4)
 *    (src/pvq_encoder.c:712); beta[band] from OD_PVQ_BETA (src/pvq.c:230).
 *    Output record layout: od_hip_pvq_band (one per block per band, blocks in
 *    raster order of the level), pulses y as documented at od_hip_pvq_download. */
typedef struct od_hip_pvq_band {
  double cg;          /* companded gain of the input, cg in pvq_theta */
  double g;           /* raw gain */
  double cos_dist[2];
  double dist[2];     /* gain_weight*(qcg-cg)^2 + qcg*cg*(2-2*cos_dist) */
  int32_t qg[2];      /* candidate gain index i */
  int32_t k[2];       /* pulses */
  int32_t ncand;      /* 0..2 */
  int32_t pad;
} od_hip_pvq_band;

/* Bands of an n x n block (OD_BAND_OFFSETS src/partition.c:77-91): returns the
 * band count and fills off[0..nb] (coding-order boundaries). */
int od_hip_band_offsets(int bs, int off[11]);

/* Runs the search for pyramid level `level` of plane pli for slots
 * [slot0, slot0+nslots).  Results stay in HBM until downloaded.
 *
 * The device evaluates NO transcendental: the one libm call on this path, pow() in
 * od_gain_compand (src/pvq.c:422-425, beta = 1.5 with activity masking), is made by the
 * HOST's libm - glibc's pow is not correctly rounded, so only the host's own libm gives
 * the reference's bits.  Hence three steps:
 *   od_hip_pvq_gains          device: exact g = sqrt(sum) of every band (src/pvq.c:456-464)
 *   od_hip_pvq_compand_level  host: g down, cg = od_gain_compand(g, q[band], beta[band])
 *                             with this process's libm, cg up (synchronous)
 *   od_hip_pvq_search         device: candidates, K, codeword search, distortion
 * od_hip_pvq_noref_search = the three in a row.  od_hip_pvq_compand is the bare host loop
 * (cg[i] = od_gain_compand(g[i], q0, beta)). */
int od_hip_pvq_noref_search(od_hip_ctx *ctx, int slot0, int nslots, int pli,
 int level, const int16_t *qm, const int32_t *q, const double *beta);
int od_hip_pvq_gains(od_hip_ctx *ctx, int slot0, int nslots, int pli, int level,
 const int16_t *qm, const int32_t *q, const double *beta);
int od_hip_pvq_compand_level(od_hip_ctx *ctx, int slot0, int nslots, int pli, int level,
 const int32_t *q, const double *beta);
int od_hip_pvq_search(od_hip_ctx *ctx, int slot0, int nslots, int pli, int level,
 const int16_t *qm, const int32_t *q, const double *beta);
int od_hip_pvq_compand(int count, const double *g, int q0, double beta, double *cg);
/* Measurement only: algorithmic work of the searches.  enable = 1 zeroes three device
 * counters that every following search launch of the context adds to and returns their
 * previous values in out: [0] element steps of the greedy scans (pulses placed x band size,
 * src/pvq_encoder.c:166-188), [1] of the RDO scans (:193-220), [2] candidates searched;
 * enable = 0 reads them and stops counting. */
int od_hip_pvq_stats(od_hip_ctx *ctx, int enable, uint64_t out[3]);

/* Number of blocks of that level in one frame; the y array of one slot holds
 * nblocks * 2 * ncoded int32 (ncoded = min(n*n, 512)), block-major, then
 * candidate, then coding order (DC slot unused). */
int od_hip_pvq_nblocks(od_hip_ctx *ctx, int pli, int level);
int od_hip_pvq_download(od_hip_ctx *ctx, int slot, int pli, int level,
 od_hip_pvq_band *bands, int32_t *y);

/* ---------------------------------------------------------------------------
 * 4b. Encoder feed: hands the state-free luma work of a batch of keyframes to
 *    the host's serial stage.  For keyframe luma pvq_theta ALWAYS runs its
 *    no-reference search (src/pvq_encoder.c:452 `is_keyframe && pli == 0`), on
 *    inputs that depend only on the picture (the block's fDCT at that level), so
 *    every pvq_search_rdo_double(x1, n, k, y_tmp, qcg*cg) call of :463 made by the
 *    block-size RDO pass and by the final pass is known before the entropy coder
 *    starts.  od_hip_enc_feed_run = od_hip_forward_pyramid + od_hip_pvq_noref_search
 *    of the 4 luma levels for the slots, then asynchronous copies of the results
 *    into pinned host memory on a second stream, one completion event per slot.
 *    It is the sequence of three phases that a multi-threaded driver calls itself:
 *      od_hip_enc_feed_gains(slot0, nslots)   device: pyramid + exact gains, g -> host
 *      od_hip_enc_feed_compand(slot)          host, any thread, slot by slot: cg with the
 *                                             host's libm (see section 4), cg -> device
 *      od_hip_enc_feed_search(slot0, nslots)  device: searches, records -> host;
 *    od_hip_enc_feed_view blocks until that slot has landed and returns host
 *    pointers in the device's own band-major layout (no repacking):
 *      record r = band*nblk + block (blocks in raster order of the level)
 *      cg[r], ncand[r], qg[c*nrec + r], k[c*nrec + r], cos_dist[c*nrec + r]
 *                                                          c = candidate 0/1
 *      y of band b, 16-bit.  A candidate whose K exceeds 32767 (quantizer 1-2 on noise-like
 *      content) is not searched: its qg[] reads 0, the consumer must search it itself:
 *           y + 2*nblk*yo[b] + (c*nblk + block)*ns[b]
 *           ns[b] = the band's size rounded up to even (one pad entry after the 15 pulses of
 *           band 0), yo[0] = 0, yo[b] = off[b] for b >= 1
 *    level l holds the (32 >> l)-sized luma blocks.  Per frame slot the device hands over ONE
 *    block of records + pulses, one of gains and one of the four level planes (three
 *    transfers), and takes one block of companded gains + search work lists. */
typedef struct od_hip_enc_feed od_hip_enc_feed;

typedef struct od_hip_feed_level {
  int32_t n;                /* block size 32 >> level */
  int32_t nbands;
  int32_t nblk;             /* blocks of this level in one frame */
  int32_t nbx;              /* blocks per row */
  int32_t off[11];          /* band boundaries in coding order */
  int32_t pad;
  const double *cg;         /* [nbands*nblk] companded gain of the band (the host's od_gain_compand of g) */
  const double *g;          /* [nbands*nblk] its uncompanded gain sqrt(acc) (src/pvq.c:456-464): exact */
  const int32_t *ncand;     /* [nbands*nblk] 0..2 */
  const int32_t *qg;        /* [2][nbands*nblk] gain index i; the search's g2 = qg*cg */
  const int32_t *k;         /* [2][nbands*nblk] */
  const double *cos_dist;   /* [2][nbands*nblk] return value of the search */
  const int16_t *y;         /* pulses, see above */
  /* The level's plane of the forward pyramid itself (od_hip_forward_pyramid): the fDCT of
     every n x n block of the lapped picture, i.e. what fdct_2d produces at
     src/encode.c:1139 (block-size RDO pass) and :1308 (od_compute_dcts) for luma.
     Row-major, `lev_stride` elements per row, block (bx, by) at lev + by*n*lev_stride + bx*n. */
  const od_coeff *lev;
  int32_t lev_stride;
  int32_t pad2;
} od_hip_feed_level;

od_hip_enc_feed *od_hip_enc_feed_create(od_hip_ctx *ctx);
void od_hip_enc_feed_destroy(od_hip_enc_feed *feed);
/* Per-level PVQ parameters (as od_hip_pvq_noref_search), set once per stream. */
int od_hip_enc_feed_set_level(od_hip_enc_feed *feed, int level, const int16_t *qm,
 const int32_t *q, const double *beta);
int od_hip_enc_feed_run(od_hip_enc_feed *feed, int slot0, int nslots);
int od_hip_enc_feed_gains(od_hip_enc_feed *feed, int slot0, int nslots);
int od_hip_enc_feed_compand(od_hip_enc_feed *feed, int slot);
int od_hip_enc_feed_search(od_hip_enc_feed *feed, int slot0, int nslots);
int od_hip_enc_feed_view(od_hip_enc_feed *feed, int slot, od_hip_feed_level lev[4]);
/* The chroma planes in the feed (keyframes).  Their forward transforms and their no-reference
 * candidates are as state-free as luma's: pvq_theta runs the chroma no-reference search whenever
 * the CfL reference correlates badly or the gain is small (src/pvq_encoder.c:449-455) on an input
 * that depends on the picture alone.  od_hip_enc_feed_set_level_plane(feed, pli, ...) for every
 * level of a chroma plane (level l holds its ((32 >> xdec) >> l)-sized blocks) puts the plane
 * into every following run; od_hip_enc_feed_view_plane returns its levels (entries beyond the
 * plane's level count are zero).  The two luma-only names above are pli = 0. */
int od_hip_enc_feed_set_level_plane(od_hip_enc_feed *feed, int pli, int level, const int16_t *qm,
 const int32_t *q, const double *beta);
int od_hip_enc_feed_view_plane(od_hip_enc_feed *feed, int slot, int pli, od_hip_feed_level lev[4]);
/* On the coding rank of a superblock-row sharded frame, after od_hip_gather_strips: fetches
 * the slots' host mirrors again (the other ranks' strips arrived device to device). */
int od_hip_enc_feed_refresh(od_hip_enc_feed *feed, int slot0, int nslots);
/* Lossless frames (quantizer 0): the encoder codes the Haar wavelet of every whole
 * superblock (od_haar, src/encode.c:1305; no lapping, DCT or PVQ).  _run_lossless =
 * od_hip_forward_haar of the slots + the three coefficient planes to pinned host memory;
 * _haar_view blocks until the slot has landed: planes[pli] is what od_haar writes into
 * dtmp[pli] superblock by superblock (row-major, strides[pli] elements per row). */
int od_hip_enc_feed_run_lossless(od_hip_enc_feed *feed, int slot0, int nslots);
int od_hip_enc_feed_haar_view(od_hip_enc_feed *feed, int slot, const od_coeff *planes[3],
 int strides[3]);

/* ---------------------------------------------------------------------------
 * 4d. P-frame feed.  On an inter frame the reference of every band is the transform of the
 *    motion-compensated prediction (od_encode_compute_pred, src/encode.c:749-755) - no H/V
 *    intra prediction, no CfL - so both inputs of pvq_theta (src/pvq_encoder.c:311) are known
 *    for every block size of every plane before the first symbol is coded: the device
 *    enumerates the COMPLETE candidate list of every band - the with-reference (gain, theta)
 *    candidates of :399-448 in the reference's loop order (slots 0 .. nref_slots - 1) and the
 *    no-reference ones of :452-481 (the last two slots) - and the host prices them.  Two device
 *    passes around the host's libm (pow, acos, sin are this process's):
 *      od_hip_pfeed_gains       uploads the padded input planes and the prediction, forward
 *                               pyramids of both, g / gr / correlation sum of every band;
 *      od_hip_pfeed_host_stage  cg, cgr, theta, sin(theta), which searches run - any thread,
 *                               disjoint record ranges concurrently;
 *      od_hip_pfeed_search      every candidate's Householder reflection + codeword search.
 *    One object holds one frame.  View layout: record r = band*nblk + block;
 *      cos_dist[c*nrec + r], k[c*nrec + r] (K of slot c, -1: the reference's loops do not reach
 *      that slot; the host re-derives the loops and must find the same K);
 *      y of band b, slot c: y + nslots*nblk*yo[b] + (c*nblk + block)*ns[b], 16-bit, yo / ns as
 *      in section 4b; with-reference codewords have n - 1 entries (src/pvq_encoder.c:426).
 *    Blocks of superblocks that contain padding must not be taken from the feed: the encoder
 *    overwrites the padded part of its input with the prediction's (src/encode.c:2443-2457). */
typedef struct od_hip_pfeed od_hip_pfeed;
typedef struct od_hip_pfeed_level {
  int32_t n, nbands, nblk, nbx;
  int32_t off[11];
  int32_t nslots, nref_slots, pad;
  const double *g, *gr;       /* [nrec] raw gains of input and reference (src/pvq.c:456-464) */
  const double *corr;         /* [nrec] normalised, clamped correlation (:380-381) */
  const int32_t *isnull;      /* [nrec] the reference vector is all zero */
  const double *cg, *cgr;     /* [nrec] od_gain_compand of g, gr (host libm) */
  const double *theta;        /* [nrec] acos(corr) where the theta search runs */
  const int32_t *flags;       /* [nrec] bit 0: theta search ran, bit 1: no-reference search ran */
  const double *cos_dist;     /* [nslots][nrec] */
  const int32_t *k;           /* [nslots][nrec] */
  const int16_t *y;
} od_hip_pfeed_level;
od_hip_pfeed *od_hip_pfeed_create(int device, const od_hip_geometry *geo);
void od_hip_pfeed_destroy(od_hip_pfeed *pf);
int od_hip_pfeed_set_level(od_hip_pfeed *pf, int pli, int level, const int16_t *qm, const int32_t *q,
 const double *beta);
int od_hip_pfeed_gains(od_hip_pfeed *pf, const unsigned char *const planes_in[], const int stride_in[],
 const unsigned char *const planes_pred[], const int stride_pred[]);
int od_hip_pfeed_nrec(od_hip_pfeed *pf, int pli, int level);
int od_hip_pfeed_host_stage(od_hip_pfeed *pf, int pli, int level, long rec0, long rec1);
int od_hip_pfeed_search(od_hip_pfeed *pf);
int od_hip_pfeed_view(od_hip_pfeed *pf, int pli, int level, od_hip_pfeed_level *v);

/* ---------------------------------------------------------------------------
 * 4e. Decoder-side PVQ synthesis of an inter frame: pvq_synthesis (src/pvq_decoder.c:104-118) =
 *    od_compute_householder + od_pvq_synthesis_partial (src/pvq.c:364-413, :552-585) of every
 *    coded band, od_init_skipped_coeffs (src/state.c:1351-1357), the DC (src/decode.c:599-609)
 *    and od_coding_order_to_raster (src/partition.c:176) of every block, written into the
 *    context's coefficient planes of slot 0 - what od_hip_decode_tail reads.  On a P frame the
 *    reference of every band is the transform of the motion-compensated prediction: the object
 *    is attached to a context whose slot 0 holds od_hip_forward_pyramid of the prediction.
 *    The host only parses symbols:
 *      od_hip_dsynth_ref_gains  gr = od_pvq_compute_gain's *g (src/pvq.c:456-464) of every band
 *                               of every block size, [band][block at that level]: the parse
 *                               needs it (src/pvq_decoder.c:221-240);
 *      od_hip_dsynth_buffers    page-locked record buffers the parse fills in place;
 *      od_hip_dsynth_run        uploads nblocks / nbands / npulses entries of them and runs.
 *    Every sample of the frame must belong to exactly one block record.  Bands that copy the
 *    reference (OD_PVQ_SKIP_COPY, skipped blocks, the uncoded half of a 32x32) need no record.
 *    g is od_gain_expand's result, sin_theta / cos_theta the host's libm of the decoded theta. */
typedef struct od_hip_dsynth od_hip_dsynth;
typedef struct od_hip_dsynth_block {
  int32_t org;          /* raster origin in the plane: (by*4)*(frame_width >> xdec) + bx*4 */
  int32_t dc;           /* decoded DC minus the prediction's: pred[0]*dc_quant (src/decode.c:608) */
  uint8_t pli, bs, pad[2];
} od_hip_dsynth_block;
#define OD_HIP_DSYNTH_ZERO 0     /* OD_PVQ_SKIP_ZERO: the band is cleared */
#define OD_HIP_DSYNTH_NOREF 1    /* synthesis without reference, n pulses */
#define OD_HIP_DSYNTH_REF 2      /* synthesis with reference, n - 1 pulses */
#define OD_HIP_DSYNTH_WIDE 4     /* or-ed in: a pulse does not fit 16 bits, every pulse is two entries (low, high half) */
typedef struct od_hip_dsynth_band {
  uint32_t block;       /* index of the band's block in the block records */
  uint8_t band, mode, pad[2];
  uint32_t yoff;        /* first pulse of the band in the pulse buffer */
  uint32_t pad2;
  double g, sin_theta, cos_theta;
} od_hip_dsynth_band;
od_hip_dsynth *od_hip_dsynth_create(od_hip_ctx *ctx);
void od_hip_dsynth_destroy(od_hip_dsynth *s);
int od_hip_dsynth_set_level(od_hip_dsynth *s, int pli, int level, const int16_t *qm, const int16_t *qm_inv);
int od_hip_dsynth_ref_gains(od_hip_dsynth *s, const double *gr[3][4]);
int od_hip_dsynth_buffers(od_hip_dsynth *s, od_hip_dsynth_block **blocks, long *max_blocks,
 od_hip_dsynth_band **bands, long *max_bands, int16_t **pulses, long *max_pulses);
int od_hip_dsynth_run(od_hip_dsynth *s, long nblocks, long nbands, long npulses);

/* ---------------------------------------------------------------------------
 * 4c. Encoder-side deringing: od_dering() (src/filter.c:1835) of EVERY 32x32 superblock of
 *    one frame in one pass - what the encoder's filter on/off loop (src/encode.c:2552-2685)
 *    asks for superblock by superblock.  in[pli]: the unfiltered post-filter planes
 *    (state.etmp[pli], int16 like the reference keeps them, frame_width >> xdec samples per row); bskip / skip_stride as
 *    state.bskip; threshold[pli] = (int)pow(quantizer[pli], 0.84182) computed by the host;
 *    out[pli]: int16 planes of the same geometry.  One object per host worker (own stream
 *    and staging); 4:2:0 / 4:4:4, 8 bit. */
typedef struct od_hip_dering od_hip_dering;
od_hip_dering *od_hip_dering_create(int device, int frame_width, int frame_height,
 int nplanes, const int *xdec);
void od_hip_dering_destroy(od_hip_dering *d);
int od_hip_dering_run(od_hip_dering *d, const int16_t *const in[],
 const unsigned char *const bskip[], int skip_stride, const int32_t *threshold,
 const int32_t *quantizer, int16_t *const out[]);
/* The same plus the two distortions the encoder's on/off decision compares for every 32x32
 * luma superblock (src/encode.c:2606-2636): od_compute_dist(original, unfiltered) and
 * od_compute_dist(original, deringed) (A22, src/encode.c:1032), up to the activity power:
 * per superblock (raster order) and 8x8 sub-block (raster order, 16 per superblock)
 *   dist_arg[sb*16 + k]         the argument .25 + var_stat/256 of pow(., -1/6) (:1007),
 *   dist_unfiltered / _filtered the weighted error energy sum + vardist (:1029),
 * so that od_compute_dist = 1.7 * sum_k (calibration*pow(arg_k, -1./6))^2 * energy_k with
 * the HOST's pow.  orig_luma: the padded 8-bit input plane (frame_width x frame_height
 * used, orig_stride bytes per row); mag2[64]: as od_hip_compute_dist_blocks, bs = 3. */
int od_hip_dering_run_dist(od_hip_dering *d, const int16_t *const in[],
 const unsigned char *const bskip[], int skip_stride, const int32_t *threshold,
 const int32_t *quantizer, int16_t *const out[], const unsigned char *orig_luma,
 int orig_stride, const double *mag2, int activity_masking, double *dist_arg,
 double *dist_unfiltered, double *dist_filtered);

/* Stand-alone batched pieces for parity tests (host memory):
 * nvec band vectors of length n. */
int od_hip_pvq_search_vectors(int n, int nvec, const double *x, const int32_t *k,
 const double *g2, int32_t *y, double *cos_dist);
int od_hip_pvq_synthesis_noref(int n, int nvec, const int32_t *y,
 const double *g, const int16_t *qm_inv, od_coeff *out);

/* Complete candidate enumeration of pvq_theta (src/pvq_encoder.c:311-481) for
 * nvec band vectors of length n, WITHOUT the rate term: with-reference (gain i,
 * angle j) candidates in the reference's loop order, then the no-reference
 * ones.  The host reproduces the decision with cost = dist + lambda*od_pvq_rate,
 * '<' for with-reference and '<=' for no-reference candidates (:435, :469).
 * x0/r0: [nvec][n] input and prediction, qm: [n] (shared), q0: [nvec].
 * y_ref: [nvec][12][n] (n-1 entries used), y_noref: [nvec][2][n]. */
typedef struct od_hip_pvq_theta_out {
  double cg, cgr, g, gr, corr, theta, gain_offset, skip_dist, null_dist;
  int32_t icgr, m, s, nref, nnoref, theta_searched, noref_searched, pad;
  int32_t ref_qg[12], ref_itheta[12], ref_ts[12], ref_k[12];
  double ref_qtheta[12], ref_cos_dist[12], ref_dist[12];
  int32_t nr_qg[2], nr_k[2];
  double nr_cos_dist[2], nr_dist[2];
} od_hip_pvq_theta_out;

int od_hip_pvq_theta_vectors(int n, int nvec, const od_coeff *x0, const od_coeff *r0,
 const int16_t *qm, const int32_t *q0, double beta, int robust, int is_keyframe,
 int pli, od_hip_pvq_theta_out *out, int32_t *y_ref, int32_t *y_noref);

/* Decoder-side synthesis of nvec bands, pvq_synthesis (src/pvq_decoder.c:104-118)
 * = Householder rebuild from the reference + od_pvq_synthesis_partial
 * (src/pvq.c:552-585), both the no-reference and the with-reference branch.
 * y, ref, out: [nvec][n]; gr, noref, g, theta: [nvec]; qm, qm_inv: [n]. */
int od_hip_pvq_synthesis_vectors(int n, int nvec, const int32_t *y, const od_coeff *ref,
 const double *gr, const int32_t *noref, const double *g, const double *theta,
 const int16_t *qm, const int16_t *qm_inv, od_coeff *out);

/* ---------------------------------------------------------------------------
 * Superblock-row sharding of one frame over the GPUs of a node (BASELINE
 * configs[2]).  One process per GPU, each with the whole input frame in its slot.
 *   od_hip_set_strip(ctx, r0, r1)   restricts od_hip_forward_pyramid and the PVQ passes
 *       (od_hip_pvq_gains / _compand_level / _search / _noref_search) to superblock rows
 *       [r0, r1); the kernels read their lapping halo from the pixels, so a strip needs
 *       nothing from its neighbours.  (0, nvsb) - the default - is the whole frame.
 *   od_hip_comm_unique_id / od_hip_comm_create   an RCCL communicator over the ranks (the
 *       128-byte id travels by whatever the launcher offers, e.g. torch.distributed);
 *   od_hip_gather_strips   device-to-device gather over xGMI to the CODING rank (rank 0):
 *       afterwards rank 0's slot holds the complete pyramid (and PVQ records), sb_rows[r] ..
 *       sb_rows[r + 1] being rank r's strip.  Every owner packs its strip into one buffer
 *       (one kernel), sends it with ONE ncclSend, rank 0 posts one ncclRecv per owner in one
 *       group and unpacks; the only collective of the path, nothing crosses the host.
 *   od_hip_strip_bytes / _export / _import   the same packed strip through host memory, for
 *       launchers whose ranks have no RCCL between them (a gloo rehearsal on one GPU). */
typedef struct od_hip_comm od_hip_comm;
int od_hip_set_strip(od_hip_ctx *ctx, int sb_row0, int sb_row1);
int od_hip_comm_unique_id(unsigned char id[128]);
od_hip_comm *od_hip_comm_create(int device, int world, int rank, const unsigned char id[128]);
void od_hip_comm_destroy(od_hip_comm *comm);
int od_hip_gather_strips(od_hip_ctx *ctx, od_hip_comm *comm, int slot, const int *sb_rows,
 int with_pvq);
long od_hip_strip_bytes(od_hip_ctx *ctx, int slot, int sb_row0, int sb_row1, int with_pvq);
int od_hip_strip_export(od_hip_ctx *ctx, int slot, int sb_row0, int sb_row1, int with_pvq,
 void *host, long cap);
int od_hip_strip_import(od_hip_ctx *ctx, int slot, int sb_row0, int sb_row1, int with_pvq,
 const void *host, long bytes);

/* F3 (inter frames, first kernel): overlapped block motion compensation of a list of
 * prediction blocks = the leaves od_state_mc_predict (src/state.c:993) visits through
 * od_state_pred_block -> od_state_pred_block_from_setup (:689) -> od_mc_predict
 * (src/mc.c:2006): per block four od_mc_predict1fmv8_c predictions (src/mc.c:94), one per
 * corner motion vector, blended by od_mc_blend_full8_c (s == 3, :352) or
 * od_mc_blend_full_split8_c (:1104).  8-bit references.
 *   refs[k]: reference plane k, ref_h rows of ref_stride bytes, the picture's (0, 0) at
 *            (org_x, org_y) - the reference keeps OD_UMV padding around its frames;
 *   block:   (x, y) in the picture, 2^log_xblk_sz x 2^log_yblk_sz samples (4..64), corner k
 *            reads reference ref[k] with vector (mvx[k], mvy[k]) in 1/8 sample of THIS plane
 *            (the host applies OD_DIV_POW2_RE for chroma, :719-720); oc, s as the reference;
 *   dst:     in/out plane, dst_h rows of dst_stride bytes; only the blocks are written. */
typedef struct od_hip_mc_block {
  int32_t x, y;
  int32_t log_xblk_sz, log_yblk_sz;
  int32_t ref[4];
  int32_t mvx[4], mvy[4];
  int32_t oc, s;
} od_hip_mc_block;
int od_hip_mc_predict_blocks(int nref, const unsigned char *const refs[], int ref_stride,
 int ref_h, int org_x, int org_y, const od_hip_mc_block *blocks, int nblocks,
 unsigned char *dst, int dst_stride, int dst_h);

/* The same with the reference frames RESIDENT on the device: one object per host thread that
 * predicts frames (own stream and staging, safe beside any other object).
 *   od_hip_mc_set_ref   uploads reference image k of plane pli (geometry as refs[k] above);
 *                       call it when that image has changed - not per predicted plane;
 *   od_hip_mc_predict   predicts the blocks of plane pli from the resident references into
 *                       dst: dst_w x dst_h samples are written (the blocks tile that area),
 *                       rows dst_stride bytes apart; nothing else of dst is touched. */
typedef struct od_hip_mc od_hip_mc;
od_hip_mc *od_hip_mc_create(int device, int nref);
void od_hip_mc_destroy(od_hip_mc *mc);
int od_hip_mc_set_ref(od_hip_mc *mc, int pli, int k, const unsigned char *plane, int ref_stride,
 int ref_h, int org_x, int org_y);
int od_hip_mc_predict(od_hip_mc *mc, int pli, const od_hip_mc_block *blocks, int nblocks,
 unsigned char *dst, int dst_stride, int dst_w, int dst_h);
/* The same prediction written into picture plane pli of slot `slot` of a context on the same
 * device (the blocks must tile the context's padded plane): for a decoder, whose only consumer
 * of the prediction is od_hip_forward_pyramid of that context - nothing crosses PCIe.  The
 * context's stream waits for the prediction on the device; the call does not block. */
int od_hip_mc_predict_ctx(od_hip_mc *mc, int pli, const od_hip_mc_block *blocks, int nblocks,
 od_hip_ctx *ctx, int slot);
/* Reference image k of plane pli taken from the RECONSTRUCTION plane of slot `slot` of a
 * context on the same device (what od_hip_decode_tail left there), padded on the device as
 * od_img_edge_ext pads a reference image (src/state.c:1100-1171: the frame's edge samples
 * replicated): a decoded frame becomes the next frame's reference without crossing PCIe
 * (src/decode.c:1267-1270 on the device).  Geometry arguments as od_hip_mc_set_ref. */
int od_hip_mc_set_ref_ctx(od_hip_mc *mc, int pli, int k, od_hip_ctx *ctx, int slot, int ref_stride,
 int ref_h, int org_x, int org_y);

/* F3, second half: batched OBMC prediction + SAD for the motion search.  What
 * od_mv_est_sad (src/mcenc.c:2271-2300) computes for one block - od_state_pred_block_from_setup
 * (src/state.c:689) of every plane, od_enc_sad (src/mcenc.c:1615: the block clipped against the
 * picture) with the chroma sums >> OD_MC_CHROMA_SCALE (:53) - for a whole list of (block,
 * exterior corner, split state) items against the references resident in the object: the batch
 * stages of od_mv_est, where the vector grid stands still - od_mv_est_calc_sads (:3761: every
 * block of two sizes x four split states) and the top-level blocks of od_mv_est_init_du (:3953).
 *   x, y, log_blk_sz: luma position and log2 size (3..6); mvx/mvy: the four corner vectors in
 *   luma 1/8-sample units as the grid holds them (chroma vectors: OD_DIV_POW2_RE on the device);
 *   ref[k]: index into the object's reference images; oc, s as od_mc_predict (src/mc.c:2006).
 * od_hip_mc_set_src uploads plane pli of the frame being coded (w x h samples, the padded input
 * plane; xdec/ydec: its decimation); od_hip_mc_sad_items writes sad[i] for every item, planes
 * 0..nplanes-1 summed.  pic_w/pic_h: the luma picture size the SAD is clipped to. */
typedef struct od_hip_mc_sad_item {
  int32_t x, y;
  int32_t log_blk_sz;
  int32_t oc, s;
  int32_t ref[4];
  int32_t mvx[4], mvy[4];
  int32_t reserved;
} od_hip_mc_sad_item;
int od_hip_mc_set_src(od_hip_mc *mc, int pli, const unsigned char *plane, int stride, int w, int h,
 int xdec, int ydec);
int od_hip_mc_sad_items(od_hip_mc *mc, int nplanes, int pic_w, int pic_h,
 const od_hip_mc_sad_item *items, int nitems, int32_t *sad);

/* F3: dense block-matching windows for the EPZS initialisation of the motion search
 * (od_mv_est_init_mv, src/mcenc.c:2511).  od_mv_est_bma_sad (:2228-2268) - ONE single-vector
 * prediction of the block centred on a grid vertex (od_mc_predict1fmv8_c on every plane, the
 * vector in half samples) + od_enc_sad (:1615) clipped against the picture, chroma >>
 * OD_MC_CHROMA_SCALE - for EVERY half-sample vector within `radius` of the record's centre (the
 * vertex's median predictor, known for a whole level of the grid before the level starts), all
 * records in one launch, against the object's resident references and source planes.
 *   bx, by: luma position of the block (multiples of 4; may be negative or hang over the frame);
 *   cx, cy: window centre; xmin..ymax: the vertex's vector limits (od_mv_est_limits), inclusive.
 * out: [nrec][(2 radius + 1)^2] row-major in (dy, dx); -1 where the vector is outside the limits. */
typedef struct od_hip_mc_bma_rec {
  int32_t bx, by;
  int32_t log_blk_sz;
  int32_t ref;
  int32_t cx, cy;
  int32_t xmin, xmax, ymin, ymax;
} od_hip_mc_bma_rec;
int od_hip_mc_bma_windows(od_hip_mc *mc, int nplanes, int pic_w, int pic_h,
 const od_hip_mc_bma_rec *recs, int nrec, int radius, int32_t *out);

/* A11: od_raster_to_coding_order (to_raster = 0, src/partition.c:144) and
 * od_coding_order_to_raster (to_raster = 1, :176) for nblocks dense n x n blocks
 * (n = 4 << bs).  dst is in/out: entries the permutation does not write (a 32x32 block
 * codes 512 of its 1024 coefficients) keep the caller's values, like the reference's
 * buffers (od_init_skipped_coeffs before the scatter, src/encode.c:1219-1220). */
int od_hip_coding_order_blocks(int bs, int to_raster, od_coeff *inout_dst, const od_coeff *src,
 int nblocks);

/* Keyframe luma predictor of od_encode_compute_pred (src/encode.c:732-737):
 * OD_CLEAR + od_hv_intra_pred (src/intra.c:37-61) for nblk blocks of size bs at
 * 4x4-unit positions (bx[i], by[i]) of the w x h coefficient plane d.
 * pred: [nblk][n*n]. */
int od_hip_hv_intra_pred_blocks(const od_coeff *d, int w, int h,
 const unsigned char *bsize, int bstride, int bs, int nblk, const int32_t *bx,
 const int32_t *by, od_coeff *pred);

/* od_compute_dist (src/encode.c:1032-1058, HVS quantisation matrix) for nblk pairs
 * of dense n x n blocks (n = 4 << bs, bs = 1..3): the perceptual distortion the
 * block-size RDO compares (src/encode.c:1631-1632).  mag2[64]: squared weights
 * (16/OD_QM8_Q4_HVS[i][j] * OD_BASIS_MAG[0][bs][i<<(bs-1)] * ...[j<<(bs-1)])^2,
 * src/encode.c:1018-1025, computed by the host from its tables. */
int od_hip_compute_dist_blocks(int bs, int nblk, const od_coeff *x, const od_coeff *y,
 const double *mag2, int activity_masking, double *dist);

/* Diagnostic: the device's pow/acos/sin/cos/sqrt/divide on n doubles, so tests can
 * quantify agreement with the host libm the reference uses.
 * fn: 0 pow(x,y), 1 acos(x), 2 sin(x), 3 cos(x) (OCML; no product kernel calls them: the
 * transcendentals of the path are evaluated by the host's libm between device passes),
 * 4 sqrt(x), 5 x/y (used everywhere: must be identical to the host's). */
int od_hip_libm_probe(int fn, int n, const double *x, const double *y, double *out);

/* Profiling aid: streams a `bytes`-sized device buffer once with the access width
 * the transform kernels use (mode 0: dword loads, 1: int4 loads, 2: int4 stores) so
 * that the rocprofv3 FETCH_SIZE / WRITE_SIZE counters can be calibrated against a
 * known byte count (tools/profile_round.sh). */
int od_hip_calibrate_traffic(int mode, size_t bytes);

/* Page-locks (hipHostRegister) / unlocks a host buffer the caller keeps handing to the
 * upload/download entry points - e.g. the reference's dtmp coefficient planes in the
 * decoder binding - so that those copies run as direct DMA instead of through a
 * staging buffer.  Optional: unregistered memory works, slower. */
int od_hip_host_register(void *ptr, size_t bytes);
int od_hip_host_unregister(void *ptr);

/* Synchronise the context's stream / time its last batch (ms, HIP events). */
int od_hip_sync(od_hip_ctx *ctx);
/* hipDeviceSynchronize on `device`: every stream of every object of this process on it. */
int od_hip_device_sync(int device);

/* Kernel timing hooks for bench.py: HIP events recorded on the context's own
 * stream around every launch of the named kernel since the last reset.
 * Returns the number of launches and sum of durations in ms. */
int od_hip_timing_reset(od_hip_ctx *ctx);
int od_hip_timing_get(od_hip_ctx *ctx, const char *kernel, int *launches,
 double *total_ms);

#ifdef __cplusplus
}
#endif
#endif
