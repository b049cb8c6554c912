/* hip_enc_glue.h - reference-side binding of the batched frame seam
 * (seam 2).  The glue is C that is compiled INTO a build of the
 * reference encoder (it includes the reference's internal headers); together with
 * libdaala_hip.so it turns daala_encode_img_in() into "device does the state-free
 * keyframe-luma PVQ searches for a batch of frames, N host workers run the serial
 * entropy/RDO stage".  Plain C ABI. */
#ifndef HIP_ENC_GLUE_H
#define HIP_ENC_GLUE_H

#include <stdint.h>
#include "../../include/daala_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct od_hipenc_params {
  int32_t pic_width, pic_height;   /* 4:2:0, 8 bit */
  int32_t quant;                   /* OD_SET_QUANT */
  int32_t complexity;              /* OD_SET_COMPLEXITY */
  int32_t masking;                 /* OD_SET_ACTIVITY_MASKING */
  int32_t nworkers;                /* host threads, one reference encoder context each */
  int32_t check;                   /* OD_CHECKGPU: run the C search as well and compare */
  int32_t batch;                   /* frame slots resident on the device (0: default) */
  int32_t keyframe_rate;           /* 0 or 1: every frame a keyframe (frames independent: N workers);
                                      > 1: inter frames between keyframes - ONE worker codes the
                                      stream in order (nworkers is forced to 1) */
} od_hipenc_params;

typedef struct od_hipenc_stats {
  int64_t dev_hits;        /* pvq_search_rdo_double calls answered from the device feed */
  int64_t cpu_noref_luma;  /* keyframe-luma no-reference searches that ran on the host anyway */
  int64_t cpu_other;       /* with-reference and chroma searches (serial-state dependent) */
  int64_t g2_mismatch;     /* candidate present but host g2 != device qg*cg (libm pow, 1 ulp) */
  int64_t lost_sync;       /* blocks whose call sequence did not match the feed */
  int64_t check_fail;      /* check mode: device answer != C answer (must be 0) */
  int64_t resampled;       /* feed candidates re-searched on the host and compared (check_fail counts mismatches) */
  int64_t pvq_check_fail;  /* check mode: block result != the reference's od_pvq_encode (must be 0) */
  double search_cpu_s;     /* seconds inside the C pvq_search_rdo_double, all workers */
  double search_class_s[4];/* ... split: luma no-ref, luma with-ref, chroma no-ref, chroma with-ref */
  int64_t fdct_hits;       /* luma fdct_2d calls answered from the device pyramid */
  int64_t haar_hits;       /* lossless frames: od_haar calls (whole superblocks) answered from the device's Haar planes */
  int64_t fdct_check_fail; /* check mode: device block != C transform (must be 0) */
  int64_t dering_dev_sbs;  /* od_dering calls (superblock, plane) answered from the device pass */
  int64_t dering_check_fail; /* check mode: device block != C od_dering (must be 0) */
  int64_t dist_dev;        /* od_compute_dist calls of the deringing on/off loop answered from the device pass */
  int64_t dist_check_fail; /* check mode: device distortion != the reference's od_compute_dist (must be 0) */
  int64_t pfeed_frames;    /* inter frames whose bands took the complete candidate lists of the P-frame feed */
  double t_pfeed_s;        /* seconds the coding thread waited for P-frame feeds (device passes + libm stage) */
  /* HIPENC_TIME=1 only (0 otherwise): host time classes, summed over workers */
  double rate_s;           /* inside the rate-only pricing of codewords */
  double rate_state_free_s; /* a state-free skeleton of the same pricing run beside it (see hip_pvq_host.c) */
  int64_t rate_calls;
  double frame_cpu_s;      /* inside daala_encode_img_in + packet_out, all frames */
  double pre_mc_s;         /* P frames: frame start -> od_state_mc_predict (input copy + od_mv_est) */
  double mv_stage_s[8];    /* od_mv_est by stage (always on; mcenc_tail.c): EPZS initialisation of the previous
                              reference, of the golden/next one, od_mv_est_calc_sads, the rest of od_mv_est_init_dus,
                              the decimation loop, the refinement loop, sub-pel refinement, the whole od_mv_est */
  int64_t mv_dev_calls;    /* batched OBMC + SAD passes of the motion search answered by the device */
  int64_t mv_dev_sads;     /* block SADs in them */
  double mv_dev_wait_s;    /* seconds the coding thread spent in those calls */
  int64_t mv_check_fail;   /* check mode: device SAD != the reference's od_mv_est_calc_sads (must be 0) */
  int64_t mv_bma_calls;    /* EPZS initialisation: device calls (one per level and vertex parity) */
  int64_t mv_bma_windows;  /* EPZS initialisation: vertices whose block-matching window came from the device */
  int64_t mv_bma_hits;     /* ... od_mv_est_bma_sad calls answered from a window */
  int64_t mv_bma_misses;   /* ... calls for a vector outside the vertex's window (reference code on the host) */
  int64_t mv_level_walks;  /* od_mv_est_init_mvs calls that walked the grid level by level (one per reference and P frame) */
  double t_setup_s;        /* encoder/device context creation (not in t_total_s) */
  double t_upload_s;       /* pad + upload phase, wall */
  double t_launch_s;       /* upload done -> device batch enqueued (includes t_compand_s), wall */
  double t_compand_s;      /* upload done -> every frame's gains companded by the host's libm, wall */
  double t_total_s;        /* wall: first frame in -> last packet out */
  int64_t pkt_bytes_needed; /* size of the packet blob (with the 4-byte length prefixes) */
} od_hipenc_stats;

/* Encodes nframes dense 4:2:0 frames (Y then U then V, picture size) as
 * independent keyframes (keyframe_rate 1), bit-identical to one reference encoder
 * fed the frames in order.  Packets are written to pkt_out in frame order, each
 * prefixed by its 4-byte little-endian length.  views: NULL, or
 * [nframes][4] host-fabricated feed views (tests); use_device: take the feed from
 * the HIP device `device` (fails loudly when there is none); neither: the plain
 * reference search on nworkers threads.  Returns total packet bytes or < 0.
 * All packets or none: when the blob (stats->pkt_bytes_needed bytes) does not fit
 * pkt_cap nothing is written and OD_HIP_ENOSPC is returned. */
long od_hipenc_encode_frames(const od_hipenc_params *p, int nframes,
 const unsigned char *frames, const od_hip_feed_level *views, int use_device,
 int device, unsigned char *pkt_out, long pkt_cap, od_hipenc_stats *stats);

/* The same as a session: od_hipenc_open creates the p->nworkers host workers (one
 * reference encoder context each) and, with use_device, the device context + encoder feed
 * with p->batch frame slots (0: a default bounded by pinned host memory) ONCE;
 * od_hipenc_encode codes one stream of nframes independent keyframes whose first frame
 * has stream index frame0 (a stream longer than the slots runs through two half-buffers);
 * views only without a device (tests).  A session codes one stream at a time.
 * od_hipenc_open returns NULL and sets *err (OD_HIP_ENODEV: no device, no encode). */
typedef struct od_hipenc od_hipenc;
od_hipenc *od_hipenc_open(const od_hipenc_params *p, int use_device, int device, int *err);
long od_hipenc_encode(od_hipenc *s, int nframes, long frame0, const unsigned char *frames,
 const od_hip_feed_level *views, unsigned char *pkt_out, long pkt_cap,
 od_hipenc_stats *stats);
void od_hipenc_close(od_hipenc *s);

/* The per-level PVQ parameters the device needs, read from a live reference
 * encoder context of these settings (what bench/tests pass to
 * od_hip_enc_feed_set_level): for level l (block size 32 >> l) qm[l][1024],
 * q[l][11], beta[l][11]. */
int od_hipenc_level_params(const od_hipenc_params *p, int16_t qm[4][1024],
 int32_t q[4][11], double beta[4][11]);

/* Padded input planes of one frame exactly as daala_encode_img_in() codes them
 * (od_img_copy_pad, src/encode.c:1728): planes[pli] receives frame_width x
 * frame_height (chroma half size) dense bytes; planes may be NULL to query the size. */
int od_hipenc_pad_frame(const od_hipenc_params *p, const unsigned char *frame,
 unsigned char *const planes[3], int *frame_width, int *frame_height);

/* Header packets of a stream with these settings (length-prefixed, 3 packets). */
long od_hipenc_headers(const od_hipenc_params *p, unsigned char *out, long cap);

/* Decoder side of the seam (hip_dec_glue.c): decodes nframes keyframe packets
 * (length-prefixed, as written by od_hipenc_encode_frames) on p->nworkers threads,
 * one reference decoder context each.  use_device: the symbol parse stays the
 * reference's C code, the whole pixel-domain stage of every frame (iDCT, split and
 * frame post-filters, deringing, bilinear smoothing, 8-bit clamp) is one
 * od_hip_decode_tail pass per frame; otherwise the plain reference decoder.
 * frames_out: nframes dense 4:2:0 pictures.  Returns nframes or < 0; fails with
 * OD_HIP_ENODEV when use_device is set and there is no device. */
long od_hipdec_decode_frames(const od_hipenc_params *p, const unsigned char *hdr,
 long hdr_bytes, int nframes, const unsigned char *pkts, long pkt_bytes,
 int use_device, int device, unsigned char *frames_out, double *seconds,
 double *device_seconds);

/* After od_hipdec_decode_frames: out[0] = inter frames whose motion-compensated prediction
 * (od_state_mc_predict, src/state.c:993) came from the device, out[1] = check-mode mismatches
 * against the reference's own prediction (must be 0). */
void od_hipdec_mc_stats(long out[2]);

#ifdef __cplusplus
}
#endif
#endif
