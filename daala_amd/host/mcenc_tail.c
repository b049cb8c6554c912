/* Appended to the reference's src/mcenc.c in the HIP build (Makefile: sed | cat | gcc).
 * The stage functions of od_mv_est (src/mcenc.c:6390) run as the reference wrote them
 * (*_cpu); the definitions here are what their call sites bind to:
 *   - every stage is timed (a handful of clock reads per frame) into the worker's statistics
 *     (od_hipenc_stats.mv_stage_s), which is where profiles/r04_mvest_stages.md comes from. */
void od_hipenc_mv_stage(int stage, double seconds);   /* hip_enc_glue.c */
double od_hipenc_now(void);

enum {
  OD_HIPENC_MV_INIT_PREV = 0,    /* od_mv_est_init_mvs(OD_FRAME_PREV): EPZS initialisation */
  OD_HIPENC_MV_INIT_OTHER = 1,   /* ... of the golden / next reference */
  OD_HIPENC_MV_CALC_SADS = 2,    /* od_mv_est_calc_sads */
  OD_HIPENC_MV_INIT_DUS = 3,     /* od_mv_est_init_dus without od_mv_est_calc_sads */
  OD_HIPENC_MV_DECIMATE = 4,     /* od_mv_est_decimate without od_mv_est_init_dus */
  OD_HIPENC_MV_REFINE = 5,       /* the od_mv_est_refine loop (what is left of od_mv_est) */
  OD_HIPENC_MV_SUBPEL = 6,       /* od_mv_subpel_refine */
  OD_HIPENC_MV_TOTAL = 7         /* od_mv_est */
};

static __thread double mv_inner;     /* time of stages nested in the one being timed */

static void od_mv_est_init_mvs(od_mv_est_ctx *est, int ref, int must_update) {
  double t0;
  t0 = od_hipenc_now();
  od_mv_est_init_mvs_cpu(est, ref, must_update);
  od_hipenc_mv_stage(ref == OD_FRAME_PREV ? OD_HIPENC_MV_INIT_PREV : OD_HIPENC_MV_INIT_OTHER,
   od_hipenc_now() - t0);
}

/* od_mv_est_calc_sads (src/mcenc.c:3761-3823) with its SADs from the device.  The vector grid
   stands still during the call, so every od_mv_est_sad of its loop nest - every block of the
   block sizes the level limits admit x the split states 0 .. smax - 1 - is known up front: the
   loops below walk the reference's order once to list the items (what
   od_state_pred_block_from_setup, src/state.c:689-734, reads from the grid for each), one device
   call returns the sums (od_hip_mc_sad_items: OBMC of every plane fused with the clipped SAD),
   and a second walk stores them and the blocks' set-up state exactly as the reference does.
   Returns 0 when the call is not the device's (no device thread, B frames, SATD, level_max <= 0:
   the tail loop of the reference): od_mv_est_calc_sads_cpu runs instead. */
#include "../../include/daala_hip.h"
int od_hipenc_mv_sad_items(daala_enc_ctx *enc, int nplanes, const od_hip_mc_sad_item *items,
 int nitems, int32_t *sad);
void od_hipenc_mv_check_fail(long n);
int od_hipenc_check_mode(void);

static __thread od_hip_mc_sad_item *mv_items;
static __thread int32_t *mv_sads;
static __thread int mv_items_cap;
/* The SADs of the top-level (64x64) blocks, which od_mv_est_init_du asks for once per vertex
   whose error domain touches them (src/mcenc.c:3937-3942: the same block again and again while
   the grid stands still): they ride the same device call and od_mv_est_sad answers from here
   while od_mv_est_init_dus runs. */
static __thread int32_t *mv_top;
static __thread int mv_top_cap;
static __thread int mv_top_nh;           /* top-level blocks per row; 0: table not valid */

static int mv_calc_sads_device(od_mv_est_ctx *est) {
  od_state *state;
  int nplanes;
  int pass;
  int n;
  int ntop;
  int rc;
  state = &est->enc->state;
  ntop = 0;
  mv_top_nh = 0;
  if (est->level_max <= 0 || state->frame_type != OD_P_FRAME || est->compute_distortion != od_enc_sad) return 0;
  nplanes = (est->flags & OD_MC_USE_CHROMA) ? est->enc->input_img[est->enc->curr_frame].nplanes : 1;
  if (nplanes != 1 && nplanes != 3) return 0;
  n = 0;
  rc = 0;
  /* pass 0: count; pass 1: list the items; pass 2 (after the device call): store */
  for (pass = 0; pass < 3; pass++) {
    int nhmvbs;
    int nvmvbs;
    int log_mvb_sz;
    int i;
    nhmvbs = state->nhmvbs;
    nvmvbs = state->nvmvbs;
    i = 0;
    for (log_mvb_sz = 0; log_mvb_sz < OD_LOG_MVB_DELTA0; log_mvb_sz++) {
      if (est->level_max >= OD_MC_LEVEL_MAX - 1 - 2*log_mvb_sz
       && est->level_min <= OD_MC_LEVEL_MAX - 2*log_mvb_sz) {
        int smax;
        int vx;
        int vy;
        int s;
        smax = est->level_max >= OD_MC_LEVEL_MAX - 2*log_mvb_sz ? 4 : 1;
        for (vy = 0; vy < nvmvbs; vy++) {
          for (vx = 0; vx < nhmvbs; vx++) {
            int oc;
            oc = (vx & 1) ^ ((vy & 1) << 1 | (vy & 1));
            for (s = 0; s < smax; s++, i++) {
              if (pass == 1) {
                od_hip_mc_sad_item *it;
                const int *dxp;
                const int *dyp;
                int k;
                it = mv_items + i;
                dxp = OD_VERT_SETUP_DX[oc][s];
                dyp = OD_VERT_SETUP_DY[oc][s];
                it->x = vx << log_mvb_sz << OD_LOG_MVBSIZE_MIN;
                it->y = vy << log_mvb_sz << OD_LOG_MVBSIZE_MIN;
                it->log_blk_sz = log_mvb_sz + OD_LOG_MVBSIZE_MIN;
                it->oc = oc;
                it->s = s;
                it->reserved = 0;
                for (k = 0; k < 4; k++) {
                  const od_mv_grid_pt *g;
                  g = state->mv_grid[(vy + dyp[k]) << log_mvb_sz] + ((vx + dxp[k]) << log_mvb_sz);
                  if (g->ref == OD_FRAME_NEXT || state->ref_imgi[g->ref] < 0) return 0;
                  it->mvx[k] = g->mv[0];
                  it->mvy[k] = g->mv[1];
                  it->ref[k] = state->ref_imgi[g->ref];
                }
              }
              else if (pass == 2) est->sad_cache[log_mvb_sz][vy][vx][s] = mv_sads[i];
            }
            /* "While we're here, fill in the block's setup state." (:3798-3803) */
            if (pass == 2 && est->level_max <= OD_MC_LEVEL_MAX - 2*log_mvb_sz) {
              od_mv_node *mv;
              mv = est->mvs[vy << log_mvb_sz] + (vx << log_mvb_sz);
              mv->oc = oc;
              mv->log_mvb_sz = log_mvb_sz;
              mv->s = smax - 1;
              mv->sad = est->sad_cache[log_mvb_sz][vy][vx][smax - 1];
            }
          }
        }
      }
      nhmvbs >>= 1;
      nvmvbs >>= 1;
    }
    if (pass == 0) {
      n = i;
      if (n == 0) return 0;
      ntop = (state->nhmvbs >> OD_LOG_MVB_DELTA0)*(state->nvmvbs >> OD_LOG_MVB_DELTA0);
      if (n + ntop > mv_items_cap) {
        free(mv_items);
        free(mv_sads);
        mv_items = (od_hip_mc_sad_item *)malloc(sizeof(*mv_items)*(n + ntop));
        mv_sads = (int32_t *)malloc(sizeof(*mv_sads)*(n + ntop));
        mv_items_cap = mv_items != NULL && mv_sads != NULL ? n + ntop : 0;
        if (mv_items_cap == 0) return 0;
      }
      if (ntop > mv_top_cap) {
        free(mv_top);
        mv_top = (int32_t *)malloc(sizeof(*mv_top)*ntop);
        mv_top_cap = mv_top != NULL ? ntop : 0;
        if (mv_top_cap == 0) ntop = 0;
      }
    }
    else if (pass == 1) {
      /* the top-level blocks behind the regular items: oc 0, s 3 (src/mcenc.c:3940) */
      int t;
      for (t = 0; t < ntop; t++) {
        od_hip_mc_sad_item *it;
        int tvx;
        int tvy;
        int k;
        it = mv_items + n + t;
        tvx = (t%(state->nhmvbs >> OD_LOG_MVB_DELTA0)) << OD_LOG_MVB_DELTA0;
        tvy = (t/(state->nhmvbs >> OD_LOG_MVB_DELTA0)) << OD_LOG_MVB_DELTA0;
        it->x = tvx << OD_LOG_MVBSIZE_MIN;
        it->y = tvy << OD_LOG_MVBSIZE_MIN;
        it->log_blk_sz = OD_LOG_MVB_DELTA0 + OD_LOG_MVBSIZE_MIN;
        it->oc = 0;
        it->s = 3;
        it->reserved = 0;
        for (k = 0; k < 4; k++) {
          const od_mv_grid_pt *g;
          g = state->mv_grid[tvy + (OD_VERT_SETUP_DY[0][3][k] << OD_LOG_MVB_DELTA0)]
           + tvx + (OD_VERT_SETUP_DX[0][3][k] << OD_LOG_MVB_DELTA0);
          if (g->ref == OD_FRAME_NEXT || state->ref_imgi[g->ref] < 0) return 0;
          it->mvx[k] = g->mv[0];
          it->mvy[k] = g->mv[1];
          it->ref[k] = state->ref_imgi[g->ref];
        }
      }
      rc = od_hipenc_mv_sad_items(est->enc, nplanes, mv_items, n + ntop, mv_sads);
      if (rc <= 0) return 0;          /* rc < 0: the frame is marked failed; memory stays defined */
      for (t = 0; t < ntop; t++) mv_top[t] = mv_sads[n + t];
      mv_top_nh = ntop > 0 ? state->nhmvbs >> OD_LOG_MVB_DELTA0 : 0;
    }
  }
  return 1;
}

/* per-thread clean-up (worker exit) */
void od_hipenc_mv_thread_cleanup(void) {
  free(mv_items);
  free(mv_sads);
  free(mv_top);
  mv_items = NULL;
  mv_sads = NULL;
  mv_top = NULL;
  mv_items_cap = 0;
  mv_top_cap = 0;
  mv_top_nh = 0;
}

static void od_mv_est_calc_sads(od_mv_est_ctx *est) {
  double t0;
  double dt;
  t0 = od_hipenc_now();
  if (!mv_calc_sads_device(est)) od_mv_est_calc_sads_cpu(est);
  else if (od_hipenc_check_mode()) {
    /* OD_CHECKASM: the reference's loop on the same grid; its results stay */
    long bad;
    int l;
    int i;
    int n;
    bad = 0;
    for (l = 0, n = 0; l < OD_LOG_MVB_DELTA0; l++) n += (est->enc->state.nhmvbs >> l)*(est->enc->state.nvmvbs >> l)*4;
    {
      int32_t *keep;
      int32_t *q;
      keep = (int32_t *)malloc(sizeof(*keep)*(n > 0 ? n : 1));
      q = keep;
      for (l = 0; keep != NULL && l < OD_LOG_MVB_DELTA0; l++) {
        int cells;
        cells = (est->enc->state.nhmvbs >> l)*(est->enc->state.nvmvbs >> l);
        if (cells > 0) memcpy(q, est->sad_cache[l][0], sizeof(od_sad4)*cells);
        q += cells*4;
      }
      od_mv_est_calc_sads_cpu(est);
      q = keep;
      for (l = 0; keep != NULL && l < OD_LOG_MVB_DELTA0; l++) {
        int cells;
        int smax;
        cells = (est->enc->state.nhmvbs >> l)*(est->enc->state.nvmvbs >> l);
        smax = est->level_max >= OD_MC_LEVEL_MAX - 1 - 2*l && est->level_min <= OD_MC_LEVEL_MAX - 2*l ?
         (est->level_max >= OD_MC_LEVEL_MAX - 2*l ? 4 : 1) : 0;
        for (i = 0; i < cells*4; i++) bad += (i & 3) < smax && q[i] != est->sad_cache[l][0][i >> 2][i & 3];
        q += cells*4;
      }
      if (keep == NULL) bad++;
      free(keep);
    }
    od_hipenc_mv_check_fail(bad);
  }
  dt = od_hipenc_now() - t0;
  od_hipenc_mv_stage(OD_HIPENC_MV_CALC_SADS, dt);
  mv_inner += dt;
}

/* od_mv_est_sad (src/mcenc.c:2271) as its callers see it: a top-level block asked for while
   od_mv_est_init_dus runs (the grid's vectors stand still from od_mv_est_calc_sads to its end) is
   answered from the device call of od_mv_est_calc_sads; everything else is the reference's. */
static int32_t od_mv_est_sad(od_mv_est_ctx *est, int vx, int vy, int oc, int s, int log_mvb_sz) {
  if (mv_top_nh > 0 && log_mvb_sz == OD_LOG_MVB_DELTA0 && oc == 0 && s == 3
   && !(vx & OD_MVB_MASK) && !(vy & OD_MVB_MASK)) {
    int32_t v;
    v = mv_top[(vy >> OD_LOG_MVB_DELTA0)*mv_top_nh + (vx >> OD_LOG_MVB_DELTA0)];
    if (od_hipenc_check_mode() && v != od_mv_est_sad_cpu(est, vx, vy, oc, s, log_mvb_sz)) od_hipenc_mv_check_fail(1);
    return v;
  }
  return od_mv_est_sad_cpu(est, vx, vy, oc, s, log_mvb_sz);
}

static void od_mv_est_init_dus(od_mv_est_ctx *est) {
  double t0;
  double dt;
  double keep;
  keep = mv_inner;
  mv_inner = 0;
  t0 = od_hipenc_now();
  od_mv_est_init_dus_cpu(est);
  mv_top_nh = 0;                 /* the grid moves again from here on */
  dt = od_hipenc_now() - t0;
  od_hipenc_mv_stage(OD_HIPENC_MV_INIT_DUS, dt - mv_inner);
  mv_inner = keep + dt;
}

static void od_mv_est_decimate(od_mv_est_ctx *est) {
  double t0;
  double dt;
  double keep;
  keep = mv_inner;
  mv_inner = 0;
  t0 = od_hipenc_now();
  od_mv_est_decimate_cpu(est);
  dt = od_hipenc_now() - t0;
  od_hipenc_mv_stage(OD_HIPENC_MV_DECIMATE, dt - mv_inner);
  mv_inner = keep + dt;
}

void od_mv_subpel_refine(od_mv_est_ctx *est, int cost_thresh) {
  double t0;
  t0 = od_hipenc_now();
  od_mv_subpel_refine_cpu(est, cost_thresh);
  od_hipenc_mv_stage(OD_HIPENC_MV_SUBPEL, od_hipenc_now() - t0);
}

void od_mv_est(od_mv_est_ctx *est, int lambda) {
  double t0;
  t0 = od_hipenc_now();
  od_mv_est_cpu(est, lambda);
  od_hipenc_mv_stage(OD_HIPENC_MV_TOTAL, od_hipenc_now() - t0);
}
