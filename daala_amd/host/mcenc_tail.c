/* Appended to the reference's src/mcenc.c in the HIP build (Makefile: sed | cat | gcc).
 * The stage functions of od_mv_est (src/mcenc.c:6390) run as the reference wrote them
 * (*_cpu); the definitions here are what their call sites bind to:
 *   - every stage is timed (a handful of clock reads per frame) into the worker's statistics
 *     (od_hipenc_stats.mv_stage_s), which is where profiles/r04*_mvest_stages*.md come from;
 *   - od_mv_est_init_mvs walks levels >= 1 over the whole grid and od_mv_est_bma_sad answers from
 *     device block-matching windows (first part of this file);
 *   - od_mv_est_calc_sads is one fused OBMC + SAD device call per frame, and od_mv_est_sad answers
 *     the top-level blocks of od_mv_est_init_dus from the same call (second part).
 * The DP refinement (od_mv_est_refine, od_mv_subpel_refine) is the reference's code, timed only. */
#include "../../include/daala_hip.h"
void od_hipenc_mv_stage(int stage, double seconds);   /* hip_enc_glue.c */
void od_hipenc_mv_check_fail(long n);
int od_hipenc_check_mode(void);
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

/* ---------------------------------------------------------------------------------------------
 * od_mv_est_init_mvs (src/mcenc.c:3036) with the block-matching SADs of levels >= 1 from the
 * device.  od_mv_est_init_mv (:2511), the reference's own function, keeps making every decision;
 * what changes is (a) the ORDER in which the vertices of the grid are visited and (b) where
 * od_mv_est_bma_sad's numbers come from.
 * (a) The reference walks motion vector blocks and, inside each, levels ("for cache coherency");
 *     its comment (:3052-3062) states the contract: a level needs the levels below it; level 0
 *     needs raster order; "order within a level does not matter".  Every predictor a level >= 1
 *     vertex reads (od_state_get_predictor, od_mc_get_ref_predictor, the cneighbors /
 *     pneighbors of od_mv_est_init_mv) is a vertex of a lower level inside its own block - the
 *     even-level vertices on a block's right / bottom edge explicitly drop the neighbours that
 *     belong to the next block (:2660-2663) - or the previous frame's vectors.  So the grid is
 *     walked level by level here: level 0 exactly as the reference does, then every vertex of
 *     level 1, of level 2, ...
 * (b) Before a level starts, the median predictor of each of its vertices - the centre of the
 *     candidates od_mv_est_init_mv will ask for - is known.  ONE device call evaluates the whole
 *     (2R + 1)^2 window of half-sample vectors around every vertex's centre
 *     (od_hip_mc_bma_windows); od_mv_est_bma_sad is rebound and answers from the vertex's window
 *     when the vector lies inside it, and runs the reference's code when it does not.  The table
 *     is indexed by the absolute vector: a wrong guess of the centre costs hits, never bits.
 * Check mode compares every table answer with the reference's function (mv_check_fail). */
int od_hipenc_mv_bma_windows(daala_enc_ctx *enc, int nplanes, const od_hip_mc_bma_rec *recs, int nrec,
 int radius, int32_t *out);
void od_hipenc_mv_bma_stats(long hits, long misses);

#define MV_WIN_RADIUS (4)
#define MV_WIN_W (2*MV_WIN_RADIUS + 1)

static __thread struct {
  int active;
  int ref;
  int bx;
  int by;
  int log_mvb_sz;
  int cx;
  int cy;
  const int32_t *win;
  int cx2;                  /* second window (around the coarser neighbour's vector), win2 == NULL: none */
  int cy2;
  const int32_t *win2;
  long hits;
  long misses;
} mv_win;

static __thread od_hip_mc_bma_rec *mv_recs;   /* up to two records per vertex */
static __thread int32_t *mv_wins;
static __thread int (*mv_verts)[2];
static __thread int *mv_rec2;                  /* vertex -> index of its second record, or -1 */
static __thread int mv_recs_cap;

static int32_t od_mv_est_bma_sad(od_mv_est_ctx *est, int ref, int bx, int by, int mvx, int mvy,
 int log_mvb_sz) {
  if (mv_win.active && ref == mv_win.ref && bx == mv_win.bx && by == mv_win.by
   && log_mvb_sz == mv_win.log_mvb_sz) {
    int w;
    for (w = 0; w < 2; w++) {
      const int32_t *win;
      int dx;
      int dy;
      win = w == 0 ? mv_win.win : mv_win.win2;
      if (win == NULL) continue;
      dx = mvx - (w == 0 ? mv_win.cx : mv_win.cx2);
      dy = mvy - (w == 0 ? mv_win.cy : mv_win.cy2);
      if (dx >= -MV_WIN_RADIUS && dx <= MV_WIN_RADIUS && dy >= -MV_WIN_RADIUS && dy <= MV_WIN_RADIUS) {
        int32_t v;
        v = win[(dy + MV_WIN_RADIUS)*MV_WIN_W + dx + MV_WIN_RADIUS];
        if (v >= 0) {
          mv_win.hits++;
          if (od_hipenc_check_mode() && v != od_mv_est_bma_sad_cpu(est, ref, bx, by, mvx, mvy, log_mvb_sz)) {
            od_hipenc_mv_check_fail(1);
          }
          return v;
        }
      }
    }
    mv_win.misses++;
  }
  return od_mv_est_bma_sad_cpu(est, ref, bx, by, mvx, mvy, log_mvb_sz);
}

/* the vertices of one level (n of them, in mv_verts): windows from the device, then the
   reference's od_mv_est_init_mv for each */
static int mv_level_with_windows(od_mv_est_ctx *est, int ref, int must_update, int n) {
  od_state *state;
  int nplanes;
  int nrec;
  int i;
  int rc;
  state = &est->enc->state;
  nplanes = (est->flags & OD_MC_USE_CHROMA) ? est->enc->input_img[est->enc->curr_frame].nplanes : 1;
  nrec = n;
  for (i = 0; i < n; i++) {
    od_hip_mc_bma_rec *r;
    od_mv_limits limits;
    int pred[2];
    int vx;
    int vy;
    int level;
    int log_mvb_sz;
    int mvb_sz;
    vx = mv_verts[i][0];
    vy = mv_verts[i][1];
    level = OD_MC_LEVEL[vy & OD_MVB_MASK][vx & OD_MVB_MASK];
    log_mvb_sz = (OD_MC_LEVEL_MAX - level) >> 1;
    mvb_sz = 1 << log_mvb_sz;
    r = mv_recs + i;
    /* src/mcenc.c:2585-2616: the limits in half samples, the block centred on the vertex, the
       clamped median predictor */
    od_mv_est_limits(state, &limits, vx, vy, log_mvb_sz + OD_LOG_MVBSIZE_MIN);
    r->xmin = limits.xmin*2;
    r->xmax = limits.xmax*2;
    r->ymin = limits.ymin*2;
    r->ymax = limits.ymax*2;
    r->bx = (vx << OD_LOG_MVBSIZE_MIN) - (mvb_sz << (OD_LOG_MVBSIZE_MIN - 1));
    r->by = (vy << OD_LOG_MVBSIZE_MIN) - (mvb_sz << (OD_LOG_MVBSIZE_MIN - 1));
    r->log_blk_sz = log_mvb_sz + OD_LOG_MVBSIZE_MIN;
    r->ref = state->ref_imgi[ref];
    od_state_get_predictor(state, pred, vx, vy, level, 2, ref);
    r->cx = OD_CLAMPI(r->xmin, pred[0], r->xmax);
    r->cy = OD_CLAMPI(r->ymin, pred[1], r->ymax);
    /* A second window where the search usually ends when it does not end at the predictor: around
       the vector block matching found for the first Set B neighbour (:2620-2660: a vertex of a
       coarser level, done before this level started) against this reference.  The golden reference's
       searches end there more often than at their predictor (which follows the vectors the grid
       holds, mostly of the previous reference).  Only when the first window does not cover it. */
    mv_rec2[i] = -1;
    if (state->frame_type == OD_P_FRAME) {
      const od_mv_node *nb;
      int cx2;
      int cy2;
      nb = NULL;
      if (level & 1) nb = est->mvs[vy - mvb_sz] + vx - mvb_sz;
      else if (vy >= mvb_sz) nb = est->mvs[vy - mvb_sz] + vx;
      cx2 = OD_CLAMPI(r->xmin, nb != NULL ? nb->bma_mvs[0][ref][0] : 0, r->xmax);
      cy2 = OD_CLAMPI(r->ymin, nb != NULL ? nb->bma_mvs[0][ref][1] : 0, r->ymax);
      if (abs(cx2 - r->cx) > MV_WIN_RADIUS/2 || abs(cy2 - r->cy) > MV_WIN_RADIUS/2) {
        mv_recs[nrec] = *r;
        mv_recs[nrec].cx = cx2;
        mv_recs[nrec].cy = cy2;
        mv_rec2[i] = nrec++;
      }
    }
  }
  rc = od_hipenc_mv_bma_windows(est->enc, nplanes, mv_recs, nrec, MV_WIN_RADIUS, mv_wins);
  for (i = 0; i < n; i++) {
    if (rc > 0) {
      mv_win.active = 1;
      mv_win.ref = ref;
      mv_win.bx = mv_recs[i].bx;
      mv_win.by = mv_recs[i].by;
      mv_win.log_mvb_sz = mv_recs[i].log_blk_sz - OD_LOG_MVBSIZE_MIN;
      mv_win.cx = mv_recs[i].cx;
      mv_win.cy = mv_recs[i].cy;
      mv_win.win = mv_wins + (size_t)i*MV_WIN_W*MV_WIN_W;
      mv_win.win2 = NULL;
      if (mv_rec2[i] >= 0) {
        mv_win.cx2 = mv_recs[mv_rec2[i]].cx;
        mv_win.cy2 = mv_recs[mv_rec2[i]].cy;
        mv_win.win2 = mv_wins + (size_t)mv_rec2[i]*MV_WIN_W*MV_WIN_W;
      }
    }
    od_mv_est_init_mv(est, ref, mv_verts[i][0], mv_verts[i][1], must_update);
    mv_win.active = 0;
  }
  return rc;
}

static int mv_init_mvs_levels(od_mv_est_ctx *est, int ref, int must_update) {
  od_state *state;
  int nhmvbs;
  int nvmvbs;
  int vx;
  int vy;
  int log_mvb_sz;
  int level;
  int cap;
  state = &est->enc->state;
  nhmvbs = state->nhmvbs;
  nvmvbs = state->nvmvbs;
  if (state->frame_type != OD_P_FRAME || ref == OD_FRAME_NEXT || state->ref_imgi[ref] < 0
   || est->level_max < 1 || !od_hipenc_mv_bma_windows(est->enc, 0, NULL, 0, MV_WIN_RADIUS, NULL)) {
    return 0;
  }
  cap = (nhmvbs + 1)*(nvmvbs + 1);
  if (cap > mv_recs_cap) {
    free(mv_recs);
    free(mv_wins);
    free(mv_verts);
    free(mv_rec2);
    mv_recs = (od_hip_mc_bma_rec *)malloc(sizeof(*mv_recs)*2*cap);
    mv_wins = (int32_t *)malloc(sizeof(*mv_wins)*2*cap*MV_WIN_W*MV_WIN_W);
    mv_verts = (int (*)[2])malloc(sizeof(*mv_verts)*cap);
    mv_rec2 = (int *)malloc(sizeof(*mv_rec2)*cap);
    mv_recs_cap = mv_recs != NULL && mv_wins != NULL && mv_verts != NULL && mv_rec2 != NULL ? cap : 0;
    if (mv_recs_cap == 0) return 0;
  }
  /* "Move the motion vector predictors back a frame." (:3045-3056) */
  if (ref == OD_FRAME_PREV) {
    OD_MOVE(est->bma_history_time + 1, est->bma_history_time + 0, 2);
    est->bma_history_time[0] = est->enc->curr_display_order;
    for (vy = 0; vy <= nvmvbs; vy++) {
      for (vx = 0; vx <= nhmvbs; vx++) {
        od_mv_node *mv;
        mv = est->mvs[vy] + vx;
        OD_MOVE(mv->bma_mvs + 1, mv->bma_mvs + 0, 2);
      }
    }
  }
  /* level 0: raster order, as the reference visits it (:3069-3076), SADs from the host */
  for (vx = 0; vx <= nhmvbs; vx += OD_MVB_DELTA0) od_mv_est_init_mv(est, ref, vx, 0, must_update);
  for (vy = 0; vy < nvmvbs; vy += OD_MVB_DELTA0) {
    od_mv_est_init_mv(est, ref, 0, vy + OD_MVB_DELTA0, must_update);
    for (vx = 0; vx < nhmvbs; vx += OD_MVB_DELTA0) {
      od_mv_est_init_mv(est, ref, vx + OD_MVB_DELTA0, vy + OD_MVB_DELTA0, must_update);
    }
  }
  /* the other levels, each over the whole grid: the vertex sets of :3078-3111 per block */
  for (log_mvb_sz = OD_LOG_MVB_DELTA0, level = 1; log_mvb_sz-- > 0 && est->level_max >= level; level++) {
    int mvb_sz;
    int n;
    int cx;
    int cy;
    mvb_sz = 1 << log_mvb_sz;
    /* odd level */
    n = 0;
    for (vy = 0; vy < nvmvbs; vy += OD_MVB_DELTA0) {
      for (vx = 0; vx < nhmvbs; vx += OD_MVB_DELTA0) {
        for (cy = vy + mvb_sz; cy < vy + OD_MVB_DELTA0; cy += 2*mvb_sz) {
          for (cx = vx + mvb_sz; cx < vx + OD_MVB_DELTA0; cx += 2*mvb_sz) {
            mv_verts[n][0] = cx;
            mv_verts[n][1] = cy;
            n++;
          }
        }
      }
    }
    if (mv_level_with_windows(est, ref, must_update, n) < 0) return -1;
    level++;
    if (est->level_max < level) break;
    /* even level (the quincunx of :3095-3109; a block owns its bottom and right edges, the frame's
       first row and column of blocks their top and left edges too) */
    n = 0;
    for (vy = 0; vy < nvmvbs; vy += OD_MVB_DELTA0) {
      for (vx = 0; vx < nhmvbs; vx += OD_MVB_DELTA0) {
        for (cy = vy + mvb_sz*!!vy; cy <= vy + OD_MVB_DELTA0; cy += mvb_sz) {
          for (cx = vx + (cy & mvb_sz ? 2*mvb_sz*!!vx : mvb_sz); cx <= vx + OD_MVB_DELTA0; cx += 2*mvb_sz) {
            mv_verts[n][0] = cx;
            mv_verts[n][1] = cy;
            n++;
          }
        }
      }
    }
    if (mv_level_with_windows(est, ref, must_update, n) < 0) return -1;
  }
  od_hipenc_mv_bma_stats(mv_win.hits, mv_win.misses);
  mv_win.hits = mv_win.misses = 0;
  return 1;
}

static void od_mv_est_init_mvs(od_mv_est_ctx *est, int ref, int must_update) {
  double t0;
  t0 = od_hipenc_now();
  /* < 0: a device stage failed mid-way (the frame is marked failed); the grid is left as the
     completed levels made it - the frame's result is discarded by the caller */
  if (mv_init_mvs_levels(est, ref, must_update) == 0) od_mv_est_init_mvs_cpu(est, ref, must_update);
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
int od_hipenc_mv_sad_items(daala_enc_ctx *enc, int nplanes, const od_hip_mc_sad_item *items,
 int nitems, int32_t *sad);

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
  free(mv_recs);
  free(mv_wins);
  free(mv_verts);
  free(mv_rec2);
  mv_recs = NULL;
  mv_wins = NULL;
  mv_verts = NULL;
  mv_rec2 = NULL;
  mv_recs_cap = 0;
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
