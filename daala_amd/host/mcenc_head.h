/* Force-included (-include) ahead of the reference's src/mcenc.c in the HIP build.  The build
 * recipe keeps the reference's definitions of the motion search's stage functions under *_cpu
 * names (od_mv_est_init_mvs :3036, od_mv_est_bma_sad :2228, od_mv_est_sad :2271, od_mv_est_calc_sads :3761, od_mv_est_init_dus :3970,
 * od_mv_est_decimate :4024, od_mv_subpel_refine :6325, od_mv_est :6390), so their call sites
 * bind to the definitions appended by mcenc_tail.c.  Only declarations: the struct tag stands
 * for the typedef mcenc.c's own includes provide later. */
#include <stdint.h>
struct od_mv_est_ctx;
static int32_t od_mv_est_sad(struct od_mv_est_ctx *est, int vx, int vy, int oc, int s, int log_mvb_sz);
static int32_t od_mv_est_bma_sad(struct od_mv_est_ctx *est, int ref, int bx, int by, int mvx, int mvy,
 int log_mvb_sz);
static void od_mv_est_init_mvs(struct od_mv_est_ctx *est, int ref, int must_update);
static void od_mv_est_calc_sads(struct od_mv_est_ctx *est);
static void od_mv_est_init_dus(struct od_mv_est_ctx *est);
static void od_mv_est_decimate(struct od_mv_est_ctx *est);
void od_mv_subpel_refine(struct od_mv_est_ctx *est, int cost_thresh);
