// C-ABI implementation of include/daala_hip.h (gfx950 only).
// Host-side plumbing: device buffers, launches, timing events.  No CPU fallback:
// every compute entry point fails with OD_HIP_ENODEV when HIP is unusable.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/daala_hip.h"
#include "coding_order_tables.h"
#include "pvq_kernels.hpp"
#include "pvq_theta_kernels.hpp"
#include "pvq_pfeed_kernels.hpp"
#include "mc_kernels.hpp"
#include "xform_kernels.hpp"
#include "xform_rt_kernels.hpp"
#include "tail_kernels.hpp"
#include "haar_kernels.hpp"

static_assert(sizeof(PvqBandRec) == sizeof(od_hip_pvq_band), "record layout");
static_assert(sizeof(od_hip_pvq_band) == 72, "record layout");
static_assert(sizeof(PvqThetaOut) == sizeof(od_hip_pvq_theta_out), "record layout");

namespace {

thread_local std::string g_err;

int fail(int code, const char *what, hipError_t e = hipSuccess) {
  g_err = what;
  if (e != hipSuccess) {
    g_err += ": ";
    g_err += hipGetErrorString(e);
  }
  return code;
}

#define HIPCHK(expr)                                         \
  do {                                                       \
    hipError_t e_ = (expr);                                  \
    if (e_ != hipSuccess) return fail(OD_HIP_ENODEV, #expr, e_); \
  } while (0)

int ensure_device() {
  static int state = 0;     // 0 unknown, 1 ok, -1 none
  if (state == 0) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    state = (e == hipSuccess && n > 0) ? 1 : -1;
  }
  if (state < 0) return fail(OD_HIP_ENODEV, "no HIP device available");
  return 0;
}

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    HIPCHK(hipMalloc(&p, bytes));
    cap = bytes;
    return 0;
  }
};

// Scratch for the host-pointer entry points (sections 1, 2 of the header): one set per
// process, so those entry points take g_scratch_mu for their whole round trip (they are
// parity/drop-in paths, not the fast path; the context objects carry their own buffers).
DevBuf g_in, g_out, g_aux0, g_aux1, g_aux2, g_aux3, g_aux4, g_aux5, g_aux6;
std::recursive_mutex g_scratch_mu;
#define SCRATCH_LOCK std::lock_guard<std::recursive_mutex> scratch_lock_(g_scratch_mu)

template <int N, bool INV>
int launch_dct_blocks(int32_t *out, const int32_t *in, int nblocks, hipStream_t s) {
  constexpr int BPW = 256/N;
  int grid = (nblocks + BPW - 1)/BPW;
  hipLaunchKernelGGL((k_dct_blocks<N, INV>), dim3(grid), dim3(256), 0, s, out, in, nblocks);
  HIPCHK(hipGetLastError());
  return 0;
}

int dct_blocks_dev(int bs, bool inv, int32_t *out, const int32_t *in, int nblocks,
                   hipStream_t s) {
  switch (bs*2 + (inv ? 1 : 0)) {
    case 0: return launch_dct_blocks<4, false>(out, in, nblocks, s);
    case 1: return launch_dct_blocks<4, true>(out, in, nblocks, s);
    case 2: return launch_dct_blocks<8, false>(out, in, nblocks, s);
    case 3: return launch_dct_blocks<8, true>(out, in, nblocks, s);
    case 4: return launch_dct_blocks<16, false>(out, in, nblocks, s);
    case 5: return launch_dct_blocks<16, true>(out, in, nblocks, s);
    case 6: return launch_dct_blocks<32, false>(out, in, nblocks, s);
    case 7: return launch_dct_blocks<32, true>(out, in, nblocks, s);
  }
  return fail(OD_HIP_EINVAL, "block size out of range");
}

int dct_blocks_host(int bs, bool inv, od_coeff *out, const od_coeff *in, int nblocks) {
  if (!out || !in) return fail(OD_HIP_EFAULT, "null pointer");
  if (bs < 0 || bs >= OD_HIP_NBSIZES || nblocks < 0) return fail(OD_HIP_EINVAL, "bad bs/nblocks");
  if (int rc = ensure_device()) return rc;
  if (nblocks == 0) return 0;
  size_t n = 4u << bs, bytes = (size_t)nblocks*n*n*sizeof(int32_t);
  SCRATCH_LOCK;
  if (int rc = g_in.reserve(bytes)) return rc;
  if (int rc = g_out.reserve(bytes)) return rc;
  HIPCHK(hipMemcpy(g_in.p, in, bytes, hipMemcpyHostToDevice));
  if (int rc = dct_blocks_dev(bs, inv, (int32_t *)g_out.p, (const int32_t *)g_in.p, nblocks, 0))
    return rc;
  HIPCHK(hipMemcpy(out, g_out.p, bytes, hipMemcpyDeviceToHost));
  return 0;
}

// vtable-signature single block: gather the strided block, run, scatter.
void dct_one(int bs, bool inv, od_coeff *out, int ostride, const od_coeff *in, int istride) {
  int n = 4 << bs;
  od_coeff tmp[32*32];
  for (int i = 0; i < n; i++) memcpy(tmp + i*n, in + (size_t)i*istride, n*sizeof(od_coeff));
  int rc = dct_blocks_host(bs, inv, tmp, tmp, 1);
  if (rc != 0) {
    fprintf(stderr, "daala_hip: fatal: %s\n", g_err.c_str());
    abort();
  }
  for (int i = 0; i < n; i++) memcpy(out + (size_t)i*ostride, tmp + i*n, n*sizeof(od_coeff));
}

}  // namespace

// ---------------------------------------------------------------------------
// Layout of one (plane, level) inside the plane's per-slot arenas (byte offsets from the
// slot's start), see PvqSoA in pvq_kernels.hpp.
struct PvqLevelLayout {
  int n = 0, bs = 0, nb = 0, nblk = 0, ncoded = 0, off[11] = {};
  size_t nrec = 0, ny = 0;                   // records per frame; int16 pulse entries per frame
  size_t o_cd = 0, o_qg = 0, o_k = 0, o_nc = 0, o_y = 0;   // out arena
  size_t o_g = 0;                            // g arena
  size_t o_cg = 0, o_perm = 0;               // in arena
};
struct PvqArena {
  PvqLevelLayout lev[4];
  size_t out_slot = 0, g_slot = 0, in_slot = 0;   // bytes per frame slot
  char *out = nullptr, *g = nullptr, *in = nullptr;
  double *dist = nullptr;                    // [slot][level...] not transferred
  size_t dist_slot = 0, o_dist[4] = {0, 0, 0, 0};
};

inline int pvq_yo(const int *off, int b) { return b == 0 ? 0 : off[b]; }
inline int pvq_ns(const int *off, int b) { return ((off[b + 1] - off[b]) + 1) & ~1; }

struct od_hip_ctx {
  od_hip_geometry geo;
  int device;
  hipStream_t stream;
  int nhsb, nvsb;
  int pw[OD_HIP_NPLANES_MAX], ph[OD_HIP_NPLANES_MAX], nlev[OD_HIP_NPLANES_MAX];
  size_t psz[OD_HIP_NPLANES_MAX];            // samples per plane
  uint8_t *pix[OD_HIP_NPLANES_MAX];          // [slot][h][w]
  int32_t *lev[OD_HIP_NPLANES_MAX];          // [slot][level][h][w]
  int32_t *d[OD_HIP_NPLANES_MAX];            // [slot][h][w]
  int32_t *c[OD_HIP_NPLANES_MAX];            // [slot][h][w]
  uint8_t *rec[OD_HIP_NPLANES_MAX];          // [slot][h][w]
  uint8_t *bsize;                            // [slot][nvsb*4][nhsb*4]
  int32_t *p32[OD_HIP_NPLANES_MAX];          // [slot][h][w] post-filtered planes (decoder tail)
  int16_t *rs[OD_HIP_NPLANES_MAX] = {nullptr, nullptr, nullptr, nullptr};   // fused inverse: row strips (lazy)
  int16_t *cs[OD_HIP_NPLANES_MAX] = {nullptr, nullptr, nullptr, nullptr};   // fused inverse: column strips
  uint8_t *dflags;                           // [slot][nvsb*nhsb] dering flags
  uint8_t *bskip[OD_HIP_NPLANES_MAX];        // [slot][(fh/4)*(fw/4)]
  size_t bsize_sz;
  uint16_t *tab[OD_HIP_NBSIZES];             // coding tables on device
  int16_t *qm_dev;                           // scratch QM (1024 int16)
  double *rsq;                               // 1/sqrt(i) table (pvq_rsqrt_tab)
  // The PVQ launches of a step (35 of them, all independent: they only read pyramid
  // levels) go round-robin over a few side streams so that the tail of one kernel - a
  // few waves with large K - overlaps the next kernels instead of idling the chip.
  // They are joined back into `stream` lazily, before anything else touches the context.
  static constexpr int NAUX = 16;
  hipStream_t aux[NAUX] = {};
  hipEvent_t aux_done[NAUX] = {};
  hipEvent_t aux_dep = nullptr;
  int naux = 3;                              // OD_HIP_PVQ_STREAMS (0: everything on `stream`); 3 measured best
  int aux_rr = 0;
  bool aux_pending = false;
  hipEvent_t phase_a = nullptr;              // start of the PVQ batch in flight (timing only)
  int16_t *qm_slots = nullptr;               // [plane][level][1024]: one QM copy per (plane, level)
  // PVQ results per (plane, level): pointers of slot 0 into the plane's arenas
  PvqSoA pvq[OD_HIP_NPLANES_MAX][4];
  bool pvq_alloc[OD_HIP_NPLANES_MAX][4];     // the plane's arenas exist (all levels at once)
  PvqArena arena[OD_HIP_NPLANES_MAX];
  // timing
  struct Span { hipEvent_t a, b; };
  std::map<std::string, std::vector<Span>> spans;
  std::vector<hipEvent_t> pool;
  bool timing = false;
  // strip window (od_hip_set_strip): superblock rows [strip0, strip1) are computed by the
  // forward pyramid and the PVQ passes; the whole frame by default
  int strip0 = 0, strip1 = 0;
  unsigned long long *pvq_stats = nullptr;   // od_hip_pvq_stats: device work counters (measurement)
  struct StripCache *strips = nullptr;       // comm.hpp: segment lists and the staging buffer of strip transfers
  int *order_scratch = nullptr;              // k_pvq_order_*: [2][slot][band][256] key histograms and cursors
};
static void strip_cache_free(od_hip_ctx *ctx);   // comm.hpp

namespace {

hipEvent_t get_event(od_hip_ctx *ctx) {
  if (!ctx->pool.empty()) {
    hipEvent_t e = ctx->pool.back();
    ctx->pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

struct Timed {
  od_hip_ctx *ctx;
  const char *name;
  hipEvent_t a = nullptr;
  hipStream_t st;
  Timed(od_hip_ctx *c, const char *n, hipStream_t s = nullptr) : ctx(c), name(n), st(s ? s : c->stream) {
    if (ctx->timing) {
      a = get_event(ctx);
      if (a) (void)hipEventRecord(a, st);
    }
  }
  ~Timed() {
    if (a) {
      hipEvent_t b = get_event(ctx);
      if (b) {
        (void)hipEventRecord(b, st);
        ctx->spans[name].push_back({a, b});
      }
    }
  }
};

// Joins the side streams back into ctx->stream (device-side dependency, no host wait).
int join_aux(od_hip_ctx *ctx) {
  if (!ctx->aux_pending) return 0;
  for (int i = 0; i < ctx->naux; i++) {
    HIPCHK(hipEventRecord(ctx->aux_done[i], ctx->aux[i]));
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->aux_done[i], 0));
  }
  ctx->aux_pending = false;
  if (ctx->phase_a) {
    // wall time of the whole concurrent PVQ batch (the per-kernel spans overlap)
    hipEvent_t b = get_event(ctx);
    if (b) {
      (void)hipEventRecord(b, ctx->stream);
      ctx->spans["pvq_phase"].push_back({ctx->phase_a, b});
    }
    ctx->phase_a = nullptr;
  }
  return 0;
}

int check_slots(od_hip_ctx *ctx, int slot0, int nslots, bool join = true) {
  if (!ctx) return fail(OD_HIP_EFAULT, "null context");
  if (slot0 < 0 || nslots < 1 || slot0 + nslots > ctx->geo.nslots)
    return fail(OD_HIP_EINVAL, "slot range out of bounds");
  HIPCHK(hipSetDevice(ctx->device));
  if (join) return join_aux(ctx);
  return 0;
}

}  // namespace

namespace {
template <int N>
void launch_pvq(const PvqLevelArgs &a, int nlist, long nblk_unused, int nslots, hipStream_t s,
                bool gain_only, const double *rsq, unsigned long long *stats) {
  constexpr int BPW = PvqGeom<N>::BPW;
  const long nblk = a.blk_end - a.blk_first;
  if (nblk <= 0) return;
  (void)nblk_unused;
  PvqLevelArgs3 aa;
  aa.a = a;
  aa.rsq = rsq;
  aa.stats = stats;
  if (gain_only) {
    dim3 grid((unsigned)((nblk + BPW - 1)/BPW), nlist, nslots);
    hipLaunchKernelGGL((k_pvq_gain<N>), grid, dim3(64), 0, s, aa);
  }
  else {
    // one lane group per candidate: the band's work list has two entries per block
    dim3 grid((unsigned)((2*nblk + BPW - 1)/BPW), nlist, nslots);
    hipLaunchKernelGGL((k_pvq_cand<N>), grid, dim3(64), 0, s, aa);
  }
}

// od_gain_compand (src/pvq.c:422-425) with the HOST's libm: the value the reference
// computes in this very process (same expression, same pow).
inline double host_gain_compand(double g, int q0, double beta) {
  if (beta == 1) return g/q0;
  return PVQ_COMPAND_SCALE*pow(g*(1./PVQ_COMPAND_SCALE), 1./beta)/q0;
}

// an integer switch from the environment, read once by its caller (static const)
inline int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return e ? atoi(e) : dflt;
}

inline bool pvq_sort_enabled() {
  static const int on = env_int("OD_HIP_PVQ_SORT", 1);
  return on != 0;
}

// Work list of one band for the search kernel (performance only, any permutation of the
// entries is correct): entry 2*block + candidate for both gain candidates of every block,
// ordered by descending K (counting sort), so that the lane slots of a wave - which runs
// until its slowest slot is done - hold similar K, the long searches start first and the
// candidates that do not exist (K = 0) gather at the tail.  Candidates as pvq_theta's
// no-reference loop (src/pvq_encoder.c:457): i = max(1, floor(cg)) + c while i <= ceil(cg);
// K as od_pvq_compute_k (src/pvq.c:508-514).  perm: 2*count entries.
inline void pvq_block_order(const double *cg, long first, long count, int n, double beta, bool sort,
                            int32_t *perm) {
  if (!sort) {
    for (long i = 0; i < 2*count; i++) perm[i] = (int32_t)(2*first + i);
    return;
  }
  std::vector<uint8_t> key(2*count);
  long hist[257];
  for (int i = 0; i < 257; i++) hist[i] = 0;
  const double sq = sqrt((double)((n + 3)/2));
  for (long i = 0; i < count; i++) {
    const double lo = floor(cg[i]) < 1 ? 1 : floor(cg[i]), hi = ceil(cg[i]);
    for (int c = 0; c < 2; c++) {
      const double q = lo + c;
      int k = 0;
      if (q <= hi) {
        if (n == 15 && q == 1 && beta > 1.25) k = 1;
        else {
          const double v = floor(.5 + (q - .2)*sq/beta);
          k = v < 1 ? 1 : v > 255 ? 255 : (int)v;
        }
      }
      key[2*i + c] = (uint8_t)(255 - k);
      hist[key[2*i + c] + 1]++;
    }
  }
  for (int i = 0; i < 256; i++) hist[i + 1] += hist[i];
  for (long i = 0; i < 2*count; i++) perm[hist[key[i]]++] = (int32_t)(2*first + i);
}
}  // namespace

extern "C" {

const char *od_hip_last_error(void) { return g_err.c_str(); }

const char *od_hip_version(void) { return "daala_hip 0.1 (gfx950)"; }

int od_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void od_hip_bin_fdct4x4(od_coeff *y, int ys, const od_coeff *x, int xs) { dct_one(0, false, y, ys, x, xs); }
void od_hip_bin_fdct8x8(od_coeff *y, int ys, const od_coeff *x, int xs) { dct_one(1, false, y, ys, x, xs); }
void od_hip_bin_fdct16x16(od_coeff *y, int ys, const od_coeff *x, int xs) { dct_one(2, false, y, ys, x, xs); }
void od_hip_bin_fdct32x32(od_coeff *y, int ys, const od_coeff *x, int xs) { dct_one(3, false, y, ys, x, xs); }
void od_hip_bin_idct4x4(od_coeff *x, int xs, const od_coeff *y, int ys) { dct_one(0, true, x, xs, y, ys); }
void od_hip_bin_idct8x8(od_coeff *x, int xs, const od_coeff *y, int ys) { dct_one(1, true, x, xs, y, ys); }
void od_hip_bin_idct16x16(od_coeff *x, int xs, const od_coeff *y, int ys) { dct_one(2, true, x, xs, y, ys); }
void od_hip_bin_idct32x32(od_coeff *x, int xs, const od_coeff *y, int ys) { dct_one(3, true, x, xs, y, ys); }

int od_hip_vtbl_fill(od_dct_func_2d fdct_2d[OD_HIP_NBSIZES + 1],
                     od_dct_func_2d idct_2d[OD_HIP_NBSIZES + 1]) {
  if (!fdct_2d || !idct_2d) return fail(OD_HIP_EFAULT, "null vtable");
  if (int rc = ensure_device()) return rc;
  fdct_2d[0] = od_hip_bin_fdct4x4;
  fdct_2d[1] = od_hip_bin_fdct8x8;
  fdct_2d[2] = od_hip_bin_fdct16x16;
  fdct_2d[3] = od_hip_bin_fdct32x32;
  idct_2d[0] = od_hip_bin_idct4x4;
  idct_2d[1] = od_hip_bin_idct8x8;
  idct_2d[2] = od_hip_bin_idct16x16;
  idct_2d[3] = od_hip_bin_idct32x32;
  return 0;
}

int od_hip_fdct_blocks(int bs, od_coeff *out, const od_coeff *in, int nblocks) {
  return dct_blocks_host(bs, false, out, in, nblocks);
}

int od_hip_idct_blocks(int bs, od_coeff *out, const od_coeff *in, int nblocks) {
  return dct_blocks_host(bs, true, out, in, nblocks);
}

int od_hip_haar_blocks(int bs, int inverse, od_coeff *out, const od_coeff *in, int nblocks) {
  if (!out || !in) return fail(OD_HIP_EFAULT, "null pointer");
  if (bs < 0 || bs >= OD_HIP_NBSIZES || nblocks < 0) return fail(OD_HIP_EINVAL, "bad bs/nblocks");
  if (int rc = ensure_device()) return rc;
  if (nblocks == 0) return 0;
  size_t n = 4u << bs, bytes = (size_t)nblocks*n*n*sizeof(int32_t);
  SCRATCH_LOCK;
  if (int rc = g_in.reserve(bytes)) return rc;
  if (int rc = g_out.reserve(bytes)) return rc;
  HIPCHK(hipMemcpy(g_in.p, in, bytes, hipMemcpyHostToDevice));
  dim3 grid((nblocks + 63)/64), blk(64);
  int32_t *o = (int32_t *)g_out.p;
  const int32_t *i = (const int32_t *)g_in.p;
  switch (bs) {
    case 0: hipLaunchKernelGGL(k_haar_blocks<2>, grid, blk, 0, 0, o, i, nblocks, inverse); break;
    case 1: hipLaunchKernelGGL(k_haar_blocks<3>, grid, blk, 0, 0, o, i, nblocks, inverse); break;
    case 2: hipLaunchKernelGGL(k_haar_blocks<4>, grid, blk, 0, 0, o, i, nblocks, inverse); break;
    default: hipLaunchKernelGGL(k_haar_blocks<5>, grid, blk, 0, 0, o, i, nblocks, inverse); break;
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, g_out.p, bytes, hipMemcpyDeviceToHost));
  return 0;
}

namespace {
// blocks of one size class form one launch (mc_kernels.hpp: McTiles); prediction blocks are
// independent and may be reordered, so the list is bucketed by class first: at most three launches
inline int mc_block_log(const McBlock &b) { return b.log_xblk_sz > b.log_yblk_sz ? b.log_xblk_sz : b.log_yblk_sz; }
void mc_bucket_blocks(McBlock *dst, const od_hip_mc_block *src, int n) {
  const McBlock *in = reinterpret_cast<const McBlock *>(src);
  int at = 0;
  for (int cls = 4; cls <= 6; cls++) {
    for (int i = 0; i < n; i++) if (mc_size_class(mc_block_log(in[i])) == cls) dst[at++] = in[i];
  }
}
void mc_launch_predict(const McArgs &a0, const McBlock *host_sorted, int n, hipStream_t st) {
  mc_launch_runs(n, [&](int i) { return mc_block_log(host_sorted[i]); }, [&](int cls, int first, int cnt) {
    McArgs a = a0;
    a.blocks = a0.blocks + first;
    a.nblocks = cnt;
    if (cls == 4) hipLaunchKernelGGL(k_mc_predict_blocks<4>, dim3(cnt), dim3(MC_THREADS), 0, st, a);
    else if (cls == 5) hipLaunchKernelGGL(k_mc_predict_blocks<5>, dim3(cnt), dim3(MC_THREADS), 0, st, a);
    else hipLaunchKernelGGL(k_mc_predict_blocks<6>, dim3(cnt), dim3(MC_THREADS), 0, st, a);
  });
}
void mc_launch_sad(const McSadArgs &a0, const McSadItem *host_items, int n, hipStream_t st) {
  mc_launch_runs(n, [&](int i) { return host_items[i].log_blk_sz; }, [&](int cls, int first, int cnt) {
    McSadArgs a = a0;
    a.items = a0.items + first;
    a.sad = a0.sad + first;
    a.nitems = cnt;
    if (cls == 4) hipLaunchKernelGGL(k_mc_sad_items<4>, dim3(cnt), dim3(MC_SAD_THREADS), 0, st, a);
    else if (cls == 5) hipLaunchKernelGGL(k_mc_sad_items<5>, dim3(cnt), dim3(MC_SAD_THREADS), 0, st, a);
    else hipLaunchKernelGGL(k_mc_sad_items<6>, dim3(cnt), dim3(MC_SAD_THREADS), 0, st, a);
  });
}
}  // namespace

// F3: OBMC prediction of a list of blocks (mc_kernels.hpp).
int od_hip_mc_predict_blocks(int nref, const unsigned char *const refs[], int ref_stride, int ref_h,
                             int org_x, int org_y, const od_hip_mc_block *blocks, int nblocks,
                             unsigned char *dst, int dst_stride, int dst_h) {
  if (!refs || !blocks || !dst) return fail(OD_HIP_EFAULT, "null pointer");
  if (nref < 1 || nref > 8 || ref_stride < 1 || ref_h < 1 || nblocks < 0 || dst_stride < 1 || dst_h < 1)
    return fail(OD_HIP_EINVAL, "bad geometry");
  for (int k = 0; k < nref; k++) if (!refs[k]) return fail(OD_HIP_EFAULT, "null reference plane");
  // operand shapes are checked on the host before anything is launched
  for (int b = 0; b < nblocks; b++) {
    const od_hip_mc_block &m = blocks[b];
    if (m.log_xblk_sz < 2 || m.log_xblk_sz > 6 || m.log_yblk_sz < 2 || m.log_yblk_sz > 6 || m.x < 0 || m.y < 0
        || m.x + (1 << m.log_xblk_sz) > dst_stride || m.y + (1 << m.log_yblk_sz) > dst_h
        || m.oc < 0 || m.oc > 3 || m.s < 0 || m.s > 3)
      return fail(OD_HIP_EINVAL, "bad prediction block");
    for (int k = 0; k < 4; k++) if (m.ref[k] < 0 || m.ref[k] >= nref) return fail(OD_HIP_EINVAL, "bad reference index");
  }
  if (int rc = ensure_device()) return rc;
  if (nblocks == 0) return 0;
  static_assert(sizeof(McBlock) == sizeof(od_hip_mc_block), "McBlock mirrors od_hip_mc_block");
  const size_t plane = (size_t)ref_stride*ref_h, dbytes = (size_t)dst_stride*dst_h;
  SCRATCH_LOCK;
  if (int rc = g_in.reserve(plane*nref)) return rc;
  if (int rc = g_aux0.reserve((size_t)nblocks*sizeof(McBlock))) return rc;
  if (int rc = g_out.reserve(dbytes)) return rc;
  for (int k = 0; k < nref; k++)
    HIPCHK(hipMemcpy((uint8_t *)g_in.p + plane*k, refs[k], plane, hipMemcpyHostToDevice));
  std::vector<McBlock> sorted((size_t)nblocks);
  mc_bucket_blocks(sorted.data(), blocks, nblocks);
  HIPCHK(hipMemcpy(g_aux0.p, sorted.data(), (size_t)nblocks*sizeof(McBlock), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_out.p, dst, dbytes, hipMemcpyHostToDevice));     // blocks need not cover the plane
  McArgs a;
  a.refs = (const uint8_t *)g_in.p;
  a.ref_plane = plane;
  a.ref_stride = ref_stride;
  a.ref_h = ref_h;
  a.org_x = org_x;
  a.org_y = org_y;
  a.blocks = (const McBlock *)g_aux0.p;
  a.nblocks = nblocks;
  a.dst = (uint8_t *)g_out.p;
  a.dst_stride = dst_stride;
  mc_launch_predict(a, sorted.data(), nblocks, 0);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(dst, g_out.p, dbytes, hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Streams of the per-worker objects (od_hip_mc, od_hip_dering).  A session has one such object
// per host worker (30 in the bench); with a stream each the process held 30+ streams and every
// launch issued afterwards - on ANY stream of the process - showed 60-100 us more between its
// bracketing events (profiles/r04_launch_gaps.md has the experiment).  The
// objects' device work is short and each call ends with a synchronisation, so they lease a
// stream from a small per-device pool instead (OD_HIP_STREAM_POOL streams, default 4;
// 0 = one stream per object as before).  Pooled streams live as long as the process.
namespace {
std::mutex g_spool_mu;
std::map<int, std::vector<hipStream_t>> g_spool;
unsigned g_spool_rr = 0;
hipStream_t lease_stream(int device, bool &owned) {
  static const int pool = [] { const char *e = getenv("OD_HIP_STREAM_POOL"); return e ? atoi(e) : 4; }();
  hipStream_t st = nullptr;
  if (pool <= 0) {
    owned = true;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return nullptr;
    return st;
  }
  owned = false;
  std::lock_guard<std::mutex> lk(g_spool_mu);
  auto &v = g_spool[device];
  if ((int)v.size() < pool) {
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return nullptr;
    v.push_back(st);
    return st;
  }
  return v[g_spool_rr++ % v.size()];
}
}  // namespace

// ---------------------------------------------------------------------------
// Resident motion-compensation object (one per host thread that predicts frames): own
// stream, page-locked staging, the reference frames of every plane resident in HBM - a
// reference image is uploaded when it changes, not once per predicted plane - block list and
// prediction buffers.  od_hip_mc_predict_blocks above stays as the stateless parity entry.
struct od_hip_mc {
  int device = 0, nref = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;          // false: leased from the per-device pool
  struct Plane {
    int ref_stride = 0, ref_h = 0, org_x = 0, org_y = 0;
    size_t plane = 0;                 // bytes of one reference plane
    uint8_t *d_refs = nullptr;        // [nref][ref_h][ref_stride]
    uint8_t *h_ref = nullptr;         // pinned staging, one plane
    uint8_t *d_dst = nullptr, *h_dst = nullptr;
    size_t dst_cap = 0;
  } pl[3];
  McBlock *d_blocks = nullptr, *h_blocks = nullptr;
  size_t blocks_cap = 0;
  hipEvent_t done = nullptr;        // od_hip_mc_predict_ctx: the prediction is in the context's plane
  hipEvent_t ctx_ready = nullptr;   // ... and: the context's stream has reached the call (its planes may be touched)
  // od_hip_mc_sad_items: the frame being coded (one dense plane each) and the item / result lists
  struct Src {
    uint8_t *d = nullptr, *h = nullptr;
    size_t cap = 0;
    int w = 0, h_rows = 0, xdec = 0, ydec = 0;
  } src[3];
  McSadItem *d_items = nullptr, *h_items = nullptr;
  int32_t *d_sad = nullptr, *h_sad = nullptr;
  size_t items_cap = 0;
  // od_hip_mc_bma_windows: vertex records and their SAD windows
  McBmaRec *d_recs = nullptr, *h_recs = nullptr;
  int32_t *d_win = nullptr, *h_win = nullptr;
  size_t recs_cap = 0, win_cap = 0;
};

extern "C" {

void od_hip_mc_destroy(od_hip_mc *m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  for (auto &p : m->pl) {
    if (p.d_refs) (void)hipFree(p.d_refs);
    if (p.h_ref) (void)hipHostFree(p.h_ref);
    if (p.d_dst) (void)hipFree(p.d_dst);
    if (p.h_dst) (void)hipHostFree(p.h_dst);
  }
  if (m->d_blocks) (void)hipFree(m->d_blocks);
  if (m->h_blocks) (void)hipHostFree(m->h_blocks);
  for (auto &q : m->src) {
    if (q.d) (void)hipFree(q.d);
    if (q.h) (void)hipHostFree(q.h);
  }
  if (m->d_items) (void)hipFree(m->d_items);
  if (m->h_items) (void)hipHostFree(m->h_items);
  if (m->d_sad) (void)hipFree(m->d_sad);
  if (m->h_sad) (void)hipHostFree(m->h_sad);
  if (m->d_recs) (void)hipFree(m->d_recs);
  if (m->h_recs) (void)hipHostFree(m->h_recs);
  if (m->d_win) (void)hipFree(m->d_win);
  if (m->h_win) (void)hipHostFree(m->h_win);
  if (m->done) (void)hipEventDestroy(m->done);
  if (m->ctx_ready) (void)hipEventDestroy(m->ctx_ready);
  if (m->stream && m->own_stream) (void)hipStreamDestroy(m->stream);
  delete m;
}

od_hip_mc *od_hip_mc_create(int device, int nref) {
  if (nref < 1 || nref > 8) { fail(OD_HIP_EINVAL, "bad reference count"); return nullptr; }
  if (ensure_device()) return nullptr;
  if (hipSetDevice(device) != hipSuccess) { fail(OD_HIP_ENODEV, "hipSetDevice failed"); return nullptr; }
  od_hip_mc *m = new od_hip_mc();
  m->device = device;
  m->nref = nref;
  m->stream = lease_stream(device, m->own_stream);
  if (!m->stream) {
    fail(OD_HIP_ENODEV, "stream creation failed");
    delete m;
    return nullptr;
  }
  return m;
}

int od_hip_mc_set_ref(od_hip_mc *m, int pli, int k, const unsigned char *plane, int ref_stride, int ref_h,
                      int org_x, int org_y) {
  if (!m || !plane) return fail(OD_HIP_EFAULT, "null pointer");
  if (pli < 0 || pli > 2 || k < 0 || k >= m->nref || ref_stride < 1 || ref_h < 1) return fail(OD_HIP_EINVAL, "bad reference plane");
  HIPCHK(hipSetDevice(m->device));
  auto &P = m->pl[pli];
  const size_t bytes = (size_t)ref_stride*ref_h;
  if (P.plane != bytes || P.ref_stride != ref_stride) {
    // geometry (re)defined: every reference of this plane has to be set again by the caller
    HIPCHK(hipStreamSynchronize(m->stream));
    if (P.d_refs) (void)hipFree(P.d_refs);
    if (P.h_ref) (void)hipHostFree(P.h_ref);
    P.d_refs = P.h_ref = nullptr;
    HIPCHK(hipMalloc((void **)&P.d_refs, bytes*m->nref));
    HIPCHK(hipHostMalloc((void **)&P.h_ref, bytes));
    P.plane = bytes;
    P.ref_stride = ref_stride;
    P.ref_h = ref_h;
  }
  P.org_x = org_x;
  P.org_y = org_y;
  HIPCHK(hipStreamSynchronize(m->stream));          // the staging plane may still be in flight
  memcpy(P.h_ref, plane, bytes);
  HIPCHK(hipMemcpyAsync(P.d_refs + bytes*k, P.h_ref, bytes, hipMemcpyHostToDevice, m->stream));
  return 0;
}

}  // extern "C"

namespace {
// One plane's prediction.  dev_dst == nullptr: into the object's own dense plane, then to the
// caller's host plane (dst).  dev_dst != nullptr: straight into that dense device plane (a
// context's picture plane), nothing comes to the host; the launch is left in flight on m->stream.
int mc_predict_impl(od_hip_mc *m, int pli, const od_hip_mc_block *blocks, int nblocks, unsigned char *dst,
                    int dst_stride, int dst_w, int dst_h, uint8_t *dev_dst) {
  if (!m || !blocks || (!dst && !dev_dst)) return fail(OD_HIP_EFAULT, "null pointer");
  if (pli < 0 || pli > 2 || nblocks < 0 || dst_stride < 1 || dst_w < 1 || dst_w > dst_stride || dst_h < 1)
    return fail(OD_HIP_EINVAL, "bad geometry");
  auto &P = m->pl[pli];
  if (!P.d_refs) return fail(OD_HIP_EINVAL, "no reference planes set for this plane");
  // operand shapes are checked on the host before anything is launched
  for (int b = 0; b < nblocks; b++) {
    const od_hip_mc_block &q = blocks[b];
    if (q.log_xblk_sz < 2 || q.log_xblk_sz > 6 || q.log_yblk_sz < 2 || q.log_yblk_sz > 6 || q.x < 0 || q.y < 0
        || q.x + (1 << q.log_xblk_sz) > dst_w || q.y + (1 << q.log_yblk_sz) > dst_h
        || q.oc < 0 || q.oc > 3 || q.s < 0 || q.s > 3)
      return fail(OD_HIP_EINVAL, "bad prediction block");
    for (int k = 0; k < 4; k++) if (q.ref[k] < 0 || q.ref[k] >= m->nref) return fail(OD_HIP_EINVAL, "bad reference index");
  }
  if (nblocks == 0) return 0;
  HIPCHK(hipSetDevice(m->device));
  const size_t dbytes = (size_t)dst_w*dst_h;        // dense picture area on the device
  if (!dev_dst && P.dst_cap < dbytes) {
    HIPCHK(hipStreamSynchronize(m->stream));
    if (P.d_dst) (void)hipFree(P.d_dst);
    if (P.h_dst) (void)hipHostFree(P.h_dst);
    P.d_dst = P.h_dst = nullptr;
    P.dst_cap = 0;
    HIPCHK(hipMalloc((void **)&P.d_dst, dbytes));
    HIPCHK(hipHostMalloc((void **)&P.h_dst, dbytes));
    P.dst_cap = dbytes;
  }
  if (m->blocks_cap < (size_t)nblocks) {
    HIPCHK(hipStreamSynchronize(m->stream));
    if (m->d_blocks) (void)hipFree(m->d_blocks);
    if (m->h_blocks) (void)hipHostFree(m->h_blocks);
    m->d_blocks = m->h_blocks = nullptr;
    m->blocks_cap = 0;
    const size_t cap = (size_t)nblocks*2;
    HIPCHK(hipMalloc((void **)&m->d_blocks, cap*sizeof(McBlock)));
    HIPCHK(hipHostMalloc((void **)&m->h_blocks, cap*sizeof(McBlock)));
    m->blocks_cap = cap;
  }
  HIPCHK(hipStreamSynchronize(m->stream));          // the previous plane's list may still be read
  mc_bucket_blocks(m->h_blocks, blocks, nblocks);
  HIPCHK(hipMemcpyAsync(m->d_blocks, m->h_blocks, (size_t)nblocks*sizeof(McBlock), hipMemcpyHostToDevice, m->stream));
  McArgs a;
  a.refs = P.d_refs;
  a.ref_plane = P.plane;
  a.ref_stride = P.ref_stride;
  a.ref_h = P.ref_h;
  a.org_x = P.org_x;
  a.org_y = P.org_y;
  a.blocks = m->d_blocks;
  a.nblocks = nblocks;
  a.dst = dev_dst ? dev_dst : P.d_dst;
  a.dst_stride = dst_w;
  mc_launch_predict(a, m->h_blocks, nblocks, m->stream);
  HIPCHK(hipGetLastError());
  if (dev_dst) return 0;
  HIPCHK(hipMemcpyAsync(P.h_dst, P.d_dst, dbytes, hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipStreamSynchronize(m->stream));
  // the blocks tile the picture area; only that area of the caller's plane is written
  for (int y = 0; y < dst_h; y++) memcpy(dst + (size_t)y*dst_stride, P.h_dst + (size_t)y*dst_w, dst_w);
  return 0;
}
}  // namespace

extern "C" {

int od_hip_mc_predict(od_hip_mc *m, int pli, const od_hip_mc_block *blocks, int nblocks, unsigned char *dst,
                      int dst_stride, int dst_w, int dst_h) {
  if (!dst) return fail(OD_HIP_EFAULT, "null pointer");
  return mc_predict_impl(m, pli, blocks, nblocks, dst, dst_stride, dst_w, dst_h, nullptr);
}

// The same prediction written into picture plane `pli` of slot `slot` of a context on the same
// device (the blocks must tile the context's frame_width >> xdec x frame_height >> ydec plane):
// what a decoder needs when the prediction's only consumer is the forward pyramid of that
// context - no copy to the host and back.  The context's stream waits for the prediction on the
// device; the call itself does not block.
static int check_plane(od_hip_ctx *ctx, int slot, int pli);
int od_hip_mc_predict_ctx(od_hip_mc *m, int pli, const od_hip_mc_block *blocks, int nblocks, od_hip_ctx *ctx,
                          int slot) {
  if (!m || !ctx) return fail(OD_HIP_EFAULT, "null pointer");
  if (int rc = check_plane(ctx, slot, pli)) return rc;
  if (ctx->device != m->device) return fail(OD_HIP_EINVAL, "prediction object and context live on different devices");
  long area = 0;
  for (int b = 0; blocks && b < nblocks; b++) area += 1L << (blocks[b].log_xblk_sz + blocks[b].log_yblk_sz);
  if (area != (long)ctx->pw[pli]*ctx->ph[pli]) return fail(OD_HIP_EINVAL, "the blocks do not tile the context's plane");
  // work still queued on the context's stream may read or write that picture plane (the forward
  // pyramid or the tail of the previous frame): the prediction is ordered behind it
  if (!m->ctx_ready) HIPCHK(hipEventCreateWithFlags(&m->ctx_ready, hipEventDisableTiming));
  HIPCHK(hipEventRecord(m->ctx_ready, ctx->stream));
  HIPCHK(hipStreamWaitEvent(m->stream, m->ctx_ready, 0));
  if (int rc = mc_predict_impl(m, pli, blocks, nblocks, nullptr, ctx->pw[pli], ctx->pw[pli], ctx->ph[pli],
                               ctx->pix[pli] + (size_t)slot*ctx->psz[pli])) return rc;
  if (!m->done) HIPCHK(hipEventCreateWithFlags(&m->done, hipEventDisableTiming));
  HIPCHK(hipEventRecord(m->done, m->stream));
  HIPCHK(hipStreamWaitEvent(ctx->stream, m->done, 0));
  return 0;
}

// Reference image k of plane pli taken from the reconstruction plane of a context on the same
// device (what od_hip_decode_tail / od_hip_inverse left in slot `slot`): the frame never visits
// the host on its way to becoming a reference.  Geometry as od_hip_mc_set_ref; the padding is
// filled on the device as od_img_edge_ext fills it (src/state.c:1100-1171).
int od_hip_mc_set_ref_ctx(od_hip_mc *m, int pli, int k, od_hip_ctx *ctx, int slot, int ref_stride, int ref_h,
                          int org_x, int org_y) {
  if (!m || !ctx) return fail(OD_HIP_EFAULT, "null pointer");
  if (int rc = check_plane(ctx, slot, pli)) return rc;
  if (ctx->device != m->device) return fail(OD_HIP_EINVAL, "prediction object and context live on different devices");
  if (pli > 2 || k < 0 || k >= m->nref || org_x < 0 || org_y < 0 || ref_stride < org_x + ctx->pw[pli]
      || ref_h < org_y + ctx->ph[pli]) return fail(OD_HIP_EINVAL, "bad reference plane");
  HIPCHK(hipSetDevice(m->device));
  auto &P = m->pl[pli];
  const size_t bytes = (size_t)ref_stride*ref_h;
  if (P.plane != bytes || P.ref_stride != ref_stride) {
    // geometry (re)defined: every reference of this plane has to be set again by the caller
    HIPCHK(hipStreamSynchronize(m->stream));
    if (P.d_refs) (void)hipFree(P.d_refs);
    if (P.h_ref) (void)hipHostFree(P.h_ref);
    P.d_refs = P.h_ref = nullptr;
    HIPCHK(hipMalloc((void **)&P.d_refs, bytes*m->nref));
    HIPCHK(hipHostMalloc((void **)&P.h_ref, bytes));
    P.plane = bytes;
    P.ref_stride = ref_stride;
    P.ref_h = ref_h;
  }
  P.org_x = org_x;
  P.org_y = org_y;
  // the reconstruction is produced on the context's stream
  if (!m->ctx_ready) HIPCHK(hipEventCreateWithFlags(&m->ctx_ready, hipEventDisableTiming));
  HIPCHK(hipEventRecord(m->ctx_ready, ctx->stream));
  HIPCHK(hipStreamWaitEvent(m->stream, m->ctx_ready, 0));
  McRefArgs a;
  a.rec = ctx->rec[pli] + (size_t)slot*ctx->psz[pli];
  a.pw = ctx->pw[pli];
  a.ph = ctx->ph[pli];
  a.ref = P.d_refs + bytes*k;
  a.ref_stride = ref_stride;
  a.ref_h = ref_h;
  a.org_x = org_x;
  a.org_y = org_y;
  hipLaunchKernelGGL(k_mc_ref_from_rec, dim3((ref_stride/4 + 256)/256, ref_h), dim3(256), 0, m->stream, a);
  HIPCHK(hipGetLastError());
  // the context may overwrite its reconstruction plane with the next frame: that waits for the copy
  if (!m->done) HIPCHK(hipEventCreateWithFlags(&m->done, hipEventDisableTiming));
  HIPCHK(hipEventRecord(m->done, m->stream));
  HIPCHK(hipStreamWaitEvent(ctx->stream, m->done, 0));
  return 0;
}

#ifdef TAIL_STAMPS
// diagnostic build only: reads and clears the phase cycle counters of k_decode_tail
int od_hip_tail_stamps(unsigned long long out[16]) {
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_stamps), 16*sizeof(unsigned long long)));
  unsigned long long z[16] = {};
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_tail_stamps), z, sizeof(z)));
  return 0;
}
#endif

// F3, second half (mc_kernels.hpp: k_mc_sad_items).  od_hip_mc_set_src: plane pli of the frame
// being coded, w x h samples (the encoder's padded input plane), resident until the next call.
int od_hip_mc_set_src(od_hip_mc *m, int pli, const unsigned char *plane, int stride, int w, int h, int xdec,
                      int ydec) {
  if (!m || !plane) return fail(OD_HIP_EFAULT, "null pointer");
  if (pli < 0 || pli > 2 || w < 1 || h < 1 || stride < w || xdec < 0 || xdec > 1 || ydec < 0 || ydec > 1)
    return fail(OD_HIP_EINVAL, "bad source plane");
  HIPCHK(hipSetDevice(m->device));
  auto &S = m->src[pli];
  const size_t bytes = (size_t)w*h;
  HIPCHK(hipStreamSynchronize(m->stream));          // the staging plane may still be in flight
  if (S.cap < bytes) {
    if (S.d) (void)hipFree(S.d);
    if (S.h) (void)hipHostFree(S.h);
    S.d = S.h = nullptr;
    S.cap = 0;
    HIPCHK(hipMalloc((void **)&S.d, bytes));
    HIPCHK(hipHostMalloc((void **)&S.h, bytes));
    S.cap = bytes;
  }
  S.w = w;
  S.h_rows = h;
  S.xdec = xdec;
  S.ydec = ydec;
  for (int y = 0; y < h; y++) memcpy(S.h + (size_t)y*w, plane + (size_t)y*stride, w);
  HIPCHK(hipMemcpyAsync(S.d, S.h, bytes, hipMemcpyHostToDevice, m->stream));
  return 0;
}

// SAD of the OBMC prediction of every item against the source planes, planes 0 .. nplanes - 1
// summed as od_mv_est_sad does (src/mcenc.c:2271-2300; chroma >> OD_MC_CHROMA_SCALE).
int od_hip_mc_sad_items(od_hip_mc *m, int nplanes, int pic_w, int pic_h, const od_hip_mc_sad_item *items,
                        int nitems, int32_t *sad) {
  if (!m || !items || !sad) return fail(OD_HIP_EFAULT, "null pointer");
  if (nplanes < 1 || nplanes > 3 || pic_w < 1 || pic_h < 1 || nitems < 0) return fail(OD_HIP_EINVAL, "bad geometry");
  for (int pli = 0; pli < nplanes; pli++) {
    if (!m->pl[pli].d_refs) return fail(OD_HIP_EINVAL, "no reference planes set for this plane");
    if (!m->src[pli].d) return fail(OD_HIP_EINVAL, "no source plane set for this plane");
  }
  // operand shapes are checked on the host before anything is launched
  for (int b = 0; b < nitems; b++) {
    const od_hip_mc_sad_item &q = items[b];
    if (q.log_blk_sz < 3 || q.log_blk_sz > 6 || q.x < 0 || q.y < 0 || q.oc < 0 || q.oc > 3 || q.s < 0 || q.s > 3)
      return fail(OD_HIP_EINVAL, "bad SAD item");
    for (int pli = 0; pli < nplanes; pli++) {
      const auto &S = m->src[pli];
      if ((q.x & ((1 << S.xdec) - 1)) || (q.y & ((1 << S.ydec) - 1))
          || (q.x >> S.xdec) + (1 << (q.log_blk_sz - S.xdec)) > S.w
          || (q.y >> S.ydec) + (1 << (q.log_blk_sz - S.ydec)) > S.h_rows)
        return fail(OD_HIP_EINVAL, "SAD item outside the source plane");
    }
    for (int k = 0; k < 4; k++) if (q.ref[k] < 0 || q.ref[k] >= m->nref) return fail(OD_HIP_EINVAL, "bad reference index");
  }
  if (nitems == 0) return 0;
  HIPCHK(hipSetDevice(m->device));
  static_assert(sizeof(McSadItem) == sizeof(od_hip_mc_sad_item), "McSadItem mirrors od_hip_mc_sad_item");
  if (m->items_cap < (size_t)nitems) {
    HIPCHK(hipStreamSynchronize(m->stream));
    if (m->d_items) (void)hipFree(m->d_items);
    if (m->h_items) (void)hipHostFree(m->h_items);
    if (m->d_sad) (void)hipFree(m->d_sad);
    if (m->h_sad) (void)hipHostFree(m->h_sad);
    m->d_items = m->h_items = nullptr;
    m->d_sad = m->h_sad = nullptr;
    m->items_cap = 0;
    const size_t cap = (size_t)nitems + nitems/2;
    HIPCHK(hipMalloc((void **)&m->d_items, cap*sizeof(McSadItem)));
    HIPCHK(hipHostMalloc((void **)&m->h_items, cap*sizeof(McSadItem)));
    HIPCHK(hipMalloc((void **)&m->d_sad, cap*sizeof(int32_t)));
    HIPCHK(hipHostMalloc((void **)&m->h_sad, cap*sizeof(int32_t)));
    m->items_cap = cap;
  }
  HIPCHK(hipStreamSynchronize(m->stream));
  memcpy(m->h_items, items, (size_t)nitems*sizeof(McSadItem));
  HIPCHK(hipMemcpyAsync(m->d_items, m->h_items, (size_t)nitems*sizeof(McSadItem), hipMemcpyHostToDevice, m->stream));
  McSadArgs a;
  for (int pli = 0; pli < 3; pli++) {
    const int q = pli < nplanes ? pli : 0;
    const auto &P = m->pl[q];
    const auto &S = m->src[q];
    a.pl[pli].R.refs = P.d_refs;
    a.pl[pli].R.ref_plane = P.plane;
    a.pl[pli].R.ref_stride = P.ref_stride;
    a.pl[pli].R.ref_h = P.ref_h;
    a.pl[pli].R.org_x = P.org_x;
    a.pl[pli].R.org_y = P.org_y;
    a.pl[pli].src = S.d;
    a.pl[pli].src_stride = S.w;
    a.pl[pli].xdec = S.xdec;
    a.pl[pli].ydec = S.ydec;
    a.pl[pli].clip_w = (pic_w + (1 << S.xdec) - 1) >> S.xdec;
    a.pl[pli].clip_h = (pic_h + (1 << S.ydec) - 1) >> S.ydec;
    a.pl[pli].shift = q > 0 ? 2 : 0;                 // OD_MC_CHROMA_SCALE (src/mcenc.c:53)
  }
  a.nplanes = nplanes;
  a.items = m->d_items;
  a.nitems = nitems;
  a.sad = m->d_sad;
  mc_launch_sad(a, m->h_items, nitems, m->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(m->h_sad, m->d_sad, (size_t)nitems*sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipStreamSynchronize(m->stream));
  memcpy(sad, m->h_sad, (size_t)nitems*sizeof(int32_t));
  return 0;
}

// F3: dense block-matching windows (mc_kernels.hpp: k_mc_bma_windows) - od_mv_est_bma_sad
// (src/mcenc.c:2228-2268) of every half-sample vector within `radius` of each record's centre,
// against the resident references and source planes.  out: [nrec][(2 radius + 1)^2], -1 where the
// vector lies outside the record's limits.
int od_hip_mc_bma_windows(od_hip_mc *m, int nplanes, int pic_w, int pic_h, const od_hip_mc_bma_rec *recs,
                          int nrec, int radius, int32_t *out) {
  if (!m || !recs || !out) return fail(OD_HIP_EFAULT, "null pointer");
  if (nplanes < 1 || nplanes > 3 || pic_w < 1 || pic_h < 1 || nrec < 0 || nrec > (1 << 22) || radius < 0 || radius > 16)
    return fail(OD_HIP_EINVAL, "bad geometry");
  for (int pli = 0; pli < nplanes; pli++) {
    if (!m->pl[pli].d_refs) return fail(OD_HIP_EINVAL, "no reference planes set for this plane");
    if (!m->src[pli].d) return fail(OD_HIP_EINVAL, "no source plane set for this plane");
  }
  // operand shapes are checked on the host before anything is launched: a block may hang over
  // the frame (it is centred on a grid vertex) by at most its own size; what it reads of the
  // references is clamped to the plane by the kernel
  for (int b = 0; b < nrec; b++) {
    const od_hip_mc_bma_rec &q = recs[b];
    const int n = 1 << q.log_blk_sz;
    if (q.log_blk_sz < 3 || q.log_blk_sz > 6 || q.ref < 0 || q.ref >= m->nref || (q.bx & 3) || (q.by & 3)
        || q.bx < -n || q.by < -n || q.bx > m->src[0].w || q.by > m->src[0].h_rows
        || q.xmin > q.xmax || q.ymin > q.ymax || q.xmin < -(1 << 14) || q.xmax > (1 << 14)
        || q.ymin < -(1 << 14) || q.ymax > (1 << 14))
      return fail(OD_HIP_EINVAL, "bad block-matching record");
  }
  if (nrec == 0) return 0;
  HIPCHK(hipSetDevice(m->device));
  static_assert(sizeof(McBmaRec) == sizeof(od_hip_mc_bma_rec), "McBmaRec mirrors od_hip_mc_bma_rec");
  const int W = 2*radius + 1;
  const size_t nwin = (size_t)nrec*W*W;
  if (m->recs_cap < (size_t)nrec || m->win_cap < nwin) {
    HIPCHK(hipStreamSynchronize(m->stream));
    if (m->d_recs) (void)hipFree(m->d_recs);
    if (m->h_recs) (void)hipHostFree(m->h_recs);
    if (m->d_win) (void)hipFree(m->d_win);
    if (m->h_win) (void)hipHostFree(m->h_win);
    m->d_recs = m->h_recs = nullptr;
    m->d_win = m->h_win = nullptr;
    m->recs_cap = m->win_cap = 0;
    const size_t rc = (size_t)nrec + nrec/2, wc = nwin + nwin/2;
    HIPCHK(hipMalloc((void **)&m->d_recs, rc*sizeof(McBmaRec)));
    HIPCHK(hipHostMalloc((void **)&m->h_recs, rc*sizeof(McBmaRec)));
    HIPCHK(hipMalloc((void **)&m->d_win, wc*sizeof(int32_t)));
    HIPCHK(hipHostMalloc((void **)&m->h_win, wc*sizeof(int32_t)));
    m->recs_cap = rc;
    m->win_cap = wc;
  }
  HIPCHK(hipStreamSynchronize(m->stream));
  memcpy(m->h_recs, recs, (size_t)nrec*sizeof(McBmaRec));
  HIPCHK(hipMemcpyAsync(m->d_recs, m->h_recs, (size_t)nrec*sizeof(McBmaRec), hipMemcpyHostToDevice, m->stream));
  McBmaArgs a;
  for (int pli = 0; pli < 3; pli++) {
    const int q = pli < nplanes ? pli : 0;
    const auto &P = m->pl[q];
    const auto &S = m->src[q];
    a.pl[pli].R.refs = P.d_refs;
    a.pl[pli].R.ref_plane = P.plane;
    a.pl[pli].R.ref_stride = P.ref_stride;
    a.pl[pli].R.ref_h = P.ref_h;
    a.pl[pli].R.org_x = P.org_x;
    a.pl[pli].R.org_y = P.org_y;
    a.pl[pli].src = S.d;
    a.pl[pli].src_stride = S.w;
    a.pl[pli].xdec = S.xdec;
    a.pl[pli].ydec = S.ydec;
    a.pl[pli].clip_w = min((pic_w + (1 << S.xdec) - 1) >> S.xdec, S.w);
    a.pl[pli].clip_h = min((pic_h + (1 << S.ydec) - 1) >> S.ydec, S.h_rows);
    a.pl[pli].shift = q > 0 ? 2 : 0;                 // OD_MC_CHROMA_SCALE (src/mcenc.c:53)
  }
  a.nplanes = nplanes;
  a.radius = radius;
  // the records are the grid's y extent (at most 65 535): a 4K frame's finest level has more
  constexpr int CHUNK = 32768;
  for (int r0 = 0; r0 < nrec; r0 += CHUNK) {
    a.recs = m->d_recs + r0;
    a.nrec = min(CHUNK, nrec - r0);
    a.sad = m->d_win + (size_t)r0*W*W;
    int worst = 4;
    for (int b = 0; b < a.nrec; b++) worst = max(worst, mc_size_class(recs[r0 + b].log_blk_sz));
    // OD_HIP_BMA_V1=1: one wave per (vertex, offset), every offset filtered by itself (A/B);
    // default: one wave per vertex, phase planes shared by the window's offsets (radius <= 4)
    const int v1 = env_int("OD_HIP_BMA_V1", 0);           // read per call: the parity test runs both kernels in one process
    if (v1 || radius > 4) {
      if (worst == 4) hipLaunchKernelGGL(k_mc_bma_windows<4>, dim3(W*W, a.nrec), dim3(MC_SAD_THREADS), 0, m->stream, a);
      else if (worst == 5) hipLaunchKernelGGL(k_mc_bma_windows<5>, dim3(W*W, a.nrec), dim3(MC_SAD_THREADS), 0, m->stream, a);
      else hipLaunchKernelGGL(k_mc_bma_windows<6>, dim3(W*W, a.nrec), dim3(MC_SAD_THREADS), 0, m->stream, a);
    }
    else {
      if (worst == 4) hipLaunchKernelGGL((k_mc_bma_windows_v2<4, 64>), dim3(a.nrec), dim3(64), 0, m->stream, a);
      else if (worst == 5) hipLaunchKernelGGL((k_mc_bma_windows_v2<5, 256>), dim3(a.nrec), dim3(256), 0, m->stream, a);
      else hipLaunchKernelGGL((k_mc_bma_windows_v2<6, 256>), dim3(a.nrec), dim3(256), 0, m->stream, a);
    }
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipMemcpyAsync(m->h_win, m->d_win, nwin*sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipStreamSynchronize(m->stream));
  memcpy(out, m->h_win, nwin*sizeof(int32_t));
  return 0;
}

// A11: od_raster_to_coding_order / od_coding_order_to_raster (src/partition.c:144-194) for
// nblocks dense n x n blocks.  One thread per (block, coding index).
__global__ void k_coding_order_blocks(int nn, int ncoded, long total, int to_raster,
                                      const uint16_t *__restrict__ tab,
                                      const int32_t *__restrict__ in, int32_t *__restrict__ out) {
  const long t = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= total) return;
  const long b = t/ncoded;
  const int i = (int)(t - b*ncoded);
  if (to_raster) out[b*nn + tab[i]] = in[b*nn + i];
  else out[b*nn + i] = in[b*nn + tab[i]];
}

int od_hip_coding_order_blocks(int bs, int to_raster, od_coeff *inout_dst, const od_coeff *src,
                               int nblocks) {
  if (!inout_dst || !src) return fail(OD_HIP_EFAULT, "null pointer");
  if (bs < 0 || bs >= OD_HIP_NBSIZES || nblocks < 0) return fail(OD_HIP_EINVAL, "bad bs/nblocks");
  if (int rc = ensure_device()) return rc;
  if (nblocks == 0) return 0;
  const uint16_t *tabs[4] = {CODING_TO_RASTER_4, CODING_TO_RASTER_8, CODING_TO_RASTER_16,
                             CODING_TO_RASTER_32};
  const int tabn[4] = {CODING_NCODED_4, CODING_NCODED_8, CODING_NCODED_16, CODING_NCODED_32};
  const int n = 4 << bs, nn = n*n, ncoded = tabn[bs];
  const size_t bytes = (size_t)nblocks*nn*4;
  SCRATCH_LOCK;
  if (int rc = g_in.reserve(bytes)) return rc;
  if (int rc = g_out.reserve(bytes)) return rc;
  if (int rc = g_aux0.reserve((size_t)ncoded*2)) return rc;
  HIPCHK(hipMemcpy(g_in.p, src, bytes, hipMemcpyHostToDevice));
  // positions the permutation does not touch (a 32x32 block codes 512 of its 1024
  // coefficients) keep the caller's values, as in the reference (src/encode.c:1219-1220)
  HIPCHK(hipMemcpy(g_out.p, inout_dst, bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux0.p, tabs[bs], (size_t)ncoded*2, hipMemcpyHostToDevice));
  const long total = (long)nblocks*ncoded;
  hipLaunchKernelGGL(k_coding_order_blocks, dim3((unsigned)((total + 255)/256)), dim3(256), 0, 0, nn,
                     ncoded, total, to_raster, (const uint16_t *)g_aux0.p, (const int32_t *)g_in.p,
                     (int32_t *)g_out.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(inout_dst, g_out.p, bytes, hipMemcpyDeviceToHost));
  return 0;
}

int od_hip_filter4_vectors(int inverse, od_coeff *out, const od_coeff *in, int nvec) {
  if (!out || !in) return fail(OD_HIP_EFAULT, "null pointer");
  if (nvec < 0) return fail(OD_HIP_EINVAL, "bad nvec");
  if (int rc = ensure_device()) return rc;
  if (nvec == 0) return 0;
  size_t bytes = (size_t)nvec*4*sizeof(int32_t);
  SCRATCH_LOCK;
  if (int rc = g_in.reserve(bytes)) return rc;
  if (int rc = g_out.reserve(bytes)) return rc;
  HIPCHK(hipMemcpy(g_in.p, in, bytes, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_filter4_vectors, dim3((nvec + 255)/256), dim3(256), 0, 0,
                     (int32_t *)g_out.p, (const int32_t *)g_in.p, nvec, inverse);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, g_out.p, bytes, hipMemcpyDeviceToHost));
  return 0;
}

int od_hip_filter_vectors(int n, int inverse, od_coeff *out, const od_coeff *in, int nvec) {
  if (!out || !in) return fail(OD_HIP_EFAULT, "null pointer");
  if (nvec < 0 || (n != 4 && n != 8 && n != 16 && n != 32)) return fail(OD_HIP_EINVAL, "bad n/nvec");
  if (int rc = ensure_device()) return rc;
  if (nvec == 0) return 0;
  size_t bytes = (size_t)nvec*n*sizeof(int32_t);
  SCRATCH_LOCK;
  if (int rc = g_in.reserve(bytes)) return rc;
  if (int rc = g_out.reserve(bytes)) return rc;
  HIPCHK(hipMemcpy(g_in.p, in, bytes, hipMemcpyHostToDevice));
  dim3 grid((nvec + 63)/64), blk(64);
  int32_t *o = (int32_t *)g_out.p;
  const int32_t *i = (const int32_t *)g_in.p;
  switch (n) {
    case 4: hipLaunchKernelGGL(k_filter_vectors<4>, grid, blk, 0, 0, o, i, nvec, inverse); break;
    case 8: hipLaunchKernelGGL(k_filter_vectors<8>, grid, blk, 0, 0, o, i, nvec, inverse); break;
    case 16: hipLaunchKernelGGL(k_filter_vectors<16>, grid, blk, 0, 0, o, i, nvec, inverse); break;
    default: hipLaunchKernelGGL(k_filter_vectors<32>, grid, blk, 0, 0, o, i, nvec, inverse); break;
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, g_out.p, bytes, hipMemcpyDeviceToHost));
  return 0;
}

int od_hip_resample_luma_420(od_coeff *pred, const od_coeff *luma, size_t luma_len,
                             int lstride, const int32_t *luma_off, int nblk, int bs,
                             int chroma_bs) {
  return od_hip_resample_luma(pred, luma, luma_len, lstride, luma_off, nblk, bs, chroma_bs, 1, 1);
}

int od_hip_resample_luma(od_coeff *pred, const od_coeff *luma, size_t luma_len,
                         int lstride, const int32_t *luma_off, int nblk, int bs,
                         int chroma_bs, int xdec, int ydec) {
  if (!pred || !luma || !luma_off) return fail(OD_HIP_EFAULT, "null pointer");
  if (bs < 0 || bs >= OD_HIP_NBSIZES || nblk < 0 || (chroma_bs == 0 && bs != 0) ||
      xdec < 0 || xdec > 1 || ydec < 0 || ydec > 1)
    return fail(OD_HIP_EINVAL, "bad block size / decimation");
  // without decimation the chroma block is a plain copy whatever chroma_bs says
  // (src/intra.c:76: the TF merge only happens for a decimated plane)
  if (!xdec && !ydec && chroma_bs == 0) chroma_bs = 1;
  const int mode = xdec && ydec ? 0 : xdec ? 1 : 2;
  if (int rc = ensure_device()) return rc;
  if (nblk == 0) return 0;
  size_t n = 4u << bs, obytes = (size_t)nblk*n*n*sizeof(int32_t);
  SCRATCH_LOCK;
  if (int rc = g_in.reserve(luma_len*sizeof(int32_t))) return rc;
  if (int rc = g_out.reserve(obytes)) return rc;
  if (int rc = g_aux0.reserve((size_t)nblk*sizeof(int32_t))) return rc;
  HIPCHK(hipMemcpy(g_in.p, luma, luma_len*sizeof(int32_t), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux0.p, luma_off, (size_t)nblk*sizeof(int32_t), hipMemcpyHostToDevice));
  long total = (long)nblk*n*n;
  hipLaunchKernelGGL(k_resample_luma_420, dim3((total + 255)/256), dim3(256), 0, 0,
                     (int32_t *)g_out.p, (const int32_t *)g_in.p, lstride,
                     (const int32_t *)g_aux0.p, nblk, bs, chroma_bs, mode);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(pred, g_out.p, obytes, hipMemcpyDeviceToHost));
  return 0;
}

// ---------------------------------------------------------------------------
od_hip_ctx *od_hip_ctx_create(int device, const od_hip_geometry *geo) {
  if (!geo) { fail(OD_HIP_EFAULT, "null geometry"); return nullptr; }
  if (ensure_device()) return nullptr;
  if (geo->frame_width <= 0 || geo->frame_height <= 0 || geo->frame_width%32 ||
      geo->frame_height%32 || geo->pic_width > geo->frame_width ||
      geo->pic_height > geo->frame_height || geo->pic_width <= 0 || geo->pic_height < 0 ||
      geo->nplanes < 1 || geo->nplanes > 3 || geo->nslots < 1) {
    fail(OD_HIP_EINVAL, "invalid geometry");
    return nullptr;
  }
  for (int p = 0; p < geo->nplanes; p++) {
    if (geo->xdec[p] < 0 || geo->xdec[p] > 1 || (p == 0 && geo->xdec[p] != 0)) {
      fail(OD_HIP_EINVAL, "unsupported decimation");
      return nullptr;
    }
  }
  if (hipSetDevice(device) != hipSuccess) { fail(OD_HIP_ENODEV, "hipSetDevice failed"); return nullptr; }
  od_hip_ctx *ctx = new od_hip_ctx();
  ctx->geo = *geo;
  ctx->device = device;
  ctx->nhsb = geo->frame_width/32;
  ctx->nvsb = geo->frame_height/32;
  ctx->strip0 = 0;
  ctx->strip1 = ctx->nvsb;
  memset(ctx->pix, 0, sizeof(ctx->pix));
  memset(ctx->lev, 0, sizeof(ctx->lev));
  memset(ctx->d, 0, sizeof(ctx->d));
  memset(ctx->c, 0, sizeof(ctx->c));
  memset(ctx->rec, 0, sizeof(ctx->rec));
  memset(ctx->p32, 0, sizeof(ctx->p32));
  memset(ctx->bskip, 0, sizeof(ctx->bskip));
  ctx->dflags = nullptr;
  memset(ctx->pvq, 0, sizeof(ctx->pvq));
  memset(ctx->pvq_alloc, 0, sizeof(ctx->pvq_alloc));
  memset(ctx->tab, 0, sizeof(ctx->tab));
  ctx->bsize = nullptr;
  ctx->qm_dev = nullptr;
  ctx->rsq = nullptr;
  bool ok = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess;
  size_t ns = geo->nslots;
  for (int p = 0; ok && p < geo->nplanes; p++) {
    int dec = geo->xdec[p];
    ctx->pw[p] = geo->frame_width >> dec;
    ctx->ph[p] = geo->frame_height >> dec;
    ctx->nlev[p] = 4 - dec;
    ctx->psz[p] = (size_t)ctx->pw[p]*ctx->ph[p];
    ok = ok && hipMalloc((void **)&ctx->pix[p], ns*ctx->psz[p]) == hipSuccess;
    ok = ok && hipMalloc((void **)&ctx->lev[p], ns*ctx->nlev[p]*ctx->psz[p]*4) == hipSuccess;
    ok = ok && hipMalloc((void **)&ctx->d[p], ns*ctx->psz[p]*4) == hipSuccess;
    ok = ok && hipMalloc((void **)&ctx->c[p], ns*ctx->psz[p]*4) == hipSuccess;
    ok = ok && hipMalloc((void **)&ctx->rec[p], ns*ctx->psz[p]) == hipSuccess;
  }
  ctx->bsize_sz = (size_t)ctx->nhsb*4*ctx->nvsb*4;
  ok = ok && hipMalloc((void **)&ctx->bsize, ns*ctx->bsize_sz) == hipSuccess;
  ok = ok && hipMemsetAsync(ctx->bsize, 3, ns*ctx->bsize_sz, ctx->stream) == hipSuccess;
  ok = ok && hipMalloc((void **)&ctx->qm_dev, 4*1024*sizeof(int16_t)) == hipSuccess;
  ok = ok && hipMalloc((void **)&ctx->rsq, PVQ_RSQ_TAB*sizeof(double)) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(k_pvq_fill_rsqrt, dim3(PVQ_RSQ_TAB/256), dim3(256), 0, ctx->stream, ctx->rsq);
    ok = hipGetLastError() == hipSuccess;
  }
  if (const char *e = getenv("OD_HIP_PVQ_STREAMS")) {
    int v = atoi(e);
    ctx->naux = v < 0 ? 0 : v > od_hip_ctx::NAUX ? od_hip_ctx::NAUX : v;
  }
  for (int i = 0; ok && i < ctx->naux; i++) {
    ok = ok && hipStreamCreateWithFlags(&ctx->aux[i], hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&ctx->aux_done[i], hipEventDisableTiming) == hipSuccess;
  }
  ok = ok && hipEventCreateWithFlags(&ctx->aux_dep, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipMalloc((void **)&ctx->qm_slots, 3*4*1024*sizeof(int16_t)) == hipSuccess;
  const uint16_t *tabs[4] = {CODING_TO_RASTER_4, CODING_TO_RASTER_8, CODING_TO_RASTER_16,
                             CODING_TO_RASTER_32};
  const int tabn[4] = {CODING_NCODED_4, CODING_NCODED_8, CODING_NCODED_16, CODING_NCODED_32};
  for (int b = 0; ok && b < 4; b++) {
    ok = ok && hipMalloc((void **)&ctx->tab[b], tabn[b]*sizeof(uint16_t)) == hipSuccess;
    ok = ok && hipMemcpyAsync(ctx->tab[b], tabs[b], tabn[b]*sizeof(uint16_t),
                              hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
  }
  // everything above was issued on ctx->stream; nothing may still be in flight when the
  // first caller-visible operation starts
  ok = ok && hipStreamSynchronize(ctx->stream) == hipSuccess;
  if (!ok) {
    fail(OD_HIP_ENODEV, "device allocation failed");
    od_hip_ctx_destroy(ctx);
    return nullptr;
  }
  return ctx;
}

void od_hip_ctx_destroy(od_hip_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  for (int i = 0; i < od_hip_ctx::NAUX; i++) if (ctx->aux[i]) (void)hipStreamSynchronize(ctx->aux[i]);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (int p = 0; p < OD_HIP_NPLANES_MAX; p++) {
    if (ctx->pix[p]) (void)hipFree(ctx->pix[p]);
    if (ctx->lev[p]) (void)hipFree(ctx->lev[p]);
    if (ctx->d[p]) (void)hipFree(ctx->d[p]);
    if (ctx->c[p]) (void)hipFree(ctx->c[p]);
    if (ctx->rec[p]) (void)hipFree(ctx->rec[p]);
    if (ctx->p32[p]) (void)hipFree(ctx->p32[p]);
    if (ctx->rs[p]) (void)hipFree(ctx->rs[p]);
    if (ctx->cs[p]) (void)hipFree(ctx->cs[p]);
    if (ctx->bskip[p]) (void)hipFree(ctx->bskip[p]);
    {
      PvqArena &A = ctx->arena[p];
      void *ptrs[] = {A.out, A.g, A.in, A.dist};
      for (void *q : ptrs) if (q) (void)hipFree(q);
    }
  }
  for (int b = 0; b < 4; b++) if (ctx->tab[b]) (void)hipFree(ctx->tab[b]);
  if (ctx->pvq_stats) (void)hipFree(ctx->pvq_stats);
  if (ctx->order_scratch) (void)hipFree(ctx->order_scratch);
  strip_cache_free(ctx);
  if (ctx->bsize) (void)hipFree(ctx->bsize);
  if (ctx->dflags) (void)hipFree(ctx->dflags);
  for (int i = 0; i < od_hip_ctx::NAUX; i++) {
    if (ctx->aux[i]) { (void)hipStreamSynchronize(ctx->aux[i]); (void)hipStreamDestroy(ctx->aux[i]); }
    if (ctx->aux_done[i]) (void)hipEventDestroy(ctx->aux_done[i]);
  }
  if (ctx->aux_dep) (void)hipEventDestroy(ctx->aux_dep);
  if (ctx->qm_slots) (void)hipFree(ctx->qm_slots);
  if (ctx->qm_dev) (void)hipFree(ctx->qm_dev);
  if (ctx->rsq) (void)hipFree(ctx->rsq);
  for (auto &kv : ctx->spans) for (auto &s : kv.second) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
  for (auto e : ctx->pool) (void)hipEventDestroy(e);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int od_hip_upload_planes(od_hip_ctx *ctx, int slot, const unsigned char *const planes[],
                         const int ystride[]) {
  if (int rc = check_slots(ctx, slot, 1)) return rc;
  if (!planes || !ystride) return fail(OD_HIP_EFAULT, "null pointer");
  for (int p = 0; p < ctx->geo.nplanes; p++) {
    if (!planes[p] || ystride[p] < ctx->pw[p]) return fail(OD_HIP_EINVAL, "bad plane");
    HIPCHK(hipMemcpy2DAsync(ctx->pix[p] + (size_t)slot*ctx->psz[p], ctx->pw[p], planes[p],
                            ystride[p], ctx->pw[p], ctx->ph[p], hipMemcpyHostToDevice,
                            ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int od_hip_set_bsize(od_hip_ctx *ctx, int slot, const unsigned char *bsize, int bstride) {
  if (int rc = check_slots(ctx, slot, 1)) return rc;
  if (!bsize || bstride < ctx->nhsb*4) return fail(OD_HIP_EINVAL, "bad bsize map");
  // validate: values 0..3 and quadtree consistency, so the kernels' per-block
  // top-left-cell test is exact
  int bw = ctx->nhsb*4, bh = ctx->nvsb*4;
  for (int y = 0; y < bh; y++) {
    for (int x = 0; x < bw; x++) {
      int v = bsize[(size_t)y*bstride + x];
      if (v > 3) return fail(OD_HIP_EINVAL, "bsize value out of range");
      if (v >= 2) {
        int m = (v == 3) ? 4 : 2;
        int v0 = bsize[(size_t)(y/m*m)*bstride + (x/m*m)];
        if (v0 != v) return fail(OD_HIP_EINVAL, "bsize map is not quadtree-consistent");
      }
    }
  }
  for (int y = 0; y < bh; y += 2) for (int x = 0; x < bw; x += 2) {
    // a 16x16 area is either all 2 (or part of a 3) or all < 2
    int hi = 0, lo = 0;
    for (int j = 0; j < 2; j++) for (int i = 0; i < 2; i++) {
      if (bsize[(size_t)(y + j)*bstride + x + i] >= 2) hi++; else lo++;
    }
    if (hi && lo) return fail(OD_HIP_EINVAL, "bsize map is not quadtree-consistent");
  }
  for (int y = 0; y < bh; y += 4) for (int x = 0; x < bw; x += 4) {
    int hi = 0, lo = 0;
    for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) {
      if (bsize[(size_t)(y + j)*bstride + x + i] == 3) hi++; else lo++;
    }
    if (hi && lo) return fail(OD_HIP_EINVAL, "bsize map is not quadtree-consistent");
  }
  HIPCHK(hipMemcpy2DAsync(ctx->bsize + (size_t)slot*ctx->bsize_sz, bw, bsize, bstride, bw, bh,
                          hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

static FwdArgs fwd_args(od_hip_ctx *ctx, int p, int slot0, bool known, int keyframe) {
  FwdArgs a;
  a.pix = ctx->pix[p] + (size_t)slot0*ctx->psz[p];
  a.pix_fstride = ctx->psz[p];
  a.pstride = ctx->pw[p];
  if (known) {
    a.out = ctx->d[p] + (size_t)slot0*ctx->psz[p];
    a.out_lstride = 0;
    a.out_fstride = ctx->psz[p];
  }
  else {
    a.out = ctx->lev[p] + (size_t)slot0*ctx->nlev[p]*ctx->psz[p];
    a.out_lstride = ctx->psz[p];
    a.out_fstride = (size_t)ctx->nlev[p]*ctx->psz[p];
  }
  a.bsize = ctx->bsize + (size_t)slot0*ctx->bsize_sz;
  a.bsize_fstride = ctx->bsize_sz;
  a.bstride = ctx->nhsb*4;
  a.w = ctx->pw[p];
  a.h = ctx->ph[p];
  a.nhsb = ctx->nhsb;
  a.nvsb = ctx->nvsb;
  a.pic_w = ctx->geo.pic_width;
  a.pic_h = ctx->geo.pic_height;
  a.dec = ctx->geo.xdec[p];
  a.keyframe = keyframe;
  a.sby0 = 0;
  return a;
}

int od_hip_set_strip(od_hip_ctx *ctx, int sb_row0, int sb_row1) {
  if (!ctx) return fail(OD_HIP_EFAULT, "null context");
  if (sb_row0 < 0 || sb_row1 < sb_row0 || sb_row1 > ctx->nvsb) return fail(OD_HIP_EINVAL, "bad strip");
  ctx->strip0 = sb_row0;
  ctx->strip1 = sb_row1;
  return 0;
}

int od_hip_forward_pyramid(od_hip_ctx *ctx, int slot0, int nslots) {
  if (int rc = check_slots(ctx, slot0, nslots)) return rc;
  const int rows = ctx->strip1 - ctx->strip0;
  if (rows <= 0) return 0;                             // an empty strip (more ranks than SB rows)
  for (int p = 0; p < ctx->geo.nplanes; p++) {
    FwdArgs a = fwd_args(ctx, p, slot0, false, 0);
    a.sby0 = ctx->strip0;
    if (a.dec == 0) {
      Timed tm(ctx, "k_forward_pyramid_luma");
      hipLaunchKernelGGL((k_forward_rt<32, 4, false>), dim3((ctx->nhsb + 1)/2, rows, nslots),
                           dim3(128), 0, ctx->stream, a);
    }
    else {
      Timed tm(ctx, "k_forward_pyramid_chroma");
      hipLaunchKernelGGL((k_forward_rt<16, 3, false>), dim3((ctx->nhsb + 3)/4, rows, nslots),
                           dim3(128), 0, ctx->stream, a);
    }
    HIPCHK(hipGetLastError());
  }
  return 0;
}

int od_hip_forward_known(od_hip_ctx *ctx, int slot0, int nslots, int keyframe) {
  if (int rc = check_slots(ctx, slot0, nslots)) return rc;
  for (int p = 0; p < ctx->geo.nplanes; p++) {
    FwdArgs a = fwd_args(ctx, p, slot0, true, keyframe);
    if (a.dec == 0) {
      Timed tm(ctx, "k_forward_known_luma");
      hipLaunchKernelGGL((k_forward_rt<32, 4, true>), dim3((ctx->nhsb + 1)/2, ctx->nvsb, nslots),
                           dim3(128), 0, ctx->stream, a);
    }
    else {
      Timed tm(ctx, "k_forward_known_chroma");
      hipLaunchKernelGGL((k_forward_rt<16, 3, true>), dim3((ctx->nhsb + 3)/4, ctx->nvsb, nslots),
                           dim3(128), 0, ctx->stream, a);
    }
    HIPCHK(hipGetLastError());
  }
  return 0;
}

int od_hip_inverse(od_hip_ctx *ctx, int slot0, int nslots) {
  if (int rc = check_slots(ctx, slot0, nslots)) return rc;
  // OD_HIP_INVERSE_IMPL=1: the two-kernel form through the int32 work plane (what the
  // decoder tail uses, where deringing needs that plane anyway); default: fused (no work plane)
  static const int impl = env_int("OD_HIP_INVERSE_IMPL", 2);
  for (int p = 0; p < ctx->geo.nplanes; p++) {
    InvArgs a;
    a.d = ctx->d[p] + (size_t)slot0*ctx->psz[p];
    a.c = ctx->c[p] + (size_t)slot0*ctx->psz[p];
    a.fstride = ctx->psz[p];
    a.bsize = ctx->bsize + (size_t)slot0*ctx->bsize_sz;
    a.bsize_fstride = ctx->bsize_sz;
    a.bstride = ctx->nhsb*4;
    a.w = ctx->pw[p]; a.h = ctx->ph[p]; a.nhsb = ctx->nhsb; a.nvsb = ctx->nvsb;
    a.pic_w = ctx->geo.pic_width; a.pic_h = ctx->geo.pic_height;
    a.dec = ctx->geo.xdec[p];
    if (impl != 1) {
      const int sb = 32 >> a.dec;
      const int ntx = (a.w + 63)/64;
      const size_t rsz = (size_t)ctx->nvsb*4*a.w, csz = (size_t)ntx*a.h*4;
      if (!ctx->rs[p]) {
        HIPCHK(hipMalloc((void **)&ctx->rs[p], (size_t)ctx->geo.nslots*rsz*sizeof(int16_t)));
        HIPCHK(hipMalloc((void **)&ctx->cs[p], (size_t)ctx->geo.nslots*csz*sizeof(int16_t)));
      }
      a.rec = ctx->rec[p] + (size_t)slot0*ctx->psz[p];
      a.rs = ctx->rs[p] + (size_t)slot0*rsz;
      a.cs = ctx->cs[p] + (size_t)slot0*csz;
      a.rs_fstride = rsz;
      a.cs_fstride = csz;
      a.ntx = ntx;
      (void)sb;
      const dim3 grid(ntx, ctx->nvsb, nslots);
      const dim3 gridf((ntx + RT_SEG - 1)/RT_SEG, ctx->nvsb, nslots);   // RT_SEG tiles per workgroup
      if (a.dec == 0) {
        { Timed tm(ctx, "k_inverse_sb_luma");
          hipLaunchKernelGGL((k_inverse_rt_fused<32, 4>), gridf, dim3(64), 0, ctx->stream, a); }
        HIPCHK(hipGetLastError());
        { Timed tm(ctx, "k_inverse_strips_luma");
          hipLaunchKernelGGL((k_inverse_strips<32>), grid, dim3(64), 0, ctx->stream, a); }
      }
      else {
        { Timed tm(ctx, "k_inverse_sb_chroma");
          hipLaunchKernelGGL((k_inverse_rt_fused<16, 3>), gridf, dim3(64), 0, ctx->stream, a); }
        HIPCHK(hipGetLastError());
        { Timed tm(ctx, "k_inverse_strips_chroma");
          hipLaunchKernelGGL((k_inverse_strips<16>), grid, dim3(64), 0, ctx->stream, a); }
      }
      HIPCHK(hipGetLastError());
      continue;
    }
    PostArgs q;
    q.c = a.c; q.c_fstride = ctx->psz[p];
    q.rec = ctx->rec[p] + (size_t)slot0*ctx->psz[p]; q.rec_fstride = ctx->psz[p];
    q.out32 = nullptr;
    q.w = a.w; q.h = a.h; q.nhsb = a.nhsb; q.nvsb = a.nvsb;
    dim3 grid2(ctx->nhsb + 1, ctx->nvsb + 1, nslots);
    if (a.dec == 0) {
      { Timed tm(ctx, "k_inverse_sb_luma");
        hipLaunchKernelGGL((k_inverse_rt<32, 4>), dim3((ctx->nhsb + 1)/2, ctx->nvsb, nslots),
                             dim3(64), 0, ctx->stream, a); }
      HIPCHK(hipGetLastError());
      { Timed tm(ctx, "k_postfilter_clamp_luma");
        hipLaunchKernelGGL((k_postfilter_clamp<32>), grid2, dim3(256), 0, ctx->stream, q); }
    }
    else {
      { Timed tm(ctx, "k_inverse_sb_chroma");
        hipLaunchKernelGGL((k_inverse_rt<16, 3>), dim3((ctx->nhsb + 3)/4, ctx->nvsb, nslots),
                             dim3(64), 0, ctx->stream, a); }
      HIPCHK(hipGetLastError());
      { Timed tm(ctx, "k_postfilter_clamp_chroma");
        hipLaunchKernelGGL((k_postfilter_clamp<16>), grid2, dim3(64), 0, ctx->stream, q); }
    }
    HIPCHK(hipGetLastError());
  }
  return 0;
}

// Lossless frames: whole-superblock Haar, no lapping, coefficient shift 0.
static int haar_planes(od_hip_ctx *ctx, int slot0, int nslots, bool inverse) {
  if (int rc = check_slots(ctx, slot0, nslots)) return rc;
  for (int p = 0; p < ctx->geo.nplanes; p++) {
    HaarArgs a;
    a.pix = (inverse ? ctx->rec[p] : ctx->pix[p]) + (size_t)slot0*ctx->psz[p];
    a.d = ctx->d[p] + (size_t)slot0*ctx->psz[p];
    a.fstride = ctx->psz[p];
    a.w = ctx->pw[p];
    dim3 grid(ctx->nhsb, ctx->nvsb, nslots);
    const bool luma = ctx->geo.xdec[p] == 0;
    Timed tm(ctx, inverse ? (luma ? "k_haar_inverse_luma" : "k_haar_inverse_chroma")
                          : (luma ? "k_haar_forward_luma" : "k_haar_forward_chroma"));
    if (luma) {
      if (inverse) hipLaunchKernelGGL(k_haar_inverse_plane<32>, grid, dim3(64), 0, ctx->stream, a);
      else hipLaunchKernelGGL(k_haar_forward_plane<32>, grid, dim3(64), 0, ctx->stream, a);
    }
    else {
      if (inverse) hipLaunchKernelGGL(k_haar_inverse_plane<16>, grid, dim3(64), 0, ctx->stream, a);
      else hipLaunchKernelGGL(k_haar_forward_plane<16>, grid, dim3(64), 0, ctx->stream, a);
    }
    HIPCHK(hipGetLastError());
  }
  return 0;
}

int od_hip_forward_haar(od_hip_ctx *ctx, int slot0, int nslots) {
  return haar_planes(ctx, slot0, nslots, false);
}

int od_hip_inverse_haar(od_hip_ctx *ctx, int slot0, int nslots) {
  return haar_planes(ctx, slot0, nslots, true);
}

static int check_plane(od_hip_ctx *ctx, int slot, int pli) {
  if (int rc = check_slots(ctx, slot, 1)) return rc;
  if (pli < 0 || pli >= ctx->geo.nplanes) return fail(OD_HIP_EINVAL, "plane out of range");
  return 0;
}

int od_hip_download_level(od_hip_ctx *ctx, int slot, int pli, int level, od_coeff *dst) {
  if (int rc = check_plane(ctx, slot, pli)) return rc;
  if (!dst) return fail(OD_HIP_EFAULT, "null pointer");
  if (level < 0 || level >= ctx->nlev[pli]) return fail(OD_HIP_EINVAL, "level out of range");
  // stream-ordered: copies on the context's own stream, then wait (the null stream a
  // plain hipMemcpy uses is not ordered against ctx->stream, a non-blocking stream)
  HIPCHK(hipMemcpyAsync(dst, ctx->lev[pli] + ((size_t)slot*ctx->nlev[pli] + level)*ctx->psz[pli],
                        ctx->psz[pli]*4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int od_hip_download_coeffs(od_hip_ctx *ctx, int slot, int pli, od_coeff *dst) {
  if (int rc = check_plane(ctx, slot, pli)) return rc;
  if (!dst) return fail(OD_HIP_EFAULT, "null pointer");
  HIPCHK(hipMemcpyAsync(dst, ctx->d[pli] + (size_t)slot*ctx->psz[pli], ctx->psz[pli]*4,
                        hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int od_hip_upload_coeffs(od_hip_ctx *ctx, int slot, int pli, const od_coeff *src) {
  if (int rc = check_plane(ctx, slot, pli)) return rc;
  if (!src) return fail(OD_HIP_EFAULT, "null pointer");
  HIPCHK(hipMemcpyAsync(ctx->d[pli] + (size_t)slot*ctx->psz[pli], src, ctx->psz[pli]*4,
                        hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));      // src may be reused by the caller
  return 0;
}

int od_hip_download_recon(od_hip_ctx *ctx, int slot, int pli, unsigned char *dst) {
  if (int rc = check_plane(ctx, slot, pli)) return rc;
  if (!dst) return fail(OD_HIP_EFAULT, "null pointer");
  HIPCHK(hipMemcpyAsync(dst, ctx->rec[pli] + (size_t)slot*ctx->psz[pli], ctx->psz[pli],
                        hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

// ---------------------------------------------------------------------------
int od_hip_band_offsets(int bs, int off[11]) {
  static const int all[] = {1, 16, 24, 32, 64, 96, 128, 256, 384, 512};
  if (bs < 0 || bs >= OD_HIP_NBSIZES || !off) return fail(OD_HIP_EINVAL, "bad bs");
  int nb = bs == 0 ? 1 : bs == 1 ? 4 : bs == 2 ? 7 : 9;
  for (int i = 0; i <= nb; i++) off[i] = all[i];
  return nb;
}

int od_hip_pvq_nblocks(od_hip_ctx *ctx, int pli, int level) {
  if (!ctx) return fail(OD_HIP_EFAULT, "null context");
  if (pli < 0 || pli >= ctx->geo.nplanes || level < 0 || level >= ctx->nlev[pli])
    return fail(OD_HIP_EINVAL, "plane/level out of range");
  int n = (32 >> ctx->geo.xdec[pli]) >> level;
  return (ctx->pw[pli]/n)*(ctx->ph[pli]/n);
}

namespace {
struct PvqCall {
  PvqLevelArgs a;
  int nblk, n, bs;
  size_t nrec, ny;
  PvqSoA *o;
};

// The plane's arenas (all levels), allocated on first use.
int pvq_arena(od_hip_ctx *ctx, int pli) {
  PvqArena &A = ctx->arena[pli];
  if (A.out) return 0;
  const size_t ns = ctx->geo.nslots;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t out = 0, gb = 0, in = 0, db = 0;
  for (int l = 0; l < ctx->nlev[pli]; l++) {
    PvqLevelLayout &L = A.lev[l];
    L.n = (32 >> ctx->geo.xdec[pli]) >> l;
    L.bs = L.n == 4 ? 0 : L.n == 8 ? 1 : L.n == 16 ? 2 : 3;
    L.nb = od_hip_band_offsets(L.bs, L.off);
    L.nblk = (ctx->pw[pli]/L.n)*(ctx->ph[pli]/L.n);
    L.ncoded = L.n*L.n < 512 ? L.n*L.n : 512;
    L.nrec = (size_t)L.nb*L.nblk;
    L.ny = (size_t)2*L.nblk*L.ncoded;
    L.o_cd = out; out = al(out + 2*L.nrec*8);
    L.o_qg = out; out = al(out + 2*L.nrec*4);
    L.o_k = out; out = al(out + 2*L.nrec*4);
    L.o_nc = out; out = al(out + L.nrec*4);
    L.o_y = out; out = al(out + L.ny*2);
    L.o_g = gb; gb = al(gb + L.nrec*8);
    L.o_cg = in; in = al(in + L.nrec*8);
    L.o_perm = in; in = al(in + 2*L.nrec*4);
    A.o_dist[l] = db; db = al(db + 2*L.nrec*8);
  }
  A.out_slot = out; A.g_slot = gb; A.in_slot = in; A.dist_slot = db;
  HIPCHK(hipMalloc((void **)&A.out, ns*out));
  HIPCHK(hipMalloc((void **)&A.g, ns*gb));
  HIPCHK(hipMalloc((void **)&A.in, ns*in));
  HIPCHK(hipMalloc((void **)&A.dist, ns*db));
  HIPCHK(hipMemsetAsync(A.in, 0xff, ns*in, ctx->stream));     // no work list yet: every slot idle
  HIPCHK(hipMemsetAsync(A.out, 0, ns*out, ctx->stream));
  for (int l = 0; l < ctx->nlev[pli]; l++) {
    const PvqLevelLayout &L = A.lev[l];
    PvqSoA &o = ctx->pvq[pli][l];
    o.cos_dist = (double *)(A.out + L.o_cd);
    o.qg = (int32_t *)(A.out + L.o_qg);
    o.k = (int32_t *)(A.out + L.o_k);
    o.ncand = (int32_t *)(A.out + L.o_nc);
    o.y = (int16_t *)(A.out + L.o_y);
    o.g = (double *)(A.g + L.o_g);
    o.cg = (double *)(A.in + L.o_cg);
    o.perm = (const int32_t *)(A.in + L.o_perm);
    o.dist = (double *)((char *)A.dist + A.o_dist[l]);
    o.fs_cd = out/8; o.fs_qg = o.fs_k = o.fs_nc = out/4; o.fs_y = out/2;
    o.fs_g = gb/8;
    o.fs_cg = in/8; o.fs_perm = in/4;
    o.fs_dist = db/8;
    ctx->pvq_alloc[pli][l] = true;
  }
  return 0;
}

// pointers of frame slot `slot` (PvqSoA holds slot 0)
PvqSoA pvq_slot(const PvqSoA &o, size_t slot) {
  PvqSoA r = o;
  r.cg += slot*o.fs_cg; r.g += slot*o.fs_g; r.cos_dist += slot*o.fs_cd; r.dist += slot*o.fs_dist;
  r.qg += slot*o.fs_qg; r.k += slot*o.fs_k; r.ncand += slot*o.fs_nc; r.y += slot*o.fs_y;
  r.perm += slot*o.fs_perm;
  return r;
}

// Argument block of one (plane, level) PVQ launch group (+ lazy allocation of its outputs).
int pvq_prepare(od_hip_ctx *ctx, int slot0, int nslots, int pli, int level, const int16_t *qm,
                const int32_t *q, const double *beta, PvqCall &c) {
  if (int rc = check_slots(ctx, slot0, nslots, false)) return rc;
  if (!q || !beta) return fail(OD_HIP_EFAULT, "null pointer");
  c.nblk = od_hip_pvq_nblocks(ctx, pli, level);
  if (c.nblk < 0) return c.nblk;
  if (int rc = pvq_arena(ctx, pli)) return rc;
  const PvqLevelLayout &L = ctx->arena[pli].lev[level];
  const int n = c.n = L.n;
  const int bs = c.bs = L.bs;
  PvqLevelArgs &a = c.a;
  a.nbands = od_hip_band_offsets(bs, a.off);
  c.nrec = L.nrec;
  c.ny = L.ny;
  PvqSoA &o = ctx->pvq[pli][level];
  c.o = &o;
  // one QM copy per (plane, level): kernels of earlier calls may still be running on the
  // side streams when the next call uploads its table
  int16_t *qm_d = ctx->qm_slots + ((size_t)pli*4 + level)*1024;
  if (qm) HIPCHK(hipMemcpyAsync(qm_d, qm, (size_t)n*n*sizeof(int16_t), hipMemcpyHostToDevice, ctx->stream));
  a.lev = ctx->lev[pli] + ((size_t)slot0*ctx->nlev[pli] + level)*ctx->psz[pli];
  a.lev_fstride = (size_t)ctx->nlev[pli]*ctx->psz[pli];
  a.w = ctx->pw[pli];
  a.n = n;
  a.nbx = ctx->pw[pli]/n;
  a.nby = ctx->ph[pli]/n;
  for (int i = 0; i < a.nbands; i++) { a.q[i] = q[i]; a.beta[i] = beta[i]; }
  a.tab = ctx->tab[bs];
  a.qm = qm_d;
  a.out = pvq_slot(o, slot0);                   // cg and the work lists are written by the companding stage
  // the strip's blocks of this level: block rows are superblock rows times 32/n (luma units)
  {
    const int per_sb = (32 >> ctx->geo.xdec[pli])/n;
    a.blk_first = (long)ctx->strip0*per_sb*a.nbx;
    a.blk_end = (long)ctx->strip1*per_sb*a.nbx;
  }
  return 0;
}

// One launch per distinct band size (the no-reference sizes are 15, 8, 32, 128).  The
// search launches go round-robin over the side streams (their long tails overlap); the
// cheap gain launches stay on the context's stream.
int pvq_launch(od_hip_ctx *ctx, PvqCall &c, int nslots, bool gain_only) {
  PvqLevelArgs &a = c.a;
  const bool side = !gain_only && ctx->naux > 0;
  if (side) {
    if (ctx->timing && !ctx->aux_pending && !ctx->phase_a) {
      ctx->phase_a = get_event(ctx);
      if (ctx->phase_a) (void)hipEventRecord(ctx->phase_a, ctx->stream);
    }
    HIPCHK(hipEventRecord(ctx->aux_dep, ctx->stream));   // after the pyramid, the table and cg
  }
  static const int sizes[4] = {15, 8, 32, 128};
  for (int si = 0; si < 4; si++) {
    int nlist = 0;
    for (int b = 0; b < a.nbands; b++) {
      if (a.off[b + 1] - a.off[b] == sizes[si]) a.band_list[nlist++] = b;
    }
    if (!nlist) continue;
    hipStream_t ls = ctx->stream;
    if (side) {
      // OD_HIP_PVQ_ASSIGN=1: by kernel class (128 | 32 | 15+8) instead of round-robin
      static const int by_class = env_int("OD_HIP_PVQ_ASSIGN", 0);
      const int cls = sizes[si] == 128 ? 0 : sizes[si] == 32 ? 1 : 2;
      ls = ctx->aux[(by_class ? cls : ctx->aux_rr++) % ctx->naux];
      HIPCHK(hipStreamWaitEvent(ls, ctx->aux_dep, 0));
      ctx->aux_pending = true;
    }
    const char *nm = gain_only ? (sizes[si] == 15 ? "k_pvq_gain<15>" : sizes[si] == 8 ? "k_pvq_gain<8>"
                                  : sizes[si] == 32 ? "k_pvq_gain<32>" : "k_pvq_gain<128>")
                               : (sizes[si] == 15 ? "k_pvq_noref<15>" : sizes[si] == 8 ? "k_pvq_noref<8>"
                                  : sizes[si] == 32 ? "k_pvq_noref<32>" : "k_pvq_noref<128>");
    Timed tm(ctx, nm, ls);
    switch (sizes[si]) {
      case 15: launch_pvq<15>(a, nlist, c.nblk, nslots, ls, gain_only, ctx->rsq, gain_only ? nullptr : ctx->pvq_stats); break;
      case 8: launch_pvq<8>(a, nlist, c.nblk, nslots, ls, gain_only, ctx->rsq, gain_only ? nullptr : ctx->pvq_stats); break;
      case 32: launch_pvq<32>(a, nlist, c.nblk, nslots, ls, gain_only, ctx->rsq, gain_only ? nullptr : ctx->pvq_stats); break;
      default: launch_pvq<128>(a, nlist, c.nblk, nslots, ls, gain_only, ctx->rsq, gain_only ? nullptr : ctx->pvq_stats); break;
    }
    HIPCHK(hipGetLastError());
  }
  return 0;
}
// A level whose bands all have beta == 1 (everything with activity masking off; with it on: the
// chroma planes and the 4x4 luma level) needs no libm: its gain launch already wrote cg = g/q0
// and k_pvq_order builds the work lists - no transfer, no host stage.  OD_HIP_PVQ_DEV_COMPAND=0
// sends those levels through the host stage as well (rounds 1-3; A/B).
bool pvq_level_on_device(const od_hip_ctx *ctx, const double *beta, int nbands) {
  static const int on = env_int("OD_HIP_PVQ_DEV_COMPAND", 1);
  // strips (od_hip_set_strip) and unsorted lists keep the host stage, which writes identity lists
  if (!on || ctx->strip0 != 0 || ctx->strip1 != ctx->nvsb || !pvq_sort_enabled()) return false;
  for (int b = 0; b < nbands; b++) if (beta[b] != 1) return false;
  return true;
}

// the work lists of every band of the level, on the context's stream (after its gain launch):
// count pass, scatter pass (pvq_kernels.hpp); the (frame, band) histograms and cursors live in a
// small per-context scratch that every level reuses in stream order
int pvq_order_launch(od_hip_ctx *ctx, PvqCall &c, int nslots) {
  const long count = c.a.blk_end - c.a.blk_first;
  if (count <= 0) return 0;
  const size_t ints = (size_t)ctx->geo.nslots*10*256;
  if (!ctx->order_scratch) HIPCHK(hipMalloc((void **)&ctx->order_scratch, 2*ints*sizeof(int)));
  PvqOrderArgs oa;
  oa.a = c.a;
  oa.nlist = c.a.nbands;
  oa.gh = ctx->order_scratch;
  oa.gc = ctx->order_scratch + ints;
  for (int b = 0; b < c.a.nbands; b++) oa.a.band_list[b] = b;
  HIPCHK(hipMemsetAsync(ctx->order_scratch, 0, 2*ints*sizeof(int), ctx->stream));
  const dim3 grid((unsigned)((count + PVQ_ORDER_CHUNK - 1)/PVQ_ORDER_CHUNK), c.a.nbands, nslots);
  Timed tm(ctx, "k_pvq_order");
  hipLaunchKernelGGL(k_pvq_order_count, grid, dim3(PVQ_ORDER_THREADS), 0, ctx->stream, oa);
  hipLaunchKernelGGL(k_pvq_order_scatter, grid, dim3(PVQ_ORDER_THREADS), 0, ctx->stream, oa);
  HIPCHK(hipGetLastError());
  return 0;
}
}  // namespace

// Measurement: enable = 1 allocates/zeroes three device counters that every following search
// launch of this context adds to; the call returns their current values (element steps of the
// greedy scans, of the RDO scans, candidates searched); enable = 0 stops counting.
int od_hip_pvq_stats(od_hip_ctx *ctx, int enable, uint64_t out[3]) {
  if (int rc = check_slots(ctx, 0, 1)) return rc;
  if (out) {
    out[0] = out[1] = out[2] = 0;
    if (ctx->pvq_stats) {
      uint64_t all[64];
      HIPCHK(hipMemcpyAsync(all, ctx->pvq_stats, sizeof(all), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      out[0] = all[0]; out[1] = all[1]; out[2] = all[2];
      if (getenv("OD_HIP_PVQ_STAMPS")) {           // diagnostic builds (-DPVQ_STAMPS): cycles per phase
        static const char *cls[4] = {"15", "8", "32", "128"};
        for (int c = 0; c < 4; c++) {
          const uint64_t *p = all + 8 + 8*c;
          if (p[7]) fprintf(stderr, "pvq_stamps N=%s waves=%llu list=%.0f gather=%.0f norms=%.0f search=%.0f out=%.0f (cycles per wave)\n",
                            cls[c], (unsigned long long)p[7], (double)p[0]/p[7], (double)p[1]/p[7], (double)p[2]/p[7],
                            (double)p[3]/p[7], (double)p[4]/p[7]);
        }
      }
    }
  }
  if (enable) {
    if (!ctx->pvq_stats) HIPCHK(hipMalloc((void **)&ctx->pvq_stats, 64*8));
    HIPCHK(hipMemsetAsync(ctx->pvq_stats, 0, 64*8, ctx->stream));
  }
  else if (ctx->pvq_stats) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    (void)hipFree(ctx->pvq_stats);
    ctx->pvq_stats = nullptr;
  }
  return 0;
}

int od_hip_pvq_compand(int count, const double *g, int q0, double beta, double *cg) {
  if (!g || !cg) return fail(OD_HIP_EFAULT, "null pointer");
  if (count < 0 || q0 < 1) return fail(OD_HIP_EINVAL, "bad count/quantiser");
  for (int i = 0; i < count; i++) cg[i] = host_gain_compand(g[i], q0, beta);
  return 0;
}

int od_hip_pvq_gains(od_hip_ctx *ctx, int slot0, int nslots, int pli, int level,
                     const int16_t *qm, const int32_t *q, const double *beta) {
  PvqCall c;
  if (!qm) return fail(OD_HIP_EFAULT, "null pointer");
  if (int rc = pvq_prepare(ctx, slot0, nslots, pli, level, qm, q, beta, c)) return rc;
  return pvq_launch(ctx, c, nslots, true);
}

int od_hip_pvq_compand_level(od_hip_ctx *ctx, int slot0, int nslots, int pli, int level,
                             const int32_t *q, const double *beta) {
  PvqCall c;
  if (int rc = pvq_prepare(ctx, slot0, nslots, pli, level, nullptr, q, beta, c)) return rc;
  // only the strip's records travel: [band][blk_first, blk_end) of every slot
  const long first = c.a.blk_first, count = c.a.blk_end - c.a.blk_first;
  if (count <= 0) return 0;
  if (pvq_level_on_device(ctx, beta, c.a.nbands)) {
    return pvq_order_launch(ctx, c, nslots);        // cg is there since the gain launch: nothing crosses PCIe
  }
  const size_t per = (size_t)count, tot = (size_t)nslots*c.a.nbands*per;
  std::vector<double> g(tot), cg(tot);
  for (int s = 0; s < nslots; s++) {
    for (int b = 0; b < c.a.nbands; b++) {
      HIPCHK(hipMemcpyAsync(g.data() + ((size_t)s*c.a.nbands + b)*per,
                            c.a.out.g + (size_t)s*c.a.out.fs_g + (size_t)b*c.nblk + first, per*8,
                            hipMemcpyDeviceToHost, ctx->stream));
    }
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  std::vector<int32_t> perm(2*tot);
  const bool whole = ctx->strip0 == 0 && ctx->strip1 == ctx->nvsb && pvq_sort_enabled();
  for (int s = 0; s < nslots; s++) {
    for (int b = 0; b < c.a.nbands; b++) {
      const size_t o = ((size_t)s*c.a.nbands + b)*per;
      for (size_t i = 0; i < per; i++) cg[o + i] = host_gain_compand(g[o + i], q[b], beta[b]);
      pvq_block_order(cg.data() + o, first, count, c.a.off[b + 1] - c.a.off[b], beta[b], whole, perm.data() + 2*o);
    }
  }
  for (int s = 0; s < nslots; s++) {
    for (int b = 0; b < c.a.nbands; b++) {
      HIPCHK(hipMemcpyAsync(c.a.out.cg + (size_t)s*c.a.out.fs_cg + (size_t)b*c.nblk + first,
                            cg.data() + ((size_t)s*c.a.nbands + b)*per, per*8,
                            hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(const_cast<int32_t *>(c.a.out.perm) + (size_t)s*c.a.out.fs_perm + (size_t)b*2*c.nblk + 2*first,
                            perm.data() + 2*((size_t)s*c.a.nbands + b)*per, 2*per*4,
                            hipMemcpyHostToDevice, ctx->stream));
    }
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int od_hip_pvq_search(od_hip_ctx *ctx, int slot0, int nslots, int pli, int level,
                      const int16_t *qm, const int32_t *q, const double *beta) {
  PvqCall c;
  if (!qm) return fail(OD_HIP_EFAULT, "null pointer");
  if (int rc = pvq_prepare(ctx, slot0, nslots, pli, level, qm, q, beta, c)) return rc;
  return pvq_launch(ctx, c, nslots, false);
}

int od_hip_pvq_noref_search(od_hip_ctx *ctx, int slot0, int nslots, int pli, int level,
                            const int16_t *qm, const int32_t *q, const double *beta) {
  if (int rc = od_hip_pvq_gains(ctx, slot0, nslots, pli, level, qm, q, beta)) return rc;
  if (int rc = od_hip_pvq_compand_level(ctx, slot0, nslots, pli, level, q, beta)) return rc;
  return od_hip_pvq_search(ctx, slot0, nslots, pli, level, qm, q, beta);
}

int od_hip_pvq_download(od_hip_ctx *ctx, int slot, int pli, int level,
                        od_hip_pvq_band *bands, int32_t *y) {
  if (int rc = check_plane(ctx, slot, pli)) return rc;
  if (int rc = join_aux(ctx)) return rc;
  int nblk = od_hip_pvq_nblocks(ctx, pli, level);
  if (nblk < 0) return nblk;
  if (!ctx->pvq_alloc[pli][level]) return fail(OD_HIP_EINVAL, "no PVQ results for this level");
  const PvqLevelLayout &L = ctx->arena[pli].lev[level];
  const int *off = L.off;
  const int nb = L.nb, ncoded = L.ncoded;
  const size_t nrec = L.nrec, ny = L.ny;
  const PvqSoA o = pvq_slot(ctx->pvq[pli][level], slot);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (bands) {
    std::vector<double> cg(nrec), g(nrec), cd(2*nrec), di(2*nrec);
    std::vector<int32_t> qg(2*nrec), k(2*nrec), nc(nrec);
    HIPCHK(hipMemcpyAsync(cg.data(), o.cg, nrec*8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(g.data(), o.g, nrec*8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(cd.data(), o.cos_dist, 2*nrec*8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(di.data(), o.dist, 2*nrec*8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(qg.data(), o.qg, 2*nrec*4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(k.data(), o.k, 2*nrec*4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(nc.data(), o.ncand, nrec*4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < nb; b++) {
      for (int blk = 0; blk < nblk; blk++) {
        size_t r = (size_t)b*nblk + blk;
        od_hip_pvq_band &d = bands[(size_t)blk*nb + b];
        d.cg = cg[r]; d.g = g[r]; d.ncand = nc[r]; d.pad = 0;
        for (int c = 0; c < 2; c++) {
          d.cos_dist[c] = cd[c*nrec + r]; d.dist[c] = di[c*nrec + r];
          d.qg[c] = qg[c*nrec + r]; d.k[c] = k[c*nrec + r];
        }
      }
    }
  }
  if (y) {
    // device: band-major int16 [band][cand][block][ns_b]  ->  API: int32 [block][cand][ncoded]
    std::vector<int16_t> yd(ny);
    HIPCHK(hipMemcpyAsync(yd.data(), o.y, ny*2, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    memset(y, 0, (size_t)nblk*2*ncoded*sizeof(int32_t));
    for (int b = 0; b < nb; b++) {
      const int nbnd = off[b + 1] - off[b], ns = pvq_ns(off, b);
      const int16_t *src = yd.data() + (size_t)2*nblk*pvq_yo(off, b);
      for (int c = 0; c < 2; c++) {
        for (int blk = 0; blk < nblk; blk++) {
          int32_t *dst = y + ((size_t)blk*2 + c)*ncoded + off[b];
          const int16_t *sp = src + ((size_t)c*nblk + blk)*ns;
          for (int j = 0; j < nbnd; j++) dst[j] = sp[j];
        }
      }
    }
  }
  return 0;
}

int od_hip_pvq_search_vectors(int n, int nvec, const double *x, const int32_t *k,
                              const double *g2, int32_t *y, double *cos_dist) {
  if (!x || !k || !g2 || !y || !cos_dist) return fail(OD_HIP_EFAULT, "null pointer");
  if (n < 1 || n > PVQ_MAXN || nvec < 0) return fail(OD_HIP_EINVAL, "bad n/nvec");
  if (int rc = ensure_device()) return rc;
  if (nvec == 0) return 0;
  size_t nv = nvec;
  SCRATCH_LOCK;
  if (int rc = g_in.reserve(nv*n*8)) return rc;
  if (int rc = g_out.reserve(nv*n*4)) return rc;
  if (int rc = g_aux0.reserve(nv*4)) return rc;
  if (int rc = g_aux1.reserve(nv*8)) return rc;
  if (int rc = g_aux2.reserve(nv*8)) return rc;
  HIPCHK(hipMemcpy(g_in.p, x, nv*n*8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux0.p, k, nv*4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux1.p, g2, nv*8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_pvq_search_vectors, dim3((nvec + 63)/64), dim3(64), 0, 0, n, nvec,
                     (const double *)g_in.p, (const int32_t *)g_aux0.p,
                     (const double *)g_aux1.p, (int32_t *)g_out.p, (double *)g_aux2.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(y, g_out.p, nv*n*4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(cos_dist, g_aux2.p, nv*8, hipMemcpyDeviceToHost));
  return 0;
}

int od_hip_pvq_synthesis_noref(int n, int nvec, const int32_t *y, const double *g,
                               const int16_t *qm_inv, od_coeff *out) {
  if (!y || !g || !qm_inv || !out) return fail(OD_HIP_EFAULT, "null pointer");
  if (n < 1 || nvec < 0) return fail(OD_HIP_EINVAL, "bad n/nvec");
  if (int rc = ensure_device()) return rc;
  if (nvec == 0) return 0;
  size_t nv = nvec;
  SCRATCH_LOCK;
  if (int rc = g_in.reserve(nv*n*4)) return rc;
  if (int rc = g_out.reserve(nv*n*4)) return rc;
  if (int rc = g_aux0.reserve(nv*n*2)) return rc;
  if (int rc = g_aux1.reserve(nv*8)) return rc;
  HIPCHK(hipMemcpy(g_in.p, y, nv*n*4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux0.p, qm_inv, nv*n*2, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux1.p, g, nv*8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_pvq_synthesis_noref, dim3((nvec + 63)/64), dim3(64), 0, 0, n, nvec,
                     (const int32_t *)g_in.p, (const double *)g_aux1.p,
                     (const int16_t *)g_aux0.p, (int32_t *)g_out.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, g_out.p, nv*n*4, hipMemcpyDeviceToHost));
  return 0;
}

// od_pvq_compute_max_theta / _theta / _k (src/pvq.c:476-535) on the host: the candidate
// enumeration of pvq_theta is scalar work around acos/sin/cos, so it runs here, between
// the two device passes.
namespace {
inline int host_max_theta(double qcg, double beta) {
  int ts = (int)floor(.5 + qcg*M_PI/(2*beta));
  if (qcg < 1.4) ts = 1;
  return ts;
}
inline double host_theta(int t, int max_theta) {
  if (max_theta != 0) return (t < max_theta - 1 ? t : max_theta - 1)*.5*M_PI/max_theta;
  return 0;
}
inline int host_k(double qcg, int itheta, double theta, int noref, int n, double beta, int nodesync) {
  if (noref) {
    if (qcg == 0) return 0;
    if (n == 15 && qcg == 1 && beta > 1.25) return 1;
    const int k = (int)floor(.5 + (qcg - .2)*sqrt((n + 3)/2)/beta);
    return k > 1 ? k : 1;
  }
  if (itheta == 0) return 0;
  int k;
  if (nodesync) k = (int)floor(.5 + (itheta - .2)*sqrt((n + 2)/2));
  else k = (int)floor(.5 + (qcg*sin(theta) - .2)*sqrt((n + 2)/2)/beta);
  return k > 1 ? k : 1;
}
}  // namespace

int od_hip_pvq_theta_vectors(int n, int nvec, const od_coeff *x0, const od_coeff *r0,
                             const int16_t *qm, const int32_t *q0, double beta, int robust,
                             int is_keyframe, int pli, od_hip_pvq_theta_out *out,
                             int32_t *y_ref, int32_t *y_noref) {
  if (!x0 || !r0 || !qm || !q0 || !out || !y_ref || !y_noref) return fail(OD_HIP_EFAULT, "null pointer");
  if (n < 2 || n > PVQ_MAXN || nvec < 0) return fail(OD_HIP_EINVAL, "bad n/nvec");
  if (int rc = ensure_device()) return rc;
  if (nvec == 0) return 0;
  size_t nv = nvec;
  SCRATCH_LOCK;
  if (int rc = g_in.reserve(nv*n*4)) return rc;
  if (int rc = g_aux0.reserve(nv*n*4)) return rc;
  if (int rc = g_aux1.reserve((size_t)n*2)) return rc;
  if (int rc = g_aux2.reserve(nv*sizeof(PvqThetaPrep))) return rc;
  if (int rc = g_aux5.reserve(nv*sizeof(PvqThetaCands))) return rc;
  if (int rc = g_out.reserve(nv*sizeof(PvqThetaRes))) return rc;
  if (int rc = g_aux3.reserve(nv*12*n*4)) return rc;
  if (int rc = g_aux4.reserve(nv*2*n*4)) return rc;
  HIPCHK(hipMemcpy(g_in.p, x0, nv*n*4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux0.p, r0, nv*n*4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux1.p, qm, (size_t)n*2, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(g_aux3.p, 0, nv*12*n*4));
  HIPCHK(hipMemset(g_aux4.p, 0, nv*2*n*4));
  // pass 1: gains and correlation sums (exact)
  hipLaunchKernelGGL(k_pvq_theta_prep, dim3((nvec + 63)/64), dim3(64), 0, 0, n, nvec,
                     (const int32_t *)g_in.p, (const int32_t *)g_aux0.p, (const int16_t *)g_aux1.p,
                     (PvqThetaPrep *)g_aux2.p);
  HIPCHK(hipGetLastError());
  std::vector<PvqThetaPrep> prep(nv);
  std::vector<PvqThetaCands> cands(nv);
  std::vector<PvqThetaRes> res(nv);
  HIPCHK(hipMemcpy(prep.data(), g_aux2.p, nv*sizeof(PvqThetaPrep), hipMemcpyDeviceToHost));
  // host stage: everything of pvq_theta (src/pvq_encoder.c:359-417, :452-462) that goes
  // through libm, in the reference's own expressions
  const double gain_weight = 1.4;
  const int nodesync = robust || is_keyframe;
  const int cfl_enabled = is_keyframe && pli != 0;
  memset(out, 0, nv*sizeof(*out));
  for (size_t v = 0; v < nv; v++) {
    od_hip_pvq_theta_out &o = out[v];
    PvqThetaCands &c = cands[v];
    memset(&c, 0, sizeof(c));
    const double g = prep[v].g, gr = prep[v].gr;
    const double cg = host_gain_compand(g, q0[v], beta);
    double cgr = host_gain_compand(gr, q0[v], beta);
    if (cfl_enabled) cgr = 1;
    const int icgr = (int)floor(.5 + cgr);
    const double gain_offset = cgr - icgr;
    double corr = prep[v].corr_sum/(1e-100 + g*gr);
    corr = corr < 1. ? corr : 1.;
    corr = corr > -1. ? corr : -1.;
    o.null_dist = gain_weight*cg*cg;
    if (is_keyframe) o.skip_dist = gain_weight*cg*cg;
    else o.skip_dist = gain_weight*(cg - cgr)*(cg - cgr) + cgr*cg*(2 - 2*corr);
    double theta = 0;
    if (n <= PVQ_MAXN && !prep[v].isnull && corr > 0) {
      o.theta_searched = c.theta_searched = 1;
      theta = acos(corr);
      int i = (int)floor(cg - gain_offset) - 1;
      if (i < 1) i = 1;
      for (; i <= (int)ceil(cg - gain_offset); i++) {
        const double qcg = i + gain_offset;
        const int ts = host_max_theta(qcg, beta);
        int j = (int)floor(.5 + theta*2/M_PI*ts) - 2;
        if (j < 0) j = 0;
        int jhi = (int)ceil(theta*2/M_PI*ts);
        if (jhi > ts - 1) jhi = ts - 1;
        for (; j <= jhi; j++) {
          const int ci = o.nref;
          if (ci >= 12) break;
          const double qtheta = host_theta(j, ts);
          const int k = host_k(qcg, j, qtheta, 0, n, beta, nodesync);
          o.ref_qg[ci] = i; o.ref_itheta[ci] = j; o.ref_ts[ci] = ts; o.ref_k[ci] = k;
          o.ref_qtheta[ci] = qtheta;
          c.ref_k[ci] = k;
          c.ref_g2[ci] = qcg*cg*sin(theta)*sin(qtheta);
          o.nref++;
        }
      }
      c.nref = o.nref;
    }
    if (n <= PVQ_MAXN && ((is_keyframe && pli == 0) || corr < .5 || cg < 2.)) {
      o.noref_searched = c.noref_searched = 1;
      int i = (int)floor(cg);
      if (i < 1) i = 1;
      for (; i <= ceil(cg) && o.nnoref < 2; i++) {
        const int ci = o.nnoref;
        const double qcg = i;
        const int k = host_k(qcg, -1, -1, 1, n, beta, nodesync);
        o.nr_qg[ci] = i; o.nr_k[ci] = k;
        c.nr_k[ci] = k;
        c.nr_g2[ci] = qcg*cg;
        o.nnoref++;
      }
      c.nnoref = o.nnoref;
    }
    o.cg = cg; o.cgr = cgr; o.g = g; o.gr = gr; o.corr = corr; o.theta = theta;
    o.gain_offset = gain_offset; o.icgr = icgr;
  }
  HIPCHK(hipMemcpy(g_aux5.p, cands.data(), nv*sizeof(PvqThetaCands), hipMemcpyHostToDevice));
  // pass 2: Householder + codeword searches (exact)
  hipLaunchKernelGGL(k_pvq_theta_search, dim3((nvec + 63)/64), dim3(64), 0, 0, n, nvec,
                     (const int32_t *)g_in.p, (const int32_t *)g_aux0.p, (const int16_t *)g_aux1.p,
                     (const PvqThetaPrep *)g_aux2.p, (const PvqThetaCands *)g_aux5.p,
                     (PvqThetaRes *)g_out.p, (int32_t *)g_aux3.p, (int32_t *)g_aux4.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(res.data(), g_out.p, nv*sizeof(PvqThetaRes), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(y_ref, g_aux3.p, nv*12*n*4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(y_noref, g_aux4.p, nv*2*n*4, hipMemcpyDeviceToHost));
  // host stage 2: the distortions (:429-431, :465)
  for (size_t v = 0; v < nv; v++) {
    od_hip_pvq_theta_out &o = out[v];
    const double cg = o.cg, theta = o.theta;
    o.m = res[v].m;
    o.s = res[v].s;
    for (int ci = 0; ci < o.nref; ci++) {
      const double qcg = o.ref_qg[ci] + o.gain_offset, qtheta = o.ref_qtheta[ci];
      const double cos_dist = res[v].ref_cos_dist[ci];
      const double dist_theta = 2 - 2*cos(theta - qtheta)
                                + sin(theta)*sin(qtheta)*(2 - 2*cos_dist);
      o.ref_cos_dist[ci] = cos_dist;
      o.ref_dist[ci] = gain_weight*(qcg - cg)*(qcg - cg) + qcg*cg*dist_theta;
    }
    for (int ci = 0; ci < o.nnoref; ci++) {
      const double qcg = o.nr_qg[ci], cd = res[v].nr_cos_dist[ci];
      o.nr_cos_dist[ci] = cd;
      o.nr_dist[ci] = gain_weight*(qcg - cg)*(qcg - cg) + qcg*cg*(2 - 2*cd);
    }
  }
  return 0;
}

int od_hip_pvq_synthesis_vectors(int n, int nvec, const int32_t *y, const od_coeff *ref,
                                 const double *gr, const int32_t *noref, const double *g,
                                 const double *theta, const int16_t *qm, const int16_t *qm_inv,
                                 od_coeff *out) {
  if (!y || !ref || !gr || !noref || !g || !theta || !qm || !qm_inv || !out)
    return fail(OD_HIP_EFAULT, "null pointer");
  if (n < 1 || n > PVQ_MAXN || nvec < 0) return fail(OD_HIP_EINVAL, "bad n/nvec");
  if (int rc = ensure_device()) return rc;
  if (nvec == 0) return 0;
  size_t nv = nvec;
  SCRATCH_LOCK;
  if (int rc = g_in.reserve(nv*n*4)) return rc;
  if (int rc = g_aux0.reserve(nv*n*4)) return rc;
  if (int rc = g_aux1.reserve(nv*8)) return rc;
  if (int rc = g_aux2.reserve(nv*4)) return rc;
  if (int rc = g_aux3.reserve(nv*8)) return rc;
  if (int rc = g_aux4.reserve(nv*16)) return rc;
  if (int rc = g_aux5.reserve((size_t)n*2)) return rc;
  if (int rc = g_aux6.reserve((size_t)n*2)) return rc;
  if (int rc = g_out.reserve(nv*n*4)) return rc;
  // sin(theta), cos(theta) of od_pvq_synthesis_partial (src/pvq.c:574-577): host libm
  std::vector<double> sc(2*nv);
  for (size_t v = 0; v < nv; v++) {
    sc[v] = sin(theta[v]);
    sc[nv + v] = cos(theta[v]);
  }
  HIPCHK(hipMemcpy(g_in.p, y, nv*n*4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux0.p, ref, nv*n*4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux1.p, gr, nv*8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux2.p, noref, nv*4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux3.p, g, nv*8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux4.p, sc.data(), nv*16, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux5.p, qm, (size_t)n*2, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux6.p, qm_inv, (size_t)n*2, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_pvq_synthesis_vectors, dim3((nvec + 63)/64), dim3(64), 0, 0, n, nvec,
                     (const int32_t *)g_in.p, (const int32_t *)g_aux0.p, (const double *)g_aux1.p,
                     (const int32_t *)g_aux2.p, (const double *)g_aux3.p, (const double *)g_aux4.p,
                     (const double *)g_aux4.p + nv,
                     (const int16_t *)g_aux5.p, (const int16_t *)g_aux6.p, (int32_t *)g_out.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, g_out.p, nv*n*4, hipMemcpyDeviceToHost));
  return 0;
}

int od_hip_hv_intra_pred_blocks(const od_coeff *d, int w, int h, const unsigned char *bsize,
                                int bstride, int bs, int nblk, const int32_t *bx,
                                const int32_t *by, od_coeff *pred) {
  if (!d || !bsize || !bx || !by || !pred) return fail(OD_HIP_EFAULT, "null pointer");
  if (bs < 0 || bs >= OD_HIP_NBSIZES || nblk < 0 || w <= 0 || h <= 0 || w%8 || h%8 ||
      bstride < w/8) return fail(OD_HIP_EINVAL, "bad geometry");
  int n = 4 << bs;
  for (int i = 0; i < nblk; i++) {
    if (bx[i] < 0 || by[i] < 0 || (bx[i] << 2) + n > w || (by[i] << 2) + n > h)
      return fail(OD_HIP_EINVAL, "block outside the plane");
  }
  if (int rc = ensure_device()) return rc;
  if (nblk == 0) return 0;
  size_t nb = nblk, bh = h/8;
  SCRATCH_LOCK;
  if (int rc = g_in.reserve((size_t)w*h*4)) return rc;
  if (int rc = g_aux0.reserve(bh*bstride)) return rc;
  if (int rc = g_aux1.reserve(nb*4)) return rc;
  if (int rc = g_aux2.reserve(nb*4)) return rc;
  if (int rc = g_out.reserve(nb*n*n*4)) return rc;
  HIPCHK(hipMemcpy(g_in.p, d, (size_t)w*h*4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux0.p, bsize, bh*bstride, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux1.p, bx, nb*4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux2.p, by, nb*4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_hv_intra_pred_blocks, dim3((nblk + 63)/64), dim3(64), 0, 0,
                     (const int32_t *)g_in.p, w, (const uint8_t *)g_aux0.p, bstride, bs, nblk,
                     (const int32_t *)g_aux1.p, (const int32_t *)g_aux2.p, (int32_t *)g_out.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(pred, g_out.p, nb*n*n*4, hipMemcpyDeviceToHost));
  return 0;
}

int od_hip_compute_dist_blocks(int bs, int nblk, const od_coeff *x, const od_coeff *y,
                               const double *mag2, int activity_masking, double *dist) {
  if (!x || !y || !mag2 || !dist) return fail(OD_HIP_EFAULT, "null pointer");
  if (bs < 1 || bs >= OD_HIP_NBSIZES || nblk < 0) return fail(OD_HIP_EINVAL, "bad bs/nblk");
  if (int rc = ensure_device()) return rc;
  if (nblk == 0) return 0;
  size_t n = 4u << bs, bytes = (size_t)nblk*n*n*4;
  const size_t per = (n/8)*(n/8), nsub = (size_t)nblk*per;
  SCRATCH_LOCK;
  if (int rc = g_in.reserve(bytes)) return rc;
  if (int rc = g_aux0.reserve(bytes)) return rc;
  if (int rc = g_aux1.reserve(64*8)) return rc;
  if (int rc = g_out.reserve(nsub*16)) return rc;
  HIPCHK(hipMemcpy(g_in.p, x, bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux0.p, y, bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux1.p, mag2, 64*8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_compute_dist_blocks, dim3((unsigned)((nsub + 63)/64)), dim3(64), 0, 0, (int)n, nblk,
                     (const int32_t *)g_in.p, (const int32_t *)g_aux0.p, (const double *)g_aux1.p,
                     activity_masking, (double *)g_out.p, (double *)g_out.p + nsub);
  HIPCHK(hipGetLastError());
  std::vector<double> ae(2*nsub);
  HIPCHK(hipMemcpy(ae.data(), g_out.p, nsub*16, hipMemcpyDeviceToHost));
  // od_compute_dist_8x8's activity (src/encode.c:997-1007, :1029) with the host's libm,
  // sub-blocks summed in raster order (:1045-1050), then the 1.7 of :1055
  const double calibration = activity_masking ? 1.95 : 1.62;
  for (int b = 0; b < nblk; b++) {
    double sum = 0;
    for (size_t sb = 0; sb < per; sb++) {
      const double activity = calibration*pow(ae[b*per + sb], -1./6);
      sum += activity*activity*ae[nsub + b*per + sb];
    }
    dist[b] = sum*1.7;
  }
  return 0;
}

namespace {
int ensure_tail_buffers(od_hip_ctx *ctx) {
  if (ctx->dflags) return 0;
  size_t ns = ctx->geo.nslots;
  for (int p = 0; p < ctx->geo.nplanes; p++) {
    HIPCHK(hipMalloc((void **)&ctx->p32[p], ns*ctx->psz[p]*4));
    HIPCHK(hipMalloc((void **)&ctx->bskip[p], ns*(size_t)(ctx->geo.frame_width/4)*(ctx->geo.frame_height/4)));
    HIPCHK(hipMemsetAsync(ctx->bskip[p], 0, ns*(size_t)(ctx->geo.frame_width/4)*(ctx->geo.frame_height/4), ctx->stream));
  }
  HIPCHK(hipMalloc((void **)&ctx->dflags, ns*ctx->nhsb*ctx->nvsb));
  // on the context's stream: a plain hipMemset runs on the null stream, asynchronously to
  // the host, and ctx->stream (non-blocking) does not wait for it - the uploads that
  // follow could be overwritten by a late memset (seen as rare +-1 pixels in multi-worker
  // decodes: zeroed dering flags)
  HIPCHK(hipMemsetAsync(ctx->dflags, 0, ns*ctx->nhsb*ctx->nvsb, ctx->stream));
  return 0;
}
}  // namespace

int od_hip_set_decode_info(od_hip_ctx *ctx, int slot, const unsigned char *dering_flags,
                           const unsigned char *const bskip[], int skip_stride) {
  if (int rc = check_slots(ctx, slot, 1)) return rc;
  if (!dering_flags || !bskip) return fail(OD_HIP_EFAULT, "null pointer");
  if (int rc = ensure_tail_buffers(ctx)) return rc;
  int fw4 = ctx->geo.frame_width/4, fh4 = ctx->geo.frame_height/4;
  size_t nsb = (size_t)ctx->nhsb*ctx->nvsb;
  HIPCHK(hipMemcpyAsync(ctx->dflags + slot*nsb, dering_flags, nsb, hipMemcpyHostToDevice, ctx->stream));
  for (int p = 0; p < ctx->geo.nplanes; p++) {
    if (!bskip[p] || skip_stride < (fw4 >> ctx->geo.xdec[p])) return fail(OD_HIP_EINVAL, "bad skip map");
    int pw4 = fw4 >> ctx->geo.xdec[p], ph4 = fh4 >> ctx->geo.xdec[p];
    HIPCHK(hipMemcpy2DAsync(ctx->bskip[p] + (size_t)slot*fw4*fh4, fw4, bskip[p], skip_stride, pw4, ph4,
                            hipMemcpyHostToDevice, ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int od_hip_decode_tail(od_hip_ctx *ctx, int slot0, int nslots, const int32_t *threshold,
                       const int32_t *quantizer, int is_keyframe) {
  if (int rc = check_slots(ctx, slot0, nslots)) return rc;
  if (!threshold || !quantizer) return fail(OD_HIP_EFAULT, "null pointer");
  if (int rc = ensure_tail_buffers(ctx)) return rc;
  // 1. iDCT + split post-filters (k_inverse_rt) -> c; 2. frame post-filter -> p32
  for (int p = 0; p < ctx->geo.nplanes; p++) {
    InvArgs a;
    a.d = ctx->d[p] + (size_t)slot0*ctx->psz[p];
    a.c = ctx->c[p] + (size_t)slot0*ctx->psz[p];
    a.fstride = ctx->psz[p];
    a.bsize = ctx->bsize + (size_t)slot0*ctx->bsize_sz;
    a.bsize_fstride = ctx->bsize_sz;
    a.bstride = ctx->nhsb*4;
    a.w = ctx->pw[p]; a.h = ctx->ph[p]; a.nhsb = ctx->nhsb; a.nvsb = ctx->nvsb;
    a.pic_w = ctx->geo.pic_width; a.pic_h = ctx->geo.pic_height;
    a.dec = ctx->geo.xdec[p];
    PostArgs q;
    q.c = a.c; q.c_fstride = ctx->psz[p];
    q.rec = nullptr; q.rec_fstride = ctx->psz[p];
    q.out32 = ctx->p32[p] + (size_t)slot0*ctx->psz[p];
    q.w = a.w; q.h = a.h; q.nhsb = a.nhsb; q.nvsb = a.nvsb;
    dim3 grid2(ctx->nhsb + 1, ctx->nvsb + 1, nslots);
    if (a.dec == 0) {
      { Timed tm(ctx, "k_inverse_sb_luma");
        hipLaunchKernelGGL((k_inverse_rt<32, 4>), dim3((ctx->nhsb + 1)/2, ctx->nvsb, nslots), dim3(64), 0, ctx->stream, a); }
      { Timed tm(ctx, "k_postfilter_i32_luma");
        hipLaunchKernelGGL((k_postfilter_clamp<32, false>), grid2, dim3(256), 0, ctx->stream, q); }
    }
    else {
      { Timed tm(ctx, "k_inverse_sb_chroma");
        hipLaunchKernelGGL((k_inverse_rt<16, 3>), dim3((ctx->nhsb + 3)/4, ctx->nvsb, nslots), dim3(64), 0, ctx->stream, a); }
      { Timed tm(ctx, "k_postfilter_i32_chroma");
        hipLaunchKernelGGL((k_postfilter_clamp<16, false>), grid2, dim3(64), 0, ctx->stream, q); }
    }
    HIPCHK(hipGetLastError());
  }
  // 3. dering + smoothing + clamp
  TailArgs t;
  size_t fw4 = ctx->geo.frame_width/4, fh4 = ctx->geo.frame_height/4;
  for (int p = 0; p < 3; p++) {
    bool on = p < ctx->geo.nplanes;
    t.p[p] = on ? ctx->p32[p] + (size_t)slot0*ctx->psz[p] : nullptr;
    t.rec[p] = on ? ctx->rec[p] + (size_t)slot0*ctx->psz[p] : nullptr;
    t.fstride[p] = on ? ctx->psz[p] : 0;
    t.bskip[p] = on ? ctx->bskip[p] + (size_t)slot0*fw4*fh4 : nullptr;
    t.xdec[p] = on ? ctx->geo.xdec[p] : 0;
    t.thr[p] = on ? threshold[p] : 0;
    t.q[p] = on ? quantizer[p] : 0;
  }
  t.flags = ctx->dflags + (size_t)slot0*ctx->nhsb*ctx->nvsb;
  t.bskip_fstride = fw4*fh4;
  t.bsize = ctx->bsize + (size_t)slot0*ctx->bsize_sz;
  t.bsize_fstride = ctx->bsize_sz;
  t.bstride = ctx->nhsb*4;
  t.fw = ctx->geo.frame_width; t.fh = ctx->geo.frame_height;
  t.nhsb = ctx->nhsb; t.nvsb = ctx->nvsb; t.nplanes = ctx->geo.nplanes;
  t.is_keyframe = is_keyframe;
  {
    Timed tm(ctx, "k_decode_tail");
    tail_launch(t, dim3(ctx->nhsb, ctx->nvsb, nslots), ctx->stream);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

int od_hip_libm_probe(int fn, int n, const double *x, const double *y, double *out) {
  if (!x || !y || !out) return fail(OD_HIP_EFAULT, "null pointer");
  if (n < 0 || fn < 0 || fn > 5) return fail(OD_HIP_EINVAL, "bad arguments");
  if (int rc = ensure_device()) return rc;
  if (n == 0) return 0;
  size_t nb = (size_t)n*8;
  SCRATCH_LOCK;
  if (int rc = g_in.reserve(nb)) return rc;
  if (int rc = g_aux0.reserve(nb)) return rc;
  if (int rc = g_out.reserve(nb)) return rc;
  HIPCHK(hipMemcpy(g_in.p, x, nb, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_aux0.p, y, nb, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_libm_probe, dim3((n + 255)/256), dim3(256), 0, 0, fn, n,
                     (const double *)g_in.p, (const double *)g_aux0.p, (double *)g_out.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, g_out.p, nb, hipMemcpyDeviceToHost));
  return 0;
}

int od_hip_calibrate_traffic(int mode, size_t bytes) {
  if (mode < 0 || mode > 2 || bytes < 1024) return fail(OD_HIP_EINVAL, "bad arguments");
  if (int rc = ensure_device()) return rc;
  bytes &= ~(size_t)1023;
  SCRATCH_LOCK;
  if (int rc = g_in.reserve(bytes)) return rc;
  if (int rc = g_aux0.reserve(64)) return rc;
  HIPCHK(hipMemset(g_in.p, 1, bytes));
  HIPCHK(hipDeviceSynchronize());
  dim3 grid(2048), blk(256);
  if (mode == 0)
    hipLaunchKernelGGL(k_calib_read_dword, grid, blk, 0, 0, (const uint32_t *)g_in.p, bytes/4, (uint32_t *)g_aux0.p);
  else if (mode == 1)
    hipLaunchKernelGGL(k_calib_read_int4, grid, blk, 0, 0, (const int4 *)g_in.p, bytes/16, (uint32_t *)g_aux0.p);
  else
    hipLaunchKernelGGL(k_calib_write_int4, grid, blk, 0, 0, (int4 *)g_in.p, bytes/16);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  return 0;
}

int od_hip_host_register(void *ptr, size_t bytes) {
  if (!ptr || !bytes) return fail(OD_HIP_EFAULT, "null pointer");
  if (int rc = ensure_device()) return rc;
  HIPCHK(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  return 0;
}

int od_hip_host_unregister(void *ptr) {
  if (!ptr) return fail(OD_HIP_EFAULT, "null pointer");
  HIPCHK(hipHostUnregister(ptr));
  return 0;
}

int od_hip_device_sync(int device) {
  if (int rc = ensure_device()) return rc;
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipDeviceSynchronize());
  return 0;
}

int od_hip_sync(od_hip_ctx *ctx) {
  if (!ctx) return fail(OD_HIP_EFAULT, "null context");
  HIPCHK(hipSetDevice(ctx->device));
  if (int rc = join_aux(ctx)) return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int od_hip_timing_reset(od_hip_ctx *ctx) {
  if (!ctx) return fail(OD_HIP_EFAULT, "null context");
  if (int rc = join_aux(ctx)) return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (auto &kv : ctx->spans) {
    for (auto &s : kv.second) { ctx->pool.push_back(s.a); ctx->pool.push_back(s.b); }
  }
  ctx->spans.clear();
  ctx->timing = true;
  return 0;
}

int od_hip_timing_get(od_hip_ctx *ctx, const char *kernel, int *launches, double *total_ms) {
  if (!ctx || !kernel || !launches || !total_ms) return fail(OD_HIP_EFAULT, "null pointer");
  if (int rc = join_aux(ctx)) return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *launches = 0;
  *total_ms = 0;
  auto it = ctx->spans.find(kernel);
  if (it == ctx->spans.end()) return 0;
  for (auto &s : it->second) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, s.a, s.b));
    *total_ms += ms;
    (*launches)++;
  }
  return 0;
}

}  // extern "C"

#include "enc_feed.hpp"
#include "pfeed.hpp"
#include "dsynth.hpp"
#include "comm.hpp"
