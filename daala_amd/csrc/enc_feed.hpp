// Encoder feed (include/daala_hip.h section 4b): batch launch of the state-free
// keyframe-luma work + asynchronous hand-over to pinned host memory, one event per
// frame slot so that host workers start on a frame as soon as its records landed.
// Included at the end of daala_hip.hip (needs od_hip_ctx internals).
#pragma once

struct od_hip_enc_feed {
  od_hip_ctx *ctx = nullptr;
  hipStream_t copy = nullptr;
  hipStream_t up = nullptr;               // small transfers of the companding round trip (g down, cg up)
  hipEvent_t gains_done = nullptr;
  hipEvent_t computed = nullptr;
  hipEvent_t transformed = nullptr;       // forward pyramid done (the level planes can be copied)
  std::vector<hipEvent_t> ready;          // one per slot: records landed on the host
  std::vector<hipEvent_t> gready;         // one per slot: gains landed on the host
  std::vector<hipEvent_t> cgup;           // one per slot: companded gains uploaded
  std::vector<char> companded;            // slot went through od_hip_enc_feed_compand since its gains
  od_coeff *haar[3] = {nullptr, nullptr, nullptr};   // lossless frames: pinned [slot][h][w] Haar planes (lazy)
  std::vector<char> lossless;             // slot holds Haar planes, not the PVQ feed
  std::vector<char> pending;              // slot has a copy in flight / landed
  // Per plane in the feed (luma always; the chroma planes once od_hip_enc_feed_set_level_plane
  // has named their levels): pinned host mirrors of the plane's per-slot arenas (PvqArena) - ONE
  // transfer per frame slot and direction: records + pulses down, gains down, companded gains +
  // work lists up - and one for the level planes of a slot (they are contiguous in the context).
  struct Lev {
    bool set = false;
    std::vector<int16_t> qm;
    int32_t q[11];
    double beta[11];
  };
  struct Plane {
    bool on = false;
    char *h_out = nullptr, *h_g = nullptr, *h_in = nullptr;
    od_coeff *h_planes = nullptr;         // [slot][level][h][w]
    Lev lev[4];
  } P[3];
};

namespace {
// pinned mirrors of plane p (its arenas are allocated with it)
int feed_plane_on(od_hip_enc_feed *f, int p) {
  od_hip_ctx *ctx = f->ctx;
  auto &Q = f->P[p];
  if (Q.on) return 0;
  if (int rc = pvq_arena(ctx, p)) return rc;
  const PvqArena &A = ctx->arena[p];
  const size_t ns = ctx->geo.nslots;
  HIPCHK(hipHostMalloc((void **)&Q.h_out, ns*A.out_slot));
  HIPCHK(hipHostMalloc((void **)&Q.h_g, ns*A.g_slot));
  HIPCHK(hipHostMalloc((void **)&Q.h_in, ns*A.in_slot));
  HIPCHK(hipHostMalloc((void **)&Q.h_planes, ns*ctx->nlev[p]*ctx->psz[p]*sizeof(od_coeff)));
  Q.on = true;
  return 0;
}
}  // namespace

extern "C" {

void od_hip_enc_feed_destroy(od_hip_enc_feed *f) {
  if (!f) return;
  (void)hipSetDevice(f->ctx->device);
  if (f->copy) (void)hipStreamSynchronize(f->copy);
  for (auto &Q : f->P) {
    if (Q.h_out) (void)hipHostFree(Q.h_out);
    if (Q.h_g) (void)hipHostFree(Q.h_g);
    if (Q.h_in) (void)hipHostFree(Q.h_in);
    if (Q.h_planes) (void)hipHostFree(Q.h_planes);
  }
  for (auto &h : f->haar) if (h) (void)hipHostFree(h);
  for (auto e : f->ready) if (e) (void)hipEventDestroy(e);
  for (auto e : f->gready) if (e) (void)hipEventDestroy(e);
  for (auto e : f->cgup) if (e) (void)hipEventDestroy(e);
  if (f->gains_done) (void)hipEventDestroy(f->gains_done);
  if (f->up) { (void)hipStreamSynchronize(f->up); (void)hipStreamDestroy(f->up); }
  if (f->computed) (void)hipEventDestroy(f->computed);
  if (f->transformed) (void)hipEventDestroy(f->transformed);
  if (f->copy) (void)hipStreamDestroy(f->copy);
  delete f;
}

od_hip_enc_feed *od_hip_enc_feed_create(od_hip_ctx *ctx) {
  if (!ctx) { fail(OD_HIP_EFAULT, "null context"); return nullptr; }
  if (hipSetDevice(ctx->device) != hipSuccess) { fail(OD_HIP_ENODEV, "hipSetDevice failed"); return nullptr; }
  od_hip_enc_feed *f = new od_hip_enc_feed();
  f->ctx = ctx;
  size_t ns = ctx->geo.nslots;
  bool ok = hipStreamCreateWithFlags(&f->copy, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipStreamCreateWithFlags(&f->up, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&f->computed, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&f->gains_done, hipEventDisableTiming) == hipSuccess;
  f->gready.assign(ns, nullptr);
  f->cgup.assign(ns, nullptr);
  f->companded.assign(ns, 0);
  f->lossless.assign(ns, 0);
  for (size_t s = 0; ok && s < ns; s++) {
    ok = hipEventCreateWithFlags(&f->gready[s], hipEventDisableTiming) == hipSuccess
      && hipEventCreateWithFlags(&f->cgup[s], hipEventDisableTiming) == hipSuccess;
  }
  ok = ok && hipEventCreateWithFlags(&f->transformed, hipEventDisableTiming) == hipSuccess;
  f->ready.assign(ns, nullptr);
  f->pending.assign(ns, 0);
  for (size_t s = 0; ok && s < ns; s++)
    ok = hipEventCreateWithFlags(&f->ready[s], hipEventDisableTiming) == hipSuccess;
  ok = ok && feed_plane_on(f, 0) == 0;
  if (!ok) {
    fail(OD_HIP_ENODEV, "encoder feed allocation failed");
    od_hip_enc_feed_destroy(f);
    return nullptr;
  }
  return f;
}

// Level parameters of plane pli.  Naming a chroma plane's level puts that plane into the feed:
// its forward transforms and its no-reference candidates - state-free on a keyframe like luma's
// (src/pvq_encoder.c:449-455 runs the chroma search when the CfL reference correlates badly) -
// then travel with luma's.  All levels of a plane in the feed must be set before a run.
int od_hip_enc_feed_set_level_plane(od_hip_enc_feed *f, int pli, int level, const int16_t *qm,
                                    const int32_t *q, const double *beta) {
  if (!f || !qm || !q || !beta) return fail(OD_HIP_EFAULT, "null pointer");
  od_hip_ctx *ctx = f->ctx;
  if (pli < 0 || pli >= ctx->geo.nplanes || pli > 2) return fail(OD_HIP_EINVAL, "plane out of range");
  if (level < 0 || level >= ctx->nlev[pli]) return fail(OD_HIP_EINVAL, "level out of range");
  HIPCHK(hipSetDevice(ctx->device));
  if (int rc = feed_plane_on(f, pli)) return rc;
  auto &L = f->P[pli].lev[level];
  const PvqLevelLayout &Y = ctx->arena[pli].lev[level];
  L.qm.assign(qm, qm + Y.n*Y.n);
  for (int b = 0; b < Y.nb; b++) { L.q[b] = q[b]; L.beta[b] = beta[b]; }
  L.set = true;
  return 0;
}

int od_hip_enc_feed_set_level(od_hip_enc_feed *f, int level, const int16_t *qm,
                              const int32_t *q, const double *beta) {
  return od_hip_enc_feed_set_level_plane(f, 0, level, qm, q, beta);
}

// Phase 1: forward pyramid + exact gains g of every band of the four luma levels; the
// level planes and the gains start travelling to the host.
int od_hip_enc_feed_gains(od_hip_enc_feed *f, int slot0, int nslots) {
  if (!f) return fail(OD_HIP_EFAULT, "null feed");
  od_hip_ctx *ctx = f->ctx;
  if (int rc = check_slots(ctx, slot0, nslots)) return rc;
  for (int p = 0; p < 3; p++) {
    if (!f->P[p].on) continue;
    for (int l = 0; l < ctx->nlev[p]; l++) if (!f->P[p].lev[l].set) return fail(OD_HIP_EINVAL, "feed level parameters not set");
  }
  // a slot's host mirror must not be overwritten while a previous copy is in flight
  for (int s = slot0; s < slot0 + nslots; s++) {
    if (f->pending[s]) HIPCHK(hipEventSynchronize(f->ready[s]));
    f->pending[s] = 0;
    f->companded[s] = 0;
    f->lossless[s] = 0;
  }
  if (int rc = od_hip_forward_pyramid(ctx, slot0, nslots)) return rc;
  // the level planes are final here: their copies overlap everything that follows
  HIPCHK(hipEventRecord(f->transformed, ctx->stream));
  HIPCHK(hipStreamWaitEvent(f->copy, f->transformed, 0));
  for (int s = slot0; s < slot0 + nslots; s++) {
    for (int p = 0; p < 3; p++) {
      if (!f->P[p].on) continue;
      const size_t pl = (size_t)ctx->nlev[p]*ctx->psz[p];
      HIPCHK(hipMemcpyAsync(f->P[p].h_planes + (size_t)s*pl, ctx->lev[p] + (size_t)s*pl, pl*sizeof(od_coeff),
                            hipMemcpyDeviceToHost, f->copy));
    }
  }
  for (int p = 0; p < 3; p++) {
    if (!f->P[p].on) continue;
    for (int l = 0; l < ctx->nlev[p]; l++) {
      auto &L = f->P[p].lev[l];
      if (int rc = od_hip_pvq_gains(ctx, slot0, nslots, p, l, L.qm.data(), L.q, L.beta)) return rc;
    }
  }
  HIPCHK(hipEventRecord(f->gains_done, ctx->stream));
  // Levels whose bands all have beta == 1 need nothing from the host: their gain launch wrote
  // cg = g/q0, the work lists are built on the device and the searches start right here - they
  // run while the host compands the other levels (masking off: every level, no host stage at all).
  for (int p = 0; p < 3; p++) {
    if (!f->P[p].on) continue;
    for (int l = 0; l < ctx->nlev[p]; l++) {
      auto &L = f->P[p].lev[l];
      if (!pvq_level_on_device(ctx, L.beta, ctx->arena[p].lev[l].nb)) continue;
      if (int rc = od_hip_pvq_compand_level(ctx, slot0, nslots, p, l, L.q, L.beta)) return rc;
      if (int rc = od_hip_pvq_search(ctx, slot0, nslots, p, l, L.qm.data(), L.q, L.beta)) return rc;
    }
  }
  HIPCHK(hipStreamWaitEvent(f->up, f->gains_done, 0));
  for (int s = slot0; s < slot0 + nslots; s++) {
    for (int p = 0; p < 3; p++) {
      if (!f->P[p].on) continue;
      const PvqArena &A = ctx->arena[p];
      HIPCHK(hipMemcpyAsync(f->P[p].h_g + (size_t)s*A.g_slot, A.g + (size_t)s*A.g_slot, A.g_slot, hipMemcpyDeviceToHost, f->up));
    }
    HIPCHK(hipEventRecord(f->gready[s], f->up));
  }
  return 0;
}

// Phase 2, per slot, any host thread (slots are independent): cg = od_gain_compand(g) with
// the host's libm for every band of the slot and the work lists of the searches, then the upload.
int od_hip_enc_feed_compand(od_hip_enc_feed *f, int slot) {
  if (!f) return fail(OD_HIP_EFAULT, "null feed");
  od_hip_ctx *ctx = f->ctx;
  if (slot < 0 || slot >= ctx->geo.nslots) return fail(OD_HIP_EINVAL, "slot out of range");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipEventSynchronize(f->gready[slot]));
  for (int p = 0; p < 3; p++) {
    if (!f->P[p].on) continue;
    auto &Q = f->P[p];
    const PvqArena &A = ctx->arena[p];
    for (int l = 0; l < ctx->nlev[p]; l++) {
      auto &L = Q.lev[l];
      const PvqLevelLayout &Y = A.lev[l];
      const double *g = (const double *)(Q.h_g + (size_t)slot*A.g_slot + Y.o_g);
      double *cg = (double *)(Q.h_in + (size_t)slot*A.in_slot + Y.o_cg);
      int32_t *perm = (int32_t *)(Q.h_in + (size_t)slot*A.in_slot + Y.o_perm);
      // the strip of the context (od_hip_set_strip; the whole frame by default): its blocks'
      // gains are companded and its work list ordered - the search launch walks the list
      // positions of the strip's blocks only
      const int per_sb = (32 >> ctx->geo.xdec[p])/Y.n;
      const long nbx = ctx->pw[p]/Y.n;
      const long first = (long)ctx->strip0*per_sb*nbx, count = (long)(ctx->strip1 - ctx->strip0)*per_sb*nbx;
      const bool on_device = pvq_level_on_device(ctx, L.beta, Y.nb);
      for (int b = 0; b < Y.nb; b++) {
        const int q0 = L.q[b];
        const double beta = L.beta[b];
        const size_t o = (size_t)b*Y.nblk + first;
        for (long i = 0; i < count; i++) cg[o + i] = host_gain_compand(g[o + i], q0, beta);
        // a level the device companded itself: the host keeps cg (the same quotient g/q0) for
        // the consumer of the feed; the work list exists on the device already
        if (!on_device) pvq_block_order(cg + o, first, count, Y.off[b + 1] - Y.off[b], beta, pvq_sort_enabled(), perm + 2*o);
      }
      // one transfer per level that went through the host: its companded gains and work lists
      if (!on_device) {
        const size_t bytes = Y.o_perm + 2*Y.nrec*4 - Y.o_cg;
        HIPCHK(hipMemcpyAsync(A.in + (size_t)slot*A.in_slot + Y.o_cg, Q.h_in + (size_t)slot*A.in_slot + Y.o_cg, bytes,
                              hipMemcpyHostToDevice, f->up));
      }
    }
  }
  HIPCHK(hipEventRecord(f->cgup[slot], f->up));
  f->companded[slot] = 1;
  return 0;
}

// Phase 3: the codeword searches with the uploaded cg, then the records travel to the host.
int od_hip_enc_feed_search(od_hip_enc_feed *f, int slot0, int nslots) {
  if (!f) return fail(OD_HIP_EFAULT, "null feed");
  od_hip_ctx *ctx = f->ctx;
  if (int rc = check_slots(ctx, slot0, nslots)) return rc;
  for (int s = slot0; s < slot0 + nslots; s++) {
    if (!f->companded[s]) return fail(OD_HIP_EINVAL, "slot was not companded (od_hip_enc_feed_compand)");
    HIPCHK(hipStreamWaitEvent(ctx->stream, f->cgup[s], 0));
  }
  for (int p = 0; p < 3; p++) {
    if (!f->P[p].on) continue;
    for (int l = 0; l < ctx->nlev[p]; l++) {
      auto &L = f->P[p].lev[l];
      if (pvq_level_on_device(ctx, L.beta, ctx->arena[p].lev[l].nb)) continue;     // searched since od_hip_enc_feed_gains
      if (int rc = od_hip_pvq_search(ctx, slot0, nslots, p, l, L.qm.data(), L.q, L.beta)) return rc;
    }
  }
  if (int rc = join_aux(ctx)) return rc;          // the PVQ launches run on side streams
  HIPCHK(hipEventRecord(f->computed, ctx->stream));
  HIPCHK(hipStreamWaitEvent(f->copy, f->computed, 0));
  for (int s = slot0; s < slot0 + nslots; s++) {
    for (int p = 0; p < 3; p++) {
      if (!f->P[p].on) continue;
      const PvqArena &A = ctx->arena[p];
      HIPCHK(hipMemcpyAsync(f->P[p].h_out + (size_t)s*A.out_slot, A.out + (size_t)s*A.out_slot, A.out_slot,
                            hipMemcpyDeviceToHost, f->copy));
    }
    HIPCHK(hipEventRecord(f->ready[s], f->copy));
    f->pending[s] = 1;
  }
  return 0;
}

// After od_hip_gather_strips on the coding rank: the slots' device buffers now hold the whole
// frame (the other ranks' strips arrived device to device), so the host mirrors are fetched
// again - level planes, gains, companded gains, records - and the slots' events re-armed.
int od_hip_enc_feed_refresh(od_hip_enc_feed *f, int slot0, int nslots) {
  if (!f) return fail(OD_HIP_EFAULT, "null feed");
  od_hip_ctx *ctx = f->ctx;
  if (int rc = check_slots(ctx, slot0, nslots)) return rc;
  HIPCHK(hipEventRecord(f->computed, ctx->stream));
  HIPCHK(hipStreamWaitEvent(f->copy, f->computed, 0));
  for (int s = slot0; s < slot0 + nslots; s++) {
    for (int p = 0; p < 3; p++) {
      if (!f->P[p].on) continue;
      auto &Q = f->P[p];
      const PvqArena &A = ctx->arena[p];
      const size_t pl = (size_t)ctx->nlev[p]*ctx->psz[p];
      HIPCHK(hipMemcpyAsync(Q.h_planes + (size_t)s*pl, ctx->lev[p] + (size_t)s*pl, pl*sizeof(od_coeff),
                            hipMemcpyDeviceToHost, f->copy));
      HIPCHK(hipMemcpyAsync(Q.h_g + (size_t)s*A.g_slot, A.g + (size_t)s*A.g_slot, A.g_slot, hipMemcpyDeviceToHost, f->copy));
      HIPCHK(hipMemcpyAsync(Q.h_in + (size_t)s*A.in_slot, A.in + (size_t)s*A.in_slot, A.in_slot, hipMemcpyDeviceToHost, f->copy));
      HIPCHK(hipMemcpyAsync(Q.h_out + (size_t)s*A.out_slot, A.out + (size_t)s*A.out_slot, A.out_slot,
                            hipMemcpyDeviceToHost, f->copy));
    }
    HIPCHK(hipEventRecord(f->ready[s], f->copy));
    f->pending[s] = 1;
    f->lossless[s] = 0;
  }
  return 0;
}

// Lossless frames (quantizer 0): no lapping, no DCT, no PVQ - the encoder
// transforms every whole superblock with od_haar (src/encode.c:1305, :1129).  One pass
// (od_hip_forward_haar) produces the three coefficient planes of every slot; they travel to
// pinned host memory, one completion event per slot.
int od_hip_enc_feed_run_lossless(od_hip_enc_feed *f, int slot0, int nslots) {
  if (!f) return fail(OD_HIP_EFAULT, "null feed");
  od_hip_ctx *ctx = f->ctx;
  if (int rc = check_slots(ctx, slot0, nslots)) return rc;
  if (ctx->geo.nplanes != 3) return fail(OD_HIP_EINVAL, "lossless feed needs 3 planes");
  const size_t ns = ctx->geo.nslots;
  for (int p = 0; p < 3; p++) {
    if (!f->haar[p]) HIPCHK(hipHostMalloc((void **)&f->haar[p], ns*ctx->psz[p]*sizeof(od_coeff)));
  }
  for (int s = slot0; s < slot0 + nslots; s++) {
    if (f->pending[s]) HIPCHK(hipEventSynchronize(f->ready[s]));
    f->pending[s] = 0;
  }
  if (int rc = od_hip_forward_haar(ctx, slot0, nslots)) return rc;
  HIPCHK(hipEventRecord(f->computed, ctx->stream));
  HIPCHK(hipStreamWaitEvent(f->copy, f->computed, 0));
  for (int s = slot0; s < slot0 + nslots; s++) {
    for (int p = 0; p < 3; p++) {
      HIPCHK(hipMemcpyAsync(f->haar[p] + (size_t)s*ctx->psz[p], ctx->d[p] + (size_t)s*ctx->psz[p],
                            ctx->psz[p]*sizeof(od_coeff), hipMemcpyDeviceToHost, f->copy));
    }
    HIPCHK(hipEventRecord(f->ready[s], f->copy));
    f->pending[s] = 1;
    f->lossless[s] = 1;
  }
  return 0;
}

int od_hip_enc_feed_haar_view(od_hip_enc_feed *f, int slot, const od_coeff *planes[3], int strides[3]) {
  if (!f || !planes || !strides) return fail(OD_HIP_EFAULT, "null pointer");
  if (slot < 0 || slot >= f->ctx->geo.nslots) return fail(OD_HIP_EINVAL, "slot out of range");
  if (!f->pending[slot] || !f->lossless[slot]) return fail(OD_HIP_EINVAL, "no lossless run covers this slot");
  HIPCHK(hipSetDevice(f->ctx->device));
  HIPCHK(hipEventSynchronize(f->ready[slot]));
  for (int p = 0; p < 3; p++) {
    planes[p] = f->haar[p] + (size_t)slot*f->ctx->psz[p];
    strides[p] = f->ctx->pw[p];
  }
  return 0;
}

// The three phases back to back on the calling thread.
int od_hip_enc_feed_run(od_hip_enc_feed *f, int slot0, int nslots) {
  if (int rc = od_hip_enc_feed_gains(f, slot0, nslots)) return rc;
  for (int s = slot0; s < slot0 + nslots; s++) {
    if (int rc = od_hip_enc_feed_compand(f, s)) return rc;
  }
  return od_hip_enc_feed_search(f, slot0, nslots);
}

int od_hip_enc_feed_view_plane(od_hip_enc_feed *f, int slot, int pli, od_hip_feed_level lev[4]) {
  if (!f || !lev) return fail(OD_HIP_EFAULT, "null pointer");
  if (slot < 0 || slot >= f->ctx->geo.nslots) return fail(OD_HIP_EINVAL, "slot out of range");
  if (pli < 0 || pli > 2 || !f->P[pli].on) return fail(OD_HIP_EINVAL, "plane is not in the feed");
  if (!f->pending[slot] || f->lossless[slot]) return fail(OD_HIP_EINVAL, "no feed run covers this slot");
  HIPCHK(hipSetDevice(f->ctx->device));       // callers are host worker threads
  HIPCHK(hipEventSynchronize(f->ready[slot]));
  const auto &Q = f->P[pli];
  const PvqArena &A = f->ctx->arena[pli];
  const char *out = Q.h_out + (size_t)slot*A.out_slot;
  const char *in = Q.h_in + (size_t)slot*A.in_slot;
  const char *hg = Q.h_g + (size_t)slot*A.g_slot;
  memset(lev, 0, 4*sizeof(lev[0]));
  for (int l = 0; l < f->ctx->nlev[pli]; l++) {
    const PvqLevelLayout &Y = A.lev[l];
    od_hip_feed_level &v = lev[l];
    v.n = Y.n;
    v.nbands = Y.nb;
    v.nblk = Y.nblk;
    v.nbx = f->ctx->pw[pli]/Y.n;
    for (int i = 0; i < 11; i++) v.off[i] = i <= Y.nb ? Y.off[i] : 0;
    v.pad = 0;
    v.ncand = (const int32_t *)(out + Y.o_nc);
    v.k = (const int32_t *)(out + Y.o_k);
    v.qg = (const int32_t *)(out + Y.o_qg);
    v.cos_dist = (const double *)(out + Y.o_cd);
    v.y = (const int16_t *)(out + Y.o_y);
    v.cg = (const double *)(in + Y.o_cg);
    v.g = (const double *)(hg + Y.o_g);
    v.lev = Q.h_planes + ((size_t)slot*f->ctx->nlev[pli] + l)*f->ctx->psz[pli];
    v.lev_stride = f->ctx->pw[pli];
    v.pad2 = 0;
  }
  return 0;
}

int od_hip_enc_feed_view(od_hip_enc_feed *f, int slot, od_hip_feed_level lev[4]) {
  return od_hip_enc_feed_view_plane(f, slot, 0, lev);
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Encoder-side deringing (include/daala_hip.h section 4c): a small per-worker object -
// own stream, page-locked staging, three int32 input planes and three int16 output
// planes on the device - that runs od_dering() for EVERY superblock of one frame in one
// launch of k_decode_tail's encoder mode.
struct od_hip_dering {
  int device = 0, fw = 0, fh = 0, nplanes = 0, xdec[3] = {0, 0, 0};
  hipStream_t stream = nullptr;
  bool own_stream = false;        // false: leased from the per-device pool (daala_hip.hip: lease_stream)
  size_t psz[3] = {0, 0, 0};
  int16_t *d_in[3] = {nullptr, nullptr, nullptr};
  int16_t *d_out[3] = {nullptr, nullptr, nullptr};
  uint8_t *d_bskip[3] = {nullptr, nullptr, nullptr};
  int16_t *h_in[3] = {nullptr, nullptr, nullptr};      // pinned
  int16_t *h_out[3] = {nullptr, nullptr, nullptr};     // pinned
  // distortions of the on/off decision (od_hip_dering_run_dist): lazily allocated
  uint8_t *d_orig = nullptr, *h_orig = nullptr;        // padded 8-bit luma input
  double *d_mag2 = nullptr, *d_dist = nullptr, *h_dist = nullptr;   // [3][nsb*16]: arg, unfiltered, filtered
};

extern "C" {

void od_hip_dering_destroy(od_hip_dering *d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  if (d->stream) (void)hipStreamSynchronize(d->stream);
  for (int p = 0; p < 3; p++) {
    if (d->d_in[p]) (void)hipFree(d->d_in[p]);
    if (d->d_out[p]) (void)hipFree(d->d_out[p]);
    if (d->d_bskip[p]) (void)hipFree(d->d_bskip[p]);
    if (d->h_in[p]) (void)hipHostFree(d->h_in[p]);
    if (d->h_out[p]) (void)hipHostFree(d->h_out[p]);
  }
  if (d->d_orig) (void)hipFree(d->d_orig);
  if (d->h_orig) (void)hipHostFree(d->h_orig);
  if (d->d_mag2) (void)hipFree(d->d_mag2);
  if (d->d_dist) (void)hipFree(d->d_dist);
  if (d->h_dist) (void)hipHostFree(d->h_dist);
  if (d->stream && d->own_stream) (void)hipStreamDestroy(d->stream);
  delete d;
}

od_hip_dering *od_hip_dering_create(int device, int frame_width, int frame_height, int nplanes,
                                    const int *xdec) {
  if (!xdec || nplanes < 1 || nplanes > 3 || frame_width <= 0 || frame_height <= 0 ||
      frame_width%32 || frame_height%32) {
    fail(OD_HIP_EINVAL, "invalid geometry");
    return nullptr;
  }
  if (ensure_device()) return nullptr;
  if (hipSetDevice(device) != hipSuccess) { fail(OD_HIP_ENODEV, "hipSetDevice failed"); return nullptr; }
  od_hip_dering *d = new od_hip_dering();
  d->device = device;
  d->fw = frame_width;
  d->fh = frame_height;
  d->nplanes = nplanes;
  d->stream = lease_stream(device, d->own_stream);
  bool ok = d->stream != nullptr;
  for (int p = 0; ok && p < nplanes; p++) {
    d->xdec[p] = xdec[p];
    d->psz[p] = (size_t)(frame_width >> xdec[p])*(frame_height >> xdec[p]);
    const size_t nsk = (size_t)(frame_width/4)*(frame_height/4);
    ok = ok && hipMalloc((void **)&d->d_in[p], d->psz[p]*2) == hipSuccess;
    ok = ok && hipMalloc((void **)&d->d_out[p], d->psz[p]*2) == hipSuccess;
    ok = ok && hipMalloc((void **)&d->d_bskip[p], nsk) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&d->h_in[p], d->psz[p]*2) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&d->h_out[p], d->psz[p]*2) == hipSuccess;
  }
  if (!ok) {
    fail(OD_HIP_ENODEV, "dering object allocation failed");
    od_hip_dering_destroy(d);
    return nullptr;
  }
  return d;
}

static int dering_run_impl(od_hip_dering *d, const int16_t *const in[], const unsigned char *const bskip[],
                           int skip_stride, const int32_t *threshold, const int32_t *quantizer,
                           int16_t *const out[], const unsigned char *orig, int orig_stride,
                           const double *mag2, int masking, double *dist_arg, double *dist_unf,
                           double *dist_filt) {
  if (!d || !in || !bskip || !threshold || !quantizer || !out) return fail(OD_HIP_EFAULT, "null pointer");
  HIPCHK(hipSetDevice(d->device));
  const int fw4 = d->fw/4, fh4 = d->fh/4;
  const bool dist = orig != nullptr;
  const size_t nsub = (size_t)(d->fw/32)*(d->fh/32)*16;
  if (dist) {
    if (!mag2 || !dist_arg || !dist_unf || !dist_filt || orig_stride < d->fw) return fail(OD_HIP_EFAULT, "bad distortion arguments");
    if (!d->d_orig) {
      HIPCHK(hipMalloc((void **)&d->d_orig, d->psz[0]));
      HIPCHK(hipHostMalloc((void **)&d->h_orig, d->psz[0]));
      HIPCHK(hipMalloc((void **)&d->d_mag2, 64*8));
      HIPCHK(hipMalloc((void **)&d->d_dist, 3*nsub*8));
      HIPCHK(hipHostMalloc((void **)&d->h_dist, 3*nsub*8));
    }
    for (int y = 0; y < d->fh; y++) memcpy(d->h_orig + (size_t)y*d->fw, orig + (size_t)y*orig_stride, d->fw);
    HIPCHK(hipMemcpyAsync(d->d_orig, d->h_orig, d->psz[0], hipMemcpyHostToDevice, d->stream));
    HIPCHK(hipMemcpyAsync(d->d_mag2, mag2, 64*8, hipMemcpyHostToDevice, d->stream));
  }
  TailArgs t;
  memset(&t, 0, sizeof(t));
  for (int p = 0; p < d->nplanes; p++) {
    if (!in[p] || !bskip[p] || !out[p] || skip_stride < (fw4 >> d->xdec[p]))
      return fail(OD_HIP_EINVAL, "bad plane");
    memcpy(d->h_in[p], in[p], d->psz[p]*2);                      // pageable -> pinned staging
    HIPCHK(hipMemcpyAsync(d->d_in[p], d->h_in[p], d->psz[p]*2, hipMemcpyHostToDevice, d->stream));
    HIPCHK(hipMemsetAsync(d->d_bskip[p], 0, (size_t)fw4*fh4, d->stream));
    HIPCHK(hipMemcpy2DAsync(d->d_bskip[p], fw4, bskip[p], skip_stride, fw4 >> d->xdec[p],
                            fh4 >> d->xdec[p], hipMemcpyHostToDevice, d->stream));
    t.p16[p] = d->d_in[p];
    t.o16[p] = d->d_out[p];
    t.fstride[p] = d->psz[p];
    t.bskip[p] = d->d_bskip[p];
    t.xdec[p] = d->xdec[p];
    t.thr[p] = threshold[p];
    t.q[p] = quantizer[p];
  }
  t.flags = nullptr;                  // encoder mode: every superblock, int16 out
  t.bskip_fstride = (size_t)fw4*fh4;
  t.fw = d->fw; t.fh = d->fh; t.nhsb = d->fw/32; t.nvsb = d->fh/32; t.nplanes = d->nplanes;
  t.is_keyframe = 1;
  tail_launch(t, dim3(t.nhsb, t.nvsb, 1), d->stream);
  HIPCHK(hipGetLastError());
  if (dist) {
    const long total = (long)t.nhsb*t.nvsb*32;
    hipLaunchKernelGGL(k_dering_dist, dim3((unsigned)((total + 63)/64)), dim3(64), 0, d->stream, t.nhsb, t.nvsb,
                       d->fw, (const uint8_t *)d->d_orig, d->fw, (const int16_t *)d->d_in[0],
                       (const int16_t *)d->d_out[0], (const double *)d->d_mag2, masking, d->d_dist,
                       d->d_dist + nsub, d->d_dist + 2*nsub);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(d->h_dist, d->d_dist, 3*nsub*8, hipMemcpyDeviceToHost, d->stream));
  }
  for (int p = 0; p < d->nplanes; p++)
    HIPCHK(hipMemcpyAsync(d->h_out[p], d->d_out[p], d->psz[p]*2, hipMemcpyDeviceToHost, d->stream));
  HIPCHK(hipStreamSynchronize(d->stream));
  for (int p = 0; p < d->nplanes; p++) memcpy(out[p], d->h_out[p], d->psz[p]*2);
  if (dist) {
    memcpy(dist_arg, d->h_dist, nsub*8);
    memcpy(dist_unf, d->h_dist + nsub, nsub*8);
    memcpy(dist_filt, d->h_dist + 2*nsub, nsub*8);
  }
  return 0;
}

int od_hip_dering_run(od_hip_dering *d, const int16_t *const in[], const unsigned char *const bskip[],
                      int skip_stride, const int32_t *threshold, const int32_t *quantizer,
                      int16_t *const out[]) {
  return dering_run_impl(d, in, bskip, skip_stride, threshold, quantizer, out, nullptr, 0, nullptr, 0,
                         nullptr, nullptr, nullptr);
}

int od_hip_dering_run_dist(od_hip_dering *d, const int16_t *const in[], const unsigned char *const bskip[],
                           int skip_stride, const int32_t *threshold, const int32_t *quantizer,
                           int16_t *const out[], const unsigned char *orig_luma, int orig_stride,
                           const double *mag2, int activity_masking, double *dist_arg,
                           double *dist_unfiltered, double *dist_filtered) {
  if (!orig_luma) return fail(OD_HIP_EFAULT, "null pointer");
  return dering_run_impl(d, in, bskip, skip_stride, threshold, quantizer, out, orig_luma, orig_stride,
                         mag2, activity_masking, dist_arg, dist_unfiltered, dist_filtered);
}

}  // extern "C"
