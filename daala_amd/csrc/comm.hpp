// Superblock-row sharding of ONE frame across the GPUs of a node (BASELINE
// configs[2]): every rank computes the forward pyramid and the PVQ passes of its strip of
// superblock rows (od_hip_set_strip) from the replicated input frame - the kernels read
// their 2-sample lapping halo from the pixels, so a strip needs no neighbour's results and
// no halo recomputation.  The strips are then gathered DEVICE TO DEVICE onto the coding rank
// (rank 0, the one that runs the frame's serial entropy/RDO stage) with RCCL over xGMI:
//   1. a small ncclAllGather of every rank's {PVQ plane mask, geometry}: all ranks see all
//      entries and take the SAME decision - a rank whose arenas differ makes everybody return
//      OD_HIP_EINVAL before any payload moves (mismatched send/recv counts are undefined in RCCL);
//   2. every owner packs its strip - planes, gains, records, pulses of every level, a segment
//      list built once per (slot, strip, mask) and cached on the context - into one staging
//      buffer that starts with a 16-byte header {magic, mask, r0, r1} and issues ONE ncclSend;
//   3. rank 0 posts one ncclRecv per owner inside one group, checks every header, unpacks.
// Strips need not be equal (nvsb is rarely a multiple of the rank count).  The same packed
// strip can cross the host instead (od_hip_strip_export / _import) for launchers without RCCL.
// Included at the end of daala_hip.hip.
#pragma once
#include <deque>
#include <rccl/rccl.h>

struct od_hip_comm {
  ncclComm_t comm = nullptr;
  int world = 0, rank = 0, device = 0;
  int32_t *d_meta = nullptr;     // [world + 1][4]: the allgathered {mask, nvsb, width, height}; own entry at [world]
};

#define NCCLCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) \
  return fail(OD_HIP_ENODEV, ncclGetErrorString(r_)); } while (0)

extern "C" {

int od_hip_comm_unique_id(unsigned char id[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "the C-ABI passes the RCCL id as 128 bytes");
  if (!id) return fail(OD_HIP_EFAULT, "null pointer");
  ncclUniqueId u;
  NCCLCHK(ncclGetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return 0;
}

void od_hip_comm_destroy(od_hip_comm *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  if (c->d_meta) (void)hipFree(c->d_meta);
  delete c;
}

od_hip_comm *od_hip_comm_create(int device, int world, int rank, const unsigned char id[128]) {
  if (!id || world < 1 || rank < 0 || rank >= world) { fail(OD_HIP_EINVAL, "bad communicator arguments"); return nullptr; }
  if (ensure_device()) return nullptr;
  if (hipSetDevice(device) != hipSuccess) { fail(OD_HIP_ENODEV, "hipSetDevice failed"); return nullptr; }
  od_hip_comm *c = new od_hip_comm();
  c->world = world; c->rank = rank; c->device = device;
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) {
    fail(OD_HIP_ENODEV, ncclGetErrorString(r));
    delete c;
    return nullptr;
  }
  return c;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// A strip's results as ONE packed buffer.  What a rank computed for its superblock rows lies
// scattered over the frame-shaped buffers: a row range of every level plane and, per band and
// array of the PVQ arenas, a block range.  strip_segments() lists those ranges in a fixed
// order (the same on the owner and on the coder); k_copy_segments packs them into a staging
// buffer on the owner and unpacks them into the frame layout on the coder - so a strip travels
// as one message (round 2: one broadcast per array, band, level and plane).
struct StripSeg {
  int base;                 // which buffer: 0..3 level planes of plane p, 4..7 out arenas, 8..11 g arenas, 12..15 in arenas
  size_t off;               // byte offset inside that buffer
  size_t packed;            // byte offset inside the staging buffer
  size_t bytes;             // multiple of 4
};

namespace {

// pvq_mask: bit p = the PVQ arenas of plane p travel too.  The packed buffer starts with a 16-byte
// header {magic, pvq_mask, r0, r1} so that an importer lays the segments out as the exporter did.
#define STRIP_MAGIC 0x53545250
int strip_mask(const od_hip_ctx *ctx, int with_pvq) {
  int m = 0;
  if (with_pvq) for (int p = 0; p < ctx->geo.nplanes; p++) if (ctx->arena[p].out) m |= 1 << p;
  return m;
}

size_t strip_segments(od_hip_ctx *ctx, int slot, int r0, int r1, int pvq_mask, std::vector<StripSeg> &segs) {
  size_t packed = 16;
  auto add = [&](int base, size_t off, size_t bytes) {
    if (!bytes) return;
    segs.push_back({base, off, packed, bytes});
    packed += (bytes + 15) & ~(size_t)15;
  };
  for (int p = 0; p < ctx->geo.nplanes; p++) {
    const int sb = 32 >> ctx->geo.xdec[p];
    const size_t off = (size_t)r0*sb*ctx->pw[p], cnt = (size_t)(r1 - r0)*sb*ctx->pw[p];
    for (int l = 0; l < ctx->nlev[p]; l++) add(p, (((size_t)slot*ctx->nlev[p] + l)*ctx->psz[p] + off)*4, cnt*4);
    if (!((pvq_mask >> p) & 1) || !ctx->arena[p].out) continue;
    const PvqArena &A = ctx->arena[p];
    for (int l = 0; l < ctx->nlev[p]; l++) {
      const PvqLevelLayout &Y = A.lev[l];
      const long nbx = ctx->pw[p]/Y.n;
      const size_t first = (size_t)r0*(sb/Y.n)*nbx, count = (size_t)(r1 - r0)*(sb/Y.n)*nbx;
      for (int b = 0; b < Y.nb; b++) {
        const size_t e = (size_t)b*Y.nblk + first;
        const size_t ns = pvq_ns(Y.off, b), yo = pvq_yo(Y.off, b);
        for (int cd = 0; cd < 2; cd++) {
          const size_t e2 = (size_t)cd*Y.nrec + e;
          add(4 + p, (size_t)slot*A.out_slot + Y.o_cd + e2*8, count*8);
          add(4 + p, (size_t)slot*A.out_slot + Y.o_qg + e2*4, count*4);
          add(4 + p, (size_t)slot*A.out_slot + Y.o_k + e2*4, count*4);
          add(4 + p, (size_t)slot*A.out_slot + Y.o_y + ((size_t)2*Y.nblk*yo + ((size_t)cd*Y.nblk + first)*ns)*2, count*ns*2);
        }
        add(4 + p, (size_t)slot*A.out_slot + Y.o_nc + e*4, count*4);
        add(8 + p, (size_t)slot*A.g_slot + Y.o_g + e*8, count*8);
        add(12 + p, (size_t)slot*A.in_slot + Y.o_cg + e*8, count*8);
      }
    }
  }
  return packed;
}

struct CopySegArgs {
  char *base[16];
  char *staging;
  const StripSeg *segs;
  int nsegs;
  int unpack;
  int32_t hdr[4];           // pack: written to the first 16 bytes of the staging buffer
};

// one workgroup per (segment, 16 KB chunk)
__global__ __launch_bounds__(256) void k_copy_segments(CopySegArgs a) {
  if (!a.unpack && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 4) {
    reinterpret_cast<int32_t *>(a.staging)[threadIdx.x] = a.hdr[threadIdx.x];
  }
  const StripSeg s = a.segs[blockIdx.x];
  const size_t chunk = (size_t)blockIdx.y*16384;
  if (chunk >= s.bytes) return;
  const size_t lim = s.bytes - chunk < 16384 ? s.bytes - chunk : 16384;
  const uint32_t *src = (const uint32_t *)((a.unpack ? a.staging + s.packed : a.base[s.base] + s.off) + chunk);
  uint32_t *dst = (uint32_t *)((a.unpack ? a.base[s.base] + s.off : a.staging + s.packed) + chunk);
  for (size_t i = threadIdx.x; i < lim/4; i += 256) dst[i] = src[i];
}

struct StripPlan {            // one per (slot, r0, r1, mask), built once, kept on the context
  StripSeg *d_segs = nullptr;
  size_t nsegs = 0, packed = 0, max_seg = 0;
  int slot = -1, r0 = -1, r1 = -1, mask = -1;
};

}  // namespace

struct StripCache {
  std::deque<StripPlan> plans;  // references stay valid while plans are added
  char *staging = nullptr;    // grow-only device staging buffer (pack / receive / unpack)
  size_t staging_cap = 0;
};

static void strip_cache_free(od_hip_ctx *ctx) {
  if (!ctx->strips) return;
  for (auto &p : ctx->strips->plans) if (p.d_segs) (void)hipFree(p.d_segs);
  if (ctx->strips->staging) (void)hipFree(ctx->strips->staging);
  delete ctx->strips;
  ctx->strips = nullptr;
}

namespace {

StripPlan *strip_plan(od_hip_ctx *ctx, int slot, int r0, int r1, int mask) {
  if (!ctx->strips) ctx->strips = new StripCache();
  for (auto &p : ctx->strips->plans) if (p.slot == slot && p.r0 == r0 && p.r1 == r1 && p.mask == mask) return &p;
  std::vector<StripSeg> segs;
  StripPlan P;
  P.packed = strip_segments(ctx, slot, r0, r1, mask, segs);
  P.nsegs = segs.size();
  for (auto &s : segs) P.max_seg = s.bytes > P.max_seg ? s.bytes : P.max_seg;
  P.slot = slot; P.r0 = r0; P.r1 = r1; P.mask = mask;
  if (P.nsegs) {
    if (hipMalloc((void **)&P.d_segs, P.nsegs*sizeof(StripSeg)) != hipSuccess) { fail(OD_HIP_ENODEV, "segment list"); return nullptr; }
    if (hipMemcpy(P.d_segs, segs.data(), P.nsegs*sizeof(StripSeg), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(P.d_segs);
      fail(OD_HIP_ENODEV, "segment list upload failed");
      return nullptr;
    }
  }
  ctx->strips->plans.push_back(P);
  return &ctx->strips->plans.back();
}

// the context's staging buffer, at least `bytes` long (contents are not preserved when it grows)
char *strip_staging(od_hip_ctx *ctx, size_t bytes) {
  if (!ctx->strips) ctx->strips = new StripCache();
  StripCache &C = *ctx->strips;
  if (C.staging_cap < bytes) {
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { fail(OD_HIP_ENODEV, "stream"); return nullptr; }
    if (C.staging) (void)hipFree(C.staging);
    C.staging = nullptr;
    C.staging_cap = 0;
    if (hipMalloc((void **)&C.staging, bytes) != hipSuccess) { fail(OD_HIP_ENODEV, "strip staging buffer"); return nullptr; }
    C.staging_cap = bytes;
  }
  return C.staging;
}

// pack (unpack = 0: frame layout -> staging, header written) or unpack; left in flight on ctx->stream
int strip_copy(od_hip_ctx *ctx, const StripPlan *P, char *staging, int unpack) {
  if (!P->nsegs) return 0;
  CopySegArgs a;
  memset(&a, 0, sizeof(a));
  for (int p = 0; p < ctx->geo.nplanes; p++) {
    a.base[p] = (char *)ctx->lev[p];
    a.base[4 + p] = ctx->arena[p].out;
    a.base[8 + p] = ctx->arena[p].g;
    a.base[12 + p] = ctx->arena[p].in;
  }
  a.staging = staging;
  a.segs = P->d_segs;
  a.nsegs = (int)P->nsegs;
  a.unpack = unpack;
  a.hdr[0] = STRIP_MAGIC; a.hdr[1] = P->mask; a.hdr[2] = P->r0; a.hdr[3] = P->r1;
  hipLaunchKernelGGL(k_copy_segments, dim3((unsigned)P->nsegs, (unsigned)((P->max_seg + 16383)/16384)), dim3(256), 0,
                     ctx->stream, a);
  HIPCHK(hipGetLastError());
  return 0;
}

// does a 16-byte header describe strip [r0, r1) packed with `mask`?
bool strip_header_ok(const int32_t hdr[4], int mask, int r0, int r1) {
  return hdr[0] == STRIP_MAGIC && hdr[1] == mask && hdr[2] == r0 && hdr[3] == r1;
}

}  // namespace

extern "C" {

// Size of rank-strip [r0, r1)'s packed buffer.
long od_hip_strip_bytes(od_hip_ctx *ctx, int slot, int sb_row0, int sb_row1, int with_pvq) {
  if (!ctx) return fail(OD_HIP_EFAULT, "null context");
  if (sb_row0 < 0 || sb_row1 > ctx->nvsb || sb_row1 < sb_row0) return fail(OD_HIP_EINVAL, "bad strip");
  HIPCHK(hipSetDevice(ctx->device));
  const StripPlan *P = strip_plan(ctx, slot, sb_row0, sb_row1, strip_mask(ctx, with_pvq));
  return P ? (long)P->packed : OD_HIP_ENODEV;
}

// Host transports (a launcher without RCCL between its ranks, e.g. a gloo rehearsal on one
// GPU): the owner exports its strip as one packed host buffer, the coder imports it.
int od_hip_strip_export(od_hip_ctx *ctx, int slot, int sb_row0, int sb_row1, int with_pvq, void *host, long cap) {
  if (!ctx || !host) return fail(OD_HIP_EFAULT, "null pointer");
  if (int rc = check_slots(ctx, slot, 1)) return rc;
  if (sb_row0 < 0 || sb_row1 > ctx->nvsb || sb_row1 < sb_row0) return fail(OD_HIP_EINVAL, "bad strip");
  const StripPlan *P = strip_plan(ctx, slot, sb_row0, sb_row1, strip_mask(ctx, with_pvq));
  if (!P) return OD_HIP_ENODEV;
  if (cap < (long)P->packed) return fail(OD_HIP_ENOSPC, "strip buffer too small");
  char *st = strip_staging(ctx, P->packed);
  if (!st) return OD_HIP_ENODEV;
  if (int rc = strip_copy(ctx, P, st, 0)) return rc;
  HIPCHK(hipMemcpyAsync(host, st, P->packed, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (!P->nsegs) {           // an empty strip: nothing was launched, the header is written here
    const int32_t hdr[4] = {STRIP_MAGIC, P->mask, sb_row0, sb_row1};
    memcpy(host, hdr, sizeof(hdr));
  }
  return 0;
}

int od_hip_strip_import(od_hip_ctx *ctx, int slot, int sb_row0, int sb_row1, int with_pvq, const void *host, long bytes) {
  if (!ctx || !host) return fail(OD_HIP_EFAULT, "null pointer");
  if (int rc = check_slots(ctx, slot, 1)) return rc;
  if (sb_row0 < 0 || sb_row1 > ctx->nvsb || sb_row1 < sb_row0) return fail(OD_HIP_EINVAL, "bad strip");
  if (bytes < 16) return fail(OD_HIP_EINVAL, "strip buffer has no header");
  int32_t hdr[4];
  memcpy(hdr, host, sizeof(hdr));
  if (hdr[0] != STRIP_MAGIC || hdr[2] != sb_row0 || hdr[3] != sb_row1 || (hdr[1] & ~((1 << ctx->geo.nplanes) - 1))
      || (!with_pvq && hdr[1])) return fail(OD_HIP_EINVAL, "strip buffer does not describe this strip");
  // the arenas the exporter packed must exist here BEFORE the layout is computed
  for (int p = 0; p < ctx->geo.nplanes; p++) if ((hdr[1] >> p) & 1) if (int rc = pvq_arena(ctx, p)) return rc;
  const StripPlan *P = strip_plan(ctx, slot, sb_row0, sb_row1, hdr[1]);
  if (!P) return OD_HIP_ENODEV;
  if (bytes != (long)P->packed) return fail(OD_HIP_EINVAL, "strip buffer has the wrong size");
  char *st = strip_staging(ctx, P->packed);
  if (!st) return OD_HIP_ENODEV;
  HIPCHK(hipMemcpyAsync(st, host, P->packed, hipMemcpyHostToDevice, ctx->stream));
  if (int rc = strip_copy(ctx, P, st, 1)) return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

// sb_rows[world + 1]: rank r owns superblock rows [sb_rows[r], sb_rows[r + 1]).  After the
// call the CODING rank's (rank 0's) slot holds the complete pyramid (all planes, all levels)
// and - with_pvq - the complete PVQ records of every plane whose searches ran: the frame
// gather of the north star (steps 1-3 of this file's header).  Every rank calls it; every rank
// gets the same return code for a disagreement about what travels.
int od_hip_gather_strips(od_hip_ctx *ctx, od_hip_comm *c, int slot, const int *sb_rows, int with_pvq) {
  if (!ctx || !c || !sb_rows) return fail(OD_HIP_EFAULT, "null pointer");
  if (int rc = check_slots(ctx, slot, 1)) return rc;         // joins the PVQ side streams
  if (sb_rows[0] != 0 || sb_rows[c->world] != ctx->nvsb) return fail(OD_HIP_EINVAL, "strips do not cover the frame");
  for (int r = 0; r < c->world; r++) if (sb_rows[r + 1] < sb_rows[r]) return fail(OD_HIP_EINVAL, "strips not ordered");
  if (c->world == 1) return 0;
  HIPCHK(hipSetDevice(ctx->device));
  const int mask = strip_mask(ctx, with_pvq);
  // 1. agree on what travels: every rank's {mask, geometry} to every rank
  {
    const int W = c->world;
    if (!c->d_meta) HIPCHK(hipMalloc((void **)&c->d_meta, (size_t)(W + 1)*4*sizeof(int32_t)));
    const int32_t mine[4] = {mask, ctx->nvsb, ctx->geo.frame_width, ctx->geo.frame_height};
    HIPCHK(hipMemcpyAsync(c->d_meta + 4*W, mine, sizeof(mine), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));                 // `mine` is on the stack
    NCCLCHK(ncclAllGather(c->d_meta + 4*W, c->d_meta, 4, ncclInt32, c->comm, ctx->stream));
    std::vector<int32_t> all((size_t)W*4);
    HIPCHK(hipMemcpyAsync(all.data(), c->d_meta, all.size()*sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int r = 1; r < W; r++) {
      if (memcmp(&all[4*r], &all[0], 4*sizeof(int32_t)) != 0)
        return fail(OD_HIP_EINVAL, "the ranks disagree about the planes or the geometry of the strips");
    }
  }
  // 2./3. payload: sizes follow from the agreed mask and the common geometry
  if (c->rank != 0) {
    const StripPlan *P = strip_plan(ctx, slot, sb_rows[c->rank], sb_rows[c->rank + 1], mask);
    if (!P) return OD_HIP_ENODEV;
    if (!P->nsegs) return 0;                                   // an empty strip sends nothing
    char *st = strip_staging(ctx, P->packed);
    if (!st) return OD_HIP_ENODEV;
    if (int rc = strip_copy(ctx, P, st, 0)) return rc;
    NCCLCHK(ncclSend(st, P->packed, ncclChar, 0, c->comm, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
  }
  std::vector<const StripPlan *> plan(c->world, nullptr);
  size_t total = 0;
  for (int r = 1; r < c->world; r++) {
    plan[r] = strip_plan(ctx, slot, sb_rows[r], sb_rows[r + 1], mask);
    if (!plan[r]) return OD_HIP_ENODEV;
    if (plan[r]->nsegs) total += (plan[r]->packed + 255) & ~(size_t)255;
  }
  if (total == 0) return 0;
  char *st = strip_staging(ctx, total);
  if (!st) return OD_HIP_ENODEV;
  int rc = 0;
  ncclResult_t g_ = ncclGroupStart();
  if (g_ != ncclSuccess) rc = fail(OD_HIP_ENODEV, ncclGetErrorString(g_));
  size_t o = 0;
  for (int r = 1; r < c->world && rc == 0; r++) {
    if (!plan[r]->nsegs) continue;
    ncclResult_t r_ = ncclRecv(st + o, plan[r]->packed, ncclChar, r, c->comm, ctx->stream);
    if (r_ != ncclSuccess) rc = fail(OD_HIP_ENODEV, ncclGetErrorString(r_));
    o += (plan[r]->packed + 255) & ~(size_t)255;
  }
  g_ = ncclGroupEnd();                                 // closed on the error path too
  if (g_ != ncclSuccess && rc == 0) rc = fail(OD_HIP_ENODEV, ncclGetErrorString(g_));
  if (rc) return rc;
  // every header is checked before anything is unpacked into the frame
  std::vector<int32_t> hdrs((size_t)c->world*4, 0);
  o = 0;
  for (int r = 1; r < c->world; r++) {
    if (!plan[r]->nsegs) continue;
    HIPCHK(hipMemcpyAsync(&hdrs[4*r], st + o, 16, hipMemcpyDeviceToHost, ctx->stream));
    o += (plan[r]->packed + 255) & ~(size_t)255;
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (int r = 1; r < c->world; r++) {
    if (plan[r]->nsegs && !strip_header_ok(&hdrs[4*r], mask, sb_rows[r], sb_rows[r + 1]))
      return fail(OD_HIP_EINVAL, "a received strip does not describe the rows its owner was given");
  }
  o = 0;
  for (int r = 1; r < c->world; r++) {
    if (!plan[r]->nsegs) continue;
    if (int rc2 = strip_copy(ctx, plan[r], st + o, 1)) return rc2;
    o += (plan[r]->packed + 255) & ~(size_t)255;
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

}  // extern "C"
