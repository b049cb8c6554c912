// Superblock-row sharding of ONE frame across the GPUs of a node (BASELINE
// configs[2]): every rank computes the forward pyramid and the PVQ passes of its strip of
// superblock rows (od_hip_set_strip) from the replicated input frame - the kernels read
// their 2-sample lapping halo from the pixels, so a strip needs no neighbour's results and
// no halo recomputation - and the strips are then gathered DEVICE TO DEVICE with RCCL over
// xGMI: one in-place ncclBroadcast per (owner rank, contiguous chunk) inside a single group,
// because the strips need not be equal (nvsb is rarely a multiple of the rank count) and
// every buffer already has the frame's layout on every rank.  Included at the end of
// daala_hip.hip.  (Round 3: the strips travel packed, one message per owner, to the coding rank.)
#pragma once
#include <rccl/rccl.h>

struct od_hip_comm {
  ncclComm_t comm = nullptr;
  int world = 0, rank = 0, device = 0;
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
};

// one workgroup per (segment, 16 KB chunk)
__global__ __launch_bounds__(256) void k_copy_segments(CopySegArgs a) {
  const StripSeg s = a.segs[blockIdx.x];
  const size_t chunk = (size_t)blockIdx.y*16384;
  if (chunk >= s.bytes) return;
  const size_t lim = s.bytes - chunk < 16384 ? s.bytes - chunk : 16384;
  const uint32_t *src = (const uint32_t *)((a.unpack ? a.staging + s.packed : a.base[s.base] + s.off) + chunk);
  uint32_t *dst = (uint32_t *)((a.unpack ? a.base[s.base] + s.off : a.staging + s.packed) + chunk);
  for (size_t i = threadIdx.x; i < lim/4; i += 256) dst[i] = src[i];
}

struct StripPlan {            // cached on the context per (slot, r0, r1, with_pvq)
  std::vector<StripSeg> segs;
  StripSeg *d_segs = nullptr;
  size_t packed = 0, max_seg = 0;
  int slot = -1, r0 = -1, r1 = -1, with_pvq = -1;
};

int strip_copy(od_hip_ctx *ctx, int slot, int r0, int r1, int pvq_mask, char *staging, int unpack) {
  std::vector<StripSeg> segs;
  strip_segments(ctx, slot, r0, r1, pvq_mask, segs);
  if (segs.empty()) return 0;
  StripSeg *d = nullptr;
  HIPCHK(hipMalloc((void **)&d, segs.size()*sizeof(StripSeg)));
  HIPCHK(hipMemcpyAsync(d, segs.data(), segs.size()*sizeof(StripSeg), hipMemcpyHostToDevice, ctx->stream));
  CopySegArgs a;
  memset(&a, 0, sizeof(a));
  size_t mx = 0;
  for (auto &s : segs) mx = s.bytes > mx ? s.bytes : mx;
  for (int p = 0; p < ctx->geo.nplanes; p++) {
    a.base[p] = (char *)ctx->lev[p];
    a.base[4 + p] = ctx->arena[p].out;
    a.base[8 + p] = ctx->arena[p].g;
    a.base[12 + p] = ctx->arena[p].in;
  }
  a.staging = staging;
  a.segs = d;
  a.nsegs = (int)segs.size();
  a.unpack = unpack;
  hipLaunchKernelGGL(k_copy_segments, dim3((unsigned)segs.size(), (unsigned)((mx + 16383)/16384)), dim3(256), 0,
                     ctx->stream, a);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));       // the segment list is freed below
  (void)hipFree(d);
  return 0;
}

}  // namespace

extern "C" {

// Size of rank-strip [r0, r1)'s packed buffer.
long od_hip_strip_bytes(od_hip_ctx *ctx, int slot, int sb_row0, int sb_row1, int with_pvq) {
  if (!ctx) return fail(OD_HIP_EFAULT, "null context");
  if (sb_row0 < 0 || sb_row1 > ctx->nvsb || sb_row1 < sb_row0) return fail(OD_HIP_EINVAL, "bad strip");
  std::vector<StripSeg> segs;
  return (long)strip_segments(ctx, slot, sb_row0, sb_row1, strip_mask(ctx, with_pvq), segs);
}

// Host transports (a launcher without RCCL between its ranks, e.g. a gloo rehearsal on one
// GPU): the owner exports its strip as one packed host buffer, the coder imports it.
int od_hip_strip_export(od_hip_ctx *ctx, int slot, int sb_row0, int sb_row1, int with_pvq, void *host, long cap) {
  if (!ctx || !host) return fail(OD_HIP_EFAULT, "null pointer");
  if (int rc = check_slots(ctx, slot, 1)) return rc;
  const long need = od_hip_strip_bytes(ctx, slot, sb_row0, sb_row1, with_pvq);
  if (need < 0) return (int)need;
  if (cap < need) return fail(OD_HIP_ENOSPC, "strip buffer too small");
  char *st = nullptr;
  HIPCHK(hipMalloc((void **)&st, need));
  const int mask = strip_mask(ctx, with_pvq);
  int rc = strip_copy(ctx, slot, sb_row0, sb_row1, mask, st, 0);
  if (rc == 0 && hipMemcpy(host, st, need, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(OD_HIP_ENODEV, "strip download failed");
  (void)hipFree(st);
  if (rc == 0) {
    const int32_t hdr[4] = {STRIP_MAGIC, mask, sb_row0, sb_row1};
    memcpy(host, hdr, sizeof(hdr));
  }
  return rc;
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
  std::vector<StripSeg> segs;
  const long need = (long)strip_segments(ctx, slot, sb_row0, sb_row1, hdr[1], segs);
  if (bytes != need) return fail(OD_HIP_EINVAL, "strip buffer has the wrong size");
  char *st = nullptr;
  HIPCHK(hipMalloc((void **)&st, need));
  int rc = 0;
  if (hipMemcpy(st, host, need, hipMemcpyHostToDevice) != hipSuccess) rc = fail(OD_HIP_ENODEV, "strip upload failed");
  if (rc == 0) rc = strip_copy(ctx, slot, sb_row0, sb_row1, hdr[1], st, 1);
  (void)hipFree(st);
  return rc;
}

// sb_rows[world + 1]: rank r owns superblock rows [sb_rows[r], sb_rows[r + 1]).  After the
// call the CODING rank's (rank 0's) slot holds the complete pyramid (all planes, all levels)
// and - with_pvq - the complete PVQ records of every plane whose searches ran: the frame
// gather of the north star.  One ncclSend per owner, one ncclRecv per owner on rank 0, all in
// one group, device to device over xGMI; pack and unpack are one kernel each.
int od_hip_gather_strips(od_hip_ctx *ctx, od_hip_comm *c, int slot, const int *sb_rows, int with_pvq) {
  if (!ctx || !c || !sb_rows) return fail(OD_HIP_EFAULT, "null pointer");
  if (int rc = check_slots(ctx, slot, 1)) return rc;         // joins the PVQ side streams
  if (sb_rows[0] != 0 || sb_rows[c->world] != ctx->nvsb) return fail(OD_HIP_EINVAL, "strips do not cover the frame");
  for (int r = 0; r < c->world; r++) if (sb_rows[r + 1] < sb_rows[r]) return fail(OD_HIP_EINVAL, "strips not ordered");
  if (c->world == 1) return 0;
  std::vector<long> need(c->world, 0);
  long total = 0;
  for (int r = 1; r < c->world; r++) {
    need[r] = od_hip_strip_bytes(ctx, slot, sb_rows[r], sb_rows[r + 1], with_pvq);
    if (need[r] < 0) return (int)need[r];
    total += need[r];
  }
  char *st = nullptr;
  int rc = 0;
  if (c->rank != 0) {
    if (need[c->rank] == 0) return 0;
    HIPCHK(hipMalloc((void **)&st, need[c->rank]));
    rc = strip_copy(ctx, slot, sb_rows[c->rank], sb_rows[c->rank + 1], strip_mask(ctx, with_pvq), st, 0);
    if (rc == 0) {
      ncclResult_t r_ = ncclSend(st, need[c->rank], ncclChar, 0, c->comm, ctx->stream);
      if (r_ != ncclSuccess) rc = fail(OD_HIP_ENODEV, ncclGetErrorString(r_));
    }
    if (rc == 0 && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(OD_HIP_ENODEV, "strip send failed");
    (void)hipFree(st);
    return rc;
  }
  if (total == 0) return 0;
  // every rank runs the same passes, so the same planes have arenas everywhere: rank 0's own
  // set defines what is packed (a rank that searched other planes would send another size -
  // ncclRecv then fails instead of misplacing data)
  HIPCHK(hipMalloc((void **)&st, total));
  ncclResult_t g_ = ncclGroupStart();
  if (g_ != ncclSuccess) rc = fail(OD_HIP_ENODEV, ncclGetErrorString(g_));
  long o = 0;
  for (int r = 1; r < c->world && rc == 0; r++) {
    if (!need[r]) continue;
    ncclResult_t r_ = ncclRecv(st + o, need[r], ncclChar, r, c->comm, ctx->stream);
    if (r_ != ncclSuccess) rc = fail(OD_HIP_ENODEV, ncclGetErrorString(r_));
    o += need[r];
  }
  g_ = ncclGroupEnd();                                 // closed on the error path too
  if (g_ != ncclSuccess && rc == 0) rc = fail(OD_HIP_ENODEV, ncclGetErrorString(g_));
  o = 0;
  for (int r = 1; r < c->world && rc == 0; r++) {
    if (!need[r]) continue;
    rc = strip_copy(ctx, slot, sb_rows[r], sb_rows[r + 1], strip_mask(ctx, with_pvq), st + o, 1);
    o += need[r];
  }
  if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == 0) rc = fail(OD_HIP_ENODEV, "strip receive failed");
  (void)hipFree(st);
  return rc;
}

}  // extern "C"
