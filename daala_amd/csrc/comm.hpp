// Superblock-row sharding of ONE frame across the GPUs of a node (BASELINE
// configs[2]): every rank computes the forward pyramid and the PVQ passes of its strip of
// superblock rows (od_hip_set_strip) from the replicated input frame - the kernels read
// their 2-sample lapping halo from the pixels, so a strip needs no neighbour's results and
// no halo recomputation - and the strips are then gathered DEVICE TO DEVICE with RCCL over
// xGMI: one in-place ncclBroadcast per (owner rank, contiguous chunk) inside a single group,
// because the strips need not be equal (nvsb is rarely a multiple of the rank count) and
// every buffer already has the frame's layout on every rank.  Included at the end of
// daala_hip.hip.
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

// sb_rows[world + 1]: rank r owns superblock rows [sb_rows[r], sb_rows[r + 1]).  After the
// call every rank's slot holds the complete pyramid (all planes, all levels) and - with_pvq -
// the complete PVQ records of every (plane, level) whose search ran.
int od_hip_gather_strips(od_hip_ctx *ctx, od_hip_comm *c, int slot, const int *sb_rows, int with_pvq) {
  if (!ctx || !c || !sb_rows) return fail(OD_HIP_EFAULT, "null pointer");
  if (int rc = check_slots(ctx, slot, 1)) return rc;         // joins the PVQ side streams
  if (sb_rows[0] != 0 || sb_rows[c->world] != ctx->nvsb) return fail(OD_HIP_EINVAL, "strips do not cover the frame");
  for (int r = 0; r < c->world; r++) if (sb_rows[r + 1] < sb_rows[r]) return fail(OD_HIP_EINVAL, "strips not ordered");
  NCCLCHK(ncclGroupStart());
  for (int r = 0; r < c->world; r++) {
    const int r0 = sb_rows[r], r1 = sb_rows[r + 1];
    if (r1 <= r0) continue;
    for (int p = 0; p < ctx->geo.nplanes; p++) {
      const int sb = 32 >> ctx->geo.xdec[p];
      const size_t off = (size_t)r0*sb*ctx->pw[p], cnt = (size_t)(r1 - r0)*sb*ctx->pw[p];
      for (int l = 0; l < ctx->nlev[p]; l++) {
        int32_t *q = ctx->lev[p] + ((size_t)slot*ctx->nlev[p] + l)*ctx->psz[p] + off;
        NCCLCHK(ncclBroadcast(q, q, cnt*sizeof(int32_t), ncclChar, r, c->comm, ctx->stream));
      }
      if (!with_pvq) continue;
      for (int l = 0; l < ctx->nlev[p]; l++) {
        if (!ctx->pvq_alloc[p][l]) continue;
        const PvqLevelLayout &Y = ctx->arena[p].lev[l];
        const int n = Y.n, nb = Y.nb;
        const long nbx = ctx->pw[p]/n, nblk = Y.nblk;
        const size_t nrec = Y.nrec;
        const long first = (long)r0*(sb/n)*nbx, count = (long)(r1 - r0)*(sb/n)*nbx;
        const PvqSoA o = pvq_slot(ctx->pvq[p][l], slot);
        for (int b = 0; b < nb; b++) {
          const size_t e = (size_t)b*nblk + first;
          NCCLCHK(ncclBroadcast(o.cg + e, o.cg + e, count*8, ncclChar, r, c->comm, ctx->stream));
          NCCLCHK(ncclBroadcast(o.g + e, o.g + e, count*8, ncclChar, r, c->comm, ctx->stream));
          NCCLCHK(ncclBroadcast(o.ncand + e, o.ncand + e, count*4, ncclChar, r, c->comm, ctx->stream));
          const int ns = pvq_ns(Y.off, b);
          for (int cd = 0; cd < 2; cd++) {
            const size_t e2 = (size_t)cd*nrec + (size_t)b*nblk + first;
            NCCLCHK(ncclBroadcast(o.qg + e2, o.qg + e2, count*4, ncclChar, r, c->comm, ctx->stream));
            NCCLCHK(ncclBroadcast(o.k + e2, o.k + e2, count*4, ncclChar, r, c->comm, ctx->stream));
            NCCLCHK(ncclBroadcast(o.cos_dist + e2, o.cos_dist + e2, count*8, ncclChar, r, c->comm, ctx->stream));
            NCCLCHK(ncclBroadcast(o.dist + e2, o.dist + e2, count*8, ncclChar, r, c->comm, ctx->stream));
            int16_t *y = o.y + (size_t)2*nblk*pvq_yo(Y.off, b) + ((size_t)cd*nblk + first)*ns;
            NCCLCHK(ncclBroadcast(y, y, (size_t)count*ns*2, ncclChar, r, c->comm, ctx->stream));
          }
        }
      }
    }
  }
  NCCLCHK(ncclGroupEnd());
  return 0;
}

}  // extern "C"
