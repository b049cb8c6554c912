// Decoder-side PVQ synthesis of an inter frame on the device (SURVEY 8 row A17 live; C-ABI in
// include/daala_hip.h section 4e).  On a P frame the reference of every band is the transform
// of the motion-compensated prediction (od_decode_compute_pred, src/decode.c:349-357: pred = md),
// which the decoder's context already holds as the forward pyramid of the prediction - so
// pvq_synthesis (src/pvq_decoder.c:104-118) needs nothing from the serial symbol parse but the
// symbols themselves.  The host parses (qg, theta, pulses) into records; here:
//   od_hip_dsynth_ref_gains   gr of every band of every level (od_pvq_compute_gain's *g,
//                             src/pvq.c:456-464; the parse needs it to de-interleave the gain)
//   od_hip_dsynth_run         k_dsynth_blocks: od_init_skipped_coeffs (src/state.c:1351-1357: the
//                             block starts as the prediction's transform) + the DC
//                             (src/decode.c:599-609); k_dsynth_bands: per coded band
//                             od_compute_householder + od_pvq_synthesis_partial
//                             (src/pvq.c:364-413, :552-585) or the zero of OD_PVQ_SKIP_ZERO,
//                             scattered from coding order to the raster (src/partition.c:176)
//                             into the context's coefficient planes - where od_hip_decode_tail
//                             expects them.  No coefficient plane crosses PCIe in either direction.
// sin(theta), cos(theta) and the gain expansion are the host's libm; one
// thread per band runs the reference's loops in their order, so every coefficient is
// bit-identical by construction.  Included at the end of daala_hip.hip.
#pragma once

struct DsBlock {             // == od_hip_dsynth_block
  int32_t org;               // raster origin of the block in its plane
  int32_t dc;                // DC minus the prediction's DC
  uint8_t pli, bs, pad[2];
};
struct DsBand {              // == od_hip_dsynth_band
  uint32_t block;            // index into the block list
  uint8_t band, mode, pad[2];
  uint32_t yoff;             // first pulse (16-bit entries)
  uint32_t pad2;
  double g, sin_theta, cos_theta;
};
enum { DS_ZERO = 0, DS_NOREF = 1, DS_REF = 2, DS_WIDE = 4 };   // WIDE: a pulse is two 16-bit entries (low, high)

struct DsPlanes {
  const int32_t *lev[3][4];  // prediction pyramid level planes (slot 0)
  int32_t *d[3];
  int w[3];
  const uint16_t *tab[4];
  const int16_t *qm[3][4], *qm_inv[3][4];
};

__global__ __launch_bounds__(64) void k_dsynth_blocks(DsPlanes P, const DsBlock *__restrict__ blocks, int nblocks) {
  const int b = blockIdx.x;
  if (b >= nblocks) return;
  const DsBlock q = blocks[b];
  const int n = 4 << q.bs, w = P.w[q.pli];
  const int32_t *md = P.lev[q.pli][(q.pli ? 2 : 3) - q.bs] + q.org;
  int32_t *d = P.d[q.pli] + q.org;
  for (int e = threadIdx.x; e < n*n; e += 64) {
    const int r = e/n, c = e - r*n;
    int32_t v = md[(size_t)r*w + c];
    if (e == 0) v += q.dc;
    d[(size_t)r*w + c] = v;
  }
}

__global__ __launch_bounds__(64) void k_dsynth_bands(DsPlanes P, const DsBlock *__restrict__ blocks,
                                                     const DsBand *__restrict__ bands, int nbands,
                                                     const int16_t *__restrict__ pulses) {
  const long v = (long)blockIdx.x*64 + threadIdx.x;
  if (v >= nbands) return;
  const DsBand B = bands[v];
  const DsBlock q = blocks[B.block];
  static const int off_all[] = {1, 16, 24, 32, 64, 96, 128, 256, 384, 512};
  const int o0 = off_all[B.band], n = off_all[B.band + 1] - o0;
  const int bn = 4 << q.bs, lg = 2 + q.bs, w = P.w[q.pli];
  const int lvl = (q.pli ? 2 : 3) - q.bs;
  const uint16_t *tab = P.tab[q.bs] + o0;
  const int32_t *md = P.lev[q.pli][lvl] + q.org;
  int32_t *d = P.d[q.pli] + q.org;
  auto pos = [&](int i) { const int rt = tab[i]; return (size_t)(rt >> lg)*w + (rt & (bn - 1)); };
  if ((B.mode & 3) == DS_ZERO) {
    for (int i = 0; i < n; i++) d[pos(i)] = 0;
    return;
  }
  const int16_t *yq = pulses + B.yoff;
  const int16_t *qm = P.qm[q.pli][lvl] + o0, *qm_inv = P.qm_inv[q.pli][lvl] + o0;
  const double g = B.g;
  const int noref = (B.mode & 3) == DS_NOREF;
  const bool wide = B.mode & DS_WIDE;
  auto ypv = [&](int i) -> int32_t {
    return wide ? (int32_t)((uint32_t)(uint16_t)yq[2*i] | ((uint32_t)(uint16_t)yq[2*i + 1] << 16)) : (int32_t)yq[i];
  };
  const int nn = n - !noref;
  int yy = 0;
  for (int i = 0; i < nn; i++) yy += ypv(i)*ypv(i);
  double scale = yy == 0 ? 0 : g/sqrt((double)yy);
  if (noref) {
    for (int i = 0; i < n; i++) {
      d[pos(i)] = (int32_t)floor(.5 + (ypv(i)*scale)*(qm_inv[i]*PVQ_QM_INV_SCALE_1));
    }
    return;
  }
  // the reference vector, its gain, the reflection (pvq_synthesis, od_compute_householder)
  double acc = 0, maxr = 0;
  int m = 0;
  for (int i = 0; i < n; i++) {
    const int32_t x = md[pos(i)];
    acc += x*(double)x*qm[i]*PVQ_QM_SCALE_1*qm[i]*PVQ_QM_SCALE_1;
    const double r = x*qm[i]*PVQ_QM_SCALE_1;
    if (fabs(r) > maxr) { maxr = fabs(r); m = i; }
  }
  const double gr = sqrt(acc);
  auto rref = [&](int i) { return md[pos(i)]*qm[i]*PVQ_QM_SCALE_1; };
  const int s = rref(m) > 0 ? 1 : -1;
  auto rv = [&](int i) { double r = rref(i); if (i == m) r += gr*s; return r; };
  scale *= B.sin_theta;
  auto xv = [&](int i) { return i < m ? ypv(i)*scale : i == m ? -s*g*B.cos_theta : ypv(i - 1)*scale; };
  // od_apply_householder
  double l2r = 0, proj = 0;
  for (int i = 0; i < n; i++) { const double r = rv(i); l2r += r*r; }
  for (int i = 0; i < n; i++) proj += rv(i)*xv(i);
  const double proj_1 = proj*2./(1e-100 + l2r);
  for (int i = 0; i < n; i++) {
    const double x = xv(i) - rv(i)*proj_1;
    d[pos(i)] = (int32_t)floor(.5 + (x*(qm_inv[i]*PVQ_QM_INV_SCALE_1)));
  }
}

struct od_hip_dsynth {
  od_hip_ctx *ctx = nullptr;
  int16_t *d_qm[3][4] = {}, *d_qmi[3][4] = {};
  bool level_set[3][4] = {};
  char *h_g[3] = {nullptr, nullptr, nullptr};          // pinned mirrors of the planes' gain arenas (slot 0)
  long cap_blocks = 0, cap_bands = 0, cap_pulses = 0;
  DsBlock *h_blocks = nullptr, *d_blocks = nullptr;
  DsBand *h_bands = nullptr, *d_bands = nullptr;
  int16_t *h_pulses = nullptr, *d_pulses = nullptr;
};

extern "C" {

void od_hip_dsynth_destroy(od_hip_dsynth *s) {
  if (!s) return;
  if (s->ctx) {
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
  }
  for (int p = 0; p < 3; p++) {
    for (int l = 0; l < 4; l++) {
      if (s->d_qm[p][l]) (void)hipFree(s->d_qm[p][l]);
      if (s->d_qmi[p][l]) (void)hipFree(s->d_qmi[p][l]);
    }
    if (s->h_g[p]) (void)hipHostFree(s->h_g[p]);
  }
  void *dv[] = {s->d_blocks, s->d_bands, s->d_pulses};
  for (void *p : dv) if (p) (void)hipFree(p);
  void *hv[] = {s->h_blocks, s->h_bands, s->h_pulses};
  for (void *p : hv) if (p) (void)hipHostFree(p);
  delete s;
}

od_hip_dsynth *od_hip_dsynth_create(od_hip_ctx *ctx) {
  if (!ctx) { fail(OD_HIP_EFAULT, "null context"); return nullptr; }
  if (ctx->geo.nplanes != 3) { fail(OD_HIP_EINVAL, "decoder synthesis wants three planes"); return nullptr; }
  if (hipSetDevice(ctx->device) != hipSuccess) { fail(OD_HIP_ENODEV, "hipSetDevice failed"); return nullptr; }
  od_hip_dsynth *s = new od_hip_dsynth();
  s->ctx = ctx;
  // capacities from the geometry: at most one block per 4x4, one band per 15 coefficients
  long samples = 0;
  for (int p = 0; p < 3; p++) samples += (long)ctx->psz[p];
  s->cap_blocks = samples/16;
  s->cap_bands = samples/16;
  s->cap_pulses = 2*samples;                     // a band whose pulses do not fit 16 bits takes two entries per pulse
  bool ok = hipHostMalloc((void **)&s->h_blocks, s->cap_blocks*sizeof(DsBlock)) == hipSuccess;
  ok = ok && hipHostMalloc((void **)&s->h_bands, s->cap_bands*sizeof(DsBand)) == hipSuccess;
  ok = ok && hipHostMalloc((void **)&s->h_pulses, s->cap_pulses*sizeof(int16_t)) == hipSuccess;
  ok = ok && hipMalloc((void **)&s->d_blocks, s->cap_blocks*sizeof(DsBlock)) == hipSuccess;
  ok = ok && hipMalloc((void **)&s->d_bands, s->cap_bands*sizeof(DsBand)) == hipSuccess;
  ok = ok && hipMalloc((void **)&s->d_pulses, s->cap_pulses*sizeof(int16_t)) == hipSuccess;
  for (int p = 0; ok && p < 3; p++) {
    for (int l = 0; ok && l < ctx->nlev[p]; l++) {
      ok = ok && hipMalloc((void **)&s->d_qm[p][l], 1024*sizeof(int16_t)) == hipSuccess;
      ok = ok && hipMalloc((void **)&s->d_qmi[p][l], 1024*sizeof(int16_t)) == hipSuccess;
    }
  }
  if (!ok) {
    fail(OD_HIP_ENODEV, "decoder synthesis allocation failed");
    od_hip_dsynth_destroy(s);
    return nullptr;
  }
  return s;
}

int od_hip_dsynth_buffers(od_hip_dsynth *s, od_hip_dsynth_block **blocks, long *max_blocks,
                          od_hip_dsynth_band **bands, long *max_bands, int16_t **pulses, long *max_pulses) {
  if (!s || !blocks || !max_blocks || !bands || !max_bands || !pulses || !max_pulses)
    return fail(OD_HIP_EFAULT, "null pointer");
  static_assert(sizeof(DsBlock) == sizeof(od_hip_dsynth_block), "DsBlock mirrors od_hip_dsynth_block");
  static_assert(sizeof(DsBand) == sizeof(od_hip_dsynth_band), "DsBand mirrors od_hip_dsynth_band");
  *blocks = (od_hip_dsynth_block *)s->h_blocks; *max_blocks = s->cap_blocks;
  *bands = (od_hip_dsynth_band *)s->h_bands; *max_bands = s->cap_bands;
  *pulses = s->h_pulses; *max_pulses = s->cap_pulses;
  return 0;
}

int od_hip_dsynth_set_level(od_hip_dsynth *s, int pli, int level, const int16_t *qm, const int16_t *qm_inv) {
  if (!s || !qm || !qm_inv) return fail(OD_HIP_EFAULT, "null pointer");
  od_hip_ctx *ctx = s->ctx;
  if (pli < 0 || pli > 2 || level < 0 || level >= ctx->nlev[pli]) return fail(OD_HIP_EINVAL, "plane/level out of range");
  HIPCHK(hipSetDevice(ctx->device));
  const int n = (32 >> ctx->geo.xdec[pli]) >> level;
  HIPCHK(hipMemcpyAsync(s->d_qm[pli][level], qm, (size_t)n*n*sizeof(int16_t), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(s->d_qmi[pli][level], qm_inv, (size_t)n*n*sizeof(int16_t), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));       // the caller's tables may go away
  s->level_set[pli][level] = true;
  return 0;
}

// gr of every band of every block size of the pyramid in slot 0 -> pinned host; gr[pli][level]
// points at [band][block] (block = raster index at that level).
int od_hip_dsynth_ref_gains(od_hip_dsynth *s, const double *gr[3][4]) {
  if (!s || !gr) return fail(OD_HIP_EFAULT, "null pointer");
  od_hip_ctx *ctx = s->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  static const int32_t q1[11] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
  static const double b1[11] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
  for (int p = 0; p < 3; p++) {
    for (int l = 0; l < ctx->nlev[p]; l++) {
      if (!s->level_set[p][l]) return fail(OD_HIP_EINVAL, "od_hip_dsynth_set_level has not run for every level");
      PvqCall c;
      // the gain pass reads its own copy of the table (ctx->qm_slots): hand it the device copy
      if (int rc = pvq_prepare(ctx, 0, 1, p, l, nullptr, q1, b1, c)) return rc;
      const int n = c.n;
      HIPCHK(hipMemcpyAsync((void *)c.a.qm, s->d_qm[p][l], (size_t)n*n*sizeof(int16_t), hipMemcpyDeviceToDevice, ctx->stream));
      if (int rc = pvq_launch(ctx, c, 1, true)) return rc;
    }
    PvqArena &A = ctx->arena[p];
    if (!s->h_g[p]) HIPCHK(hipHostMalloc((void **)&s->h_g[p], A.g_slot));
    HIPCHK(hipMemcpyAsync(s->h_g[p], A.g, A.g_slot, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (int p = 0; p < 3; p++) {
    for (int l = 0; l < 4; l++) gr[p][l] = l < ctx->nlev[p] ? (const double *)(s->h_g[p] + ctx->arena[p].lev[l].o_g) : nullptr;
  }
  return 0;
}

int od_hip_dsynth_run(od_hip_dsynth *s, long nblocks, long nbands, long npulses) {
  if (!s) return fail(OD_HIP_EFAULT, "null pointer");
  od_hip_ctx *ctx = s->ctx;
  if (nblocks < 0 || nblocks > s->cap_blocks || nbands < 0 || nbands > s->cap_bands || npulses < 0
      || npulses > s->cap_pulses) return fail(OD_HIP_EINVAL, "record counts beyond the buffers");
  HIPCHK(hipSetDevice(ctx->device));
  // host-side validation of what the kernels index with: a corrupt record must not become a
  // device fault
  for (long b = 0; b < nblocks; b++) {
    const DsBlock &q = s->h_blocks[b];
    if (q.pli > 2 || q.bs > 3 || (q.pli > 0 && q.bs > 2)) return fail(OD_HIP_EINVAL, "bad synthesis block");
    const int n = 4 << q.bs, w = ctx->pw[q.pli], h = ctx->ph[q.pli];
    const long x = q.org%w, y = q.org/w;
    if (q.org < 0 || y + n > h || x + n > w || (x & (n - 1)) || (y & (n - 1))) return fail(OD_HIP_EINVAL, "synthesis block outside its plane");
  }
  static const int nb_of[4] = {1, 4, 7, 9};
  static const int off_all[] = {1, 16, 24, 32, 64, 96, 128, 256, 384, 512};
  for (long v = 0; v < nbands; v++) {
    const DsBand &B = s->h_bands[v];
    if (B.block >= (uint32_t)nblocks || (B.mode & 3) > DS_REF || (B.mode & ~7) || ((B.mode & DS_WIDE) && (B.mode & 3) == DS_ZERO))
      return fail(OD_HIP_EINVAL, "bad synthesis band");
    const DsBlock &q = s->h_blocks[B.block];
    if (B.band >= nb_of[q.bs]) return fail(OD_HIP_EINVAL, "band index beyond the block size");
    const int n = off_all[B.band + 1] - off_all[B.band];
    if ((B.mode & 3) != DS_ZERO && (long)B.yoff + (long)(n - ((B.mode & 3) == DS_REF))*((B.mode & DS_WIDE) ? 2 : 1) > npulses)
      return fail(OD_HIP_EINVAL, "pulses beyond the buffer");
  }
  if (nblocks == 0) return 0;
  HIPCHK(hipMemcpyAsync(s->d_blocks, s->h_blocks, nblocks*sizeof(DsBlock), hipMemcpyHostToDevice, ctx->stream));
  if (nbands) HIPCHK(hipMemcpyAsync(s->d_bands, s->h_bands, nbands*sizeof(DsBand), hipMemcpyHostToDevice, ctx->stream));
  if (npulses) HIPCHK(hipMemcpyAsync(s->d_pulses, s->h_pulses, npulses*sizeof(int16_t), hipMemcpyHostToDevice, ctx->stream));
  DsPlanes P;
  for (int p = 0; p < 3; p++) {
    for (int l = 0; l < 4; l++) {
      P.lev[p][l] = l < ctx->nlev[p] ? ctx->lev[p] + (size_t)l*ctx->psz[p] : nullptr;
      P.qm[p][l] = s->d_qm[p][l];
      P.qm_inv[p][l] = s->d_qmi[p][l];
    }
    P.d[p] = ctx->d[p];
    P.w[p] = ctx->pw[p];
  }
  for (int b = 0; b < 4; b++) P.tab[b] = ctx->tab[b];
  {
    Timed tm(ctx, "k_dsynth_blocks");
    hipLaunchKernelGGL(k_dsynth_blocks, dim3((unsigned)nblocks), dim3(64), 0, ctx->stream, P, s->d_blocks, (int)nblocks);
  }
  if (nbands) {
    Timed tm(ctx, "k_dsynth_bands");
    hipLaunchKernelGGL(k_dsynth_bands, dim3((unsigned)((nbands + 63)/64)), dim3(64), 0, ctx->stream, P, s->d_blocks,
                       s->d_bands, (int)nbands, s->d_pulses);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

}  // extern "C"
