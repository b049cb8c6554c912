// P-frame feed, host side of the C-ABI (include/daala_hip.h section 4d; kernels in
// pvq_pfeed_kernels.hpp).  One object = one inter frame in flight: an own two-slot context
// (input frame, motion-compensated prediction), the records of every band of every level of
// every plane in three arenas (pass 1 -> host, host -> pass 2, pass 2 -> host), pinned mirrors.
// Included at the end of daala_hip.hip.
#pragma once

struct PfeedLevelLayout {
  int n = 0, bs = 0, nb = 0, nblk = 0, nbx = 0, ncoded = 0, off[11] = {};
  size_t nrec = 0;
  size_t o_g = 0, o_gr = 0, o_corr = 0, o_null = 0;               // arena 1
  size_t o_cg = 0, o_cgr = 0, o_theta = 0, o_sinth = 0, o_flags = 0;   // arena A
  size_t o_cd = 0, o_k = 0, o_y = 0;                              // arena 2
  bool set = false;
  std::vector<int16_t> qm;
  int32_t q[11] = {};
  double beta[11] = {};
};

struct od_hip_pfeed {
  od_hip_ctx *ctx = nullptr;          // own context: slot 0 = input frame, slot 1 = prediction
  int nplanes = 0, nlev[3] = {0, 0, 0};
  PfeedLevelLayout L[3][4];
  size_t b1 = 0, bA = 0, b2 = 0;      // arena bytes
  char *d1 = nullptr, *dA = nullptr, *d2 = nullptr;     // device
  char *h1 = nullptr, *hA = nullptr, *h2 = nullptr;     // pinned host
  double *d_sinq = nullptr;
  bool gains_ready = false, results_ready = false;
};

extern "C" {

void od_hip_pfeed_destroy(od_hip_pfeed *f) {
  if (!f) return;
  if (f->ctx) {
    (void)hipSetDevice(f->ctx->device);
    (void)hipStreamSynchronize(f->ctx->stream);
  }
  void *dv[] = {f->d1, f->dA, f->d2, f->d_sinq};
  for (void *p : dv) if (p) (void)hipFree(p);
  void *hv[] = {f->h1, f->hA, f->h2};
  for (void *p : hv) if (p) (void)hipHostFree(p);
  if (f->ctx) od_hip_ctx_destroy(f->ctx);
  delete f;
}

od_hip_pfeed *od_hip_pfeed_create(int device, const od_hip_geometry *geo) {
  if (!geo) { fail(OD_HIP_EFAULT, "null geometry"); return nullptr; }
  od_hip_geometry g2 = *geo;
  g2.nslots = 2;
  od_hip_ctx *ctx = od_hip_ctx_create(device, &g2);
  if (!ctx) return nullptr;
  od_hip_pfeed *f = new od_hip_pfeed();
  f->ctx = ctx;
  f->nplanes = ctx->geo.nplanes;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t b1 = 0, bA = 0, b2 = 0;
  for (int p = 0; p < f->nplanes; p++) {
    f->nlev[p] = ctx->nlev[p];
    for (int l = 0; l < ctx->nlev[p]; l++) {
      PfeedLevelLayout &L = f->L[p][l];
      L.n = (32 >> ctx->geo.xdec[p]) >> l;
      L.bs = L.n == 4 ? 0 : L.n == 8 ? 1 : L.n == 16 ? 2 : 3;
      L.nb = od_hip_band_offsets(L.bs, L.off);
      L.nbx = ctx->pw[p]/L.n;
      L.nblk = L.nbx*(ctx->ph[p]/L.n);
      L.ncoded = L.n*L.n < 512 ? L.n*L.n : 512;
      L.nrec = (size_t)L.nb*L.nblk;
      L.o_g = b1; b1 = al(b1 + L.nrec*8);
      L.o_gr = b1; b1 = al(b1 + L.nrec*8);
      L.o_corr = b1; b1 = al(b1 + L.nrec*8);
      L.o_null = b1; b1 = al(b1 + L.nrec*4);
      L.o_cg = bA; bA = al(bA + L.nrec*8);
      L.o_cgr = bA; bA = al(bA + L.nrec*8);
      L.o_theta = bA; bA = al(bA + L.nrec*8);
      L.o_sinth = bA; bA = al(bA + L.nrec*8);
      L.o_flags = bA; bA = al(bA + L.nrec*4);
      L.o_cd = b2; b2 = al(b2 + (size_t)PFEED_SLOTS*L.nrec*8);
      L.o_k = b2; b2 = al(b2 + (size_t)PFEED_SLOTS*L.nrec*4);
      L.o_y = b2; b2 = al(b2 + (size_t)PFEED_SLOTS*L.nblk*L.ncoded*2);
    }
  }
  f->b1 = b1; f->bA = bA; f->b2 = b2;
  bool ok = hipMalloc((void **)&f->d1, b1) == hipSuccess && hipMalloc((void **)&f->dA, bA) == hipSuccess
    && hipMalloc((void **)&f->d2, b2) == hipSuccess
    && hipHostMalloc((void **)&f->h1, b1) == hipSuccess && hipHostMalloc((void **)&f->hA, bA) == hipSuccess
    && hipHostMalloc((void **)&f->h2, b2) == hipSuccess;
  // sin(od_pvq_compute_theta(j, ts)) (src/pvq.c:490-493), this process's libm
  std::vector<double> sq((size_t)(PFEED_TS_MAX + 1)*PFEED_TS_MAX/2 + PFEED_TS_MAX + 1, 0.);
  for (int ts = 1; ts <= PFEED_TS_MAX; ts++) {
    for (int j = 0; j < ts; j++) {
      const double qtheta = (j < ts - 1 ? j : ts - 1)*.5*M_PI/ts;
      sq[(size_t)ts*(ts - 1)/2 + j] = sin(qtheta);
    }
  }
  ok = ok && hipMalloc((void **)&f->d_sinq, sq.size()*8) == hipSuccess
    && hipMemcpy(f->d_sinq, sq.data(), sq.size()*8, hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) {
    fail(OD_HIP_ENODEV, "P-frame feed allocation failed");
    od_hip_pfeed_destroy(f);
    return nullptr;
  }
  return f;
}

int od_hip_pfeed_set_level(od_hip_pfeed *f, int pli, int level, const int16_t *qm, const int32_t *q,
                           const double *beta) {
  if (!f || !qm || !q || !beta) return fail(OD_HIP_EFAULT, "null pointer");
  if (pli < 0 || pli >= f->nplanes || level < 0 || level >= f->nlev[pli]) return fail(OD_HIP_EINVAL, "plane/level out of range");
  PfeedLevelLayout &L = f->L[pli][level];
  L.qm.assign(qm, qm + L.n*L.n);
  for (int b = 0; b < L.nb; b++) { L.q[b] = q[b]; L.beta[b] = beta[b]; }
  L.set = true;
  return 0;
}

}  // extern "C"

namespace {
template <int N>
void launch_pfeed(const PfeedArgs &pa, int nlist, long nblk, hipStream_t s, bool cand) {
  constexpr int BPW = PvqGeom<N>::BPW;
  if (!cand) {
    dim3 grid((unsigned)((nblk + BPW - 1)/BPW), nlist, 1);
    hipLaunchKernelGGL((k_pvq_pgains<N>), grid, dim3(64), 0, s, pa);
  }
  else {
    dim3 grid((unsigned)(((long)PFEED_SLOTS*nblk + BPW - 1)/BPW), nlist, 1);
    hipLaunchKernelGGL((k_pvq_pcand<N>), grid, dim3(64), 0, s, pa);
  }
}

int pfeed_pass(od_hip_pfeed *f, bool cand) {
  od_hip_ctx *ctx = f->ctx;
  for (int p = 0; p < f->nplanes; p++) {
    for (int l = 0; l < f->nlev[p]; l++) {
      PfeedLevelLayout &L = f->L[p][l];
      if (!L.set) return fail(OD_HIP_EINVAL, "P-frame feed level parameters not set");
      PfeedArgs pa;
      memset(&pa, 0, sizeof(pa));
      PvqLevelArgs &a = pa.a;
      a.nbands = L.nb;
      for (int i = 0; i <= L.nb; i++) a.off[i] = L.off[i];
      for (int b = 0; b < L.nb; b++) { a.q[b] = L.q[b]; a.beta[b] = L.beta[b]; }
      int16_t *qm_d = ctx->qm_slots + ((size_t)p*4 + l)*1024;
      if (!cand) HIPCHK(hipMemcpyAsync(qm_d, L.qm.data(), (size_t)L.n*L.n*2, hipMemcpyHostToDevice, ctx->stream));
      a.lev = ctx->lev[p] + ((size_t)0*ctx->nlev[p] + l)*ctx->psz[p];
      pa.pred = ctx->lev[p] + ((size_t)1*ctx->nlev[p] + l)*ctx->psz[p];
      a.lev_fstride = 0;
      a.w = ctx->pw[p];
      a.n = L.n;
      a.nbx = L.nbx;
      a.nby = L.nblk/L.nbx;
      a.tab = ctx->tab[L.bs];
      a.qm = qm_d;
      a.blk_first = 0;
      a.blk_end = L.nblk;
      pa.g = (double *)(f->d1 + L.o_g); pa.gr = (double *)(f->d1 + L.o_gr);
      pa.corr = (double *)(f->d1 + L.o_corr); pa.isnull = (int32_t *)(f->d1 + L.o_null);
      pa.cg = (const double *)(f->dA + L.o_cg); pa.cgr = (const double *)(f->dA + L.o_cgr);
      pa.theta = (const double *)(f->dA + L.o_theta); pa.sinth = (const double *)(f->dA + L.o_sinth);
      pa.flags = (const int32_t *)(f->dA + L.o_flags);
      pa.sinq = f->d_sinq;
      pa.cos_dist = (double *)(f->d2 + L.o_cd); pa.kout = (int32_t *)(f->d2 + L.o_k);
      pa.y = (int16_t *)(f->d2 + L.o_y);
      pa.rsq = ctx->rsq;
      static const int sizes[4] = {15, 8, 32, 128};
      for (int si = 0; si < 4; si++) {
        int nlist = 0;
        for (int b = 0; b < L.nb; b++) if (L.off[b + 1] - L.off[b] == sizes[si]) a.band_list[nlist++] = b;
        if (!nlist) continue;
        switch (sizes[si]) {
          case 15: launch_pfeed<15>(pa, nlist, L.nblk, ctx->stream, cand); break;
          case 8: launch_pfeed<8>(pa, nlist, L.nblk, ctx->stream, cand); break;
          case 32: launch_pfeed<32>(pa, nlist, L.nblk, ctx->stream, cand); break;
          default: launch_pfeed<128>(pa, nlist, L.nblk, ctx->stream, cand); break;
        }
        HIPCHK(hipGetLastError());
      }
    }
  }
  return 0;
}
}  // namespace

extern "C" {

// Pass 1.  planes_in / planes_pred: the padded 8-bit input planes of the frame and its
// motion-compensated prediction (od_state_mc_predict's output), as od_hip_upload_planes.
int od_hip_pfeed_gains(od_hip_pfeed *f, const unsigned char *const planes_in[], const int stride_in[],
                       const unsigned char *const planes_pred[], const int stride_pred[]) {
  if (!f || !planes_in || !stride_in || !planes_pred || !stride_pred) return fail(OD_HIP_EFAULT, "null pointer");
  od_hip_ctx *ctx = f->ctx;
  f->gains_ready = f->results_ready = false;
  if (int rc = od_hip_upload_planes(ctx, 0, planes_in, stride_in)) return rc;
  if (int rc = od_hip_upload_planes(ctx, 1, planes_pred, stride_pred)) return rc;
  if (int rc = od_hip_forward_pyramid(ctx, 0, 2)) return rc;
  if (int rc = pfeed_pass(f, false)) return rc;
  HIPCHK(hipMemcpyAsync(f->h1, f->d1, f->b1, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  f->gains_ready = true;
  return 0;
}

// The host's libm stage for records [rec0, rec1) of one (plane, level): any thread, disjoint
// ranges concurrently.  cg, cgr = od_gain_compand (src/pvq.c:422), the normalised and clamped
// correlation (src/pvq_encoder.c:380-381; it replaces the raw sum in the view), theta = acos,
// sin(theta), and which searches pvq_theta runs (:399, :452).
int od_hip_pfeed_host_stage(od_hip_pfeed *f, int pli, int level, long rec0, long rec1) {
  if (!f) return fail(OD_HIP_EFAULT, "null feed");
  if (pli < 0 || pli >= f->nplanes || level < 0 || level >= f->nlev[pli]) return fail(OD_HIP_EINVAL, "plane/level out of range");
  if (!f->gains_ready) return fail(OD_HIP_EINVAL, "no gains pass before the host stage");
  const PfeedLevelLayout &L = f->L[pli][level];
  if (rec0 < 0) rec0 = 0;
  if (rec1 > (long)L.nrec) rec1 = (long)L.nrec;
  const double *g = (const double *)(f->h1 + L.o_g), *gr = (const double *)(f->h1 + L.o_gr);
  double *corr = (double *)(f->h1 + L.o_corr);
  const int32_t *isnull = (const int32_t *)(f->h1 + L.o_null);
  double *cg = (double *)(f->hA + L.o_cg), *cgr = (double *)(f->hA + L.o_cgr);
  double *theta = (double *)(f->hA + L.o_theta), *sinth = (double *)(f->hA + L.o_sinth);
  int32_t *flags = (int32_t *)(f->hA + L.o_flags);
  for (long r = rec0; r < rec1; r++) {
    const int b = (int)(r/L.nblk);
    const int q0 = L.q[b];
    const double beta = L.beta[b];
    cg[r] = host_gain_compand(g[r], q0, beta);
    cgr[r] = host_gain_compand(gr[r], q0, beta);
    double c = corr[r]/(1e-100 + g[r]*gr[r]);
    c = c < 1. ? c : 1.;
    c = c > -1. ? c : -1.;
    corr[r] = c;
    int fl = 0;
    double th = 0, st = 0;
    if (!isnull[r] && c > 0) {
      fl |= 1;
      th = acos(c);
      st = sin(th);
    }
    if (c < .5 || cg[r] < 2.) fl |= 2;
    theta[r] = th;
    sinth[r] = st;
    flags[r] = fl;
  }
  return 0;
}

// Pass 2: the host stage's results up, every candidate's search, cosine distances + pulses down.
int od_hip_pfeed_search(od_hip_pfeed *f) {
  if (!f) return fail(OD_HIP_EFAULT, "null feed");
  if (!f->gains_ready) return fail(OD_HIP_EINVAL, "no gains pass before the search");
  od_hip_ctx *ctx = f->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemcpyAsync(f->dA, f->hA, f->bA, hipMemcpyHostToDevice, ctx->stream));
  if (int rc = pfeed_pass(f, true)) return rc;
  HIPCHK(hipMemcpyAsync(f->h2, f->d2, f->b2, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  f->results_ready = true;
  return 0;
}

int od_hip_pfeed_nrec(od_hip_pfeed *f, int pli, int level) {
  if (!f || pli < 0 || pli >= f->nplanes || level < 0 || level >= f->nlev[pli]) return fail(OD_HIP_EINVAL, "plane/level out of range");
  return (int)f->L[pli][level].nrec;
}

int od_hip_pfeed_view(od_hip_pfeed *f, int pli, int level, od_hip_pfeed_level *v) {
  if (!f || !v) return fail(OD_HIP_EFAULT, "null pointer");
  if (pli < 0 || pli >= f->nplanes || level < 0 || level >= f->nlev[pli]) return fail(OD_HIP_EINVAL, "plane/level out of range");
  if (!f->results_ready) return fail(OD_HIP_EINVAL, "no search results");
  const PfeedLevelLayout &L = f->L[pli][level];
  memset(v, 0, sizeof(*v));
  v->n = L.n; v->nbands = L.nb; v->nblk = L.nblk; v->nbx = L.nbx;
  for (int i = 0; i <= L.nb; i++) v->off[i] = L.off[i];
  v->nslots = PFEED_SLOTS;
  v->nref_slots = PFEED_NREF;
  v->g = (const double *)(f->h1 + L.o_g); v->gr = (const double *)(f->h1 + L.o_gr);
  v->corr = (const double *)(f->h1 + L.o_corr); v->isnull = (const int32_t *)(f->h1 + L.o_null);
  v->cg = (const double *)(f->hA + L.o_cg); v->cgr = (const double *)(f->hA + L.o_cgr);
  v->theta = (const double *)(f->hA + L.o_theta); v->flags = (const int32_t *)(f->hA + L.o_flags);
  v->cos_dist = (const double *)(f->h2 + L.o_cd); v->k = (const int32_t *)(f->h2 + L.o_k);
  v->y = (const int16_t *)(f->h2 + L.o_y);
  return 0;
}

}  // extern "C"
