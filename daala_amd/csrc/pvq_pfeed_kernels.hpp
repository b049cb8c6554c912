// P-frame feed: on an inter frame the reference of every band is the
// transform of the motion-compensated prediction (od_encode_compute_pred, src/encode.c:749-755:
// pred = md) - no H/V intra prediction, no CfL - so BOTH inputs of pvq_theta
// (src/pvq_encoder.c:311) are known for every block size of every plane before the first
// symbol is coded: x0 = the input frame's forward pyramid, r0 = the prediction's.  The device
// therefore enumerates pvq_theta's complete candidate list - the with-reference (gain, theta)
// candidates of :399-448 and the no-reference ones of :452-481 - for every band up front; the
// host prices them.  Two device passes around the host's libm stage (DESIGN.md section 5:
// pow / acos / sin are the host's, the device does + - * / sqrt floor):
//   k_pvq_pgains   g, gr (src/pvq.c:456-464), the correlation sum (:353-358), "reference is null"
//   host           cg, cgr = od_gain_compand(.), corr -> theta = acos(corr), sin(theta), which
//                  searches pvq_theta runs (:399, :452)
//   k_pvq_pcand    one lane group per (band, block, candidate slot): the slot's (i, j) by the
//                  reference's own loops, Householder reflection (src/pvq.c:364-413), the
//                  codeword search on n - 1 dimensions (pad = the dropped coefficient), or the
//                  no-reference search on n; cosine distance + 16-bit pulses out.
#pragma once
#include "pvq_kernels.hpp"

#define PFEED_NREF 12                 /* with-reference candidate slots (3 gains x 4 angles) */
#define PFEED_SLOTS (PFEED_NREF + 2)  /* + the two no-reference gains */
#define PFEED_TS_MAX 255              /* largest angular resolution the sin(qtheta) table covers */
#define PFEED_PI 3.14159265358979323846

struct PfeedArgs {
  PvqLevelArgs a;            // a.lev: input pyramid level of the frame's slot; geometry; bands
  const int32_t *pred;       // the prediction's pyramid level (same geometry)
  // pass 1 -> host, [band][block]
  double *g, *gr, *corr;
  int32_t *isnull;
  // host -> pass 2, [band][block]
  const double *cg, *cgr, *theta, *sinth;
  const int32_t *flags;      // bit 0: the theta search runs, bit 1: the no-reference search runs
  const double *sinq;        // sin(od_pvq_compute_theta(j, ts)) at ts*(ts - 1)/2 + j, ts <= PFEED_TS_MAX
  // pass 2 -> host
  double *cos_dist;          // [slot][band][block]
  int32_t *kout;             // [slot][band][block]: K of the slot, -1: slot not used
  int16_t *y;                // band b at 2... see below: [band: PFEED_SLOTS*nblk*yo][slot][block][ns]
  const double *rsq;
};

// od_pvq_compute_max_theta (src/pvq.c:476-482)
__device__ __forceinline__ int pfeed_max_theta(double qcg, double beta) {
  int ts = (int)floor(.5 + qcg*PFEED_PI/(2*beta));
  if (qcg < 1.4) ts = 1;
  return ts;
}

template <int N>
__global__ __launch_bounds__(64) void k_pvq_pgains(PfeedArgs pa) {
  const PvqLevelArgs &a = pa.a;
  constexpr int G = PvqGeom<N>::G, NL = PvqGeom<N>::NL, BPW = PvqGeom<N>::BPW;
  __shared__ int32_t Yst[BPW*(N + G + 2) + 1];
  __shared__ int32_t Org[64];
  __shared__ int32_t Pe[64];
  const int lane = threadIdx.x;
  const int g = lane%G, inst = lane/G;
  const long nblk = (long)a.nbx*a.nby;
  const long idx0 = (long)blockIdx.x*BPW;
  const int band = a.band_list[blockIdx.y];
  const int nslot_here = (int)(nblk - idx0 < BPW ? nblk - idx0 : BPW);
  int32_t x0[NL], r0[NL];
  int qi[NL];
  bool live;
  long blk;
  double unused;
  auto entry = [&]() { return idx0 + lane < nblk ? (int32_t)(2*(idx0 + lane)) : -1; };
  pvq_gather<N, decltype(entry), true>(a, 0, a.off[band], lane, g, inst, nslot_here, entry, Pe, Yst, Org, live,
                                       blk, nullptr, unused, x0, qi, pa.pred, r0);
  // od_pvq_compute_gain of x0 and of r0: five sequential multiplies per term (src/pvq.c:460-463)
  const double accx = pvq_chain_sum<N>(g, lane, [&](int j) {
    return x0[j]*(double)x0[j]*qi[j]*PVQ_QM_SCALE_1*qi[j]*PVQ_QM_SCALE_1;
  });
  const double accr = pvq_chain_sum<N>(g, lane, [&](int j) {
    return r0[j]*(double)r0[j]*qi[j]*PVQ_QM_SCALE_1*qi[j]*PVQ_QM_SCALE_1;
  });
  // corr += x[i]*r[i], x = x0*qm*scale with the reference's int*int product first (:355-357)
  const double corr = pvq_chain_sum<N>(g, lane, [&](int j) {
    return ((x0[j]*qi[j])*PVQ_QM_SCALE_1)*((r0[j]*qi[j])*PVQ_QM_SCALE_1);
  });
  int nz = 0;
#pragma unroll
  for (int j = 0; j < NL; j++) nz |= r0[j] != 0;
#pragma unroll
  for (int o = 1; o < G; o <<= 1) nz |= __shfl_xor(nz, o, 64);
  if (live && g == 0) {
    const size_t r = (size_t)band*nblk + blk;
    pa.g[r] = sqrt(accx);
    pa.gr[r] = sqrt(accr);
    pa.corr[r] = corr;
    pa.isnull[r] = !nz;
  }
}

template <int N>
__global__ __launch_bounds__(64, PVQ_V4_WAVES(N)) void k_pvq_pcand(PfeedArgs pa) {
  const PvqLevelArgs &a = pa.a;
  constexpr int G = PvqGeom<N>::G, NL = PvqGeom<N>::NL, BPW = PvqGeom<N>::BPW;
  __shared__ int32_t Yst[BPW*(N + G + 2) + 1];
  __shared__ int32_t Org[64];
  __shared__ int32_t Pe[64];
  __shared__ double RsqL[PVQ_RSQ_L(N)];
  __shared__ double Xs[BPW*(N + 1)];               // the reflected vector of every lane slot (compaction)
  const int lane = threadIdx.x;
  const int g = lane%G, inst = lane/G;
  const long nblk = (long)a.nbx*a.nby;
  const int band = a.band_list[blockIdx.y];
  const int o0 = a.off[band];
  const double beta = a.beta[band];
  // entries: candidate slot major, blocks in raster order inside a slot
  const long idx0 = (long)blockIdx.x*BPW, idx_end = (long)PFEED_SLOTS*nblk;
  const int nslot_here = (int)(idx_end - idx0 < BPW ? idx_end - idx0 : BPW);
  constexpr int NQ = PVQ_RSQ_L(N)/64;
  double rq[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) rq[q] = pa.rsq[lane + 64*q];
  int32_t x0[NL], r0[NL];
  int qi[NL];
  bool live;
  long blk;
  double unused;
  // the gather wants entries as 2*block + flag; the candidate slot is recovered from the position
  auto entry = [&]() {
    const long i = idx0 + lane;
    return (lane < BPW && i < idx_end) ? (int32_t)(2*(i%nblk)) : -1;
  };
  pvq_gather<N, decltype(entry), true>(a, 0, o0, lane, g, inst, nslot_here, entry, Pe, Yst, Org, live, blk,
                                       nullptr, unused, x0, qi, pa.pred, r0);
#pragma unroll
  for (int q = 0; q < NQ; q++) RsqL[lane + 64*q] = rq[q];
  pvq_wave_lds_sync();
  const int c = (int)((idx0 + inst)/nblk);         // candidate slot of this lane group
  const size_t rec = (size_t)band*nblk + blk;
  const size_t nrec = (size_t)a.nbands*nblk;
  const double cg = pa.cg[rec], cgr = pa.cgr[rec], theta = pa.theta[rec], sinth = pa.sinth[rec];
  const double gr = pa.gr[rec];
  const int flags = live ? pa.flags[rec] : 0;
  // ---- which candidate is slot c (the reference's loops, src/pvq_encoder.c:406-417, :457)
  bool has = false, withref = c < PFEED_NREF;
  int k = 0;
  double g2 = 0;
  if (withref) {
    if (flags & 1) {
      const int icgr = (int)floor(.5 + cgr);
      const double gain_offset = cgr - icgr;
      int i_lo = (int)floor(cg - gain_offset) - 1;
      if (i_lo < 1) i_lo = 1;
      const int i_hi = (int)ceil(cg - gain_offset);
      int slot = 0;
      for (int i = i_lo; i <= i_hi && !has && slot <= c; i++) {
        const double qcg = i + gain_offset;
        const int ts = pfeed_max_theta(qcg, beta);
        int j_lo = (int)floor(.5 + theta*2/PFEED_PI*ts) - 2;
        if (j_lo < 0) j_lo = 0;
        int j_hi = (int)ceil(theta*2/PFEED_PI*ts);
        if (j_hi > ts - 1) j_hi = ts - 1;
        const int cnt = j_hi >= j_lo ? j_hi - j_lo + 1 : 0;
        if (c < slot + cnt) {
          const int j = j_lo + (c - slot);
          has = ts <= PFEED_TS_MAX;
          // od_pvq_compute_k, with-reference form without reference-dependent terms
          // (OD_ROBUST_STREAM, src/pvq.c:526)
          if (j == 0) k = 0;
          else {
            k = (int)floor(.5 + (j - .2)*sqrt((double)((N + 2)/2)));
            if (k < 1) k = 1;
          }
          if (has) g2 = qcg*cg*sinth*pa.sinq[ts*(ts - 1)/2 + j];
        }
        slot += cnt;
      }
    }
  }
  else if (flags & 2) {
    int i0 = (int)floor(cg);
    if (i0 < 1) i0 = 1;
    const int gi = i0 + (c - PFEED_NREF);
    if (gi <= ceil(cg)) {
      const double qcg = gi;
      k = pvq_k_noref(qcg, N, beta);
      has = k <= PVQ_K_MAX16;                      // 16-bit pulses: larger K stays on the host (slot unused)
      g2 = qcg*cg;
    }
  }
  if (!has) k = 0;
  // ---- the vector that is searched
  double x[NL];
#pragma unroll
  for (int j = 0; j < NL; j++) x[j] = (x0[j]*qi[j])*PVQ_QM_SCALE_1;
  if (__any(has && withref)) {
    // od_compute_householder (src/pvq.c:364-387): m = first index of the largest |r|, r[m] += gr*s
    double r[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) r[j] = (r0[j]*qi[j])*PVQ_QM_SCALE_1;
    double mx = fabs(r[0]);
    int mpos = g*NL;
#pragma unroll
    for (int j = 1; j < NL; j++) {
      if (fabs(r[j]) > mx) { mx = fabs(r[j]); mpos = g*NL + j; }
    }
    // the reference starts from maxr = 0 and takes strictly greater: index 0 unless something is > 0,
    // which the first-index maximum reproduces (all-zero r never reaches here: isnull)
#pragma unroll
    for (int o = 1; o < G; o <<= 1) {
      const double omx = __shfl_xor(mx, o, 64);
      const int opos = __shfl_xor(mpos, o, 64);
      if (omx > mx || (omx == mx && opos < mpos)) { mx = omx; mpos = opos; }
    }
    const int m = mpos;
    double rm = 0;
#pragma unroll
    for (int j = 0; j < NL; j++) rm = (g*NL + j == m) ? r[j] : rm;
    rm = __shfl(rm, (lane/G)*G + m/NL, 64);
    const int s = rm > 0 ? 1 : -1;
#pragma unroll
    for (int j = 0; j < NL; j++) r[j] = (g*NL + j == m) ? r[j] + gr*s : r[j];
    // od_apply_householder (:395-413)
    const double l2r = pvq_chain_sum<N>(g, lane, [&](int j) { return r[j]*r[j]; });
    const double proj = pvq_chain_sum<N>(g, lane, [&](int j) { return r[j]*x[j]; });
    const double proj_1 = proj*2./(1e-100 + l2r);
    // x[i] -= r[i]*proj_1, then drop element m (:404): through LDS, the pad goes last
#pragma unroll
    for (int j = 0; j < NL; j++) Xs[inst*(N + 1) + g*NL + j] = x[j] - r[j]*proj_1;
    pvq_wave_lds_sync();
    if (has && withref) {
#pragma unroll
      for (int j = 0; j < NL; j++) {
        const int idx = g*NL + j;
        x[j] = idx < N - 1 ? Xs[inst*(N + 1) + idx + (idx >= m)] : 0.;
      }
    }
    pvq_wave_lds_sync();
  }
  PvqVec<N> v;
  v.neg = 0;
#pragma unroll
  for (int j = 0; j < NL; j++) {
    v.x[j] = fabs(x[j]);
    v.neg |= (uint32_t)(x[j] < 0) << j;
  }
  pvq_vec_finish<N>(v, g, lane);
  int y[NL];
  int npg = 0, npr = 0;
  const double cd = pvq_search_v4<N, true>(v, g, lane, k, g2, pa.rsq, RsqL, y, npg, npr, has && withref);
  if (live && g == 0) {
    pa.cos_dist[(size_t)c*nrec + rec] = has ? cd : 0;
    pa.kout[(size_t)c*nrec + rec] = has ? k : -1;
  }
  constexpr int NS = (N + 1) & ~1, NW = NS/2;
  int16_t *Y16 = reinterpret_cast<int16_t *>(Yst);
#pragma unroll
  for (int j = 0; j < NL; j++) Y16[inst*NS + g*NL + j] = (int16_t)(has ? (((v.neg >> j) & 1) ? -y[j] : y[j]) : 0);
  if (NS != N && g == G - 1) Y16[inst*NS + N] = 0;
  pvq_wave_lds_sync();
  {
    const int yo_band = o0 == 1 ? 0 : o0;
    uint32_t *yo = reinterpret_cast<uint32_t *>(pa.y + (size_t)PFEED_SLOTS*nblk*yo_band);
    const uint32_t *Yw = reinterpret_cast<const uint32_t *>(Yst);
    const int lim = nslot_here*NW;
#pragma unroll
    for (int it = 0; it < (BPW*NW + 63)/64; it++) {
      const int e = lane + 64*it;
      const int b = e/NW;
      if (e < lim && Pe[b] >= 0) {
        const long pos = idx0 + b;                 // = slot*nblk + block
        yo[(size_t)pos*NW + (e - b*NW)] = Yw[e];
      }
    }
  }
}
