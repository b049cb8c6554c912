// PVQ device code (gfx950).  All arithmetic is IEEE double evaluated in the
// reference's order; the translation unit is built with -ffp-contract=off so no
// multiply-add is fused (the reference build, gcc -O2 without -march, has none).
//
// Parallelisation (round 1, correctness first): one lane owns one band vector and
// runs the reference's sequential scans literally, so every tie-break ("first
// index wins", strict >) is reproduced by construction.  Lanes of a wave work on
// the same band of neighbouring blocks, hence the same n; only the pulse count
// diverges.  See DESIGN.md section 3.4 for the planned wave-per-vector variant.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PVQ_MAXN 128                       /* OD_MAX_PVQ_SIZE, src/pvq.h:55 */
#define PVQ_QM_SCALE_1 (1./32767)          /* OD_QM_SCALE_1, src/pvq.h:57-59 */
#define PVQ_QM_INV_SCALE_1 (1./4096)       /* OD_QM_INV_SCALE_1 */
#define PVQ_COMPAND_SCALE 4096.            /* OD_COMPAND_SCALE, src/pvq.h:68 */
#define PVQ_LAMBDA .147                    /* OD_PVQ_LAMBDA, src/pvq.h:49 */

// 1/sqrt(i): the reference's 6-digit literal table for i <= 16, exact above
// (src/pvq_encoder.c:83-91).
__device__ __forceinline__ double pvq_rsqrt_small(int i) {
  switch (i) {
    case 1: return 1.000000; case 2: return 0.707107; case 3: return 0.577350;
    case 4: return 0.500000; case 5: return 0.447214; case 6: return 0.408248;
    case 7: return 0.377964; case 8: return 0.353553; case 9: return 0.333333;
    case 10: return 0.316228; case 11: return 0.301511; case 12: return 0.288675;
    case 13: return 0.277350; case 14: return 0.267261; case 15: return 0.258199;
    case 16: return 0.250000;
    default: return 1./sqrt((double)i);
  }
}

// pvq_search_rdo_double (src/pvq_encoder.c:121-225).  xc: input vector (signed),
// x: caller scratch for |xc|, yp: pulses out.  Returns the cosine distance.
__device__ inline double pvq_search_dev(const double *xc, double *x, int n, int k,
                                        int32_t *yp, double g2) {
  double xx = 0, xy = 0, yy = 0;
  int i = 0;
  for (int j = 0; j < n; j++) {
    x[j] = fabs(xc[j]);
    xx += x[j]*x[j];
  }
  const double norm_1 = 1./sqrt(1e-30 + xx);
  const double lambda = PVQ_LAMBDA/(1e-30 + g2);
  if (k > 2) {
    double l1 = 0;
    for (int j = 0; j < n; j++) l1 += x[j];
    const double l1_inv = 1./(l1 > 1e-100 ? l1 : 1e-100);
    for (int j = 0; j < n; j++) {
      int p = (int)floor(k*x[j]*l1_inv);
      p = p > 0 ? p : 0;
      yp[j] = p;
      xy += x[j]*p;
      yy += p*p;
      i += p;
    }
  }
  else {
    for (int j = 0; j < n; j++) yp[j] = 0;
  }
  const int rdo_pulses = 1 + k/4;
  const double delta_rate = 3./n;
  for (; i < k - rdo_pulses; i++) {
    int pos = 0;
    double best_xy = -10, best_yy = 1;
    for (int j = 0; j < n; j++) {
      double txy = xy + x[j];
      double tyy = yy + 2*yp[j] + 1;
      txy *= txy;
      if (j == 0 || txy*best_yy > best_xy*tyy) {
        best_xy = txy;
        best_yy = tyy;
        pos = j;
      }
    }
    xy = xy + x[pos];
    yy = yy + 2*yp[pos] + 1;
    yp[pos]++;
  }
  for (; i < k; i++) {
    int pos = 0;
    double best_cost = -1e5;
    for (int j = 0; j < n; j++) {
      double txy = xy + x[j];
      double rs = pvq_rsqrt_small((int)(yy + 2*yp[j] + 1));
      txy = 2*txy*norm_1*rs - lambda*j*delta_rate;
      if (j == 0 || txy > best_cost) {
        best_cost = txy;
        pos = j;
      }
    }
    xy = xy + x[pos];
    yy = yy + 2*yp[pos] + 1;
    yp[pos]++;
  }
  for (int j = 0; j < n; j++) if (xc[j] < 0) yp[j] = -yp[j];
  return xy/(1e-100 + sqrt(xx*yy));
}

// od_pvq_compute_k, no-reference form (src/pvq.c:508-514).
__device__ __forceinline__ int pvq_k_noref(double qcg, int n, double beta) {
  if (qcg == 0) return 0;
  if (n == 15 && qcg == 1 && beta > 1.25) return 1;
  int k = (int)floor(.5 + (qcg - .2)*sqrt((double)((n + 3)/2))/beta);
  return k > 1 ? k : 1;
}

// od_gain_compand (src/pvq.c:422-425).  beta != 1 goes through the device pow():
// value parity with glibc pow is NOT pinned.
__device__ __forceinline__ double pvq_gain_compand(double g, int q0, double beta) {
  if (beta == 1) return g/q0;
  return PVQ_COMPAND_SCALE*pow(g*(1./PVQ_COMPAND_SCALE), 1./beta)/q0;
}

__global__ void k_pvq_search_vectors(int n, int nvec, const double *__restrict__ x,
                                     const int32_t *__restrict__ k,
                                     const double *__restrict__ g2,
                                     int32_t *__restrict__ y, double *__restrict__ cos_dist) {
  long v = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (v >= nvec) return;
  double xc[PVQ_MAXN], xa[PVQ_MAXN];
  int32_t yp[PVQ_MAXN];
  for (int j = 0; j < n; j++) xc[j] = x[v*n + j];
  cos_dist[v] = pvq_search_dev(xc, xa, n, k[v], yp, g2[v]);
  for (int j = 0; j < n; j++) y[v*n + j] = yp[j];
}

// od_pvq_synthesis_partial, noref branch (src/pvq.c:552-572).
__global__ void k_pvq_synthesis_noref(int n, int nvec, const int32_t *__restrict__ y,
                                      const double *__restrict__ g,
                                      const int16_t *__restrict__ qm_inv,
                                      int32_t *__restrict__ out) {
  long v = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (v >= nvec) return;
  int yy = 0;
  for (int i = 0; i < n; i++) yy += y[v*n + i]*y[v*n + i];
  double scale = yy == 0 ? 0 : g[v]/sqrt((double)yy);
  for (int i = 0; i < n; i++) {
    out[v*n + i] = (int32_t)floor(.5 + (y[v*n + i]*scale)*(qm_inv[v*n + i]*PVQ_QM_INV_SCALE_1));
  }
}

struct PvqBandRec {      // mirrors od_hip_pvq_band (include/daala_hip.h)
  double cg, g;
  double cos_dist[2];
  double dist[2];
  int32_t qg[2];
  int32_t k[2];
  int32_t ncand;
  int32_t pad;
};

struct PvqLevelArgs {
  const int32_t *lev;      // level plane, frame 0
  size_t lev_fstride;      // elements between frames
  int w;                   // plane stride
  int n;                   // block size
  int nbx, nby;            // blocks per row / column
  int nbands;
  int off[11];             // band boundaries (coding order)
  int q[10];
  double beta[10];
  const uint16_t *tab;     // coding index -> raster offset (y*n + x)
  const int16_t *qm;       // n*n, coding order
  PvqBandRec *bands;       // [frame][block][band]
  int32_t *y;              // [frame][block][cand][ncoded]
  int ncoded;
};

// No-reference candidates of every (block, band) of one pyramid level: the
// state-free part of pvq_theta (src/pvq_encoder.c:352-357, :452-481).
// Thread layout: x = block within frame (fastest), y = band, z = frame, so a
// wave holds the same band of 64 neighbouring blocks (uniform n).
__global__ __launch_bounds__(64) void k_pvq_noref_level(PvqLevelArgs a) {
  const long blk = (long)blockIdx.x*blockDim.x + threadIdx.x;
  const int band = blockIdx.y, f = blockIdx.z;
  const long nblk = (long)a.nbx*a.nby;
  if (blk >= nblk) return;
  const int bx = blk%a.nbx, by = blk/a.nbx;
  const int o0 = a.off[band], nn = a.off[band + 1] - o0;
  const int32_t *src = a.lev + (size_t)f*a.lev_fstride + (size_t)(by*a.n)*a.w + bx*a.n;
  const int16_t *qm = a.qm + o0;
  const int q0 = a.q[band];
  const double beta = a.beta[band];
  int32_t x0[PVQ_MAXN];
  double x1[PVQ_MAXN], xa[PVQ_MAXN];
  double acc = 0;
  for (int i = 0; i < nn; i++) {
    int ro = a.tab[o0 + i];
    int32_t c = src[(size_t)(ro/a.n)*a.w + (ro%a.n)];
    x0[i] = c;
    // od_pvq_compute_gain: five sequential multiplies per term (src/pvq.c:460-463)
    acc += c*(double)c*qm[i]*PVQ_QM_SCALE_1*qm[i]*PVQ_QM_SCALE_1;
  }
  const double g = sqrt(acc);
  const double cg = pvq_gain_compand(g, q0, beta);
  for (int i = 0; i < nn; i++) x1[i] = x0[i]*qm[i]*PVQ_QM_SCALE_1;   // int*int first
  PvqBandRec rec;
  rec.cg = cg; rec.g = g; rec.pad = 0;
  rec.qg[0] = rec.qg[1] = 0; rec.k[0] = rec.k[1] = 0;
  rec.cos_dist[0] = rec.cos_dist[1] = 0; rec.dist[0] = rec.dist[1] = 0;
  int32_t *ybase = a.y + (((size_t)f*nblk + blk)*2)*a.ncoded + o0;
  int nc = 0;
  int i0 = (int)floor(cg);
  if (i0 < 1) i0 = 1;
  for (int i = i0; i <= ceil(cg) && nc < 2; i++, nc++) {
    const double qcg = i;
    const int k = pvq_k_noref(qcg, nn, beta);
    int32_t yp[PVQ_MAXN];
    const double cd = pvq_search_dev(x1, xa, nn, k, yp, qcg*cg);
    rec.qg[nc] = i;
    rec.k[nc] = k;
    rec.cos_dist[nc] = cd;
    rec.dist[nc] = 1.4*(qcg - cg)*(qcg - cg) + qcg*cg*(2 - 2*cd);
    int32_t *yo = ybase + (size_t)nc*a.ncoded;
    for (int j = 0; j < nn; j++) yo[j] = yp[j];
  }
  rec.ncand = nc;
  a.bands[((size_t)f*nblk + blk)*a.nbands + band] = rec;
}
