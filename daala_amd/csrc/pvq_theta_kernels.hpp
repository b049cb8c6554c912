// Vector-level PVQ kernels: the COMPLETE candidate enumeration of pvq_theta
// (with-reference gain/theta search + no-reference search), the decoder-side
// synthesis and the keyframe luma H/V predictor.  One lane per band vector,
// literal sequential arithmetic (see pvq_kernels.hpp).  These are the building
// blocks for the serial-order consumers (wavefront scheduling is a later round);
// they are correctness-first and not on the bench's hot loop.
//
// Transcendentals: acos / sin / cos (and pow for beta != 1) come from ROCm OCML,
// the reference uses glibc libm.  Both are sub-ulp accurate but not bit-identical,
// so doubles that depend on them (theta, the with-reference distortions) agree to
// a few ulp, while every integer outcome (candidate ranges, K, pulses, synthesised
// coefficients) is required identical in tests/ (DESIGN.md section 5).
#pragma once
#include "pvq_kernels.hpp"
#include "gen_lift_dct.hpp"

struct PvqThetaOut {       // == od_hip_pvq_theta_out == orc_theta_out
  double cg, cgr, g, gr, corr, theta, gain_offset, skip_dist, null_dist;
  int32_t icgr, m, s, nref, nnoref, theta_searched, noref_searched, pad;
  int32_t ref_qg[12], ref_itheta[12], ref_ts[12], ref_k[12];
  double ref_qtheta[12], ref_cos_dist[12], ref_dist[12];
  int32_t nr_qg[2], nr_k[2];
  double nr_cos_dist[2], nr_dist[2];
};

#define PVQ_PI 3.14159265358979323846      /* M_PI */

// od_pvq_compute_gain (src/pvq.c:456-468)
__device__ inline double pvq_compute_gain_dev(const int32_t *x, int n, int q0, double *g,
                                              double beta, const int16_t *qm) {
  double acc = 0;
  for (int i = 0; i < n; i++) {
    acc += x[i]*(double)x[i]*qm[i]*PVQ_QM_SCALE_1*qm[i]*PVQ_QM_SCALE_1;
  }
  *g = sqrt(acc);
  return pvq_gain_compand(*g, q0, beta);
}

// od_compute_householder / od_apply_householder (src/pvq.c:364-413)
__device__ inline int pvq_compute_householder_dev(double *r, int n, double gr, int *sign) {
  int m = 0;
  double maxr = 0;
  for (int i = 0; i < n; i++) {
    if (fabs(r[i]) > maxr) { maxr = fabs(r[i]); m = i; }
  }
  const int s = r[m] > 0 ? 1 : -1;
  r[m] += gr*s;
  *sign = s;
  return m;
}

__device__ inline void pvq_apply_householder_dev(double *x, const double *r, int n) {
  double l2r = 0, proj = 0;
  for (int i = 0; i < n; i++) l2r += r[i]*r[i];
  for (int i = 0; i < n; i++) proj += r[i]*x[i];
  const double proj_1 = proj*2./(1e-100 + l2r);
  for (int i = 0; i < n; i++) x[i] -= r[i]*proj_1;
}

// od_pvq_compute_max_theta / _theta / _k (src/pvq.c:476-535)
__device__ inline int pvq_max_theta_dev(double qcg, double beta) {
  int ts = (int)floor(.5 + qcg*PVQ_PI/(2*beta));
  if (qcg < 1.4) ts = 1;
  return ts;
}

__device__ inline double pvq_theta_dev(int t, int max_theta) {
  if (max_theta != 0) return (t < max_theta - 1 ? t : max_theta - 1)*.5*PVQ_PI/max_theta;
  return 0;
}

__device__ inline int pvq_k_ref_dev(double qcg, int itheta, double theta, int n, double beta,
                                    int nodesync) {
  if (itheta == 0) return 0;
  int k;
  if (nodesync) k = (int)floor(.5 + (itheta - .2)*sqrt((double)((n + 2)/2)));
  else k = (int)floor(.5 + (qcg*sin(theta) - .2)*sqrt((double)((n + 2)/2))/beta);
  return k > 1 ? k : 1;
}

// pvq_theta minus the rate term (src/pvq_encoder.c:311-481).
__global__ void k_pvq_theta_vectors(int n, int nvec, const int32_t *__restrict__ x0a,
                                    const int32_t *__restrict__ r0a,
                                    const int16_t *__restrict__ qm, const int32_t *__restrict__ q0a,
                                    double beta, int robust, int is_keyframe, int pli,
                                    PvqThetaOut *__restrict__ outa, int32_t *__restrict__ y_ref,
                                    int32_t *__restrict__ y_noref) {
  const long v = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (v >= nvec) return;
  const int32_t *x0 = x0a + v*n, *r0 = r0a + v*n;
  const int q0 = q0a[v];
  double x[PVQ_MAXN], r[PVQ_MAXN], xa[PVQ_MAXN];
  int32_t yp[PVQ_MAXN];
  PvqThetaOut o;
  memset(&o, 0, sizeof(o));
  const double gain_weight = 1.4;
  const int nodesync = robust || is_keyframe;
  double corr = 0, g, gr, theta = 0;
  for (int i = 0; i < n; i++) {
    x[i] = x0[i]*qm[i]*PVQ_QM_SCALE_1;
    r[i] = r0[i]*qm[i]*PVQ_QM_SCALE_1;
    corr += x[i]*r[i];
  }
  const int cfl_enabled = is_keyframe && pli != 0;
  const double cg = pvq_compute_gain_dev(x0, n, q0, &g, beta, qm);
  double cgr = pvq_compute_gain_dev(r0, n, q0, &gr, beta, qm);
  if (cfl_enabled) cgr = 1;
  const int icgr = (int)floor(.5 + cgr);
  const double gain_offset = cgr - icgr;
  corr = corr/(1e-100 + g*gr);
  corr = corr < 1. ? corr : 1.;
  corr = corr > -1. ? corr : -1.;
  o.null_dist = gain_weight*cg*cg;
  if (is_keyframe) o.skip_dist = gain_weight*cg*cg;
  else o.skip_dist = gain_weight*(cg - cgr)*(cg - cgr) + cgr*cg*(2 - 2*corr);
  int m = 0, s = 1;
  bool isnull = true;
  for (int i = 0; i < n; i++) if (r0[i]) isnull = false;
  if (n <= PVQ_MAXN && !isnull && corr > 0) {
    o.theta_searched = 1;
    theta = acos(corr);
    m = pvq_compute_householder_dev(r, n, gr, &s);
    pvq_apply_householder_dev(x, r, n);
    for (int i = m; i < n - 1; i++) x[i] = x[i + 1];
    int i = (int)floor(cg - gain_offset) - 1;
    if (i < 1) i = 1;
    for (; i <= (int)ceil(cg - gain_offset); i++) {
      const double qcg = i + gain_offset;
      const int ts = pvq_max_theta_dev(qcg, beta);
      int j = (int)floor(.5 + theta*2/PVQ_PI*ts) - 2;
      if (j < 0) j = 0;
      int jhi = (int)ceil(theta*2/PVQ_PI*ts);
      if (jhi > ts - 1) jhi = ts - 1;
      for (; j <= jhi; j++) {
        const int c = o.nref;
        if (c >= 12) break;
        const double qtheta = pvq_theta_dev(j, ts);
        const int k = pvq_k_ref_dev(qcg, j, qtheta, n, beta, nodesync);
        const double cos_dist = pvq_search_dev(x, xa, n - 1, k, yp,
                                               qcg*cg*sin(theta)*sin(qtheta));
        const double dist_theta = 2 - 2*cos(theta - qtheta)
                                  + sin(theta)*sin(qtheta)*(2 - 2*cos_dist);
        o.ref_qg[c] = i; o.ref_itheta[c] = j; o.ref_ts[c] = ts; o.ref_k[c] = k;
        o.ref_qtheta[c] = qtheta; o.ref_cos_dist[c] = cos_dist;
        o.ref_dist[c] = gain_weight*(qcg - cg)*(qcg - cg) + qcg*cg*dist_theta;
        int32_t *yo = y_ref + (v*12 + c)*n;
        for (int e = 0; e < n - 1; e++) yo[e] = yp[e];
        yo[n - 1] = 0;
        o.nref++;
      }
    }
  }
  if (n <= PVQ_MAXN && ((is_keyframe && pli == 0) || corr < .5 || cg < 2.)) {
    o.noref_searched = 1;
    for (int i = 0; i < n; i++) x[i] = x0[i]*qm[i]*PVQ_QM_SCALE_1;
    int i = (int)floor(cg);
    if (i < 1) i = 1;
    for (; i <= ceil(cg) && o.nnoref < 2; i++) {
      const int c = o.nnoref;
      const double qcg = i;
      const int k = pvq_k_noref(qcg, n, beta);
      const double cd = pvq_search_dev(x, xa, n, k, yp, qcg*cg);
      o.nr_qg[c] = i; o.nr_k[c] = k; o.nr_cos_dist[c] = cd;
      o.nr_dist[c] = gain_weight*(qcg - cg)*(qcg - cg) + qcg*cg*(2 - 2*cd);
      int32_t *yo = y_noref + (v*2 + c)*n;
      for (int e = 0; e < n; e++) yo[e] = yp[e];
      o.nnoref++;
    }
  }
  o.cg = cg; o.cgr = cgr; o.g = g; o.gr = gr; o.corr = corr; o.theta = theta;
  o.gain_offset = gain_offset; o.icgr = icgr; o.m = m; o.s = s;
  outa[v] = o;
}

// pvq_synthesis of the decoder (src/pvq_decoder.c:104-118) = Householder rebuild
// + od_pvq_synthesis_partial (src/pvq.c:552-585), both branches.
__global__ void k_pvq_synthesis_vectors(int n, int nvec, const int32_t *__restrict__ ya,
                                        const int32_t *__restrict__ refa,
                                        const double *__restrict__ gra,
                                        const int32_t *__restrict__ norefa,
                                        const double *__restrict__ ga,
                                        const double *__restrict__ thetaa,
                                        const int16_t *__restrict__ qm,
                                        const int16_t *__restrict__ qm_inv,
                                        int32_t *__restrict__ outa) {
  const long v = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (v >= nvec) return;
  const int32_t *yp = ya + v*n, *ref = refa + v*n;
  int32_t *out = outa + v*n;
  const int noref = norefa[v];
  const double g = ga[v], theta = thetaa[v];
  double r[PVQ_MAXN], x[PVQ_MAXN];
  int s = 0, m = 0;
  if (!noref) {
    for (int i = 0; i < n; i++) r[i] = ref[i]*qm[i]*PVQ_QM_SCALE_1;
    m = pvq_compute_householder_dev(r, n, gra[v], &s);
  }
  const int nn = n - (!noref);
  int yy = 0;
  for (int i = 0; i < nn; i++) yy += yp[i]*yp[i];
  double scale = yy == 0 ? 0 : g/sqrt((double)yy);
  if (noref) {
    for (int i = 0; i < n; i++) {
      out[i] = (int32_t)floor(.5 + (yp[i]*scale)*(qm_inv[i]*PVQ_QM_INV_SCALE_1));
    }
  }
  else {
    scale *= sin(theta);
    for (int i = 0; i < m; i++) x[i] = yp[i]*scale;
    x[m] = -s*g*cos(theta);
    for (int i = m; i < nn; i++) x[i + 1] = yp[i]*scale;
    pvq_apply_householder_dev(x, r, n);
    for (int i = 0; i < n; i++) {
      out[i] = (int32_t)floor(.5 + (x[i]*(qm_inv[i]*PVQ_QM_INV_SCALE_1)));
    }
  }
}

// Keyframe luma predictor = OD_CLEAR + od_hv_intra_pred (src/encode.c:732-737,
// src/intra.c:37-61) for a list of blocks of size bs.  One thread per block.
__global__ void k_hv_intra_pred_blocks(const int32_t *__restrict__ d, int w,
                                       const uint8_t *__restrict__ bsize, int bstride, int bs,
                                       int nblk, const int32_t *__restrict__ bxa,
                                       const int32_t *__restrict__ bya, int32_t *__restrict__ preda) {
  const long b = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (b >= nblk) return;
  const int n = 4 << bs, bx = bxa[b], by = bya[b];
  int32_t *pred = preda + b*n*n;
  for (int e = 0; e < n*n; e++) pred[e] = 0;
  const bool top = by > 0 && bsize[((by - 1) >> 1)*bstride + (bx >> 1)] == bs;
  const bool left = bx > 0 && bsize[(by >> 1)*bstride + ((bx - 1) >> 1)] == bs;
  const int32_t *t = d + (size_t)(by << 2)*w + (bx << 2);
  double g1 = 0, g2 = 0;
  if (top) for (int i = 1; i < 4; i++) g1 += t[-n*w + i]*(double)t[-n*w + i];
  if (left) for (int i = 1; i < 4; i++) g2 += t[-n + i*w]*(double)t[-n + i*w];
  if (top) for (int i = 4; i < n; i++) pred[i] = t[-n*w + i];
  if (left) for (int i = 4; i < n; i++) pred[i*n] = t[-n + i*w];
  if (g1 > g2) {
    if (top) for (int i = 1; i < 4; i++) pred[i] = t[-n*w + i];
  }
  else if (left) for (int i = 1; i < 4; i++) pred[i*n] = t[-n + i*w];
}

// Diagnostic: evaluates the transcendental functions the PVQ path uses so that
// tests can quantify OCML vs host-libm agreement.
// fn: 0 pow(x, y), 1 acos(x), 2 sin(x), 3 cos(x), 4 sqrt(x), 5 x/y, 6 pvq_pow_2_3(x),
// 7 pvq_pow_m1_6(x).
__global__ void k_libm_probe(int fn, int n, const double *__restrict__ x,
                             const double *__restrict__ y, double *__restrict__ out) {
  const long i = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r;
  switch (fn) {
    case 0: r = pow(x[i], y[i]); break;
    case 1: r = acos(x[i]); break;
    case 2: r = sin(x[i]); break;
    case 3: r = cos(x[i]); break;
    case 4: r = sqrt(x[i]); break;
    case 6: r = pvq_pow_2_3(x[i]); break;
    case 7: r = pvq_pow_m1_6(x[i]); break;
    default: r = x[i]/y[i]; break;
  }
  out[i] = r;
}

// A22: od_compute_dist (src/encode.c:940-1058, HVS-QM branch) for nblk pairs of
// n x n blocks (n = 8, 16, 32).  One thread per pair; the 8x8 sub-block sums are
// accumulated sequentially in raster order like the reference.  mag2: the 64
// squared weights for this block size (reference tables, passed in as data).
// pow(., -1/6) comes from OCML: the value is within a few ulp of the reference's
// (DESIGN.md section 5); everything else is exact.
__device__ inline int dist_var_4x4(const int32_t *x, int stride) {
  int sum = 0, s2 = 0;
  for (int i = 0; i < 4; i++) {
    for (int j = 0; j < 4; j++) {
      const int t = x[i*stride + j] >> 2;
      sum += t;
      s2 += t*t;
    }
  }
  return s2 - (sum*sum >> 4);
}

__global__ void k_compute_dist_blocks(int n, int nblk, const int32_t *__restrict__ xa,
                                      const int32_t *__restrict__ ya,
                                      const double *__restrict__ mag2, int masking,
                                      double *__restrict__ out) {
  const long b = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (b >= nblk) return;
  const int32_t *x = xa + b*n*n, *y = ya + b*n*n;
  double total = 0;
  for (int bi = 0; bi < n; bi += 8) {
    for (int bj = 0; bj < n; bj += 8) {
      const int32_t *xs = x + bi*n + bj, *ys = y + bi*n + bj;
      double mean_var = 0, vardist = 0;
      int min_var = 2147483647;
      for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
          const int varx = dist_var_4x4(xs + 2*i*n + 2*j, n);
          const int vary = dist_var_4x4(ys + 2*i*n + 2*j, n);
          min_var = varx < min_var ? varx : min_var;
          mean_var += 1./(1 + varx);
          const double diff = sqrt((double)varx) - sqrt((double)vary);
          vardist += diff*diff;
        }
      }
      double calibration, var_stat;
      if (masking) { calibration = 1.95; var_stat = 9./mean_var; }
      else { calibration = 1.62; var_stat = min_var; }
      const double activity = calibration*pvq_pow_m1_6(.25 + var_stat/(1 << 2*4));
      // 8x8 fDCT of the error: columns into rows of z, then columns of z into rows
      int32_t z[64], et[64];
      for (int c = 0; c < 8; c++) {
        int32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = xs[k*n + c] - ys[k*n + c];
        LiftDct<8, false>::fwd(v);
#pragma unroll
        for (int k = 0; k < 8; k++) z[c*8 + k] = v[k];
      }
      for (int c = 0; c < 8; c++) {
        int32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = z[k*8 + c];
        LiftDct<8, false>::fwd(v);
#pragma unroll
        for (int k = 0; k < 8; k++) et[c*8 + k] = v[k];
      }
      double sum = 0;
      for (int i = 0; i < 64; i++) sum += et[i]*(double)et[i]*mag2[i];
      total += activity*activity*(sum + vardist);
    }
  }
  out[b] = total*1.7;
}
