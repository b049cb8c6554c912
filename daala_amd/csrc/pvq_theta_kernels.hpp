// Vector-level PVQ kernels: the COMPLETE candidate enumeration of pvq_theta
// (with-reference gain/theta search + no-reference search), the decoder-side
// synthesis and the keyframe luma H/V predictor.  One lane per band vector,
// literal sequential arithmetic (see pvq_kernels.hpp).  These are the building
// blocks for the serial-order consumers (wavefront scheduling is a later round);
// they are correctness-first and not on the bench's hot loop.
//
// Transcendentals: NONE on the device.  pow (gain companding), acos (theta), sin / cos
// (theta candidates, synthesis) and pow(., -1/6) (od_compute_dist) are evaluated by the
// host's libm between two device passes (daala_hip.hip: the reference links glibc, whose
// results are not correctly rounded, so only the same libm reproduces its bits); the
// device does +, -, *, /, sqrt, floor and compares - all exactly rounded - in the
// reference's operand order.  Every output is therefore bit-identical by construction.
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

// od_pvq_compute_gain (src/pvq.c:456-464) up to the companding: g = sqrt(acc)
__device__ inline double pvq_raw_gain_dev(const int32_t *x, int n, const int16_t *qm) {
  double acc = 0;
  for (int i = 0; i < n; i++) {
    acc += x[i]*(double)x[i]*qm[i]*PVQ_QM_SCALE_1*qm[i]*PVQ_QM_SCALE_1;
  }
  return sqrt(acc);
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

// pvq_theta minus the rate term (src/pvq_encoder.c:311-481) in two device passes around
// the host's libm stage.
struct PvqThetaPrep {      // pass 1 -> host
  double g, gr, corr_sum;
  int32_t isnull, pad;
};

struct PvqThetaCands {     // host -> pass 2: the searches to run
  int32_t theta_searched, noref_searched, nref, nnoref;
  int32_t ref_k[12], nr_k[2];
  double ref_g2[12], nr_g2[2];
};

struct PvqThetaRes {       // pass 2 -> host
  double ref_cos_dist[12], nr_cos_dist[2];
  int32_t m, s;
};

// pass 1 (:353-361): gains of x and r and the raw correlation sum
__global__ void k_pvq_theta_prep(int n, int nvec, const int32_t *__restrict__ x0a,
                                 const int32_t *__restrict__ r0a,
                                 const int16_t *__restrict__ qm, PvqThetaPrep *__restrict__ outa) {
  const long v = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (v >= nvec) return;
  const int32_t *x0 = x0a + v*n, *r0 = r0a + v*n;
  PvqThetaPrep o;
  double corr = 0;
  bool isnull = true;
  for (int i = 0; i < n; i++) {
    const double x = x0[i]*qm[i]*PVQ_QM_SCALE_1;
    const double r = r0[i]*qm[i]*PVQ_QM_SCALE_1;
    corr += x*r;
    if (r0[i]) isnull = false;
  }
  o.g = pvq_raw_gain_dev(x0, n, qm);
  o.gr = pvq_raw_gain_dev(r0, n, qm);
  o.corr_sum = corr;
  o.isnull = isnull;
  o.pad = 0;
  outa[v] = o;
}

// pass 2: Householder (:402-404) and every codeword search (:426, :463) with the K and
// distortion multiplier g2 the host derived
__global__ void k_pvq_theta_search(int n, int nvec, const int32_t *__restrict__ x0a,
                                   const int32_t *__restrict__ r0a,
                                   const int16_t *__restrict__ qm,
                                   const PvqThetaPrep *__restrict__ prep,
                                   const PvqThetaCands *__restrict__ cands,
                                   PvqThetaRes *__restrict__ resa, int32_t *__restrict__ y_ref,
                                   int32_t *__restrict__ y_noref) {
  const long v = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (v >= nvec) return;
  const int32_t *x0 = x0a + v*n, *r0 = r0a + v*n;
  const PvqThetaCands &cd = cands[v];
  double x[PVQ_MAXN], r[PVQ_MAXN], xa[PVQ_MAXN];
  int32_t yp[PVQ_MAXN];
  PvqThetaRes o;
  memset(&o, 0, sizeof(o));
  o.s = 1;
  if (cd.theta_searched) {
    for (int i = 0; i < n; i++) {
      x[i] = x0[i]*qm[i]*PVQ_QM_SCALE_1;
      r[i] = r0[i]*qm[i]*PVQ_QM_SCALE_1;
    }
    int s;
    const int m = pvq_compute_householder_dev(r, n, prep[v].gr, &s);
    pvq_apply_householder_dev(x, r, n);
    for (int i = m; i < n - 1; i++) x[i] = x[i + 1];
    o.m = m;
    o.s = s;
    for (int c = 0; c < cd.nref; c++) {
      o.ref_cos_dist[c] = pvq_search_dev(x, xa, n - 1, cd.ref_k[c], yp, cd.ref_g2[c]);
      int32_t *yo = y_ref + (v*12 + c)*n;
      for (int e = 0; e < n - 1; e++) yo[e] = yp[e];
      yo[n - 1] = 0;
    }
  }
  if (cd.noref_searched) {
    for (int i = 0; i < n; i++) x[i] = x0[i]*qm[i]*PVQ_QM_SCALE_1;
    for (int c = 0; c < cd.nnoref; c++) {
      o.nr_cos_dist[c] = pvq_search_dev(x, xa, n, cd.nr_k[c], yp, cd.nr_g2[c]);
      int32_t *yo = y_noref + (v*2 + c)*n;
      for (int e = 0; e < n; e++) yo[e] = yp[e];
    }
  }
  resa[v] = o;
}

// pvq_synthesis of the decoder (src/pvq_decoder.c:104-118) = Householder rebuild
// + od_pvq_synthesis_partial (src/pvq.c:552-585), both branches.
__global__ void k_pvq_synthesis_vectors(int n, int nvec, const int32_t *__restrict__ ya,
                                        const int32_t *__restrict__ refa,
                                        const double *__restrict__ gra,
                                        const int32_t *__restrict__ norefa,
                                        const double *__restrict__ ga,
                                        const double *__restrict__ sina,
                                        const double *__restrict__ cosa,
                                        const int16_t *__restrict__ qm,
                                        const int16_t *__restrict__ qm_inv,
                                        int32_t *__restrict__ outa) {
  const long v = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (v >= nvec) return;
  const int32_t *yp = ya + v*n, *ref = refa + v*n;
  int32_t *out = outa + v*n;
  const int noref = norefa[v];
  const double g = ga[v], sin_theta = sina[v], cos_theta = cosa[v];     // host libm
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
    scale *= sin_theta;
    for (int i = 0; i < m; i++) x[i] = yp[i]*scale;
    x[m] = -s*g*cos_theta;
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
// fn: 0 pow(x, y), 1 acos(x), 2 sin(x), 3 cos(x) - OCML, NOT used by any product kernel:
// this probe documents why (they differ from glibc in the last place) -, 4 sqrt(x), 5 x/y
// (used everywhere, must be identical to the host's).
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
    default: r = x[i]/y[i]; break;
  }
  out[i] = r;
}

// A22: od_compute_dist (src/encode.c:940-1058, HVS-QM branch) for nblk pairs of
// n x n blocks (n = 8, 16, 32).  One thread per pair and 8x8 sub-block: the device
// produces, per sub-block, the argument of the activity power (.25 + var_stat/256) and
// the weighted error energy sum + vardist; the host applies calibration*pow(arg, -1/6)
// with its libm and adds the sub-blocks up in the reference's raster order.  mag2: the 64
// squared weights for this block size (reference tables, passed in as data).
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

// od_compute_dist_8x8 (src/encode.c:962-1030) up to the activity power: the argument of
// pow() and the weighted error energy of one 8x8 sub-block.  x(i, j), y(i, j): the two
// blocks' samples.
template <typename FX, typename FY>
__device__ __forceinline__ void dist8x8_core(FX x, FY y, const double *__restrict__ mag2, int masking,
                                             double *arg, double *energy) {
  double mean_var = 0, vardist = 0;
  int min_var = 2147483647;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) {
      int sx = 0, s2x = 0, sy = 0, s2y = 0;
      for (int u = 0; u < 4; u++) {
        for (int v = 0; v < 4; v++) {
          const int tx = x(2*i + u, 2*j + v) >> 2, ty = y(2*i + u, 2*j + v) >> 2;
          sx += tx; s2x += tx*tx;
          sy += ty; s2y += ty*ty;
        }
      }
      const int varx = s2x - (sx*sx >> 4), vary = s2y - (sy*sy >> 4);
      min_var = varx < min_var ? varx : min_var;
      mean_var += 1./(1 + varx);
      const double diff = sqrt((double)varx) - sqrt((double)vary);
      vardist += diff*diff;
    }
  }
  const double var_stat = masking ? 9./mean_var : (double)min_var;
  // 8x8 fDCT of the error: columns into rows of z, then columns of z into rows
  int32_t z[64], et[64];
  for (int c = 0; c < 8; c++) {
    int32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = x(k, c) - y(k, c);
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
  *arg = .25 + var_stat/(1 << 2*4);
  *energy = sum + vardist;
}

__global__ void k_compute_dist_blocks(int n, int nblk, const int32_t *__restrict__ xa,
                                      const int32_t *__restrict__ ya,
                                      const double *__restrict__ mag2, int masking,
                                      double *__restrict__ arg, double *__restrict__ energy) {
  const int per = (n/8)*(n/8);
  const long t = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= (long)nblk*per) return;
  const long b = t/per;
  const int sb = (int)(t%per), bi = (sb/(n/8))*8, bj = (sb%(n/8))*8;
  const int32_t *xs = xa + b*n*n + bi*n + bj, *ys = ya + b*n*n + bi*n + bj;
  dist8x8_core([&](int i, int j) { return xs[i*n + j]; }, [&](int i, int j) { return ys[i*n + j]; },
               mag2, masking, &arg[t], &energy[t]);
}

// The encoder's deringing on/off decision (src/encode.c:2606-2636) compares, for every 32x32
// luma superblock, od_compute_dist(original, unfiltered reconstruction) with
// od_compute_dist(original, deringed reconstruction): both for EVERY superblock of the frame
// in one launch, right after the deringing pass that produced the second operand.  One
// thread per (superblock, 8x8 sub-block, operand).  orig: the padded 8-bit input plane
// ((p - 128) << 4 on the fly, od_ref_buf_to_coeff :2608).
__global__ void k_dering_dist(int nhsb, int nvsb, int w, const uint8_t *__restrict__ orig, int ostride,
                              const int16_t *__restrict__ unf, const int16_t *__restrict__ filt,
                              const double *__restrict__ mag2, int masking,
                              double *__restrict__ arg, double *__restrict__ e_unf,
                              double *__restrict__ e_filt) {
  const long t = (long)blockIdx.x*blockDim.x + threadIdx.x;
  const long total = (long)nhsb*nvsb*32;
  if (t >= total) return;
  const int which = (int)(t & 1), sub = (int)((t >> 1) & 15);
  const long sb = t >> 5;
  const int sbx = (int)(sb%nhsb), sby = (int)(sb/nhsb);
  const int y0 = sby*32 + (sub >> 2)*8, x0 = sbx*32 + (sub & 3)*8;
  const uint8_t *o = orig + (size_t)y0*ostride + x0;
  const int16_t *r = (which ? filt : unf) + (size_t)y0*w + x0;
  double a, e;
  dist8x8_core([&](int i, int j) { return ((int)o[i*ostride + j] - 128) << 4; },
               [&](int i, int j) { return (int)r[i*w + j]; }, mag2, masking, &a, &e);
  const long idx = sb*16 + sub;
  if (which) e_filt[idx] = e;
  else { e_unf[idx] = e; arg[idx] = a; }
}
