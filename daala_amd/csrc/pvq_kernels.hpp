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

struct PvqBandRec {      // mirrors od_hip_pvq_band (include/daala_hip.h); host-side AoS
  double cg, g;
  double cos_dist[2];
  double dist[2];
  int32_t qg[2];
  int32_t k[2];
  int32_t ncand;
  int32_t pad;
};

// Device-side result layout of one (plane, level), per frame slot: structure of
// arrays, band-major, so that a wave (64 consecutive blocks of one band) writes
// contiguous memory.  All arrays are [band][block]:
struct PvqSoA {
  double *cg, *g;           // [nbands*nblk]
  double *cos_dist, *dist;  // [2][nbands*nblk] per frame
  int32_t *qg, *k;          // [2][nbands*nblk] per frame
  int32_t *ncand;           // [nbands*nblk]
  // pulses: band b occupies y + 2*nblk*(off[b]-1), laid out [cand][block][n_b]
  int32_t *y;
  // work-balancing order (performance only): perm[band*nblk + i] = the block the i-th lane
  // slot of that band processes; written by the host with the companded gains, which fix
  // every K.  nullptr: blocks in raster order.
  const int32_t *perm;
  int32_t *perm_rw;         // host side: the allocation perm points into (all slots)
};

struct PvqLevelArgs {
  const int32_t *lev;      // level plane, frame 0
  size_t lev_fstride;      // elements between frames
  int w;                   // plane stride
  int n;                   // block size
  int nbx, nby;            // blocks per row / column
  int nbands;              // bands of this block size
  int off[11];             // band boundaries (coding order)
  int q[10];
  double beta[10];
  int band_list[10];       // bands handled by this launch (all of size N)
  const uint16_t *tab;     // coding index -> raster offset (y*n + x)
  const int16_t *qm;       // n*n, coding order
  PvqSoA out;              // frame 0
  size_t rec_fstride;      // nbands*nblk (elements between frames of the record arrays)
  size_t y_fstride;        // 2*nblk*(ncoded-1)
  long blk_first, blk_end; // blocks [blk_first, blk_end) of the level are processed (a strip of SB rows)
};

// ===========================================================================
// v3: register-resident search.  A band of N coefficients is owned by G lanes
// (G = 1 for N <= 32, G = 4 for N = 127/128), NL = ceil(N/G) consecutive
// coefficients per lane, all in VGPRs (every array index is a compile-time
// constant after unrolling).  Exactness:
//   * floating-point sums (gain, xx, l1, projection xy) are order dependent, so
//     they run as ONE sequential chain in index order; with G > 1 the running sum
//     is handed from lane g to lane g+1 (G rounds in lock step).
//   * RDO-phase argmax compares plain doubles with strict '>' and first-index
//     wins: a total order, so per-lane scans + an ordered combine are exact.
//   * greedy-phase argmax uses the reference's cross-multiplied compare
//     fl(a_j*b_best) > fl(a_best*b_j), which is NOT guaranteed transitive under
//     rounding.  With G > 1 the per-lane scans + ordered combine give a candidate
//     w; w is then VERIFIED: it is the sequential result if it strictly beats
//     every earlier element and no later element strictly beats it.  If any band
//     of the wave fails the check (only possible at ~1-ulp near-ties) the wave
//     redoes that pulse with the literal sequential scan (lane to lane hand-off).
// ===========================================================================

// 1/sqrt(i) table (i < PVQ_RSQ_TAB) filled on the device by k_pvq_fill_rsqrt with
// the same operations as pvq_rsqrt_small, so lookups are bit-identical.
#define PVQ_RSQ_TAB 4096
__global__ void k_pvq_fill_rsqrt(double *tab) {
  int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i < PVQ_RSQ_TAB) tab[i] = i >= 1 ? pvq_rsqrt_small(i) : 0;
}

__device__ __attribute__((noinline)) double pvq_rsqrt_slow(int i) {
  return 1./sqrt((double)i);
}

#ifndef PVQ_RSQ_LDS
#define PVQ_RSQ_LDS 0         /* entries of the table also kept in LDS per workgroup (0: none) */
#endif
struct PvqRsq {
  const double *tab;          // global, PVQ_RSQ_TAB entries
  const double *lds;          // LDS copy of the first PVQ_RSQ_LDS entries, or nullptr
};
__device__ __forceinline__ double pvq_rsqrt_tab(const PvqRsq &r, int i) {
  if (PVQ_RSQ_LDS > 0 && i < PVQ_RSQ_LDS) return r.lds[i];
  if (__builtin_expect(i < PVQ_RSQ_TAB, 1)) return r.tab[i];
  return pvq_rsqrt_slow(i);
}

#ifndef PVQ_G128
#define PVQ_G128 8            /* lanes per 128-coefficient band (A/B on MI355X: 4 -> 8.5 ms, 8 -> 4.6, 16 -> 5.9) */
#endif
#ifndef PVQ_G32
#define PVQ_G32 4             /* lanes per 31/32-coefficient band (1: LDS kernel 3.05 ms, 2 -> 2.84, 4 -> 2.30, 8 -> 3.18) */
#endif
#ifndef PVQ_G16
#define PVQ_G16 1             /* lanes per 7..15-coefficient band */
#endif
template <int N>
struct PvqGeom {
  static constexpr int G = N > 32 ? PVQ_G128 : N > 16 ? PVQ_G32 : PVQ_G16;
  static constexpr int NL = (N + G - 1)/G;
  static constexpr int BPW = 64/G;               // bands per wave
};

// Per-band search state shared by all candidates of one input vector.
template <int N>
struct PvqVec {
  static constexpr int NL = PvqGeom<N>::NL;
  double x[NL];        // |x_j| of this lane's chunk, 0 beyond the band
  uint32_t neg;        // sign bits of the chunk
  double xx, l1_inv, norm_1;
};

// Sequential-order sum of term(j) over the whole band, result on every lane.
template <int N, typename F>
__device__ __forceinline__ double pvq_chain_sum(int g, int lane, F term) {
  constexpr int G = PvqGeom<N>::G, NL = PvqGeom<N>::NL;
  double acc = 0;
  if (G == 1) {
#pragma unroll
    for (int j = 0; j < NL; j++) acc += term(j);
    return acc;
  }
#pragma unroll
  for (int r = 0; r < G; r++) {
    if (g == r) {
#pragma unroll
      for (int j = 0; j < NL; j++) {
        if (r*NL + j < N) acc += term(j);
      }
    }
    if (r + 1 < G) {
      double up = __shfl_up(acc, 1, 64);
      if (g == r + 1) acc = up;
    }
  }
  return __shfl(acc, (lane/G)*G + G - 1, 64);
}

template <int N>
__device__ __forceinline__ void pvq_vec_finish(PvqVec<N> &v, int g, int lane) {
  v.xx = pvq_chain_sum<N>(g, lane, [&](int j) { return v.x[j]*v.x[j]; });
  double l1 = pvq_chain_sum<N>(g, lane, [&](int j) { return v.x[j]; });
  v.norm_1 = 1./sqrt(1e-30 + v.xx);
  v.l1_inv = 1./(l1 > 1e-100 ? l1 : 1e-100);
}

// One codeword search (pvq_search_rdo_double, src/pvq_encoder.c:121-225) for the
// band owned by this lane group.  `act`: this band really has a candidate (lanes
// of inactive bands run with k = 0).  Returns the cosine distance; y = unsigned
// pulses of this lane's chunk.
template <int N>
__device__ __forceinline__ double pvq_search_v3(const PvqVec<N> &v, int g, int lane, int k,
                                                double g2, const PvqRsq &rsq,
                                                int (&y)[PvqGeom<N>::NL], int &npulse_greedy,
                                                int &npulse_rdo) {
  constexpr int G = PvqGeom<N>::G, NL = PvqGeom<N>::NL;
  const int base = (lane/G)*G;                  // first lane of my band
  const int nvalid = (N - g*NL) < NL ? (N - g*NL) : NL;
  const double lambda = PVQ_LAMBDA/(1e-30 + g2);
  const double delta_rate = 3./N;
  double xy = 0, yy = 0;
  int i = 0;
  if (k > 2) {
#pragma unroll
    for (int j = 0; j < NL; j++) {
      int p = (int)floor(k*v.x[j]*v.l1_inv);
      y[j] = p > 0 ? p : 0;
    }
    xy = pvq_chain_sum<N>(g, lane, [&](int j) { return v.x[j]*y[j]; });
    int s2 = 0, s1 = 0;
#pragma unroll
    for (int j = 0; j < NL; j++) { s2 += y[j]*y[j]; s1 += y[j]; }
#pragma unroll
    for (int o = 1; o < G; o <<= 1) {
      s2 += __shfl_xor(s2, o, 64);
      s1 += __shfl_xor(s1, o, 64);
    }
    // yy accumulates exact integers in double (< 2^53): any order is exact.
    // NOTE: the reference adds ypulse[j]*ypulse[j] as int products one by one;
    // each partial sum is an exactly representable integer, so the value is the same.
    yy = (double)s2;
    i = s1;
  }
  else {
#pragma unroll
    for (int j = 0; j < NL; j++) y[j] = 0;
  }
  const int rdo_pulses = 1 + k/4;
  // ---- greedy phase ----------------------------------------------------------
  // wave-uniform trip count: bands that are done idle (masked) meanwhile
  while (__any(i < k - rdo_pulses)) {
    const bool run = i < k - rdo_pulses;
    double ba = 0, bb = 1;
    int bpos = 0;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      if (G*NL == N || j < nvalid) {
        double a = xy + v.x[j];
        const double b = yy + (2*y[j] + 1);
        a *= a;
        if (j == 0 || a*bb > ba*b) { ba = a; bb = b; bpos = j; }
      }
    }
    int wl = bpos, wg = 0;            // winner: local index and owning lane group index
    if (G > 1) {
      double ca = __shfl(ba, base, 64), cb = __shfl(bb, base, 64);
      int cp = __shfl(bpos, base, 64);
      wg = 0;
#pragma unroll
      for (int r = 1; r < G; r++) {
        const double ra = __shfl(ba, base + r, 64), rb = __shfl(bb, base + r, 64);
        const int rp = __shfl(bpos, base + r, 64);
        if (ra*cb > ca*rb) { ca = ra; cb = rb; cp = rp; wg = r; }
      }
      wl = cp;
      // verification: w = (wg, wl) must strictly beat every earlier element and
      // must not be strictly beaten by any later one
      bool ok = true;
#pragma unroll
      for (int j = 0; j < NL; j++) {
        if (j < nvalid) {
          double a = xy + v.x[j];
          const double b = yy + (2*y[j] + 1);
          a *= a;
          const bool before = g < wg || (g == wg && j < wl);
          const bool after = g > wg || (g == wg && j > wl);
          if (before) ok = ok && (ca*b > a*cb);
          if (after) ok = ok && !(a*cb > ca*b);
        }
      }
      if (__any(run && !ok)) {
        // literal sequential scan, incumbent handed from lane to lane
        double ia = 0, ib = 1;
        int ip = 0, ig = 0;
#pragma unroll
        for (int r = 0; r < G; r++) {
          if (g == r) {
#pragma unroll
            for (int j = 0; j < NL; j++) {
              if (j < nvalid) {
                double a = xy + v.x[j];
                const double b = yy + (2*y[j] + 1);
                a *= a;
                if ((r == 0 && j == 0) || a*ib > ia*b) { ia = a; ib = b; ip = j; ig = r; }
              }
            }
          }
          if (r + 1 < G) {
            const double ua = __shfl_up(ia, 1, 64), ub = __shfl_up(ib, 1, 64);
            const int up = __shfl_up(ip, 1, 64), ug = __shfl_up(ig, 1, 64);
            if (g == r + 1) { ia = ua; ib = ub; ip = up; ig = ug; }
          }
        }
        wl = __shfl(ip, base + G - 1, 64);
        wg = __shfl(ig, base + G - 1, 64);
      }
    }
    // apply the pulse
    double xw = 0;
    int yw = 0;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      if (j == wl) { xw = v.x[j]; yw = y[j]; }
    }
    if (G > 1) {
      xw = __shfl(xw, base + wg, 64);
      yw = __shfl(yw, base + wg, 64);
    }
    if (run) {
      xy = xy + xw;
      yy = yy + (2*yw + 1);
      if (g == wg) {
#pragma unroll
        for (int j = 0; j < NL; j++) y[j] += (j == wl);
      }
      i++;
      npulse_greedy++;
    }
  }
  // ---- RDO phase -------------------------------------------------------------
  // cost_j = 2*(xy + x_j)*norm_1*rsqrt(yy + 2*y_j + 1) - lambda*j*delta_rate, plain
  // double compare, strict '>', first index wins (src/pvq_encoder.c:195-220).
  while (__any(i < k)) {
    const bool run = i < k;
    const int iy = (int)yy;
    // the reference tabulates y = 0..3 per pulse (:205); same values here
    const double r0 = pvq_rsqrt_tab(rsq, iy + 1), r1 = pvq_rsqrt_tab(rsq, iy + 3);
    const double r2 = pvq_rsqrt_tab(rsq, iy + 5), r3 = pvq_rsqrt_tab(rsq, iy + 7);
    double bc = 0;
    int bpos = 0;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      if (G*NL == N || j < nvalid) {
        const int yj = y[j];
        double rs = yj == 0 ? r0 : yj == 1 ? r1 : yj == 2 ? r2 : r3;
        if (__builtin_expect(yj > 3, 0)) rs = pvq_rsqrt_tab(rsq, iy + 2*yj + 1);
        double c = xy + v.x[j];
        c = 2*c*v.norm_1*rs - lambda*(g*NL + j)*delta_rate;
        if (j == 0 || c > bc) { bc = c; bpos = j; }
      }
    }
    int wl = bpos, wg = 0;
    if (G > 1) {
      double cc = __shfl(bc, base, 64);
      int cp = __shfl(bpos, base, 64);
#pragma unroll
      for (int r = 1; r < G; r++) {
        const double rc = __shfl(bc, base + r, 64);
        const int rp = __shfl(bpos, base + r, 64);
        if (rc > cc) { cc = rc; cp = rp; wg = r; }
      }
      wl = cp;
    }
    double xw = 0;
    int yw = 0;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      if (j == wl) { xw = v.x[j]; yw = y[j]; }
    }
    if (G > 1) {
      xw = __shfl(xw, base + wg, 64);
      yw = __shfl(yw, base + wg, 64);
    }
    if (run) {
      xy = xy + xw;
      yy = yy + (2*yw + 1);
      if (g == wg) {
#pragma unroll
        for (int j = 0; j < NL; j++) y[j] += (j == wl);
      }
      i++;
      npulse_rdo++;
    }
  }
  return xy/(1e-100 + sqrt(v.xx*yy));
}

struct PvqLevelArgs3 {
  PvqLevelArgs a;
  const double *rsq;
  // optional work counters (measurement only): [0] pulses placed by greedy scans x N,
  // [1] pulses placed by RDO scans x N, [2] candidates searched - the algorithmic element
  // steps of pvq_search_rdo_double (src/pvq_encoder.c:166-220), summed over the launch
  unsigned long long *stats;
};

// No-reference candidates (state-free part of pvq_theta, src/pvq_encoder.c:352-357,
// :452-481) of the bands of size N of one pyramid level, register-resident.
#ifndef PVQ_STAGE_MIN_N
#define PVQ_STAGE_MIN_N 32    /* bands longer than this are gathered through LDS (measured: pays for 128 only) */
#endif
#ifndef PVQ_V3_WAVES
#define PVQ_V3_WAVES(N) 3     /* min waves/SIMD: 3 measured best for every N (4+ spills, 1-2 starves) */
#endif
// Two launches per level: GAIN_ONLY computes the exact uncompanded gain g = sqrt(acc) of
// every band (A13 up to the companding); the host then turns g into cg with ITS libm
// (od_gain_compand's pow, src/pvq.c:422 - the one operation on this path that is not
// +,-,*,/,sqrt,floor, and the only one whose result depends on the libm in use) and the
// search launch reads cg back: the device never evaluates a transcendental.
template <int N, bool GAIN_ONLY>
__global__ __launch_bounds__(64, PVQ_V3_WAVES(N)) void k_pvq_noref_v3(PvqLevelArgs3 aa) {
  const PvqLevelArgs &a = aa.a;
  constexpr int G = PvqGeom<N>::G, NL = PvqGeom<N>::NL, BPW = PvqGeom<N>::BPW;
  __shared__ int32_t Yst[BPW*(N + G + 2) + 1];      // >= BPW*N (output) and >= BPW*LDSN (staging)
  __shared__ int32_t Org[64];
  const int lane = threadIdx.x;
  const int g = lane%G, inst = lane/G;
  __shared__ int32_t Pb[64];                        // the blocks of this wave's lane slots
  const long idx0 = a.blk_first + (long)blockIdx.x*BPW;
  const int band = a.band_list[blockIdx.y], f = blockIdx.z;
  const long nblk = (long)a.nbx*a.nby;
  // Waves run until their slowest band is done (trip count = max K of the wave), so the
  // host orders the blocks of every band by K: neighbouring lane slots get similar K.
  const int32_t *perm = (!GAIN_ONLY && a.out.perm) ? a.out.perm + (size_t)f*a.rec_fstride + (size_t)band*nblk : nullptr;
  if (lane < BPW) {
    const long i = idx0 + lane;
    Pb[lane] = i < a.blk_end ? (perm ? perm[i] : (int32_t)i) : -1;
  }
  __syncthreads();
  const bool live = idx0 + inst < a.blk_end;
  const long blk = live ? Pb[inst] : 0;
  const long blk0 = idx0;
  const int o0 = a.off[band];
  const int q0 = a.q[band];
  const double beta = a.beta[band];
  const size_t rin = (size_t)band*nblk + (live ? blk : 0);
  const size_t rec = (size_t)f*a.rec_fstride + rin;
  const size_t rec2 = (size_t)f*2*a.rec_fstride + rin;
  // Gather the band of the wave's BPW blocks.  Reading "lane = block" straight
  // from the raster level plane costs one L1 transaction per lane (64 per load):
  // the first profile showed the N = 15 kernel bound by exactly that rate.  So
  // the wave loads cooperatively with lane = coefficient-within-block (lanes that
  // share a block share its few 64-byte sectors), stages in LDS and every lane
  // then picks up its own chunk.
  constexpr int CH = NL + 1;                       // padded chunk stride in LDS
  constexpr int LDSN = (G*CH) | 1;                 // odd per-band stride
  int32_t cf[NL];
  int qi[NL];
  constexpr bool STAGE = N > PVQ_STAGE_MIN_N;
  if (!STAGE) {
    const long bsafe = live ? blk : 0;
    const int bx = bsafe%a.nbx, by = bsafe/a.nbx;
    const int32_t *src = a.lev + (size_t)f*a.lev_fstride + (size_t)(by*a.n)*a.w + bx*a.n;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      const int jj = g*NL + j;
      if (live && jj < N) {
        const int ro = a.tab[o0 + jj];
        cf[j] = src[(size_t)(ro/a.n)*a.w + (ro%a.n)];
        qi[j] = a.qm[o0 + jj];
      }
      else { cf[j] = 0; qi[j] = 0; }
    }
  }
  else {
    const int32_t *plane = a.lev + (size_t)f*a.lev_fstride;
    const int nb_here = (int)(a.blk_end - blk0 < BPW ? a.blk_end - blk0 : BPW);
    if (lane < nb_here) {                          // block origins: one division per block
      const long bb = Pb[lane];
      const int bx = bb%a.nbx, by = bb/a.nbx;
      Org[lane] = (by*a.n)*a.w + bx*a.n;
    }
    __syncthreads();
    const int lg = a.n == 4 ? 2 : a.n == 8 ? 3 : a.n == 16 ? 4 : 5;
    for (int e = lane; e < nb_here*N; e += 64) {
      const int b = e/N, jj = e%N;
      const int ro = a.tab[o0 + jj];
      Yst[b*LDSN + (jj/NL)*CH + jj%NL] = plane[(size_t)Org[b] + (ro >> lg)*a.w + (ro & (a.n - 1))];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NL; j++) {
      const int jj = g*NL + j;
      if (live && jj < N) {
        cf[j] = Yst[inst*LDSN + g*CH + j];
        qi[j] = a.qm[o0 + jj];
      }
      else { cf[j] = 0; qi[j] = 0; }
    }
    __syncthreads();
  }
  if (GAIN_ONLY) {
    // od_pvq_compute_gain: five sequential multiplies per term (src/pvq.c:460-463)
    const double acc = pvq_chain_sum<N>(g, lane, [&](int j) {
      return cf[j]*(double)cf[j]*qi[j]*PVQ_QM_SCALE_1*qi[j]*PVQ_QM_SCALE_1;
    });
    if (live && g == 0) a.out.g[rec] = sqrt(acc);
    return;
  }
#if PVQ_RSQ_LDS > 0
  __shared__ double RsqL[PVQ_RSQ_LDS];
  for (int e = lane; e < PVQ_RSQ_LDS; e += 64) RsqL[e] = aa.rsq[e];
  __syncthreads();
  const PvqRsq rsq{aa.rsq, RsqL};
#else
  const PvqRsq rsq{aa.rsq, nullptr};
#endif
  const double cg = a.out.cg[rec];                 // companded on the host from out.g
  PvqVec<N> v;
  v.neg = 0;
#pragma unroll
  for (int j = 0; j < NL; j++) {
    const int pr = cf[j]*qi[j];                    // the reference's int*int product (:455)
    v.x[j] = fabs(pr*PVQ_QM_SCALE_1);
    v.neg |= (uint32_t)(pr < 0) << j;
  }
  pvq_vec_finish<N>(v, g, lane);
  int i0 = (int)floor(cg);
  if (i0 < 1) i0 = 1;
  const int i1 = live ? (int)ceil(cg) : 0;
  int nc = 0;
  for (int c = 0; c < 2; c++) {
    const int gi = i0 + c;
    const bool has = live && gi <= i1;
    const double qcg = gi;
    const int k = has ? pvq_k_noref(qcg, N, beta) : 0;
    int y[NL];
    int npg = 0, npr = 0;
    const double cd = pvq_search_v3<N>(v, g, lane, k, qcg*cg, rsq, y, npg, npr);
    if (aa.stats && has && g == 0) {
      atomicAdd(&aa.stats[0], (unsigned long long)npg*N);
      atomicAdd(&aa.stats[1], (unsigned long long)npr*N);
      atomicAdd(&aa.stats[2], 1ull);
    }
    if (live && g == 0) {
      a.out.qg[c*a.rec_fstride + rec2] = has ? gi : 0;
      a.out.k[c*a.rec_fstride + rec2] = k;
      a.out.cos_dist[c*a.rec_fstride + rec2] = has ? cd : 0;
      a.out.dist[c*a.rec_fstride + rec2] =
          has ? 1.4*(qcg - cg)*(qcg - cg) + qcg*cg*(2 - 2*cd) : 0;
    }
    nc += has;
    // signed pulses -> LDS -> coalesced [cand][block][N] store
#pragma unroll
    for (int j = 0; j < NL; j++) {
      const int jj = g*NL + j;
      if (jj < N) Yst[inst*N + jj] = has ? (((v.neg >> j) & 1) ? -y[j] : y[j]) : 0;
    }
    __syncthreads();
    {
      // [cand][block][N]: every block's N pulses are one contiguous run (whole runs of
      // neighbouring blocks when the order is the raster order)
      int32_t *yo = a.out.y + (size_t)f*a.y_fstride + (size_t)2*nblk*(o0 - 1) + (size_t)c*nblk*N;
      const long lim = (a.blk_end - blk0 < BPW ? a.blk_end - blk0 : BPW)*N;
      for (int e = lane; e < lim; e += 64) {
        const int b = e/N;
        yo[(size_t)Pb[b]*N + (e - b*N)] = Yst[e];
      }
    }
    __syncthreads();
  }
  if (live && g == 0) a.out.ncand[rec] = nc;
  (void)q0;
  (void)beta;
}
