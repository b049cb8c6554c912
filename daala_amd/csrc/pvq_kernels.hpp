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
#define PVQ_K_MAX16 32767                 /* largest K whose pulses fit the 16-bit records */
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

// Device-side result layout of one (plane, level): structure of arrays, band-major, so that
// a wave writes contiguous memory.  All record arrays are [band][block]; every array has its
// own distance between frame slots (fs_*, in elements) because the arrays that cross PCIe
// live in per-slot arenas - one transfer per frame slot and direction:
//   out arena  [slot]{level 0: cos_dist, qg, k, ncand, y; level 1: ...}    device -> host
//   g arena    [slot]{level 0: g; level 1: ...}                            device -> host
//   in arena   [slot]{level 0: cg, perm; level 1: ...}                     host -> device
struct PvqSoA {
  double *cg, *g;           // [nbands*nblk]
  double *cos_dist, *dist;  // [2][nbands*nblk]
  int32_t *qg, *k;          // [2][nbands*nblk]
  int32_t *ncand;           // [nbands*nblk]
  // pulses, int16 (candidates with K > PVQ_K_MAX16 are left to the host): band b occupies y + 2*nblk*yo[b], laid out
  // [cand][block][ns[b]] with ns = the band size rounded up to even (one pad entry for the
  // 15-coefficient band: runs start 4-byte aligned) and yo = 0, off[1], off[2], ...
  int16_t *y;
  // work list of the search kernel (performance only): perm[band*2*nblk + i] = entry
  // 2*block + candidate of the i-th lane slot of that band; written by the host with the
  // companded gains, which fix every K.
  const int32_t *perm;
  size_t fs_cg, fs_g, fs_cd, fs_dist, fs_qg, fs_k, fs_nc, fs_y, fs_perm;
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
  PvqSoA out;              // frame 0 of the launch
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

#ifndef PVQ_G128
#define PVQ_G128 16           /* lanes per 128-coefficient band (round 3, tree combine: 8 -> 0.97 ms per launch, 16 -> 0.90) */
#endif
#ifndef PVQ_G32
#define PVQ_G32 4             /* lanes per 31/32-coefficient band (1: LDS kernel 3.05 ms, 2 -> 2.84, 4 -> 2.30, 8 -> 3.18) */
#endif
#ifndef PVQ_G16
#define PVQ_G16 1             /* lanes per 7..15-coefficient band */
#endif
#ifndef PVQ_G8
#define PVQ_G8 PVQ_G16        /* lanes per 8-coefficient band; 2 measured in round 4: launch alone 0.053 -> 0.060 ms,
                                 overlapped 0.105 -> 0.128 (tree + verification cost more than the shorter scan saves) */
#endif
template <int N>
struct PvqGeom {
  static constexpr int G = N > 32 ? PVQ_G128 : N > 16 ? PVQ_G32 : N == 8 ? PVQ_G8 : PVQ_G16;
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

// 1/sqrt(i) for the RDO scans: the first PVQ_RSQ_L entries of the table live in LDS (one
// copy per workgroup), the rest in global memory, anything beyond PVQ_RSQ_TAB is computed.
#ifndef PVQ_RSQ_L
#define PVQ_RSQ_L(N) ((N) == 128 ? 1024 : (N) == 32 ? 512 : 256)
#endif

// Data-parallel-primitive moves inside a row of 16 lanes (the G <= 16 lanes of a band are
// consecutive and aligned, so a band never straddles a row): lane i reads lane i + S of its
// row (row_shl:S), lanes beyond the row read 0.  VALU rate, no LDS round trip.
template <int S>
__device__ __forceinline__ int pvq_dpp_next_i(int v) {
  static_assert(S >= 1 && S <= 15, "row shift");
  return __builtin_amdgcn_update_dpp(0, v, 0x100 | S, 0xf, 0xf, true);
}
template <int S>
__device__ __forceinline__ double pvq_dpp_next_d(double v) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = pvq_dpp_next_i<S>(u.i[0]);
  u.i[1] = pvq_dpp_next_i<S>(u.i[1]);
  return u.d;
}

// Ordered tree combine of the lanes' RDO-scan winners: plain doubles, a total order, so the tree's winner IS the
// sequential scan's (strictly greater replaces, the lower index stays on ties).
template <int G, int S = 1>
__device__ __forceinline__ void pvq_tree_rdo(double &c, double &t, int &code, int &yw) {
  if constexpr (S < G) {
    const double rc = pvq_dpp_next_d<S>(c), rt = pvq_dpp_next_d<S>(t);
    const int rcode = pvq_dpp_next_i<S>(code), ry = pvq_dpp_next_i<S>(yw);
    if (rc > c) { c = rc; t = rt; code = rcode; yw = ry; }
    pvq_tree_rdo<G, 2*S>(c, t, code, yw);
  }
}

// Ordered tree combine of the lanes' greedy-scan winners (a, b, t, code): after log2(G) steps
// the band's first lane holds the winner of the pairwise rounded compares, the lower lane
// kept on "not strictly greater" (first index wins).  The tree need not agree with the
// sequential scan when rounding makes the compare non-transitive: its winner is verified.
template <int G, int S = 1>
__device__ __forceinline__ void pvq_tree_greedy(double &a, double &b, double &t, int &code) {
  if constexpr (S < G) {
    const double ra = pvq_dpp_next_d<S>(a), rb = pvq_dpp_next_d<S>(b), rt = pvq_dpp_next_d<S>(t);
    const int rc = pvq_dpp_next_i<S>(code);
    if (ra*b > a*rb) { a = ra; b = rb; t = rt; code = rc; }
    pvq_tree_greedy<G, 2*S>(a, b, t, code);
  }
}

// One codeword search (pvq_search_rdo_double, src/pvq_encoder.c:121-225) for the
// band owned by this lane group.  Lanes of bands without a candidate run with k = 0.
// Returns the cosine distance; y = unsigned pulses of this lane's chunk.
//
// Round 3 (v4).  What the round-2 profile showed as "waves parked 64-77 %" was serialised
// memory latency, not arithmetic: the per-element `y > 3` table look-up of the RDO scan sat in
// its own branch (one global load + wait per element and pulse) and the scans carried the
// winner's x and y out through a second select chain.  Now: every 1/sqrt of a pulse is read
// in ONE batch (LDS copy of the table, or one batch of independent global loads beyond it),
// the scans carry the unsquared sum and the new yy of the winner (they ARE the reference's
// next xy and yy: same operands, same operation), and no branch is left inside a scan.
// PAD: the band may be one coefficient short of N (the with-reference searches of pvq_theta
// run on n - 1 dimensions, src/pvq_encoder.c:404,426): with `padlast` the last coefficient of the
// band's last lane is a pad - its |x| is 0, so every sum is unchanged - that no scan may select.
template <int N, bool PAD = false>
__device__ __forceinline__ double pvq_search_v4(const PvqVec<N> &v, int g, int lane, int k,
                                                double g2, const double *__restrict__ rsq,
                                                const double *rsqL,
                                                int (&y)[PvqGeom<N>::NL], int &npulse_greedy,
                                                int &npulse_rdo, bool padlast = false) {
  constexpr int G = PvqGeom<N>::G, NL = PvqGeom<N>::NL;
  static_assert(G*NL == N, "every lane of a band owns NL coefficients (no padding elements in the scans)");
  const int base = (lane/G)*G;                  // first lane of my band
  const bool padl = PAD && padlast && g == G - 1;   // this lane's element NL - 1 is the pad
  const double lambda = PVQ_LAMBDA/(1e-30 + g2);
  const double delta_rate = 3./(N - ((PAD && padlast) ? 1 : 0));   // 3./n of the searched vector
  double xy = 0, yy = 0;
  int i = 0;
  int ymax = 0;                                 // largest pulse count of this lane's chunk
  if (k > 2) {
#pragma unroll
    for (int j = 0; j < NL; j++) {
      int p = (int)floor(k*v.x[j]*v.l1_inv);
      y[j] = p > 0 ? p : 0;
      ymax = y[j] > ymax ? y[j] : ymax;
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
    // local scan of this lane's chunk; bt/bb = the winner's xy + x and yy + 2y + 1
    double bt = xy + v.x[0];
    double bb = yy + (double)(2*y[0] + 1);
    double ba = bt*bt;
    int bpos = 0;
#pragma unroll
    for (int j = 1; j < NL; j++) {
      const double t = xy + v.x[j];
      const double b = yy + (double)(2*y[j] + 1);
      const double a = t*t;
      if (a*bb > ba*b && !(PAD && j == NL - 1 && padl)) { ba = a; bb = b; bt = t; bpos = j; }
    }
    int wl = bpos, wg = 0;            // winner: local index and owning lane of the group
    double nxy = bt, nyy = bb;
    if (G > 1) {
      double ca = ba, cb = bb, ct = bt;
      int code = g*NL + bpos;
      pvq_tree_greedy<G>(ca, cb, ct, code);
      ca = __shfl(ca, base, 64);
      cb = __shfl(cb, base, 64);
      ct = __shfl(ct, base, 64);
      code = __shfl(code, base, 64);
      wg = code/NL;
      wl = code%NL;
      // verification: w = (wg, wl) must strictly beat every earlier element and
      // must not be strictly beaten by any later one
      bool ok = true;
#pragma unroll
      for (int j = 0; j < NL; j++) {
        {
          const double t = xy + v.x[j];
          const double b = yy + (double)(2*y[j] + 1);
          const double a = t*t;
          const double p1 = ca*b, p2 = a*cb;
          const bool before = g < wg || (g == wg && j < wl);
          const bool self = g == wg && j == wl;
          // before: p1 > p2 required; after: !(p2 > p1)
          ok = ok && (self || (PAD && j == NL - 1 && padl) || (before ? p1 > p2 : !(p2 > p1)));
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
              {
                double a = xy + v.x[j];
                const double b = yy + (2*y[j] + 1);
                a *= a;
                if ((r == 0 && j == 0) || (a*ib > ia*b && !(PAD && j == NL - 1 && padl))) { ia = a; ib = b; ip = j; ig = r; }
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
        cb = __shfl(ib, base + G - 1, 64);
        // the sequential winner need not be a local winner: its lane recomputes xy + x
        double tw = 0;
#pragma unroll
        for (int j = 0; j < NL; j++) tw = j == wl ? xy + v.x[j] : tw;
        ct = __shfl(tw, base + wg, 64);
      }
      nxy = ct;
      nyy = cb;
    }
    // apply the pulse
    const int ap = (run && (G == 1 || g == wg)) ? wl : -1;
#pragma unroll
    for (int j = 0; j < NL; j++) y[j] += (j == ap);
    if (run) {
      // the winner now holds (nyy - yy + 1)/2 pulses (an upper bound for every lane of the band)
      const int yn = ((int)(nyy - yy) + 1) >> 1;
      ymax = yn > ymax ? yn : ymax;
      xy = nxy;
      yy = nyy;
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
    const int hi = iy + 2*ymax + 1;           // largest table index this lane asks for
    // keep lambda*j*delta_rate inside the loop: hoisted, it would pin 2 NL registers per lane
    double lam = lambda;
    asm volatile("" : "+v"(lam));
    // The scans carry the winner's table index ix = iy + 2*y + 1 (not y itself: a select chain
    // over y[] is folded into a dynamically indexed read of y, which sends y[] to scratch).
    double bt, bc;
    int bpos, bix;
    if (!__any(hi >= PVQ_RSQ_L(N))) {
      bix = iy + 2*y[0] + 1;
      bt = xy + v.x[0];
      bc = 2*bt*v.norm_1*rsqL[bix] - lam*(g*NL)*delta_rate;
      bpos = 0;
#pragma unroll
      for (int j = 1; j < NL; j++) {
        const int ix = iy + 2*y[j] + 1;
        const double t = xy + v.x[j];
        const double c = 2*t*v.norm_1*rsqL[ix] - lam*(g*NL + j)*delta_rate;
        if (c > bc && !(PAD && j == NL - 1 && padl)) { bc = c; bt = t; bpos = j; bix = ix; }
      }
    }
    else if (N >= 32 && !__any(ymax > 3)) {
      // long bands, beyond the LDS copy but every y <= 3: the reference's own 4-entry table of
      // the pulse (od_fill_dynamic_rqrt_table, src/pvq_encoder.c:205): one batch of four loads
      const double r0 = rsq[iy + 1 < PVQ_RSQ_TAB ? iy + 1 : 0], r1 = rsq[iy + 3 < PVQ_RSQ_TAB ? iy + 3 : 0];
      const double r2 = rsq[iy + 5 < PVQ_RSQ_TAB ? iy + 5 : 0], r3 = rsq[iy + 7 < PVQ_RSQ_TAB ? iy + 7 : 0];
      double q0 = r0, q1 = r1, q2 = r2, q3 = r3;
      if (__any(iy + 7 >= PVQ_RSQ_TAB)) {
        if (iy + 1 >= PVQ_RSQ_TAB) q0 = pvq_rsqrt_slow(iy + 1);
        if (iy + 3 >= PVQ_RSQ_TAB) q1 = pvq_rsqrt_slow(iy + 3);
        if (iy + 5 >= PVQ_RSQ_TAB) q2 = pvq_rsqrt_slow(iy + 5);
        if (iy + 7 >= PVQ_RSQ_TAB) q3 = pvq_rsqrt_slow(iy + 7);
      }
      bt = 0; bc = 0; bpos = 0; bix = 0;
#pragma unroll
      for (int j = 0; j < NL; j++) {
        const int yj = y[j];
        const double rs = yj == 0 ? q0 : yj == 1 ? q1 : yj == 2 ? q2 : q3;
        const double t = xy + v.x[j];
        const double c = 2*t*v.norm_1*rs - lam*(g*NL + j)*delta_rate;
        if (j == 0 || (c > bc && !(PAD && j == NL - 1 && padl))) { bc = c; bt = t; bpos = j; bix = iy + 2*yj + 1; }
      }
    }
    else {
      // batches of independent loads from the global table (it stays in L1/L2), then the
      // rare values beyond the table
      constexpr int CHK = NL < 8 ? NL : 8;
      bt = 0; bc = 0; bpos = 0; bix = 0;
#pragma unroll
      for (int j0 = 0; j0 < NL; j0 += CHK) {
        double rs[CHK];
#pragma unroll
        for (int j = 0; j < CHK; j++) {
          if (j0 + j < NL) {
            const int ix = iy + 2*y[j0 + j] + 1;
            rs[j] = rsq[ix < PVQ_RSQ_TAB ? ix : PVQ_RSQ_TAB - 1];
          }
        }
        if (__any(hi >= PVQ_RSQ_TAB)) {
#pragma unroll
          for (int j = 0; j < CHK; j++) {
            if (j0 + j < NL) {
              const int ix = iy + 2*y[j0 + j] + 1;
              if (ix >= PVQ_RSQ_TAB) rs[j] = pvq_rsqrt_slow(ix);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < CHK; j++) {
          if (j0 + j < NL) {
            int ix = iy + 2*y[j0 + j] + 1;
            asm volatile("" : "+v"(ix));       // opaque: no "bix = table[bpos]" rewrite through scratch
            const double t = xy + v.x[j0 + j];
            const double c = 2*t*v.norm_1*rs[j] - lam*(g*NL + j0 + j)*delta_rate;
            if (j0 + j == 0 || (c > bc && !(PAD && j0 + j == NL - 1 && padl))) { bc = c; bt = t; bpos = j0 + j; bix = ix; }
          }
        }
      }
    }
    int wl = bpos, wg = 0;
    double nxy = bt;
    if (G > 1) {
      double cc = bc, ct = bt;
      int code = g*NL + bpos, cix = bix;
      pvq_tree_rdo<G>(cc, ct, code, cix);
      nxy = __shfl(ct, base, 64);
      code = __shfl(code, base, 64);
      bix = __shfl(cix, base, 64);
      wg = code/NL;
      wl = code%NL;
    }
    const int ap = (run && (G == 1 || g == wg)) ? wl : -1;
#pragma unroll
    for (int j = 0; j < NL; j++) y[j] += (j == ap);
    if (run) {
      // bix - iy = 2*y + 1 of the winner before its new pulse
      const int yn = ((bix - iy) + 1) >> 1;
      ymax = yn > ymax ? yn : ymax;
      xy = nxy;
      yy = yy + (bix - iy);
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
#ifndef PVQ_RANK_MAJOR
#define PVQ_RANK_MAJOR 0      /* 1: list position runs slowest in dispatch order (experiment, measured: see k_pvq_cand) */
#endif
#ifndef PVQ_V4_WAVES
/* min waves/SIMD asked of the register allocator (512/w registers per lane) */
#define PVQ_V4_WAVES(N) 3     /* no spills at 168 registers for every N; 4 and more spill into the scans */
#endif

// LDS hand-over inside ONE wave (these kernels run single-wave workgroups): the LDS
// operations of a wave execute in issue order, so a read issued after a write sees it; only the
// compiler must keep the order, and the counter wait must not cover vector memory - a
// workgroup barrier's fence would drain every load in flight and turn the kernel's start-up
// into a chain of full memory round trips.  lgkmcnt(0); vmcnt / expcnt untouched.
__device__ __forceinline__ void pvq_wave_lds_sync() {
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
}

// Gather of the band of the wave's lane slots (shared by the gain and the search kernel).
// The wave loads cooperatively with lane = coefficient-within-block - consecutive lanes read
// one block's few 64-byte segments - all NL loads of a lane issued back to back (the raster
// offsets of the band come from an LDS copy, not from a dependent global load), stages the
// values in LDS and every lane then picks up its own chunk.  Two memory round trips in all:
// {work list, coding-order table, QM} and {coefficients}; the sampled in-kernel stamps of
// round 3 (profiles/r03_pvq_stamps.txt) showed four, each 2-6 k cycles, against 5-15 k cycles
// of search in a short-band wave.
// pe: this lane's work-list entry as loaded (lane < BPW), or anything for other lanes.
template <int N, typename ENTRY, bool TWO = false>
__device__ __forceinline__ void pvq_gather(const PvqLevelArgs &a, int f, int o0, int lane, int g, int inst,
                                           int nslot_here, ENTRY entry_of_lane, int32_t *Pe, int32_t *Yst,
                                           int32_t *Org, bool &live, long &blk, const double *cg_band,
                                           double &cg, int32_t (&cf)[PvqGeom<N>::NL],
                                           int (&qi)[PvqGeom<N>::NL], const int32_t *plane2 = nullptr,
                                           int32_t *cf2 = nullptr) {
  constexpr int G = PvqGeom<N>::G, NL = PvqGeom<N>::NL, BPW = PvqGeom<N>::BPW;
  constexpr int CH = NL + 1;                       // padded chunk stride in LDS
  constexpr int LDSN = (G*CH) | 1;                 // odd per-band stride
  constexpr int NR = (N + 63)/64;
  __shared__ int32_t Ro[N];                        // raster offset of every coefficient of the band
  const int lg = a.n == 4 ? 2 : a.n == 8 ? 3 : a.n == 16 ? 4 : 5;
  // round trip 1: list entry, table, QM
  const int32_t e_mine = entry_of_lane();
  int rt[NR];
#pragma unroll
  for (int q = 0; q < NR; q++) rt[q] = a.tab[o0 + (lane + 64*q < N ? lane + 64*q : 0)];
#pragma unroll
  for (int j = 0; j < NL; j++) qi[j] = a.qm[o0 + g*NL + j];
  if (lane < BPW) {
    Pe[lane] = e_mine;
    const long bb = e_mine >= 0 ? e_mine >> 1 : 0;   // idle slots read block 0: a valid address
    const int bx = bb%a.nbx, by = bb/a.nbx;
    Org[lane] = (by*a.n)*a.w + bx*a.n;
  }
#pragma unroll
  for (int q = 0; q < NR; q++) {
    if (lane + 64*q < N) Ro[lane + 64*q] = (rt[q] >> lg)*a.w + (rt[q] & (a.n - 1));
  }
  pvq_wave_lds_sync();
  // round trip 2: the coefficients (of both planes when there are two)
  const int32_t *plane = a.lev + (size_t)f*a.lev_fstride;
  const int total = nslot_here*N;
  int32_t val[NL], val2[TWO ? NL : 1];
#pragma unroll
  for (int it = 0; it < NL; it++) {
    const int e = lane + 64*it;
    const int ec = e < total ? e : 0;
    const int b = ec/N, jj = ec - b*N;
    const size_t o = (size_t)Org[b] + Ro[jj];
    val[it] = plane[o];
    if (TWO) val2[it] = plane2[o];
  }
  const int32_t en = Pe[inst];
  live = en >= 0;
  blk = live ? en >> 1 : 0;
  cg = cg_band ? cg_band[blk] : 0;                 // the band's companded gain rides the same round trip
#pragma unroll
  for (int it = 0; it < NL; it++) {
    const int e = lane + 64*it;
    const int b = e/N, jj = e - b*N;
    if (e < total) Yst[b*LDSN + (jj/NL)*CH + jj%NL] = val[it];
  }
  pvq_wave_lds_sync();
#pragma unroll
  for (int j = 0; j < NL; j++) {
    cf[j] = live ? Yst[inst*LDSN + g*CH + j] : 0;
    qi[j] = live ? qi[j] : 0;
  }
  pvq_wave_lds_sync();                             // Yst is reused (second plane, pulses)
  if (TWO) {
#pragma unroll
    for (int it = 0; it < NL; it++) {
      const int e = lane + 64*it;
      const int b = e/N, jj = e - b*N;
      if (e < total) Yst[b*LDSN + (jj/NL)*CH + jj%NL] = val2[it];
    }
    pvq_wave_lds_sync();
#pragma unroll
    for (int j = 0; j < NL; j++) cf2[j] = live ? Yst[inst*LDSN + g*CH + j] : 0;
    pvq_wave_lds_sync();
  }
}

// Two launches per level.  k_pvq_gain computes the exact uncompanded gain g = sqrt(acc) of
// every band (A13 up to the companding); the host then turns g into cg with ITS libm
// (od_gain_compand's pow, src/pvq.c:422 - the one operation on this path that is not
// +,-,*,/,sqrt,floor, and the only one whose result depends on the libm in use) and the
// search launch reads cg back: the device never evaluates a transcendental.
template <int N>
__global__ __launch_bounds__(64) void k_pvq_gain(PvqLevelArgs3 aa) {
  const PvqLevelArgs &a = aa.a;
  constexpr int G = PvqGeom<N>::G, NL = PvqGeom<N>::NL, BPW = PvqGeom<N>::BPW;
  __shared__ int32_t Yst[BPW*(N + G + 2) + 1];
  __shared__ int32_t Org[64];
  __shared__ int32_t Pe[64];
  const int lane = threadIdx.x;
  const int g = lane%G, inst = lane/G;
  const long idx0 = a.blk_first + (long)blockIdx.x*BPW;
  const int band = a.band_list[blockIdx.y], f = blockIdx.z;
  const long nblk = (long)a.nbx*a.nby;
  const int nslot_here = (int)(a.blk_end - idx0 < BPW ? a.blk_end - idx0 : BPW);
  int32_t cf[NL];
  int qi[NL];
  bool live;
  long blk;
  double cg_unused;
  // entries in raster order: "candidate 0" of block idx0 + lane
  pvq_gather<N>(a, f, a.off[band], lane, g, inst, nslot_here,
                [&]() { return idx0 + lane < a.blk_end ? (int32_t)(2*(idx0 + lane)) : -1; },
                Pe, Yst, Org, live, blk, nullptr, cg_unused, cf, qi);
  // od_pvq_compute_gain: five sequential multiplies per term (src/pvq.c:460-463)
  const double acc = pvq_chain_sum<N>(g, lane, [&](int j) {
    return cf[j]*(double)cf[j]*qi[j]*PVQ_QM_SCALE_1*qi[j]*PVQ_QM_SCALE_1;
  });
  if (live && g == 0) {
    const double gain = sqrt(acc);
    a.out.g[(size_t)f*a.out.fs_g + (size_t)band*nblk + blk] = gain;
    // beta == 1: od_gain_compand is g/q0 (src/pvq.c:423), one exactly rounded division - the
    // companded gain needs no libm and no host (the host computes the same quotient for its own
    // use); k_pvq_order then builds the band's work list on the device
    if (a.beta[band] == 1) a.out.cg[(size_t)f*a.out.fs_cg + (size_t)band*nblk + blk] = gain/a.q[band];
  }
}

// Work list of a band whose companded gains were computed on the device (beta == 1): the
// counting sort of pvq_block_order (daala_hip.hip) - entries 2*block + candidate by descending K
// - in two passes of one workgroup per (chunk of blocks, band, frame).  Performance only: any permutation of the entries is a
// correct list, so the order inside a K class (atomic cursors) need not be reproducible.
struct PvqOrderArgs {
  PvqLevelArgs a;
  int nlist;
  int *gh, *gc;          // [frame][band][256]: key histogram, reservation cursors (zeroed before the launch)
};

#define PVQ_ORDER_THREADS 256
#define PVQ_ORDER_CHUNK 4096        /* blocks per workgroup */
// Takes one slot of LDS counter ctr[key] for every active lane and returns the lane's slot.  The
// three most common keys of the wave are served by ONE atomic each (leader adds the population
// count, lanes rank themselves by ballot): in the short-band levels most lanes of a wave share a
// key and a per-lane atomic would serialise on one address.  Whatever is left - the long bands'
// K are spread over the whole range - takes plain per-lane atomics, which only collide by chance.
__device__ __forceinline__ int pvq_order_take(int *ctr, int key, bool active) {
  const int lane = threadIdx.x & 63;
  int slot = 0;
  unsigned long long todo = __ballot(active);
#pragma unroll 1
  for (int it = 0; it < 3 && todo; it++) {
    const int leader = __ffsll((long long)todo) - 1;
    const int k = __shfl(key, leader, 64);
    const unsigned long long same = __ballot(active && key == k) & todo;
    int base = 0;
    if (lane == leader) base = atomicAdd(&ctr[k], __popcll(same));
    base = __shfl(base, leader, 64);
    if ((same >> lane) & 1) slot = base + __popcll(same & ((1ull << lane) - 1));
    todo &= ~same;
  }
  if ((todo >> lane) & 1) slot = atomicAdd(&ctr[key], 1);
  return slot;
}

// sort key of candidate `cand` of a band with companded gain c: 255 - K (descending K), K as
// od_pvq_compute_k (src/pvq.c:508-514) clamped to a byte - the key function of pvq_block_order
// (sqb = sqrt((n + 3)/2)/beta: the list is performance only, so the quotient may be formed once)
__device__ __forceinline__ int pvq_order_key(double c, int cand, int n, double beta, double sqb) {
  const double fl = floor(c);
  const double lo = fl < 1 ? 1 : fl, hi = ceil(c);
  const double q = lo + cand;
  int k = 0;
  if (q <= hi) {
    if (n == 15 && q == 1 && beta > 1.25) k = 1;
    else {
      const double v = floor(.5 + (q - .2)*sqb);
      k = v < 1 ? 1 : v > 255 ? 255 : (int)v;
    }
  }
  return 255 - k;
}

// Pass 1: every workgroup counts the keys of its chunk of blocks (LDS, wave-aggregated) and adds
// them to the (frame, band) histogram gh[(f*nbands + band)*256 + key].
__global__ __launch_bounds__(PVQ_ORDER_THREADS) void k_pvq_order_count(PvqOrderArgs aa) {
  const PvqLevelArgs &a = aa.a;
  __shared__ int hist[256];
  const int t = threadIdx.x;
  const int band = a.band_list[blockIdx.y], f = blockIdx.z;
  const long nblk = (long)a.nbx*a.nby;
  const long first = a.blk_first + (long)blockIdx.x*PVQ_ORDER_CHUNK;
  const long end = first + PVQ_ORDER_CHUNK < a.blk_end ? first + PVQ_ORDER_CHUNK : a.blk_end;
  if (first >= end) return;
  const int n = a.off[band + 1] - a.off[band];
  const double beta = a.beta[band], sq = sqrt((double)((n + 3)/2))/beta;
  const double *cg = a.out.cg + (size_t)f*a.out.fs_cg + (size_t)band*nblk;
  hist[t] = 0;
  // the chunk's gains are loaded in one batch (clamped index: no load behind a branch), then counted
  constexpr int ROUNDS = PVQ_ORDER_CHUNK/PVQ_ORDER_THREADS;
  double cv[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const long i = first + (long)r*PVQ_ORDER_THREADS + t;
    cv[r] = cg[i < end ? i : end - 1];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const bool on = first + (long)r*PVQ_ORDER_THREADS + t < end;
    (void)pvq_order_take(hist, pvq_order_key(cv[r], 0, n, beta, sq), on);
    (void)pvq_order_take(hist, pvq_order_key(cv[r], 1, n, beta, sq), on);
  }
  __syncthreads();
  if (hist[t]) atomicAdd(&aa.gh[((size_t)f*a.nbands + band)*256 + t], hist[t]);
}

// Pass 2: the workgroup turns the (frame, band) histogram into bin bases (exclusive scan),
// reserves its chunk's share of every bin (one atomic per non-empty bin on the cursors gc) and
// scatters its entries.  Which chunk comes first inside a bin is left to the atomics: any
// permutation of a K class is a correct list.
__global__ __launch_bounds__(PVQ_ORDER_THREADS) void k_pvq_order_scatter(PvqOrderArgs aa) {
  const PvqLevelArgs &a = aa.a;
  __shared__ int hist[256];
  __shared__ int base[256];
  const int t = threadIdx.x;
  const int band = a.band_list[blockIdx.y], f = blockIdx.z;
  const long nblk = (long)a.nbx*a.nby;
  const long first = a.blk_first + (long)blockIdx.x*PVQ_ORDER_CHUNK;
  const long end = first + PVQ_ORDER_CHUNK < a.blk_end ? first + PVQ_ORDER_CHUNK : a.blk_end;
  if (first >= end) return;
  const int n = a.off[band + 1] - a.off[band];
  const double beta = a.beta[band], sq = sqrt((double)((n + 3)/2))/beta;
  const double *cg = a.out.cg + (size_t)f*a.out.fs_cg + (size_t)band*nblk;
  int32_t *perm = const_cast<int32_t *>(a.out.perm) + (size_t)f*a.out.fs_perm + (size_t)band*2*nblk + 2*a.blk_first;
  const size_t gb = ((size_t)f*a.nbands + band)*256;
  hist[t] = 0;
  base[t] = aa.gh[gb + t];
  // the chunk's gains in one batch, with the histogram read (clamped index: no load behind a branch)
  constexpr int ROUNDS = PVQ_ORDER_CHUNK/PVQ_ORDER_THREADS;
  double cv[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const long i = first + (long)r*PVQ_ORDER_THREADS + t;
    cv[r] = cg[i < end ? i : end - 1];
  }
  __syncthreads();
  // exclusive scan of the 256 global bins: one wave, 4 bins per lane
  if (t < 64) {
    int v[4], sum = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) { v[q] = base[4*t + q]; sum += v[q]; }
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(inc, o, 64);
      if (t >= o) inc += up;
    }
    int run = inc - sum;
#pragma unroll
    for (int q = 0; q < 4; q++) { base[4*t + q] = run; run += v[q]; }
  }
  // the keys of this thread's entries, two bytes per round, kept for the scatter below
  uint32_t keys[ROUNDS/2];
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const long i = first + (long)r*PVQ_ORDER_THREADS + t;
    const bool on = i < end;
    const double c = cv[r];
    const int k0 = pvq_order_key(c, 0, n, beta, sq), k1 = pvq_order_key(c, 1, n, beta, sq);
    (void)pvq_order_take(hist, k0, on);
    (void)pvq_order_take(hist, k1, on);
    const uint32_t pair = (uint32_t)k0 | (uint32_t)k1 << 8;
    if (r & 1) keys[r >> 1] |= pair << 16;
    else keys[r >> 1] = pair;
  }
  __syncthreads();
  // this chunk's slice of bin t starts at base[t] + (what earlier reservations took)
  {
    const int mine = hist[t];
    int start = base[t];
    if (mine) start += atomicAdd(&aa.gc[gb + t], mine);
    __syncthreads();
    hist[t] = start;               // from here on: the chunk's write cursor of bin t
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const long i = first + (long)r*PVQ_ORDER_THREADS + t;
    const bool on = i < end;
    const uint32_t pair = keys[r >> 1] >> (16*(r & 1));
#pragma unroll
    for (int cand = 0; cand < 2; cand++) {
      const int pos = pvq_order_take(hist, (int)((pair >> (8*cand)) & 255), on);
      if (on) perm[pos] = (int32_t)(2*i + cand);
    }
  }
}


// The searches: one lane group per CANDIDATE (round 2: per band, its two gain candidates one
// after the other).  The host's companding stage knows cg of every band, hence which
// candidates exist and their K; it writes, per band, the work list a.out.perm of entries
// 2*block + candidate ordered by descending K (counting sort; any order is correct).  Lane
// slots of a wave then hold similar K - a wave runs until its slowest slot is done - the long
// searches start first, and candidates that do not exist (K = 0: they only write their zero
// records) gather at the tail.  No-reference candidates of pvq_theta (src/pvq_encoder.c:
// 352-357, :452-481): gain i = max(1, floor(cg)) + candidate while i <= ceil(cg).
// (Measured and dropped, tools/ab_libs.sh on MI355X: waves that walk several batches of the
// list and load batch i + 1 while they search batch i - 0.160 -> 0.185 ms for N = 15, the
// start-up loads are not what these waves wait for; a single-precision candidate scan in
// front of the verification for G > 1 - no gain, the scans are bound by their dependent
// compare -> select chain, not by double-precision issue.)
#ifdef PVQ_STAMPS
/* diagnostic build only (OD_HIP_PVQ_STAMPS=1, od_hip_pvq_stats): cycles of a wave's life per
   phase, kept in registers and added to stats[8 + 8*class + phase] when the wave ends (class:
   N = 15, 8, 32, 128); no output depends on them */
#define PVQ_STAMP(ph) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
  t_acc[ph] = t_ - t_prev; t_prev = t_; } while (0)
#else
#define PVQ_STAMP(ph) do { } while (0)
#endif
template <int N>
__global__ __launch_bounds__(64, PVQ_V4_WAVES(N)) void k_pvq_cand(PvqLevelArgs3 aa) {
  const PvqLevelArgs &a = aa.a;
#ifdef PVQ_STAMPS
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
  unsigned long long t_acc[5] = {0, 0, 0, 0, 0};
#endif
  constexpr int G = PvqGeom<N>::G, NL = PvqGeom<N>::NL, BPW = PvqGeom<N>::BPW;
  __shared__ int32_t Yst[BPW*(N + G + 2) + 1];      // >= BPW*N (output) and >= BPW*LDSN (staging)
  __shared__ int32_t Org[64];
  __shared__ int32_t Pe[64];                        // its work-list entry (2*block + candidate), -1: idle
  // the head of the 1/sqrt table in LDS (one copy per workgroup): an RDO pulse reads NL
  // entries per lane
  __shared__ double RsqL[PVQ_RSQ_L(N)];
  const int lane = threadIdx.x;
  const int g = lane%G, inst = lane/G;
  // Which (list position, band, frame) this workgroup takes.  The hardware hands workgroups out
  // with x running fastest: the K-sorted list of frame 0 is walked from its longest search to its
  // shortest before frame 1's longest search starts, and the last frame's K = 700 waves begin when
  // the launch is almost over.  PVQ_RANK_MAJOR = 1 splits the place in dispatch order so that the
  // list position runs SLOWEST (all bands and frames start their longest candidates first; every
  // (x, y, z) still taken exactly once).  Measured round 4, alternating builds on one box, launches
  // alone on the chip: <128> 0.812 -> 0.787 ms, <32> 0.219 -> 0.215, <15> 0.142 -> 0.139, <8> 0.053 ->
  // 0.058; the overlapped phase 5.87 -> 6.02 ms.  The launches are bound by the SUM of wave time,
  // not by their last waves (which is also why candidates above a K threshold were not given 64
  // lanes: that shortens a wave at more lane time).  Off.
  unsigned wx = blockIdx.x, wy = blockIdx.y, wz = blockIdx.z;
#if PVQ_RANK_MAJOR
  {
    const unsigned gyz = gridDim.y*gridDim.z;
    const unsigned lin = blockIdx.x + gridDim.x*(blockIdx.y + gridDim.y*blockIdx.z);
    const unsigned r = lin%gyz;
    wx = lin/gyz;
    wy = r%gridDim.y;
    wz = r/gridDim.y;
  }
#endif
  const long idx0 = 2*a.blk_first + (long)wx*BPW;              // position in the band's work list
  const long idx_end = 2*a.blk_end;
  const int band = a.band_list[wy], f = (int)wz;
  const long nblk = (long)a.nbx*a.nby;
  const int32_t *list = a.out.perm ? a.out.perm + (size_t)f*a.out.fs_perm + (size_t)band*2*nblk : nullptr;
  const int nslot_here = (int)(idx_end - idx0 < BPW ? idx_end - idx0 : BPW);
  const int o0 = a.off[band];
  const double beta = a.beta[band];
  // the table head travels with the first round trip of the gather
  constexpr int NQ = PVQ_RSQ_L(N)/64;
  double rq[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) rq[q] = aa.rsq[lane + 64*q];
  PvqVec<N> v;
  bool live;
  long blk;
  double cg;                                       // companded on the host from out.g
  {
    int32_t cf[NL];
    int qi[NL];
    pvq_gather<N>(a, f, o0, lane, g, inst, nslot_here,
                  [&]() {
                    const long i = idx0 + lane;
                    int32_t e = (lane < BPW && i < idx_end) ? (list ? list[i] : (int32_t)i) : -1;
                    if (e < 0 || (e >> 1) >= nblk) e = -1;   // a list that was never written: idle slot
                    return e;
                  },
                  Pe, Yst, Org, live, blk, a.out.cg + (size_t)f*a.out.fs_cg + (size_t)band*nblk, cg, cf, qi);
    v.neg = 0;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      const int pr = cf[j]*qi[j];                  // the reference's int*int product (:455)
      v.x[j] = fabs(pr*PVQ_QM_SCALE_1);
      v.neg |= (uint32_t)(pr < 0) << j;
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; q++) RsqL[lane + 64*q] = rq[q];
  pvq_wave_lds_sync();
  PVQ_STAMP(1);                                    // gather
  const int c = live ? Pe[inst] & 1 : 0;
  const size_t rin = (size_t)band*nblk + blk;       // record of the band
  const size_t rin2 = (size_t)c*a.nbands*nblk + rin;   // record of the candidate
  pvq_vec_finish<N>(v, g, lane);
  int i0 = (int)floor(cg);
  if (i0 < 1) i0 = 1;
  const int i1 = (int)ceil(cg);
  const int gi = i0 + c;
  const double qcg = gi;
  // pulses travel as int16: a candidate whose K does not fit (quantizer 1-2 on noise-like
  // content) is left to the host - its qg stays 0, which the consumer reads as "not fed"
  const int kraw = live && gi <= i1 ? pvq_k_noref(qcg, N, beta) : 0;
  const bool has = live && gi <= i1 && kraw <= PVQ_K_MAX16;
  const int k = has ? kraw : 0;
  int y[NL];
  int npg = 0, npr = 0;
  PVQ_STAMP(2);                                    // norms + companded gain
  const double cd = pvq_search_v4<N>(v, g, lane, k, qcg*cg, aa.rsq, RsqL, y, npg, npr);
  PVQ_STAMP(3);                                    // the search
#ifndef PVQ_STAMPS               /* the diagnostic build times the phases without this traffic */
  if (aa.stats && has && g == 0) {
    atomicAdd(&aa.stats[0], (unsigned long long)npg*N);
    atomicAdd(&aa.stats[1], (unsigned long long)npr*N);
    atomicAdd(&aa.stats[2], 1ull);
  }
#endif
  if (live && g == 0) {
    a.out.qg[(size_t)f*a.out.fs_qg + rin2] = has ? gi : 0;
    a.out.k[(size_t)f*a.out.fs_k + rin2] = k;
    a.out.cos_dist[(size_t)f*a.out.fs_cd + rin2] = has ? cd : 0;
    a.out.dist[(size_t)f*a.out.fs_dist + rin2] = has ? 1.4*(qcg - cg)*(qcg - cg) + qcg*cg*(2 - 2*cd) : 0;
    if (c == 0) a.out.ncand[(size_t)f*a.out.fs_nc + rin] = (i0 <= i1) + (i0 + 1 <= i1);
  }
  // signed pulses -> LDS (int16 pairs) -> coalesced [cand][block][NS] store, one dword = two
  // pulses: every candidate's run is contiguous and 4-byte aligned
  constexpr int NS = (N + 1) & ~1, NW = NS/2;
  int16_t *Y16 = reinterpret_cast<int16_t *>(Yst);
#pragma unroll
  for (int j = 0; j < NL; j++) Y16[inst*NS + g*NL + j] = (int16_t)(has ? (((v.neg >> j) & 1) ? -y[j] : y[j]) : 0);
  if (NS != N && g == G - 1) Y16[inst*NS + N] = 0;
  pvq_wave_lds_sync();
  {
    const int yo_band = o0 == 1 ? 0 : o0;
    uint32_t *yo = reinterpret_cast<uint32_t *>(a.out.y + (size_t)f*a.out.fs_y + (size_t)2*nblk*yo_band);
    const uint32_t *Yw = reinterpret_cast<const uint32_t *>(Yst);
    const int lim = nslot_here*NW;
#pragma unroll
    for (int it = 0; it < (BPW*NW + 63)/64; it++) {
      const int e = lane + 64*it;
      const int b = e/NW;
      const int32_t en = e < lim ? Pe[b] : -1;
      if (en >= 0) yo[((size_t)(en & 1)*nblk + (en >> 1))*NW + (e - b*NW)] = Yw[e];
    }
  }
  PVQ_STAMP(4);                                    // records + pulses issued
#ifdef PVQ_STAMPS
  if (aa.stats && threadIdx.x == 0 && (blockIdx.x & 63) == 0) {      // one wave in 64: no hot-address traffic
    unsigned long long *st = aa.stats + 8 + 8*(N == 15 ? 0 : N == 8 ? 1 : N == 32 ? 2 : 3);
    for (int q = 0; q < 5; q++) atomicAdd(&st[q], t_acc[q]);
    atomicAdd(&st[7], 1ull);
  }
#endif
}
