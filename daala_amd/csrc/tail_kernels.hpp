// Decoder pixel-domain tail: directional deringing and keyframe
// bilinear smoothing between the frame post-filter and the 8-bit clamp
// (reference src/decode.c:1040-1155, src/filter.c:1655-2040).
//
// One WAVE per 32x32 deringing superblock (round 4; rounds 1-3: one workgroup of four waves
// with ~30 workgroup barriers per superblock, waves parked half of their time), all planes
// of the superblock in turn (chroma reuses the directions found on luma).  A single-wave
// workgroup never waits at an s_barrier: its LDS hand-overs only drain its own LDS queue.
// Deringing reads only the unfiltered post-filter output (the reference's etmp copy,
// :1046-1056) and every superblock writes its own region, so superblocks are independent;
// smoothing works inside one 32x32 block.  Dering + smoothing + clamp are fused: the tile is
// read once (int32) and written once (u8); the filtered samples stay in registers between
// the orthogonal filter, the smoothing and the clamp.
#pragma once
#include "xform_kernels.hpp"

#define TAIL_BSTRIDE 38                    /* OD_FILT_BSTRIDE */
#define TAIL_VERY_LARGE 30000              /* OD_DERING_VERY_LARGE */

struct TailArgs {
  const int32_t *p[3];       // post-filtered planes (frame 0)
  uint8_t *rec[3];           // 8-bit output planes (frame 0)
  size_t fstride[3];         // samples per plane
  const uint8_t *flags;      // [frame][nvsb*nhsb]
  const uint8_t *bskip[3];   // [frame][(fh/4)*(fw/4)], row stride fw/4
  size_t bskip_fstride;
  const uint8_t *bsize;      // [frame][nvsb*4][nhsb*4]
  size_t bsize_fstride;
  int bstride;
  int fw, fh, nhsb, nvsb, nplanes;
  int xdec[3];
  int thr[3];                // (int)pow(quantizer, 0.84182)  (src/filter.c:1878)
  int q[3];
  int is_keyframe;
  // encoder mode (od_hip_dering_run): every superblock is deringed (flags == NULL) and the
  // result is written as int16 planes without smoothing or clamping - exactly what
  // od_dering() hands back (src/filter.c:1835), for every superblock of the frame at once
  int16_t *o16[3];
  const int16_t *p16[3];     // encoder mode input: the reference's int16 etmp planes
};

__constant__ int8_t TAIL_DIR[8][3][2] = {
  {{-1, 1}, {-2, 2}, {-3, 3}}, {{0, 1}, {-1, 2}, {-1, 3}}, {{0, 1}, {0, 2}, {0, 3}},
  {{0, 1}, {1, 2}, {1, 3}}, {{1, 1}, {2, 2}, {3, 3}}, {{1, 0}, {2, 1}, {3, 1}},
  {{1, 0}, {2, 0}, {3, 0}}, {{1, 0}, {2, -1}, {3, -1}}};

__constant__ int16_t TAIL_THRESH_Q8[18] = {128, 134, 150, 168, 188, 210, 234, 262, 292,
  327, 365, 408, 455, 509, 569, 635, 710, 768};

// Sums over aligned groups of 8 / 16 / 32 neighbouring lanes without the LDS pipe: quad
// permutes (xor 1, xor 2), row_half_mirror (lane i <-> 7 - i: the other quad of the eight),
// row_mirror (i <-> 15 - i: the other eight of the row); only the last step of a 32-lane sum
// crosses rows (one ds_bpermute).  Every lane of the group ends up with the group's sum.
__device__ __forceinline__ int tail_sum8(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);    // row_half_mirror
  return v;
}
template <int N>
__device__ __forceinline__ int tail_sum_group(int v) {
  v = tail_sum8(v);
  if (N >= 16) v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);   // row_mirror
  if (N >= 32) v += __shfl_xor(v, 16);
  return v;
}

#define TAIL_PSTRIDE 72                    /* line sums of one block: 4 directions x 16 lines, padded
                                              so that the 8 blocks of a wave instruction spread over the banks */

// od_dir_find8 (src/filter.c:1655-1708) for the 16 8x8 blocks of a 32x32 luma tile by one
// wave.  The reference accumulates, per block and direction d, the sums of the pixels along
// the <= 15 lines of that direction (partial[d][line]); integer sums are associative, so a
// lane owns one row of one block (8 pixels, read once, kept in registers for all eight
// directions) and adds it into LDS counters; the direction is uniform across the wave (no
// divergence).  Four directions at a time fit the scratch area: D0 = 0 or 4.
template <int D0>
__device__ __forceinline__ void tail_dir_accumulate(const int (&x)[8], int32_t *p, int i) {
#pragma unroll
  for (int dd = 0; dd < 4; dd++) {
    const int d = D0 + dd;
    int32_t *q = p + dd*16;
    if (d == 0) {
#pragma unroll
      for (int j = 0; j < 8; j++) atomicAdd(&q[i + j], x[j]);
    }
    else if (d == 1) {
#pragma unroll
      for (int j = 0; j < 4; j++) atomicAdd(&q[i + j], x[2*j] + x[2*j + 1]);
    }
    else if (d == 2) q[i] = x[0] + x[1] + x[2] + x[3] + x[4] + x[5] + x[6] + x[7];   // a line per row: no other writer
    else if (d == 3) {
#pragma unroll
      for (int j = 0; j < 4; j++) atomicAdd(&q[3 + i - j], x[2*j] + x[2*j + 1]);
    }
    else if (d == 4) {
#pragma unroll
      for (int j = 0; j < 8; j++) atomicAdd(&q[7 + i - j], x[j]);
    }
    else if (d == 5) {
#pragma unroll
      for (int j = 0; j < 8; j++) atomicAdd(&q[3 - i/2 + j], x[j]);
    }
    else if (d == 6) {
      // line j = column j: the 8 rows of a block sit in 8 neighbouring lanes - summed
      // there (three DPP steps per column), one lane writes
      int c[8];
#pragma unroll
      for (int j = 0; j < 8; j++) c[j] = tail_sum8(x[j]);
      if (i == 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) q[j] = c[j];
      }
    }
    else {
#pragma unroll
      for (int j = 0; j < 8; j++) atomicAdd(&q[i/2 + j], x[j]);
    }
  }
}

// cost of one (block, direction) from its line sums (src/filter.c:1676-1697)
__device__ __forceinline__ int tail_dir_cost(const int32_t *p, int d) {
  int cost = 0;
  if (d == 2 || d == 6) {
#pragma unroll
    for (int i = 0; i < 8; i++) cost += p[i]*p[i] >> 3;
  }
  else if (d == 0 || d == 4) {
#pragma unroll
    for (int i = 0; i < 7; i++) {
      cost += (int)((uint32_t)(p[i]*p[i])/(uint32_t)(i + 1))
              + (int)((uint32_t)(p[14 - i]*p[14 - i])/(uint32_t)(i + 1));
    }
    cost += p[7]*p[7] >> 3;
  }
  else {
#pragma unroll
    for (int j = 0; j < 5; j++) cost += p[3 + j]*p[3 + j] >> 3;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      cost += (int)((uint32_t)(p[j]*p[j])/(uint32_t)(2*j + 2))
              + (int)((uint32_t)(p[10 - j]*p[10 - j])/(uint32_t)(2*j + 2));
    }
  }
  return cost;
}

#ifdef TAIL_STAMPS
/* diagnostic build only (make FLAGS+=-DTAIL_STAMPS, tools/tail_stamps.py): cycles of a wave's
   life per phase of the luma plane, summed over one wave in 16; no output depends on them */
__device__ unsigned long long g_tail_stamps[16];
#define TAIL_STAMP(ph) do { if (pli == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
  t_acc[ph] += t_ - t_prev; t_prev = t_; } } while (0)
#else
#define TAIL_STAMP(ph) do { } while (0)
#endif

struct TailShared {
  int16_t in0[TAIL_BSTRIDE*TAIL_BSTRIDE];   // unfiltered tile + border
  // Three tenants, one after the other: the line sums of the direction search (luma only, four
  // directions at a time), then the filtered tile (border + direction-filtered interior), then
  // the packed output bytes.  Sharing the space takes the workgroup from 11.6 to 8.7 KB of LDS:
  // 18 instead of 13 single-wave workgroups per CU.
  union alignas(16) {
    int16_t in1[TAIL_BSTRIDE*TAIL_BSTRIDE];
    int32_t scratch[16*TAIL_PSTRIDE];
  };
  int32_t cost[128];                        // [block][direction]
  int dirs[16], vars[16], thresh[16];
  int doff[16*3];                           // tap offsets of each block's direction
  int16_t tab[64];                          // [0, 24): TAIL_DIR as tile offsets (dy*TAIL_BSTRIDE + dx), [direction][tap];
                                            // [32, 50): TAIL_THRESH_Q8 - one entry per lane, one store
};

// A plane's (n + 6)^2 tile on its way from HBM to LDS, held in registers between the ISSUE of
// its loads and their COMMIT to LDS: the kernel issues the loads of all three planes of a
// 4:2:0 superblock up front, so the chroma tiles (and the skip maps) travel while the luma
// plane is being filtered - one memory round trip per superblock instead of one per plane
// plus one per skip test (stamps of the first single-wave version, profiles/r04_tail_stamps_*:
// a third of a wave's life went into waiting for tile loads).
template <int LN>
struct TailTile {
  static constexpr int n = 1 << LN, tw = n + 6;
  static constexpr int LPR = n/4, RPP = 64/LPR;            // interior: lanes per row, rows per pass
  static constexpr int NPASS = (tw + RPP - 1)/RPP, NB = (6*tw + 63)/64;
  int4 q[NPASS];            // four interior samples per pass (int16 input: two dwords used)
  int b[NB];                // border columns, one sample per pass (raw: the commit converts)
  int sk[16];               // the skip flags of a block's 4x4 neighbourhood (lanes & 15 = block)
};

__device__ __forceinline__ int tail_clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

// ISSUE: nothing but address arithmetic and loads - no load sits behind a lane-dependent branch
// and no loaded value is touched, so the whole tile (interior passes, border columns, skip flags)
// leaves in one batch.  Rows, columns and flags outside the frame or the neighbourhood are read
// from the nearest valid position instead and discarded at the commit, which repeats the test.
// (The first form had each load in its own `if`, its value converted right there: the compiler
// waited for every one of them in turn - 25 memory round trips for a luma tile.)
template <int LN>
__device__ __forceinline__ void tail_tile_issue(TailTile<LN> &T, const TailArgs &a, int pli, int sbx, int sby,
                                                int f, bool enc_mode, int lane) {
  using TT = TailTile<LN>;
  constexpr int n = TT::n, tw = TT::tw, xdec = 5 - LN;
  const int w = a.fw >> xdec;
  const size_t porg = (size_t)f*a.fstride[pli] + (size_t)(sby << LN)*w + (sbx << LN);
  // tile with a 3-sample border; outside the frame: OD_DERING_VERY_LARGE (filled at commit)
  const int lo_i = -3*(sby != 0), hi_i = n + 3*(sby != a.nvsb - 1);
  const int lo_j = -3*(sbx != 0), hi_j = n + 3*(sbx != a.nhsb - 1);
  // interior columns 0 .. n - 1 of rows -3 .. n + 2: always inside the frame horizontally and
  // aligned - four samples per lane and load
  const int c4 = (lane % TT::LPR)*4, r0 = lane/TT::LPR;
  int bofs[TT::NB];
#pragma unroll
  for (int it = 0; it < TT::NB; it++) {
    const int e = min(lane + 64*it, 6*tw - 1);
    const int r = e/6, c6 = e - 6*r;
    const int ii = tail_clampi(r - 3, lo_i, hi_i - 1);
    const int jj = tail_clampi(c6 < 3 ? c6 - 3 : n + c6 - 3, lo_j, hi_j - 1);
    bofs[it] = ii*w + jj;
  }
  if (enc_mode) {
    const int16_t *P16 = a.p16[pli] + porg;
#pragma unroll
    for (int pass = 0; pass < TT::NPASS; pass++) {
      const int ii = tail_clampi(pass*TT::RPP + r0 - 3, lo_i, hi_i - 1);
      const int2 u = *reinterpret_cast<const int2 *>(P16 + (ptrdiff_t)ii*w + c4);
      T.q[pass] = make_int4(u.x, u.y, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < TT::NB; it++) T.b[it] = P16[bofs[it]];
  }
  else {
    const int32_t *P = a.p[pli] + porg;
#pragma unroll
    for (int pass = 0; pass < TT::NPASS; pass++) {
      const int ii = tail_clampi(pass*TT::RPP + r0 - 3, lo_i, hi_i - 1);
      T.q[pass] = *reinterpret_cast<const int4 *>(P + (ptrdiff_t)ii*w + c4);
    }
#pragma unroll
    for (int it = 0; it < TT::NB; it++) T.b[it] = P[bofs[it]];
  }
  // skipped neighbourhood => no filtering (src/filter.c:1898-1917): every flag of the block's
  // neighbourhood is read (no early exit: the loads leave together)
  {
    const int by = (lane >> 2) & 3, bx = lane & 3;
    const int sstride = a.fw/4;
    const uint8_t *bs = a.bskip[pli] + (size_t)f*a.bskip_fstride +
                        (size_t)(sby << (3 - xdec))*sstride + (sbx << (3 - xdec));
    const int xstart = sbx == 0 ? 0 : -1, ystart = sby == 0 ? 0 : -1;
    const int xend = (2 >> xdec) + (sbx != a.nhsb - 1), yend = (2 >> xdec) + (sby != a.nvsb - 1);
#pragma unroll
    for (int ii = -1; ii < 3; ii++) {
#pragma unroll
      for (int jj = -1; jj < 3; jj++) {
        const int ci = tail_clampi(ii, ystart, yend - 1), cj = tail_clampi(jj, xstart, xend - 1);
        T.sk[(ii + 1)*4 + jj + 1] = bs[(ptrdiff_t)((by << 1 >> xdec) + ci)*sstride + (bx << 1 >> xdec) + cj];
      }
    }
  }
}

// the block's whole neighbourhood was skipped (lanes 0 .. 15 ask): flags outside the
// neighbourhood - the issue read a neighbour of theirs - do not count
template <int LN>
__device__ __forceinline__ int tail_tile_skip(const TailTile<LN> &T, const TailArgs &a, int sbx, int sby) {
  constexpr int xdec = 5 - LN;
  const int xstart = sbx == 0 ? 0 : -1, ystart = sby == 0 ? 0 : -1;
  const int xend = (2 >> xdec) + (sbx != a.nhsb - 1), yend = (2 >> xdec) + (sby != a.nvsb - 1);
  int all = 1;
#pragma unroll
  for (int ii = -1; ii < 3; ii++) {
#pragma unroll
    for (int jj = -1; jj < 3; jj++) {
      if (ii >= ystart && ii < yend && jj >= xstart && jj < xend) all &= T.sk[(ii + 1)*4 + jj + 1] != 0;
    }
  }
  return all;
}

// IN1: the filtered tile's border is written too (not for luma, whose direction search uses
// that space first: tail_border_copy fills it afterwards)
template <int LN, bool IN1>
__device__ __forceinline__ void tail_tile_commit(const TailTile<LN> &T, const TailArgs &a, TailShared &S, int sbx,
                                                 int sby, bool enc_mode, int lane) {
  using TT = TailTile<LN>;
  constexpr int n = TT::n, tw = TT::tw;
  const int lo_i = -3*(sby != 0), hi_i = n + 3*(sby != a.nvsb - 1);
  const int c4 = (lane % TT::LPR)*4, r0 = lane/TT::LPR;
#pragma unroll
  for (int pass = 0; pass < TT::NPASS; pass++) {
    const int r = pass*TT::RPP + r0;
    if (r < tw) {
      const int ii = r - 3;
      int v0 = TAIL_VERY_LARGE, v1 = TAIL_VERY_LARGE, v2 = TAIL_VERY_LARGE, v3 = TAIL_VERY_LARGE;
      if (ii >= lo_i && ii < hi_i) {
        if (enc_mode) {
          v0 = (int16_t)(T.q[pass].x & 0xffff); v1 = T.q[pass].x >> 16;
          v2 = (int16_t)(T.q[pass].y & 0xffff); v3 = T.q[pass].y >> 16;
        }
        else {
          v0 = (int16_t)T.q[pass].x; v1 = (int16_t)T.q[pass].y; v2 = (int16_t)T.q[pass].z; v3 = (int16_t)T.q[pass].w;
        }
      }
      int16_t *d0 = S.in0 + r*TAIL_BSTRIDE + c4 + 3;
      d0[0] = (int16_t)v0; d0[1] = (int16_t)v1; d0[2] = (int16_t)v2; d0[3] = (int16_t)v3;
      if (IN1 && (unsigned)ii >= (unsigned)n) {        // a border row: the filtered tile keeps it
        int16_t *d1 = S.in1 + r*TAIL_BSTRIDE + c4 + 3;
        d1[0] = (int16_t)v0; d1[1] = (int16_t)v1; d1[2] = (int16_t)v2; d1[3] = (int16_t)v3;
      }
    }
  }
  const int lo_j = -3*(sbx != 0), hi_j = n + 3*(sbx != a.nhsb - 1);
#pragma unroll
  for (int it = 0; it < TT::NB; it++) {
    const int e = lane + 64*it;
    const int r = e/6, c6 = e - 6*r;
    const int ii = r - 3, jj = c6 < 3 ? c6 - 3 : n + c6 - 3;
    if (e < 6*tw) {
      const bool in = ii >= lo_i && ii < hi_i && jj >= lo_j && jj < hi_j;
      const int16_t v = in ? (int16_t)T.b[it] : (int16_t)TAIL_VERY_LARGE;
      S.in0[r*TAIL_BSTRIDE + jj + 3] = v;
      if (IN1) S.in1[r*TAIL_BSTRIDE + jj + 3] = v;
    }
  }
}

// What the orthogonal filter reads beyond the interior of the filtered tile - two samples
// straight above, below, left and right of it, never a corner (src/filter.c:1753-1793: taps at
// +-1 and +-2 along a row or a column) - copied from the unfiltered tile, LDS to LDS.
template <int LN>
__device__ __forceinline__ void tail_border_copy(TailShared &S, int lane) {
  constexpr int n = 1 << LN;
#pragma unroll
  for (int it = 0; it < 4*n/64; it++) {
    const int e = lane + 64*it;
    const int k = e >> LN, c = e & (n - 1);            // k: which of the four lines
    const int off = k < 2 ? k + 1 : n + k + 1;         // tile rows / columns 1, 2, n + 3, n + 4
    S.in1[off*TAIL_BSTRIDE + c + 3] = S.in0[off*TAIL_BSTRIDE + c + 3];
    S.in1[(c + 3)*TAIL_BSTRIDE + off] = S.in0[(c + 3)*TAIL_BSTRIDE + off];
  }
}

// p when -t < p < t, else 0; tb = t - 1, tl = 2 t - 1 (0 when t = 0).  |p| <= 65535 and t is a
// filter threshold (< 2^16): no wrap.
__device__ __forceinline__ int tail_in_range(int p, int tb, unsigned tl) {
  return (unsigned)(p + tb) < tl ? p : 0;
}

// One plane of one superblock.  LN = 5: 32x32 samples (luma, 4:4:4 chroma), LN = 4: 16x16.
// A lane owns the samples e = lane + 64 r: column j = lane & (n - 1), rows (64 >> LN) apart.
template <int LN, bool PRE>
__device__ __forceinline__ void tail_plane(const TailArgs &a, TailShared &S, int pli, int sbx, int sby, int f,
                                           bool enc_mode, bool dering_on, bool smooth_on, int lane,
                                           TailTile<LN> &T) {
  constexpr int n = 1 << LN, NPX = n*n/64, RSTEP = 64 >> LN;   // rows covered by one wave pass
  constexpr int xdec = 5 - LN, bsz = 3 - xdec;
  const int w = a.fw >> xdec;
  const size_t porg = (size_t)f*a.fstride[pli] + (size_t)(sby << LN)*w + (sbx << LN);
  const int32_t *P = enc_mode ? nullptr : a.p[pli] + porg;
  const int16_t *P16 = enc_mode ? a.p16[pli] + porg : nullptr;
  const int j = lane & (n - 1), i0 = lane >> LN;
  int o[NPX];
#ifdef TAIL_STAMPS
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
  unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  if (dering_on) {
    if (!PRE) tail_tile_issue<LN>(T, a, pli, sbx, sby, f, enc_mode, lane);
    if (pli == 0) tail_tile_commit<LN, false>(T, a, S, sbx, sby, enc_mode, lane);
    else tail_tile_commit<LN, true>(T, a, S, sbx, sby, enc_mode, lane);
    __syncthreads();
    TAIL_STAMP(0);                                   // tile in LDS
    const int16_t *in = S.in0 + 3*TAIL_BSTRIDE + 3;
    if (pli == 0) {
      // two rows of the 16 blocks' 128 per lane: block (lane >> 3) + 8 q, row lane & 7
      int x[2][8];
      const int bi = lane & 7;
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int blk = (lane >> 3) + 8*q;
        const int16_t *row = in + (8*(blk >> 2) + bi)*TAIL_BSTRIDE + 8*(blk & 3);
#pragma unroll
        for (int jj = 0; jj < 8; jj++) x[q][jj] = row[jj] >> 4;
      }
#pragma unroll
      for (int h = 0; h < 2; h++) {
        for (int e = lane; e < 16*TAIL_PSTRIDE/4; e += 64) reinterpret_cast<int4 *>(S.scratch)[e] = make_int4(0, 0, 0, 0);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; q++) {
          int32_t *p = S.scratch + ((lane >> 3) + 8*q)*TAIL_PSTRIDE;
          if (h == 0) tail_dir_accumulate<0>(x[q], p, bi);
          else tail_dir_accumulate<4>(x[q], p, bi);
        }
        __syncthreads();
        {
          // 64 (block, direction) costs of this half, one per lane
          const int blk = lane >> 2, dd = lane & 3;
          S.cost[blk*8 + 4*h + dd] = tail_dir_cost(S.scratch + blk*TAIL_PSTRIDE + dd*16, dd);   // the cost's form repeats with period 4
        }
        __syncthreads();
      }
      if (lane < 16) {
        // first direction with the largest cost (strict '>' scan from best_cost = 0)
        int best_cost = 0, best_dir = 0;
#pragma unroll
        for (int d = 0; d < 8; d++) {
          const int c = S.cost[lane*8 + d];
          if (c > best_cost) { best_cost = c; best_dir = d; }
        }
        S.dirs[lane] = best_dir;
        S.vars[lane] = best_cost - S.cost[lane*8 + ((best_dir + 4) & 7)];
      }
      __syncthreads();
      // the three tap offsets of every block's direction, once per superblock (chroma
      // reuses them): keeps the per-pixel loop free of constant-memory gathers
      if (lane < 48) {
        const int blk = lane/3, k = lane - 3*blk;
        S.doff[lane] = S.tab[S.dirs[blk]*3 + k];
      }
      if (lane < 16) {
        int varsum = 0;
        for (int k = 0; k < 16; k++) varsum += S.vars[k];
        int v1 = S.vars[lane] >> 6, v2 = varsum/1024;
        v1 = v1 > 32767 ? 32767 : v1;
        v2 = v2 > 32767 ? 32767 : v2;
        const uint32_t pr = (uint32_t)(v1*v2);
        int il = pr ? 32 - __clz(pr) : 0;
        il = il - 9;
        il = il < 0 ? 0 : il > 17 ? 17 : il;
        S.thresh[lane] = a.thr[0]*S.tab[32 + il] >> 8;
      }
      tail_border_copy<LN>(S, lane);       // the line sums are done with: the space is the filtered tile now
    }
    else if (lane < 16) S.thresh[lane] = a.thr[pli];
    TAIL_STAMP(1);                                   // directions
    if (lane < 16 && tail_tile_skip<LN>(T, a, sbx, sby)) S.thresh[lane] = 0;     // a lane only rewrites its own entry
    __syncthreads();
    TAIL_STAMP(2);                                   // thresholds, skip test
    // direction filter (src/filter.c:1714-1740); a lane stays in one block column, and in one
    // block for n/8 consecutive passes: its parameters are read once per block row
    constexpr int PER_BROW = NPX/4;          // passes per block row
    const int bcol = j >> bsz;
#pragma unroll
    for (int rb = 0; rb < 4; rb++) {
      const int blk = rb*4 + bcol;
      const int th = S.thresh[blk];
      const int tb = th - 1;
      const unsigned tl = th > 0 ? 2u*(unsigned)th - 1u : 0u;
      const int off0 = S.doff[blk*3], off1 = S.doff[blk*3 + 1], off2 = S.doff[blk*3 + 2];
#pragma unroll
      for (int k = 0; k < PER_BROW; k++) {
        const int r = rb*PER_BROW + k;
        const int i = r*RSTEP + i0;
        const int16_t *c = in + i*TAIL_BSTRIDE + j;
        const int xx = c[0];
        // "if (abs(p) < th) sum += tap*p" (src/filter.c:1731-1736) as one unsigned range test:
        // -th < p < th  <=>  (unsigned)(p + th - 1) < 2 th - 1 (th = 0: the bound is 0, never true)
        const int a0 = tail_in_range(c[off0] - xx, tb, tl) + tail_in_range(c[-off0] - xx, tb, tl);
        const int a1 = tail_in_range(c[off1] - xx, tb, tl) + tail_in_range(c[-off1] - xx, tb, tl)
                     + tail_in_range(c[off2] - xx, tb, tl) + tail_in_range(c[-off2] - xx, tb, tl);
        const int sum = 3*a0 + 2*a1;
        S.in1[(i + 3)*TAIL_BSTRIDE + j + 3] = (int16_t)(xx + ((sum + 8) >> 4));
      }
    }
    __syncthreads();
    TAIL_STAMP(3);                                   // direction filter
    // orthogonal filter (src/filter.c:1753-1793)
    const int16_t *inf = S.in1 + 3*TAIL_BSTRIDE + 3;
#pragma unroll
    for (int rb = 0; rb < 4; rb++) {
      const int blk = rb*4 + bcol;
      const int th = S.thresh[blk], th3 = th/3;
      const int offset = S.dirs[blk] <= 4 ? TAIL_BSTRIDE : 1;
#pragma unroll
      for (int k = 0; k < PER_BROW; k++) {
        const int r = rb*PER_BROW + k;
        const int i = r*RSTEP + i0;
        const int16_t *c = inf + i*TAIL_BSTRIDE + j;
        const int yy = c[0];
        int athresh = th3 + abs(yy - in[i*TAIL_BSTRIDE + j]);
        athresh = th < athresh ? th : athresh;
        const int ab = athresh - 1;
        const unsigned al = athresh > 0 ? 2u*(unsigned)athresh - 1u : 0u;
        const int sum = tail_in_range(c[offset] - yy, ab, al) + tail_in_range(c[-offset] - yy, ab, al)
                      + tail_in_range(c[2*offset] - yy, ab, al) + tail_in_range(c[-2*offset] - yy, ab, al);
        o[r] = (int16_t)(yy + ((3*sum + 8) >> 4));
      }
    }
    __syncthreads();           // the tiles are rewritten by the next plane
    TAIL_STAMP(4);                                   // orthogonal filter
  }
  else if (enc_mode) {
#pragma unroll
    for (int r = 0; r < NPX; r++) o[r] = P16[(size_t)(r*RSTEP + i0)*w + j];
  }
  else {
#pragma unroll
    for (int r = 0; r < NPX; r++) o[r] = P[(size_t)(r*RSTEP + i0)*w + j];
  }
  if (smooth_on) {
    // od_bilinear_smooth (src/filter.c:1952-2008) on the whole n x n tile: corners from the
    // lanes that hold them
    const int32_t x00 = __shfl(o[0], 0), x01 = __shfl(o[0], n - 1);
    const int32_t x10 = __shfl(o[NPX - 1], 64 - n), x11 = __shfl(o[NPX - 1], 63);
    const int32_t a00 = x00;
    int32_t a01 = x01 - x00, a10 = x10 - x00, a11 = x11 + x00 - x10 - x01;
    a01 += (a01 + n/2) >> LN;
    a10 += (a10 + n/2) >> LN;
    a11 += (2*a10 + n/2) >> LN;
    int shift = 2*4 + 2*LN - 16;
    shift = shift < 0 ? 0 : shift;
    // per row: sum of squared differences >> shift (the row sits in n neighbouring lanes)
    int32_t yv[NPX];
    int32_t dist = 0;
#pragma unroll
    for (int r = 0; r < NPX; r++) {
      const int i = r*RSTEP + i0;
      yv[r] = a00 + ((j*a01 + i*a10 + (j*i*a11 >> LN) + n/2) >> LN);
      const int32_t dd = yv[r] - o[r];
      dist += tail_sum_group<n>(dd*dd) >> shift;          // every lane of the row holds the row's term
    }
    // one lane per row group holds what the rows it saw add up to: lanes 0, n, 2n, ...
    {
      int32_t tot = 0;
#pragma unroll
      for (int q = 0; q < 64; q += n) tot += __builtin_amdgcn_readlane(dist, q);
      dist = tot;
    }
    dist += n/2;
    dist >>= 2*LN - shift;
    const int strength = (pli == 1 || pli == 2) ? 20 : 5;
    int wq = strength*a.q[pli]*a.q[pli]/(1 + 12*dist);
    wq = wq > 1024 ? 1024 : wq;
    wq = wq*wq >> 12;
#pragma unroll
    for (int r = 0; r < NPX; r++) o[r] = o[r] - ((wq*(o[r] - yv[r]) + 128) >> 8);
  }
  TAIL_STAMP(5);                                     // smoothing (or the plain load)
  if (enc_mode) {
    int16_t *O = a.o16[pli] + porg;
#pragma unroll
    for (int r = 0; r < NPX; r++) O[(size_t)(r*RSTEP + i0)*w + j] = (int16_t)o[r];
    return;
  }
  // od_coeff_to_ref_buf (src/state.c:1274-1300); the bytes of four neighbouring samples are
  // gathered through LDS so that the plane is written as dwords
  uint8_t *bytes = reinterpret_cast<uint8_t *>(S.scratch);
#pragma unroll
  for (int r = 0; r < NPX; r++) {
    int v = ((o[r] + 8) >> 4) + 128;
    v = v < 0 ? 0 : v > 255 ? 255 : v;
    bytes[(r*RSTEP + i0)*n + j] = (uint8_t)v;
  }
  __syncthreads();
  uint8_t *R = a.rec[pli] + porg;
#pragma unroll
  for (int q = 0; q < n*n/256; q++) {
    const int e4 = 4*(lane + 64*q);
    *reinterpret_cast<uint32_t *>(R + (size_t)(e4 >> LN)*w + (e4 & (n - 1))) = S.scratch[lane + 64*q];
  }
  __syncthreads();
  TAIL_STAMP(6);                                     // clamp, pack, store
#ifdef TAIL_STAMPS
  if (pli == 0 && lane == 0 && ((sbx + sby) & 15) == 0) {
    for (int q = 0; q < 7; q++) atomicAdd(&g_tail_stamps[q], t_acc[q]);
    atomicAdd(&g_tail_stamps[15], 1ull);
  }
#endif
}

#define TAIL_THREADS 64
// F420: three planes, chroma decimated in both directions (the host knows the geometry and picks
// the instantiation): a plane's loads leave while the plane before it is filtered.  Otherwise
// (one plane, 4:4:4, ...) the planes simply follow one another.
template <bool F420>
__global__ __launch_bounds__(TAIL_THREADS, F420 ? 4 : 2) void k_decode_tail(TailArgs a) {
  __shared__ TailShared S;
  const int lane = threadIdx.x;
  int sbx, sby, f;
  xcd_tile_coords(sbx, sby, f);        // 3-sample borders: neighbours share lines
  const bool enc_mode = a.flags == nullptr;
  // the superblock's flag and block size and the two small tables of the direction search: four
  // independent reads, one wait (every lane reads a valid table entry; the tables go to LDS so
  // that their reads do not stand in the dependent chain after the search)
  int flag = 1, sb_bsize = 0;
  if (!enc_mode) {
    flag = a.flags[(size_t)f*a.nhsb*a.nvsb + sby*a.nhsb + sbx];
    sb_bsize = a.bsize[(size_t)f*a.bsize_fstride + (size_t)(sby*4)*a.bstride + sbx*4];
  }
  const int lo_ = lane < 24 ? lane : 23, lt_ = lane >= 32 && lane < 50 ? lane - 32 : 0;
  const int t_dy = TAIL_DIR[lo_/3][lo_%3][0], t_dx = TAIL_DIR[lo_/3][lo_%3][1], t_thr = TAIL_THRESH_Q8[lt_];
  const bool dering_on = a.q[0] > 0 && flag;
  const bool smooth_on = !enc_mode && a.q[0] > 0 && a.is_keyframe && sb_bsize == 3;
  S.tab[lane] = (int16_t)(lane < 24 ? t_dy*TAIL_BSTRIDE + t_dx : t_thr);      // no branch: the reads stay up here
  if (F420) {
    TailTile<5> t0;
    TailTile<4> t1, t2;
    if (dering_on) {
      tail_tile_issue<5>(t0, a, 0, sbx, sby, f, enc_mode, lane);
      tail_tile_issue<4>(t1, a, 1, sbx, sby, f, enc_mode, lane);
    }
    tail_plane<5, true>(a, S, 0, sbx, sby, f, enc_mode, dering_on, smooth_on, lane, t0);
    if (dering_on) tail_tile_issue<4>(t2, a, 2, sbx, sby, f, enc_mode, lane);   // travels while the first chroma plane is filtered
    tail_plane<4, true>(a, S, 1, sbx, sby, f, enc_mode, dering_on, smooth_on, lane, t1);
    tail_plane<4, true>(a, S, 2, sbx, sby, f, enc_mode, dering_on, smooth_on, lane, t2);
  }
  else {
    for (int pli = 0; pli < a.nplanes; pli++) {
      if (a.xdec[pli]) {
        TailTile<4> t;
        tail_plane<4, false>(a, S, pli, sbx, sby, f, enc_mode, dering_on, smooth_on, lane, t);
      }
      else {
        TailTile<5> t;
        tail_plane<5, false>(a, S, pli, sbx, sby, f, enc_mode, dering_on, smooth_on, lane, t);
      }
    }
  }
}

// host side: the instantiation that fits the geometry
static inline void tail_launch(const TailArgs &t, dim3 grid, hipStream_t stream) {
  if (t.nplanes == 3 && !t.xdec[0] && t.xdec[1] && t.xdec[2]) {
    hipLaunchKernelGGL(k_decode_tail<true>, grid, dim3(TAIL_THREADS), 0, stream, t);
  }
  else hipLaunchKernelGGL(k_decode_tail<false>, grid, dim3(TAIL_THREADS), 0, stream, t);
}
