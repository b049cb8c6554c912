// F3, first inter kernel: overlapped block motion compensation of a list
// of prediction blocks - what od_state_mc_predict (src/state.c:993) produces leaf by leaf
// through od_state_pred_block_from_setup -> od_mc_predict (src/mc.c:2006): four
// single-vector predictions (one per block corner: od_mc_predict1fmv8_c, src/mc.c:94, a
// separable 6-tap 1/8-sample interpolation) blended by position (od_mc_blend_full8_c :352,
// or od_mc_blend_full_split8_c :1104 with the weights of od_mc_setup_s_split :1056 when an
// edge of the block is not split).  The leaf list is a pure function of the decoded motion
// vector grid and the reference frames, so the whole prediction of a frame is one launch.
//
// One workgroup of two waves per block (blocks are 4x4 .. 64x64: OD_MVBSIZE_MAX, src/internal.h:64).  Per corner: horizontal pass of the
// (yblk + 5) rows into an int16 LDS tile, vertical pass into a u8 LDS tile; corners that
// share reference and vector with an earlier corner reuse its tile (the reference's own
// shortcut, src/mc.c:1979-1999).  Then one blending pass writes the block.  All integer.
#pragma once
#include <stdint.h>

struct McBlock {          // == od_hip_mc_block (include/daala_hip.h)
  int32_t x, y;
  int32_t log_xblk_sz, log_yblk_sz;
  int32_t ref[4];
  int32_t mvx[4], mvy[4];
  int32_t oc, s;
};

struct McArgs {
  const uint8_t *refs;    // nref planes of ref_h x ref_stride bytes, back to back
  size_t ref_plane;       // bytes per reference plane
  int ref_stride, ref_h;
  int org_x, org_y;       // picture origin inside a reference plane (the padding)
  const McBlock *blocks;
  int nblocks;
  uint8_t *dst;
  int dst_stride;
};

__constant__ int16_t MC_SUBPEL[8][6] = {     // OD_SUBPEL_FILTER_SET (src/mc.c:66-77)
  {0, 0, 128, 0, 0, 0}, {1, -9, 122, 18, -5, 1}, {3, -15, 112, 37, -11, 2},
  {3, -18, 97, 58, -15, 3}, {4, -20, 80, 80, -20, 4}, {3, -15, 58, 97, -18, 3},
  {2, -11, 37, 112, -15, 3}, {1, -5, 18, 122, -9, 1}};

#define MC_THREADS 128
__global__ __launch_bounds__(MC_THREADS) void k_mc_predict_blocks(McArgs a) {
  __shared__ int16_t buff[(64 + 5)*64];
  __shared__ uint8_t pred[4][64*64];
  const int lane = threadIdx.x;
  const int bidx = blockIdx.x;
  if (bidx >= a.nblocks) return;
  const McBlock blk = a.blocks[bidx];
  const int lx = blk.log_xblk_sz, ly = blk.log_yblk_sz;
  const int xblk = 1 << lx, yblk = 1 << ly, npix = xblk*yblk;
  int alias[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    alias[k] = k;
    for (int e = 0; e < k; e++) {
      if (alias[k] == k && blk.ref[e] == blk.ref[k] && blk.mvx[e] == blk.mvx[k]
          && blk.mvy[e] == blk.mvy[k]) alias[k] = alias[e];
    }
  }
  for (int k = 0; k < 4; k++) {
    if (alias[k] != k) continue;                       // wave-uniform
    const int mvx = blk.mvx[k], mvy = blk.mvy[k];
    const int mvxf = mvx & 7, mvyf = mvy & 7;
    const uint8_t *plane = a.refs + (size_t)blk.ref[k]*a.ref_plane;
    const int sx0 = a.org_x + blk.x + (mvx >> 3), sy0 = a.org_y + blk.y + (mvy >> 3);
    auto px = [&](int yy, int xx) -> int {
      // the reference relies on the padding of its reference frames; the clamp only keeps
      // a vector that points beyond it from reading outside the buffer
      yy = yy < 0 ? 0 : yy >= a.ref_h ? a.ref_h - 1 : yy;
      xx = xx < 0 ? 0 : xx >= a.ref_stride ? a.ref_stride - 1 : xx;
      return plane[(size_t)yy*a.ref_stride + xx];
    };
    if (mvxf || mvyf) {
      // 1st stage: rows -2 .. yblk + 2 (src/mc.c:145-172)
      const int nrow = yblk + 5;
      for (int e = lane; e < nrow*xblk; e += MC_THREADS) {
        const int j = (e >> lx) - 2, i = e & (xblk - 1);
        int v;
        if (mvxf) {
          int sum = 0;
#pragma unroll
          for (int t = 0; t < 6; t++) sum += px(sy0 + j, sx0 + i + t - 2)*MC_SUBPEL[mvxf][t];
          v = sum - (128 << 7);
        }
        else v = (px(sy0 + j, sx0 + i) << 7) - (128 << 7);
        buff[e] = (int16_t)v;
      }
      __syncthreads();
      // 2nd stage (src/mc.c:174-198)
      for (int e = lane; e < npix; e += MC_THREADS) {
        const int j = e >> lx, i = e & (xblk - 1);
        int v;
        if (mvyf) {
          int sum = 0;
#pragma unroll
          for (int t = 0; t < 6; t++) sum += buff[(j + t)*xblk + i]*MC_SUBPEL[mvyf][t];
          v = (sum + (1 << 13) + (128 << 14)) >> 14;
        }
        else v = (buff[(j + 2)*xblk + i] + (1 << 6) + (128 << 7)) >> 7;
        pred[k][e] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
      }
      __syncthreads();
    }
    else {
      for (int e = lane; e < npix; e += MC_THREADS) pred[k][e] = (uint8_t)px(sy0 + (e >> lx), sx0 + (e & (xblk - 1)));
      __syncthreads();
    }
  }
  const uint8_t *p0 = pred[alias[0]], *p1 = pred[alias[1]], *p2 = pred[alias[2]], *p3 = pred[alias[3]];
  uint8_t *d = a.dst + (size_t)blk.y*a.dst_stride + blk.x;
  const int l2 = lx + ly;
  if (blk.s == 3) {
    // od_mc_blend_full8_c
    const int round = 1 << (l2 - 1);
    for (int e = lane; e < npix; e += MC_THREADS) {
      const int j = e >> lx, i = e & (xblk - 1);
      int av = p0[e], bv = p3[e];
      av = (av << lx) + (p1[e] - av)*i;
      bv = (bv << lx) + (p2[e] - bv)*i;
      d[(size_t)j*a.dst_stride + i] = (uint8_t)(((av << ly) + (bv - av)*j + round) >> l2);
    }
  }
  else {
    // od_mc_setup_s_split + od_mc_blend_full_split8_c: the weight of corner c at (i, j) is
    // s0[c] + j*dsdj[c] + i*(dsdi[c] + j*ddsdidj[c]) - the closed form of its row/column
    // increments
    int s0[4] = {2 << l2, 0, 0, 0};
    int dsdi[4] = {-(2 << lx), 2 << lx, 0, 0};
    int dsdj[4] = {-(2 << ly), 0, 0, 2 << ly};
    int dd[4] = {2, -2, 2, -2};
    const int oc = blk.oc & 3;
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const bool on = t == 0 ? !(blk.s & 1) : !(blk.s & 2);
      const int c = t == 0 ? (oc + 1) & 3 : (oc + 3) & 3;
      if (on) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (q == c) { s0[q] >>= 1; dsdi[q] >>= 1; dsdj[q] >>= 1; dd[q] >>= 1; }
        }
        int hs = 0, hi = 0, hj = 0, hd = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (q == c) { hs = s0[q]; hi = dsdi[q]; hj = dsdj[q]; hd = dd[q]; }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (q == oc) { s0[q] += hs; dsdi[q] += hi; dsdj[q] += hj; dd[q] += hd; }
        }
      }
    }
    const int round = 1 << l2;
    for (int e = lane; e < npix; e += MC_THREADS) {
      const int j = e >> lx, i = e & (xblk - 1);
      const int w1 = s0[1] + j*dsdj[1] + i*(dsdi[1] + j*dd[1]);
      const int w2 = s0[2] + j*dsdj[2] + i*(dsdi[2] + j*dd[2]);
      const int w3 = s0[3] + j*dsdj[3] + i*(dsdi[3] + j*dd[3]);
      const int av = p0[e];
      const int bv = (p1[e] - av)*w1, cv = (p2[e] - av)*w2, dv = (p3[e] - av)*w3;
      d[(size_t)j*a.dst_stride + i] = (uint8_t)(((av << (l2 + 1)) + bv + cv + dv + round) >> (l2 + 1));
    }
  }
}
