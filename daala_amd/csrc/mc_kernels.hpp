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

// The reference planes of one image plane as a kernel sees them.
struct McPlaneRef {
  const uint8_t *refs;    // nref planes of ref_h x ref_stride bytes, back to back
  size_t ref_plane;       // bytes per reference plane
  int ref_stride, ref_h;
  int org_x, org_y;       // picture origin inside a reference plane (the padding)
};

// LDS of a prediction kernel instantiated for blocks up to (1 << LM)^2 samples: the reference
// window of one corner (rows -2 .. n+2, columns -2 .. n+2), the horizontal pass' 16-bit rows, and
// NP prediction tiles.  Sized by the LARGEST block a launch holds (the host launches runs of
// blocks of one size class): the 64x64 layout is 30 KB per workgroup - five single-wave
// workgroups per CU - the 16x16 one 2 KB.
template <int LM, int NP>
struct McTiles {
  static constexpr int N = 1 << LM;
  static constexpr int SS = N + 8;                   // window stride (N + 5 used)
  uint8_t stage[(N + 5)*SS];
  int16_t buff[(N + 5)*N];
  uint8_t pred[NP*N*N];
};

// The four single-vector predictions of a block (one per corner) into T.pred[k][...] (N*N bytes
// apart), NT threads cooperating; corners that share reference and vector with an earlier corner
// are not predicted again: alias[k] names the tile that holds corner k's prediction.
// A corner's reference window is staged in LDS ONCE (clamped byte loads, a row of the window per
// row of lanes) and both filter passes read LDS: the first form fetched every tap of every sample
// from global memory through two clamps - six loads and twenty-four min/max per output sample
// of the horizontal pass.
template <int NT, int LM, int NP>
__device__ __forceinline__ void mc_predict_corners(const McPlaneRef &R, int bx, int by, int lx, int ly,
                                                   const int32_t *ref, const int32_t *cmvx, const int32_t *cmvy,
                                                   McTiles<LM, NP> &T, int *alias, int lane) {
  constexpr int PN = McTiles<LM, NP>::N*McTiles<LM, NP>::N, SS = McTiles<LM, NP>::SS;
  const int xblk = 1 << lx, yblk = 1 << ly, npix = xblk*yblk;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    alias[k] = k;
    for (int e = 0; e < k; e++) {
      if (alias[k] == k && ref[e] == ref[k] && cmvx[e] == cmvx[k] && cmvy[e] == cmvy[k]) alias[k] = alias[e];
    }
  }
  for (int k = 0; k < 4; k++) {
    if (alias[k] != k) continue;                       // wave-uniform
    const int mvx = cmvx[k], mvy = cmvy[k];
    const int mvxf = mvx & 7, mvyf = mvy & 7;
    const uint8_t *plane = R.refs + (size_t)ref[k]*R.ref_plane;
    const int sx0 = R.org_x + bx + (mvx >> 3), sy0 = R.org_y + by + (mvy >> 3);
    uint8_t *pk = T.pred + (NP == 1 ? 0 : k*PN);
    auto px = [&](int yy, int xx) -> int {
      // the reference relies on the padding of its reference frames; the clamp only keeps
      // a vector that points beyond it from reading outside the buffer
      yy = yy < 0 ? 0 : yy >= R.ref_h ? R.ref_h - 1 : yy;
      xx = xx < 0 ? 0 : xx >= R.ref_stride ? R.ref_stride - 1 : xx;
      return plane[(size_t)yy*R.ref_stride + xx];
    };
    if (mvxf || mvyf) {
      // the window: rows -2 .. yblk + 2, columns -2 .. xblk + 2 of the displaced block
      const int nrow = yblk + 5, ncol = xblk + 5;
      const float ncol_1 = 1.0f/(float)ncol;
      for (int e = lane; e < nrow*ncol; e += NT) {
        const int r = (int)(((float)e + 0.5f)*ncol_1);   // e/ncol, exact for these small integers
        const int c = e - r*ncol;
        T.stage[r*SS + c] = (uint8_t)px(sy0 + r - 2, sx0 + c - 2);
      }
      __syncthreads();
      // 1st stage: rows -2 .. yblk + 2 (src/mc.c:145-172)
      for (int e = lane; e < nrow*xblk; e += NT) {
        const int r = e >> lx, i = e & (xblk - 1);
        const uint8_t *w = T.stage + r*SS + i;          // columns i - 2 .. i + 3 of the window row
        int v;
        if (mvxf) {
          int sum = 0;
#pragma unroll
          for (int t = 0; t < 6; t++) sum += w[t]*MC_SUBPEL[mvxf][t];
          v = sum - (128 << 7);
        }
        else v = (w[2] << 7) - (128 << 7);
        T.buff[e] = (int16_t)v;
      }
      __syncthreads();
      // 2nd stage (src/mc.c:174-198)
      for (int e = lane; e < npix; e += NT) {
        const int j = e >> lx, i = e & (xblk - 1);
        int v;
        if (mvyf) {
          int sum = 0;
#pragma unroll
          for (int t = 0; t < 6; t++) sum += T.buff[(j + t)*xblk + i]*MC_SUBPEL[mvyf][t];
          v = (sum + (1 << 13) + (128 << 14)) >> 14;
        }
        else v = (T.buff[(j + 2)*xblk + i] + (1 << 6) + (128 << 7)) >> 7;
        pk[e] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
      }
      __syncthreads();
    }
    else {
      for (int e = lane; e < npix; e += NT) pk[e] = (uint8_t)px(sy0 + (e >> lx), sx0 + (e & (xblk - 1)));
      __syncthreads();
    }
  }
}

// Blending weights of a block: od_mc_blend_full8_c (s == 3) or od_mc_setup_s_split +
// od_mc_blend_full_split8_c - the weight of corner c at (i, j) is
// s0[c] + j*dsdj[c] + i*(dsdi[c] + j*dd[c]), the closed form of its row/column increments.
struct McBlend {
  int full, lx, ly, l2;
  int s0[4], dsdi[4], dsdj[4], dd[4];
  __device__ __forceinline__ void setup(int lx_, int ly_, int oc_, int s) {
    lx = lx_;
    ly = ly_;
    l2 = lx + ly;
    full = s == 3;
    if (full) return;
    const int is0[4] = {2 << l2, 0, 0, 0};
    const int idsdi[4] = {-(2 << lx), 2 << lx, 0, 0};
    const int idsdj[4] = {-(2 << ly), 0, 0, 2 << ly};
    const int idd[4] = {2, -2, 2, -2};
#pragma unroll
    for (int q = 0; q < 4; q++) { s0[q] = is0[q]; dsdi[q] = idsdi[q]; dsdj[q] = idsdj[q]; dd[q] = idd[q]; }
    const int oc = oc_ & 3;
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const bool on = t == 0 ? !(s & 1) : !(s & 2);
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
  }
  __device__ __forceinline__ int value(int v0, int v1, int v2, int v3, int i, int j) const {
    if (full) {
      int av = v0, bv = v3;
      av = (av << lx) + (v1 - av)*i;
      bv = (bv << lx) + (v2 - bv)*i;
      return ((av << ly) + (bv - av)*j + (1 << (l2 - 1))) >> l2;
    }
    const int w1 = s0[1] + j*dsdj[1] + i*(dsdi[1] + j*dd[1]);
    const int w2 = s0[2] + j*dsdj[2] + i*(dsdi[2] + j*dd[2]);
    const int w3 = s0[3] + j*dsdj[3] + i*(dsdi[3] + j*dd[3]);
    return ((v0 << (l2 + 1)) + (v1 - v0)*w1 + (v2 - v0)*w2 + (v3 - v0)*w3 + (1 << l2)) >> (l2 + 1);
  }
};

#define MC_THREADS 128
// LM: log2 of the largest block side of the launch (4, 5 or 6; the host launches runs of one class)
template <int LM>
__global__ __launch_bounds__(MC_THREADS) void k_mc_predict_blocks(McArgs a) {
  __shared__ McTiles<LM, 4> T;
  constexpr int PN = McTiles<LM, 4>::N*McTiles<LM, 4>::N;
  const int lane = threadIdx.x;
  const int bidx = blockIdx.x;
  if (bidx >= a.nblocks) return;
  const McBlock blk = a.blocks[bidx];
  const int lx = blk.log_xblk_sz, ly = blk.log_yblk_sz;
  const int xblk = 1 << lx, npix = xblk << ly;
  McPlaneRef R;
  R.refs = a.refs;
  R.ref_plane = a.ref_plane;
  R.ref_stride = a.ref_stride;
  R.ref_h = a.ref_h;
  R.org_x = a.org_x;
  R.org_y = a.org_y;
  int alias[4];
  mc_predict_corners<MC_THREADS, LM, 4>(R, blk.x, blk.y, lx, ly, blk.ref, blk.mvx, blk.mvy, T, alias, lane);
  const uint8_t *p0 = T.pred + alias[0]*PN, *p1 = T.pred + alias[1]*PN, *p2 = T.pred + alias[2]*PN,
                *p3 = T.pred + alias[3]*PN;
  uint8_t *d = a.dst + (size_t)blk.y*a.dst_stride + blk.x;
  McBlend W;
  W.setup(lx, ly, blk.oc, blk.s);
  for (int e = lane; e < npix; e += MC_THREADS) {
    const int j = e >> lx, i = e & (xblk - 1);
    d[(size_t)j*a.dst_stride + i] = (uint8_t)W.value(p0[e], p1[e], p2[e], p3[e], i, j);
  }
}

// F3, second half: the OBMC prediction of a block FUSED with its SAD against the frame being
// coded - od_mv_est_sad (src/mcenc.c:2271-2300): od_state_pred_block_from_setup (src/state.c:689)
// of every plane + od_enc_sad (:1615, the block clipped against the picture) with the chroma sums
// scaled down by OD_MC_CHROMA_SCALE (:53).  The prediction never leaves LDS.  One item = one
// (block, exterior corner, split state) with the four corner vectors in luma units; chroma
// vectors and positions are derived here (OD_DIV_POW2_RE, src/odintrin.h:142).  One WAVE per
// item (no workgroup barrier is ever waited on by a second wave), planes in turn.  What
// od_mv_est_calc_sads (:3761) computes block by block for a fixed vector grid - every block of
// two sizes x four split states - is one launch.
struct McSadItem {        // == od_hip_mc_sad_item (include/daala_hip.h)
  int32_t x, y;           // luma position
  int32_t log_blk_sz;     // luma log2 size, 3 .. 6
  int32_t oc, s;
  int32_t ref[4];
  int32_t mvx[4], mvy[4];
  int32_t reserved;
};

struct McSadPlane {
  McPlaneRef R;
  const uint8_t *src;     // the frame being coded, src_stride bytes per row
  int src_stride;
  int xdec, ydec;
  int clip_w, clip_h;     // picture size in this plane's samples (od_enc_sad's clip)
  int shift;              // 0 for luma, OD_MC_CHROMA_SCALE for chroma
};

struct McSadArgs {
  McSadPlane pl[3];
  int nplanes;
  const McSadItem *items;
  int nitems;
  int32_t *sad;
};

__device__ __forceinline__ int mc_div_pow2_re(int x, int shift) {
  return (x + (((1 << shift) + ((x >> shift) & 1) - 1) >> 1)) >> shift;
}

#define MC_SAD_THREADS 64
template <int LM>
__global__ __launch_bounds__(MC_SAD_THREADS) void k_mc_sad_items(McSadArgs a) {
  __shared__ McTiles<LM, 4> T;
  constexpr int PN = McTiles<LM, 4>::N*McTiles<LM, 4>::N;
  const int lane = threadIdx.x;
  const int idx = blockIdx.x;
  if (idx >= a.nitems) return;
  const McSadItem it = a.items[idx];
  int total = 0;
  for (int pli = 0; pli < a.nplanes; pli++) {
    const McSadPlane &P = a.pl[pli];
    const int lx = it.log_blk_sz - P.xdec, ly = it.log_blk_sz - P.ydec;
    const int bx = it.x >> P.xdec, by = it.y >> P.ydec;
    const int xblk = 1 << lx, yblk = 1 << ly, npix = xblk*yblk;
    int32_t cmvx[4], cmvy[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      cmvx[k] = mc_div_pow2_re(it.mvx[k], P.xdec);
      cmvy[k] = mc_div_pow2_re(it.mvy[k], P.ydec);
    }
    int alias[4];
    mc_predict_corners<MC_SAD_THREADS, LM, 4>(P.R, bx, by, lx, ly, it.ref, cmvx, cmvy, T, alias, lane);
    const uint8_t *p0 = T.pred + alias[0]*PN, *p1 = T.pred + alias[1]*PN, *p2 = T.pred + alias[2]*PN,
                  *p3 = T.pred + alias[3]*PN;
    McBlend W;
    W.setup(lx, ly, it.oc, it.s);
    const int w = min(xblk, P.clip_w - bx), h = min(yblk, P.clip_h - by);
    const uint8_t *src = P.src + (size_t)by*P.src_stride + bx;
    int acc = 0;
    for (int e = lane; e < npix; e += MC_SAD_THREADS) {
      const int j = e >> lx, i = e & (xblk - 1);
      if (i < w && j < h) {
        const int v = W.value(p0[e], p1[e], p2[e], p3[e], i, j) & 255;     // the (unsigned char) store of the blend
        acc += abs(v - (int)src[(size_t)j*P.src_stride + i]);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    total += acc >> P.shift;
    __syncthreads();                                   // the tiles are rewritten by the next plane
  }
  if (lane == 0) a.sad[idx] = total;
}

// A reference plane built ON the device from a context's reconstruction plane: the frame area
// copied, the padding filled by replicating the frame's edge samples - what
// od_coeff_to_ref_plane + od_img_edge_ext (src/state.c:1100-1171: left/right from the row's
// edge sample, top/bottom from the extended edge rows) leave in a reference image.  Four
// output bytes per thread.
struct McRefArgs {
  const uint8_t *rec;     // pw x ph, dense
  int pw, ph;
  uint8_t *ref;           // ref_h rows of ref_stride bytes
  int ref_stride, ref_h, org_x, org_y;
};

__global__ __launch_bounds__(256) void k_mc_ref_from_rec(McRefArgs a) {
  const int x4 = (blockIdx.x*256 + threadIdx.x)*4, y = blockIdx.y;
  if (x4 >= a.ref_stride) return;
  int sy = y - a.org_y;
  sy = sy < 0 ? 0 : sy >= a.ph ? a.ph - 1 : sy;
  const uint8_t *row = a.rec + (size_t)sy*a.pw;
  uint32_t word = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int sx = x4 + k - a.org_x;
    sx = sx < 0 ? 0 : sx >= a.pw ? a.pw - 1 : sx;
    word |= (uint32_t)row[sx] << (8*k);
  }
  if (x4 + 4 <= a.ref_stride) *reinterpret_cast<uint32_t *>(a.ref + (size_t)y*a.ref_stride + x4) = word;
  else for (int k = 0; x4 + k < a.ref_stride; k++) a.ref[(size_t)y*a.ref_stride + x4 + k] = (uint8_t)(word >> (8*k));
}

// F3: dense block-matching windows for the EPZS initialisation of the motion search.
// od_mv_est_init_mv (src/mcenc.c:2511) judges candidate vectors of one grid vertex by
// od_mv_est_bma_sad (:2228-2268): ONE single-vector prediction of the block centred on the vertex
// (od_mc_predict1fmv8_c, every plane, vectors in half samples: mvx*(1 << (2 - xdec))) and
// od_enc_sad against the frame being coded (the block clipped against the picture on every side,
// it may hang over the frame's edge), chroma >> OD_MC_CHROMA_SCALE.  Which candidates it asks for
// depends on the SADs it has seen, but they cluster around the median predictor - known for every
// vertex of a level before the level starts - so the device evaluates the WHOLE (2R + 1)^2 window
// of half-sample vectors around it for every vertex of the level in one launch and the host's
// decision logic (the reference's own function) looks its SADs up.  One wave per (vertex, offset).
struct McBmaRec {         // == od_hip_mc_bma_rec (include/daala_hip.h)
  int32_t bx, by;         // luma position of the block's upper-left corner (may be negative)
  int32_t log_blk_sz;     // luma log2 size, 3 .. 6
  int32_t ref;            // reference image index
  int32_t cx, cy;         // window centre, half samples
  int32_t xmin, xmax, ymin, ymax;    // the vertex's vector limits, half samples (inclusive)
};

struct McBmaArgs {
  McSadPlane pl[3];
  int nplanes;
  const McBmaRec *recs;
  int nrec;
  int radius;             // R: offsets -R .. R in both directions
  int32_t *sad;           // [nrec][(2R + 1)^2]; -1: outside the limits, not evaluated
};

template <int LM>
__global__ __launch_bounds__(MC_SAD_THREADS) void k_mc_bma_windows(McBmaArgs a) {
  __shared__ McTiles<LM, 1> T;
  const uint8_t *pred = T.pred;
  const int lane = threadIdx.x;
  const int W = 2*a.radius + 1;
  const int rec = blockIdx.y, o = blockIdx.x;
  if (rec >= a.nrec || o >= W*W) return;
  const McBmaRec r = a.recs[rec];
  const int mvx = r.cx + o%W - a.radius, mvy = r.cy + o/W - a.radius;
  int32_t *out = a.sad + (size_t)rec*W*W + o;
  if (mvx < r.xmin || mvx > r.xmax || mvy < r.ymin || mvy > r.ymax) {
    if (lane == 0) *out = -1;
    return;
  }
  int total = 0;
  for (int pli = 0; pli < a.nplanes; pli++) {
    const McSadPlane &P = a.pl[pli];
    const int lx = r.log_blk_sz - P.xdec, ly = r.log_blk_sz - P.ydec;
    const int bx = r.bx >> P.xdec, by = r.by >> P.ydec;          // arithmetic: positions are multiples of 4
    const int xblk = 1 << lx, yblk = 1 << ly, npix = xblk*yblk;
    int32_t ref4[4], cmvx[4], cmvy[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      ref4[k] = r.ref;
      cmvx[k] = mvx*(1 << (2 - P.xdec));
      cmvy[k] = mvy*(1 << (2 - P.ydec));
    }
    int alias[4];
    mc_predict_corners<MC_SAD_THREADS, LM, 1>(P.R, bx, by, lx, ly, ref4, cmvx, cmvy, T, alias, lane);   // one tile
    // od_enc_sad: the block clipped against [0, clip_w) x [0, clip_h)
    int acc = 0;
    for (int e = lane; e < npix; e += MC_SAD_THREADS) {
      const int j = e >> lx, i = e & (xblk - 1);
      const int sx = bx + i, sy = by + j;
      if (sx >= 0 && sx < P.clip_w && sy >= 0 && sy < P.clip_h) {
        acc += abs((int)pred[e] - (int)P.src[(size_t)sy*P.src_stride + sx]);
      }
    }
#pragma unroll
    for (int q = 32; q > 0; q >>= 1) acc += __shfl_xor(acc, q);
    total += acc >> P.shift;
    __syncthreads();
  }
  if (lane == 0) *out = total;
}

// The same windows, one WAVE PER VERTEX (k_mc_bma_windows above: one wave per vertex AND offset,
// which filters the block anew for each of the (2R + 1)^2 offsets).  The offsets of a window share
// very few sub-sample phases: a half-sample vector is mvx*4 eighths of a luma sample - phase 0 or
// 4 - and mvx*2 eighths of a 4:2:0 chroma sample - phase 0, 2, 4 or 6; their integer parts span
// R + 1 (luma) or R/2 + 2 (chroma) positions.  So per plane: the reference window is staged once,
// every horizontal phase is filtered once over the whole region (od_mc_predict1fmv8_c's first
// stage, src/mc.c:145-172), every (horizontal, vertical) phase pair once (second stage, :174-198
// - with phase 0 the two stages reduce to the copy the reference takes for whole-sample vectors),
// and each offset's SAD (od_enc_sad: the block clipped against the picture) reads its phase
// plane at its integer displacement.  Same numbers, about a tenth of the arithmetic.
template <int LM>
struct McBmaTiles {
  static constexpr int N = 1 << LM;
  static constexpr int WMAX = 6;                       // integer displacements of a window (R <= 4)
  static constexpr int PW = N + WMAX - 1;              // phase plane side
  static constexpr int SW = PW + 5, SS = SW + 3;       // staged window: 2 samples left/above, 3 right/below
  uint8_t stage[SW*SS];
  int16_t hbuf[SW*PW];
  uint8_t plane[PW*PW];
  uint8_t srcb[N*N];                                   // the block of the frame being coded
  int32_t tot[96];                                     // the window's SAD sums
  int32_t part[96];                                    // ... of the plane in hand
};

// NT threads per vertex: one wave for blocks up to 16x16, four for the larger ones (a level of large
// blocks has few vertices: one wave each would leave most of the chip idle)
template <int LM, int NT>
__global__ __launch_bounds__(NT) void k_mc_bma_windows_v2(McBmaArgs a) {
  using TT = McBmaTiles<LM>;
  __shared__ TT T;
  const int lane = threadIdx.x;
  const int R = a.radius, W = 2*R + 1, NO = W*W;
  const int rec = blockIdx.x;
  if (rec >= a.nrec) return;
  const McBmaRec r = a.recs[rec];
  for (int o = lane; o < NO; o += NT) T.tot[o] = 0;
  for (int pli = 0; pli < a.nplanes; pli++) {
    const McSadPlane &P = a.pl[pli];
    const int lx = r.log_blk_sz - P.xdec, ly = r.log_blk_sz - P.ydec;
    const int bx = r.bx >> P.xdec, by = r.by >> P.ydec;          // arithmetic: positions are multiples of 4
    const int xblk = 1 << lx, yblk = 1 << ly, npix = xblk*yblk;
    const int shx = 2 - P.xdec, shy = 2 - P.ydec;                // half samples -> eighths of this plane's samples
    // integer displacements the window spans, and the region the phase planes cover
    const int ix0 = ((r.cx - R)*(1 << shx)) >> 3, ix1 = ((r.cx + R)*(1 << shx)) >> 3;
    const int iy0 = ((r.cy - R)*(1 << shy)) >> 3, iy1 = ((r.cy + R)*(1 << shy)) >> 3;
    const int pw = xblk + ix1 - ix0, ph = yblk + iy1 - iy0;      // <= TT::PW
    const int sw = pw + 5, sh = ph + 5;
    const uint8_t *plane_ref = P.R.refs + (size_t)r.ref*P.R.ref_plane;
    const int sx0 = P.R.org_x + bx + ix0 - 2, sy0 = P.R.org_y + by + iy0 - 2;
    {
      const float sw_1 = 1.0f/(float)sw;
      for (int e = lane; e < sh*sw; e += NT) {
        const int rr = (int)(((float)e + 0.5f)*sw_1), c = e - rr*sw;      // e/sw, exact for these small integers
        int yy = sy0 + rr, xx = sx0 + c;
        yy = yy < 0 ? 0 : yy >= P.R.ref_h ? P.R.ref_h - 1 : yy;
        xx = xx < 0 ? 0 : xx >= P.R.ref_stride ? P.R.ref_stride - 1 : xx;
        T.stage[rr*TT::SS + c] = plane_ref[(size_t)yy*P.R.ref_stride + xx];
      }
      // the block of the frame being coded; samples outside the picture never count (od_enc_sad)
      for (int e = lane; e < npix; e += NT) {
        const int j = e >> lx, i = e & (xblk - 1);
        const int sx = bx + i, sy = by + j;
        const bool in = sx >= 0 && sx < P.clip_w && sy >= 0 && sy < P.clip_h;
        T.srcb[e] = in ? P.src[(size_t)sy*P.src_stride + sx] : 0;
      }
    }
    __syncthreads();
    for (int o = lane; o < NO; o += NT) T.part[o] = 0;       // this plane's sums (the chroma shift applies to the whole sum)
    const int fstep = 1 << shx;                                  // 4 (luma: phases 0, 4) or 2 (phases 0, 2, 4, 6)
    const float pw_1 = 1.0f/(float)pw;
    for (int fx = 0; fx < 8; fx += fstep) {
      // is any offset of the window at this horizontal phase?  (W >= 2 consecutive vectors: yes
      // for luma; chroma windows of R >= 2 cover all four)
      bool anyx = false;
      for (int ox = -R; ox <= R; ox++) anyx = anyx || ((((r.cx + ox)*(1 << shx)) & 7) == fx);
      if (!anyx) continue;
      // first stage over rows 0 .. sh - 1, columns 0 .. pw - 1 of the region
      for (int e = lane; e < sh*pw; e += NT) {
        const int rr = (int)(((float)e + 0.5f)*pw_1), c = e - rr*pw;
        const uint8_t *w = T.stage + rr*TT::SS + c;              // window columns c .. c + 5 = samples c - 2 .. c + 3
        int v;
        if (fx) {
          int sum = 0;
#pragma unroll
          for (int t = 0; t < 6; t++) sum += w[t]*MC_SUBPEL[fx][t];
          v = sum - (128 << 7);
        }
        else v = (w[2] << 7) - (128 << 7);
        T.hbuf[rr*TT::PW + c] = (int16_t)v;
      }
      __syncthreads();
      for (int fy = 0; fy < 8; fy += (1 << shy)) {
        bool anyy = false;
        for (int oy = -R; oy <= R; oy++) anyy = anyy || ((((r.cy + oy)*(1 << shy)) & 7) == fy);
        if (!anyy) continue;
        // second stage: region rows 0 .. ph - 1 from first-stage rows rr .. rr + 5
        for (int e = lane; e < ph*pw; e += NT) {
          const int rr = (int)(((float)e + 0.5f)*pw_1), c = e - rr*pw;
          const int16_t *h = T.hbuf + rr*TT::PW + c;
          int v;
          if (fy) {
            int sum = 0;
#pragma unroll
            for (int t = 0; t < 6; t++) sum += h[t*TT::PW]*MC_SUBPEL[fy][t];
            v = (sum + (1 << 13) + (128 << 14)) >> 14;
          }
          else v = (h[2*TT::PW] + (1 << 6) + (128 << 7)) >> 7;
          T.plane[rr*TT::PW + c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
        __syncthreads();
        // the offsets at this phase pair
        for (int oy = -R; oy <= R; oy++) {
          const int my = (r.cy + oy)*(1 << shy);
          if ((my & 7) != fy) continue;
          const int dy = (my >> 3) - iy0;
          for (int ox = -R; ox <= R; ox++) {
            const int mx = (r.cx + ox)*(1 << shx);
            if ((mx & 7) != fx) continue;
            const int dx = (mx >> 3) - ix0;
            int acc = 0;
            for (int e = lane; e < npix; e += NT) {
              const int j = e >> lx, i = e & (xblk - 1);
              const int sx = bx + i, sy = by + j;
              if (sx >= 0 && sx < P.clip_w && sy >= 0 && sy < P.clip_h) {
                acc += abs((int)T.plane[(dy + j)*TT::PW + dx + i] - (int)T.srcb[e]);
              }
            }
#pragma unroll
            for (int q = 32; q > 0; q >>= 1) acc += __shfl_xor(acc, q);
            if ((lane & 63) == 0) atomicAdd(&T.part[(oy + R)*W + ox + R], acc);
          }
        }
        __syncthreads();                                       // the plane is rewritten by the next phase pair
      }
    }
    for (int o = lane; o < NO; o += NT) T.tot[o] += T.part[o] >> P.shift;
    __syncthreads();                                           // stage / srcb / part are rewritten by the next plane
  }
  for (int o = lane; o < NO; o += NT) {
    const int mvx = r.cx + o%W - R, mvy = r.cy + o/W - R;
    const bool in = mvx >= r.xmin && mvx <= r.xmax && mvy >= r.ymin && mvy <= r.ymax;
    a.sad[(size_t)rec*NO + o] = in ? T.tot[o] : -1;
  }
}

// host side: a launch handles blocks of one size class (LDS sized by the class: McTiles)
static inline int mc_size_class(int log_sz) { return log_sz <= 4 ? 4 : log_sz == 5 ? 5 : 6; }

// calls launch(cls, first, count) for maximal runs of equal class; lists that change class too
// often go out as one launch of the largest class (correct for every size, only slower)
template <typename GetLog, typename Launch>
static inline void mc_launch_runs(int n, GetLog log_of, Launch launch) {
  int runs = 0, worst = 4;
  for (int i = 0, c = -1; i < n; i++) {
    const int ci = mc_size_class(log_of(i));
    worst = ci > worst ? ci : worst;
    if (ci != c) { runs++; c = ci; }
  }
  if (runs > 32) { launch(worst, 0, n); return; }
  for (int i = 0; i < n;) {
    const int c = mc_size_class(log_of(i));
    int j = i + 1;
    while (j < n && mc_size_class(log_of(j)) == c) j++;
    launch(c, i, j - i);
    i = j;
  }
}
