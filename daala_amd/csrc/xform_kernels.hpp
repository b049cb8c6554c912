// HIP kernels of the transform path (gfx950, wave64): dense block batches, the
// corner-centred frame post-filter, small helpers and the argument structs.  The
// frame pipeline's forward / inverse kernels live in xform_rt_kernels.hpp
// (the first, one-workgroup-per-superblock versions are in the git history:
// 0.44 ms vs 0.30 ms per 30-frame luma pyramid).
#pragma once
#include "xform_device.hpp"

// ---------------------------------------------------------------------------
// Dense block batch: nblocks independent N x N blocks (tests, per-call drop-ins,
// IEEE-1180 style runs).  256 threads = 256/N blocks per workgroup.
template <int N, bool INV>
__global__ __launch_bounds__(256) void k_dct_blocks(int32_t *__restrict__ out,
                                                    const int32_t *__restrict__ in,
                                                    int nblocks) {
  constexpr int BPW = 256 / N;
  constexpr int LD = N + 1;
  __shared__ int32_t A[BPW*N*LD];
  __shared__ int32_t Z[BPW*N*LD];
  const int t = threadIdx.x;
  const long b0 = (long)blockIdx.x*BPW;
  for (int e = t; e < BPW*N*N; e += 256) {
    int b = e/(N*N), r = (e/N)%N, c = e%N;
    A[(b*N + r)*LD + c] = (b0 + b < nblocks) ? in[(b0 + b)*N*N + r*N + c] : 0;
  }
  __syncthreads();
  const int b = t/N, i = t%N;
  int32_t v[N];
  if (!INV) {
    // columns of x -> rows of z -> (columns of z) -> rows of y   (src/dct.c:138-141)
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = A[(b*N + k)*LD + i];
    LiftDct<N>::fwd(v);
#pragma unroll
    for (int k = 0; k < N; k++) Z[(b*N + i)*LD + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = Z[(b*N + k)*LD + i];
    LiftDct<N>::fwd(v);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) A[(b*N + i)*LD + k] = v[k];
  }
  else {
    // rows of y -> columns of z -> (rows of z) -> columns of x   (src/dct.c:145-148)
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = A[(b*N + i)*LD + k];
    LiftDct<N>::inv(v);
#pragma unroll
    for (int k = 0; k < N; k++) Z[(b*N + k)*LD + i] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = Z[(b*N + i)*LD + k];
    LiftDct<N>::inv(v);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) A[(b*N + k)*LD + i] = v[k];
  }
  __syncthreads();
  for (int e = t; e < BPW*N*N; e += 256) {
    int bb = e/(N*N), r = (e/N)%N, c = e%N;
    if (b0 + bb < nblocks) out[(b0 + bb)*N*N + r*N + c] = A[(bb*N + r)*LD + c];
  }
}

// Multi-level 2-D Haar of dense blocks (lossless path, src/dct.c:1960-2026).
// One thread per block with a private copy: tiny, only used by the lossless
// configuration (not on the lossy hot path).
template <int LN>
__global__ void k_haar_blocks(int32_t *__restrict__ out, const int32_t *__restrict__ in,
                              int nblocks, int inverse) {
  constexpr int N = 1 << LN;
  long b = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  int32_t tmp[N*N];
  int32_t *y = out + b*N*N;
  for (int e = 0; e < N*N; e++) tmp[e] = in[b*N*N + e];
  if (!inverse) {
    for (int level = 0; level < LN; level++) {
      int np = N >> level >> 1;
      for (int i = 0; i < np; i++) {
        for (int j = 0; j < np; j++) {
          int32_t a = tmp[2*i*N + 2*j], bb = tmp[(2*i + 1)*N + 2*j];
          int32_t c = tmp[2*i*N + 2*j + 1], d = tmp[(2*i + 1)*N + 2*j + 1];
          haar2x2(a, bb, c, d);
          tmp[i*N + j] = a;
          y[i*N + j + np] = bb;
          y[(i + np)*N + j] = c;
          y[(i + np)*N + j + np] = d;
        }
      }
    }
    y[0] = tmp[0];
  }
  else {
    int32_t xo[N*N];
    xo[0] = tmp[0];
    for (int level = LN - 1; level >= 0; level--) {
      int np = 1 << (LN - 1 - level);
      for (int i = np - 1; i >= 0; i--) {
        for (int j = np - 1; j >= 0; j--) {
          int32_t a = xo[i*N + j], bb = tmp[i*N + j + np];
          int32_t c = tmp[(i + np)*N + j], d = tmp[(i + np)*N + j + np];
          haar2x2(a, bb, c, d);
          xo[2*i*N + 2*j] = a;
          xo[(2*i + 1)*N + 2*j] = bb;
          xo[2*i*N + 2*j + 1] = c;
          xo[(2*i + 1)*N + 2*j + 1] = d;
        }
      }
    }
    for (int e = 0; e < N*N; e++) y[e] = xo[e];
  }
}

// n-point lapping filters od_pre/post_filter{4,8,16,32} (src/filter.c:174-249, :306-440,
// :546-808, :879-1380; TYPE3 rotation structure, tables in filter_params.h).  One lane per
// vector; only the 4-point pair is on the codec's path, the others exist for parity
// with the reference's transform test tools.  Generic int32 multiplies (arbitrary input).
#include "filter_params.h"
template <int N>
__global__ void k_filter_vectors(int32_t *__restrict__ out, const int32_t *__restrict__ in,
                                 int nvec, int inverse) {
  constexpr int H = N/2;
  constexpr int P4[] = LAP_PARAMS4, P8[] = LAP_PARAMS8, P16[] = LAP_PARAMS16, P32[] = LAP_PARAMS32;
  const long v = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (v >= nvec) return;
  auto P = [&](int i) -> int { return N == 4 ? P4[i] : N == 8 ? P8[i] : N == 16 ? P16[i] : P32[i]; };
  int32_t x[N], t[N];
#pragma unroll
  for (int i = 0; i < N; i++) x[i] = in[v*N + i];
#pragma unroll
  for (int i = 0; i < H; i++) t[N - 1 - i] = x[i] - x[N - 1 - i];
#pragma unroll
  for (int i = 0; i < H; i++) t[i] = x[i] - (t[N - 1 - i] >> 1);
  if (!inverse) {
#pragma unroll
    for (int i = 0; i < H; i++) {
      if (P(i) != 64) {
        t[H + i] = (t[H + i]*P(i)) >> 6;
        t[H + i] += t[H + i] > 0;
      }
    }
#pragma unroll
    for (int j = N - 2; j >= H; j--) {
      t[j + 1] += (t[j]*P(H + (j - H)) + 32) >> 6;
      t[j] += (t[j + 1]*P(2*H - 1 + (j - H)) + 32) >> 6;
    }
  }
  else {
#pragma unroll
    for (int j = H; j <= N - 2; j++) {
      t[j] -= (t[j + 1]*P(2*H - 1 + (j - H)) + 32) >> 6;
      t[j + 1] -= (t[j]*P(H + (j - H)) + 32) >> 6;
    }
#pragma unroll
    for (int i = 0; i < H; i++) {
      if (P(i) != 64) t[H + i] = (t[H + i]*64)/P(i);      // C truncating division
    }
  }
#pragma unroll
  for (int i = 0; i < H; i++) t[i] += t[N - 1 - i] >> 1;
#pragma unroll
  for (int i = 0; i < H; i++) {
    out[v*N + i] = t[i];
    out[v*N + N - 1 - i] = t[i] - t[N - 1 - i];
  }
}

__global__ void k_filter4_vectors(int32_t *__restrict__ out, const int32_t *__restrict__ in,
                                  int nvec, int inverse) {
  long v = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (v >= nvec) return;
  int4 x = reinterpret_cast<const int4 *>(in)[v];
  if (inverse) lap4_post(x.x, x.y, x.z, x.w); else lap4_pre(x.x, x.y, x.z, x.w);
  reinterpret_cast<int4 *>(out)[v] = x;
}

// ---------------------------------------------------------------------------
// Tile geometry of the corner-centred post-filter kernel below.
template <int SB> struct SbTile {
  static constexpr int TH = SB*SB/4;        // threads per workgroup
  static constexpr int LD = SB + 1;         // padded LDS stride
};

struct FwdArgs {
  const uint8_t *pix;      // padded 8-bit plane, frame 0
  size_t pix_fstride;      // bytes between frames
  int pstride;
  int32_t *out;            // PYRAMID: level 0 plane; KNOWN: d plane (frame 0)
  size_t out_lstride;      // elements between levels (PYRAMID)
  size_t out_fstride;      // elements between frames
  const uint8_t *bsize;    // KNOWN: luma block-size map, 1 byte / 8x8
  size_t bsize_fstride;
  int bstride;
  int w, h, nhsb, nvsb;    // plane geometry (padded)
  int pic_w, pic_h;        // LUMA picture size (quirk, src/encode.c:1318-1320)
  int dec;
  int keyframe;
  int sby0;                // first superblock row of the launch (a strip of the frame; 0: whole frame)
};

struct InvArgs {
  const int32_t *d;        // coefficient plane, frame 0
  int32_t *c;              // work plane (spatial, lapped domain)
  size_t fstride;          // elements between frames (both planes)
  const uint8_t *bsize;
  size_t bsize_fstride;
  int bstride;
  int w, h, nhsb, nvsb;
  int pic_w, pic_h;
  int dec;
  // fused inverse (k_inverse_rt_fused / k_inverse_strips): 8-bit output and the edge strips
  uint8_t *rec;            // [frame][h][w]
  int16_t *rs;             // row strips  [frame][nvsb][4][w]: tile rows 0, 1, SB-2, SB-1 (int16, escape in c)
  int16_t *cs;             // column strips [frame][ntx][h][4]: tile columns 0, 1, tw-2, tw-1
  size_t rs_fstride, cs_fstride;
  int ntx;
};

struct PostArgs {
  const int32_t *c;
  size_t c_fstride;
  uint8_t *rec;
  size_t rec_fstride;
  int32_t *out32;            // CLAMP == false: post-filtered plane (same strides as c)
  int w, h, nhsb, nvsb;
};

// Frame post-filter + clamp (src/filter.c:1588-1646 + src/state.c:1274-1300).
// Workgroup (bx,by) owns the SB x SB region centred on superblock corner
// (bx*SB, by*SB), i.e. [bx*SB - SB/2, +SB): every SB boundary tap it needs lies
// inside that region, so no halo and no inter-workgroup dependency.  Vertical
// boundaries (row taps) first, then horizontal boundaries (column taps).
template <int SB, bool CLAMP = true>
__global__ __launch_bounds__(SB*SB/4) void k_postfilter_clamp(PostArgs a) {
  using T = SbTile<SB>;
  __shared__ int32_t X[SB*T::LD];
  const int t = threadIdx.x;
  int bx, by, f;
  xcd_tile_coords(bx, by, f);          // corner-centred tiles straddle cache lines of 4 neighbours
  const int x0 = bx*SB - SB/2, y0 = by*SB - SB/2;
  const int32_t *c = a.c + (size_t)f*a.c_fstride;
  uint8_t *rec = a.rec + (size_t)f*a.rec_fstride;
  const int r = t/(SB/4), c4 = (t%(SB/4))*4;
  const int gy = y0 + r, gx = x0 + c4;
  const bool inside = gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
  if (inside) {
    int4 v = *reinterpret_cast<const int4 *>(c + (size_t)gy*a.w + gx);
    X[r*T::LD + c4] = v.x; X[r*T::LD + c4 + 1] = v.y;
    X[r*T::LD + c4 + 2] = v.z; X[r*T::LD + c4 + 3] = v.w;
  }
  __syncthreads();
  const bool vbound = bx > 0 && bx < a.nhsb;   // internal x boundary
  const bool hbound = by > 0 && by < a.nvsb;
  if (vbound && t < SB) {
    int gyy = y0 + t;
    if (gyy >= 0 && gyy < a.h) {
      int32_t *p = X + t*T::LD + SB/2 - 2;
      lap4_post(p[0], p[1], p[2], p[3]);
    }
  }
  __syncthreads();
  if (hbound && t < SB) {
    int gxx = x0 + t;
    if (gxx >= 0 && gxx < a.w) {
      int32_t *p = X + (SB/2 - 2)*T::LD + t;
      lap4_post(p[0], p[T::LD], p[2*T::LD], p[3*T::LD]);
    }
  }
  __syncthreads();
  if (inside && !CLAMP) {
    // decoder flow: deringing / smoothing still follow (tail_kernels.hpp)
    int32_t *o = a.out32 + (size_t)f*a.c_fstride + (size_t)gy*a.w + gx;
    *reinterpret_cast<int4 *>(o) = make_int4(X[r*T::LD + c4], X[r*T::LD + c4 + 1],
                                             X[r*T::LD + c4 + 2], X[r*T::LD + c4 + 3]);
  }
  if (inside && CLAMP) {
    uint32_t pk = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      int v = ((X[r*T::LD + c4 + j] + 8) >> 4) + 128;
      v = v < 0 ? 0 : v > 255 ? 255 : v;
      pk |= (uint32_t)v << (8*j);
    }
    *reinterpret_cast<uint32_t *>(rec + (size_t)gy*a.w + gx) = pk;
  }
}

// CfL luma resample for 4:2:0 (src/intra.c:72-109).  One thread per output
// coefficient.  chroma_bs==0: 2x2 TF merge of four 4x4 luma blocks + scaling.
// mode: 0 = 4:2:0 (od_tf_up_hv_lp + CfL scaling), 1 = horizontally decimated chroma
// (od_tf_up_h_lp, src/tf.c:38-58), 2 = vertically decimated (od_tf_up_v_lp, :60-80).
__global__ void k_resample_luma_420(int32_t *__restrict__ pred, const int32_t *__restrict__ luma,
                                    int lstride, const int32_t *__restrict__ luma_off,
                                    int nblk, int bs, int chroma_bs, int mode) {
  const int n = 4 << bs;
  long e = (long)blockIdx.x*blockDim.x + threadIdx.x;
  if (e >= (long)nblk*n*n) return;
  const int b = e/(n*n), y = (e/n)%n, x = e%n;
  const int32_t *src = luma + luma_off[b];
  if (chroma_bs != 0) {
    pred[e] = src[y*lstride + x];
    return;
  }
  if (mode == 1) {
    // pair (ll, lh) = src[y][xx], src[y][xx + n]; lh = ll - lh; ll -= LIFT_HALF(lh);
    // outputs at 2*xx + sw (ll) and 2*xx + 1 - sw (lh), sw = xx & 1
    const int xx = x >> 1;
    int32_t ll = src[y*lstride + xx], lh = src[y*lstride + xx + n];
    lh = ll - lh;
    ll -= LIFT_HALF(lh);
    pred[e] = (((x & 1) ^ (xx & 1)) == 0) ? ll : lh;
    return;
  }
  if (mode == 2) {
    const int yy = y >> 1;
    int32_t ll = src[yy*lstride + x], hl = src[(yy + n)*lstride + x];
    hl = ll - hl;
    ll -= LIFT_HALF(hl);
    pred[e] = (((y & 1) ^ (yy & 1)) == 0) ? ll : hl;
    return;
  }
  // od_tf_up_hv_lp(dst, src, dx=n, dy=n, n): output (oy,ox) comes from the 2x2
  // group at (yy,xx) = (oy>>1, ox>>1) of the four source blocks.
  const int yy = y >> 1, xx = x >> 1;
  int32_t ll = src[yy*lstride + xx], lh = src[yy*lstride + xx + n];
  int32_t hl = src[(yy + n)*lstride + xx], hh = src[(yy + n)*lstride + xx + n];
  haar2x2(ll, hl, lh, hh);
  const int vs = yy & 1, hs = xx & 1;
  const int ry = (y & 1) ^ vs, rx = (x & 1) ^ hs;   // 0 => "low" output of the pair
  int32_t v = ry == 0 ? (rx == 0 ? ll : lh) : (rx == 0 ? hl : hh);
  const int16_t S[4][4] = {{128, 128, 100, 36}, {128, 80, 71, 35},
                           {100, 71, 35, 31}, {36, 35, 31, 18}};
  if (y < 4 && x < 4) v = (S[x][y]*v + 64) >> 7;
  pred[e] = v;
}

// Calibration kernels for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters
// (MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE under-reports wide
// streaming reads by 2x and other widths are uncalibrated): stream a buffer of
// known size with the SAME access widths the transform kernels use - dword loads
// (the u8 pixel tile), int4 loads (coefficient tiles) and int4 stores.
__global__ void k_calib_read_dword(const uint32_t *__restrict__ src, size_t n, uint32_t *sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x)
    acc ^= src[i];
  if (acc == 0x12345678u) *sink = acc;
}

__global__ void k_calib_read_int4(const int4 *__restrict__ src, size_t n, uint32_t *sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x) {
    int4 v = src[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

__global__ void k_calib_write_int4(int4 *__restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x)
    dst[i] = make_int4((int)i, 1, 2, 3);
}
