// Lossless frames (BASELINE configs[4]): the reference codes
// them with a multi-level 2-D Haar of whole superblocks - 32x32 luma, 16x16 4:2:0
// chroma - without lapping or DCT and with coefficient shift 0
// (src/encode.c:3002,3090-3092,2427-2430; od_haar / od_haar_inv src/dct.c:1960-2026;
// od_ref_buf_to_coeff / od_coeff_to_ref_buf with lossless_p, src/state.c:1209,1274).
//
// One wave per superblock, grid (nhsb, nvsb, frames).  The tile lives in LDS
// (stride SB+1); every level maps one 2x2 group to one lane-iteration.  The low band
// ping-pongs between two tiles because level l reads (2i,2j) while writing (i,j).
// HBM-bound: forward reads 1 B and writes 4 B per sample, inverse the opposite
// (5 B/sample algorithmic either way); rows of a superblock are 32 B (u8) / 128 B
// (int32) contiguous segments.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "xform_device.hpp"

struct HaarArgs {
  uint8_t *pix;          // [frame][h][w] input (forward) or output (inverse)
  int32_t *d;            // [frame][h][w] coefficients
  size_t fstride;        // samples per frame plane
  int w;                 // plane stride
};

template <int SB>
__global__ __launch_bounds__(64) void k_haar_forward_plane(HaarArgs a) {
  constexpr int LD = SB + 1;
  constexpr int LN = SB == 32 ? 5 : 4;
  __shared__ int32_t T[2][SB*LD];
  __shared__ int32_t Y[SB*LD];
  const int lane = threadIdx.x;
  int sbx, sby, fr;
  xcd_tile_coords(sbx, sby, fr);   // 4 adjacent luma SBs share each 128-byte line of the 8-bit plane
  const size_t org = (size_t)fr*a.fstride + (size_t)(sby*SB)*a.w + sbx*SB;
  // load: dword = 4 pixels; SB*SB/4 dwords over 64 lanes
  for (int e = lane; e < SB*SB/4; e += 64) {
    const int r = e/(SB/4), c4 = e%(SB/4);
    const uint32_t v = *reinterpret_cast<const uint32_t *>(a.pix + org + (size_t)r*a.w + 4*c4);
#pragma unroll
    for (int k = 0; k < 4; k++) T[0][r*LD + 4*c4 + k] = (int32_t)((v >> (8*k)) & 255) - 128;
  }
  __syncthreads();
  int cur = 0;
#pragma unroll
  for (int level = 0; level < LN; level++) {
    const int np = SB >> level >> 1;
    for (int e = lane; e < np*np; e += 64) {
      const int i = e/np, j = e%np;
      int32_t ll = T[cur][(2*i)*LD + 2*j], lh = T[cur][(2*i + 1)*LD + 2*j];
      int32_t hl = T[cur][(2*i)*LD + 2*j + 1], hh = T[cur][(2*i + 1)*LD + 2*j + 1];
      haar2x2(ll, lh, hl, hh);
      T[cur ^ 1][i*LD + j] = ll;
      Y[i*LD + j + np] = lh;
      Y[(i + np)*LD + j] = hl;
      Y[(i + np)*LD + j + np] = hh;
    }
    cur ^= 1;
    __syncthreads();
  }
  if (lane == 0) Y[0] = T[cur][0];
  __syncthreads();
  for (int e = lane; e < SB*SB/4; e += 64) {
    const int r = e/(SB/4), c4 = e%(SB/4);
    int4 v = make_int4(Y[r*LD + 4*c4], Y[r*LD + 4*c4 + 1], Y[r*LD + 4*c4 + 2], Y[r*LD + 4*c4 + 3]);
    *reinterpret_cast<int4 *>(a.d + org + (size_t)r*a.w + 4*c4) = v;
  }
}

template <int SB>
__global__ __launch_bounds__(64) void k_haar_inverse_plane(HaarArgs a) {
  constexpr int LD = SB + 1;
  constexpr int LN = SB == 32 ? 5 : 4;
  __shared__ int32_t X[2][SB*LD];
  __shared__ int32_t Y[SB*LD];
  const int lane = threadIdx.x;
  int sbx, sby, fr;
  xcd_tile_coords(sbx, sby, fr);   // 4 adjacent luma SBs share each 128-byte line of the 8-bit plane
  const size_t org = (size_t)fr*a.fstride + (size_t)(sby*SB)*a.w + sbx*SB;
  for (int e = lane; e < SB*SB/4; e += 64) {
    const int r = e/(SB/4), c4 = e%(SB/4);
    const int4 v = *reinterpret_cast<const int4 *>(a.d + org + (size_t)r*a.w + 4*c4);
    Y[r*LD + 4*c4] = v.x;
    Y[r*LD + 4*c4 + 1] = v.y;
    Y[r*LD + 4*c4 + 2] = v.z;
    Y[r*LD + 4*c4 + 3] = v.w;
  }
  __syncthreads();
  if (lane == 0) X[0][0] = Y[0];
  __syncthreads();
  int cur = 0;
#pragma unroll
  for (int level = LN - 1; level >= 0; level--) {
    const int np = 1 << (LN - 1 - level);
    for (int e = lane; e < np*np; e += 64) {
      const int i = e/np, j = e%np;
      int32_t ll = X[cur][i*LD + j], lh = Y[i*LD + j + np];
      int32_t hl = Y[(i + np)*LD + j], hh = Y[(i + np)*LD + j + np];
      haar2x2(ll, lh, hl, hh);
      X[cur ^ 1][(2*i)*LD + 2*j] = ll;
      X[cur ^ 1][(2*i + 1)*LD + 2*j] = lh;
      X[cur ^ 1][(2*i)*LD + 2*j + 1] = hl;
      X[cur ^ 1][(2*i + 1)*LD + 2*j + 1] = hh;
    }
    cur ^= 1;
    __syncthreads();
  }
  for (int e = lane; e < SB*SB/4; e += 64) {
    const int r = e/(SB/4), c4 = e%(SB/4);
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int32_t p = X[cur][r*LD + 4*c4 + k] + 128;       // shift 0: OD_CLAMP255(c + 128)
      p = p < 0 ? 0 : p > 255 ? 255 : p;
      v |= (uint32_t)p << (8*k);
    }
    *reinterpret_cast<uint32_t *>(a.pix + org + (size_t)r*a.w + 4*c4) = v;
  }
}
