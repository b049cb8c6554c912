// Lossless frames (BASELINE configs[4]): the reference codes
// them with a multi-level 2-D Haar of whole superblocks - 32x32 luma, 16x16 4:2:0
// chroma - without lapping or DCT and with coefficient shift 0
// (src/encode.c:3002,3090-3092,2427-2430; od_haar / od_haar_inv src/dct.c:1960-2026;
// od_ref_buf_to_coeff / od_coeff_to_ref_buf with lossless_p, src/state.c:1209,1274).
//
// One wave per superblock, grid (nhsb, nvsb, frames).  The tile lives in LDS
// (stride SB+1); every level maps one 2x2 group to one lane-iteration.  The low band
// ping-pongs between two tiles because level l reads (2i,2j) while writing (i,j); the second
// tile needs only half the rows (10.6 KB of LDS per wave instead of 12.7: 15 waves per CU).
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
  // the low band ping-pongs between the full tile and a half-height one (level 0 writes 16x16)
  __shared__ int32_t T0[SB*LD];
  __shared__ int32_t T1[(SB/2)*LD];
  __shared__ int32_t Y[SB*LD];
  const int lane = threadIdx.x;
  int sbx, sby, fr;
  xcd_tile_coords(sbx, sby, fr);   // 4 adjacent luma SBs share each 128-byte line of the 8-bit plane
  const size_t org = (size_t)fr*a.fstride + (size_t)(sby*SB)*a.w + sbx*SB;
  // load: dword = 4 pixels; SB*SB/4 dwords over 64 lanes, all of a lane's loads issued before the
  // first is unpacked (the rolled loop waited for each in turn: waves parked 76 % of their life)
  constexpr int NLD = SB*SB/4/64;
  uint32_t pv[NLD];
#pragma unroll
  for (int q = 0; q < NLD; q++) {
    const int e = lane + 64*q;
    pv[q] = *reinterpret_cast<const uint32_t *>(a.pix + org + (size_t)(e/(SB/4))*a.w + 4*(e%(SB/4)));
  }
#pragma unroll
  for (int q = 0; q < NLD; q++) {
    const int e = lane + 64*q;
    const int r = e/(SB/4), c4 = e%(SB/4);
#pragma unroll
    for (int k = 0; k < 4; k++) T0[r*LD + 4*c4 + k] = (int32_t)((pv[q] >> (8*k)) & 255) - 128;
  }
  __syncthreads();
  int cur = 0;
#pragma unroll
  for (int level = 0; level < LN; level++) {
    const int np = SB >> level >> 1;
    int32_t *src = cur ? T1 : T0, *dst = cur ? T0 : T1;
    for (int e = lane; e < np*np; e += 64) {
      const int i = e/np, j = e%np;
      int32_t ll = src[(2*i)*LD + 2*j], lh = src[(2*i + 1)*LD + 2*j];
      int32_t hl = src[(2*i)*LD + 2*j + 1], hh = src[(2*i + 1)*LD + 2*j + 1];
      haar2x2(ll, lh, hl, hh);
      dst[i*LD + j] = ll;
      Y[i*LD + j + np] = lh;
      Y[(i + np)*LD + j] = hl;
      Y[(i + np)*LD + j + np] = hh;
    }
    cur ^= 1;
    __syncthreads();
  }
  if (lane == 0) Y[0] = (cur ? T1 : T0)[0];
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
  // the low band grows 1x1 -> SBxSB through LN levels, ping-ponging between a half-height tile
  // and the full one; the parity is chosen so that the last level (the only one that needs SB
  // rows) writes the full tile
  __shared__ int32_t XF[SB*LD];
  __shared__ int32_t XH[(SB/2)*LD];
  __shared__ int32_t Y[SB*LD];
  const int lane = threadIdx.x;
  int sbx, sby, fr;
  xcd_tile_coords(sbx, sby, fr);   // 4 adjacent luma SBs share each 128-byte line of the 8-bit plane
  const size_t org = (size_t)fr*a.fstride + (size_t)(sby*SB)*a.w + sbx*SB;
  constexpr int NLD = SB*SB/4/64;
  int4 cv[NLD];
#pragma unroll
  for (int q = 0; q < NLD; q++) {      // all loads first (see the forward kernel)
    const int e = lane + 64*q;
    cv[q] = *reinterpret_cast<const int4 *>(a.d + org + (size_t)(e/(SB/4))*a.w + 4*(e%(SB/4)));
  }
#pragma unroll
  for (int q = 0; q < NLD; q++) {
    const int e = lane + 64*q;
    const int r = e/(SB/4), c4 = e%(SB/4);
    Y[r*LD + 4*c4] = cv[q].x;
    Y[r*LD + 4*c4 + 1] = cv[q].y;
    Y[r*LD + 4*c4 + 2] = cv[q].z;
    Y[r*LD + 4*c4 + 3] = cv[q].w;
  }
  __syncthreads();
  // LN levels: the write of step s (s = 0 .. LN - 1) goes to XF when (LN - 1 - s) is even
  int cur = (LN & 1) ? 1 : 0;          // 0: the low band is in XF, 1: in XH; the first write goes to the other
  if (lane == 0) (cur ? XH : XF)[0] = Y[0];
  __syncthreads();
#pragma unroll
  for (int level = LN - 1; level >= 0; level--) {
    const int np = 1 << (LN - 1 - level);
    const int32_t *src = cur ? XH : XF;
    int32_t *dst = cur ? XF : XH;
    for (int e = lane; e < np*np; e += 64) {
      const int i = e/np, j = e%np;
      int32_t ll = src[i*LD + j], lh = Y[i*LD + j + np];
      int32_t hl = Y[(i + np)*LD + j], hh = Y[(i + np)*LD + j + np];
      haar2x2(ll, lh, hl, hh);
      dst[(2*i)*LD + 2*j] = ll;
      dst[(2*i + 1)*LD + 2*j] = lh;
      dst[(2*i)*LD + 2*j + 1] = hl;
      dst[(2*i + 1)*LD + 2*j + 1] = hh;
    }
    cur ^= 1;
    __syncthreads();
  }
  const int32_t *X = cur ? XH : XF;    // == XF: LN steps from the chosen parity end in the full tile
  for (int e = lane; e < SB*SB/4; e += 64) {
    const int r = e/(SB/4), c4 = e%(SB/4);
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int32_t p = X[r*LD + 4*c4 + k] + 128;            // shift 0: OD_CLAMP255(c + 128)
      p = p < 0 ? 0 : p > 255 ? 255 : p;
      v |= (uint32_t)p << (8*k);
    }
    *reinterpret_cast<uint32_t *>(a.pix + org + (size_t)r*a.w + 4*c4) = v;
  }
}
