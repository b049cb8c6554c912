// Device-side building blocks of the transform path (gfx950).
//   - 4-point lapping pre/post filter   (reference src/filter.c:174-249)
//   - 2x2 Haar kernel                   (reference src/tf.h:34-45)
// The reversible DCT lifting steps live in gen_lift_dct.hpp (generated).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gen_lift_dct.hpp"

// Forward lapping across a block edge: x0,x1 | x2,x3.  Floor shifts, "+1 if
// positive" after each scale (src/filter.c:190-198).
__device__ __forceinline__ void lap4_pre(int32_t &x0, int32_t &x1, int32_t &x2,
                                         int32_t &x3) {
  int32_t d3 = x0 - x3;
  int32_t d2 = x1 - x2;
  int32_t s1 = x1 - (d2 >> 1);
  int32_t s0 = x0 - (d3 >> 1);
  d2 = (d2*85) >> 6;
  d2 += (d2 > 0);
  d3 = (d3*75) >> 6;
  d3 += (d3 > 0);
  d3 += (d2*-15 + 32) >> 6;
  d2 += (d3*33 + 32) >> 6;
  s0 += d3 >> 1;
  s1 += d2 >> 1;
  x0 = s0;
  x1 = s1;
  x2 = s1 - d2;
  x3 = s0 - d3;
}

// Same arithmetic with the 24-bit multiplier (full rate) for callers whose data is
// known to be small: the row-tile forward kernels keep the lapped tile as int16
// (|value| <= 6452, tools/range_analysis.py), so every product fits easily.  The
// generic form above compiles its "a*C + 32" steps to v_mad_u64_u32 / v_mul_lo_u32,
// both quarter rate.
__device__ __forceinline__ void lap4_pre24(int32_t &x0, int32_t &x1, int32_t &x2,
                                           int32_t &x3) {
  int32_t d3 = x0 - x3;
  int32_t d2 = x1 - x2;
  int32_t s1 = x1 - (d2 >> 1);
  int32_t s0 = x0 - (d3 >> 1);
  d2 = lift_mul24<85>(d2) >> 6;
  d2 += (d2 > 0);
  d3 = lift_mul24<75>(d3) >> 6;
  d3 += (d3 > 0);
  int32_t t;
  // multiplier and rounding term are inline constants: one v_mad_i32_i24 each
  asm("v_mad_i32_i24 %0, %1, -15, 32" : "=v"(t) : "v"(d2));
  d3 += t >> 6;
  asm("v_mad_i32_i24 %0, %1, 33, 32" : "=v"(t) : "v"(d3));
  d2 += t >> 6;
  s0 += d3 >> 1;
  s1 += d2 >> 1;
  x0 = s0;
  x1 = s1;
  x2 = s1 - d2;
  x3 = s0 - d3;
}

// Inverse lapping; undoes the scale with C truncating division
// (src/filter.c:237-241).
__device__ __forceinline__ void lap4_post(int32_t &x0, int32_t &x1, int32_t &x2,
                                          int32_t &x3) {
  int32_t d3 = x0 - x3;
  int32_t d2 = x1 - x2;
  int32_t s1 = x1 - (d2 >> 1);
  int32_t s0 = x0 - (d3 >> 1);
  d2 -= (d3*33 + 32) >> 6;
  d3 -= (d2*-15 + 32) >> 6;
  d3 = d3*64/75;
  d2 = d2*64/85;
  s0 += d3 >> 1;
  s1 += d2 >> 1;
  x0 = s0;
  x1 = s1;
  x2 = s1 - d2;
  x3 = s0 - d3;
}

// XCD-aware workgroup order (cdna guide T1).  Workgroups are dealt round-robin over the
// 8 XCDs, each with its own L2, while neighbouring tiles share cache lines (halo rows,
// unaligned or sub-line row segments): the plain order makes several XCDs fetch the
// same lines from HBM (rocprofv3 FETCH_SIZE showed 2-4x the input bytes).  Each XCD is
// given a CONTIGUOUS run of the (x, y, z) grid instead.  Pure speed: any mapping is
// correct.  Bijective for every grid size.
__device__ __forceinline__ void xcd_tile_coords(int &bx, int &by, int &bz) {
  const unsigned gx = gridDim.x, gy = gridDim.y;
  const unsigned total = gx*gy*gridDim.z;
  const unsigned lin = blockIdx.x + gx*(blockIdx.y + gy*blockIdx.z);
  const unsigned q = total/8, r = total%8, xcd = lin%8, idx = lin/8;
  const unsigned m = (xcd < r ? xcd*(q + 1) : r*(q + 1) + (xcd - r)*q) + idx;
  bx = m%gx;
  by = (m/gx)%gy;
  bz = m/(gx*gy);
}

__device__ __forceinline__ void haar2x2(int32_t &ll, int32_t &lh, int32_t &hl,
                                        int32_t &hh) {
  ll += hl;
  hh -= lh;
  int32_t m = (ll - hh) >> 1;
  lh = m - lh;
  hl = m - hl;
  ll -= lh;
  hh += hl;
}
