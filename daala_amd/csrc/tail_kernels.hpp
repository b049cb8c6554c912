// Decoder pixel-domain tail: directional deringing and keyframe
// bilinear smoothing between the frame post-filter and the 8-bit clamp
// (reference src/decode.c:1040-1155, src/filter.c:1655-2040).
//
// One workgroup per 32x32 deringing superblock, all planes of the superblock in
// turn (chroma reuses the directions found on luma).  Deringing reads only the
// unfiltered post-filter output (the reference's etmp copy, :1046-1056) and every
// superblock writes its own region, so superblocks are independent; smoothing
// works inside one 32x32 block.  Dering + smoothing + clamp are fused: the tile is
// read once (int32) and written once (u8).
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

// od_dir_find8 (src/filter.c:1655-1708) for the 16 8x8 blocks of a 32x32 luma tile, by
// the whole workgroup.  The reference accumulates, per block and direction d, the sums of
// the pixels along the <= 15 lines of that direction (partial[d][line]); integer sums are
// associative, so they are built here by 1024 (block, direction, row) work items - 4 per
// thread - that add their row's 8 pixels into LDS counters (pixels that fall on the same
// line are merged in registers first), instead of 16 threads walking 64 pixels x 8
// directions with a 120-entry private array.  part: [16][8][16] zero-initialised.
__device__ __forceinline__ void tail_dir_accumulate(const int16_t *in, int32_t *part, int t) {
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int item = t + 256*r;
    const int blk = item >> 6, d = (item >> 3) & 7, i = item & 7;
    const int16_t *row = in + (8*(blk >> 2) + i)*TAIL_BSTRIDE + 8*(blk & 3);
    int x[8];
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = row[j] >> 4;
    int32_t *p = part + (blk*8 + d)*16;
    switch (d) {
      case 0:
#pragma unroll
        for (int j = 0; j < 8; j++) atomicAdd(&p[i + j], x[j]);
        break;
      case 1:
#pragma unroll
        for (int j = 0; j < 4; j++) atomicAdd(&p[i + j], x[2*j] + x[2*j + 1]);
        break;
      case 2:
        atomicAdd(&p[i], x[0] + x[1] + x[2] + x[3] + x[4] + x[5] + x[6] + x[7]);
        break;
      case 3:
#pragma unroll
        for (int j = 0; j < 4; j++) atomicAdd(&p[3 + i - j], x[2*j] + x[2*j + 1]);
        break;
      case 4:
#pragma unroll
        for (int j = 0; j < 8; j++) atomicAdd(&p[7 + i - j], x[j]);
        break;
      case 5:
#pragma unroll
        for (int j = 0; j < 8; j++) atomicAdd(&p[3 - i/2 + j], x[j]);
        break;
      case 6:
#pragma unroll
        for (int j = 0; j < 8; j++) atomicAdd(&p[j], x[j]);
        break;
      default:
#pragma unroll
        for (int j = 0; j < 8; j++) atomicAdd(&p[i/2 + j], x[j]);
        break;
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

#ifndef TAIL_WAVES
#define TAIL_WAVES 6           /* min waves per SIMD asked of the compiler (VGPR cap): 92 -> <= 85 VGPRs, tail 1.61 -> 1.52 ms */
#endif
__global__ __launch_bounds__(256, TAIL_WAVES) void k_decode_tail(TailArgs a) {
  __shared__ int16_t in0[TAIL_BSTRIDE*TAIL_BSTRIDE];   // unfiltered tile + border
  __shared__ int16_t in1[TAIL_BSTRIDE*TAIL_BSTRIDE];   // border + direction-filtered interior
  __shared__ int32_t out[32*32];
  __shared__ int32_t rowsum[32];
  __shared__ int32_t part[16*8*16];                    // line sums, then costs in [..][0]
  __shared__ int dirs[16], vars[16], thresh[16];
  __shared__ int doff[16*3];                           // tap offsets of each block's direction
  __shared__ int sh_w;
  const int t = threadIdx.x;
  int sbx, sby, f;
  xcd_tile_coords(sbx, sby, f);        // 3-sample borders: neighbours share lines
  const bool enc_mode = a.flags == nullptr;
  const int flag = enc_mode ? 1 : a.flags[(size_t)f*a.nhsb*a.nvsb + sby*a.nhsb + sbx];
  const bool dering_on = a.q[0] > 0 && flag;
  const int sb_bsize = enc_mode ? 0 : a.bsize[(size_t)f*a.bsize_fstride + (size_t)(sby*4)*a.bstride + sbx*4];
  const bool smooth_on = !enc_mode && a.q[0] > 0 && a.is_keyframe && sb_bsize == 3;
  for (int pli = 0; pli < a.nplanes; pli++) {
    const int xdec = a.xdec[pli], ln = 5 - xdec, n = 1 << ln;
    const int w = a.fw >> xdec;
    const size_t porg = (size_t)f*a.fstride[pli] + (size_t)(sby << ln)*w + (sbx << ln);
    const int32_t *P = enc_mode ? nullptr : a.p[pli] + porg;
    const int16_t *P16 = enc_mode ? a.p16[pli] + porg : nullptr;
    if (dering_on) {
      const int bsz = 3 - xdec, nb = n >> bsz;            // 4 blocks per side
      // tile with a 3-sample border; outside the frame: OD_DERING_VERY_LARGE
      const int lo_i = -3*(sby != 0), hi_i = n + 3*(sby != a.nvsb - 1);
      const int lo_j = -3*(sbx != 0), hi_j = n + 3*(sbx != a.nhsb - 1);
      // only the (n + 6)^2 corner of the 38 x 38 buffers is ever read: a chroma tile loads
      // 22 x 22 cells, not 38 x 38
      const int tw = n + 6;
      for (int e = t; e < tw*tw; e += 256) {
        const int r = xdec ? e/22 : e/38;
        const int i = r - 3, j = e - r*tw - 3;
        int16_t v = TAIL_VERY_LARGE;
        if (i >= lo_i && i < hi_i && j >= lo_j && j < hi_j) {
          v = enc_mode ? P16[(ptrdiff_t)i*w + j] : (int16_t)P[(ptrdiff_t)i*w + j];
        }
        in0[r*TAIL_BSTRIDE + j + 3] = v;
        in1[r*TAIL_BSTRIDE + j + 3] = v;
      }
      __syncthreads();
      const int16_t *in = in0 + 3*TAIL_BSTRIDE + 3;
      if (pli == 0) {
        for (int e = t; e < 16*8*16; e += 256) part[e] = 0;
        __syncthreads();
        tail_dir_accumulate(in, part, t);
        __syncthreads();
        int cost = 0;
        if (t < 128) cost = tail_dir_cost(part + t*16, t & 7);
        __syncthreads();
        if (t < 128) part[t*16] = cost;
        __syncthreads();
        if (t < 16) {
          // first direction with the largest cost (strict '>' scan from best_cost = 0)
          int best_cost = 0, best_dir = 0;
#pragma unroll
          for (int d = 0; d < 8; d++) {
            const int c = part[(t*8 + d)*16];
            if (c > best_cost) { best_cost = c; best_dir = d; }
          }
          dirs[t] = best_dir;
          vars[t] = best_cost - part[(t*8 + ((best_dir + 4) & 7))*16];
        }
        __syncthreads();
        // the three tap offsets of every block's direction, once per superblock (chroma
        // reuses them): keeps the per-pixel loop free of constant-memory gathers
        if (t < 48) {
          const int blk = t/3, k = t - 3*blk;
          doff[t] = TAIL_DIR[dirs[blk]][k][0]*TAIL_BSTRIDE + TAIL_DIR[dirs[blk]][k][1];
        }
        if (t < 16) {
          int varsum = 0;
          for (int k = 0; k < 16; k++) varsum += vars[k];
          int v1 = vars[t] >> 6, v2 = varsum/1024;
          v1 = v1 > 32767 ? 32767 : v1;
          v2 = v2 > 32767 ? 32767 : v2;
          const uint32_t pr = (uint32_t)(v1*v2);
          int il = pr ? 32 - __clz(pr) : 0;
          il = il - 9;
          il = il < 0 ? 0 : il > 17 ? 17 : il;
          thresh[t] = a.thr[0]*TAIL_THRESH_Q8[il] >> 8;
        }
      }
      else if (t < 16) thresh[t] = a.thr[pli];
      __syncthreads();
      if (t < nb*nb) {
        // skipped neighbourhood => no filtering (src/filter.c:1898-1917)
        const int by = t/nb, bx = t%nb;
        const int sstride = a.fw/4;
        const uint8_t *bs = a.bskip[pli] + (size_t)f*a.bskip_fstride +
                            (size_t)(sby << (3 - xdec))*sstride + (sbx << (3 - xdec));
        const int xstart = sbx == 0 ? 0 : -1, ystart = sby == 0 ? 0 : -1;
        const int xend = (2 >> xdec) + (sbx != a.nhsb - 1), yend = (2 >> xdec) + (sby != a.nvsb - 1);
        int skip = 1;
        for (int i = ystart; i < yend; i++) {
          for (int j = xstart; j < xend; j++) {
            skip = skip && bs[(ptrdiff_t)((by << 1 >> xdec) + i)*sstride + (bx << 1 >> xdec) + j];
          }
        }
        if (skip) thresh[by*4 + bx] = 0;
      }
      __syncthreads();
      // direction filter (src/filter.c:1714-1740)
      for (int e = t; e < n*n; e += 256) {
        const int i = e >> ln, j = e & (n - 1);
        const int blk = (i >> bsz)*4 + (j >> bsz);
        const int th = thresh[blk];
        const int xx = in[i*TAIL_BSTRIDE + j];
        int sum = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const int off = doff[blk*3 + k];
          const int tap = k == 0 ? 3 : 2;
          const int p0 = in[i*TAIL_BSTRIDE + j + off] - xx;
          const int p1 = in[i*TAIL_BSTRIDE + j - off] - xx;
          if (abs(p0) < th) sum += tap*p0;
          if (abs(p1) < th) sum += tap*p1;
        }
        in1[(i + 3)*TAIL_BSTRIDE + j + 3] = (int16_t)(xx + ((sum + 8) >> 4));
      }
      __syncthreads();
      // orthogonal filter (src/filter.c:1753-1793)
      const int16_t *inf = in1 + 3*TAIL_BSTRIDE + 3;
      for (int e = t; e < n*n; e += 256) {
        const int i = e >> ln, j = e & (n - 1);
        const int blk = (i >> bsz)*4 + (j >> bsz);
        const int th = thresh[blk], dir = dirs[blk];
        const int offset = dir <= 4 ? TAIL_BSTRIDE : 1;
        const int yy = inf[i*TAIL_BSTRIDE + j];
        int athresh = th/3 + abs(yy - in[i*TAIL_BSTRIDE + j]);
        athresh = th < athresh ? th : athresh;
        int sum = 0, p;
        p = inf[i*TAIL_BSTRIDE + j + offset] - yy;   if (abs(p) < athresh) sum += p;
        p = inf[i*TAIL_BSTRIDE + j - offset] - yy;   if (abs(p) < athresh) sum += p;
        p = inf[i*TAIL_BSTRIDE + j + 2*offset] - yy; if (abs(p) < athresh) sum += p;
        p = inf[i*TAIL_BSTRIDE + j - 2*offset] - yy; if (abs(p) < athresh) sum += p;
        out[e] = (int16_t)(yy + ((3*sum + 8) >> 4));
      }
    }
    else {
      for (int e = t; e < n*n; e += 256) {
        const int i = e >> ln, j = e & (n - 1);
        out[e] = enc_mode ? (int32_t)P16[(size_t)i*w + j] : P[(size_t)i*w + j];
      }
    }
    __syncthreads();
    if (smooth_on) {
      // od_bilinear_smooth (src/filter.c:1952-2008) on the whole n x n tile
      const int32_t x00 = out[0], x01 = out[n - 1], x10 = out[(n - 1)*n];
      const int32_t x11 = out[(n - 1)*n + n - 1];
      const int32_t a00 = x00;
      int32_t a01 = x01 - x00, a10 = x10 - x00, a11 = x11 + x00 - x10 - x01;
      a01 += (a01 + n/2) >> ln;
      a10 += (a10 + n/2) >> ln;
      a11 += (2*a10 + n/2) >> ln;
      int shift = 2*4 + 2*ln - 16;
      shift = shift < 0 ? 0 : shift;
      __syncthreads();
      if (t < n) {
        int32_t partial = 0;
        for (int j = 0; j < n; j++) {
          const int32_t yv = a00 + ((j*a01 + t*a10 + (j*t*a11 >> ln) + n/2) >> ln);
          const int32_t dd = yv - out[t*n + j];
          partial += dd*dd;
        }
        rowsum[t] = partial >> shift;
      }
      __syncthreads();
      if (t == 0) {
        int32_t dist = 0;
        for (int i = 0; i < n; i++) dist += rowsum[i];
        dist += n/2;
        dist >>= 2*ln - shift;
        const int strength = (pli == 1 || pli == 2) ? 20 : 5;
        int wq = strength*a.q[pli]*a.q[pli]/(1 + 12*dist);
        wq = wq > 1024 ? 1024 : wq;
        sh_w = wq*wq >> 12;
      }
      __syncthreads();
      const int wq = sh_w;
      for (int e = t; e < n*n; e += 256) {
        const int i = e >> ln, j = e & (n - 1);
        const int32_t yv = a00 + ((j*a01 + i*a10 + (j*i*a11 >> ln) + n/2) >> ln);
        const int32_t xv = out[e];
        out[e] = xv - ((wq*(xv - yv) + 128) >> 8);
      }
      __syncthreads();
    }
    if (enc_mode) {
      int16_t *O = a.o16[pli] + (size_t)f*a.fstride[pli] + (size_t)(sby << ln)*w + (sbx << ln);
      for (int e = t; e < n*n; e += 256) {
        const int i = e >> ln, j = e & (n - 1);
        O[(size_t)i*w + j] = (int16_t)out[e];
      }
      __syncthreads();
      continue;
    }
    // od_coeff_to_ref_buf (src/state.c:1274-1300)
    uint8_t *R = a.rec[pli] + (size_t)f*a.fstride[pli] + (size_t)(sby << ln)*w + (sbx << ln);
    for (int e = t; e < n*n/4; e += 256) {
      const int i = (e*4) >> ln, j = (e*4) & (n - 1);
      uint32_t pk = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        int v = ((out[i*n + j + k] + 8) >> 4) + 128;
        v = v < 0 ? 0 : v > 255 ? 255 : v;
        pk |= (uint32_t)v << (8*k);
      }
      *reinterpret_cast<uint32_t *>(R + (size_t)i*w + j) = pk;
    }
    __syncthreads();
  }
}
