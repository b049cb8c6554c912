// Row-tile transform kernels (gfx950, wave64) - the fast path of the frame
// pipeline.
//
// One WAVE (a 64-thread workgroup) owns a tile of 64 x SB samples = 64/SB
// horizontally adjacent superblocks of one plane of one frame:
//   * no inter-wave barriers at all (the PMC profile of the first, 4-waves-per-SB
//     version showed 72 % of wave cycles parked at s_barrier / s_waitcnt);
//   * every transform level has >= 64 1-D transforms per pass, so all lanes work
//     (a single 32x32 SB only has 32 columns);
//   * HBM rows are 256 B contiguous (64 int32) instead of 128 B;
//   * the two separable passes share ONE stride-65 LDS tile: pass 2 reads its
//     column into VGPRs and writes its output row back in place;
//   * the lifting multiplies use the 24-bit multiplier (v_mad_i32_i24, full rate;
//     v_mul_lo_u32 is quarter rate): exact because every operand is < 2^18 for
//     pixel-driven data (tools/range_analysis.py), and range-checked per tile on
//     the inverse path.
#pragma once
#include "xform_kernels.hpp"

template <int SB> struct RowTile {
  static constexpr int W = 64;               // tile width (samples)
  static constexpr int NSB = W/SB;           // superblocks per tile
  static constexpr int LDZ = W + 1;          // coefficient tile stride
  static constexpr int HAW = W + 4;          // lapped tile incl. 2-sample halo
  static constexpr int HAH = SB + 4;
  static constexpr int LDA = W + 5;
};

// Store an SB x 64 int32 tile (stride 65 in LDS) to a plane; 4 rows x 256 B per
// wave instruction, columns beyond the plane are masked.
template <int SB>
__device__ __forceinline__ void rt_store_tile(int32_t *__restrict__ dst, int w, int x0,
                                              const int32_t *Z) {
  using T = RowTile<SB>;
  const int lane = threadIdx.x;
  const int r0 = lane & 3, c4 = (lane >> 2)*4;
  if (x0 + c4 < w) {
#pragma unroll
    for (int it = 0; it < SB/4; it++) {
      const int r = it*4 + r0;
      const int32_t *p = Z + r*T::LDZ + c4;
      *reinterpret_cast<int4 *>(dst + (size_t)r*w + c4) = make_int4(p[0], p[1], p[2], p[3]);
    }
  }
}

template <int SB>
__device__ __forceinline__ void rt_load_tile(int32_t *Z, const int32_t *__restrict__ src, int w,
                                             int x0) {
  using T = RowTile<SB>;
  const int lane = threadIdx.x;
  const int r0 = lane & 3, c4 = (lane >> 2)*4;
#pragma unroll
  for (int it = 0; it < SB/4; it++) {
    const int r = it*4 + r0;
    int4 v = make_int4(0, 0, 0, 0);
    if (x0 + c4 < w) v = *reinterpret_cast<const int4 *>(src + (size_t)r*w + c4);
    int32_t *p = Z + r*T::LDZ + c4;
    p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
  }
}

// Forward path of one row tile.  PYRAMID: all blocks of all levels (stores one
// plane per level).  KNOWN: only the quadtree leaves + keyframe DC merge
// (od_compute_dcts, src/encode.c:1286-1343).
template <int SB, int NLEV, bool KNOWN>
__global__ __launch_bounds__(64) void k_forward_rt(FwdArgs a) {
  using T = RowTile<SB>;
  __shared__ int32_t A[T::HAH*T::LDA];
  __shared__ int32_t Z[SB*T::LDZ];
  __shared__ uint8_t bsz[16*T::NSB];
  const int lane = threadIdx.x;
  const int tx = blockIdx.x, sby = blockIdx.y, f = blockIdx.z;
  const int x0 = tx*T::W, y0 = sby*SB;
  const int sbx0 = tx*T::NSB;                          // first superblock of the tile
  const int nsb = min(T::NSB, a.nhsb - sbx0);          // superblocks really present
  const int dec = a.dec;
  if (KNOWN) {
    for (int e = lane; e < 16*T::NSB; e += 64) {
      const int s = e >> 4, c = e & 15;
      bsz[e] = s < nsb ? a.bsize[(size_t)f*a.bsize_fstride +
                                 (size_t)(sby*4 + (c >> 2))*a.bstride + (sbx0 + s)*4 + (c & 3)]
                       : 3;
    }
  }
  {
    // A1: 8-bit pixels -> (p - 128) << 4, dword loads starting 4 bytes left of the tile
    const uint8_t *pix = a.pix + (size_t)f*a.pix_fstride;
    constexpr int DW = T::HAW/4 + 1;                   // 18 dwords cover x0-4 .. x0+67
    for (int e = lane; e < T::HAH*DW; e += 64) {
      const int ty = e/DW, dx = e%DW;
      const int gy = min(max(y0 - 2 + ty, 0), a.h - 1);
      const int gx = min(max(x0 - 4 + dx*4, 0), a.w - 4);
      const uint32_t v = *reinterpret_cast<const uint32_t *>(pix + (size_t)gy*a.pstride + gx);
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int txx = dx*4 + b - 2;
        if (txx >= 0 && txx < T::HAW) {
          A[ty*T::LDA + txx] = ((int32_t)((v >> (8*b)) & 255) - 128) << 4;
        }
      }
    }
  }
  __syncthreads();
  // A4: frame lapping.  Horizontal SB boundaries first (vertical taps over every
  // column of the halo'd tile), then vertical boundaries (src/filter.c:1566-1584).
  for (int c = lane; c < T::HAW; c += 64) {
    if (sby > 0) {
      int32_t *p = A + c;
      lap4_pre(p[0], p[T::LDA], p[2*T::LDA], p[3*T::LDA]);
    }
    if (sby < a.nvsb - 1) {
      int32_t *p = A + SB*T::LDA + c;
      lap4_pre(p[0], p[T::LDA], p[2*T::LDA], p[3*T::LDA]);
    }
  }
  __syncthreads();
  for (int e = lane; e < (T::NSB + 1)*SB; e += 64) {
    const int s = e/SB, row = e%SB;                    // boundary left of tile SB s
    const int gb = sbx0 + s;                           // global boundary index
    if (gb >= 1 && gb <= a.nhsb - 1 && s <= nsb) {
      int32_t *p = A + (2 + row)*T::LDA + s*SB;
      lap4_pre(p[0], p[1], p[2], p[3]);
    }
  }
  __syncthreads();
  auto cell = [&](int byi, int bxt, int n) -> int {    // max(obs, dec) of a tile block
    const int s = (bxt*n)/SB, bxs = bxt - s*(SB/n);
    const int nl = n << dec;
    const int o = bsz[s*16 + ((byi*nl) >> 3)*4 + ((bxs*nl) >> 3)];
    return o > dec ? o : dec;
  };
  int32_t *out = a.out + (size_t)f*a.out_fstride + (size_t)y0*a.w + x0;
#define RT_FWD_LEVEL(K)                                                                \
  if constexpr (K < NLEV) {                                                            \
    constexpr int N = SB >> K;                                                         \
    constexpr int NBY = SB/N;                                                          \
    const int col = lane, bxt = col/N, i = col%N;                                      \
    const bool present = (bxt*N)/SB < nsb;                                             \
    for (int byi = 0; byi < NBY; byi++) {                                              \
      const bool go = present && (!KNOWN || cell(byi, bxt, N) == 3 - K);               \
      int32_t v[N];                                                                    \
      if (go) {                                                                        \
        _Pragma("unroll") for (int k = 0; k < N; k++)                                  \
          v[k] = A[(2 + byi*N + k)*T::LDA + 2 + col];                                  \
        LiftDct<N, true>::fwd(v);                                                      \
        _Pragma("unroll") for (int k = 0; k < N; k++)                                  \
          Z[(byi*N + i)*T::LDZ + bxt*N + k] = v[k];                                    \
      }                                                                                \
      __syncthreads();                                                                 \
      if (go) {                                                                        \
        _Pragma("unroll") for (int k = 0; k < N; k++)                                  \
          v[k] = Z[(byi*N + k)*T::LDZ + col];                                          \
        LiftDct<N, true>::fwd(v);                                                      \
      }                                                                                \
      __syncthreads();                                                                 \
      if (go) {                                                                        \
        _Pragma("unroll") for (int k = 0; k < N; k++)                                  \
          Z[(byi*N + i)*T::LDZ + bxt*N + k] = v[k];                                    \
      }                                                                                \
    }                                                                                  \
    __syncthreads();                                                                   \
    if (!KNOWN) {                                                                      \
      rt_store_tile<SB>(out + (size_t)K*a.out_lstride, a.w, x0, Z);                    \
      __syncthreads();                                                                 \
    }                                                                                  \
    if constexpr (K + 1 < NLEV) {                                                      \
      /* A5: split lapping of the blocks that are split at this level */              \
      for (int byi = 0; byi < NBY; byi++) {                                            \
        const bool sp = present && (!KNOWN || cell(byi, bxt, N) < 3 - K);              \
        if (sp && (tx*(T::W/N) + bxt + 1)*N <= a.pic_w) {                              \
          int32_t *p = A + (2 + byi*N + N/2 - 2)*T::LDA + 2 + col;                     \
          lap4_pre(p[0], p[T::LDA], p[2*T::LDA], p[3*T::LDA]);                         \
        }                                                                              \
      }                                                                                \
      __syncthreads();                                                                 \
      for (int e = lane; e < SB*(T::W/N); e += 64) {                                   \
        const int bx2 = e/SB, row = e%SB, by2 = row/N;                                 \
        const bool sp = (bx2*N)/SB < nsb && (!KNOWN || cell(by2, bx2, N) < 3 - K);     \
        if (sp && (sby*NBY + by2 + 1)*N <= a.pic_h) {                                  \
          int32_t *p = A + (2 + row)*T::LDA + 2 + bx2*N + N/2 - 2;                     \
          lap4_pre(p[0], p[1], p[2], p[3]);                                            \
        }                                                                              \
      }                                                                                \
      __syncthreads();                                                                 \
    }                                                                                  \
  }
  RT_FWD_LEVEL(0)
  RT_FWD_LEVEL(1)
  RT_FWD_LEVEL(2)
  RT_FWD_LEVEL(3)
#undef RT_FWD_LEVEL
  if (KNOWN) {
    if (a.keyframe) {
      // Haar merge of the four child DCs of every split block, finest first
      for (int k = NLEV - 2; k >= 0; k--) {
        const int n = SB >> k, nby = SB/n, nbx = T::W/n, hh = n/2;
        if (lane < nby*nbx) {
          const int byi = lane/nbx, bxt = lane%nbx;
          if ((bxt*n)/SB < nsb && cell(byi, bxt, n) < 3 - k) {
            int32_t *p = Z + (byi*n)*T::LDZ + bxt*n;
            int32_t q0 = p[0], q1 = p[hh], q2 = p[hh*T::LDZ], q3 = p[hh*T::LDZ + hh];
            haar2x2(q0, q2, q1, q3);
            p[0] = q0; p[hh] = q1; p[hh*T::LDZ] = q2; p[hh*T::LDZ + hh] = q3;
          }
        }
        __syncthreads();
      }
    }
    rt_store_tile<SB>(out, a.w, x0, Z);
  }
}

// Inverse path of one row tile: iDCT of every quadtree leaf, split post-filters
// finest first (src/decode.c:843-866, src/filter.c:1537-1552).  The lifting
// multiplies use the 24-bit multiplier when every coefficient of the tile is
// <= 2^18 in magnitude (tools/range_analysis.py: safe up to 2^18.17 for the
// 32-point transform); otherwise the generic int32 form runs.  Same results.
template <int SB, int NLEV, bool M24>
__device__ __forceinline__ void rt_inverse_body(int32_t *Z, const uint8_t *bsz, int nsb, int dec,
                                                int tx, int sby, int pic_w, int pic_h) {
  using T = RowTile<SB>;
  const int lane = threadIdx.x;
  auto cell = [&](int byi, int bxt, int n) -> int {
    const int s = (bxt*n)/SB, bxs = bxt - s*(SB/n);
    const int nl = n << dec;
    const int o = bsz[s*16 + ((byi*nl) >> 3)*4 + ((bxs*nl) >> 3)];
    return o > dec ? o : dec;
  };
#define RT_INV_LEVEL(K)                                                                \
  if constexpr (K < NLEV) {                                                            \
    constexpr int N = SB >> K;                                                         \
    /* item = (block column bxt, row): bxt slowest, so 64 consecutive items cover    */ \
    /* whole blocks and both in-place passes stay inside the items' own blocks        */ \
    for (int e = lane; e < SB*(T::W/N); e += 64) {                                     \
      const int bxt = e/SB, row = e%SB, byi = row/N, i = row%N;                        \
      const bool go = (bxt*N)/SB < nsb && cell(byi, bxt, N) == 3 - K;                  \
      int32_t v[N];                                                                    \
      if (go) {                                                                        \
        _Pragma("unroll") for (int k = 0; k < N; k++) v[k] = Z[row*T::LDZ + bxt*N + k];\
        LiftDct<N, M24>::inv(v);                                                       \
      }                                                                                \
      __syncthreads();                                                                 \
      if (go) {                                                                        \
        _Pragma("unroll") for (int k = 0; k < N; k++)                                  \
          Z[(byi*N + k)*T::LDZ + bxt*N + i] = v[k];                                    \
      }                                                                                \
      __syncthreads();                                                                 \
      if (go) {                                                                        \
        _Pragma("unroll") for (int k = 0; k < N; k++) v[k] = Z[row*T::LDZ + bxt*N + k];\
        LiftDct<N, M24>::inv(v);                                                       \
      }                                                                                \
      __syncthreads();                                                                 \
      if (go) {                                                                        \
        _Pragma("unroll") for (int k = 0; k < N; k++)                                  \
          Z[(byi*N + k)*T::LDZ + bxt*N + i] = v[k];                                    \
      }                                                                                \
      __syncthreads();                                                                 \
    }                                                                                  \
  }
  RT_INV_LEVEL(0)
  RT_INV_LEVEL(1)
  RT_INV_LEVEL(2)
  RT_INV_LEVEL(3)
#undef RT_INV_LEVEL
  for (int k = NLEV - 2; k >= 0; k--) {
    const int n = SB >> k, nby = SB/n, nbx = T::W/n;
    for (int e = lane; e < SB*nbx; e += 64) {          // taps across vertical centre lines
      const int bxt = e/SB, row = e%SB, byi = row/n;
      if ((bxt*n)/SB < nsb && cell(byi, bxt, n) < 3 - k && (sby*nby + byi + 1)*n <= pic_h) {
        int32_t *p = Z + row*T::LDZ + bxt*n + n/2 - 2;
        lap4_post(p[0], p[1], p[2], p[3]);
      }
    }
    __syncthreads();
    for (int byi = 0; byi < nby; byi++) {              // taps across horizontal centre lines
      const int col = lane, bxt = col/n;
      if ((bxt*n)/SB < nsb && cell(byi, bxt, n) < 3 - k && (tx*nbx + bxt + 1)*n <= pic_w) {
        int32_t *p = Z + (byi*n + n/2 - 2)*T::LDZ + col;
        lap4_post(p[0], p[T::LDZ], p[2*T::LDZ], p[3*T::LDZ]);
      }
    }
    __syncthreads();
  }
}

template <int SB, int NLEV>
__global__ __launch_bounds__(64) void k_inverse_rt(InvArgs a) {
  using T = RowTile<SB>;
  __shared__ int32_t Z[SB*T::LDZ];
  __shared__ uint8_t bsz[16*T::NSB];
  const int lane = threadIdx.x;
  const int tx = blockIdx.x, sby = blockIdx.y, f = blockIdx.z;
  const int x0 = tx*T::W, y0 = sby*SB;
  const int sbx0 = tx*T::NSB;
  const int nsb = min(T::NSB, a.nhsb - sbx0);
  for (int e = lane; e < 16*T::NSB; e += 64) {
    const int s = e >> 4, c = e & 15;
    bsz[e] = s < nsb ? a.bsize[(size_t)f*a.bsize_fstride +
                               (size_t)(sby*4 + (c >> 2))*a.bstride + (sbx0 + s)*4 + (c & 3)]
                     : 3;
  }
  rt_load_tile<SB>(Z, a.d + (size_t)f*a.fstride + (size_t)y0*a.w + x0, a.w, x0);
  __syncthreads();
  // range check for the 24-bit multiplier
  int mx = 0;
  for (int e = lane; e < SB*T::W; e += 64) {
    const int v = Z[(e >> 6)*T::LDZ + (e & 63)];
    mx = max(mx, v < 0 ? -(v + 1) : v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
  if (mx <= (1 << 18)) {
    rt_inverse_body<SB, NLEV, true>(Z, bsz, nsb, a.dec, tx, sby, a.pic_w, a.pic_h);
  }
  else {
    rt_inverse_body<SB, NLEV, false>(Z, bsz, nsb, a.dec, tx, sby, a.pic_w, a.pic_h);
  }
  rt_store_tile<SB>(a.c + (size_t)f*a.fstride + (size_t)y0*a.w + x0, a.w, x0, Z);
}
