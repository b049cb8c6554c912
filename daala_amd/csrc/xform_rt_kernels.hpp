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

// A row-tile workgroup is exactly ONE wave, so no s_barrier is needed: DS
// operations of a wave execute in order, and the only thing to wait for before
// another lane's LDS data is read is the wave's own outstanding LDS traffic.
// Unlike __syncthreads() this does NOT wait for global stores (vmcnt), so the
// level stores stay in flight while the next level is computed.
__device__ __forceinline__ void rt_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void rt_lap4_pre16(int16_t *p, int stride) {
  int32_t x0 = p[0], x1 = p[stride], x2 = p[2*stride], x3 = p[3*stride];
  lap4_pre24(x0, x1, x2, x3);      // int16 data: 24-bit products are exact
  p[0] = (int16_t)x0; p[stride] = (int16_t)x1; p[2*stride] = (int16_t)x2; p[3*stride] = (int16_t)x3;
}

// XCD-aware tile order (cdna guide T1).  Workgroups are dealt round-robin over the
// 8 XCDs, each with its own L2; neighbouring tiles share input sectors (2-sample
// halo, 64-byte sectors), so each XCD is given a CONTIGUOUS run of tiles.  Pure
// speed: any mapping is correct.  Bijective for every grid size.
__device__ __forceinline__ void rt_tile_coords(int &tx, int &sby, int &f) {
  const unsigned gx = gridDim.x, gy = gridDim.y;
  const unsigned total = gx*gy*gridDim.z;
  const unsigned lin = blockIdx.x + gx*(blockIdx.y + gy*blockIdx.z);
  const unsigned q = total/8, r = total%8, xcd = lin%8, idx = lin/8;
  const unsigned m = (xcd < r ? xcd*(q + 1) : r*(q + 1) + (xcd - r)*q) + idx;
  tx = m%gx;
  sby = (m/gx)%gy;
  f = m/(gx*gy);
}

template <int SB> struct RowTile {
  static constexpr int W = 64;               // tile width (samples)
  static constexpr int NSB = W/SB;           // superblocks per tile
  static constexpr int LDZ = W + 1;          // coefficient tile stride
  static constexpr int HAW = W + 4;          // lapped tile incl. 2-sample halo
  static constexpr int HAH = SB + 4;
  static constexpr int LDA = W + 6;          // int16 tile: even stride keeps rows dword aligned
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

// Load stage of the inverse tile kernels (one wave): the tile's block sizes (16 bytes per
// superblock) and its SB x 64 coefficients, every global load issued before the first wait - the
// rolled form waited for the block-size byte, then for the tile in two halves, three memory round
// trips in a wave's life.  Returns the wave's max |coefficient| for the 24-bit-multiplier check
// (from the registers: the tile is not read back from LDS for it); -v - 1 for negative v, as the
// range analysis counts it.
template <int SB>
__device__ __forceinline__ int rt_load_tile_bsz(int32_t *Z, uint8_t *bsz, const int32_t *__restrict__ src,
                                                int w, int x0, const uint8_t *__restrict__ bsize,
                                                int bstride, int sby, int sbx0, int nsb) {
  using T = RowTile<SB>;
  static_assert(16*T::NSB <= 64, "one block-size byte per lane");
  const int lane = threadIdx.x;
  const int r0 = lane & 3, c4 = (lane >> 2)*4;
  int b = 3;
  {
    const int s = lane >> 4, c = lane & 15;
    if (lane < 16*T::NSB && s < nsb) b = bsize[(size_t)(sby*4 + (c >> 2))*bstride + (sbx0 + s)*4 + (c & 3)];
  }
  int4 v[SB/4];
#pragma unroll
  for (int it = 0; it < SB/4; it++) v[it] = make_int4(0, 0, 0, 0);
  if (x0 + c4 < w) {
    // one address per lane and a uniform row step (no 64-bit multiply-add per load)
    const int32_t *q = src + (size_t)(unsigned)__umul24(r0, w) + c4;
    const size_t step = (size_t)4*(unsigned)w;
#pragma unroll
    for (int it = 0; it < SB/4; it++) {
      v[it] = *reinterpret_cast<const int4 *>(q);
      q += step;
    }
  }
  if (lane < 16*T::NSB) bsz[lane] = (uint8_t)b;
  int hi = 0, lo = 0;
#pragma unroll
  for (int it = 0; it < SB/4; it++) {
    int32_t *p = Z + (it*4 + r0)*T::LDZ + c4;
    p[0] = v[it].x; p[1] = v[it].y; p[2] = v[it].z; p[3] = v[it].w;
    hi = max(max(hi, v[it].x), max(v[it].y, max(v[it].z, v[it].w)));
    lo = min(min(lo, v[it].x), min(v[it].y, min(v[it].z, v[it].w)));
  }
  int mx = max(hi, -(lo + 1));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
  return mx;
}

// Raw workgroup barrier for the two waves of a row-tile workgroup: waits for the
// wave's LDS traffic only (no vmcnt: global stores stay in flight).
__device__ __forceinline__ void rt_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// RT_NT_STORES: the level planes are written once and read much later by other kernels (1 GB per
// 30-frame luma launch, far beyond the L2): non-temporal stores.  Measured, alternating builds on
// one box: luma pyramid 0.2956 / 0.2944 ms with plain stores, 0.2864 / 0.2855 ms with these (-3 %).
#ifndef RT_NT_STORES
#define RT_NT_STORES 1
#endif
// Store a rectangle of the coefficient tile: rows [row0, row0+NR), columns
// [col0, col0+NC) (NC in {32, 64}), one int4 per lane.
template <int SB, int NR, int NC>
__device__ __forceinline__ void rt_store_rect(int32_t *__restrict__ dst, int w, int x0,
                                              const int32_t *Z, int row0, int col0, int lane) {
  using T = RowTile<SB>;
  constexpr int LPR = NC/4;                     // lanes per row
  constexpr int RPI = 64/LPR;                   // rows per wave instruction
  const int rr = lane % RPI, c4 = col0 + (lane/RPI)*4;
  if (x0 + c4 < w) {
    // one address per lane, then a uniform row step: the 64-bit multiply-add per store
    // the plain form compiles to is quarter rate (v_mad_u64_u32)
    int32_t *q = dst + (size_t)(unsigned)__umul24(row0 + rr, w) + c4;
    const size_t step = (size_t)RPI*(unsigned)w;
#pragma unroll
    for (int it = 0; it < (NR + RPI - 1)/RPI; it++) {
      const int r = row0 + it*RPI + rr;
      if (NR % RPI == 0 || it*RPI + rr < NR) {
        const int32_t *p = Z + r*T::LDZ + c4;
#if RT_NT_STORES
        typedef int rt_v4i __attribute__((ext_vector_type(4)));
        rt_v4i nv = {p[0], p[1], p[2], p[3]};
        __builtin_nontemporal_store(nv, reinterpret_cast<rt_v4i *>(q));
#else
        *reinterpret_cast<int4 *>(q) = make_int4(p[0], p[1], p[2], p[3]);
#endif
      }
      q += step;
    }
  }
}

// Forward path of one row tile, TWO waves per tile.
//   level 0 (block size SB): the tile has exactly 64 column transforms and 64 row
//     transforms per pass, so wave 0 runs the whole level with all 64 lanes busy while
//     wave 1 waits at the barrier (splitting the level by columns issued the same
//     instruction stream twice with half the lanes masked).
//   levels >= 1: wave w owns the rows [w*SB/2, (w+1)*SB/2) of the tile at every
//     level (blocks nest): nothing crosses waves.
// So the only workgroup barriers are: after the load, between the two frame-lapping
// phases, after them, and between level 0 and level 1.
// PMC profile (rocprofv3 SQ_INSTS_VALU / SQ_WAVES): ~2800 VALU instructions per wave =
// 0.28 ms of pure VALU issue for a 30-frame luma launch - this kernel is bound by VALU
// issue about as much as by HBM, hence the instruction-count work: one-instruction
// v_mad_i32_i24 lifting steps, 24-bit lapping, 24-bit index arithmetic, pointer stepping
// instead of a 64-bit multiply-add per store.
// PYRAMID: all blocks of all levels (one plane per level).  KNOWN: only the quadtree
// leaves + keyframe DC merge (od_compute_dcts, src/encode.c:1286-1343).
template <int SB, int NLEV, bool KNOWN>
__global__ __launch_bounds__(128) void k_forward_rt(FwdArgs a) {
  using T = RowTile<SB>;
  // lapped spatial tile as int16: |value| <= 6452 for pixel-driven data
  // (tools/range_analysis.py); halves its LDS footprint
  __shared__ __attribute__((aligned(8))) int16_t A_[T::HAH*T::LDA + 2];
  int16_t *const A = A_ + 2;             // the load stage writes the pair (-2, -1) of row 0
  __shared__ int32_t Z[SB*T::LDZ];
  __shared__ uint8_t bsz[16*T::NSB];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index: scalar
  int tx, sby, f;
  rt_tile_coords(tx, sby, f);
  sby += a.sby0;                                       // strip launches cover rows sby0 .. sby0 + gridDim.y
  const int x0 = tx*T::W, y0 = sby*SB;
  const int sbx0 = tx*T::NSB;                          // first superblock of the tile
  const int nsb = min(T::NSB, a.nhsb - sbx0);          // superblocks really present
  const int dec = a.dec;
  static_assert(16*T::NSB <= 128, "one block-size byte per thread");
  int bsz_mine = 3;                                    // stored behind the pixel loads: one wait for both
  if (KNOWN) {
    const int s = tid >> 4, c = tid & 15;
    if (tid < 16*T::NSB && s < nsb) {
      bsz_mine = a.bsize[(size_t)f*a.bsize_fstride + (size_t)(sby*4 + (c >> 2))*a.bstride + (sbx0 + s)*4 + (c & 3)];
    }
  }
  {
    // A1: 8-bit pixels -> (p - 128) << 4, dword loads starting 4 bytes left of the tile.  All of
    // a thread's loads are issued before the first conversion (the rolled loop waited for every
    // load by itself), and a dword's four samples are converted as two packed 16-bit pairs and
    // stored as two dwords: sample 4*dx + b - 2 of the halo'd row; the pairs (-2, -1) and (68, 69)
    // fall into the row padding (LDA = HAW + 2; the tile starts two entries into its array).
    const uint8_t *pix = a.pix + (size_t)f*a.pix_fstride;
    constexpr int DW = T::HAW/4 + 1;                   // 18 dwords cover x0-4 .. x0+67
    static_assert(DW == 18, "the division by DW below is spelled for 18");
    static_assert(T::LDA == T::HAW + 2 && T::LDA % 2 == 0, "row padding takes the two outer pairs");
    constexpr int NE = T::HAH*DW, IT = (NE + 127)/128;
    typedef unsigned short rt_u16x2 __attribute__((ext_vector_type(2)));
    uint32_t v[IT];
    int at[IT];
#pragma unroll
    for (int it = 0; it < IT; it++) {
      const int e = min(tid + it*128, NE - 1);
      const int ty = __umul24(e, 3641) >> 16, dx = e - ty*DW;       // e/18, exact for e < 5000
      const int gy = min(max(y0 - 2 + ty, 0), a.h - 1);
      const int gx = min(max(x0 - 4 + dx*4, 0), a.w - 4);
      v[it] = *reinterpret_cast<const uint32_t *>(pix + ((unsigned)__umul24(gy, a.pstride) + gx));
      at[it] = ty*T::LDA + dx*4 - 2;
    }
    if (KNOWN && tid < 16*T::NSB) bsz[tid] = (uint8_t)bsz_mine;
    // threads beyond the last entry repeat it (same address, same value): no branch around the
    // last round, so its load is issued with the others
#pragma unroll
    for (int it = 0; it < IT; it++) {
      {
        union { uint32_t u; rt_u16x2 h; } lo, hi;
        lo.u = __builtin_amdgcn_perm(0u, v[it], 0x0c010c00u);      // bytes 0, 1 -> the two halves
        hi.u = __builtin_amdgcn_perm(0u, v[it], 0x0c030c02u);      // bytes 2, 3
        const rt_u16x2 k16 = {16, 16}, kb = {0xf800, 0xf800};       // (p << 4) - (128 << 4), mod 2^16
        lo.h = lo.h*k16 + kb;
        hi.h = hi.h*k16 + kb;
        uint32_t *q = reinterpret_cast<uint32_t *>(A + at[it]);
        q[0] = lo.u;
        q[1] = hi.u;
      }
    }
  }
  rt_barrier();
  // A4: frame lapping.  Horizontal SB boundaries first (vertical taps over every
  // column of the halo'd tile), then vertical boundaries (src/filter.c:1566-1584).
  for (int c = tid; c < T::HAW; c += 128) {
    if (sby > 0) rt_lap4_pre16(A + c, T::LDA);
    if (sby < a.nvsb - 1) rt_lap4_pre16(A + SB*T::LDA + c, T::LDA);
  }
  rt_barrier();
  for (int e = tid; e < (T::NSB + 1)*SB; e += 128) {
    const int s = e/SB, row = e%SB;                    // boundary left of tile SB s
    const int gb = sbx0 + s;                           // global boundary index
    if (gb >= 1 && gb <= a.nhsb - 1 && s <= nsb) rt_lap4_pre16(A + (2 + row)*T::LDA + s*SB, 1);
  }
  rt_barrier();
  auto cell = [&](int byi, int bxt, int n) -> int {    // max(obs, dec) of a tile block
    const int s = (bxt*n)/SB, bxs = bxt - s*(SB/n);
    const int nl = n << dec;
    const int o = bsz[s*16 + ((byi*nl) >> 3)*4 + ((bxs*nl) >> 3)];
    return o > dec ? o : dec;
  };
  int32_t *out = a.out + (size_t)f*a.out_fstride + (size_t)y0*a.w + x0;
  // One transform level.  K == 0: this wave's 32 columns (lanes 0..31), the whole
  // tile height; K >= 1: all 64 columns, this wave's block rows.
#define RT_FWD_LEVEL(K)                                                                \
  if constexpr (K < NLEV) {                                                            \
    constexpr int N = SB >> K;                                                         \
    constexpr int NBY = SB/N;                                                          \
    constexpr int BY_PER_WAVE = K == 0 ? 1 : NBY/2;                                    \
    const int col = lane;                                                              \
    const int bxt = col/N, i = col%N;                                                  \
    const bool present = (bxt*N)/SB < nsb;                                             \
    /* level 0 has only 64 column / 64 row transforms per pass: ONE wave runs it with  \
       all 64 lanes busy, the other goes straight to the barrier (a wave split by      \
       columns would issue the same instructions with half the lanes masked: the PMC   \
       profile showed the kernel VALU-issue bound, not HBM bound) */                   \
    if (K != 0 || wv == 0) {                                                           \
    for (int bb = 0; bb < BY_PER_WAVE; bb++) {                                         \
      const int byi = K == 0 ? 0 : wv*BY_PER_WAVE + bb;                                \
      const bool go = present && (!KNOWN || cell(byi, bxt, N) == 3 - K);               \
      int32_t v[N];                                                                    \
      if (go) {                                                                        \
        _Pragma("unroll") for (int k = 0; k < N; k++)                                  \
          v[k] = A[(2 + byi*N + k)*T::LDA + 2 + col];                                  \
        LiftDct<N, true>::fwd(v);                                                      \
        _Pragma("unroll") for (int k = 0; k < N; k++)                                  \
          Z[(byi*N + i)*T::LDZ + bxt*N + k] = v[k];                                    \
      }                                                                                \
      rt_sync();                                                                       \
      if (go) {                                                                        \
        _Pragma("unroll") for (int k = 0; k < N; k++)                                  \
          v[k] = Z[(byi*N + k)*T::LDZ + col];                                          \
        LiftDct<N, true>::fwd(v);                                                      \
      }                                                                                \
      rt_sync();                                                                       \
      if (go) {                                                                        \
        _Pragma("unroll") for (int k = 0; k < N; k++)                                  \
          Z[(byi*N + i)*T::LDZ + bxt*N + k] = v[k];                                    \
      }                                                                                \
      rt_sync();                                                                       \
      if (!KNOWN) {                                                                    \
        rt_store_rect<SB, N, 64>(out + (size_t)K*a.out_lstride, a.w, x0, Z, byi*N, 0, lane); \
        rt_sync();                                                                     \
      }                                                                                \
      if constexpr (K + 1 < NLEV) {                                                    \
        /* A5: split lapping of this block row: taps across the horizontal centre */   \
        /* line of each block (per column), then across the vertical centre line   */  \
        const bool sp = present && (!KNOWN || cell(byi, bxt, N) < 3 - K);              \
        if (sp && (tx*(T::W/N) + bxt + 1)*N <= a.pic_w) {                              \
          rt_lap4_pre16(A + (2 + byi*N + N/2 - 2)*T::LDA + 2 + col, T::LDA);           \
        }                                                                              \
        rt_sync();                                                                     \
        constexpr int NBX = T::W/N;                      /* blocks per tile row */     \
        for (int e = lane; e < N*NBX; e += 64) {                                       \
          const int bx2 = e/N, row = byi*N + e%N;                                      \
          const bool sp2 = (bx2*N)/SB < nsb && (!KNOWN || cell(byi, bx2, N) < 3 - K);  \
          if (sp2 && (sby*NBY + byi + 1)*N <= a.pic_h) {                               \
            rt_lap4_pre16(A + (2 + row)*T::LDA + 2 + bx2*N + N/2 - 2, 1);              \
          }                                                                            \
        }                                                                              \
        rt_sync();                                                                     \
      }                                                                                \
    }                                                                                  \
    }                                                                                  \
    if (K == 0) rt_barrier();    /* level >= 1 regions mix the whole level-0 tile */   \
  }
  RT_FWD_LEVEL(0)
  RT_FWD_LEVEL(1)
  RT_FWD_LEVEL(2)
  RT_FWD_LEVEL(3)
#undef RT_FWD_LEVEL
  if (KNOWN) {
    rt_barrier();
    if (a.keyframe) {
      // Haar merge of the four child DCs of every split block, finest first
      for (int k = NLEV - 2; k >= 0; k--) {
        const int n = SB >> k, nby = SB/n, nbx = T::W/n, hh = n/2;
        if (tid < nby*nbx) {
          const int byi = tid/nbx, bxt = tid%nbx;
          if ((bxt*n)/SB < nsb && cell(byi, bxt, n) < 3 - k) {
            int32_t *p = Z + (byi*n)*T::LDZ + bxt*n;
            int32_t q0 = p[0], q1 = p[hh], q2 = p[hh*T::LDZ], q3 = p[hh*T::LDZ + hh];
            haar2x2(q0, q2, q1, q3);
            p[0] = q0; p[hh] = q1; p[hh*T::LDZ] = q2; p[hh*T::LDZ + hh] = q3;
          }
        }
        rt_barrier();
      }
    }
    rt_store_rect<SB, SB/2, 64>(out, a.w, x0, Z, wv*(SB/2), 0, lane);
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
  // RT_INV_COMPACT: with a quadtree of mixed block sizes a level's instruction stream would run
  // with most lanes masked (an iteration of 64 consecutive items executes when ANY of its blocks
  // has this size).  The blocks of the level that really have this size are listed first (ballot +
  // prefix count), and lane groups of N take them in list order: ceil(active*N/64) iterations.
  // Measured bound (uniform maps): tile kernel 0.117-0.125 ms against 0.164 on random maps.
  __shared__ uint8_t blist[(SB/4)*(T::W/4)];
#define RT_INV_LEVEL(K)                                                                \
  if constexpr (K < NLEV) {                                                            \
    constexpr int N = SB >> K;                                                         \
    constexpr int NBY = SB/N, NB = NBY*(T::W/N);                                       \
    int nact = 0;                                                                      \
    for (int c0 = 0; c0 < NB; c0 += 64) {                                              \
      const int c = c0 + lane;                                                         \
      const int cb = c/NBY, cy = c%NBY;             /* block column slowest, as before */ \
      const bool act = c < NB && (cb*N)/SB < nsb && cell(cy, cb, N) == 3 - K;          \
      const unsigned long long m = __ballot(act);                                      \
      if (act) blist[nact + __popcll(m & ((1ull << lane) - 1))] = (uint8_t)c;          \
      nact += __popcll(m);                                                             \
    }                                                                                  \
    rt_sync();                                                                         \
    /* item = (listed block, row): N consecutive lanes share a block, so both in-place */ \
    /* passes stay inside the items' own blocks                                        */ \
    for (int e0 = 0; e0 < nact*N; e0 += 64) {                                          \
      const int e = e0 + lane;                                                         \
      const bool go = e < nact*N;                                                      \
      const int blk = go ? blist[e/N] : 0;                                             \
      const int bxt = blk/NBY, byi = blk%NBY, i = e%N, row = byi*N + i;                \
      int32_t v[N];                                                                    \
      if (go) {                                                                        \
        _Pragma("unroll") for (int k = 0; k < N; k++) v[k] = Z[row*T::LDZ + bxt*N + k];\
        LiftDct<N, M24>::inv(v);                                                       \
      }                                                                                \
      rt_sync();                                                                 \
      if (go) {                                                                        \
        _Pragma("unroll") for (int k = 0; k < N; k++)                                  \
          Z[(byi*N + k)*T::LDZ + bxt*N + i] = v[k];                                    \
      }                                                                                \
      rt_sync();                                                                 \
      if (go) {                                                                        \
        _Pragma("unroll") for (int k = 0; k < N; k++) v[k] = Z[row*T::LDZ + bxt*N + k];\
        LiftDct<N, M24>::inv(v);                                                       \
      }                                                                                \
      rt_sync();                                                                 \
      if (go) {                                                                        \
        _Pragma("unroll") for (int k = 0; k < N; k++)                                  \
          Z[(byi*N + k)*T::LDZ + bxt*N + i] = v[k];                                    \
      }                                                                                \
      rt_sync();                                                                 \
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
    rt_sync();
    for (int byi = 0; byi < nby; byi++) {              // taps across horizontal centre lines
      const int col = lane, bxt = col/n;
      if ((bxt*n)/SB < nsb && cell(byi, bxt, n) < 3 - k && (tx*nbx + bxt + 1)*n <= pic_w) {
        int32_t *p = Z + (byi*n + n/2 - 2)*T::LDZ + col;
        lap4_post(p[0], p[T::LDZ], p[2*T::LDZ], p[3*T::LDZ]);
      }
    }
    rt_sync();
  }
}

template <int SB, int NLEV>
__global__ __launch_bounds__(64) void k_inverse_rt(InvArgs a) {
  using T = RowTile<SB>;
  __shared__ int32_t Z[SB*T::LDZ];
  __shared__ uint8_t bsz[16*T::NSB];
  int tx, sby, f;
  rt_tile_coords(tx, sby, f);
  const int x0 = tx*T::W, y0 = sby*SB;
  const int sbx0 = tx*T::NSB;
  const int nsb = min(T::NSB, a.nhsb - sbx0);
  // range check for the 24-bit multiplier: the tile's largest magnitude
  const int mx = rt_load_tile_bsz<SB>(Z, bsz, a.d + (size_t)f*a.fstride + (size_t)y0*a.w + x0, a.w, x0,
                                      a.bsize + (size_t)f*a.bsize_fstride, a.bstride, sby, sbx0, nsb);
  rt_sync();
  if (mx <= (1 << 18)) {
    rt_inverse_body<SB, NLEV, true>(Z, bsz, nsb, a.dec, tx, sby, a.pic_w, a.pic_h);
  }
  else {
    rt_inverse_body<SB, NLEV, false>(Z, bsz, nsb, a.dec, tx, sby, a.pic_w, a.pic_h);
  }
  rt_store_tile<SB>(a.c + (size_t)f*a.fstride + (size_t)y0*a.w + x0, a.w, x0, Z);
}


// ---------------------------------------------------------------------------------
// Inverse + frame post-filter + clamp WITHOUT the int32 work plane (A7 + A5 + A4 + A2).
// k_inverse_rt above writes the whole lapped-domain tile as int32 and k_postfilter_clamp
// reads it back: 13 B/sample of traffic for 5 B/sample of work.  Here the tile finishes
// everything that does not involve another tile:
//   * the frame post-filter's first pass (od_apply_postfilter_frame_sbs, src/filter.c:1627-
//     1633: 4-tap across every vertical superblock boundary, all rows) on the boundaries
//     INSIDE the 64-wide tile;
//   * rows 2 .. SB-3 x columns 2 .. tw-3 are then final (the second pass touches only rows
//     within 2 of a horizontal boundary, the first pass of a neighbouring tile only columns
//     within 2 of the tile edge): clamped to 8 bit and written once;
//   * the 2-sample edge strips (tile rows 0, 1, SB-2, SB-1; tile columns 0, 1, tw-2, tw-1)
//     go to two small int16 strip buffers; a value that does not fit int16 (possible only
//     for out-of-range streams) is stored as the marker -32768 and its int32 value goes to
//     the work plane c at its own position, where the strip kernel picks it up: exact for
//     every input, 2 bytes per strip sample in the normal case.
// k_inverse_strips then finishes the strips of tile (tx, sby): the first pass across its
// left tile boundary for its interior rows, and the second pass (vertical 4-tap across the
// horizontal superblock boundary above it) for all its columns - after redoing the first
// pass locally for the four rows x two tile boundaries involved, so that no workgroup
// depends on another's output.  Frame edges have no boundary: those strips are only
// clamped.  Order of the two passes and every tap are the reference's.
#define STRIP_ESC (-32768)
__device__ __forceinline__ int16_t strip_put(int32_t v, int32_t *esc) {
  if (v > STRIP_ESC && v <= 32767) return (int16_t)v;
  *esc = v;
  return (int16_t)STRIP_ESC;
}
__device__ __forceinline__ int32_t strip_get(int16_t s, const int32_t *esc) {
  return s == STRIP_ESC ? *esc : (int32_t)s;
}

__device__ __forceinline__ uint32_t clamp8(int32_t v) {
  v = ((v + 8) >> 4) + 128;
  return (uint32_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}

// RT_SEG > 1 (experiment, off): one workgroup walks RT_SEG consecutive tiles of a tile row and
// finishes the boundaries between them itself - lane r keeps the last four columns of row r
// of the tile just done, the first pass across the boundary (4-tap on the previous tile's
// last two and this tile's first two columns, all rows - src/filter.c:1627-1633) runs once
// the next tile is in LDS, and the delayed four bytes go out as one aligned word; column
// strips and the strip kernel's first pass are then needed at SEGMENT edges only.  Measured on
// MI355X (30 luma frames per launch, tile kernel + strip kernel, ms): RT_SEG 1: 0.155 + 0.037,
// 2: 0.266 + 0.041, 4: 0.215 + 0.036, 8: 0.251 + 0.033; WRITE_SIZE of the pair 141 MB (1) vs
// 138 MB (4), FETCH_SIZE equal: the delayed word is itself a lone partial-sector store, so the
// counted traffic does not move while one wave doing its tiles back to back (141 VGPRs instead
// of 76) costs time.  Default: one tile per workgroup.  Only stores of whole 64-byte row
// segments that straddle the tile boundary (previous tile's right half kept in LDS) would
// remove the partial sectors; not built - the pair is VALU bound, not traffic bound.
#ifndef RT_SEG
#define RT_SEG 1
#endif
#ifndef RT_FUSED_WAVES
#define RT_FUSED_WAVES 1      /* min waves per SIMD asked of the compiler (VGPR cap) */
#endif
template <int SB, int NLEV>
__global__ __launch_bounds__(64, RT_FUSED_WAVES) void k_inverse_rt_fused(InvArgs a) {
  using T = RowTile<SB>;
  __shared__ int32_t Z[SB*T::LDZ];
  __shared__ uint8_t bsz[16*T::NSB];
  const int lane = threadIdx.x;
  int seg, sby, f;
  rt_tile_coords(seg, sby, f);
  const int y0 = sby*SB;
  int32_t pend[4] = {0, 0, 0, 0};     // lane r < SB: columns tw-4 .. tw-1 of row r of the previous tile
  for (int t = 0; t < RT_SEG; t++) {
    const int tx = seg*RT_SEG + t;
    if (tx >= a.ntx) break;
    const int x0 = tx*T::W;
    const int sbx0 = tx*T::NSB;
    const int nsb = min(T::NSB, a.nhsb - sbx0);
    const int tw = min(T::W, a.w - x0);
    const bool first = t == 0;                                   // the segment's left edge
    const bool lastt = t == RT_SEG - 1 || tx == a.ntx - 1;       // the segment's right edge
    const int mx = rt_load_tile_bsz<SB>(Z, bsz, a.d + (size_t)f*a.fstride + (size_t)y0*a.w + x0, a.w, x0,
                                        a.bsize + (size_t)f*a.bsize_fstride, a.bstride, sby, sbx0, nsb);
    rt_sync();
    if (mx <= (1 << 18)) rt_inverse_body<SB, NLEV, true>(Z, bsz, nsb, a.dec, tx, sby, a.pic_w, a.pic_h);
    else rt_inverse_body<SB, NLEV, false>(Z, bsz, nsb, a.dec, tx, sby, a.pic_w, a.pic_h);
    // first pass on the tile-internal vertical superblock boundaries
    if (lane < (T::NSB - 1)*SB) {
      const int k = lane/SB + 1, r = lane%SB;
      if (k < nsb) {
        int32_t *p = Z + r*T::LDZ + k*SB - 2;
        lap4_post(p[0], p[1], p[2], p[3]);
      }
    }
    uint8_t *rec = a.rec + (size_t)f*a.fstride + (size_t)y0*a.w + x0;
    int32_t *escp = a.c + (size_t)f*a.fstride + (size_t)y0*a.w + x0;
    // first pass across the boundary to the previous tile of this segment; its delayed word
    if (!first && lane < SB) {
      int32_t *p = Z + lane*T::LDZ;
      lap4_post(pend[2], pend[3], p[0], p[1]);
      if (lane >= 2 && lane < SB - 2) {
        *reinterpret_cast<uint32_t *>(rec + (size_t)lane*a.w - 4) =
            clamp8(pend[0]) | clamp8(pend[1]) << 8 | clamp8(pend[2]) << 16 | clamp8(pend[3]) << 24;
      }
      else {
        // a strip row: the previous tile left the last two entries of its row strip to this pass
        const int k = lane < 2 ? lane : lane - (SB - 4);
        int16_t *q = a.rs + (size_t)f*a.rs_fstride + ((size_t)sby*4 + k)*a.w + x0 - 2;
        int32_t *e = escp + (size_t)lane*a.w - 2;
        *reinterpret_cast<uint32_t *>(q) =
            (uint16_t)strip_put(pend[2], e) | (uint32_t)(uint16_t)strip_put(pend[3], e + 1) << 16;
      }
    }
    rt_sync();
    // interior: final, 8 bit
    {
      const int r0 = lane & 3, c4 = (lane >> 2)*4;
#pragma unroll
      for (int it = 0; it < SB/4; it++) {
        const int r = it*4 + r0;
        if (r >= 2 && r < SB - 2 && c4 < tw) {
          const int32_t *p = Z + r*T::LDZ + c4;
          uint8_t *q = rec + (size_t)r*a.w + c4;
          const uint32_t word = clamp8(p[0]) | clamp8(p[1]) << 8 | clamp8(p[2]) << 16 | clamp8(p[3]) << 24;
          if (c4 == tw - 4 && !lastt) { /* delayed: written with the next tile's boundary pass */ }
          else if (c4 == 0 && first && c4 == tw - 4) { /* a 4-column tile: strips only */ }
          else if (c4 == 0 && first) *reinterpret_cast<uint16_t *>(q + 2) = (uint16_t)(word >> 16);
          else if (c4 == tw - 4 && lastt) *reinterpret_cast<uint16_t *>(q) = (uint16_t)word;
          else *reinterpret_cast<uint32_t *>(q) = word;
        }
      }
    }
    // row strips: tile rows 0, 1, SB-2, SB-1
    {
      const int k = lane >> 4, c4 = (lane & 15)*4;
      const int r = k < 2 ? k : SB - 4 + k;
      if (c4 < tw) {
        const int32_t *p = Z + r*T::LDZ + c4;
        int32_t *e = escp + (size_t)r*a.w + c4;
        int16_t *q = a.rs + (size_t)f*a.rs_fstride + ((size_t)sby*4 + k)*a.w + x0 + c4;
        const uint32_t lo = (uint16_t)strip_put(p[0], e) | (uint32_t)(uint16_t)strip_put(p[1], e + 1) << 16;
        if (c4 == tw - 4 && !lastt) {
          // the last two columns wait for the pass across the boundary to the next tile
          *reinterpret_cast<uint32_t *>(q) = lo;
        }
        else {
          const uint32_t hi = (uint16_t)strip_put(p[2], e + 2) | (uint32_t)(uint16_t)strip_put(p[3], e + 3) << 16;
          *reinterpret_cast<uint2 *>(q) = make_uint2(lo, hi);
        }
      }
    }
    // column strips at the segment's edges; the last four columns of every row for the next tile
    if (lane < SB) {
      const int32_t *p = Z + lane*T::LDZ;
      int32_t *e = escp + (size_t)lane*a.w;
      int16_t *q = a.cs + (size_t)f*a.cs_fstride + ((size_t)tx*a.h + y0 + lane)*4;
      if (first) {
        *reinterpret_cast<uint32_t *>(q) =
            (uint16_t)strip_put(p[0], e) | (uint32_t)(uint16_t)strip_put(p[1], e + 1) << 16;
      }
      if (lastt) {
        *reinterpret_cast<uint32_t *>(q + 2) =
            (uint16_t)strip_put(p[tw - 2], e + tw - 2) | (uint32_t)(uint16_t)strip_put(p[tw - 1], e + tw - 1) << 16;
      }
      else {
        pend[0] = p[tw - 4]; pend[1] = p[tw - 3]; pend[2] = p[tw - 2]; pend[3] = p[tw - 1];
      }
    }
    rt_sync();                       // Z is reloaded by the next tile
  }
}

template <int SB>
__global__ __launch_bounds__(64) void k_inverse_strips(InvArgs a) {
  using T = RowTile<SB>;
  __shared__ int32_t E[6][4];          // edge columns {0, 1, tw-2, tw-1} of rows y0-2 .. y0+1, then the frame's last two rows
  const int lane = threadIdx.x;
  int tx, sby, f;
  rt_tile_coords(tx, sby, f);
  const int x0 = tx*T::W, y0 = sby*SB;
  const int tw = min(T::W, a.w - x0);
  const int16_t *cs = a.cs + (size_t)f*a.cs_fstride;
  const int16_t *rs = a.rs + (size_t)f*a.rs_fstride;
  const int32_t *esc = a.c + (size_t)f*a.fstride;
  uint8_t *rec = a.rec + (size_t)f*a.fstride;
  // tile boundaries inside a segment were finished by k_inverse_rt_fused (values in the row
  // strips and bytes already final there): only segment edges are handled here
  const bool lseg = tx%RT_SEG == 0, rseg = tx%RT_SEG == RT_SEG - 1 || tx == a.ntx - 1;
  const bool left = tx > 0 && lseg, right = tx < a.ntx - 1 && rseg, top = sby > 0, last = sby == a.nvsb - 1;
  // Every strip value the wave needs is loaded FIRST (raw int16: up to four 8-byte column-strip rows
  // and six row-strip entries per lane, all independent), then the rare escapes are resolved: the
  // straight-line form loaded a row, waited, looked for the marker, loaded the next - some twenty
  // memory round trips in a wave that computes for a microsecond.
  auto csraw = [&](int t, int y) -> uint2 {
    return *reinterpret_cast<const uint2 *>(cs + ((size_t)t*a.h + y)*4);
  };
  // the four strip columns of row y of tile t: its columns 0, 1, tw_t-2, tw_t-1
  auto csval = [&](uint2 u, int t, int y) -> int4 {
    const int xt = t*T::W, twt = min(T::W, a.w - xt);
    const int32_t *e = esc + (size_t)y*a.w + xt;
    return make_int4(strip_get((int16_t)(u.x & 0xffff), e), strip_get((int16_t)(u.x >> 16), e + 1),
                     strip_get((int16_t)(u.y & 0xffff), e + twt - 2), strip_get((int16_t)(u.y >> 16), e + twt - 1));
  };
  const bool p1 = lane < SB, p1cur = p1 && (lseg || tx == a.ntx - 1), p1prv = p1 && lseg && left;
  uint2 u1c = make_uint2(0, 0), u1p = make_uint2(0, 0);
  if (p1cur) u1c = csraw(tx, y0 + lane);
  if (p1prv) u1p = csraw(tx - 1, y0 + lane);
  // the rows of the horizontal boundary above this tile (y0-2 .. y0+1) and, on the last tile row,
  // the frame's last two rows: lane = 2*slot + side
  const int k2 = lane >> 1, side = lane & 1;           // row slot, 0 = left boundary / 1 = right boundary
  const int y2 = k2 < 4 ? y0 - 2 + k2 : y0 + SB - 6 + k2;   // slots 4, 5 -> y0 + SB - 2, y0 + SB - 1
  const bool p2 = lane < 12 && (k2 < 4 ? (y2 >= 0) : last) && (side == 0 ? lseg : rseg);
  const bool p2oth = p2 && (side == 0 ? left : right);
  uint2 u2c = make_uint2(0, 0), u2o = make_uint2(0, 0);
  if (p2) u2c = csraw(tx, y2);
  if (p2oth) u2o = csraw(side == 0 ? tx - 1 : tx + 1, y2);
  const bool p3 = lane < tw;
  const size_t col = (size_t)x0 + lane;
  int16_t r3[6] = {0, 0, 0, 0, 0, 0};
  if (p3) {
    if (top) {
      r3[0] = rs[(size_t)((sby - 1)*4 + 2)*a.w + col];
      r3[1] = rs[(size_t)((sby - 1)*4 + 3)*a.w + col];
    }
    r3[2] = rs[(size_t)(sby*4 + 0)*a.w + col];
    r3[3] = rs[(size_t)(sby*4 + 1)*a.w + col];
    if (last) {
      r3[4] = rs[(size_t)(sby*4 + 2)*a.w + col];
      r3[5] = rs[(size_t)(sby*4 + 3)*a.w + col];
    }
  }
  // first pass across the left segment boundary, this tile's interior rows
  if (p1) {
    const int r = lane;
    const bool inner = r >= 2 && r < SB - 2;
    uint8_t *q = rec + (size_t)(y0 + r)*a.w + x0;
    int4 cur = make_int4(0, 0, 0, 0);
    if (p1cur) cur = csval(u1c, tx, y0 + r);
    if (lseg) {
      int32_t v0 = 0, v1 = 0, v2 = cur.x, v3 = cur.y;
      if (left) {
        const int4 prv = csval(u1p, tx - 1, y0 + r);
        v0 = prv.z; v1 = prv.w;
        lap4_post(v0, v1, v2, v3);
      }
      if (inner) {
        if (left) *reinterpret_cast<uint16_t *>(q - 2) = (uint16_t)(clamp8(v0) | clamp8(v1) << 8);
        *reinterpret_cast<uint16_t *>(q) = (uint16_t)(clamp8(v2) | clamp8(v3) << 8);
      }
    }
    if (tx == a.ntx - 1 && inner) {
      // the frame's right edge has no boundary: its last two columns are only clamped
      *reinterpret_cast<uint16_t *>(q + tw - 2) = (uint16_t)(clamp8(cur.z) | clamp8(cur.w) << 8);
    }
  }
  // the first pass redone for the boundary rows (inside a segment the tile kernel finished them)
  if (p2) {
    const int4 cur = csval(u2c, tx, y2);
    if (side == 0) {
      int32_t v0 = 0, v1 = 0, v2 = cur.x, v3 = cur.y;
      if (left) {
        const int4 prv = csval(u2o, tx - 1, y2);
        v0 = prv.z; v1 = prv.w;
        lap4_post(v0, v1, v2, v3);
      }
      E[k2][0] = v2; E[k2][1] = v3;
    }
    else {
      int32_t v0 = cur.z, v1 = cur.w, v2 = 0, v3 = 0;
      if (right) {
        const int4 nxt = csval(u2o, tx + 1, y2);
        v2 = nxt.x; v3 = nxt.y;
        lap4_post(v0, v1, v2, v3);
      }
      E[k2][2] = v0; E[k2][3] = v1;
    }
  }
  __syncthreads();
  // second pass across the horizontal boundary above this tile, one column per lane
  if (p3) {
    const int c = lane;
    const int ei = (c < 2 && lseg) ? c : (c >= tw - 2 && rseg) ? c - (tw - 4) : -1;
    int32_t v[4] = {0, 0, 0, 0};
    if (top) {
      v[0] = strip_get(r3[0], esc + (size_t)(y0 - 2)*a.w + col);
      v[1] = strip_get(r3[1], esc + (size_t)(y0 - 1)*a.w + col);
    }
    v[2] = strip_get(r3[2], esc + (size_t)y0*a.w + col);
    v[3] = strip_get(r3[3], esc + (size_t)(y0 + 1)*a.w + col);
    if (ei >= 0) {
      if (top) { v[0] = E[0][ei]; v[1] = E[1][ei]; }
      v[2] = E[2][ei]; v[3] = E[3][ei];
    }
    if (top) {
      lap4_post(v[0], v[1], v[2], v[3]);
      rec[(size_t)(y0 - 2)*a.w + col] = (uint8_t)clamp8(v[0]);
      rec[(size_t)(y0 - 1)*a.w + col] = (uint8_t)clamp8(v[1]);
    }
    rec[(size_t)y0*a.w + col] = (uint8_t)clamp8(v[2]);
    rec[(size_t)(y0 + 1)*a.w + col] = (uint8_t)clamp8(v[3]);
    if (last) {
      int32_t b0 = strip_get(r3[4], esc + (size_t)(y0 + SB - 2)*a.w + col);
      int32_t b1 = strip_get(r3[5], esc + (size_t)(y0 + SB - 1)*a.w + col);
      if (ei >= 0) { b0 = E[4][ei]; b1 = E[5][ei]; }
      rec[(size_t)(y0 + SB - 2)*a.w + col] = (uint8_t)clamp8(b0);
      rec[(size_t)(y0 + SB - 1)*a.w + col] = (uint8_t)clamp8(b1);
    }
  }
}
