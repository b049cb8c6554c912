/* hip_dct_host.c - the 2-D lifting DCTs that stay on the host, on the host's vector unit.
 *
 * The block-size RDO inverts every trial block it codes (the reconstruction is the input of
 * the next decision), chroma and 4x4 luma are transformed per block, and od_compute_dist
 * transforms an 8x8 error block per call (src/encode.c:1014): these run inside the serial
 * stage, one block at a time, so they cannot be batched onto the device - but one block is
 * N independent 1-D transforms, which is what a vector unit is for.  The 1-D networks are
 * the ones the device kernels are generated from (tools/lifting_networks.py ->
 * gen_lift_host.h: the same straight-line step lists on GCC vector types); the 2-D order is
 * the reference's (src/dct.c:335-347 and the 4/16/32 twins):
 *   forward: 1-D down every column of x, results stored as ROWS of z; the same again z -> y
 *   inverse: 1-D along every ROW of y, results stored as columns of z; the same again z -> x
 * so a pass handles eight columns (rows) per call with one 8x8 transpose on the store (load)
 * side.  Same integers as the C functions: tests/test_hipenc_cpu.py compares them on random
 * blocks, every end-to-end test compares the packets.
 * Bound as the CPU entries behind the workers' fdct_2d / idct_2d vtable hooks. */
#include <immintrin.h>
#include <stdint.h>
#include <string.h>

#include "gen_lift_host.h"
#include "hip_glue_int.h"

/* 8x8 transpose of 32-bit lanes: r[0..7] rows in, columns out */
static inline void transpose8(lift_v8 *r) {
  __m256i a0, a1, a2, a3, a4, a5, a6, a7;
  __m256i b0, b1, b2, b3, b4, b5, b6, b7;
  a0 = _mm256_unpacklo_epi32((__m256i)r[0], (__m256i)r[1]);
  a1 = _mm256_unpackhi_epi32((__m256i)r[0], (__m256i)r[1]);
  a2 = _mm256_unpacklo_epi32((__m256i)r[2], (__m256i)r[3]);
  a3 = _mm256_unpackhi_epi32((__m256i)r[2], (__m256i)r[3]);
  a4 = _mm256_unpacklo_epi32((__m256i)r[4], (__m256i)r[5]);
  a5 = _mm256_unpackhi_epi32((__m256i)r[4], (__m256i)r[5]);
  a6 = _mm256_unpacklo_epi32((__m256i)r[6], (__m256i)r[7]);
  a7 = _mm256_unpackhi_epi32((__m256i)r[6], (__m256i)r[7]);
  b0 = _mm256_unpacklo_epi64(a0, a2);
  b1 = _mm256_unpackhi_epi64(a0, a2);
  b2 = _mm256_unpacklo_epi64(a1, a3);
  b3 = _mm256_unpackhi_epi64(a1, a3);
  b4 = _mm256_unpacklo_epi64(a4, a6);
  b5 = _mm256_unpackhi_epi64(a4, a6);
  b6 = _mm256_unpacklo_epi64(a5, a7);
  b7 = _mm256_unpackhi_epi64(a5, a7);
  r[0] = (lift_v8)_mm256_permute2x128_si256(b0, b4, 0x20);
  r[1] = (lift_v8)_mm256_permute2x128_si256(b1, b5, 0x20);
  r[2] = (lift_v8)_mm256_permute2x128_si256(b2, b6, 0x20);
  r[3] = (lift_v8)_mm256_permute2x128_si256(b3, b7, 0x20);
  r[4] = (lift_v8)_mm256_permute2x128_si256(b0, b4, 0x31);
  r[5] = (lift_v8)_mm256_permute2x128_si256(b1, b5, 0x31);
  r[6] = (lift_v8)_mm256_permute2x128_si256(b2, b6, 0x31);
  r[7] = (lift_v8)_mm256_permute2x128_si256(b3, b7, 0x31);
}

static inline lift_v8 ld8(const od_coeff *p) {
  return (lift_v8)_mm256_loadu_si256((const __m256i *)p);
}
static inline void st8(od_coeff *p, lift_v8 v) {
  _mm256_storeu_si256((__m256i *)p, (__m256i)v);
}

/* forward pass: column c of `in` (stride istride) -> row c of `out` (stride ostride) */
#define FWD_PASS(N) \
  static void fwd_pass##N(od_coeff *out, int ostride, const od_coeff *in, int istride) { \
    int c0; \
    for (c0 = 0; c0 < N; c0 += 8) { \
      lift_v8 t[N]; \
      int j; \
      int k0; \
      for (j = 0; j < N; j++) t[j] = ld8(in + (size_t)j*istride + c0); \
      lift_fdct##N##_host(t); \
      for (k0 = 0; k0 < N; k0 += 8) { \
        int l; \
        transpose8(t + k0); \
        for (l = 0; l < 8; l++) st8(out + (size_t)(c0 + l)*ostride + k0, t[k0 + l]); \
      } \
    } \
  }
/* inverse pass: row i of `in` -> column i of `out` */
#define INV_PASS(N) \
  static void inv_pass##N(od_coeff *out, int ostride, const od_coeff *in, int istride) { \
    int i0; \
    for (i0 = 0; i0 < N; i0 += 8) { \
      lift_v8 t[N]; \
      int j; \
      int k0; \
      for (k0 = 0; k0 < N; k0 += 8) { \
        int l; \
        for (l = 0; l < 8; l++) t[k0 + l] = ld8(in + (size_t)(i0 + l)*istride + k0); \
        transpose8(t + k0); \
      } \
      lift_idct##N##_host(t); \
      for (j = 0; j < N; j++) st8(out + (size_t)j*ostride + i0, t[j]); \
    } \
  }
FWD_PASS(8)
FWD_PASS(16)
FWD_PASS(32)
INV_PASS(8)
INV_PASS(16)
INV_PASS(32)

#define DCT_2D(N) \
  void od_hipenc_fdct##N##x##N(od_coeff *y, int ystride, const od_coeff *x, int xstride) { \
    od_coeff z[N*N] __attribute__((aligned(32))); \
    fwd_pass##N(z, N, x, xstride); \
    fwd_pass##N(y, ystride, z, N); \
  } \
  void od_hipenc_idct##N##x##N(od_coeff *x, int xstride, const od_coeff *y, int ystride) { \
    od_coeff z[N*N] __attribute__((aligned(32))); \
    inv_pass##N(z, N, y, ystride); \
    inv_pass##N(x, xstride, z, N); \
  }
DCT_2D(8)
DCT_2D(16)
DCT_2D(32)

/* 4x4: four columns per vector, 4x4 transposes */
static inline lift_v4 ld4(const od_coeff *p) {
  return (lift_v4)_mm_loadu_si128((const __m128i *)p);
}
static inline void st4(od_coeff *p, lift_v4 v) {
  _mm_storeu_si128((__m128i *)p, (__m128i)v);
}
static inline void transpose4(lift_v4 *r) {
  __m128i a0, a1, a2, a3;
  a0 = _mm_unpacklo_epi32((__m128i)r[0], (__m128i)r[1]);
  a1 = _mm_unpackhi_epi32((__m128i)r[0], (__m128i)r[1]);
  a2 = _mm_unpacklo_epi32((__m128i)r[2], (__m128i)r[3]);
  a3 = _mm_unpackhi_epi32((__m128i)r[2], (__m128i)r[3]);
  r[0] = (lift_v4)_mm_unpacklo_epi64(a0, a2);
  r[1] = (lift_v4)_mm_unpackhi_epi64(a0, a2);
  r[2] = (lift_v4)_mm_unpacklo_epi64(a1, a3);
  r[3] = (lift_v4)_mm_unpackhi_epi64(a1, a3);
}

void od_hipenc_fdct4x4(od_coeff *y, int ystride, const od_coeff *x, int xstride) {
  lift_v4 t[4];
  int j;
  for (j = 0; j < 4; j++) t[j] = ld4(x + (size_t)j*xstride);
  lift_fdct4_host(t);
  transpose4(t);                 /* rows of z */
  lift_fdct4_host(t);            /* t[j] = row j of z; lanes = columns of z: down the columns */
  transpose4(t);
  for (j = 0; j < 4; j++) st4(y + (size_t)j*ystride, t[j]);
}

void od_hipenc_idct4x4(od_coeff *x, int xstride, const od_coeff *y, int ystride) {
  lift_v4 t[4];
  int j;
  for (j = 0; j < 4; j++) t[j] = ld4(y + (size_t)j*ystride);
  transpose4(t);                 /* lane l = row l of y */
  lift_idct4_host(t);            /* t[j][l] = z[j][l]: row l of y -> column l of z */
  transpose4(t);                 /* lane l = row l of z */
  lift_idct4_host(t);            /* t[j][l] = x[j][l] */
  for (j = 0; j < 4; j++) st4(x + (size_t)j*xstride, t[j]);
}
