// Common device helpers for the gfx950 (CDNA4, wave64) kernels of libgfv.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GFV_OK 0
#define GFV_ERR_ARG (-1)
#define GFV_ERR_LAUNCH (-2)

#define GFV_CHECK_LAUNCH()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return GFV_ERR_LAUNCH;  \
  } while (0)

typedef float floatx4 __attribute__((ext_vector_type(4)));

// exact (erf) GELU, nn.GELU() default (reference EPD.py:26).
// erfc(u), u >= 0, by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute): branch-free, 2 transcendental + ~12
// plain VALU ops per value, against ~40 with both divergent branches of the library erff.  The absolute error of
// 1 + erf is ~2e-7, i.e. GELU values differ from the library form by <= 2e-7 |x| (parity budget: 1e-5 relative).
// -DGFV_LIBM_ERF selects the library erff instead.
struct gfv_erfc_t {
  float y;  // erfc(|x| / sqrt 2)
  float e;  // exp(-x^2 / 2)
};
static __device__ __forceinline__ gfv_erfc_t gfv_erfc_half(float x) {
  const float u = fabsf(x) * 0.70710678118654752440f;
  const float t = __frcp_rn(fmaf(0.3275911f, u, 1.0f));
  gfv_erfc_t r;
  r.e = __expf(-u * u);
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  r.y = p * t * r.e;
  return r;
}
#ifndef GFV_LIBM_ERF
static __device__ __forceinline__ float gfv_gelu(float x) {
  const gfv_erfc_t r = gfv_erfc_half(x);
  // x >= 0: 0.5 x (2 - erfc) = x - h;  x < 0: 0.5 x erfc = -h;  h = 0.5 |x| erfc(|x|/sqrt 2)
  return fmaxf(x, 0.0f) - 0.5f * fabsf(x) * r.y;
}
static __device__ __forceinline__ float gfv_dgelu(float x) {
  const gfv_erfc_t r = gfv_erfc_half(x);
  const float cdf = x >= 0.0f ? 1.0f - 0.5f * r.y : 0.5f * r.y;
  return cdf + x * (0.39894228040143267794f * r.e);
}
#else
static __device__ __forceinline__ float gfv_gelu(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
static __device__ __forceinline__ float gfv_dgelu(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
#endif

// sum over the 64 lanes of a wave (all lanes get the result)
static __device__ __forceinline__ float gfv_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over aligned groups of 32 lanes
static __device__ __forceinline__ float gfv_half_sum(float v) {
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int gfv_div_up(long a, long b) { return (int)((a + b - 1) / b); }

// XCD-aware tile order.  The 256 CUs sit in 8 XCDs with one L2 each and the dispatcher deals consecutive workgroup ids
// round-robin over the XCDs (id w -> XCD w % 8).  Tiles that are neighbours in row order gather the same rows (an edge tile
// reads the rows of its end nodes, a node tile those of its neighbours; mesh numbering is spatially local), so giving each
// XCD a CONTIGUOUS range of tiles lets those gathers hit in its own L2 instead of being fetched once per XCD.
// Launch with gfv_xcd_grid(tiles) workgroups; a workgroup whose tile is >= tiles has nothing to do.
static inline int gfv_xcd_grid(int tiles) { return ((tiles + 7) / 8) * 8; }
static __device__ __forceinline__ int gfv_xcd_tile(int wg, int grid) {
#ifdef GFV_NO_XCD_REMAP
  return wg;
#else
  const int per = grid >> 3;
  return (wg & 7) * per + (wg >> 3);
#endif
}
