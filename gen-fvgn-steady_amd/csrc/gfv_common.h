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
// erfc(u), u >= 0, by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute): branch-free.  The chain kernels are bound by
// VALU issue as much as by memory latency (profiles/r01_tchain_split_pmc.txt: 8 VALU per MFMA, GELU nearly half of them),
// so the evaluation is trimmed to 11 plain + 2 transcendental instructions per value:
//   a = |x| sqrt(log2(e) / 2)  so that  exp(-x^2 / 2) = exp2(-a a)  is ONE multiplication in front of v_exp_f32;
//   t = 1 / (1 + p u),  u = |x| / sqrt 2 = a / sqrt(log2 e),  by v_rcp_f32 (1 ulp; the correctly rounded reciprocal is a
//   ten-instruction sequence and cost 2.8 % of the whole training step, profiles/tools/abn.sh frcp);
//   the polynomial's coefficients carry the factor 0.5 of  0.5 erfc.
// GELU / GELU' values differ from the exact ones by <= 3.5e-7 absolute, as with the untrimmed form (parity budget: 1e-5 relative).
// -DGFV_LIBM_ERF selects the library erff instead.
struct gfv_erfc_t {
  float y;  // 0.5 erfc(|x| / sqrt 2)
  float e;  // exp(-x^2 / 2)
};
static __device__ __forceinline__ gfv_erfc_t gfv_erfc_half(float x) {
  const float a = fabsf(x) * 0.84932180028801904272f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.2727374808792225f, a, 1.0f));   // 0.3275911 / sqrt(log2 e)
  gfv_erfc_t r;
  r.e = __builtin_amdgcn_exp2f(-(a * a));
  float p = fmaf(0.5307027145f, t, -0.7265760135f);
  p = fmaf(p, t, 0.7107068705f);
  p = fmaf(p, t, -0.142248368f);
  p = fmaf(p, t, 0.127414796f);
  r.y = (p * t) * r.e;
  return r;
}
// Two values at a time on the packed-fp32 instructions of gfx90a+ (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two fp32
// operations per lane and issue slot).  The chain kernels are bound by vector ISSUE (round 4: a fifth of the read traffic moves
// the persistent backward by 7 %, profiles/r04_colchain_phases.txt); the same operations in the same order as the scalar form
// above - bit-identical results - in 14 instead of 26 instructions per pair.  hipcc selects the packed forms for arithmetic on
// float2 ext-vectors (the SLP vectoriser, which would find them by itself, is off for these files: it shuffles registers).
typedef float gfv_f2 __attribute__((ext_vector_type(2)));
struct gfv_erfc2_t {
  gfv_f2 y, e;
};
static __device__ __forceinline__ gfv_f2 gfv_splat2(float v) { return gfv_f2{v, v}; }
static __device__ __forceinline__ gfv_erfc2_t gfv_erfc_half2(gfv_f2 x) {
  const gfv_f2 a = {fabsf(x.x) * 0.84932180028801904272f, fabsf(x.y) * 0.84932180028801904272f};
  const gfv_f2 d = __builtin_elementwise_fma(gfv_splat2(0.2727374808792225f), a, gfv_splat2(1.0f));
  const gfv_f2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  const gfv_f2 aa = a * a;
  gfv_erfc2_t r;
  r.e = gfv_f2{__builtin_amdgcn_exp2f(-aa.x), __builtin_amdgcn_exp2f(-aa.y)};
  gfv_f2 p = __builtin_elementwise_fma(gfv_splat2(0.5307027145f), t, gfv_splat2(-0.7265760135f));
  p = __builtin_elementwise_fma(p, t, gfv_splat2(0.7107068705f));
  p = __builtin_elementwise_fma(p, t, gfv_splat2(-0.142248368f));
  p = __builtin_elementwise_fma(p, t, gfv_splat2(0.127414796f));
  r.y = (p * t) * r.e;
  return r;
}
// gelu(x) and gelu'(x) of a pair from ONE erfc evaluation (what the scalar gfv_gelu / gfv_dgelu compute, value for value)
static __device__ __forceinline__ void gfv_gelu_dgelu2(gfv_f2 x, gfv_f2& g, gfv_f2& dg) {
  const gfv_erfc2_t r = gfv_erfc_half2(x);
  const gfv_f2 m = gfv_splat2(1.0f) - r.y;
  const gfv_f2 cdf = {x.x >= 0.0f ? m.x : r.y.x, x.y >= 0.0f ? m.y : r.y.y};
  dg = __builtin_elementwise_fma(x * gfv_splat2(0.39894228040143267794f), r.e, cdf);
  g = gfv_f2{fmaf(-fabsf(x.x), r.y.x, fmaxf(x.x, 0.0f)), fmaf(-fabsf(x.y), r.y.y, fmaxf(x.y, 0.0f))};
}
static __device__ __forceinline__ gfv_f2 gfv_gelu2(gfv_f2 x) {
  const gfv_erfc2_t r = gfv_erfc_half2(x);
  return gfv_f2{fmaf(-fabsf(x.x), r.y.x, fmaxf(x.x, 0.0f)), fmaf(-fabsf(x.y), r.y.y, fmaxf(x.y, 0.0f))};
}
static __device__ __forceinline__ gfv_f2 gfv_dgelu2(gfv_f2 x) {
  const gfv_erfc2_t r = gfv_erfc_half2(x);
  const gfv_f2 m = gfv_splat2(1.0f) - r.y;
  const gfv_f2 cdf = {x.x >= 0.0f ? m.x : r.y.x, x.y >= 0.0f ? m.y : r.y.y};
  return __builtin_elementwise_fma(x * gfv_splat2(0.39894228040143267794f), r.e, cdf);
}
#ifndef GFV_LIBM_ERF
static __device__ __forceinline__ float gfv_gelu(float x) {
  const gfv_erfc_t r = gfv_erfc_half(x);
  // x >= 0: 0.5 x (2 - erfc) = x - h;  x < 0: 0.5 x erfc = -h;  h = |x| 0.5 erfc(|x|/sqrt 2)
  return fmaf(-fabsf(x), r.y, fmaxf(x, 0.0f));
}
static __device__ __forceinline__ float gfv_dgelu(float x) {
  const gfv_erfc_t r = gfv_erfc_half(x);
  const float cdf = x >= 0.0f ? 1.0f - r.y : r.y;
  return fmaf(x * 0.39894228040143267794f, r.e, cdf);
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

// ---- cross-lane reductions without the LDS crossbar ---------------------------------------------------------------
// `__shfl_xor` is a ds_bpermute_b32: an LDS instruction plus an lgkmcnt wait per step.  Within a 16-lane DPP row the xor
// partners come from a DPP operand (quad_perm [1,0,3,2], [2,3,0,1], then row_half_mirror and row_mirror, which pair the
// quads / the halves once every lane of a quad / half holds the same partial); across rows gfx950's
// v_permlane16_swap / v_permlane32_swap exchange the odd and even rows / the two halves of two registers (two copies of
// x in, x and the partner's x out).  All are VALU instructions; every lane gets the result.
// (Inline asm for the swaps: with both operands the same value hipcc 7.2 folds the builtin's two results into one; the
// s_nop's are the VALU-write -> permlane-read wait states.)
#define GFV_DPP_F(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), 0xf, 0xf, true))
static __device__ __forceinline__ void gfv_lane_xor16(float x, float& a, float& b) {
  a = x; b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
static __device__ __forceinline__ void gfv_lane_xor32(float x, float& a, float& b) {
  a = x; b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
// sum over aligned groups of 16 lanes
static __device__ __forceinline__ float gfv_row16_sum(float v) {
  v += GFV_DPP_F(v, 0xB1);
  v += GFV_DPP_F(v, 0x4E);
  v += GFV_DPP_F(v, 0x141);
  v += GFV_DPP_F(v, 0x140);
  return v;
}
static __device__ __forceinline__ float gfv_row16_max(float v) {
  v = fmaxf(v, GFV_DPP_F(v, 0xB1));
  v = fmaxf(v, GFV_DPP_F(v, 0x4E));
  v = fmaxf(v, GFV_DPP_F(v, 0x141));
  v = fmaxf(v, GFV_DPP_F(v, 0x140));
  return v;
}
static __device__ __forceinline__ float gfv_row16_min(float v) {
  v = fminf(v, GFV_DPP_F(v, 0xB1));
  v = fminf(v, GFV_DPP_F(v, 0x4E));
  v = fminf(v, GFV_DPP_F(v, 0x141));
  v = fminf(v, GFV_DPP_F(v, 0x140));
  return v;
}
// sum over aligned groups of 32 lanes
static __device__ __forceinline__ float gfv_half_sum(float v) {
  float a, b;
  gfv_lane_xor16(gfv_row16_sum(v), a, b);
  return a + b;
}
// sum / maximum / minimum over the 64 lanes of a wave (all lanes get the result)
static __device__ __forceinline__ float gfv_wave_sum(float v) {
  float a, b;
  gfv_lane_xor16(gfv_row16_sum(v), a, b);
  gfv_lane_xor32(a + b, a, b);
  return a + b;
}
static __device__ __forceinline__ float gfv_wave_max(float v) {
  float a, b;
  gfv_lane_xor16(gfv_row16_max(v), a, b);
  gfv_lane_xor32(fmaxf(a, b), a, b);
  return fmaxf(a, b);
}
static __device__ __forceinline__ float gfv_wave_min(float v) {
  float a, b;
  gfv_lane_xor16(gfv_row16_min(v), a, b);
  gfv_lane_xor32(fminf(a, b), a, b);
  return fminf(a, b);
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

#include "gfv_launch.h"   // GFV_LAUNCH: every kernel launch, recordable (record.hip)
