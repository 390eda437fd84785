// fp32 -> two fp16 parts (x = hi + lo after an exact power-of-two scaling), shared by the weight-image kernel and the
// chain kernel's in-register activation split; and the bf16 single-product form (gfv_set_f16split(3)): the SAME fragment
// layouts and power-of-two scales (harmless for bf16, whose exponent range is fp32's), the high part rounded to bf16 instead of
// fp16 and multiplied on v_mfma_f32_16x16x32_bf16; the low part is not formed.
#pragma once
#include <hip/hip_runtime.h>

typedef _Float16 gfv_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gfv_f16x2 __attribute__((ext_vector_type(2)));
typedef float gfv_float2 __attribute__((ext_vector_type(2)));
typedef unsigned gfv_uint4 __attribute__((ext_vector_type(4)));

// exact power of two s with s * m in [2^13, 2^14) (fp16 overflows at 2^16); m = 0 / subnormal: capped at 2^126.
// Undo it with a multiplication by 1 / s on its own (the product of two such scales may leave the fp32 range).
__device__ __forceinline__ float gfv_pow2_scale(float m) {
  const int e = (int)((__float_as_uint(m) >> 23) & 255u);  // biased exponent of m >= 0
  const int se = min(max(267 - e, 1), 253);                // biased exponent of 2^(13 - (e - 127))
  return __uint_as_float((unsigned)se << 23);
}

// smallest power of two >= x (x > 0, finite; clamped to [2^-20, 2^40])
__device__ __forceinline__ float gfv_pow2_ceil(float x) {
  const unsigned u = __float_as_uint(x);
  int e = (int)((u >> 23) & 255u) + ((u & 0x7fffffu) ? 1 : 0);
  e = min(max(e, 107), 167);
  return __uint_as_float((unsigned)e << 23);
}

__device__ __forceinline__ unsigned gfv_pk_f16(float a, float b) {
  const gfv_f16x2 v = __builtin_convertvector(gfv_float2{a, b}, gfv_f16x2);
  return __builtin_bit_cast(unsigned, v);
}
// (a, b) -> packed hi halves, packed lo halves (lo = fp16(x - hi), exact residual before the rounding)
// The residual x - float(hi) is ONE v_fma_mix_f32 per value (x * 1.0 + (-hi) with the fp16 half read in place; exact, so the
// same bits as converting back and subtracting - checked on the GPU over 65 k random pairs): 4 instructions per pair
// instead of 5 (6 without the SLP vectoriser's packed subtraction).
__device__ __forceinline__ void gfv_split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  hi = gfv_pk_f16(a, b);
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(a), "v"(hi));
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(b), "v"(hi));
  lo = gfv_pk_f16(r0, r1);
}
__device__ __forceinline__ void gfv_split8(const float (&v)[8], gfv_uint4& hi, gfv_uint4& lo) {
  unsigned h0, h1, h2, h3, l0, l1, l2, l3;
  gfv_split_pair(v[0], v[1], h0, l0);
  gfv_split_pair(v[2], v[3], h1, l1);
  gfv_split_pair(v[4], v[5], h2, l2);
  gfv_split_pair(v[6], v[7], h3, l3);
  hi = gfv_uint4{h0, h1, h2, h3};
  lo = gfv_uint4{l0, l1, l2, l3};
}

// ---- bf16 single-product form ----
typedef __bf16 gfv_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gfv_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gfv_floatx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned gfv_pk_bf16(float a, float b) {   // v_cvt_pk_bf16_f32 (round to nearest even)
  const gfv_bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
// the high parts of a pair in the single-product form's type (BF: bf16, else fp16); `lo` is left alone
template <bool BF>
__device__ __forceinline__ unsigned gfv_hi_pair(float a, float b) {
  return BF ? gfv_pk_bf16(a, b) : gfv_pk_f16(a, b);
}
template <bool BF>
__device__ __forceinline__ void gfv_split_pair_t(float a, float b, unsigned& hi, unsigned& lo) {
  if (BF) {
    hi = gfv_pk_bf16(a, b);
    lo = 0u;
  } else {
    gfv_split_pair(a, b, hi, lo);
  }
}
template <bool BF>
__device__ __forceinline__ void gfv_split8_t(const float (&v)[8], gfv_uint4& hi, gfv_uint4& lo) {
  if (BF) {
    hi = gfv_uint4{gfv_pk_bf16(v[0], v[1]), gfv_pk_bf16(v[2], v[3]), gfv_pk_bf16(v[4], v[5]), gfv_pk_bf16(v[6], v[7])};
    lo = gfv_uint4{0u, 0u, 0u, 0u};
  } else {
    gfv_split8(v, hi, lo);
  }
}
// hi x hi product of the single-product forms: operands travel as the 16-byte fragments of the fp16 form whatever their type
template <bool BF>
__device__ __forceinline__ gfv_floatx4 gfv_mma_hh(const gfv_f16x8& a, const gfv_f16x8& b, const gfv_floatx4& c) {
  if (BF)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gfv_bf16x8, a), __builtin_bit_cast(gfv_bf16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// 1.0 in every slot of a fragment, in the form's type (the bias gradient's second operand)
template <bool BF>
__device__ __forceinline__ gfv_f16x8 gfv_frag_ones() {
  const unsigned u = BF ? 0x3f803f80u : 0x3c003c00u;
  return __builtin_bit_cast(gfv_f16x8, gfv_uint4{u, u, u, u});
}
