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

static __device__ __forceinline__ float gfv_gelu(float x) {
  // exact (erf) GELU, nn.GELU() default (reference EPD.py:26)
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
static __device__ __forceinline__ float gfv_dgelu(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

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
