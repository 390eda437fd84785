// Issue rate of scalar vs packed fp32 vector arithmetic and of the transcendental unit on gfx950 (one wave, one SIMD, or two
// waves on one SIMD): cycles per instruction from s_memtime around an unrolled loop of independent operations.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, float seed) {
  float a[8];
  f2 p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f2{seed + i, seed - i}; }
  const float m = 1.0000001f, c = 1e-9f;
  const f2 m2 = {m, m}, c2 = {c, c};
  __syncthreads();
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < 1000; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) a[i] = __builtin_fmaf(a[i], m, c);                               // v_fma_f32
        if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], m2, c2);                  // v_pk_fma_f32
        if (MODE == 2) a[i] = __builtin_amdgcn_exp2f(a[i]);                             // v_exp_f32
        if (MODE == 3) a[i] = __builtin_amdgcn_rcpf(a[i]);                              // v_rcp_f32
        if (MODE == 4) p[i] = p[i] * m2;                                                // v_pk_mul_f32
      }
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;   // per wave: the oldest wave of a SIMD wins the arbitration
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096);
  const char* names[5] = {"v_fma_f32", "v_pk_fma_f32 (2 values)", "v_exp_f32", "v_rcp_f32", "v_pk_mul_f32 (2 values)"};
  for (int waves = 1; waves <= 16; waves *= 2) {    // waves per workgroup: 1 = one wave alone on a SIMD, 4 = one per SIMD, 8 = two per SIMD, 16 = four
    for (int mode = 0; mode < 5; ++mode) {
      long long h[16] = {0};
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f);
        if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 8 * waves, hipMemcpyDeviceToHost);
      }
      long long lo = h[0], hi = h[0], sum = 0;
      for (int w = 0; w < waves; ++w) { lo = h[w] < lo ? h[w] : lo; hi = h[w] > hi ? h[w] : hi; sum += h[w]; }
      printf("%2d wave(s) per workgroup  %-26s %6.2f clock64 ticks per instruction per wave (min %.2f, max %.2f over the waves)\n", waves,
             names[mode], (double)sum / waves / 64000.0, (double)lo / 64000.0, (double)hi / 64000.0);
    }
  }
  return 0;
}
