// Feasibility / accuracy micro-benchmark behind the split-fp16 GEMM form: out[r][n] = sum_k X[r][k] W[n][k]
// (K = N = 128, weights resident in LDS in fragment order) as f16 x3 (two-part fp16 split, per-row scaling), bf16 x6, bf16 x3,
// bf16 x1 and on the f32 MFMA; prints max |err| / sum|a b| against float64 and the fp32-equivalent rate.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 split_gemm_bench.hip -o split_gemm_bench; WIDE=1 for wide-dynamic-range rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pk(float a, float b) {
  const bf16x2 v = __builtin_convertvector(float2v{a, b}, bf16x2);
  return __builtin_bit_cast(unsigned, v);
}
// two floats -> three packed bf16 pairs (x = p0 + p1 + p2 to ~2^-24)
__device__ __forceinline__ void split3_pair(float xa, float xb, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = pk(xa, xb);
  const float ra = xa - __uint_as_float(p0 << 16), rb = xb - __uint_as_float(p0 & 0xffff0000u);
  p1 = pk(ra, rb);
  const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xffff0000u);
  p2 = pk(sa, sb);
}
// 8 consecutive floats -> three bf16x8 fragments
__device__ __forceinline__ void split3_frag(const float* src, bf16x8& f0, bf16x8& f1, bf16x8& f2) {
  const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
  unsigned a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3;
  split3_pair(lo.x, lo.y, a0, b0, c0);
  split3_pair(lo.z, lo.w, a1, b1, c1);
  split3_pair(hi.x, hi.y, a2, b2, c2);
  split3_pair(hi.z, hi.w, a3, b3, c3);
  const uint4v a{a0, a1, a2, a3}, b{b0, b1, b2, b3}, c{c0, c1, c2, c3};
  f0 = __builtin_bit_cast(bf16x8, a); f1 = __builtin_bit_cast(bf16x8, b); f2 = __builtin_bit_cast(bf16x8, c);
}
__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)x;
  const float r1 = x - (float)a;
  b = (__bf16)r1;
  const float r2 = r1 - (float)b;
  c = (__bf16)r2;
}

// LDS image of the split weights: [kg 8][nt 4][part 3][lane 64] x 16 B = 96 KB
constexpr int WFRAG = 8 * 4 * 3 * 64;  // number of bf16x8 fragments

template <int TERMS>
__global__ __launch_bounds__(256, 2) void gemm_bf16x(const float* __restrict__ X, const float* __restrict__ W,
                                                     float* __restrict__ out, int reps, int kmask) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8* wl = reinterpret_cast<bf16x8*>(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;
  // stage W: fragment (kg, nt, lane') element e = W[32 nt + i][16 kg + 8 h' + e]
#pragma unroll 1
  for (int f = tid; f < (kmask + 1) * 4 * 64; f += 256) {
    const int l2 = f & 63, nt = (f >> 6) & 3, kg = f >> 8;
    const int i = l2 & 31, hh = l2 >> 5;
    bf16x8 p0, p1, p2;
    split3_frag(W + (32 * nt + i) * 128 + 16 * kg + 8 * hh, p0, p1, p2);
    wl[((kg * 4 + nt) * 3 + 0) * 64 + l2] = p0;
    wl[((kg * 4 + nt) * 3 + 1) * 64 + l2] = p1;
    wl[((kg * 4 + nt) * 3 + 2) * 64 + l2] = p2;
  }
  // activations of this wave's 32 rows: B operand, lane (j, h): X[row j][16 kg + 8 h + e]
  const int row = (blockIdx.x * 4 + wave) * 32 + j;
  bf16x8 x0[8], x1[8], x2[8];
#pragma unroll
  for (int kg = 0; kg < 8; ++kg) split3_frag(X + (size_t)row * 128 + 16 * kg + 8 * h, x0[kg], x1[kg], x2[kg]);
  __syncthreads();
  floatx16 acc[4];
  for (int n = 0; n < 4; ++n)
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
#pragma unroll 1
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const bf16x8 w0 = wl[(((kg & kmask) * 4 + nt) * 3 + 0) * 64 + lane];
        const bf16x8 w1 = wl[(((kg & kmask) * 4 + nt) * 3 + 1) * 64 + lane];
        const bf16x8 w2 = wl[(((kg & kmask) * 4 + nt) * 3 + 2) * 64 + lane];
        if (TERMS >= 6) {
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x1[kg], acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x0[kg], acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x2[kg], acc[nt], 0, 0, 0);
        }
        if (TERMS >= 3) {
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x0[kg], acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x1[kg], acc[nt], 0, 0, 0);
        }
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x0[kg], acc[nt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // D: lane (j, h) reg r -> n = 32 nt + (r & 3) + 8 (r >> 2) + 4 h, data row j
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<floatx4*>(out + (size_t)row * 128 + 32 * nt + 8 * q + 4 * h) =
          floatx4{acc[nt][4 * q], acc[nt][4 * q + 1], acc[nt][4 * q + 2], acc[nt][4 * q + 3]};
}


typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pkh(float a, float b) {
  const f16x2 v = __builtin_convertvector(float2v{a, b}, f16x2);
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split2_pair(float xa, float xb, unsigned& p0, unsigned& p1) {
  p0 = pkh(xa, xb);
  const f16x2 h = __builtin_bit_cast(f16x2, p0);
  p1 = pkh(xa - (float)h[0], xb - (float)h[1]);
}
__device__ __forceinline__ void split2_frag(const float* src, float sc, f16x8& f0, f16x8& f1) {
  const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
  unsigned a0, a1, a2, a3, b0, b1, b2, b3;
  split2_pair(lo.x * sc, lo.y * sc, a0, b0);
  split2_pair(lo.z * sc, lo.w * sc, a1, b1);
  split2_pair(hi.x * sc, hi.y * sc, a2, b2);
  split2_pair(hi.z * sc, hi.w * sc, a3, b3);
  const uint4v a{a0, a1, a2, a3}, b{b0, b1, b2, b3};
  f0 = __builtin_bit_cast(f16x8, a); f1 = __builtin_bit_cast(f16x8, b);
}
// exact power of two s with s * m in [2^13, 2^14)
__device__ __forceinline__ float pow2_scale(float m) {
  const int e = (int)((__float_as_uint(m) >> 23) & 255u);          // biased exponent of m (m >= 0)
  const int se = 127 + 13 - (e - 127);
  return m > 0.f ? __uint_as_float((unsigned)min(max(se, 1), 254) << 23) : 1.0f;
}
__global__ __launch_bounds__(256, 2) void gemm_f16x3(const float* __restrict__ X, const float* __restrict__ W,
                                                     float* __restrict__ out, int reps, int kmask, float wscale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f16x8* wl = reinterpret_cast<f16x8*>(smem);     // [kg][nt][part 2][lane 64]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;
#pragma unroll 1
  for (int f = tid; f < (kmask + 1) * 4 * 64; f += 256) {
    const int l2 = f & 63, nt = (f >> 6) & 3, kg = f >> 8;
    const int i = l2 & 31, hh = l2 >> 5;
    f16x8 p0, p1;
    split2_frag(W + (32 * nt + i) * 128 + 16 * kg + 8 * hh, wscale, p0, p1);
    wl[((kg * 4 + nt) * 2 + 0) * 64 + l2] = p0;
    wl[((kg * 4 + nt) * 2 + 1) * 64 + l2] = p1;
  }
  const int row = (blockIdx.x * 4 + wave) * 32 + j;
  // row scale: max |x| over the row's 128 values (this lane holds 64 of them, the lane with the other h the rest)
  float m = 0.f;
#pragma unroll
  for (int kg = 0; kg < 8; ++kg)
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(X[(size_t)row * 128 + 16 * kg + 8 * h + e]));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  const float sx = pow2_scale(m);
  f16x8 x0[8], x1[8];
#pragma unroll
  for (int kg = 0; kg < 8; ++kg) split2_frag(X + (size_t)row * 128 + 16 * kg + 8 * h, sx, x0[kg], x1[kg]);
  __syncthreads();
  floatx16 acc[4];
  for (int n = 0; n < 4; ++n)
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
#pragma unroll 1
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const f16x8 w0 = wl[(((kg & kmask) * 4 + nt) * 2 + 0) * 64 + lane];
        const f16x8 w1 = wl[(((kg & kmask) * 4 + nt) * 2 + 1) * 64 + lane];
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x0[kg], acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, x1[kg], acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, x0[kg], acc[nt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const float inv = 1.0f / (sx * wscale);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<floatx4*>(out + (size_t)row * 128 + 32 * nt + 8 * q + 4 * h) =
          floatx4{acc[nt][4 * q] * inv, acc[nt][4 * q + 1] * inv, acc[nt][4 * q + 2] * inv, acc[nt][4 * q + 3] * inv};
}

// f32 MFMA reference kernel, same shape: wave = 16 rows, W fragments from LDS (fp32, 64 KB)
__global__ __launch_bounds__(256, 2) void gemm_f32(const float* __restrict__ X, const float* __restrict__ W,
                                                   float* __restrict__ out, int reps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  floatx4* wl = reinterpret_cast<floatx4*>(smem);   // [t 8][nt 8][lane 64] float4: W[16 nt + i][16 t + 4 g + s]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, g = lane >> 4;
#pragma unroll 1
  for (int f = tid; f < 8 * 8 * 64; f += 256) {
    const int l2 = f & 63, nt = (f >> 6) & 7, t = f >> 9;
    floatx4 v;
    for (int s = 0; s < 4; ++s) v[s] = W[(16 * nt + (l2 & 15)) * 128 + 16 * t + 4 * (l2 >> 4) + s];
    wl[f] = v;
  }
  const int row = (blockIdx.x * 4 + wave) * 16 + i;
  float x[8][4];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float4 v = *reinterpret_cast<const float4*>(X + (size_t)row * 128 + 16 * t + 4 * g);
    x[t][0] = v.x; x[t][1] = v.y; x[t][2] = v.z; x[t][3] = v.w;
  }
  __syncthreads();
  floatx4 acc[8];
  for (int n = 0; n < 8; ++n) acc[n] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      floatx4 w[8];
#pragma unroll
      for (int n = 0; n < 8; ++n) w[n] = wl[(t * 8 + n) * 64 + lane];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n][s], x[t][s], acc[n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int n = 0; n < 8; ++n) *reinterpret_cast<floatx4*>(out + (size_t)row * 128 + 16 * n + 4 * g) = acc[n];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main() {
  const int WGS = 2048;
  const int rows = WGS * 4 * 32;
  std::vector<float> hX((size_t)rows * 128), hW(128 * 128);
  srand(1);
  for (size_t i = 0; i < hX.size(); ++i) {
    const int r = (int)(i / 128);
    float v = (float)rand() / RAND_MAX * 2.f - 1.f;
    if (getenv("WIDE")) v *= powf(10.f, -(float)(r % 12)) * powf(10.f, -6.f * (float)rand() / RAND_MAX);
    hX[i] = v;
  }
  for (auto& v : hW) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.1f;
  float *dX, *dW, *dO;
  CK(hipMalloc(&dX, hX.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&dO, hX.size() * 4));
  CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)gemm_bf16x<6>, hipFuncAttributeMaxDynamicSharedMemorySize, WFRAG * 16));
  CK(hipFuncSetAttribute((const void*)gemm_bf16x<3>, hipFuncAttributeMaxDynamicSharedMemorySize, WFRAG * 16));
  CK(hipFuncSetAttribute((const void*)gemm_bf16x<1>, hipFuncAttributeMaxDynamicSharedMemorySize, WFRAG * 16));
  CK(hipFuncSetAttribute((const void*)gemm_f32, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  std::vector<float> hO(hX.size());
  auto check = [&](const char* name) {
    CK(hipMemcpy(hO.data(), dO, 4096 * 128 * 4, hipMemcpyDeviceToHost));
    double worst = 0, worst_rel_norm = 0;
    for (int r = 0; r < 4096; ++r) {
      for (int n = 0; n < 128; ++n) {
        double ref = 0, mag = 0;
        for (int k = 0; k < 128; ++k) { const double p = (double)hX[(size_t)r * 128 + k] * hW[n * 128 + k]; ref += p; mag += fabs(p); }
        const double e = fabs(hO[(size_t)r * 128 + n] - ref) / mag;
        if (e > worst) worst = e;
      }
    }
    printf("%-10s max |err| / sum|a b| = %.3e\n", name, worst);
  };
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timeit = [&](const char* name, auto launch, int rows_done, int reps) {
    launch(reps); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); launch(reps); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double fl = 2.0 * rows_done * 128.0 * 128.0 * reps;
    printf("%-10s %8.3f ms  %7.1f fp32-equivalent TFLOP/s\n", name, ms, fl / ms * 1e-9);
  };
  auto L6 = [&](int reps) { const int km = reps > 1 ? 3 : 7; hipLaunchKernelGGL(gemm_bf16x<6>, dim3(WGS), dim3(256), (km + 1) * 4 * 3 * 64 * 16, 0, dX, dW, dO, reps, km); };
  auto L3 = [&](int reps) { const int km = reps > 1 ? 3 : 7; hipLaunchKernelGGL(gemm_bf16x<3>, dim3(WGS), dim3(256), (km + 1) * 4 * 3 * 64 * 16, 0, dX, dW, dO, reps, km); };
  auto L1 = [&](int reps) { const int km = reps > 1 ? 3 : 7; hipLaunchKernelGGL(gemm_bf16x<1>, dim3(WGS), dim3(256), (km + 1) * 4 * 3 * 64 * 16, 0, dX, dW, dO, reps, km); };
  auto LF = [&](int reps) { hipLaunchKernelGGL(gemm_f32, dim3(WGS * 2), dim3(256), 65536, 0, dX, dW, dO, reps); };
  float wmax = 0; for (auto v : hW) wmax = fmaxf(wmax, fabsf(v));
  int ex; frexpf(wmax, &ex); const float wscale = ldexpf(1.0f, 14 - ex);   // wmax * wscale in [2^13, 2^14)
  CK(hipFuncSetAttribute((const void*)gemm_f16x3, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  auto LH = [&](int reps) { const int km = reps > 1 ? 3 : 7; hipLaunchKernelGGL(gemm_f16x3, dim3(WGS), dim3(256), (km + 1) * 4 * 2 * 64 * 16, 0, dX, dW, dO, reps, km, wscale); };
  LH(1); CK(hipDeviceSynchronize()); check("f16x3");
  L6(1); CK(hipDeviceSynchronize()); check("bf16x6");
  L3(1); CK(hipDeviceSynchronize()); check("bf16x3");
  L1(1); CK(hipDeviceSynchronize()); check("bf16x1");
  LF(1); CK(hipDeviceSynchronize()); check("f32 mfma");
  timeit("f16x3", LH, rows, 64);
  timeit("bf16x6", L6, rows, 64);
  timeit("bf16x3", L3, rows, 64);
  timeit("bf16x1", L1, rows, 64);
  timeit("f32 mfma", LF, rows, 64);
  return 0;
}
