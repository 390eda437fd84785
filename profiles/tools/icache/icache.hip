// Is a short launch's time its INSTRUCTION FETCH?  (round 5.)  The chain kernels are 14 - 31 KB of mostly straight-line code
// (profiles/r05_code_sizes.txt) and a launch of theirs takes 10 - 25 us however few rows it has.  This microbenchmark runs
// kernels made of N unique straight-line fused multiply-adds (8 - 12 bytes of code each, constants as literals) ONCE per wave,
// one 64-lane workgroup per CU, and the same code TWICE (second pass: instruction cache warm), back-to-back launches of one kernel
// and alternating launches of two different kernels (does the cache survive a kernel boundary?).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/icache profiles/tools/icache/icache.hip && /tmp/icache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int N, int SEED>
__device__ __forceinline__ float body(float x) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    // distinct literal constants per step: the compiler cannot roll this back into a loop
    const float c = 1.0f + 1e-6f * (float)((i * 7919 + SEED * 104729) % 1000003);
    const float d = 1e-7f * (float)((i * 15485863 + SEED) % 999983);
    x = __builtin_fmaf(x, c, d);
  }
  return x;
}
template <int N, int SEED, int PASSES>
__global__ __launch_bounds__(64) void k_code(float* out, int passes_rt) {
  float x = (float)threadIdx.x * 1e-3f;
  for (int p = 0; p < PASSES + passes_rt; ++p) x = body<N, SEED>(x);   // (passes_rt = 0: a run-time bound keeps the loop a loop)
  out[blockIdx.x * 64 + threadIdx.x] = x;
}

template <class F>
static double time_chain(hipStream_t st, int reps, F launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 10; ++i) launch(i);
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) launch(i);
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return 1e3 * ms / reps;
}

template <int N>
static void run(hipStream_t st, float* out, int grid) {
  auto a1 = [&](int) { hipLaunchKernelGGL((k_code<N, 1, 1>), dim3(grid), dim3(64), 0, st, out, 0); };
  auto a2 = [&](int) { hipLaunchKernelGGL((k_code<N, 1, 2>), dim3(grid), dim3(64), 0, st, out, 0); };
  auto a4 = [&](int) { hipLaunchKernelGGL((k_code<N, 1, 4>), dim3(grid), dim3(64), 0, st, out, 0); };
  auto ab = [&](int i) {
    if (i & 1) hipLaunchKernelGGL((k_code<N, 2, 1>), dim3(grid), dim3(64), 0, st, out, 0);
    else hipLaunchKernelGGL((k_code<N, 1, 1>), dim3(grid), dim3(64), 0, st, out, 0);
  };
  const double t1 = time_chain(st, 400, a1), t2 = time_chain(st, 400, a2), t4 = time_chain(st, 400, a4), tab = time_chain(st, 400, ab);
  printf("N = %5d fmas (~%3d KB of code), grid %3d: one pass %6.2f us | two passes %6.2f | four passes %6.2f  => warm pass %5.2f us, "
         "cold first pass %5.2f us | two kernels alternating %6.2f us per launch\n",
         N, N * 12 / 1024, grid, t1, t2, t4, (t4 - t2) / 2.0, t1 - (t4 - t2) / 2.0, tab);
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  float* out;
  CK(hipMalloc(&out, 1 << 20));
  printf("# straight-line code executed once per wave, 64-lane workgroups; launches back-to-back on one stream (us per launch)\n");
  for (int grid : {1, 256}) {
    run<256>(st, out, grid);
    run<1024>(st, out, grid);
    run<2048>(st, out, grid);
    run<4096>(st, out, grid);
  }
  return 0;
}
