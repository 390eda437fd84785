// What does a kernel boundary cost on this GPU, and what would a device-side grid barrier cost instead?  (round 5, VERDICT r4 item 1:
// "more than half of the headline step is a size-independent latency floor").
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_floor profiles/tools/launch_floor/launch_floor.hip && /tmp/launch_floor
// Chains of DEPENDENT launches on one stream (eager and as a hipGraph replay), timed with events over the whole chain:
//   empty kernels of several grid shapes; a kernel that stages a 64 / 128 KB image in LDS and stores 16 B per thread (the skeleton of
//   lin1_kernel); a streaming copy of 1 ... 256 MB (floor + slope of a bandwidth kernel whose input the previous launch wrote);
//   a 3-round-trip pointer chase per thread (the skeleton of seg_gather_sum: rowptr -> col -> row);
// and ONE persistent launch of 256 workgroups that runs the same phases separated by grid barriers (device-scope counter):
//   barrier alone, barrier + 64 KB written per workgroup and read back by another workgroup (release / acquire traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_empty(int* p) { if (p && threadIdx.x == 100000) p[0] = 1; }

__global__ __launch_bounds__(512) void k_image(const uint4* img, int n16, float4* out) {
  extern __shared__ uint4 lds[];
  for (int i = threadIdx.x; i < n16; i += 512) lds[i] = img[i];
  __syncthreads();
  const uint4 v = lds[(threadIdx.x * 7) % n16];
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = make_float4(__uint_as_float(v.x), 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

// rows of 128 floats; thread group of 32 lanes per row: rowptr -> col -> src row (one neighbour) -> out row
__global__ __launch_bounds__(256) void k_chase(const int* __restrict__ rowptr, const int* __restrict__ col, const float4* __restrict__ src,
                                               float4* __restrict__ out, int rows) {
  const int r = (blockIdx.x * 256 + threadIdx.x) >> 5, l = threadIdx.x & 31;
  if (r >= rows) return;
  const int k = rowptr[r];
  const int c = col[k];
  out[(size_t)r * 32 + l] = src[(size_t)c * 32 + l];
}

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned& target, unsigned nwg) {
  __syncthreads();
  if (threadIdx.x == 0) {
    target += nwg;
    __threadfence();                                             // release: this workgroup's stores are visible device-wide
    atomicAdd(ctr, 1u);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    __threadfence();                                             // acquire
  }
  __syncthreads();
}

// mode 0: barriers only; 1: every workgroup writes `kb` KB, barrier, reads the KBs of workgroup (b + 97) % n (other XCD), barrier
__global__ __launch_bounds__(512) void k_persist(unsigned* ctr, int rounds, int mode, int kb, float4* buf, float* sink) {
  unsigned target = 0;
  const unsigned nwg = gridDim.x;
  float acc = 0.f;
  const size_t per = (size_t)kb * 64;   // float4 per workgroup
  for (int r = 0; r < rounds; ++r) {
    if (mode == 1) {
      float4* mine = buf + (size_t)blockIdx.x * per;
      for (size_t i = threadIdx.x; i < per; i += 512) mine[i] = make_float4((float)r, acc, 1.f, 2.f);
    }
    grid_barrier(ctr, target, nwg);
    if (mode == 1) {
      const float4* other = buf + (size_t)((blockIdx.x + 97) % nwg) * per;
      for (size_t i = threadIdx.x; i < per; i += 512) { const float4 v = other[i]; acc += v.x - (float)r; }
      grid_barrier(ctr, target, nwg);
    }
  }
  if (acc != 0.f) sink[blockIdx.x * 512 + threadIdx.x] = acc;   // (acc stays 0 when every read saw the round's value)
}

template <class F>
static double time_chain(hipStream_t st, int reps, F launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) launch();
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return 1e3 * ms / reps;
}
template <class F>
static double time_graph(hipStream_t st, int chain, int reps, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < chain; ++i) launch();
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return 1e3 * ms / (reps * chain);
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  printf("device %s, %d CUs, clock %d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
  int* dummy; CK(hipMalloc(&dummy, 64));
  printf("\n# dependent launches on one stream: microseconds per launch (eager launches | hipGraph replay of a 200-launch chain)\n");
  struct { int g, b; } shapes[] = {{1, 64}, {256, 64}, {256, 512}, {1024, 256}, {4096, 256}, {16384, 256}};
  for (auto s : shapes) {
    auto l = [&] { hipLaunchKernelGGL(k_empty, dim3(s.g), dim3(s.b), 0, st, dummy); };
    printf("empty  grid %6d x %3d : %6.2f | %6.2f\n", s.g, s.b, time_chain(st, 2000, l), time_graph(st, 200, 10, l));
  }
  // image staging
  uint4* img; float4* out;
  CK(hipMalloc(&img, 131072)); CK(hipMemset(img, 0, 131072));
  CK(hipMalloc(&out, (size_t)4096 * 512 * 16));
  CK(hipFuncSetAttribute((const void*)k_image, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  for (int kb : {16, 64, 128}) for (int g : {41, 200, 800}) {
    auto l = [&] { hipLaunchKernelGGL(k_image, dim3(g), dim3(512), kb * 1024, st, img, kb * 64, out); };
    printf("image %3d KB -> LDS, grid %4d x 512 : %6.2f | %6.2f\n", kb, g, time_chain(st, 1000, l), time_graph(st, 200, 10, l));
  }
  // streaming copy: ping-pong so that every launch reads what the previous one wrote
  printf("\n# streaming copy a -> b -> a (each launch reads what the previous one wrote): us per launch, GB/s (read + write)\n");
  const size_t maxb = (size_t)256 << 20;
  float4 *a, *b;
  CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb));
  CK(hipMemset(a, 0, maxb)); CK(hipMemset(b, 0, maxb));
  for (int mb : {0, 1, 2, 4, 8, 16, 32, 64, 128, 256}) {
    const size_t n = mb ? ((size_t)mb << 20) / 16 : 4096;
    int flip = 0;
    const int g = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    auto l = [&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, st, flip ? b : a, flip ? a : b, n); flip ^= 1; };
    const double us = time_chain(st, 400, l);
    printf("copy %4d MB (grid %4d) : %7.2f us  %7.1f GB/s\n", mb, g, us, 2.0 * n * 16 / us / 1e3);
  }
  // pointer chase
  printf("\n# rowptr -> col -> row gather (one neighbour per row, 512-byte rows): us per launch\n");
  for (int rows : {1024, 5184, 25479, 75499}) {
    std::vector<int> rp(rows + 1), cl(rows);
    for (int i = 0; i <= rows; ++i) rp[i] = i;
    for (int i = 0; i < rows; ++i) cl[i] = (int)(((long)i * 7919 + 13) % rows);
    int *drp, *dcl;
    CK(hipMalloc(&drp, (rows + 1) * 4)); CK(hipMalloc(&dcl, rows * 4));
    CK(hipMemcpy(drp, rp.data(), (rows + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dcl, cl.data(), rows * 4, hipMemcpyHostToDevice));
    int flip = 0;
    auto l = [&] { hipLaunchKernelGGL(k_chase, dim3((rows * 32 + 255) / 256), dim3(256), 0, st, drp, dcl, flip ? b : a, flip ? a : b, rows); flip ^= 1; };
    printf("gather rows %6d : %6.2f | graph %6.2f\n", rows, time_chain(st, 1000, l), time_graph(st, 200, 10, l));
    CK(hipFree(drp)); CK(hipFree(dcl));
  }
  // persistent kernel with grid barriers
  printf("\n# ONE persistent launch, 1 workgroup of 512 threads per CU, phases separated by a device-scope counter barrier: us per barrier\n");
  unsigned* ctr; float* sink;
  CK(hipMalloc(&ctr, 64)); CK(hipMalloc(&sink, (size_t)1024 * 512 * 4));
  const int nwg = pr.multiProcessorCount;
  for (int mode : {0, 1}) for (int kb : {4, 64, 256}) {
    if (mode == 0 && kb != 4) continue;
    const int rounds = 200;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemsetAsync(ctr, 0, 64, st));
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(k_persist, dim3(nwg), dim3(512), 0, st, ctr, rounds, mode, kb, a, sink);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const int nb = rounds * (mode ? 2 : 1);
    if (mode == 0) printf("barrier only                          : %6.2f us per barrier (%d barriers in %.1f us)\n", 1e3 * best / nb, nb, 1e3 * best);
    else printf("write %3d KB/WG | barrier | read other WG's | barrier : %6.2f us per (write + barrier + read + barrier) = %.1f GB/s\n", kb,
                1e3 * best / rounds, 2.0 * kb * 1024.0 * nwg / (1e3 * best / rounds) / 1e3);
  }
  return 0;
}
