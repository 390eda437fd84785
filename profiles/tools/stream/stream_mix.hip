// HBM roofline for the byte mix of the fused-MLP launches: every thread reads NR float4 streams and writes NW float4 streams
// of [rows, 128] fp32, either linearly (a wave touches 1 KB contiguous per instruction) or in the chain kernels' pattern
// (a wave instruction touches 16 rows x 64 B).  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC stream_mix.hip -o libstream.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float fx4 __attribute__((ext_vector_type(4)));

template <int NR, int NW, bool PIECES, int NT>
__global__ __launch_bounds__(256) void mix_kernel(const float4* __restrict__ in, float4* __restrict__ out, long rows, long stride4) {
  // a "row" = 32 float4; thread t of a wave: PIECES ? (row j = t & 15, piece g = t >> 4 of 16-column block b) : linear
  const long nvec = rows * 32;
  for (long base = ((long)blockIdx.x * 256 + threadIdx.x); base < nvec; base += (long)gridDim.x * 256) {
    long i = base;
    if (PIECES) {
      const long wavebase = base & ~63L;          // 64 consecutive float4 = 2 rows; remap to 16 rows x 4 float4
      const int lane = (int)(base & 63);
      const long grp = wavebase >> 9;             // 512 float4 = 16 rows
      const int blk = (int)((wavebase >> 6) & 7); // which 16-column block of the 128 columns
      i = (grp * 16 + (lane & 15)) * 32 + blk * 4 + (lane >> 4);
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      float4 x;
      if (NT & 1) { const fx4 t = __builtin_nontemporal_load(reinterpret_cast<const fx4*>(in + i + r * stride4)); x = make_float4(t[0], t[1], t[2], t[3]); }
      else x = in[i + r * stride4];
      v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      v.x += 1.0f;
      if (NT & 2) __builtin_nontemporal_store(fx4{v.x, v.y, v.z, v.w}, reinterpret_cast<fx4*>(out + i + w * stride4));
      else out[i + w * stride4] = v;
    }
  }
}

extern "C" int stream_mix(const void* in, void* out, long rows, long stride4, int nr, int nw, int pieces, int blocks, int nt, void* stream) {
#define CASE(R, W)                                                                                                              \
  if (nr == R && nw == W) {                                                                                                     \
    if (pieces && nt == 0) hipLaunchKernelGGL((mix_kernel<R, W, true, 0>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)in, (float4*)out, rows, stride4); \
    else if (pieces && nt == 2) hipLaunchKernelGGL((mix_kernel<R, W, true, 2>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)in, (float4*)out, rows, stride4); \
    else if (pieces && nt == 3) hipLaunchKernelGGL((mix_kernel<R, W, true, 3>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)in, (float4*)out, rows, stride4); \
    else if (nt == 0) hipLaunchKernelGGL((mix_kernel<R, W, false, 0>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)in, (float4*)out, rows, stride4); \
    else if (nt == 2) hipLaunchKernelGGL((mix_kernel<R, W, false, 2>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)in, (float4*)out, rows, stride4); \
    else hipLaunchKernelGGL((mix_kernel<R, W, false, 3>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)in, (float4*)out, rows, stride4); \
    return 0;                                                                                                                   \
  }
  CASE(1, 1) CASE(0, 1) CASE(2, 5) CASE(1, 5) CASE(5, 2) CASE(1, 3) CASE(3, 1) CASE(0, 4) CASE(2, 2) CASE(4, 4)
  return -1;
}
