#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* in, float* out, int nfloats) {
  __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, nfloats * 4, 0x00020000);
  __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, nfloats * 4, 0x00020000);
  const int off = (blockIdx.x * 64 + threadIdx.x) * 16;
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ri, off, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b128(v, ro, off, 0, 0);
  // out of bounds: dropped / zero
  i32x4 z = __builtin_amdgcn_raw_buffer_load_b128(ri, 0x7ffffff0, 0, 0);
  if (z[0] != 0) out[0] = -777.f;
  __builtin_amdgcn_raw_buffer_store_b128(v, ro, 0x7ffffff0, 0, 0);
}
int main() {
  const int n = 4096;
  float h[n], o[n];
  for (int i = 0; i < n; ++i) { h[i] = (float)i; o[i] = -1.f; }
  float *d, *dout;
  (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&dout, sizeof(o));
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  (void)hipMemcpy(dout, o, sizeof(o), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(8), dim3(64), 0, 0, d, dout, n / 2);   // the second half is out of bounds
  (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) { const float want = i < n / 2 ? (float)i : -1.f; if (o[i] != want) { if (bad < 5) printf("i %d got %f want %f\n", i, o[i], want); ++bad; } }
  printf("bad %d\n", bad);
  return 0;
}
