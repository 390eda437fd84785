#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(const short* in, short* out) {
  __shared__ short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  // every lane reads 8 bytes at its own address: lane l -> lds + 4 * l (shorts)  [16 lanes x 4 shorts = a 16 x 4 block]
  const int l = threadIdx.x;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + 4 * l));
  for (int e = 0; e < 4; ++e) out[4 * l + e] = v[e];
}
int main() {
  short h[4096], o[256];
  for (int i = 0; i < 4096; ++i) h[i] = (short)i;
  short *d, *dout;
  hipMalloc(&d, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3]);
  return 0;
}
