// How the achieved bandwidth of a read / write mix depends on the contiguous run a wave instruction touches per row:
// RUN float4 per row (4 = 64 B: the MFMA accumulator layout's natural store, 8 = 128 B, 16, 32 = a whole 512-B row,
// 64 = two rows = 1 KB linear); a wave instruction covers 64 / RUN rows.  NT: 0 plain, 2 nontemporal stores, 3 + loads.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float fx4 __attribute__((ext_vector_type(4)));

template <int NR, int NW, int RUN, int NT>
__global__ __launch_bounds__(256) void run_kernel(const fx4* __restrict__ in, fx4* __restrict__ out, long rows, long stride4) {
  const long nvec = rows * 32;
  for (long base = ((long)blockIdx.x * 256 + threadIdx.x); base < nvec; base += (long)gridDim.x * 256) {
    long i = base;
    if (RUN < 64) {
      // a block of 16 rows (512 float4) is covered by 8 wave instructions; instruction k of the block, lane l:
      constexpr int RPI = 64 / RUN;            // rows per instruction
      const long blk16 = base >> 9;            // 16-row block
      const int k = (int)((base >> 6) & 7), lane = (int)(base & 63);
      const int r = lane / RUN, c = lane % RUN;                 // row within the instruction, float4 within the run
      const int inst_rows = (k * RPI) % 16, inst_col = ((k * RPI) / 16) * RUN;
      i = (blk16 * 16 + inst_rows + r) * 32 + inst_col + c;
    }
    fx4 v = fx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < NR; ++r) v += (NT & 1) ? __builtin_nontemporal_load(in + i + r * stride4) : in[i + r * stride4];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      v += 1.0f;
      if (NT & 2) __builtin_nontemporal_store(v, out + i + w * stride4);
      else out[i + w * stride4] = v;
    }
  }
}

template <int NR, int NW>
static int launch(const void* in, void* out, long rows, long stride4, int run, int blocks, int nt, hipStream_t st) {
#define GO(RUN, NT) hipLaunchKernelGGL((run_kernel<NR, NW, RUN, NT>), dim3(blocks), dim3(256), 0, st, (const fx4*)in, (fx4*)out, rows, stride4)
#define PICK(RUN) if (run == RUN) { if (nt == 0) GO(RUN, 0); else if (nt == 2) GO(RUN, 2); else GO(RUN, 3); return 0; }
  PICK(4) PICK(8) PICK(16) PICK(32) PICK(64)
  return -1;
}
extern "C" int stream_run(const void* in, void* out, long rows, long stride4, int nr, int nw, int run, int blocks, int nt, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (nr == 2 && nw == 5) return launch<2, 5>(in, out, rows, stride4, run, blocks, nt, st);
  if (nr == 5 && nw == 2) return launch<5, 2>(in, out, rows, stride4, run, blocks, nt, st);
  if (nr == 2 && nw == 2) return launch<2, 2>(in, out, rows, stride4, run, blocks, nt, st);
  if (nr == 1 && nw == 1) return launch<1, 1>(in, out, rows, stride4, run, blocks, nt, st);
  return -1;
}
