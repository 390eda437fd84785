// Per-mesh preprocessing on the device: the k-hop reconstruction stencil of a mesh.
// Reference: parse_to_h5.py:228-254 (extra stencil pairs from powers of the node adjacency), Load_mesh.py:421-521
// (construct_stencil).  Contract: include/gfv.h (gfv_khop_count / gfv_khop_fill).
//
// The host code (gfv/meshgen.py, validated against the reference's pipeline) multiplies sparse adjacency matrices and
// takes np.unique over all pairs; round 1's device form did the same with torch sorts and uniques.  Here a thread owns a
// node: it walks its neighbourhood breadth first through the CSR adjacency into a private visited list (a 2-hop
// neighbourhood of a triangle mesh is ~20 nodes, the list takes 512), keeps the nodes with a larger index and sorts them.
// Nodes in order and each node's partners ascending IS the lexicographic order of np.unique(axis=1): no global sort.
// Two passes (count, exclusive scan, fill); the CSR adjacency itself is built with integer atomics (counts and cursors:
// the order inside a row is arbitrary, the sorted output does not depend on it).  Nothing here is float arithmetic.
#include "gfv_common.h"
#include "../../include/gfv.h"

namespace {

constexpr int KHOP_CAP = 512;

__global__ __launch_bounds__(256) void degree_kernel(const int64_t* __restrict__ f0, const int64_t* __restrict__ f1, int F,
                                                     int* __restrict__ deg) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= F) return;
  atomicAdd(&deg[(int)f0[e]], 1);
  atomicAdd(&deg[(int)f1[e]], 1);
}

// exclusive scan of n ints into out[0..n] (out[n] = total) by ONE workgroup of 1024 threads, chunk after chunk
__global__ __launch_bounds__(1024) void scan_kernel(const int* __restrict__ in, int* __restrict__ out, int n) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    const int v = i < n ? in[i] : 0;
    int x = v;   // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int c = carry;
    if (i < n) out[i] = c + woff + x - v;
    __syncthreads();
    if (tid == 1023) carry = c + woff + x;
    __syncthreads();
  }
  if (tid == 0) out[n] = carry;
}

__global__ __launch_bounds__(256) void csr_fill_kernel(const int64_t* __restrict__ f0, const int64_t* __restrict__ f1, int F,
                                                       const int* __restrict__ rowptr, int* __restrict__ cursor,
                                                       int* __restrict__ nbr) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= F) return;
  const int a = (int)f0[e], b = (int)f1[e];
  nbr[rowptr[a] + atomicAdd(&cursor[a], 1)] = b;
  nbr[rowptr[b] + atomicAdd(&cursor[b], 1)] = a;
}

// FILL = false: counts[i] = number of nodes j > i within k hops of i;  FILL = true: writes them, ascending, at offs[i]
template <bool FILL>
__global__ __launch_bounds__(128) void khop_kernel(const int* __restrict__ rowptr, const int* __restrict__ nbr, int N, int k,
                                                   int* __restrict__ counts, const int* __restrict__ offs,
                                                   int64_t* __restrict__ out0, int64_t* __restrict__ out1,
                                                   int* __restrict__ flag) {
  const int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= N) return;
  int buf[KHOP_CAP];
  int n = 1, lvl_beg = 0;
  buf[0] = i;
  bool over = false;
  for (int hop = 0; hop < k && !over; ++hop) {
    const int lvl_end = n;
    for (int q = lvl_beg; q < lvl_end && !over; ++q) {
      const int u = buf[q];
      for (int p = rowptr[u]; p < rowptr[u + 1]; ++p) {
        const int v = nbr[p];
        bool seen = false;
        for (int s = 0; s < n; ++s) seen = seen || (buf[s] == v);
        if (!seen) {
          if (n == KHOP_CAP) { over = true; break; }
          buf[n++] = v;
        }
      }
    }
    lvl_beg = lvl_end;
  }
  if (over) {
    atomicOr(flag, 1);
    if (!FILL) counts[i] = 0;
    return;
  }
  // the partners with a larger index, in place at the front of the list
  int m = 0;
  for (int s = 1; s < n; ++s)
    if (buf[s] > i) buf[m++] = buf[s];
  if (!FILL) {
    counts[i] = m;
    return;
  }
  for (int a = 1; a < m; ++a) {   // insertion sort (m ~ 10)
    const int v = buf[a];
    int b = a - 1;
    while (b >= 0 && buf[b] > v) { buf[b + 1] = buf[b]; --b; }
    buf[b + 1] = v;
  }
  const int o = offs[i];
  for (int s = 0; s < m; ++s) {
    out0[o + s] = i;
    out1[o + s] = buf[s];
  }
}

}  // namespace

extern "C" size_t gfv_khop_workspace_ints(int32_t N, int32_t F) { return (size_t)4 * ((size_t)N + 1) + 2 * (size_t)F + 8; }

extern "C" int gfv_khop_count(const int64_t* face0, const int64_t* face1, int32_t F, int32_t N, int32_t k, int32_t* ws,
                              void* stream_) {
  if (N < 1 || F < 0 || k < 1 || !ws) return GFV_ERR_ARG;
  // per-mesh preprocessing: its fills below are plain hipMemsetAsync calls a replay would not repeat - not part of a recorded step
  if (gfv_rec_active()) return GFV_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  int* deg = ws;                       // [N + 1] degree, then the CSR cursor
  int* rowptr = deg + (N + 1);         // [N + 1]
  int* counts = rowptr + (N + 1);      // [N + 1]
  int* offs = counts + (N + 1);        // [N + 1] exclusive scan of counts; offs[N] = number of pairs
  int* flag = offs + (N + 1);          // [8]
  int* nbr = flag + 8;                 // [2 F]
  if (hipMemsetAsync(deg, 0, sizeof(int) * (size_t)(N + 1), stream) != hipSuccess) return GFV_ERR_LAUNCH;
  if (hipMemsetAsync(flag, 0, sizeof(int) * 8, stream) != hipSuccess) return GFV_ERR_LAUNCH;
  if (F > 0) GFV_LAUNCH(degree_kernel, dim3(gfv_div_up(F, 256)), dim3(256), 0, stream, face0, face1, F, deg);
  GFV_LAUNCH(scan_kernel, dim3(1), dim3(1024), 0, stream, deg, rowptr, N);
  if (hipMemsetAsync(deg, 0, sizeof(int) * (size_t)(N + 1), stream) != hipSuccess) return GFV_ERR_LAUNCH;
  if (F > 0) GFV_LAUNCH(csr_fill_kernel, dim3(gfv_div_up(F, 256)), dim3(256), 0, stream, face0, face1, F, rowptr, deg, nbr);
  GFV_LAUNCH(khop_kernel<false>, dim3(gfv_div_up(N, 128)), dim3(128), 0, stream, rowptr, nbr, N, k, counts, offs,
                     (int64_t*)nullptr, (int64_t*)nullptr, flag);
  GFV_LAUNCH(scan_kernel, dim3(1), dim3(1024), 0, stream, counts, offs, N);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_khop_fill(int32_t N, int32_t k, const int32_t* ws, int64_t* out0, int64_t* out1, void* stream_) {
  if (N < 1 || k < 1 || !ws || !out0 || !out1) return GFV_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const int* rowptr = ws + (N + 1);
  const int* offs = ws + 3 * (size_t)(N + 1);
  int* flag = const_cast<int*>(ws) + 4 * (size_t)(N + 1);
  const int* nbr = flag + 8;
  GFV_LAUNCH(khop_kernel<true>, dim3(gfv_div_up(N, 128)), dim3(128), 0, stream, rowptr, nbr, N, k, (int*)nullptr, offs,
                     out0, out1, flag);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}
