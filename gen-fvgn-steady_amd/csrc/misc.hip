// Input normalisation, relative edge features and the fused Adam step (gfx950).  Contract: include/gfv.h.
// Reference: FVMmodel/importer.py:54-93,114-130,166-178; utils/normalization.py:32-85; torch.optim.Adam defaults
// (pre_train_Adam.py:79,191).  All HBM-bound elementwise / small-reduction work.
#include "gfv_common.h"
#include "gfv_prof.h"
#include "../../include/gfv.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
  const int tid = threadIdx.x;
  v = gfv_wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}

// per graph: mean and population std of x[:,0:3] (importer.py:80-93), two passes like the reference
__global__ __launch_bounds__(1024) void graph_norm_stats_kernel(const float* __restrict__ x, int ldx,
                                                               const int* __restrict__ gnode_ptr, float* __restrict__ stats) {
  __shared__ float red[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int beg = gnode_ptr[b], end = gnode_ptr[b + 1];
  const float cnt = fmaxf((float)(end - beg), 1.f);
  // one sweep per statistic for the three columns together (per column the same summation order as a sweep of its own)
  float mean[3];
  {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    // four rows in flight per thread (the loop was one memory round trip per row: 25 dependent trips per pass at 25 k nodes,
    // 23 us at the very start of the step), added in row order
    int i = beg + tid;
    for (; i + 3 * 1024 < end; i += 4 * 1024) {
      float v[4][3];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* r = x + (size_t)(i + 1024 * u) * ldx;
        v[u][0] = r[0]; v[u][1] = r[1]; v[u][2] = r[2];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { s0 += v[u][0]; s1 += v[u][1]; s2 += v[u][2]; }
    }
    for (; i < end; i += 1024) {
      const float* r = x + (size_t)i * ldx;
      s0 += r[0]; s1 += r[1]; s2 += r[2];
    }
    mean[0] = block_sum(s0, red) / cnt;
    mean[1] = block_sum(s1, red) / cnt;
    mean[2] = block_sum(s2, red) / cnt;
  }
  float q0 = 0.f, q1 = 0.f, q2 = 0.f;
  int i = beg + tid;
  for (; i + 3 * 1024 < end; i += 4 * 1024) {
    float v[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* r = x + (size_t)(i + 1024 * u) * ldx;
      v[u][0] = r[0]; v[u][1] = r[1]; v[u][2] = r[2];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float d0 = v[u][0] - mean[0], d1 = v[u][1] - mean[1], d2 = v[u][2] - mean[2];
      q0 += d0 * d0; q1 += d1 * d1; q2 += d2 * d2;
    }
  }
  for (; i < end; i += 1024) {
    const float* r = x + (size_t)i * ldx;
    const float d0 = r[0] - mean[0], d1 = r[1] - mean[1], d2 = r[2] - mean[2];
    q0 += d0 * d0; q1 += d1 * d1; q2 += d2 * d2;
  }
  const float var[3] = {block_sum(q0, red) / cnt, block_sum(q1, red) / cnt, block_sum(q2, red) / cnt};
  if (tid == 0) {
    for (int c = 0; c < 3; ++c) {
      stats[6 * b + c] = mean[c];
      stats[6 * b + 3 + c] = sqrtf(var[c]);
    }
  }
}

// The same statistics over many workgroups (one workgroup reads a 25 k-node graph at the bandwidth of ONE CU: 2 x 8 us): each
// of NB blocks per graph sums x and x^2 of its share of the rows in double, a second small launch folds the NB partials in
// a fixed order.  mean and E[x^2] - mean^2 in double are the exact statistics of the fp32 data (the reference's two fp32
// sweeps are one rounding error of that away).
constexpr int NORM_NB = 64;
__global__ __launch_bounds__(256) void graph_norm_partial_kernel(const float* __restrict__ x, int ldx, const int* __restrict__ gnode_ptr,
                                                                 double* __restrict__ ws) {
  __shared__ double red[4][6];
  const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  const int beg = gnode_ptr[b], end = gnode_ptr[b + 1];
  double s[3] = {0.0, 0.0, 0.0}, q[3] = {0.0, 0.0, 0.0};
  for (int i = beg + blk * 256 + tid; i < end; i += NORM_NB * 256) {
    const float* r = x + (size_t)i * ldx;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double v = (double)r[c];
      s[c] += v;
      q[c] += v * v;
    }
  }
  // fixed-order fold: lanes of a wave (xor butterflies are order-symmetric), then the four waves
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      s[c] += __shfl_xor(s[c], o);
      q[c] += __shfl_xor(q[c], o);
    }
  }
  if ((tid & 63) == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { red[tid >> 6][c] = s[c]; red[tid >> 6][3 + c] = q[c]; }
  }
  __syncthreads();
  if (tid < 6) ws[((size_t)b * NORM_NB + blk) * 6 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}
__global__ __launch_bounds__(64) void graph_norm_final_kernel(const double* __restrict__ ws, const int* __restrict__ gnode_ptr,
                                                              float* __restrict__ stats) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid >= 3) return;
  const double cnt = fmax((double)(gnode_ptr[b + 1] - gnode_ptr[b]), 1.0);
  double s = 0.0, q = 0.0;
  for (int k = 0; k < NORM_NB; ++k) {
    s += ws[((size_t)b * NORM_NB + k) * 6 + tid];
    q += ws[((size_t)b * NORM_NB + k) * 6 + 3 + tid];
  }
  const double mean = s / cnt, var = fmax(q / cnt - mean * mean, 0.0);
  stats[6 * b + tid] = (float)mean;
  stats[6 * b + 3 + tid] = (float)sqrt(var);
}

// column sums / sums of squares of x[:, 3:12] -> per-block partials [nblocks][18]
__global__ __launch_bounds__(256) void normalizer_partial_kernel(const float* __restrict__ x, int ldx, int N,
                                                                 float* __restrict__ partial) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  float s[9], q[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) { s[c] = 0.f; q[c] = 0.f; }
  for (int i = blockIdx.x * 256 + tid; i < N; i += gridDim.x * 256) {
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const float v = x[(size_t)i * ldx + 3 + c];
      s[c] += v; q[c] += v * v;
    }
  }
  for (int c = 0; c < 9; ++c) {
    const float a = block_sum(s[c], red);
    const float b = block_sum(q[c], red);
    if (tid == 0) { partial[blockIdx.x * 18 + c] = a; partial[blockIdx.x * 18 + 9 + c] = b; }
  }
}

// fold partials into the running buffers and derive mean / std (normalization.py:52-85)
__global__ void normalizer_finalize_kernel(const float* __restrict__ partial, int nblocks, float n_rows, int accumulate,
                                           float* acc_count, float* num_acc, float* acc_sum, float* acc_sq,
                                           float* __restrict__ mean_std) {
  const int c = threadIdx.x;
  if (c >= 9) return;
  if (accumulate) {
    float a = 0.f, b = 0.f;
    for (int i = 0; i < nblocks; ++i) { a += partial[i * 18 + c]; b += partial[i * 18 + 9 + c]; }
    acc_sum[c] += a;
    acc_sq[c] += b;
  }
  const float cnt_new = accumulate ? (*acc_count + n_rows) : *acc_count;
  const float safe = fmaxf(cnt_new, 1.0f);
  const float mean = acc_sum[c] / safe;
  float sd = sqrtf(acc_sq[c] / safe - mean * mean);
  if (sd < 1e-8f || !(sd == sd)) sd = (sd == sd) ? 1.0f : sd;
  mean_std[c] = mean;
  mean_std[9 + c] = sd;
  __syncthreads();
  if (c == 0 && accumulate) { *acc_count = cnt_new; *num_acc = *num_acc + 1.0f; }
}

// uv_old, per-graph standardisation of x[:,0:3], running-stat normalisation of x[:,3:12]; in place on x
__global__ __launch_bounds__(256) void node_prep_kernel(float* __restrict__ x, int ldx, const int* __restrict__ batch,
                                                        const float* __restrict__ stats, const float* __restrict__ uvp_dim,
                                                        const float* __restrict__ mean_std, int norm_global,
                                                        float* __restrict__ uv_old, int N) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int b = batch[i];
  float* xr = x + (size_t)i * ldx;
  uv_old[2 * i] = xr[0] / uvp_dim[3 * b];
  uv_old[2 * i + 1] = xr[1] / uvp_dim[3 * b + 1];
#pragma unroll
  for (int c = 0; c < 3; ++c) xr[c] = (xr[c] - stats[6 * b + c]) / (stats[6 * b + 3 + c] + 1e-8f);
  if (norm_global) {
#pragma unroll
    for (int c = 0; c < 9; ++c) xr[3 + c] = (xr[3 + c] - mean_std[c]) / mean_std[9 + c];
  }
}

// edge_attr = [x_s - x_r (12) | pos_s - pos_r (2) | norm (1)] (importer.py:54-78); padded [E,16] + optional packed [E,15]
__global__ __launch_bounds__(256) void edge_attr_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ pos,
                                                        const int* __restrict__ es, const int* __restrict__ er,
                                                        float* __restrict__ out16, float* __restrict__ out15, int E) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int s = es[e], r = er[e];
  float v[16];
#pragma unroll
  for (int c = 0; c < 12; ++c) v[c] = x[(size_t)s * ldx + c] - x[(size_t)r * ldx + c];
  const float dx = pos[2 * s] - pos[2 * r], dy = pos[2 * s + 1] - pos[2 * r + 1];
  v[12] = dx; v[13] = dy; v[14] = sqrtf(dx * dx + dy * dy); v[15] = 0.f;
  float4* o = reinterpret_cast<float4*>(out16 + (size_t)e * 16);
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
  if (out15) {
#pragma unroll
    for (int c = 0; c < 15; ++c) out15[(size_t)e * 15 + c] = v[c];
  }
}

// ---- round 6: the input preparation as TWO launches (was: state restore copy, graph_norm_partial, graph_norm_final,
// normalizer_finalize, node_prep, edge_attr - six launches of 3 - 6 us each at the very start of the step) -----------------
//
// prep_stats_kernel: the per-graph statistics exactly as graph_norm_partial + graph_norm_final form them (NORM_NB workgroups per
// graph, double partial sums, folded k = 0 .. NORM_NB-1 in order) - the fold is done by the workgroup of the graph that arrives
// LAST (integer counter per graph; the order of the fold does not depend on who does it), and - `mean_std` given - that
// workgroup of graph 0 also derives the Normalizer's (mean, std) from its running buffers (normalizer_finalize without
// accumulation: an accumulating step updates the buffers first, gfv_normalizer_update).
// With `x_raw` the launch also leaves a copy of the un-normalised rows (the drop-in path normalises graph_node.x IN PLACE,
// importer.py:123-130, and the edge features need both end nodes' normalised rows: the second launch reads the raw copy).
__global__ __launch_bounds__(256) void prep_stats_kernel(const float* __restrict__ x, int ldx, const int* __restrict__ gnode_ptr,
                                                         double* __restrict__ ws, int* __restrict__ counters,
                                                         float* __restrict__ stats, float* __restrict__ x_raw,
                                                         const float* acc_count, const float* acc_sum, const float* acc_sq,
                                                         float* __restrict__ mean_std) {
  __shared__ double red[4][6];
  __shared__ int last;
  const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  const int beg = gnode_ptr[b], end = gnode_ptr[b + 1];
  double s[3] = {0.0, 0.0, 0.0}, q[3] = {0.0, 0.0, 0.0};
  for (int i = beg + blk * 256 + tid; i < end; i += NORM_NB * 256) {
    const float* r = x + (size_t)i * ldx;
    if (x_raw) {
      float* o = x_raw + (size_t)i * 12;
#pragma unroll
      for (int c = 0; c < 12; ++c) o[c] = r[c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double v = (double)r[c];
      s[c] += v;
      q[c] += v * v;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      s[c] += __shfl_xor(s[c], o);
      q[c] += __shfl_xor(q[c], o);
    }
  }
  if ((tid & 63) == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { red[tid >> 6][c] = s[c]; red[tid >> 6][3 + c] = q[c]; }
  }
  __syncthreads();
  if (tid < 6) ws[((size_t)b * NORM_NB + blk) * 6 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    last = atomicAdd(counters + b, 1) == NORM_NB - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  // the NORM_NB x 6 partial sums: ONE round trip (a thread each) into LDS, then folded k = 0 .. NORM_NB - 1 in order as
  // graph_norm_final_kernel does (its 2 x 64 dependent loads per thread were most of that launch's 6 us)
  __shared__ double fold[NORM_NB * 6];
  for (int i = tid; i < NORM_NB * 6; i += 256)
    fold[i] = __hip_atomic_load(ws + (size_t)b * NORM_NB * 6 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (tid < 3) {
    const double cnt = fmax((double)(end - beg), 1.0);
    double ss = 0.0, qq = 0.0;
    for (int k = 0; k < NORM_NB; ++k) {
      ss += fold[k * 6 + tid];
      qq += fold[k * 6 + 3 + tid];
    }
    const double mean = ss / cnt, var = fmax(qq / cnt - mean * mean, 0.0);
    stats[6 * b + tid] = (float)mean;
    stats[6 * b + 3 + tid] = (float)sqrt(var);
  }
  if (tid == 0) counters[b] = 0;   // (reusable by the next launch)
  if (b == 0 && mean_std && tid >= 64 && tid < 64 + 9) {
    // normalizer_finalize_kernel with accumulate = 0 (normalization.py:72-85)
    const int c = tid - 64;
    const float safe = fmaxf(*acc_count, 1.0f);
    const float mean = acc_sum[c] / safe;
    float sd = sqrtf(acc_sq[c] / safe - mean * mean);
    if (sd < 1e-8f || !(sd == sd)) sd = (sd == sd) ? 1.0f : sd;
    mean_std[c] = mean;
    mean_std[9 + c] = sd;
  }
}

// prep_apply_kernel: node_prep + edge_attr in one launch.  Threads [0, N) normalise node i (raw row in, normalised row out,
// uv_old), threads [N, N + E) form edge e's relative features from the RAW rows of its two end nodes, normalising them on the
// way with the very expressions node_prep uses - the same fp32 operations on the same inputs, so the differences equal those
// of the stored normalised rows bit for bit.  x_raw != x_out (the raw rows must survive the launch: TrainStep's persistent
// backup, or the copy prep_stats left).
__device__ __forceinline__ void prep_norm_row(const float* __restrict__ xr, int b, const float* __restrict__ stats,
                                              const float* __restrict__ mean_std, int norm_global, float (&v)[12]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) v[c] = (xr[c] - stats[6 * b + c]) / (stats[6 * b + 3 + c] + 1e-8f);
#pragma unroll
  for (int c = 0; c < 9; ++c) v[3 + c] = norm_global ? (xr[3 + c] - mean_std[c]) / mean_std[9 + c] : xr[3 + c];
}
__global__ __launch_bounds__(256) void prep_apply_kernel(const float* __restrict__ x_raw, float* __restrict__ x_out,
                                                         const int* __restrict__ batch, const float* __restrict__ stats,
                                                         const float* __restrict__ uvp_dim, const float* __restrict__ mean_std,
                                                         int norm_global, float* __restrict__ uv_old, int N,
                                                         const float* __restrict__ pos, const int* __restrict__ es,
                                                         const int* __restrict__ er, float* __restrict__ out16,
                                                         float* __restrict__ out15, int E) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < N) {
    const int b = batch[t];
    const float* xr = x_raw + (size_t)t * 12;
    uv_old[2 * t] = xr[0] / uvp_dim[3 * b];
    uv_old[2 * t + 1] = xr[1] / uvp_dim[3 * b + 1];
    float v[12];
    prep_norm_row(xr, b, stats, mean_std, norm_global, v);
    float4* o = reinterpret_cast<float4*>(x_out + (size_t)t * 12);
#pragma unroll
    for (int j = 0; j < 3; ++j) o[j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
    return;
  }
  const int e = t - N;
  if (e >= E) return;
  const int s = es[e], r = er[e];
  float a[12], c2[12], v[16];
  prep_norm_row(x_raw + (size_t)s * 12, batch[s], stats, mean_std, norm_global, a);
  prep_norm_row(x_raw + (size_t)r * 12, batch[r], stats, mean_std, norm_global, c2);
#pragma unroll
  for (int c = 0; c < 12; ++c) v[c] = a[c] - c2[c];
  const float dx = pos[2 * s] - pos[2 * r], dy = pos[2 * s + 1] - pos[2 * r + 1];
  v[12] = dx; v[13] = dy; v[14] = sqrtf(dx * dx + dy * dy); v[15] = 0.f;
  float4* o = reinterpret_cast<float4*>(out16 + (size_t)e * 16);
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
  if (out15) {
#pragma unroll
    for (int c = 0; c < 15; ++c) out15[(size_t)e * 15 + c] = v[c];
  }
}

// Adam (torch.optim.Adam defaults, pre_train_Adam.py:115): the step counter and the hyper-parameters live in device
// memory, so a captured hipGraph follows lr changes (lr_scheduler.step() every epoch in both reference drivers).
// state[16] (include/gfv.h): t = completed steps; the bias corrections OF THE NEXT STEP - 1 - b1^(t+1) as a (hi, lo) float pair,
// sqrt(1 - b2^(t+1)) -; 1 - b1 and 1 - b2 rounded from DOUBLE as torch's host code forms them (1 - 0.999 in double is 0.001;
// 1.0f - 0.999f is 1.3e-5 away - the second moments of rounds 1 - 5 differed from torch's by that factor); the running powers
// b^(t+1) and the betas themselves as (hi, lo) pairs (48 bits: the powers advance by one double multiplication per step, no pow).
// Every thread forms lr / bc1 in double with the CURRENT lr (a step follows `ts.lr = ...`).
// Round 6: ONE launch per step.  The tick used to be a launch of its own (4.8 us of launch floor for two pows of one thread;
// folded into every workgroup it cost 53 us): now the workgroup that FINISHES LAST - every other one has read the state by then -
// advances t, forms the next step's corrections and publishes the status word (include/gfv.h gfv_status_mirror).
// hyper = {lr, beta1, beta2, eps, grad_scale, w_cont, w_mom, w_press}
enum { AS_T = 0, AS_BC1_HI = 1, AS_BC1_LO = 2, AS_SQRT_BC2 = 3, AS_COUNTER = 4, AS_OMB1 = 5, AS_OMB2 = 6, AS_P1 = 8, AS_P2 = 10,
       AS_B1 = 12, AS_B2 = 14 };
__device__ __forceinline__ void as_put(float* state, int at, double v) {
  const float hi = (float)v;
  state[at] = hi;
  state[at + 1] = (float)(v - (double)hi);
}
__device__ __forceinline__ double as_get(const float* state, int at) { return (double)state[at] + (double)state[at + 1]; }
__device__ __forceinline__ void adam_corrections(float* state, double p1, double p2) {   // p = beta^(t+1)
  as_put(state, AS_P1, p1);
  as_put(state, AS_P2, p2);
  as_put(state, AS_BC1_HI, 1.0 - p1);
  state[AS_SQRT_BC2] = (float)sqrt(1.0 - p2);
}
__global__ void adam_state_init_kernel(float* state, float t_done, double b1, double b2) {
  for (int i = 0; i < 16; ++i) state[i] = 0.f;
  state[AS_T] = t_done;
  reinterpret_cast<int*>(state)[AS_COUNTER] = 0;
  state[AS_OMB1] = (float)(1.0 - b1);
  state[AS_OMB2] = (float)(1.0 - b2);
  as_put(state, AS_B1, b1);
  as_put(state, AS_B2, b2);
  const double tn = (double)t_done + 1.0;
  adam_corrections(state, pow(b1, tn), pow(b2, tn));
}

constexpr int ADAM_TPB = 512, ADAM_MAX_WGS = 512;   // (few workgroups: one same-address atomic each at the end)
__global__ __launch_bounds__(ADAM_TPB) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long n, float* state,
                                                        const float* __restrict__ hyper, const int* status_dev, int* status_host) {
  const float step_size = (float)((double)hyper[0] / as_get(state, AS_BC1_HI)), bc2_sqrt = state[AS_SQRT_BC2];
  const float t_done = state[AS_T];
  const float b1 = hyper[1], b2 = hyper[2], eps = hyper[3], grad_scale = hyper[4];
  const float omb1 = state[AS_OMB1], omb2 = state[AS_OMB2];
  for (long i = (long)blockIdx.x * ADAM_TPB + threadIdx.x; i < n; i += (long)gridDim.x * ADAM_TPB) {
    const float gi = g[i] * grad_scale;
    const float mi = m[i] * b1 + omb1 * gi;
    const float vi = v[i] * b2 + omb2 * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
  __syncthreads();   // every thread of this workgroup has read the state (the values were consumed by the loop above)
  if (threadIdx.x == 0) {
    int* counter = reinterpret_cast<int*>(state) + AS_COUNTER;
    // RELAXED, no fence: the last arriver needs no DATA of the others, only the fact that they are past their reads of `state`.
    // (A release fence here writes back the L2's dirty lines - the 14 MB this launch has just written - once per workgroup: the
    // first form of this kernel took 115 us instead of 8.)
    if (__hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
      *counter = 0;
      state[AS_T] = t_done + 1.0f;
      adam_corrections(state, as_get(state, AS_P1) * as_get(state, AS_B1), as_get(state, AS_P2) * as_get(state, AS_B2));
      if (status_host) {
        const int f = *reinterpret_cast<const volatile int*>(status_dev);
        if (f) __hip_atomic_store(status_host, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// loss = mean_b log(w_p*L_p + w_c*L_c + w_m*L_mx + w_m*L_my) (pre_train_Adam.py:177-184) and its gradient wrt the
// four per-graph residuals; losses layout [B,4] = (cont, momx, momy, press)
__global__ void train_loss_kernel(const float* __restrict__ losses, int B, float wc, float wm, float wp,
                                  const float* __restrict__ hyper, float* __restrict__ loss, float* __restrict__ gloss) {
  __shared__ float red[64];
  if (hyper) { wc = hyper[5]; wm = hyper[6]; wp = hyper[7]; }   // device-resident weights (captured steps follow them)
  const int tid = threadIdx.x;
  float s = 0.f;
  for (int b = tid; b < B; b += 64) {
    const float tot = wp * losses[4 * b + 3] + wc * losses[4 * b] + wm * losses[4 * b + 1] + wm * losses[4 * b + 2];
    s += logf(tot);
    const float inv = 1.0f / (tot * (float)B);
    gloss[4 * b] = wc * inv; gloss[4 * b + 1] = wm * inv; gloss[4 * b + 2] = wm * inv; gloss[4 * b + 3] = wp * inv;
  }
  red[tid] = s;
  __syncthreads();
  if (tid == 0) {
    float a = 0.f;
    for (int i = 0; i < 64; ++i) a += red[i];
    *loss = a / (float)B;
  }
}

inline int cap_grid(long n) {
  long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int gfv_graph_norm_stats(const float* x, int32_t ldx, const int32_t* gnode_ptr, int32_t B, float* stats,
                                    void* stream) {
  GfvProfScope ps_(GFV_K_MISC, 0, 12.0 * 0.0, stream);
  if (B <= 0) return GFV_OK;
  GFV_LAUNCH(graph_norm_stats_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, x, ldx, gnode_ptr, stats);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" size_t gfv_graph_norm_workspace_bytes(int32_t B) { return (size_t)(B > 0 ? B : 0) * NORM_NB * 6 * sizeof(double); }
extern "C" int gfv_graph_norm_stats_ws(const float* x, int32_t ldx, const int32_t* gnode_ptr, int32_t B, float* stats,
                                       void* workspace, void* stream) {
  GfvProfScope ps_(GFV_K_MISC, 0, 12.0 * 0.0, stream);
  if (B <= 0) return GFV_OK;
  if (!workspace || (reinterpret_cast<size_t>(workspace) & 7)) return GFV_ERR_ARG;
  GFV_LAUNCH(graph_norm_partial_kernel, dim3(NORM_NB, B), dim3(256), 0, (hipStream_t)stream, x, ldx, gnode_ptr,
                     reinterpret_cast<double*>(workspace));
  GFV_LAUNCH(graph_norm_final_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<const double*>(workspace),
                     gnode_ptr, stats);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_normalizer_blocks(int32_t N) { int g = (N + 255) / 256; return g > 512 ? 512 : (g < 1 ? 1 : g); }

extern "C" int gfv_normalizer_update(const float* x, int32_t ldx, int32_t N, int32_t accumulate, float* acc_count,
                                     float* num_acc, float* acc_sum, float* acc_sq, float* partial_ws, float* mean_std,
                                     void* stream) {
  GfvProfScope ps_(GFV_K_MISC, 0, accumulate ? 36.0 * N : 0.0, stream);
  const int nb = gfv_normalizer_blocks(N);
  if (accumulate && N > 0) {
    GFV_LAUNCH(normalizer_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, ldx, N, partial_ws);
    GFV_CHECK_LAUNCH();
  }
  GFV_LAUNCH(normalizer_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial_ws, nb, (float)N,
                     accumulate, acc_count, num_acc, acc_sum, acc_sq, mean_std);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_node_prep(float* x, int32_t ldx, const int32_t* batch, const float* stats, const float* uvp_dim,
                             const float* mean_std, int32_t norm_global, float* uv_old, int32_t N, void* stream) {
  GfvProfScope ps_(GFV_K_MISC, 0, (96.0 + 4.0 + 8.0) * N, stream);
  if (N <= 0) return GFV_OK;
  GFV_LAUNCH(node_prep_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, ldx, batch, stats,
                     uvp_dim, mean_std, norm_global, uv_old, N);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_edge_attr(const float* x, int32_t ldx, const float* pos, const int32_t* es, const int32_t* er,
                             float* out16, float* out15, int32_t E, void* stream) {
  GfvProfScope ps_(GFV_K_MISC, 0, (8.0 + 64.0 + (out15 ? 60.0 : 0.0)) * E, stream);
  if (E <= 0) return GFV_OK;
  GFV_LAUNCH(edge_attr_kernel, dim3((E + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, ldx, pos, es, er,
                     out16, out15, E);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

// workspace = [arrival counters: PREP_MAX_GRAPHS int32, a FIXED header - a caller may reuse one workspace for batches of different
// sizes, and a counter must be where the last launch left its zero][per-graph partial sums: B x NORM_NB x 6 doubles]
constexpr int PREP_MAX_GRAPHS = 1024;
extern "C" size_t gfv_prep_workspace_bytes(int32_t B) {
  return PREP_MAX_GRAPHS * sizeof(int32_t) + (size_t)(B > 0 ? B : 0) * NORM_NB * 6 * sizeof(double);
}
extern "C" int gfv_prep_stats(const float* x, int32_t ldx, const int32_t* gnode_ptr, int32_t B, float* stats, void* workspace,
                              float* x_raw, const float* acc_count, const float* acc_sum, const float* acc_sq, float* mean_std,
                              void* stream) {
  GfvProfScope ps_(GFV_K_MISC, 0, 0.0, stream);
  if (B <= 0) return GFV_OK;
  if (!x || !gnode_ptr || !stats || !workspace || (reinterpret_cast<size_t>(workspace) & 7) || ldx < 12 || B > PREP_MAX_GRAPHS)
    return GFV_ERR_ARG;
  if (mean_std && (!acc_count || !acc_sum || !acc_sq)) return GFV_ERR_ARG;
  int* counters = reinterpret_cast<int*>(workspace);
  double* ws = reinterpret_cast<double*>(counters + PREP_MAX_GRAPHS);
  GFV_LAUNCH(prep_stats_kernel, dim3(NORM_NB, B), dim3(256), 0, (hipStream_t)stream, x, ldx, gnode_ptr, ws, counters, stats, x_raw,
             acc_count, acc_sum, acc_sq, mean_std);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}
extern "C" int gfv_prep_apply(const float* x_raw, float* x_out, const int32_t* batch, const float* stats, const float* uvp_dim,
                              const float* mean_std, int32_t norm_global, float* uv_old, int32_t N, const float* pos,
                              const int32_t* es, const int32_t* er, float* out16, float* out15, int32_t E, void* stream) {
  GfvProfScope ps_(GFV_K_MISC, 0, (96.0 + 4.0 + 8.0) * N + (8.0 + 64.0 + (out15 ? 60.0 : 0.0)) * E, stream);
  if (N <= 0) return GFV_OK;
  if (!x_raw || !x_out || x_raw == x_out || (reinterpret_cast<size_t>(x_out) & 15) || !batch || !stats || !uvp_dim || !uv_old ||
      (norm_global && !mean_std) || (E > 0 && (!pos || !es || !er || !out16)))
    return GFV_ERR_ARG;
  const long total = (long)N + (E > 0 ? E : 0);
  GFV_LAUNCH(prep_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_raw, x_out, batch, stats,
             uvp_dim, mean_std, norm_global, uv_old, N, pos, es, er, out16, out15, E > 0 ? E : 0);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

int* gfv_internal_status_ptr();        // dw.hip
int32_t* gfv_internal_status_mirror();   // dw.hip: nullptr until a host asked for the mirror

extern "C" int gfv_adam_state_init(float* state, double beta1, double beta2, float steps_done, void* stream) {
  if (!state || !(steps_done >= 0.f) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return GFV_ERR_ARG;
  GFV_LAUNCH(adam_state_init_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, steps_done, beta1, beta2);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float* state, const float* hyper,
                                 void* stream) {
  GfvProfScope ps_(GFV_K_MISC, 0, 28.0 * (double)n, stream);   // p, g, m, v in; p, m, v out
  if (n <= 0) return GFV_OK;
  if (!state || !hyper) return GFV_ERR_ARG;
  int32_t* mirror = gfv_internal_status_mirror();
  long wgs = (n + ADAM_TPB - 1) / ADAM_TPB;
  if (wgs > ADAM_MAX_WGS) wgs = ADAM_MAX_WGS;
  GFV_LAUNCH(adam_kernel, dim3((unsigned)wgs), dim3(ADAM_TPB), 0, (hipStream_t)stream, p, g, m, v, (long)n, state, hyper,
             (const int*)gfv_internal_status_ptr(), (int*)mirror);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_train_loss(const float* losses, int32_t B, float w_cont, float w_mom, float w_press, float* loss,
                              float* gloss, void* stream) {
  if (B <= 0) return GFV_ERR_ARG;
  GFV_LAUNCH(train_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, losses, B, w_cont, w_mom, w_press,
                     (const float*)nullptr, loss, gloss);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_train_loss_dev(const float* losses, int32_t B, const float* hyper, float* loss, float* gloss, void* stream) {
  GfvProfScope ps_(GFV_K_MISC, 0, 32.0 * B, stream);
  if (B <= 0 || !hyper) return GFV_ERR_ARG;
  GFV_LAUNCH(train_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, losses, B, 0.f, 0.f, 0.f, hyper, loss, gloss);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

// ---- batch assembly for the device-resident state pool ---------------------------------------
// One launch copies every per-mesh piece of every plan / field tensor of a batch to its place in the batched tensor,
// adding the index offset of the mesh (node / face / cell / incidence offsets of the block-diagonal batch,
// Graph_loader.py:405-480 __inc__ rules); kind 2 fills a constant (graph id).  32-bit words throughout.
namespace {
__global__ __launch_bounds__(256) void concat_offsets_kernel(const gfv_concat_desc_t* __restrict__ descs) {
  const gfv_concat_desc_t d = descs[blockIdx.y];
  const long n = d.n_words;
  const int* __restrict__ src = reinterpret_cast<const int*>(d.src);
  int* __restrict__ dst = reinterpret_cast<int*>(d.dst);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    int v;
    if (d.kind == 2) v = d.add;
    else {
      v = src[i];
      if (d.kind == 1) v += d.add;
    }
    dst[i] = v;
  }
}
}  // namespace

extern "C" int gfv_concat_offsets(const gfv_concat_desc_t* descs, int32_t n_desc, int32_t blocks_per_desc, void* stream) {
  if (n_desc <= 0) return GFV_OK;
  if (blocks_per_desc < 1) blocks_per_desc = 1;
  GFV_LAUNCH(concat_offsets_kernel, dim3(blocks_per_desc, n_desc), dim3(256), 0, (hipStream_t)stream, descs);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}
