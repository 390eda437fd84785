// Atomics-free segmented (CSR) gather-reduce and its companions (gfx950, wave64).  Contract: include/gfv.h.
//
// out[r,:] = scale[r] * sum_{k in row r} src[col[k],:]
//
// A destination row is owned by a sub-wave of LPR = F/4 lanes (each lane one float4 = 16 B, so a row is read in
// full 64..512-byte segments); a wave carries 64/LPR rows at a time and every (row, lane) walks its CSR
// segment with 4 gathers in flight.  No atomics, fixed summation order (deterministic), one coalesced write per
// destination row.  Rows are node/cell features that stay L2/MALL resident between kernels; indices are int32.
#include "gfv_common.h"
#include "gfv_prof.h"
#include "../../include/gfv.h"

namespace {

// LN (round 6; LPR = 16 only): the source rows are the 64-column HALVES of 128-wide pre-LayerNorm rows y (half-row c = row c >> 1,
// columns 64 (c & 1) ..), and what is summed is LayerNorm(y) = (y - mean) * rstd * gamma + beta with the row statistics the
// producing chain launch left behind (gfv_rowtile_args_t.fin_stats) - the expression, and so every bit, of that launch's own
// LayerNorm output.  The EdgeBlock forward then need not write its output a second time without the residual (blocks.py:35-42
// aggregates the MLP's output, the block returns e + output): 512 B per edge row less on a launch that runs at the rate of
// its bytes.
struct SegLn {
  const float* stats;   // [rows of y][2] = (mean, 1 / std)
  const float* gamma;   // [128]
  const float* beta;    // [128]
};

template <int LPR, bool LN = false>
__global__ __launch_bounds__(256) void seg_gather_sum_vec(const float* __restrict__ src, const int* __restrict__ rowptr,
                                                          const int* __restrict__ col, const float* __restrict__ scale,
                                                          const float* __restrict__ src_scale,
                                                          float* __restrict__ out, int n_rows, int accumulate, const SegLn ln) {
  static_assert(!LN || LPR == 16, "LayerNorm on load: halves of 128-wide rows");
  constexpr int F = LPR * 4;
  constexpr int ROWS_PER_BLOCK = 256 / LPR;
  const int sub = threadIdx.x / LPR;
  const int l = threadIdx.x % LPR;
  // XCD-aware order (gfv_common.h): neighbouring destination rows gather the same source rows; the launcher rounds the grid
  // to a multiple of 8, the map is then a permutation of the block ids
  for (int r = gfv_xcd_tile(blockIdx.x, gridDim.x) * ROWS_PER_BLOCK + sub; r < n_rows; r += gridDim.x * ROWS_PER_BLOCK) {
    const int beg = rowptr[r], end = rowptr[r + 1];
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    // the column indices of up to 8 entries come back in ONE round trip (clamped reads past the row end are cheap 4-B
    // loads that are never used); the gathers then go out in groups of four (tail: up to three at once).  Accumulation
    // pattern unchanged: full groups of four -> a0..a3, a tail of < 4 entries -> a0.
    float4 lg0, lg1, lb0, lb1;   // (selected per half-row with ?: - indexing an array by cc & 1 would put it in scratch)
    if constexpr (LN) {
      lg0 = *reinterpret_cast<const float4*>(ln.gamma + 4 * l);
      lg1 = *reinterpret_cast<const float4*>(ln.gamma + 64 + 4 * l);
      lb0 = *reinterpret_cast<const float4*>(ln.beta + 4 * l);
      lb1 = *reinterpret_cast<const float4*>(ln.beta + 64 + 4 * l);
    }
    auto gat = [&](int cc) {
      float4 v = *reinterpret_cast<const float4*>(src + (size_t)cc * F + 4 * l);
      if constexpr (LN) {
        const float2 ms = *reinterpret_cast<const float2*>(ln.stats + 2 * (size_t)(cc >> 1));
        const bool hi = (cc & 1) != 0;
        const float4 ga = make_float4(hi ? lg1.x : lg0.x, hi ? lg1.y : lg0.y, hi ? lg1.z : lg0.z, hi ? lg1.w : lg0.w);
        const float4 be = make_float4(hi ? lb1.x : lb0.x, hi ? lb1.y : lb0.y, hi ? lb1.z : lb0.z, hi ? lb1.w : lb0.w);
        v.x = (v.x - ms.x) * ms.y * ga.x + be.x;
        v.y = (v.y - ms.x) * ms.y * ga.y + be.y;
        v.z = (v.z - ms.x) * ms.y * ga.z + be.z;
        v.w = (v.w - ms.x) * ms.y * ga.w + be.w;
        return v;
      }
      if (src_scale) {
        const float sc = src_scale[cc];
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
      }
      return v;
    };
    int k = beg;
    while (k < end) {
      const int last = end - 1;
      const int c0 = col[min(k, last)], c1 = col[min(k + 1, last)], c2 = col[min(k + 2, last)], c3 = col[min(k + 3, last)];
      const int c4 = col[min(k + 4, last)], c5 = col[min(k + 5, last)], c6 = col[min(k + 6, last)], c7 = col[min(k + 7, last)];
      int t0 = c0, t1 = c1, t2 = c2;   // tail candidates
      if (k + 4 <= end) {
        const float4 v0 = gat(c0), v1 = gat(c1), v2 = gat(c2), v3 = gat(c3);
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
        a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
        a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
        k += 4;
        if (k + 4 <= end) {
          const float4 w0 = gat(c4), w1 = gat(c5), w2 = gat(c6), w3 = gat(c7);
          a0.x += w0.x; a0.y += w0.y; a0.z += w0.z; a0.w += w0.w;
          a1.x += w1.x; a1.y += w1.y; a1.z += w1.z; a1.w += w1.w;
          a2.x += w2.x; a2.y += w2.y; a2.z += w2.z; a2.w += w2.w;
          a3.x += w3.x; a3.y += w3.y; a3.z += w3.z; a3.w += w3.w;
          k += 4;
          continue;
        }
        t0 = c4; t1 = c5; t2 = c6;
      }
      const int r3 = end - k;   // 0..3 tail entries
      if (r3 > 0) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 v0 = gat(t0), v1 = (r3 > 1) ? gat(t1) : z, v2 = (r3 > 2) ? gat(t2) : z;
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        if (r3 > 1) { a0.x += v1.x; a0.y += v1.y; a0.z += v1.z; a0.w += v1.w; }
        if (r3 > 2) { a0.x += v2.x; a0.y += v2.y; a0.z += v2.z; a0.w += v2.w; }
      }
      k = end;
    }
    float4 s = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y),
                           (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
    if (scale) {
      const float sc = scale[r];
      s.x *= sc; s.y *= sc; s.z *= sc; s.w *= sc;
    }
    float4* o = reinterpret_cast<float4*>(out + (size_t)r * F + 4 * l);
    if (accumulate) {
      const float4 p = *o;
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    *o = s;
  }
}

__global__ __launch_bounds__(256) void seg_gather_sum_scalar(const float* __restrict__ src, const int* __restrict__ rowptr,
                                                             const int* __restrict__ col, const float* __restrict__ scale,
                                                             const float* __restrict__ src_scale,
                                                             float* __restrict__ out, int n_rows, int F, int accumulate) {
  const long total = (long)n_rows * F;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / F), f = (int)(i % F);
    float s = 0.f;
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
      const int c = col[k];
      const float v = src[(size_t)c * F + f];
      s += src_scale ? v * src_scale[c] : v;
    }
    if (scale) s *= scale[r];
    if (accumulate) s += out[i];
    out[i] = s;
  }
}

template <int LPR>
__global__ __launch_bounds__(256) void gather_pair_kernel(const float* __restrict__ a, const int* __restrict__ s,
                                                          const int* __restrict__ r, const float* __restrict__ base,
                                                          float* __restrict__ out, int n_edges) {
  // out row = 2F floats = 2*LPR float4; lanes [0,LPR) copy a[s[e]], lanes [LPR,2LPR) copy a[r[e]]
  constexpr int F = LPR * 4;
  constexpr int LANES = 2 * LPR;
  constexpr int EPB = 256 / LANES;
  const int sub = threadIdx.x / LANES, l = threadIdx.x % LANES;
  for (int e = blockIdx.x * EPB + sub; e < n_edges; e += gridDim.x * EPB) {
    const int node = (l < LPR) ? s[e] : r[e];
    float4 v = *reinterpret_cast<const float4*>(a + (size_t)node * F + 4 * (l % LPR));
    if (base) {
      const float4 b = *reinterpret_cast<const float4*>(base + (size_t)e * 2 * F + 4 * l);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    *reinterpret_cast<float4*>(out + (size_t)e * 2 * F + 4 * l) = v;
  }
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, int n_chunks, int n,
                                                              float* __restrict__ out, int accumulate) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = 0;
    for (; c + 4 <= n_chunks; c += 4) {
      s0 += partial[(size_t)c * n + j];
      s1 += partial[(size_t)(c + 1) * n + j];
      s2 += partial[(size_t)(c + 2) * n + j];
      s3 += partial[(size_t)(c + 3) * n + j];
    }
    for (; c < n_chunks; ++c) s0 += partial[(size_t)c * n + j];
    float s = (s0 + s1) + (s2 + s3);
    if (accumulate) s += out[j];
    out[j] = s;
  }
}

// many outputs, many chunks (weight-gradient slabs): a block = 64 column quads (float4) x 4 chunk groups; chunk group
// cg sums chunks cg, cg+4, ... with four 16-B loads in flight, the four groups are combined through LDS in a fixed
// order (deterministic).  n and the buffers are 16-B aligned (GradStore layout).
__global__ __launch_bounds__(256) void reduce_partials_vec_kernel(const float* __restrict__ partial, int n_chunks, int n4,
                                                                  float* __restrict__ out, int accumulate) {
  __shared__ float4 red[4][64];
  const int q = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + q;  // column quad
  float4 s[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) s[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (j < n4) {
    const float4* p = reinterpret_cast<const float4*>(partial) + j;
    int c = cg;
    for (; c + 12 < n_chunks; c += 16) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = p[(size_t)(c + 4 * u) * n4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { s[u].x += v[u].x; s[u].y += v[u].y; s[u].z += v[u].z; s[u].w += v[u].w; }
    }
    for (; c < n_chunks; c += 4) {
      const float4 v = p[(size_t)c * n4];
      s[0].x += v.x; s[0].y += v.y; s[0].z += v.z; s[0].w += v.w;
    }
  }
  red[cg][q] = make_float4((s[0].x + s[1].x) + (s[2].x + s[3].x), (s[0].y + s[1].y) + (s[2].y + s[3].y),
                           (s[0].z + s[1].z) + (s[2].z + s[3].z), (s[0].w + s[1].w) + (s[2].w + s[3].w));
  __syncthreads();
  if (cg == 0 && j < n4) {
    const float4 a = red[0][q], b = red[1][q], c = red[2][q], d = red[3][q];
    float4 t = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z),
                           (a.w + b.w) + (c.w + d.w));
    float4* o = reinterpret_cast<float4*>(out) + j;
    if (accumulate) { const float4 pv = *o; t.x += pv.x; t.y += pv.y; t.z += pv.z; t.w += pv.w; }
    *o = t;
  }
}

// 2-D form: out[r * ld_out + c] = sum_chunks partial[chunk * chunk_stride + r * cols + c]  (r < rows, c < cols): reduces a
// [rows, cols] sub-block of the slab workspace straight into a column block of a wider gradient matrix (W1[:, 256:384]
// of the factored EdgeBlock) - same thread layout and summation order as reduce_partials_vec_kernel.
__global__ __launch_bounds__(256) void reduce_partials_2d_kernel(const float* __restrict__ partial, int n_chunks,
                                                                 long chunk_stride4, int n4, int cols4, int ld_out4,
                                                                 float* __restrict__ out) {
  __shared__ float4 red[4][64];
  const int q = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + q;
  float4 s[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) s[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (j < n4) {
    const float4* p = reinterpret_cast<const float4*>(partial) + j;
    int c = cg;
    for (; c + 12 < n_chunks; c += 16) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = p[(size_t)(c + 4 * u) * chunk_stride4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { s[u].x += v[u].x; s[u].y += v[u].y; s[u].z += v[u].z; s[u].w += v[u].w; }
    }
    for (; c < n_chunks; c += 4) {
      const float4 v = p[(size_t)c * chunk_stride4];
      s[0].x += v.x; s[0].y += v.y; s[0].z += v.z; s[0].w += v.w;
    }
  }
  red[cg][q] = make_float4((s[0].x + s[1].x) + (s[2].x + s[3].x), (s[0].y + s[1].y) + (s[2].y + s[3].y),
                           (s[0].z + s[1].z) + (s[2].z + s[3].z), (s[0].w + s[1].w) + (s[2].w + s[3].w));
  __syncthreads();
  if (cg == 0 && j < n4) {
    const float4 a = red[0][q], b = red[1][q], c = red[2][q], d = red[3][q];
    const int r = j / cols4, cc = j - r * cols4;
    reinterpret_cast<float4*>(out)[(size_t)r * ld_out4 + cc] =
        make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
  }
}

// segmented form: out[b, :] = sum of the chunk rows seg_ptr[b] .. seg_ptr[b+1]-1 (blockIdx.y = b); same thread layout
// and summation order as reduce_partials_vec_kernel.  Used to pre-reduce the per-chunk slice tokens of every graph, so
// that the (graph, head) attention blocks - only 8 per graph - do not walk hundreds of chunk partials serially.
// (1 024 threads: 16 chunk groups per column.  With the 4 groups of the first version a graph of 398 chunks was 100
// dependent round trips per thread on 17 workgroups - 12 us for 7 MB, four times per step.)
__global__ __launch_bounds__(1024) void reduce_partials_seg_kernel(const float* __restrict__ partial,
                                                                   const int* __restrict__ seg_ptr, int n4,
                                                                   float* __restrict__ out) {
  constexpr int CG = 16;
  __shared__ float4 red[CG][64];
  const int q = threadIdx.x & 63, cg = threadIdx.x >> 6, b = blockIdx.y;
  const int j = blockIdx.x * 64 + q;
  const int c0 = seg_ptr[b], c1 = seg_ptr[b + 1];
  float4 s[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) s[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (j < n4) {
    const float4* p = reinterpret_cast<const float4*>(partial) + j;
    int c = c0 + cg;
    for (; c + 3 * CG < c1; c += 4 * CG) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = p[(size_t)(c + CG * u) * n4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { s[u].x += v[u].x; s[u].y += v[u].y; s[u].z += v[u].z; s[u].w += v[u].w; }
    }
    for (; c < c1; c += CG) {
      const float4 v = p[(size_t)c * n4];
      s[0].x += v.x; s[0].y += v.y; s[0].z += v.z; s[0].w += v.w;
    }
  }
  red[cg][q] = make_float4((s[0].x + s[1].x) + (s[2].x + s[3].x), (s[0].y + s[1].y) + (s[2].y + s[3].y),
                           (s[0].z + s[1].z) + (s[2].z + s[3].z), (s[0].w + s[1].w) + (s[2].w + s[3].w));
  __syncthreads();
  if (cg == 0 && j < n4) {
    float4 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {   // fixed order: ((0+1)+(2+3)) per group of four, then the four groups pairwise
      const float4 a = red[4 * u][q], bb = red[4 * u + 1][q], c = red[4 * u + 2][q], d = red[4 * u + 3][q];
      t[u] = make_float4((a.x + bb.x) + (c.x + d.x), (a.y + bb.y) + (c.y + d.y), (a.z + bb.z) + (c.z + d.z), (a.w + bb.w) + (c.w + d.w));
    }
    reinterpret_cast<float4*>(out)[(size_t)b * n4 + j] =
        make_float4((t[0].x + t[1].x) + (t[2].x + t[3].x), (t[0].y + t[1].y) + (t[2].y + t[3].y),
                    (t[0].z + t[1].z) + (t[2].z + t[3].z), (t[0].w + t[1].w) + (t[2].w + t[3].w));
  }
}

// few outputs, many chunks (bias / LayerNorm partials): 32 columns x 8 chunk lanes per block, LDS tree at the end
__global__ __launch_bounds__(1024) void reduce_partials_small_kernel(const float* __restrict__ partial, int n_chunks, int n,
                                                                     float* __restrict__ out, int accumulate) {
  __shared__ float red[32][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + cx;
  // 8 loads in flight per thread (the loop was one memory round trip per pair of chunks); fixed summation order
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (j < n) {
    int c = ry;
    for (; c + 7 * 32 < n_chunks; c += 8 * 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += partial[(size_t)(c + 32 * u) * n + j];
    }
    for (int u = 0; c < n_chunks; c += 32, ++u) s[u] += partial[(size_t)c * n + j];
  }
  red[ry][cx] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (ry == 0 && j < n) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) t += red[r][cx];
    if (accumulate) t += out[j];
    out[j] = t;
  }
}

// ---- several reductions in ONE launch ---------------------------------------------------------------------------------
// blockIdx.y picks a piece (gfv_reduce_piece_t), blockIdx.x a group of 16 float4 output columns; 16 lanes of chunks per
// output column, each lane sums every 16th chunk with 4 loads in flight, a fixed-order LDS tree folds the 16 lanes.
// Serves the slab partials of a weight-gradient launch (hundreds of chunks x 64 KB), the per-tile LayerNorm partials
// (a thousand chunks x 1 KB) and the small attention / slice-projection partials alike, so the parameter gradients of one
// MLP need one reduction launch instead of two to five.
struct ReduceMulti {
  gfv_reduce_piece_t piece[12];
};
__global__ __launch_bounds__(256) void reduce_multi_kernel(const ReduceMulti A) {
  __shared__ float4 red[16][17];
  const gfv_reduce_piece_t P = A.piece[blockIdx.y];
  const int cols4 = P.cols >> 2;
  const long n4 = (long)P.rows * cols4;
  const int q = threadIdx.x & 15, cg = threadIdx.x >> 4;
  const long j = (long)blockIdx.x * 16 + q;
  if ((long)blockIdx.x * 16 >= n4) return;
  float4 s[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) s[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (j < n4) {
    const int r = (int)(j / cols4), cc = (int)(j - (long)r * cols4);
    const float4* p = reinterpret_cast<const float4*>(P.partial + (size_t)r * P.ld_in) + cc;
    const size_t cs4 = (size_t)(P.chunk_stride >> 2);
    int c = cg;
    for (; c + 48 < P.n_chunks; c += 64) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = p[(size_t)(c + 16 * u) * cs4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { s[u].x += v[u].x; s[u].y += v[u].y; s[u].z += v[u].z; s[u].w += v[u].w; }
    }
    for (; c < P.n_chunks; c += 16) {
      const float4 v = p[(size_t)c * cs4];
      s[0].x += v.x; s[0].y += v.y; s[0].z += v.z; s[0].w += v.w;
    }
  }
  red[cg][q] = make_float4((s[0].x + s[1].x) + (s[2].x + s[3].x), (s[0].y + s[1].y) + (s[2].y + s[3].y),
                           (s[0].z + s[1].z) + (s[2].z + s[3].z), (s[0].w + s[1].w) + (s[2].w + s[3].w));
  __syncthreads();
  if (cg == 0 && j < n4) {
    float4 t = red[0][q];
#pragma unroll
    for (int k = 1; k < 16; ++k) { const float4 o = red[k][q]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
    const int r = (int)(j / cols4), cc = (int)(j - (long)r * cols4);
    reinterpret_cast<float4*>(P.out + (size_t)r * P.ld_out)[cc] = t;
  }
}

__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ out,
                                                        int rows, int cols) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = by + j, c = bx + tx;
    tile[j][tx] = (r < rows && c < cols) ? in[(size_t)r * ld_in + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = bx + j, r = by + tx;  // out[c][r]
    if (c < cols && r < rows) out[(size_t)c * rows + r] = tile[tx][j];
  }
}

inline int grid_for(long work_items, int per_block) {
  long g = (work_items + per_block - 1) / per_block;
  if (g > 256L * 16) g = 256L * 16;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int gfv_seg_gather_sum_nnz(const float* src, const int32_t* rowptr, const int32_t* col, const float* scale,
                                      const float* src_scale, float* out, int32_t n_rows, int32_t F, int32_t accumulate,
                                      int64_t nnz_hint, void* stream);

extern "C" int gfv_seg_gather_sum(const float* src, const int32_t* rowptr, const int32_t* col, const float* scale,
                                  const float* src_scale, float* out, int32_t n_rows, int32_t F, int32_t accumulate, void* stream) {
  return gfv_seg_gather_sum_nnz(src, rowptr, col, scale, src_scale, out, n_rows, F, accumulate, -1, stream);
}

// nnz_hint: number of gathered rows (for the algorithmic-byte count of the profiler only; -1 = unknown)
extern "C" int gfv_seg_gather_sum_nnz(const float* src, const int32_t* rowptr, const int32_t* col, const float* scale,
                                      const float* src_scale, float* out, int32_t n_rows, int32_t F, int32_t accumulate,
                                      int64_t nnz_hint, void* stream) {
  return gfv_seg_gather_sum_ex(src, rowptr, col, scale, src_scale, out, n_rows, F, accumulate, nnz_hint, -1, stream);
}

// n_src_hint: rows of `src` (distinct source rows a gather can touch; -1 = unknown: priced as if every gathered row were
// its own).  Profiler bookkeeping only.
extern "C" int gfv_seg_gather_sum_ex(const float* src, const int32_t* rowptr, const int32_t* col, const float* scale,
                                     const float* src_scale, float* out, int32_t n_rows, int32_t F, int32_t accumulate,
                                     int64_t nnz_hint, int64_t n_src_hint, void* stream) {
  if (n_rows < 0 || F < 1) return GFV_ERR_ARG;
  if (n_rows == 0) return GFV_OK;
  hipStream_t st = (hipStream_t)stream;
  void* tok = nullptr;
  if (gfv_prof_enabled() && nnz_hint >= 0) {
    // every DISTINCT input element read once, every output element written once: 4 R_src F (distinct source
    // rows, at most the gathered ones) + w nnz (indices, w = 4) + 4 R F (output rows; read too when accumulating) + 4 R (row
    // pointers).  The gathered-row figure 4 nnz F - what the L2 / Infinity-Cache side serves - travels in the record's flops
    // field (bench.py reports it as l2_side_gbs; the kernel has no flops worth pricing).
    const double rsrc = (n_src_hint >= 0 && n_src_hint < nnz_hint) ? (double)n_src_hint : (double)nnz_hint;
    const double by = 4.0 * rsrc * F + 4.0 * (double)nnz_hint + 4.0 * (double)n_rows * F * (accumulate ? 2 : 1) + 4.0 * (double)n_rows;
    tok = gfv_prof_begin(GFV_K_SEG, 4.0 * (double)nnz_hint * F, by, st);
  }
  // (Round 4 measured two more forms - all eight entries of a row requested at once, one or two rows per sub-wave - against this
  // one on the bench mesh's tables: 8 - 23 % SLOWER at 8 meshes per GPU, profiles/r04_seg_forms.txt; the unconditional requests
  // past a row's end and the lower occupancy cost more than the saved round trip.  This kernel moves 5.0 - 5.3 TB/s of counter
  // bytes = 0.63 - 0.66 of 8 TB/s at 8 meshes per GPU on three of its four launch shapes: profiles/r04_seg_pmc_b8.txt.)
#define LAUNCH_VEC(LPR)                                                                                       \
  GFV_LAUNCH((seg_gather_sum_vec<LPR>), dim3(gfv_xcd_grid(grid_for(n_rows, 256 / LPR))), dim3(256), 0, st, src, \
                     rowptr, col, scale, src_scale, out, n_rows, accumulate, SegLn{})
  switch (F) {
    case 4: LAUNCH_VEC(1); break;
    case 8: LAUNCH_VEC(2); break;
    case 16: LAUNCH_VEC(4); break;
    case 32: LAUNCH_VEC(8); break;
    case 64: LAUNCH_VEC(16); break;
    case 128: LAUNCH_VEC(32); break;
    case 256: LAUNCH_VEC(64); break;
    default:
      GFV_LAUNCH(seg_gather_sum_scalar, dim3(grid_for((long)n_rows * F, 256)), dim3(256), 0, st, src, rowptr,
                         col, scale, src_scale, out, n_rows, F, accumulate);
  }
#undef LAUNCH_VEC
  gfv_prof_end(tok, st);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_seg_gather_sum_ln(const float* y, const float* stats, const float* gamma, const float* beta, const int32_t* rowptr,
                                     const int32_t* col, float* out, int32_t n_rows, int64_t nnz_hint, int64_t n_src_hint,
                                     void* stream) {
  if (n_rows < 0 || !y || !stats || !gamma || !beta || !rowptr || !col || !out) return GFV_ERR_ARG;
  if (((reinterpret_cast<size_t>(y) | reinterpret_cast<size_t>(gamma) | reinterpret_cast<size_t>(beta) | reinterpret_cast<size_t>(out)) & 15) ||
      (reinterpret_cast<size_t>(stats) & 7))
    return GFV_ERR_ARG;
  if (n_rows == 0) return GFV_OK;
  hipStream_t st = (hipStream_t)stream;
  void* tok = nullptr;
  if (gfv_prof_enabled() && nnz_hint >= 0) {   // priced as the plain 64-wide gather (+ 8 B of statistics per source row)
    const double rsrc = (n_src_hint >= 0 && n_src_hint < nnz_hint) ? (double)n_src_hint : (double)nnz_hint;
    const double by = 4.0 * rsrc * 64 + 4.0 * rsrc + 4.0 * (double)nnz_hint + 4.0 * (double)n_rows * 64 + 4.0 * (double)n_rows;
    tok = gfv_prof_begin(GFV_K_SEG, 4.0 * (double)nnz_hint * 64, by, st);
  }
  GFV_LAUNCH((seg_gather_sum_vec<16, true>), dim3(gfv_xcd_grid(grid_for(n_rows, 16))), dim3(256), 0, st, y, rowptr, col,
             (const float*)nullptr, (const float*)nullptr, out, n_rows, 0, SegLn{stats, gamma, beta});
  gfv_prof_end(tok, st);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_gather_pair(const float* a, const int32_t* s, const int32_t* r, const float* base, float* out,
                               int32_t n_edges, int32_t F, void* stream) {
  if (n_edges < 0) return GFV_ERR_ARG;
  if (n_edges == 0) return GFV_OK;
  hipStream_t st = (hipStream_t)stream;
  if (F == 64) {
    GFV_LAUNCH((gather_pair_kernel<16>), dim3(grid_for(n_edges, 8)), dim3(256), 0, st, a, s, r, base, out, n_edges);
  } else if (F == 128) {
    GFV_LAUNCH((gather_pair_kernel<32>), dim3(grid_for(n_edges, 4)), dim3(256), 0, st, a, s, r, base, out, n_edges);
  } else {
    return GFV_ERR_ARG;
  }
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_reduce_partials(const float* partial, int32_t n_chunks, int32_t n, float* out, int32_t accumulate,
                                   void* stream) {
  GfvProfScope ps_(GFV_K_REDUCE, 0, 4.0 * ((double)n_chunks + 1.0) * n, stream);
  if (n <= 0) return GFV_OK;
  if (n <= 4096 && n_chunks >= 32) {
    GFV_LAUNCH(reduce_partials_small_kernel, dim3((n + 31) / 32), dim3(1024), 0, (hipStream_t)stream, partial,
                       n_chunks, n, out, accumulate);
    GFV_CHECK_LAUNCH();
    return GFV_OK;
  }
  if ((n & 3) == 0 && n_chunks >= 16 && ((reinterpret_cast<size_t>(partial) | reinterpret_cast<size_t>(out)) & 15) == 0) {
    GFV_LAUNCH(reduce_partials_vec_kernel, dim3((n / 4 + 63) / 64), dim3(256), 0, (hipStream_t)stream, partial,
                       n_chunks, n / 4, out, accumulate);
    GFV_CHECK_LAUNCH();
    return GFV_OK;
  }
  GFV_LAUNCH(reduce_partials_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, partial,
                     n_chunks, n, out, accumulate);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_reduce_partials_2d(const float* partial, int32_t n_chunks, int64_t chunk_stride, int32_t rows,
                                      int32_t cols, int32_t ld_out, float* out, void* stream) {
  GfvProfScope ps_(GFV_K_REDUCE, 0, 4.0 * ((double)n_chunks + 1.0) * rows * cols, stream);
  if (n_chunks <= 0 || rows <= 0 || cols <= 0) return GFV_OK;
  if ((cols & 3) || (ld_out & 3) || (chunk_stride & 3) || ld_out < cols ||
      ((reinterpret_cast<size_t>(partial) | reinterpret_cast<size_t>(out)) & 15))
    return GFV_ERR_ARG;
  const int n4 = rows * (cols / 4);
  GFV_LAUNCH(reduce_partials_2d_kernel, dim3((n4 + 63) / 64), dim3(256), 0, (hipStream_t)stream, partial, n_chunks,
                     (long)(chunk_stride / 4), n4, cols / 4, ld_out / 4, out);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_reduce_multi(const gfv_reduce_piece_t* pieces, int32_t n_pieces, void* stream) {
  if (n_pieces < 1 || n_pieces > 12) return GFV_ERR_ARG;
  ReduceMulti a;
  long maxn4 = 0;
  double by = 0;
  for (int i = 0; i < n_pieces; ++i) {
    const gfv_reduce_piece_t& p = pieces[i];
    if (p.n_chunks < 1 || p.rows < 1 || p.cols < 4 || (p.cols & 3) || (p.ld_in & 3) || (p.ld_out & 3) || (p.chunk_stride & 3) ||
        p.ld_in < p.cols || p.ld_out < p.cols || ((reinterpret_cast<size_t>(p.partial) | reinterpret_cast<size_t>(p.out)) & 15))
      return GFV_ERR_ARG;
    a.piece[i] = p;
    const long n4 = (long)p.rows * (p.cols >> 2);
    maxn4 = n4 > maxn4 ? n4 : maxn4;
    by += 4.0 * ((double)p.n_chunks + 1.0) * p.rows * p.cols;
  }
  GfvProfScope ps_(GFV_K_REDUCE, 0, by, stream);
  GFV_LAUNCH(reduce_multi_kernel, dim3((unsigned)((maxn4 + 15) / 16), n_pieces), dim3(256), 0, (hipStream_t)stream, a);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_reduce_partials_seg(const float* partial, const int32_t* seg_ptr, int32_t n_seg, int32_t n, float* out,
                                       void* stream) {
  GfvProfScope ps_(GFV_K_REDUCE, 0, 4.0 * 64.0 * n_seg * n, stream);
  if (n_seg <= 0 || n <= 0) return GFV_OK;
  if ((n & 3) || ((reinterpret_cast<size_t>(partial) | reinterpret_cast<size_t>(out)) & 15)) return GFV_ERR_ARG;
  GFV_LAUNCH(reduce_partials_seg_kernel, dim3((n / 4 + 63) / 64, n_seg), dim3(1024), 0, (hipStream_t)stream,
                     partial, seg_ptr, n / 4, out);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_transpose(const float* in, int32_t ld_in, float* out, int32_t rows, int32_t cols, void* stream) {
  if (rows <= 0 || cols <= 0) return GFV_ERR_ARG;
  GFV_LAUNCH(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)stream, in,
                     ld_in, out, rows, cols);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

namespace {
__global__ __launch_bounds__(256) void transpose_batch_kernel(const gfv_transpose_desc_t* __restrict__ descs) {
  __shared__ float tile[32][33];
  const gfv_transpose_desc_t d = descs[blockIdx.z];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  if (bx >= d.cols || by >= d.rows) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int r = by + j, c = bx + tx;
    tile[j][tx] = (r < d.rows && c < d.cols) ? d.in[(size_t)r * d.ld_in + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = bx + j, r = by + tx;
    if (c < d.cols && r < d.rows) d.out[(size_t)c * (d.ld_out ? d.ld_out : d.rows) + r] = tile[tx][j];
  }
}
}  // namespace

// All weight transposes of a step in one launch; descs lives in device memory.
extern "C" int gfv_transpose_batch(const gfv_transpose_desc_t* descs, int32_t n, int32_t max_rows, int32_t max_cols,
                                   void* stream) {
  GfvProfScope ps_(GFV_K_WIMG, 0, 8.0 * (double)n * max_rows * max_cols, stream);
  if (n <= 0) return GFV_OK;
  GFV_LAUNCH(transpose_batch_kernel, dim3((max_cols + 31) / 32, (max_rows + 31) / 32, n), dim3(256), 0,
                     (hipStream_t)stream, descs);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_abi_version(void) { return GFV_ABI_VERSION; }
extern "C" int gfv_struct_size(int32_t which) {
  switch (which) {
    case 0: return (int)sizeof(gfv_seg_t);
    case 1: return (int)sizeof(gfv_layer_t);
    case 2: return (int)sizeof(gfv_rowtile_args_t);
    case 3: return (int)sizeof(gfv_wimg_desc_t);
    case 4: return (int)sizeof(gfv_dw_tile_t);
    case 5: return (int)sizeof(gfv_reduce_piece_t);
    case 6: return (int)sizeof(gfv_plan_desc_t);
    case 7: return (int)sizeof(gfv_trans_mlp_t);
    case 8: return (int)sizeof(gfv_trans_mlp_bwd_t);
    case 9: return (int)sizeof(gfv_fvm_mesh_t);
    default: return -1;
  }
}
