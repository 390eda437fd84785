// gfv-build-flags: -fno-slp-vectorize -ffp-contract=fast
// Transolver "physics attention" over per-graph slice tokens (gfx950).  Contract: include/gfv.h.
// Reference: FVMmodel/Models/GraphTransolver/GraphTransolver.py:48-95 (Graph_Physics_Attention_1D.graph_forward).
//
// Fixed geometry of the reference's block: H = 8 heads, D = 16 dims per head, G = 32 slices (TransFVGN_v2.py:28-35).
// The reference materialises [N,8,32,16] products twice per forward (28 % of its CPU time); here the per-node
// work stays in registers and only w [N,8,32] and the per-graph tokens [B,8,32,16] touch memory.
#include <cstdlib>
#include "gfv_common.h"
#include "gfv_prof.h"
#include "../../include/gfv.h"

namespace {

constexpr int H = 8, D = 16, G = 32;

// ---- w = softmax((x_mid . Ws^T + bs) / T_h) ---------------------------------------------------------------
// one thread per (node, head); a block covers 32 nodes.  Results are staged through LDS for row-contiguous stores.
__global__ __launch_bounds__(256) void slice_softmax_fwd_kernel(const float* __restrict__ xmid, const float* __restrict__ Ws,
                                                                const float* __restrict__ bs, const float* __restrict__ temp,
                                                                float* __restrict__ w, int N) {
  __shared__ float sW[G * D];
  __shared__ float sB[G];
  __shared__ __attribute__((aligned(16))) float stage[256 * (G + 4)];
  const int tid = threadIdx.x;
  for (int i = tid; i < G * D; i += 256) sW[i] = Ws[i];
  if (tid < G) sB[tid] = bs[tid];
  __syncthreads();
  const long row0 = (long)blockIdx.x * 256;  // (n,h) row index
  const long row = row0 + tid;
  const long nrows = (long)N * H;
  if (row < nrows) {
    const int h = (int)(row & (H - 1));
    float x[D];
    const float4* xp = reinterpret_cast<const float4*>(xmid + row * D);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = xp[i];
      x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
    }
    const float invT = 1.0f / temp[h];
    float l[G];
    float mx = -3.0e38f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < D; ++c) s += x[c] * sW[g * D + c];
      s = (s + sB[g]) * invT;
      l[g] = s;
      mx = fmaxf(mx, s);
    }
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      l[g] = __expf(l[g] - mx);
      sum += l[g];
    }
    const float inv = 1.0f / sum;
#pragma unroll
    for (int g = 0; g < G; g += 4)
      *reinterpret_cast<float4*>(&stage[tid * (G + 4) + g]) = make_float4(l[g] * inv, l[g + 1] * inv, l[g + 2] * inv, l[g + 3] * inv);
  }
  __syncthreads();
  // coalesced store of the block's 256 x 32 floats, float4 pieces of consecutive lanes
  for (int i = tid; i < 256 * (G / 4); i += 256) {
    const long r = row0 + i / (G / 4);
    if (r < nrows)
      reinterpret_cast<float4*>(w + r * G)[i % (G / 4)] =
          *reinterpret_cast<const float4*>(&stage[(i / (G / 4)) * (G + 4) + 4 * (i % (G / 4))]);
  }
}

// backward of the slice softmax.  gw = dL/dw.  Outputs g_xmid [N,128] and per-block partials of
// (dWs [32,16], dbs [32], dT [8]) = 552 floats per block.
__global__ __launch_bounds__(256) void slice_softmax_bwd_kernel(const float* __restrict__ xmid, const float* __restrict__ Ws,
                                                                const float* __restrict__ bs, const float* __restrict__ temp,
                                                                const float* __restrict__ w, const float* __restrict__ gw,
                                                                float* __restrict__ gxmid, float* __restrict__ partial, int N) {
  __shared__ float sW[G * D];
  __shared__ float sB[G];
  __shared__ float sGL[256 * (G + 1)];  // d logits (pre-temperature) per row
  __shared__ float sX[256 * (D + 1)];
  __shared__ float sT[256];
  const int tid = threadIdx.x;
  for (int i = tid; i < G * D; i += 256) sW[i] = Ws[i];
  if (tid < G) sB[tid] = bs[tid];
  __syncthreads();
  const long row = (long)blockIdx.x * 256 + tid;
  const long nrows = (long)N * H;
  float dT = 0.f;
  if (row < nrows) {
    const int h = (int)(row & (H - 1));
    const float invT = 1.0f / temp[h];
    float x[D];
    const float4* xp = reinterpret_cast<const float4*>(xmid + row * D);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = xp[i];
      x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
    }
    float wv[G], gv[G];
    const float4* wp = reinterpret_cast<const float4*>(w + row * G);
    const float4* gp = reinterpret_cast<const float4*>(gw + row * G);
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 a = wp[i], b = gp[i];
      wv[4 * i] = a.x; wv[4 * i + 1] = a.y; wv[4 * i + 2] = a.z; wv[4 * i + 3] = a.w;
      gv[4 * i] = b.x; gv[4 * i + 1] = b.y; gv[4 * i + 2] = b.z; gv[4 * i + 3] = b.w;
      dot += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    float gx[D];
#pragma unroll
    for (int c = 0; c < D; ++c) gx[c] = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float gz = wv[g] * (gv[g] - dot);  // grad wrt z = logit / T
      float lg = 0.f;
#pragma unroll
      for (int c = 0; c < D; ++c) lg += x[c] * sW[g * D + c];
      lg += sB[g];
      dT -= gz * lg * invT * invT;
      const float gl = gz * invT;  // grad wrt the raw logit
      sGL[tid * (G + 1) + g] = gl;
#pragma unroll
      for (int c = 0; c < D; ++c) gx[c] += gl * sW[g * D + c];
    }
    float4* op = reinterpret_cast<float4*>(gxmid + row * D);
#pragma unroll
    for (int i = 0; i < 4; ++i) op[i] = make_float4(gx[4 * i], gx[4 * i + 1], gx[4 * i + 2], gx[4 * i + 3]);
#pragma unroll
    for (int c = 0; c < D; ++c) sX[tid * (D + 1) + c] = x[c];
  } else {
#pragma unroll
    for (int g = 0; g < G; ++g) sGL[tid * (G + 1) + g] = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) sX[tid * (D + 1) + c] = 0.f;
  }
  sT[tid] = dT;
  __syncthreads();
  // dWs[g][c] = sum_rows GL[row][g] * X[row][c]: a [32 x 256] x [256 x 16] product per block -> two 16x16 MFMA tiles
  // (v_mfma_f32_16x16x4_f32, exact fp32), one per wave 0 / 1, 64 k-steps each (was: 512 scalar LDS iterations / thread)
  float* out = partial + (size_t)blockIdx.x * 552;
  {
    const int wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    if (wave < 2) {
      floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
      for (int ks = 0; ks < 64; ++ks) {
        const int r = 4 * ks + kq;
        const float a = sGL[r * (G + 1) + 16 * wave + li];   // A[i = g][k = row]
        const float b = sX[r * (D + 1) + li];                // B[k = row][j = c]
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
      }
      // D[i][j]: lane (j = li, group kq) holds rows i = 4 kq + reg
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) out[(16 * wave + 4 * kq + reg) * D + li] = acc[reg];
    }
  }
  if (tid < G) {
    float s = 0.f;
    for (int r = 0; r < 256; ++r) s += sGL[r * (G + 1) + tid];
    out[512 + tid] = s;
  }
  if (tid < H) {
    float s = 0.f;
    for (int r = tid; r < 256; r += H) s += sT[r];  // rows of head h are r = h mod 8 (block base is a multiple of 8)
    out[544 + tid] = s;
  }
}

// ---- per-chunk partial slice tokens:  T[h][g][c] = sum_n w[n,h,g] * a[n,h,c],  Nrm[h][g] = sum_n w[n,h,g] ------------
// chunk = contiguous node range inside one graph.  thread t <-> (h = t/32, g = t%32), 16 accumulators + norm.
// The rows of a chunk are contiguous in memory: 16 nodes at a time are staged in LDS with wide coalesced loads (one
// memory round trip per 16 nodes instead of one per 4), the accumulation runs out of LDS node by node (same order and
// same result as a direct loop).
__global__ __launch_bounds__(256) void slice_token_partial_kernel(const float* __restrict__ w, const float* __restrict__ a,
                                                                  const int* __restrict__ chunk_beg,
                                                                  const int* __restrict__ chunk_end,
                                                                  float* __restrict__ partial) {
  constexpr int NB = 16;                                       // nodes per staged block
  __shared__ __attribute__((aligned(16))) float sw[NB * 256];  // w[n][h*32+g]
  __shared__ __attribute__((aligned(16))) float sa[NB * 128];  // a[n][h*16+c]
  const int tid = threadIdx.x, h = tid >> 5;
  const int beg = chunk_beg[blockIdx.x], end = chunk_end[blockIdx.x];
  float acc[D];
#pragma unroll
  for (int c = 0; c < D; ++c) acc[c] = 0.f;
  float nrm = 0.f;
  for (int n0 = beg; n0 < end; n0 += NB) {
    const int nn = min(NB, end - n0);
    // w block: nn x 256 floats = nn * 64 float4; a block: nn x 128 floats = nn * 32 float4 (both contiguous)
    const float4* wsrc = reinterpret_cast<const float4*>(w + (size_t)n0 * 256);
    const float4* asrc = reinterpret_cast<const float4*>(a + (size_t)n0 * 128);
    float4 wv[4], av[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) wv[k] = (tid + 256 * k < nn * 64) ? wsrc[tid + 256 * k] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 2; ++k) av[k] = (tid + 256 * k < nn * 32) ? asrc[tid + 256 * k] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();   // the previous block has been consumed
#pragma unroll
    for (int k = 0; k < 4; ++k) reinterpret_cast<float4*>(sw)[tid + 256 * k] = wv[k];
#pragma unroll
    for (int k = 0; k < 2; ++k) reinterpret_cast<float4*>(sa)[tid + 256 * k] = av[k];
    __syncthreads();
    for (int u = 0; u < nn; ++u) {
      const float ww = sw[u * 256 + tid];
      const float4* ap = reinterpret_cast<const float4*>(&sa[u * 128 + h * D]);
      nrm += ww;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 v = ap[i];
        acc[4 * i] += ww * v.x; acc[4 * i + 1] += ww * v.y; acc[4 * i + 2] += ww * v.z; acc[4 * i + 3] += ww * v.w;
      }
    }
  }
  float* out = partial + ((size_t)blockIdx.x * 256 + tid) * 17;
#pragma unroll
  for (int c = 0; c < D; ++c) out[c] = acc[c];
  out[16] = nrm;
}

// ---- attention among the 32 slice tokens of one (graph, head) ---------------------------------------------------
struct AttnFwdArgs {
  const float* partial;      // [nchunks][256][17]
  const int* gchunk_ptr;     // [B+1] chunk range of each graph
  const float *Wq, *Wk, *Wv; // [16,16]
  float* token;              // [B,8,32,16] normalised slice tokens
  float* norm;               // [B,8,32]
  float* attn;               // [B,8,32,32]
  float* out_token;          // [B,8,32,16]
  float scale;               // dim_head ** -0.5 (GraphTransolver.py:31,80): 0.25 at hidden 128, (h / 8) ** -0.5 in general
};

__global__ __launch_bounds__(256) void slice_attention_fwd_kernel(const AttnFwdArgs A) {
  __shared__ float sTok[G][D + 1], sQ[G][D + 1], sK[G][D + 1], sV[G][D + 1], sA[G][G + 1], sNrm[G];
  __shared__ float sWq[D * D], sWk[D * D], sWv[D * D];
  const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const int bh = b * H + h;
  sWq[tid] = A.Wq[tid]; sWk[tid] = A.Wk[tid]; sWv[tid] = A.Wv[tid];
  // reduce chunk partials (fixed order): 512 token values + 32 norms
  const int c0 = A.gchunk_ptr[b], c1 = A.gchunk_ptr[b + 1];
  for (int idx = tid; idx < G * 17; idx += 256) {
    const int g = idx / 17, c = idx % 17;
    const float* pp = A.partial + ((size_t)h * G + g) * 17 + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
    int ch = c0;
    for (; ch + 8 <= c1; ch += 8) {  // 8 independent loads in flight, fixed summation order
      s0 += pp[(size_t)(ch + 0) * 4352]; s1 += pp[(size_t)(ch + 1) * 4352];
      s2 += pp[(size_t)(ch + 2) * 4352]; s3 += pp[(size_t)(ch + 3) * 4352];
      s4 += pp[(size_t)(ch + 4) * 4352]; s5 += pp[(size_t)(ch + 5) * 4352];
      s6 += pp[(size_t)(ch + 6) * 4352]; s7 += pp[(size_t)(ch + 7) * 4352];
    }
    for (; ch < c1; ++ch) s0 += pp[(size_t)ch * 4352];
    const float s = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
    if (c < D) sTok[g][c] = s; else sNrm[g] = s;
  }
  __syncthreads();
  for (int idx = tid; idx < G * D; idx += 256) {
    const int g = idx / D, c = idx % D;
    const float t = sTok[g][c] / (sNrm[g] + 1e-5f);  // GraphTransolver.py:74
    A.token[(size_t)bh * G * D + idx] = t;
  }
  if (tid < G) A.norm[bh * G + tid] = sNrm[tid];
  __syncthreads();
  for (int idx = tid; idx < G * D; idx += 256) {
    const int g = idx / D, c = idx % D;
    sTok[g][c] = sTok[g][c] / (sNrm[g] + 1e-5f);
  }
  __syncthreads();
  for (int idx = tid; idx < G * D; idx += 256) {
    const int g = idx / D, j = idx % D;
    float q = 0.f, k = 0.f, v = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) {
      const float t = sTok[g][c];
      q += t * sWq[j * D + c]; k += t * sWk[j * D + c]; v += t * sWv[j * D + c];
    }
    sQ[g][j] = q; sK[g][j] = k; sV[g][j] = v;
  }
  __syncthreads();
  for (int idx = tid; idx < G * G; idx += 256) {
    const int i = idx / G, j = idx % G;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) s += sQ[i][c] * sK[j][c];
    sA[i][j] = s * A.scale;  // dim_head ** -0.5, GraphTransolver.py:31,80
  }
  __syncthreads();
  if (tid < G) {
    float mx = -3.0e38f;
    for (int j = 0; j < G; ++j) mx = fmaxf(mx, sA[tid][j]);
    float sum = 0.f;
    for (int j = 0; j < G; ++j) { const float e = __expf(sA[tid][j] - mx); sA[tid][j] = e; sum += e; }
    const float inv = 1.0f / sum;
    for (int j = 0; j < G; ++j) sA[tid][j] *= inv;
  }
  __syncthreads();
  for (int idx = tid; idx < G * G; idx += 256) A.attn[(size_t)bh * G * G + idx] = sA[idx / G][idx % G];
  for (int idx = tid; idx < G * D; idx += 256) {
    const int i = idx / D, c = idx % D;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < G; ++j) s += sA[i][j] * sV[j][c];
    A.out_token[(size_t)bh * G * D + idx] = s;
  }
}

struct AttnBwdArgs {
  const float* gpartial;   // [nchunks][256][17] partial sums of w^T g_out_x (slot 16 unused)
  const int* gchunk_ptr;
  const float *Wq, *Wk, *Wv;
  const float* token;      // normalised tokens (saved)
  const float* norm;
  const float* attn;
  float* g_raw;            // [B,8,32,16] grad wrt the un-normalised token sums
  float* g_norm;           // [B,8,32]    grad wrt slice_norm
  float* dW_partial;       // [B*8][3][16][16]
  float scale;             // dim_head ** -0.5
};

__global__ __launch_bounds__(256) void slice_attention_bwd_kernel(const AttnBwdArgs A) {
  __shared__ float sTok[G][D + 1], sQ[G][D + 1], sK[G][D + 1], sV[G][D + 1], sA[G][G + 1], sGO[G][D + 1];
  __shared__ float sGA[G][G + 1], sGQ[G][D + 1], sGK[G][D + 1], sGV[G][D + 1], sGT[G][D + 1];
  __shared__ float sWq[D * D], sWk[D * D], sWv[D * D];
  const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const int bh = b * H + h;
  sWq[tid] = A.Wq[tid]; sWk[tid] = A.Wk[tid]; sWv[tid] = A.Wv[tid];
  const int c0 = A.gchunk_ptr[b], c1 = A.gchunk_ptr[b + 1];
  for (int idx = tid; idx < G * D; idx += 256) {
    const int g = idx / D, c = idx % D;
    const float* pp = A.gpartial + ((size_t)h * G + g) * 17 + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
    int ch = c0;
    for (; ch + 8 <= c1; ch += 8) {
      s0 += pp[(size_t)(ch + 0) * 4352]; s1 += pp[(size_t)(ch + 1) * 4352];
      s2 += pp[(size_t)(ch + 2) * 4352]; s3 += pp[(size_t)(ch + 3) * 4352];
      s4 += pp[(size_t)(ch + 4) * 4352]; s5 += pp[(size_t)(ch + 5) * 4352];
      s6 += pp[(size_t)(ch + 6) * 4352]; s7 += pp[(size_t)(ch + 7) * 4352];
    }
    for (; ch < c1; ++ch) s0 += pp[(size_t)ch * 4352];
    sGO[g][c] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));  // grad wrt out_token
    sTok[g][c] = A.token[(size_t)bh * G * D + idx];
  }
  for (int idx = tid; idx < G * G; idx += 256) sA[idx / G][idx % G] = A.attn[(size_t)bh * G * G + idx];
  __syncthreads();
  for (int idx = tid; idx < G * D; idx += 256) {
    const int g = idx / D, j = idx % D;
    float q = 0.f, k = 0.f, v = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) {
      const float t = sTok[g][c];
      q += t * sWq[j * D + c]; k += t * sWk[j * D + c]; v += t * sWv[j * D + c];
    }
    sQ[g][j] = q; sK[g][j] = k; sV[g][j] = v;
  }
  __syncthreads();
  // out = attn @ V:  gV = attn^T gO ; gAttn = gO V^T
  for (int idx = tid; idx < G * D; idx += 256) {
    const int j = idx / D, c = idx % D;
    float s = 0.f;
    for (int i = 0; i < G; ++i) s += sA[i][j] * sGO[i][c];
    sGV[j][c] = s;
  }
  for (int idx = tid; idx < G * G; idx += 256) {
    const int i = idx / G, j = idx % G;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) s += sGO[i][c] * sV[j][c];
    sGA[i][j] = s;
  }
  __syncthreads();
  if (tid < G) {  // softmax backward per row, then the 1/sqrt(D) scale
    float dot = 0.f;
    for (int j = 0; j < G; ++j) dot += sA[tid][j] * sGA[tid][j];
    for (int j = 0; j < G; ++j) sGA[tid][j] = sA[tid][j] * (sGA[tid][j] - dot) * A.scale;
  }
  __syncthreads();
  for (int idx = tid; idx < G * D; idx += 256) {
    const int i = idx / D, c = idx % D;
    float gq = 0.f, gk = 0.f;
    for (int j = 0; j < G; ++j) {
      gq += sGA[i][j] * sK[j][c];
      gk += sGA[j][i] * sQ[j][c];
    }
    sGQ[i][c] = gq; sGK[i][c] = gk;
  }
  __syncthreads();
  // q = tok Wq^T ...: g_tok = gQ Wq + gK Wk + gV Wv ; dWq[j][c] = sum_g gQ[g][j] tok[g][c]
  for (int idx = tid; idx < G * D; idx += 256) {
    const int g = idx / D, c = idx % D;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) s += sGQ[g][j] * sWq[j * D + c] + sGK[g][j] * sWk[j * D + c] + sGV[g][j] * sWv[j * D + c];
    sGT[g][c] = s;
  }
  {
    const int j = tid / D, c = tid % D;
    float dq = 0.f, dk = 0.f, dv = 0.f;
    for (int g = 0; g < G; ++g) {
      const float t = sTok[g][c];
      dq += sGQ[g][j] * t; dk += sGK[g][j] * t; dv += sGV[g][j] * t;
    }
    float* o = A.dW_partial + (size_t)bh * 3 * D * D;
    o[tid] = dq; o[D * D + tid] = dk; o[2 * D * D + tid] = dv;
  }
  __syncthreads();
  // token = raw / (norm + eps):  g_raw = g_tok / (norm+eps) ; g_norm = -sum_c g_tok * token / (norm+eps)
  for (int idx = tid; idx < G * D; idx += 256) {
    const int g = idx / D, c = idx % D;
    A.g_raw[(size_t)bh * G * D + idx] = sGT[g][c] / (A.norm[bh * G + g] + 1e-5f);
  }
  if (tid < G) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) s += sGT[tid][c] * sTok[tid][c];
    A.g_norm[bh * G + tid] = -s / (A.norm[bh * G + tid] + 1e-5f);
  }
}

// The slice tensor T[b] of a graph (8 heads x 32 x 16 floats = 16 KB) is staged in LDS when all 32 nodes of the block
// belong to one graph (head stride 516 floats: the 8 heads of a wave hit disjoint bank quads); blocks that straddle a
// graph boundary read T from L2.
constexpr int TS = G * D + 4;

__device__ __forceinline__ bool stage_T(const float* __restrict__ T, const int* __restrict__ batch, int N, float* sT) {
  const int n_first = blockIdx.x * 32;
  const int n_last = min(n_first + 31, N - 1);
  const int b0 = batch[n_first];
  const bool uniform = (batch[n_last] == b0);
  if (uniform) {
    const float4* src = reinterpret_cast<const float4*>(T + (size_t)b0 * H * G * D);
    for (int i = threadIdx.x; i < H * G * D / 4; i += 256) {
      const int h = i / (G * D / 4), r = i % (G * D / 4);
      *reinterpret_cast<float4*>(&sT[h * TS + 4 * r]) = src[i];
    }
  }
  __syncthreads();
  return uniform;
}

// ---- de-slice:  out[n,h,c] = sum_g w[n,h,g] * T[b(n),h,g,c]    (thread per (n,h)) ------------------------------------
// (a plain function, not a lambda capturing acc by reference: the closure kept acc[] in scratch memory)
__device__ __forceinline__ void deslice_row(const float4* __restrict__ wp, const float* __restrict__ Tp, float (&acc)[D]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float4 wv = wp[i];
    const float ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4* tp = reinterpret_cast<const float4*>(Tp + (4 * i + k) * D);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 t = tp[j];
        acc[4 * j] += ww[k] * t.x; acc[4 * j + 1] += ww[k] * t.y; acc[4 * j + 2] += ww[k] * t.z; acc[4 * j + 3] += ww[k] * t.w;
      }
    }
  }
}

__global__ __launch_bounds__(256) void deslice_kernel(const float* __restrict__ w, const float* __restrict__ T,
                                                      const int* __restrict__ batch, float* __restrict__ out, int N,
                                                      int accumulate) {
  __shared__ __attribute__((aligned(16))) float sT[H * TS];
  if (accumulate & 2) {   // only the workgroups whose 32 nodes span two graphs (deslice_mfma_kernel took the others)
    const int n_first = blockIdx.x * 32;
    if (batch[min(n_first + 31, N - 1)] == batch[n_first]) return;
    accumulate = 0;
  }
  const bool uniform = stage_T(T, batch, N, sT);
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  if (row >= (long)N * H) return;
  const int n = (int)(row >> 3), h = (int)(row & 7);
  float acc[D];
#pragma unroll
  for (int c = 0; c < D; ++c) acc[c] = 0.f;
  const float4* wp = reinterpret_cast<const float4*>(w + row * G);
  if (uniform) deslice_row(wp, &sT[h * TS], acc);
  else deslice_row(wp, T + ((size_t)batch[n] * H + h) * G * D, acc);
  float4* op = reinterpret_cast<float4*>(out + row * D);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float4 v = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
    if (accumulate) { const float4 p = op[j]; v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
    op[j] = v;
  }
}

// ---- gw[n,h,g] (+)= sum_c a[n,h,c] * T[b(n),h,g,c] (+ add[b(n),h,g]) --------------------------------------------------
// One thread per (node, head) computes its 32 outputs; they leave through LDS so that the block's 256 x 32 floats are
// stored (and, when accumulating, first read) as whole 128-B rows by consecutive lanes - a thread storing its own row as
// eight float4 pieces put every piece into a different 128-B line and cost 4.6x the bytes at the memory side (rocprofv3
// WRITE_SIZE: 121 MB for a 26 MB tensor).
__global__ __launch_bounds__(256) void slice_gw_kernel(const float* __restrict__ a, const float* __restrict__ T,
                                                       const float* __restrict__ add, const int* __restrict__ batch,
                                                       float* __restrict__ gw, int N, int accumulate) {
  __shared__ __attribute__((aligned(16))) float sT[H * TS];
  constexpr int SS = G + 4;   // staging row stride (floats): 16-B aligned rows
  __shared__ __attribute__((aligned(16))) float stage[256 * SS];
  const bool uniform = stage_T(T, batch, N, sT);
  const int tid = threadIdx.x;
  const long row0 = (long)blockIdx.x * 256;
  const long row = row0 + tid;
  const long nrows = (long)N * H;
  if (row < nrows) {
    const int n = (int)(row >> 3), h = (int)(row & 7);
    const size_t bh = (size_t)batch[n] * H + h;
    float x[D];
    const float4* ap = reinterpret_cast<const float4*>(a + row * D);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = ap[i];
      x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
    }
    float4 av[G / 4];   // the per-(graph, head) addend: one round trip up front instead of a load inside every dot product
#pragma unroll
    for (int i = 0; i < G / 4; ++i)
      av[i] = add ? reinterpret_cast<const float4*>(add + bh * G)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float* Tp = uniform ? &sT[h * TS] : T + bh * G * D;
#pragma unroll
    for (int i = 0; i < G / 4; ++i) {
      float r[4] = {av[i].x, av[i].y, av[i].z, av[i].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4* tp = reinterpret_cast<const float4*>(Tp + (4 * i + k) * D);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 t = tp[j];
          r[k] += x[4 * j] * t.x + x[4 * j + 1] * t.y + x[4 * j + 2] * t.z + x[4 * j + 3] * t.w;
        }
      }
      *reinterpret_cast<float4*>(&stage[tid * SS + 4 * i]) = make_float4(r[0], r[1], r[2], r[3]);
    }
  }
  __syncthreads();
  // the block's 256 rows x 32 floats leave as float4 pieces of consecutive lanes (8 per row)
  for (int i = tid; i < 256 * (G / 4); i += 256) {
    const long r = row0 + i / (G / 4);
    if (r < nrows) {
      float4 v = *reinterpret_cast<const float4*>(&stage[(i / (G / 4)) * SS + 4 * (i % (G / 4))]);
      float4* gp = reinterpret_cast<float4*>(gw + r * G) + (i % (G / 4));
      if (accumulate) { const float4 p = *gp; v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
      *gp = v;
    }
  }
}

// ---- everything behind the attention adjoint in ONE pass over the nodes ------------------------------------------------------
//   gw[n,h,g]   = sum_c g_out_x[n,h,c] T1[b,h,g,c] + sum_c fx_mid[n,h,c] T2[b,h,g,c] + g_norm[b,h,g]   (T1 = out_token, T2 = g_raw)
//   g_fx_mid    = de-slice of g_raw by w
//   g_x_mid, per-block partials of (dWs, dbs, dT): the slice-softmax adjoint from gw
// = slice_gw, deslice, slice_gw (accumulate) and slice_softmax_bwd of the four-launch form, term for term in the same
// order (the results are bit-identical, tests/test_kernels_gpu.py), without gw [N,8,32] ever leaving the registers:
// 182 MB of traffic per Transolver block become 78 MB, four launches one.  The two slice tensors of the block's graph are
// staged in LDS; the region is reused for the dWs product once every thread is through with them.
struct SlicePostArgs {
  const float* xmid; const float* Ws; const float* bs; const float* temp; const float* w; const float* gox;
  const float* T1; const float* fxm; const float* T2; const float* gnorm; const int* batch;
  float* gxmid; float* gfxmid; float* partial; int N;
  int only_straddling;   // 1: take only the workgroups whose 32 nodes span two graphs (the matrix-core form takes the others)
};

__global__ __launch_bounds__(256, 3) void slice_post_bwd_kernel(const SlicePostArgs A) {
  // 53 376 B of LDS: three workgroups per CU (the 796 workgroups of the 25 k-node bench mesh then run in one round, not 1.55)
  __shared__ float sW[G * D];
  __shared__ float sB[G];
  __shared__ __attribute__((aligned(16))) float u[256 * (G + 1) + 256 * (D + 1)];
  float* sDT = sW;   // the per-row temperature gradients take sW's place once every thread is through with it
  static_assert(2 * H * TS <= 256 * (G + 1) + 256 * (D + 1), "the two slice tensors fit the region of the dWs operands");
  float* sT1 = u;
  float* sT2 = u + H * TS;
  float* sGL = u;                    // d logits (pre-temperature) per row, stride G + 1
  float* sX = u + 256 * (G + 1);     // x_mid rows, stride D + 1
  const int tid = threadIdx.x;
  for (int i = tid; i < G * D; i += 256) sW[i] = A.Ws[i];
  if (tid < G) sB[tid] = A.bs[tid];
  const int n_first = blockIdx.x * 32;
  const int n_last = min(n_first + 31, A.N - 1);
  const int b0 = A.batch[n_first];
  const bool uniform = (A.batch[n_last] == b0);
  if (A.only_straddling && uniform) return;
  if (uniform) {
    const float4* s1 = reinterpret_cast<const float4*>(A.T1 + (size_t)b0 * H * G * D);
    const float4* s2 = reinterpret_cast<const float4*>(A.T2 + (size_t)b0 * H * G * D);
    for (int i = tid; i < H * G * D / 4; i += 256) {
      const int h = i / (G * D / 4), r = i % (G * D / 4);
      *reinterpret_cast<float4*>(&sT1[h * TS + 4 * r]) = s1[i];
      *reinterpret_cast<float4*>(&sT2[h * TS + 4 * r]) = s2[i];
    }
  }
  __syncthreads();
  const long row = (long)blockIdx.x * 256 + tid;
  const long nrows = (long)A.N * H;
  const bool live = row < nrows;
  float wv[G], gv[G];
  float dT = 0.f, invT = 0.f;
  if (live) {
    const int n = (int)(row >> 3), h = (int)(row & 7);
    const size_t bh = (size_t)A.batch[n] * H + h;
    invT = 1.0f / A.temp[h];
    float go[D], fx[D];
    const float4* gp = reinterpret_cast<const float4*>(A.gox + row * D);
    const float4* fp = reinterpret_cast<const float4*>(A.fxm + row * D);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 a = gp[i], b = fp[i];
      go[4 * i] = a.x; go[4 * i + 1] = a.y; go[4 * i + 2] = a.z; go[4 * i + 3] = a.w;
      fx[4 * i] = b.x; fx[4 * i + 1] = b.y; fx[4 * i + 2] = b.z; fx[4 * i + 3] = b.w;
    }
    const float4* wp = reinterpret_cast<const float4*>(A.w + row * G);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 a = wp[i];
      wv[4 * i] = a.x; wv[4 * i + 1] = a.y; wv[4 * i + 2] = a.z; wv[4 * i + 3] = a.w;
    }
    float4 av[G / 4];
#pragma unroll
    for (int i = 0; i < G / 4; ++i) av[i] = reinterpret_cast<const float4*>(A.gnorm + bh * G)[i];
    const float* T1p = uniform ? &sT1[h * TS] : A.T1 + bh * G * D;
    const float* T2p = uniform ? &sT2[h * TS] : A.T2 + bh * G * D;
    // per slice g: gw = (g_norm + fx_mid . T2[g]) + (0 + g_out_x . T1[g])  (slice_gw twice: the first product from zero, the
    // second on top of the per-(graph, head) addend, then their sum) and the de-slice term w[g] T2[g] (deslice_row's order);
    // a scheduling fence per group of four slices keeps the loads of later groups out of this one's registers
    float acc[D];
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = 0.f;
#pragma unroll
    for (int i = 0; i < G / 4; ++i) {
      float r1[4] = {0.f, 0.f, 0.f, 0.f};
      float r2[4] = {av[i].x, av[i].y, av[i].z, av[i].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4* t1 = reinterpret_cast<const float4*>(T1p + (4 * i + k) * D);
        const float4* t2 = reinterpret_cast<const float4*>(T2p + (4 * i + k) * D);
        const float wg = wv[4 * i + k];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 t = t1[j];
          r1[k] += go[4 * j] * t.x + go[4 * j + 1] * t.y + go[4 * j + 2] * t.z + go[4 * j + 3] * t.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 t = t2[j];
          r2[k] += fx[4 * j] * t.x + fx[4 * j + 1] * t.y + fx[4 * j + 2] * t.z + fx[4 * j + 3] * t.w;
          acc[4 * j] += wg * t.x; acc[4 * j + 1] += wg * t.y; acc[4 * j + 2] += wg * t.z; acc[4 * j + 3] += wg * t.w;
        }
        gv[4 * i + k] = r2[k] + r1[k];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    float4* op = reinterpret_cast<float4*>(A.gfxmid + row * D);
#pragma unroll
    for (int j = 0; j < 4; ++j) op[j] = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
  }
  __syncthreads();   // every thread is through with the slice tensors: their region becomes (sGL, sX)
  if (live) {
    float x[D];
    const float4* xp = reinterpret_cast<const float4*>(A.xmid + row * D);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = xp[i];
      x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
    }
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      dot += wv[4 * i] * gv[4 * i] + wv[4 * i + 1] * gv[4 * i + 1] + wv[4 * i + 2] * gv[4 * i + 2] + wv[4 * i + 3] * gv[4 * i + 3];
    float gx[D];
#pragma unroll
    for (int c = 0; c < D; ++c) gx[c] = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float gz = wv[g] * (gv[g] - dot);  // grad wrt z = logit / T
      float lg = 0.f;
#pragma unroll
      for (int c = 0; c < D; ++c) lg += x[c] * sW[g * D + c];
      lg += sB[g];
      dT -= gz * lg * invT * invT;
      const float gl = gz * invT;  // grad wrt the raw logit
      sGL[tid * (G + 1) + g] = gl;
#pragma unroll
      for (int c = 0; c < D; ++c) gx[c] += gl * sW[g * D + c];
    }
    float4* op = reinterpret_cast<float4*>(A.gxmid + row * D);
#pragma unroll
    for (int i = 0; i < 4; ++i) op[i] = make_float4(gx[4 * i], gx[4 * i + 1], gx[4 * i + 2], gx[4 * i + 3]);
#pragma unroll
    for (int c = 0; c < D; ++c) sX[tid * (D + 1) + c] = x[c];
  } else {
#pragma unroll
    for (int g = 0; g < G; ++g) sGL[tid * (G + 1) + g] = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) sX[tid * (D + 1) + c] = 0.f;
  }
  __syncthreads();   // (sW is free)
  sDT[tid] = dT;
  __syncthreads();
  float* out = A.partial + (size_t)blockIdx.x * 552;
  {
    const int wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    if (wave < 2) {
      floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
      for (int ks = 0; ks < 64; ++ks) {
        const int r = 4 * ks + kq;
        const float a = sGL[r * (G + 1) + 16 * wave + li];
        const float b = sX[r * (D + 1) + li];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) out[(16 * wave + 4 * kq + reg) * D + li] = acc[reg];
    }
  }
  if (tid < G) {
    float s = 0.f;
    for (int r = 0; r < 256; ++r) s += sGL[r * (G + 1) + tid];
    out[512 + tid] = s;
  }
  if (tid < H) {
    float s = 0.f;
    for (int r = tid; r < 256; r += H) s += sDT[r];
    out[544 + tid] = s;
  }
}

// The same pass on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products).  A workgroup takes 32 nodes of ONE graph
// (workgroups that straddle two graphs are left to slice_post_bwd_kernel), wave w the heads 2w, 2w + 1; per (head, 16-node
// tile) every product of the pass is a 16-row tile product with the contraction index permuted so that ONE float4 per lane
// feeds the four k-steps of an operand:
//   gw  [16 x 32] = fx_mid [16 x 16] T2^T + g_out_x [16 x 16] T1^T + g_norm         lg [16 x 32] = x_mid Ws^T + bs
//   gl = w (gw - <w, gw>) / T  (accumulator layout: lane (g, q) holds rows 4q .. 4q+3; the row sum is a DPP row reduction)
//   g_x_mid [16 x 16] = gl Ws   (gl to operand layout through 2 KB of wave-private LDS)      g_fx_mid [16 x 16] = w T2
//   dWs [32 x 16] += gl^T x_mid (the accumulator registers ARE the operand fragments), dbs, dT lane-private
// 48 MFMAs per tile against ~5 000 vector instructions per 64 (node, head) rows in the scalar form.
__global__ __launch_bounds__(256, 2) void slice_post_bwd_mfma_kernel(const SlicePostArgs A) {
  __shared__ __attribute__((aligned(16))) float sT1[H * TS];
  __shared__ __attribute__((aligned(16))) float sT2[H * TS];
  __shared__ __attribute__((aligned(16))) float sWs[G * D];
  __shared__ float sBs[G];
  constexpr int TRS = G + 4;                                   // row stride of the transposition tile (16-B aligned rows)
  __shared__ __attribute__((aligned(16))) float sTr[4][16 * TRS];
  __shared__ float sRed[4][552];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, q = lane >> 4;
  const int n_first = blockIdx.x * 32;
  const int n_last = min(n_first + 31, A.N - 1);
  const int b0 = A.batch[n_first];
  if (A.batch[n_last] != b0) return;   // (block-uniform) two graphs: the scalar kernel's workgroup
  {
    const float4* s1 = reinterpret_cast<const float4*>(A.T1 + (size_t)b0 * H * G * D);
    const float4* s2 = reinterpret_cast<const float4*>(A.T2 + (size_t)b0 * H * G * D);
    for (int i = tid; i < H * G * D / 4; i += 256) {
      const int h = i / (G * D / 4), r = i % (G * D / 4);
      *reinterpret_cast<float4*>(&sT1[h * TS + 4 * r]) = s1[i];
      *reinterpret_cast<float4*>(&sT2[h * TS + 4 * r]) = s2[i];
    }
    for (int i = tid; i < G * D; i += 256) sWs[i] = A.Ws[i];
    if (tid < G) sBs[tid] = A.bs[tid];
  }
  __syncthreads();
  // head-independent operands: Ws as the B operand of lg (slice li of tile t, columns 4q ..) and of g_x_mid (slice 16u+4q+s, column li)
  float bws[2][4], wsd[2][4], bsv[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float4 v = *reinterpret_cast<const float4*>(&sWs[(16 * t + li) * D + 4 * q]);
    bws[t][0] = v.x; bws[t][1] = v.y; bws[t][2] = v.z; bws[t][3] = v.w;
    bsv[t] = sBs[16 * t + li];
#pragma unroll
    for (int s = 0; s < 4; ++s) wsd[t][s] = sWs[(16 * t + 4 * q + s) * D + li];
  }
  floatx4 dws[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float dbs[2] = {0.f, 0.f}, dTw[2] = {0.f, 0.f};
  float* tr = sTr[wave];
#pragma unroll 1
  for (int hh = 0; hh < 2; ++hh) {
    const int h = 2 * wave + hh;
    const float invT = 1.0f / A.temp[h];
    float b1[2][4], b2[2][4], t2d[2][4], gnv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float4 u1 = *reinterpret_cast<const float4*>(&sT1[h * TS + (16 * t + li) * D + 4 * q]);
      const float4 u2 = *reinterpret_cast<const float4*>(&sT2[h * TS + (16 * t + li) * D + 4 * q]);
      b1[t][0] = u1.x; b1[t][1] = u1.y; b1[t][2] = u1.z; b1[t][3] = u1.w;
      b2[t][0] = u2.x; b2[t][1] = u2.y; b2[t][2] = u2.z; b2[t][3] = u2.w;
      gnv[t] = A.gnorm[((size_t)b0 * H + h) * G + 16 * t + li];
#pragma unroll
      for (int s = 0; s < 4; ++s) t2d[t][s] = sT2[h * TS + (16 * t + 4 * q + s) * D + li];
    }
#pragma unroll 1
    for (int nt = 0; nt < 2; ++nt) {
      const int n0 = n_first + 16 * nt;
      // operand layout: lane (row li, columns 4q ..)
      const int nodeA = n0 + li;
      const bool liveA = nodeA < A.N;
      const size_t rowA = (size_t)(liveA ? nodeA : A.N - 1) * H + h;
      const float mA = liveA ? 1.0f : 0.0f;
      float4 vgo = *reinterpret_cast<const float4*>(A.gox + rowA * D + 4 * q);
      float4 vfx = *reinterpret_cast<const float4*>(A.fxm + rowA * D + 4 * q);
      float4 vx = *reinterpret_cast<const float4*>(A.xmid + rowA * D + 4 * q);
      float4 vw0 = *reinterpret_cast<const float4*>(A.w + rowA * G + 4 * q);
      float4 vw1 = *reinterpret_cast<const float4*>(A.w + rowA * G + 16 + 4 * q);
      const float ago[4] = {vgo.x * mA, vgo.y * mA, vgo.z * mA, vgo.w * mA};
      const float afx[4] = {vfx.x * mA, vfx.y * mA, vfx.z * mA, vfx.w * mA};
      const float ax[4] = {vx.x * mA, vx.y * mA, vx.z * mA, vx.w * mA};
      const float aw[2][4] = {{vw0.x * mA, vw0.y * mA, vw0.z * mA, vw0.w * mA}, {vw1.x * mA, vw1.y * mA, vw1.z * mA, vw1.w * mA}};
      // accumulator layout: lane (column li, rows 4q + r)
      float wd[2][4], xb[4];
      size_t rowD[4];
      bool liveD[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nd = n0 + 4 * q + r;
        liveD[r] = nd < A.N;
        rowD[r] = (size_t)(liveD[r] ? nd : A.N - 1) * H + h;
        const float m = liveD[r] ? 1.0f : 0.0f;
        wd[0][r] = A.w[rowD[r] * G + li] * m;
        wd[1][r] = A.w[rowD[r] * G + 16 + li] * m;
        xb[r] = A.xmid[rowD[r] * D + li] * m;
      }
      floatx4 g2[2], g1[2], lg[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        g2[t] = floatx4{gnv[t], gnv[t], gnv[t], gnv[t]};
        g1[t] = floatx4{0.f, 0.f, 0.f, 0.f};
        lg[t] = floatx4{bsv[t], bsv[t], bsv[t], bsv[t]};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          g2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(afx[s], b2[t][s], g2[t], 0, 0, 0);
          g1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ago[s], b1[t][s], g1[t], 0, 0, 0);
          lg[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[s], bws[t][s], lg[t], 0, 0, 0);
        }
      }
      float gl[2][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gw0 = g2[0][r] + g1[0][r], gw1 = g2[1][r] + g1[1][r];
        const float dot = gfv_row16_sum(wd[0][r] * gw0 + wd[1][r] * gw1);
        const float gz0 = wd[0][r] * (gw0 - dot), gz1 = wd[1][r] * (gw1 - dot);
        dTw[hh] -= (gz0 * lg[0][r] + gz1 * lg[1][r]) * invT * invT;
        gl[0][r] = gz0 * invT;
        gl[1][r] = gz1 * invT;
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int s = 0; s < 4; ++s) dws[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(gl[t][s], xb[s], dws[t], 0, 0, 0);
        dbs[t] += (gl[t][0] + gl[t][1]) + (gl[t][2] + gl[t][3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) tr[(4 * q + r) * TRS + 16 * t + li] = gl[t][r];
      }
      __builtin_amdgcn_wave_barrier();
      floatx4 gxa = {0.f, 0.f, 0.f, 0.f}, gfa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float4 v = *reinterpret_cast<const float4*>(&tr[li * TRS + 16 * u + 4 * q]);
        const float agl[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          gxa = __builtin_amdgcn_mfma_f32_16x16x4f32(agl[s], wsd[u][s], gxa, 0, 0, 0);
          gfa = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[u][s], t2d[u][s], gfa, 0, 0, 0);
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (liveD[r]) {
          A.gxmid[rowD[r] * D + li] = gxa[r];
          A.gfxmid[rowD[r] * D + li] = gfa[r];
        }
      }
    }
  }
  // this wave's share of the block's partials (dWs [32,16] | dbs [32] | dT [8])
  float* red = sRed[wave];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(16 * t + 4 * q + r) * D + li] = dws[t][r];
    float a, b;
    gfv_lane_xor16(dbs[t], a, b);
    const float v = a + b;
    gfv_lane_xor32(v, a, b);
    if (q == 0) red[512 + 16 * t + li] = a + b;
  }
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const float v = gfv_wave_sum(dTw[hh]);
    if (lane == 0) red[544 + hh] = v;
  }
  __syncthreads();
  float* out = A.partial + (size_t)blockIdx.x * 552;
  for (int i = tid; i < 544; i += 256) out[i] = (sRed[0][i] + sRed[1][i]) + (sRed[2][i] + sRed[3][i]);
  if (tid < H) out[544 + tid] = sRed[tid >> 1][544 + (tid & 1)];
}

// ---- slice weights and per-chunk slice tokens on the matrix cores ------------------------------------------------------------
// SOFTMAX: w = softmax((x_mid Ws^T + bs) / T_h) is formed here (and written out) - slice_softmax_fwd and slice_token_partial
// in one pass, without the 26 MB round trip of w in between; else w is read.  One workgroup per chunk (a node range inside one
// graph), wave v the heads 2 v, 2 v + 1.  Per (head, 16-node tile): logits [16 x 32] = x Ws^T on v_mfma_f32_16x16x4_f32 (one
// float4 per lane per operand: the contraction index is permuted), the row maxima / sums of the softmax are DPP row
// reductions over the 16 lanes that hold a row's slices, and the token sums T[g][c] += w^T a take the accumulator-layout
// registers of w as their operand fragments directly (lane (g, q) holds rows 4 q .. 4 q + 3 = the four k-steps).
template <bool SOFTMAX>
__global__ __launch_bounds__(256, 2) void slice_token_mfma_kernel(const float* __restrict__ xmid, const float* __restrict__ Ws,
                                                                  const float* __restrict__ bs, const float* __restrict__ temp,
                                                                  float* __restrict__ w, const float* __restrict__ a,
                                                                  const int* __restrict__ chunk_beg, const int* __restrict__ chunk_end,
                                                                  float* __restrict__ partial) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, q = lane >> 4;
  const int beg = chunk_beg[blockIdx.x], end = chunk_end[blockIdx.x];
  float bws[2][4], bsv[2];
  if (SOFTMAX) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float4 v = *reinterpret_cast<const float4*>(Ws + (16 * t + li) * D + 4 * q);
      bws[t][0] = v.x; bws[t][1] = v.y; bws[t][2] = v.z; bws[t][3] = v.w;
      bsv[t] = bs[16 * t + li];
    }
  }
  float* out = partial + (size_t)blockIdx.x * 256 * 17;
#pragma unroll 1
  for (int hh = 0; hh < 2; ++hh) {
    const int h = 2 * wave + hh;
    const float invT = SOFTMAX ? 1.0f / temp[h] : 0.f;
    floatx4 tok[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float nrm[2] = {0.f, 0.f};
#pragma unroll 1
    for (int n0 = beg; n0 < end; n0 += 16) {
      // accumulator layout: lane (column li, rows 4 q + r)
      float wd[2][4], ab[4];
      size_t rowD[4];
      bool liveD[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nd = n0 + 4 * q + r;
        liveD[r] = nd < end;
        rowD[r] = (size_t)(liveD[r] ? nd : end - 1) * H + h;
        ab[r] = liveD[r] ? a[rowD[r] * D + li] : 0.f;
      }
      if (SOFTMAX) {
        const int nA = n0 + li;
        const bool liveA = nA < end;
        const size_t rowA = (size_t)(liveA ? nA : end - 1) * H + h;
        const float4 vx = *reinterpret_cast<const float4*>(xmid + rowA * D + 4 * q);
        const float ax[4] = {vx.x, vx.y, vx.z, vx.w};
        floatx4 lg[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          lg[t] = floatx4{bsv[t], bsv[t], bsv[t], bsv[t]};
#pragma unroll
          for (int s = 0; s < 4; ++s) lg[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[s], bws[t][s], lg[t], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float z0 = lg[0][r] * invT, z1 = lg[1][r] * invT;
          const float mx = gfv_row16_max(fmaxf(z0, z1));
          const float e0 = __expf(z0 - mx), e1 = __expf(z1 - mx);
          const float inv = 1.0f / gfv_row16_sum(e0 + e1);
          const float m = liveD[r] ? 1.0f : 0.0f;
          wd[0][r] = e0 * inv * m;
          wd[1][r] = e1 * inv * m;
          if (liveD[r]) {
            w[rowD[r] * G + li] = wd[0][r];
            w[rowD[r] * G + 16 + li] = wd[1][r];
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          wd[0][r] = liveD[r] ? w[rowD[r] * G + li] : 0.f;
          wd[1][r] = liveD[r] ? w[rowD[r] * G + 16 + li] : 0.f;
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int s = 0; s < 4; ++s) tok[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wd[t][s], ab[s], tok[t], 0, 0, 0);
        nrm[t] += (wd[t][0] + wd[t][1]) + (wd[t][2] + wd[t][3]);
      }
    }
    // T[g = 16 t + 4 q + r][c = li], norm[g = 16 t + li] (summed over the four row groups)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(size_t)(h * G + 16 * t + 4 * q + r) * 17 + li] = tok[t][r];
      float x0, x1;
      gfv_lane_xor16(nrm[t], x0, x1);
      const float v = x0 + x1;
      gfv_lane_xor32(v, x0, x1);
      if (q == 0) out[(size_t)(h * G + 16 * t + li) * 17 + 16] = x0 + x1;
    }
  }
}

// ---- de-slice on the matrix cores: out[n, h, c] = sum_g w[n, h, g] T[b(n), h, g, c] (workgroups of one graph; the others: deslice_kernel) ----
__global__ __launch_bounds__(256, 2) void deslice_mfma_kernel(const float* __restrict__ w, const float* __restrict__ T,
                                                              const int* __restrict__ batch, float* __restrict__ out, int N) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, q = lane >> 4;
  const int n_first = blockIdx.x * 32;
  const int n_last = min(n_first + 31, N - 1);
  const int b0 = batch[n_first];
  if (batch[n_last] != b0) return;
#pragma unroll 1
  for (int hh = 0; hh < 2; ++hh) {
    const int h = 2 * wave + hh;
    const float* Tp = T + ((size_t)b0 * H + h) * G * D;
    float t2d[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) t2d[u][s] = Tp[(16 * u + 4 * q + s) * D + li];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int nA = n_first + 16 * nt + li;
      const bool liveA = nA < N;
      const size_t rowA = (size_t)(liveA ? nA : N - 1) * H + h;
      const float4 v0 = *reinterpret_cast<const float4*>(w + rowA * G + 4 * q);
      const float4 v1 = *reinterpret_cast<const float4*>(w + rowA * G + 16 + 4 * q);
      const float aw[2][4] = {{v0.x, v0.y, v0.z, v0.w}, {v1.x, v1.y, v1.z, v1.w}};
      floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[u][s], t2d[u][s], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nd = n_first + 16 * nt + 4 * q + r;
        if (nd < N) out[((size_t)nd * H + h) * D + li] = acc[r];
      }
    }
  }
}

}  // namespace

extern "C" int gfv_slice_softmax_fwd(const float* xmid, const float* Ws, const float* bs, const float* temp, float* w,
                                     int32_t N, void* stream) {
  GfvProfScope ps_(GFV_K_SLICE, 0, 1536.0 * N, stream);   // x_mid [N,128] in, w [N,8,32] out
  if (N <= 0) return N == 0 ? GFV_OK : GFV_ERR_ARG;
  GFV_LAUNCH(slice_softmax_fwd_kernel, dim3(gfv_div_up((long)N * H, 256)), dim3(256), 0, (hipStream_t)stream,
                     xmid, Ws, bs, temp, w, N);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_slice_softmax_bwd_blocks(int32_t N) { return gfv_div_up((long)N * H, 256); }

extern "C" int gfv_slice_softmax_bwd(const float* xmid, const float* Ws, const float* bs, const float* temp,
                                     const float* w, const float* gw, float* gxmid, float* partial, int32_t N,
                                     void* stream) {
  GfvProfScope ps_(GFV_K_SLICE, 0, 3072.0 * N, stream);   // x_mid, w, gw in, g_x_mid out
  if (N <= 0) return N == 0 ? GFV_OK : GFV_ERR_ARG;
  GFV_LAUNCH(slice_softmax_bwd_kernel, dim3(gfv_div_up((long)N * H, 256)), dim3(256), 0, (hipStream_t)stream,
                     xmid, Ws, bs, temp, w, gw, gxmid, partial, N);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

static int slice_mfma_on() {
  static const int on = [] { const char* e = getenv("GFV_SLICE_MFMA"); return e ? atoi(e) : 1; }();
  return on;
}

extern "C" int gfv_slice_token_partial(const float* w, const float* a, const int32_t* chunk_beg, const int32_t* chunk_end,
                                       int32_t n_chunks, float* partial, void* stream) {
  GfvProfScope ps_(GFV_K_SLICE, 0, 64.0 * 1536.0 * n_chunks, stream);   // 64-node chunks: w + a rows in
  if (n_chunks <= 0) return n_chunks == 0 ? GFV_OK : GFV_ERR_ARG;
  if (slice_mfma_on())
    GFV_LAUNCH((slice_token_mfma_kernel<false>), dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, nullptr, nullptr, nullptr,
                       nullptr, const_cast<float*>(w), a, chunk_beg, chunk_end, partial);
  else
    GFV_LAUNCH(slice_token_partial_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, w, a, chunk_beg,
                       chunk_end, partial);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_slice_softmax_token(const float* xmid, const float* Ws, const float* bs, const float* temp, const float* a,
                                       const int32_t* chunk_beg, const int32_t* chunk_end, int32_t n_chunks, float* w,
                                       float* partial, void* stream) {
  GfvProfScope ps_(GFV_K_SLICE, 0, 64.0 * 2048.0 * n_chunks, stream);   // x_mid + a rows in, w out
  if (n_chunks <= 0) return n_chunks == 0 ? GFV_OK : GFV_ERR_ARG;
  GFV_LAUNCH((slice_token_mfma_kernel<true>), dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, xmid, Ws, bs, temp, w, a,
                     chunk_beg, chunk_end, partial);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_hidden_size(void);   // rowtile.hip (gfv_set_hidden_size)
static float attn_scale() { return 1.0f / sqrtf((float)(gfv_hidden_size() / 8)); }

extern "C" int gfv_slice_attention_fwd(const float* partial, const int32_t* gchunk_ptr, int32_t B, const float* Wq,
                                       const float* Wk, const float* Wv, float* token, float* norm, float* attn,
                                       float* out_token, void* stream) {
  GfvProfScope ps_(GFV_K_SLICE, 0, 60000.0 * B, stream);
  if (B <= 0) return B == 0 ? GFV_OK : GFV_ERR_ARG;
  AttnFwdArgs a{partial, gchunk_ptr, Wq, Wk, Wv, token, norm, attn, out_token, attn_scale()};
  GFV_LAUNCH(slice_attention_fwd_kernel, dim3(B, H), dim3(256), 0, (hipStream_t)stream, a);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_slice_attention_bwd(const float* gpartial, const int32_t* gchunk_ptr, int32_t B, const float* Wq,
                                       const float* Wk, const float* Wv, const float* token, const float* norm,
                                       const float* attn, float* g_raw, float* g_norm, float* dW_partial, void* stream) {
  GfvProfScope ps_(GFV_K_SLICE, 0, 100000.0 * B, stream);
  if (B <= 0) return B == 0 ? GFV_OK : GFV_ERR_ARG;
  AttnBwdArgs a{gpartial, gchunk_ptr, Wq, Wk, Wv, token, norm, attn, g_raw, g_norm, dW_partial, attn_scale()};
  GFV_LAUNCH(slice_attention_bwd_kernel, dim3(B, H), dim3(256), 0, (hipStream_t)stream, a);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_deslice(const float* w, const float* T, const int32_t* batch, float* out, int32_t N, int32_t accumulate,
                           void* stream) {
  GfvProfScope ps_(GFV_K_SLICE, 0, 1536.0 * N, stream);   // w in, out [N,128]
  if (N <= 0) return N == 0 ? GFV_OK : GFV_ERR_ARG;
  static const int dmfma = [] { const char* e = getenv("GFV_DESLICE_MFMA"); return e ? atoi(e) : 1; }();
  if (dmfma && slice_mfma_on() && !(accumulate & 1)) {
    // workgroups of one graph on the matrix cores, then (unless the caller says the batch is ONE graph: accumulate bit 2) the
    // ones that straddle two graphs
    GFV_LAUNCH(deslice_mfma_kernel, dim3(gfv_div_up((long)N * H, 256)), dim3(256), 0, (hipStream_t)stream, w, T, batch, out, N);
    if (!(accumulate & 4))
      GFV_LAUNCH(deslice_kernel, dim3(gfv_div_up((long)N * H, 256)), dim3(256), 0, (hipStream_t)stream, w, T, batch, out, N, 2);
  } else {
    accumulate &= 1;
    GFV_LAUNCH(deslice_kernel, dim3(gfv_div_up((long)N * H, 256)), dim3(256), 0, (hipStream_t)stream, w, T, batch,
                       out, N, accumulate);
  }
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_slice_gw(const float* a, const float* T, const float* add, const int32_t* batch, float* gw, int32_t N,
                            int32_t accumulate, void* stream) {
  GfvProfScope ps_(GFV_K_SLICE, 0, (accumulate ? 3072.0 : 1536.0) * N, stream);   // a in, gw out (+ gw in, fx_mid)
  if (N <= 0) return N == 0 ? GFV_OK : GFV_ERR_ARG;
  GFV_LAUNCH(slice_gw_kernel, dim3(gfv_div_up((long)N * H, 256)), dim3(256), 0, (hipStream_t)stream, a, T, add,
                     batch, gw, N, accumulate);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_slice_post_bwd(const float* xmid, const float* Ws, const float* bs, const float* temp, const float* w,
                                  const float* g_out_x, const float* out_token, const float* fx_mid, const float* g_raw,
                                  const float* g_norm, const int32_t* batch, float* g_x_mid, float* g_fx_mid, float* partial,
                                  int32_t N, int32_t n_graphs, void* stream) {
  // x_mid, g_out_x, fx_mid [N,128] + w [N,8,32] in, g_x_mid + g_fx_mid out
  GfvProfScope ps_(GFV_K_SLICE, 0, (3 * 512.0 + 1024.0 + 2 * 512.0) * N, stream);
  if (N <= 0) return N == 0 ? GFV_OK : GFV_ERR_ARG;
  SlicePostArgs a{xmid, Ws, bs, temp, w, g_out_x, out_token, fx_mid, g_raw, g_norm, batch, g_x_mid, g_fx_mid, partial, N, 0};
  static const int mfma = [] { const char* e = getenv("GFV_SLICE_MFMA"); return e ? atoi(e) : 1; }();
  const dim3 grid(gfv_div_up((long)N * H, 256));
  if (mfma) {
    // workgroups of one graph on the matrix cores; the ones that straddle two graphs (none in a one-graph batch) by the scalar form
    GFV_LAUNCH(slice_post_bwd_mfma_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    if (n_graphs != 1) {
      a.only_straddling = 1;
      GFV_LAUNCH(slice_post_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    }
  } else {
    GFV_LAUNCH(slice_post_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  }
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}
