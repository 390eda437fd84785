// gfv-build-flags: -fno-slp-vectorize
// The row-local Linear chains of a Transolver block as ONE launch each (round 5):
//   forward   fx1 = to_out(out_x) + fx_in;  z = linear_pre(LayerNorm_2(fx1));  out = linear_post(gelu(z)) + fx1
//             (GraphTransolver.py:93-95 to_out, :163-169 ln_2 / mlp / residuals: three single-layer launches of lin1.hip)
//   backward  g_z = (g W_post) gelu'(z);  g_fx1 = LayerNorm_2-backward(g_z W_pre; fx1) + g;  g_out_x = g_fx1 W_out
//             (their adjoints: the GELU' launch, the LayerNorm-backward launch, the plain one)
// Between the three Linears of a chain nothing crosses rows, so the activations stay in the wave that owns the rows: a workgroup =
// 8 waves x 16 rows (lin1.hip's geometry), a layer's split-fp16 image is staged in LDS (64 / 128 KB), the products run as in
// lin1_kernel, and the accumulator layout of a layer's output - lane (row, g) holds columns 16 nt + 4 g + r - IS the B-fragment
// layout of the next layer's input (k-group T = n-tiles 2 T, 2 T + 1: the register-resident chain's trick, tchain_kernel.h), so a
// layer's result is split in place and multiplied again.  What the launch saves over its three predecessors: two kernel
// boundaries with their ~10 us single-tile floors each (profiles/r05_latency_floor.txt), the re-reads of fx1 / z (forward) and
// g_z / g_fx1 (backward) by the next launch, two of three row loads.  Saved tensors, summation orders and scales are those of the
// separate launches (row scales for the layer inputs, LayerNorm statistics in the owning wave, (dgamma, dbeta) per 64-row tile
// through LDS) except one: gelu(z) is split behind the fixed scale the column-owner kernels use for hidden activations
// (colchain_kernel.h CC_SH) instead of a row scale over its 256 columns.
#include <atomic>
#include <cstdlib>

#include "../../include/gfv.h"
#include "tchain_kernel.h"

int* gfv_internal_status_ptr();
extern "C" int gfv_hidden_size(void);
extern "C" int gfv_f16split_enabled(void);

namespace {

constexpr float TM_SH = 16.0f, TM_SH_INV = 1.0f / 16.0f, TM_SH_LIMIT = 2048.0f;

// all 512 threads copy an image of n16 16-byte units into LDS; a barrier on both sides (the previous layer's fragment reads are
// over / the image is complete)
template <int PER>
__device__ __forceinline__ void tm_stage(gfv_uint4* lds, const void* image, int tid) {
  const gfv_uint4* src = reinterpret_cast<const gfv_uint4*>(image);
  gfv_uint4 t[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) t[u] = src[(size_t)u * 512 + tid];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PER; ++u) lds[u * 512 + tid] = t[u];
  __syncthreads();
}

// acc = sum_T W[pass p][T][nt] x[T] for one n-tile (image in LDS: [pass][T][nt][hi 64 | lo 64] x 16 B)
template <int KS, int LOWP>
__device__ __forceinline__ floatx4 tm_mma(const gfv_uint4* img, int p, int nt, int lane, const gfv_f16x8 (&xh)[KS], const gfv_f16x8 (&xl)[KS]) {
  floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int T = 0; T < KS; ++T) {
    const gfv_uint4* f = img + ((p * KS + T) * 8 + nt) * 128 + lane;
    const gfv_f16x8 wh = __builtin_bit_cast(gfv_f16x8, f[0]);
    if (!LOWP) {
      const gfv_f16x8 wl = __builtin_bit_cast(gfv_f16x8, f[64]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[T], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[T], acc, 0, 0, 0);
    }
    acc = gfv_mma_hh<LOWP == 2>(wh, xh[T], acc);
  }
  // (one n-tile's fragment reads in flight at a time: hoisted across the unrolled n-tile loops they cost the register budget)
  __builtin_amdgcn_sched_barrier(0);
  return acc;
}

// 32 values of a row per lane (8 n-tiles x 4) -> power-of-two row scale -> (hi, lo) fragments of 4 k-groups; returns the scale
template <bool BF>
__device__ __forceinline__ float tm_split_row(const float (&h)[8][4], gfv_f16x8 (&xh)[4], gfv_f16x8 (&xl)[4]) {
  float m0 = 0.f, m1 = 0.f;
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    m0 = max3_abs(m0, h[nt][0], h[nt][1]);
    m1 = max3_abs(m1, h[nt][2], h[nt][3]);
  }
  const float sx = gfv_pow2_scale(row_max4(fmaxf(m0, m1)));
#pragma unroll
  for (int T = 0; T < 4; ++T) {
    float e[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { e[r] = h[2 * T][r] * sx; e[4 + r] = h[2 * T + 1][r] * sx; }
    gfv_uint4 hi, lo;
    gfv_split8_t<BF>(e, hi, lo);
    xh[T] = __builtin_bit_cast(gfv_f16x8, hi);
    xl[T] = __builtin_bit_cast(gfv_f16x8, lo);
  }
  return sx;
}

struct TmFwdArgs {
  const float* x;      // out_x [M,128]
  const float* res;    // fx_in [M,128]
  const void* imgA;    // to_out      [128,128]
  const void* imgB;    // linear_pre  [256,128]
  const void* imgC;    // linear_post [128,256]
  const float* bA;
  const float* bB;
  const float* bC;
  const float* gamma;  // ln_2
  const float* beta;
  const float* wmax;
  float* fx1;          // [M,128] saved
  float* z;            // [M,256] saved (pre-GELU)
  float* out;          // [M,128]
  int M;
  float ln_inv_n, ln_npad;
};

template <int LOWP>
__global__ __launch_bounds__(512, 2) void trans_mlp_fwd_kernel(const TmFwdArgs A, int* status) {
  constexpr bool BF = LOWP == 2;
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  gfv_uint4* img = reinterpret_cast<gfv_uint4*>(lds_raw);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const float invw = 1.0f / gfv_pow2_scale(*A.wmax);
  float mabs = 0.f;
  {   // one 128-row block per workgroup (a persistent loop keeps every staging address alive across it: 80 registers)
    const int blk = blockIdx.x;
    const int m = blk * 128 + 16 * wave + li;
    const bool live = m < A.M;
    const size_t mr = (size_t)(live ? m : A.M - 1);
    gfv_f16x8 xh[4], xl[4];
    float inv;
    {
      float h[8][4];
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const float4 t = ld4(A.x + mr * 128 + 16 * nt + 4 * g);
        h[nt][0] = t.x; h[nt][1] = t.y; h[nt][2] = t.z; h[nt][3] = t.w;
      }
      inv = 1.0f / tm_split_row<BF>(h, xh, xl);
    }
    tm_stage<8>(img, A.imgA, tid);
    // ---- to_out + residual -> fx1 (saved); LayerNorm ln_2 ----
    float h[8][4];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const floatx4 acc = tm_mma<4, LOWP>(img, 0, nt, lane, xh, xl);
      const int col = 16 * nt + 4 * g;
      const float4 b = A.bA ? ld4(A.bA + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 r = ld4(A.res + mr * 128 + col);
      h[nt][0] = (acc[0] * inv) * invw + b.x + r.x; h[nt][1] = (acc[1] * inv) * invw + b.y + r.y;
      h[nt][2] = (acc[2] * inv) * invw + b.z + r.z; h[nt][3] = (acc[3] * inv) * invw + b.w + r.w;
      if (live) st4(A.fx1 + mr * 128 + col, h[nt]);
    }
    {
      float sm = 0.f;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) sm += (h[nt][0] + h[nt][1]) + (h[nt][2] + h[nt][3]);
      const float mean = row_sum(sm) * A.ln_inv_n;
      float qq = 0.f;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const float d0 = h[nt][0] - mean, d1 = h[nt][1] - mean, d2 = h[nt][2] - mean, d3 = h[nt][3] - mean;
        qq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
      const float rstd = rsqrtf((row_sum(qq) - A.ln_npad * (mean * mean)) * A.ln_inv_n + 1e-5f);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const float4 ga = ld4(A.gamma + 16 * nt + 4 * g), be = ld4(A.beta + 16 * nt + 4 * g);
        h[nt][0] = (h[nt][0] - mean) * rstd * ga.x + be.x; h[nt][1] = (h[nt][1] - mean) * rstd * ga.y + be.y;
        h[nt][2] = (h[nt][2] - mean) * rstd * ga.z + be.z; h[nt][3] = (h[nt][3] - mean) * rstd * ga.w + be.w;
      }
    }
    inv = 1.0f / tm_split_row<BF>(h, xh, xl);
    tm_stage<16>(img, A.imgB, tid);
    // ---- linear_pre -> z (saved); gelu(z) -> the 256-deep input of linear_post, split behind the fixed scale ----
    gfv_f16x8 yh[8], yl[8];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int tp = 0; tp < 4; ++tp) {
        float e[8];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int nt = 2 * tp + hh;
          const floatx4 acc = tm_mma<4, LOWP>(img, p, nt, lane, xh, xl);
          const int col = 128 * p + 16 * nt + 4 * g;
          const float4 b = A.bB ? ld4(A.bB + col) : make_float4(0.f, 0.f, 0.f, 0.f);
          float zz[4] = {(acc[0] * inv) * invw + b.x, (acc[1] * inv) * invw + b.y, (acc[2] * inv) * invw + b.z, (acc[3] * inv) * invw + b.w};
          if (live) st4(A.z + mr * 256 + col, zz);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a = gfv_gelu(zz[r]);
            mabs = fmaxf(mabs, live ? fabsf(a) : 0.f);
            e[4 * hh + r] = a * TM_SH;
          }
        }
        gfv_uint4 hi, lo;
        gfv_split8_t<BF>(e, hi, lo);
        yh[4 * p + tp] = __builtin_bit_cast(gfv_f16x8, hi);
        yl[4 * p + tp] = __builtin_bit_cast(gfv_f16x8, lo);
      }
    tm_stage<16>(img, A.imgC, tid);
    // ---- linear_post + residual fx1 (re-read: this lane's own rows) ----
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const floatx4 acc = tm_mma<8, LOWP>(img, 0, nt, lane, yh, yl);
      const int col = 16 * nt + 4 * g;
      const float4 b = A.bC ? ld4(A.bC + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 r = ld4(A.fx1 + mr * 128 + col);
      float o[4] = {(acc[0] * TM_SH_INV) * invw + b.x + r.x, (acc[1] * TM_SH_INV) * invw + b.y + r.y,
                    (acc[2] * TM_SH_INV) * invw + b.z + r.z, (acc[3] * TM_SH_INV) * invw + b.w + r.w};
      if (live) st4(A.out + mr * 128 + col, o);
    }
  }
  if (mabs > TM_SH_LIMIT) atomicOr(status, 2);   // GFV_FLAG_CHAIN_RANGE
}

struct TmBwdArgs {
  const float* g;        // gradient of the block's output [M,128]
  const float* g_add;    // optional addend (same rows)
  float* g_sum;          // optional [M,128]: g + g_add
  const float* z;        // [M,256] saved pre-GELU
  const float* fx1;      // [M,128] the LayerNorm's input rows
  const void* imgPt;     // linear_post^T [256,128]
  const void* imgQt;     // linear_pre^T  [128,256]
  const void* imgOt;     // to_out^T      [128,128]
  const float* gamma;    // ln_2 weight
  const float* wmax;
  float* g_z;            // [M,256]
  float* g_fx1;          // [M,128]
  float* g_out_x;        // [M,128]
  float* ln_partial;     // [n_tiles, 2, 128]
  float* gscale;         // per-16-row scales of g (+ g_add): slot 0 of gfv_rowtile_args_t.gscale
  int M, n_tiles;
  float ln_inv_n, ln_npad;
};

template <int LOWP>
__global__ __launch_bounds__(512, 2) void trans_mlp_bwd_kernel(const TmBwdArgs A, int* status) {
  constexpr bool BF = LOWP == 2;
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  gfv_uint4* img = reinterpret_cast<gfv_uint4*>(lds_raw);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const float invw = 1.0f / gfv_pow2_scale(*A.wmax);
  {   // one 128-row block per workgroup (a persistent loop keeps every staging address alive across it: 80 registers)
    const int blk = blockIdx.x;
    const int m = blk * 128 + 16 * wave + li;
    const bool live = m < A.M;
    const size_t mr = (size_t)(live ? m : A.M - 1);
    gfv_f16x8 xh[4], xl[4];
    float inv;
    {
      float h[8][4];
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const float4 t = ld4(A.g + mr * 128 + 16 * nt + 4 * g);
        h[nt][0] = t.x; h[nt][1] = t.y; h[nt][2] = t.z; h[nt][3] = t.w;
      }
      if (A.g_add) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          const float4 t = ld4(A.g_add + mr * 128 + 16 * nt + 4 * g);
          h[nt][0] += t.x; h[nt][1] += t.y; h[nt][2] += t.z; h[nt][3] += t.w;
        }
      }
      if (A.g_sum && live) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) st4(A.g_sum + mr * 128 + 16 * nt + 4 * g, h[nt]);
      }
      const float sx = tm_split_row<BF>(h, xh, xl);
      if (A.gscale) {
        const float sg = gfv_row16_min(sx);
        if (lane == 0) A.gscale[blk * 8 + wave] = sg;
      }
      inv = 1.0f / sx;
    }
    tm_stage<16>(img, A.imgPt, tid);
    // ---- g_z = (g W_post) x gelu'(z) -> saved; its 256 columns -> row scale -> fragments ----
    gfv_f16x8 yh[8], yl[8];
    float inv2;
    {
      float gz[16][4];
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          const floatx4 acc = tm_mma<4, LOWP>(img, p, nt, lane, xh, xl);
          const int col = 128 * p + 16 * nt + 4 * g;
          const float4 zz = ld4(A.z + mr * 256 + col);
          float (&v)[4] = gz[8 * p + nt];
          v[0] = ((acc[0] * inv) * invw) * gfv_dgelu(zz.x); v[1] = ((acc[1] * inv) * invw) * gfv_dgelu(zz.y);
          v[2] = ((acc[2] * inv) * invw) * gfv_dgelu(zz.z); v[3] = ((acc[3] * inv) * invw) * gfv_dgelu(zz.w);
          if (live) st4(A.g_z + mr * 256 + col, v);
        }
      float m0 = 0.f, m1 = 0.f;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        m0 = max3_abs(m0, gz[t][0], gz[t][1]);
        m1 = max3_abs(m1, gz[t][2], gz[t][3]);
      }
      const float sx = gfv_pow2_scale(row_max4(fmaxf(m0, m1)));
      inv2 = 1.0f / sx;
#pragma unroll
      for (int T = 0; T < 8; ++T) {
        float e[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { e[r] = gz[2 * T][r] * sx; e[4 + r] = gz[2 * T + 1][r] * sx; }
        gfv_uint4 hi, lo;
        gfv_split8_t<BF>(e, hi, lo);
        yh[T] = __builtin_bit_cast(gfv_f16x8, hi);
        yl[T] = __builtin_bit_cast(gfv_f16x8, lo);
      }
    }
    tm_stage<16>(img, A.imgQt, tid);
    // ---- v = g_z W_pre; LayerNorm backward of the row (tchain_kernel.h ln_bwd / lin1_lnbwd_kernel: the same sums) + g ----
    float h[8][4];   // v gamma, then (in place) the row's result
    {
      float (&vv)[8][4] = h;
      float y[8][4], dgam[8][4], dbet[8][4];
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const float4 t = ld4(A.fx1 + mr * 128 + 16 * nt + 4 * g);
        y[nt][0] = t.x; y[nt][1] = t.y; y[nt][2] = t.z; y[nt][3] = t.w;
      }
      float sm = 0.f;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) sm += (y[nt][0] + y[nt][1]) + (y[nt][2] + y[nt][3]);
      const float mean = row_sum(sm) * A.ln_inv_n;
      float qq = 0.f;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const float d0 = y[nt][0] - mean, d1 = y[nt][1] - mean, d2 = y[nt][2] - mean, d3 = y[nt][3] - mean;
        qq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
      const float rstd = rsqrtf((row_sum(qq) - A.ln_npad * (mean * mean)) * A.ln_inv_n + 1e-5f);
      const float livef = live ? 1.0f : 0.0f;   // rows past M must not reach the (dgamma, dbeta) sums
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const floatx4 acc = tm_mma<8, LOWP>(img, 0, nt, lane, yh, yl);
        const float4 ga = ld4(A.gamma + 16 * nt + 4 * g);
        const float gv[4] = {ga.x, ga.y, ga.z, ga.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = ((acc[r] * inv2) * invw) * livef;
          const float xhat = (y[nt][r] - mean) * rstd;
          dgam[nt][r] = v * xhat;
          dbet[nt][r] = v;
          vv[nt][r] = v * gv[r];
          s1 += vv[nt][r];
          s2 += vv[nt][r] * xhat;
        }
      }
      const float mm1 = row_sum(s1) * A.ln_inv_n, mm2 = row_sum(s2) * A.ln_inv_n;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const int col = 16 * nt + 4 * g;
        // the residual branch: + g (+ g_add).  (Re-read: the rows this lane loaded / saved at the top)
        float4 rr = ld4(A.g + mr * 128 + col);
        if (A.g_add) {
          const float4 t = ld4(A.g_add + mr * 128 + col);
          rr.x += t.x; rr.y += t.y; rr.z += t.z; rr.w += t.w;
        }
        const float rv[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) h[nt][r] = rstd * (vv[nt][r] - mm1 - ((y[nt][r] - mean) * rstd) * mm2) + rv[r];
        if (live) st4(A.g_fx1 + mr * 128 + col, h[nt]);
      }
      // (dgamma, dbeta): over the wave's 16 rows by DPP, over the four waves of a 64-row tile through LDS (the image is done)
      __syncthreads();
      float* red = reinterpret_cast<float*>(lds_raw);   // [8 waves][2][128]
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float dg = gfv_row16_sum(dgam[nt][r]), db = gfv_row16_sum(dbet[nt][r]);
          if (li == 0) {
            red[(wave * 2 + 0) * 128 + 16 * nt + 4 * g + r] = dg;
            red[(wave * 2 + 1) * 128 + 16 * nt + 4 * g + r] = db;
          }
        }
      __syncthreads();
      {
        const int half = tid >> 8, jj = tid & 255;   // tile 2 b + half; jj: dgamma 0..127 | dbeta 128..255
        const int tile = 2 * blk + half;
        if (tile < A.n_tiles && A.ln_partial) {
          const int w0 = 4 * half, which = jj >> 7, c = jj & 127;
          const float s = (red[((w0 + 0) * 2 + which) * 128 + c] + red[((w0 + 1) * 2 + which) * 128 + c]) +
                          (red[((w0 + 2) * 2 + which) * 128 + c] + red[((w0 + 3) * 2 + which) * 128 + c]);
          A.ln_partial[(size_t)tile * 256 + jj] = s;
        }
      }
    }
    inv = 1.0f / tm_split_row<BF>(h, xh, xl);
    tm_stage<8>(img, A.imgOt, tid);   // (its first barrier also ends the reads of `red`)
    // ---- g_out_x = g_fx1 W_out ----
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const floatx4 acc = tm_mma<4, LOWP>(img, 0, nt, lane, xh, xl);
      const float o[4] = {(acc[0] * inv) * invw, (acc[1] * inv) * invw, (acc[2] * inv) * invw, (acc[3] * inv) * invw};
      if (live) st4(A.g_out_x + mr * 128 + 16 * nt + 4 * g, o);
    }
  }
  (void)status;
}

inline bool tm_al16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }
bool tm_dyn_lds(const void* fn, std::atomic<unsigned long long>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_relaxed) & bit) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) != hipSuccess) return false;
  done.fetch_or(bit);
  return true;
}
void tm_ln(float& inv_n, float& npad) {
  const int hs = gfv_hidden_size();
  const int h = (hs > 0 && hs < 128) ? hs : 128;
  inv_n = 1.0f / (float)h;
  npad = (float)(128 - h);
}
}  // namespace

bool gfv_internal_wimg_form_ok(const float* wmax);   // wimg.hip
int gfv_internal_ctrans_fwd_try(const gfv_trans_mlp_t* a, int form, hipStream_t stream);   // ctrans.hip: the small-tile forms
int gfv_internal_ctrans_bwd_try(const gfv_trans_mlp_bwd_t* a, int form, hipStream_t stream);

extern "C" int gfv_trans_mlp_fwd(const gfv_trans_mlp_t* a, void* stream) {
  if (!a || a->M < 0) return GFV_ERR_ARG;
  if (a->M == 0) return GFV_OK;
  const int form = gfv_f16split_enabled();
  if (form == 0) return GFV_ERR_ARG;   // split-fp16 forms only: the caller keeps the three single-layer launches in the fp32-MFMA form
  const void* ptrs[] = {a->x, a->res, a->img_out, a->img_pre, a->img_post, a->gamma, a->beta, a->wmax, a->fx1, a->z, a->out};
  for (const void* p : ptrs)
    if (!p || !tm_al16(p)) return GFV_ERR_ARG;
  if (!gfv_internal_wimg_form_ok(a->wmax)) return GFV_ERR_ARG;   // (images of the other class of product form: wimg.hip)
  if ((a->b_out && !tm_al16(a->b_out)) || (a->b_pre && !tm_al16(a->b_pre)) || (a->b_post && !tm_al16(a->b_post))) return GFV_ERR_ARG;
  TmFwdArgs B{};
  B.x = a->x; B.res = a->res; B.imgA = a->img_out; B.imgB = a->img_pre; B.imgC = a->img_post;
  B.bA = a->b_out; B.bB = a->b_pre; B.bC = a->b_post; B.gamma = a->gamma; B.beta = a->beta; B.wmax = a->wmax;
  B.fx1 = a->fx1; B.z = a->z; B.out = a->out; B.M = a->M;
  tm_ln(B.ln_inv_n, B.ln_npad);
  // three Linear layers' flops; rows read: out_x, fx_in, fx1 (re-read); written: fx1, z (256 wide), out
  GfvProfScope ps_(GFV_K_LIN1, 2.0 * a->M * (128.0 * 128 + 2 * 128.0 * 256), 4.0 * a->M * (3 * 128.0 + 128 + 256 + 128), stream);
  if (gfv_internal_ctrans_fwd_try(a, form, (hipStream_t)stream)) {   // short launches: one 32-row tile per workgroup
    GFV_CHECK_LAUNCH();
    return GFV_OK;
  }
  int* st = gfv_internal_status_ptr();
  const int nblk = (a->M + 127) / 128;
  const dim3 grid(nblk), blk(512);
#define TM_F(LP)                                                                                          \
  do {                                                                                                    \
    static std::atomic<unsigned long long> done{0};                                                       \
    if (!tm_dyn_lds(reinterpret_cast<const void*>(&trans_mlp_fwd_kernel<LP>), done)) return GFV_ERR_LAUNCH; \
    GFV_LAUNCH((trans_mlp_fwd_kernel<LP>), grid, blk, 131072, (hipStream_t)stream, B, st);               \
  } while (0)
  if (form == 3) TM_F(2);
  else if (form == 2) TM_F(1);
  else TM_F(0);
#undef TM_F
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_trans_mlp_bwd(const gfv_trans_mlp_bwd_t* a, void* stream) {
  if (!a || a->M < 0) return GFV_ERR_ARG;
  if (a->M == 0) return GFV_OK;
  const int form = gfv_f16split_enabled();
  if (form == 0) return GFV_ERR_ARG;
  const void* ptrs[] = {a->g, a->z, a->fx1, a->img_post_t, a->img_pre_t, a->img_out_t, a->gamma, a->wmax, a->g_z, a->g_fx1, a->g_out_x};
  for (const void* p : ptrs)
    if (!p || !tm_al16(p)) return GFV_ERR_ARG;
  if (!gfv_internal_wimg_form_ok(a->wmax)) return GFV_ERR_ARG;
  if ((a->g_add && !tm_al16(a->g_add)) || (a->g_sum && !tm_al16(a->g_sum)) || (a->ln_partial && !tm_al16(a->ln_partial))) return GFV_ERR_ARG;
  TmBwdArgs B{};
  B.g = a->g; B.g_add = a->g_add; B.g_sum = a->g_sum; B.z = a->z; B.fx1 = a->fx1;
  B.imgPt = a->img_post_t; B.imgQt = a->img_pre_t; B.imgOt = a->img_out_t; B.gamma = a->gamma; B.wmax = a->wmax;
  B.g_z = a->g_z; B.g_fx1 = a->g_fx1; B.g_out_x = a->g_out_x; B.ln_partial = a->ln_partial; B.gscale = a->gscale;
  B.M = a->M; B.n_tiles = (a->M + 63) / 64;
  tm_ln(B.ln_inv_n, B.ln_npad);
  GfvProfScope ps_(GFV_K_LIN1, 2.0 * a->M * (128.0 * 128 + 2 * 128.0 * 256), 4.0 * a->M * (128.0 + 256 + 128 + 256 + 128 + 128 + 128), stream);
  if (gfv_internal_ctrans_bwd_try(a, form, (hipStream_t)stream)) {   // short launches: one 32-row tile per workgroup, ln_partial per 32 rows
    GFV_CHECK_LAUNCH();
    return GFV_OK;
  }
  int* st = gfv_internal_status_ptr();
  const int nblk = (a->M + 127) / 128;
  const dim3 grid(nblk), blk(512);
#define TM_B(LP)                                                                                          \
  do {                                                                                                    \
    static std::atomic<unsigned long long> done{0};                                                       \
    if (!tm_dyn_lds(reinterpret_cast<const void*>(&trans_mlp_bwd_kernel<LP>), done)) return GFV_ERR_LAUNCH; \
    GFV_LAUNCH((trans_mlp_bwd_kernel<LP>), grid, blk, 131072, (hipStream_t)stream, B, st);               \
  } while (0)
  if (form == 3) TM_B(2);
  else if (form == 2) TM_B(1);
  else TM_B(0);
#undef TM_B
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}
