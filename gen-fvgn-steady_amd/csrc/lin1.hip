// gfv-build-flags: -fno-slp-vectorize
// Lean single-layer launches of gfv_rowtile_chain in the split-fp16 form (round 3).
//
// Sixteen launches of a training step are ONE Linear over the N node rows with a light prologue / epilogue - the EdgeBlock's
// node-level projection (blocks.py:54 factored through the nodes) and five of the eight launches of a Transolver block
// (GraphTransolver.py:51-62,93-95,163-169: in_project_fx | in_project_x, to_out, linear_post and the adjoints of to_out and
// of the two input projections).  In the general chain kernel (tchain_kernel.h) such a launch costs one tile life of the
// three-layer machinery - 15 - 26 us for 13 - 26 MB: the weight image streams through LDS in four barrier-separated slices
// per layer, every workgroup of 64 rows pulls the whole 64 KB image for 32 KB of activations.  Here a workgroup takes 128
// rows (8 waves x 16), stages the layer's image in LDS ONCE (64 KB, or 128 KB for a 256-wide input or output) while its
// waves load and split their rows, and runs the products - the same three f16 MFMAs per fragment pair on the same (hi, lo)
// operands as the chain (gfv_split.h, wimg.hip's fragment order), so the results carry the chain's accuracy.
//   out_p[m, :] = sum_s op_in(seg_s[m, :] (+ in_add[m, :])) W_p,s^T + bias_p (+ res_p[m, :]),   p < N / 128, s < K / 128
// Covered: 128-wide segments without gather, in_op none / GELU / LayerNorm, in_add + in_save on a one-segment input (+ the
// per-16-row scales of those rows for the weight-gradient launch: gscale slot 0), bias / bias2, the GELU' epilogue with its
// saved pre-activations, residual addends, 128 or 256 outputs (256 inputs with 128 outputs).  Anything else stays with the
// chain kernel.
#include <atomic>
#include <cstdlib>

#include "../../include/gfv.h"
#include "gfv_common.h"
#include "gfv_split.h"

int* gfv_internal_status_ptr();
extern "C" int gfv_hidden_size(void);

namespace {

__device__ __forceinline__ float l1_max3_abs(float m, float a, float b) {
  float r;
  asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float l1_row_max4(float v) {   // over the 4 lanes (g = 0..3) that share a row
  float a, b;
  gfv_lane_xor16(v, a, b);
  v = fmaxf(a, b);
  gfv_lane_xor32(v, a, b);
  return fmaxf(a, b);
}

struct Lin1Args {
  const float* seg[2];
  int seg_ld[2];
  const float* in_add;
  float* in_save;
  const void* img;       // [pass][T][nt][hi 64 lanes | lo 64 lanes] x 16 B (gfv_weight_images)
  const float* wmax;
  const float* bias[2];
  const float* res[2];
  int res_ld[2];
  float* out[2];
  int out_ld[2];
  int M;
  const float* gamma;    // IN_OP 2 (LayerNorm prologue)
  const float* beta;
  float ln_inv_n, ln_npad;   // 1 / hidden, 128 - hidden (tchain_kernel.h ln_stats)
  const float* aux;      // DGELU: the saved pre-activations [M, 128 NP]
  float* gscale;         // optional: per-16-row scale of the prologue result (slot 0 of gfv_rowtile_args_t.gscale)
};

// KS: 32-wide k-steps (4: one 128-wide segment, 8: two); NP: 128-row passes of the image = 128-wide output chunks
// IN_OP: 0 none, 1 GELU, 2 LayerNorm (one segment); DGELU: out = (acc) * gelu'(aux)
template <int KS, int NP, int IN_OP, bool DGELU, int LOWP>   // LOWP: 0 three products, 1 / 2 the single-product forms (fp16 / bf16)
__global__ __launch_bounds__(512, 2) void lin1_kernel(const Lin1Args A, int* status) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  gfv_uint4* img = reinterpret_cast<gfv_uint4*>(lds_raw);
  constexpr int FRAGS = NP * KS * 8 * 128;   // 16-byte units of the image
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  // ---- the image goes to LDS (the loads are issued first, the row loads follow: both are in flight together) ----
  {
    const gfv_uint4* src = reinterpret_cast<const gfv_uint4*>(A.img);
    constexpr int PER = FRAGS / 512;   // 8 (64 KB) or 16 (128 KB): all of a thread's loads go out before the first LDS write
    gfv_uint4 t[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) t[u] = src[(size_t)u * 512 + tid];
#pragma unroll
    for (int u = 0; u < PER; ++u) img[u * 512 + tid] = t[u];
  }
  // Row blocks of 128: a workgroup walks blocks blockIdx.x, + gridDim.x, ... (the launcher caps the grid at a few workgroups
  // per CU: round 4).  The image is staged ONCE and never written again, so nothing below needs a barrier but the first block's:
  // the eight waves drift apart and one wave's row loads overlap another's products and stores, where one workgroup per block
  // paid the image (64 - 128 KB from L2) and a serial load -> split -> products -> store life per 128 rows.
  const int nblk = (A.M + 127) >> 7;
  bool first = true;
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
  // ---- this wave's 16 rows: lane (row li, column group g) holds columns 32 T + 4 g + (0..3) and 32 T + 16 + 4 g + (0..3) ----
  const int m = blk * 128 + 16 * wave + li;
  const bool live = m < A.M;
  const size_t mr = (size_t)(live ? m : A.M - 1);
  float v[KS][8];
#pragma unroll
  for (int T = 0; T < KS; ++T) {
    const int s = T >> 2, c = 32 * (T & 3) + 4 * g;
    const float* rp = A.seg[s] + mr * A.seg_ld[s] + c;
    const float4 a = *reinterpret_cast<const float4*>(rp), b = *reinterpret_cast<const float4*>(rp + 16);
    v[T][0] = a.x; v[T][1] = a.y; v[T][2] = a.z; v[T][3] = a.w;
    v[T][4] = b.x; v[T][5] = b.y; v[T][6] = b.z; v[T][7] = b.w;
  }
  if (A.in_add) {   // (one-segment inputs only)
#pragma unroll
    for (int T = 0; T < (KS < 4 ? KS : 4); ++T) {
      const float* rp = A.in_add + mr * A.seg_ld[0] + 32 * T + 4 * g;
      const float4 a = *reinterpret_cast<const float4*>(rp), b = *reinterpret_cast<const float4*>(rp + 16);
      v[T][0] += a.x; v[T][1] += a.y; v[T][2] += a.z; v[T][3] += a.w;
      v[T][4] += b.x; v[T][5] += b.y; v[T][6] += b.z; v[T][7] += b.w;
    }
  }
  if (A.in_save && live) {
#pragma unroll
    for (int T = 0; T < (KS < 4 ? KS : 4); ++T) {
      float* sp = A.in_save + mr * 128 + 32 * T + 4 * g;
      *reinterpret_cast<float4*>(sp) = make_float4(v[T][0], v[T][1], v[T][2], v[T][3]);
      *reinterpret_cast<float4*>(sp + 16) = make_float4(v[T][4], v[T][5], v[T][6], v[T][7]);
    }
  }
  if (IN_OP == 1) {
#pragma unroll
    for (int T = 0; T < KS; ++T)
#pragma unroll
      for (int e = 0; e < 8; ++e) v[T][e] = gfv_gelu(v[T][e]);
  }
  if (IN_OP == 2) {
    // LayerNorm of the row (tchain_kernel.h ln_stats / ln_apply: the same sums in the same order; the lane's 32 columns are
    // 16 t + 4 g + r with t = 2 T (e < 4), 2 T + 1 (e >= 4))
    float sm = 0.f;
#pragma unroll
    for (int T = 0; T < KS; ++T) {
      sm += (v[T][0] + v[T][1]) + (v[T][2] + v[T][3]);
      sm += (v[T][4] + v[T][5]) + (v[T][6] + v[T][7]);
    }
    float a, b;
    gfv_lane_xor16(sm, a, b);
    sm = a + b;
    gfv_lane_xor32(sm, a, b);
    const float mean = (a + b) * A.ln_inv_n;
    float qq = 0.f;
#pragma unroll
    for (int T = 0; T < KS; ++T) {
      const float d0 = v[T][0] - mean, d1 = v[T][1] - mean, d2 = v[T][2] - mean, d3 = v[T][3] - mean;
      qq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      const float d4 = v[T][4] - mean, d5 = v[T][5] - mean, d6 = v[T][6] - mean, d7 = v[T][7] - mean;
      qq += (d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7);
    }
    gfv_lane_xor16(qq, a, b);
    qq = a + b;
    gfv_lane_xor32(qq, a, b);
    const float rstd = rsqrtf(((a + b) - A.ln_npad * (mean * mean)) * A.ln_inv_n + 1e-5f);
#pragma unroll
    for (int T = 0; T < KS; ++T) {
      const float4 ga0 = *reinterpret_cast<const float4*>(A.gamma + 32 * T + 4 * g), ga1 = *reinterpret_cast<const float4*>(A.gamma + 32 * T + 16 + 4 * g);
      const float4 be0 = *reinterpret_cast<const float4*>(A.beta + 32 * T + 4 * g), be1 = *reinterpret_cast<const float4*>(A.beta + 32 * T + 16 + 4 * g);
      v[T][0] = (v[T][0] - mean) * rstd * ga0.x + be0.x; v[T][1] = (v[T][1] - mean) * rstd * ga0.y + be0.y;
      v[T][2] = (v[T][2] - mean) * rstd * ga0.z + be0.z; v[T][3] = (v[T][3] - mean) * rstd * ga0.w + be0.w;
      v[T][4] = (v[T][4] - mean) * rstd * ga1.x + be1.x; v[T][5] = (v[T][5] - mean) * rstd * ga1.y + be1.y;
      v[T][6] = (v[T][6] - mean) * rstd * ga1.z + be1.z; v[T][7] = (v[T][7] - mean) * rstd * ga1.w + be1.w;
    }
  }
  // power-of-two scale of the row, operand split (the chain's row_scale / to_halves)
  float m0 = 0.f, m1 = 0.f;
#pragma unroll
  for (int T = 0; T < KS; ++T) {
    m0 = l1_max3_abs(m0, v[T][0], v[T][1]);
    m1 = l1_max3_abs(m1, v[T][2], v[T][3]);
    m0 = l1_max3_abs(m0, v[T][4], v[T][5]);
    m1 = l1_max3_abs(m1, v[T][6], v[T][7]);
  }
  const float sx = gfv_pow2_scale(l1_row_max4(fmaxf(m0, m1)));
  if (A.gscale) {   // the group's scale = the smallest of its 16 rows' (tchain_kernel.h group_scale_out)
    const float sg = gfv_row16_min(sx);
    if (lane == 0) A.gscale[blk * 8 + wave] = sg;
  }
  gfv_f16x8 xh[KS], xl[KS];
#pragma unroll
  for (int T = 0; T < KS; ++T) {
    float e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = v[T][i] * sx;
    gfv_uint4 hi, lo;
    gfv_split8_t<LOWP == 2>(e, hi, lo);
    xh[T] = __builtin_bit_cast(gfv_f16x8, hi);
    xl[T] = __builtin_bit_cast(gfv_f16x8, lo);
  }
  const float inv = (1.0f / sx), invw = 1.0f / gfv_pow2_scale(*A.wmax);
  if (first) __syncthreads();   // the image is in LDS
  first = false;
  // ---- products: D[n = 16 nt + 4 g + r][row li] ----
  // The rows of the epilogue's operands (the saved pre-activations of a GELU' epilogue, the residual) are requested a GROUP of
  // four n-tiles ahead of their use: loaded where they are used - the first form of this loop - every n-tile ended in a
  // dependent global round trip (two waves per SIMD do not cover it), and the GELU' launches of the Transolver's linear_post
  // adjoint took 3.5 x the time of their plain siblings (351 against 101 us at 8 meshes per GPU, profiles/r04_ab_lin1_prefetch.txt).
  constexpr int NGRP = 2 * NP;   // groups of four n-tiles over all passes
  float4 zq[2][4], rq[2][4];
  auto fetch = [&](int grp, int buf) {
    const int p = grp >> 1, nt0 = 4 * (grp & 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int col = 16 * (nt0 + k) + 4 * g;
      if (DGELU) zq[buf][k] = *reinterpret_cast<const float4*>(A.aux + mr * (128 * NP) + 128 * p + col);
      if (A.res[p]) rq[buf][k] = *reinterpret_cast<const float4*>(A.res[p] + mr * A.res_ld[p] + col);
    }
  };
  fetch(0, 0);
#pragma unroll
  for (int grp = 0; grp < NGRP; ++grp) {
    const int p = grp >> 1, nt0 = 4 * (grp & 1), buf = grp & 1;
    float* outp = A.out[p];
    const float* resp = A.res[p];
    const float* bp = A.bias[p];
    if (grp + 1 < NGRP) fetch(grp + 1, buf ^ 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int nt = nt0 + k;
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
      const int col = 16 * nt + 4 * g;
      float4 o = make_float4((acc[0] * inv) * invw, (acc[1] * inv) * invw, (acc[2] * inv) * invw, (acc[3] * inv) * invw);
      if (bp) {
        const float4 b = *reinterpret_cast<const float4*>(bp + col);
        o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
      }
      if (DGELU) {
        const float4 z = zq[buf][k];
        o.x *= gfv_dgelu(z.x); o.y *= gfv_dgelu(z.y); o.z *= gfv_dgelu(z.z); o.w *= gfv_dgelu(z.w);
      }
      if (resp) {
        const float4 r = rq[buf][k];
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
      if (live) *reinterpret_cast<float4*>(outp + mr * A.out_ld[p] + col) = o;
    }
  }
  }   // row blocks
  (void)status;
}

// The adjoint of linear_pre behind LayerNorm ln_2 (GraphTransolver.py:163-166): out = LNbwd(g_z [M,256] W^T; fx1, gamma) + res, with
// the per-64-row-tile (dgamma, dbeta) partials the chain kernel leaves for the reduction launch (gfv_rowtile_args_t.ln_partial).
struct Lin1LnbArgs {
  Lin1Args a;
  const float* y;        // fin_aux: the LayerNorm's input rows [M,128]
  const float* fgamma;   // fin_gamma
  float* ln_partial;     // [n_tiles, 2, 128]
  int n_tiles;
};
template <int LOWP>
__global__ __launch_bounds__(512, 2) void lin1_lnbwd_kernel(const Lin1LnbArgs B, int* status) {
  constexpr int KS = 8;
  const Lin1Args& A = B.a;
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  gfv_uint4* img = reinterpret_cast<gfv_uint4*>(lds_raw);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  {
    const gfv_uint4* src = reinterpret_cast<const gfv_uint4*>(A.img);
    gfv_uint4 t[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) t[u] = src[(size_t)u * 512 + tid];
#pragma unroll
    for (int u = 0; u < 16; ++u) img[u * 512 + tid] = t[u];
  }
  const int m = blockIdx.x * 128 + 16 * wave + li;
  const bool live = m < A.M;
  const size_t mr = (size_t)(live ? m : A.M - 1);
  gfv_f16x8 xh[KS], xl[KS];
  float inv;
  {
    float v[KS][8];
#pragma unroll
    for (int T = 0; T < KS; ++T) {
      const int s = T >> 2, c = 32 * (T & 3) + 4 * g;
      const float* rp = A.seg[s] + mr * A.seg_ld[s] + c;
      const float4 a = *reinterpret_cast<const float4*>(rp), b = *reinterpret_cast<const float4*>(rp + 16);
      v[T][0] = a.x; v[T][1] = a.y; v[T][2] = a.z; v[T][3] = a.w;
      v[T][4] = b.x; v[T][5] = b.y; v[T][6] = b.z; v[T][7] = b.w;
    }
    float m0 = 0.f, m1 = 0.f;
#pragma unroll
    for (int T = 0; T < KS; ++T) {
      m0 = l1_max3_abs(m0, v[T][0], v[T][1]);
      m1 = l1_max3_abs(m1, v[T][2], v[T][3]);
      m0 = l1_max3_abs(m0, v[T][4], v[T][5]);
      m1 = l1_max3_abs(m1, v[T][6], v[T][7]);
    }
    const float sx = gfv_pow2_scale(l1_row_max4(fmaxf(m0, m1)));
    inv = 1.0f / sx;
#pragma unroll
    for (int T = 0; T < KS; ++T) {
      float e[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) e[i] = v[T][i] * sx;
      gfv_uint4 hi, lo;
      gfv_split8_t<LOWP == 2>(e, hi, lo);
      xh[T] = __builtin_bit_cast(gfv_f16x8, hi);
      xl[T] = __builtin_bit_cast(gfv_f16x8, lo);
    }
  }
  const float invw = 1.0f / gfv_pow2_scale(*A.wmax);
  // the LayerNorm input rows: in flight through the products
  float y[8][4];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    const float4 t = *reinterpret_cast<const float4*>(B.y + mr * 128 + 16 * nt + 4 * g);
    y[nt][0] = t.x; y[nt][1] = t.y; y[nt][2] = t.z; y[nt][3] = t.w;
  }
  __syncthreads();
  floatx4 acc[8];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    acc[nt] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int T = 0; T < KS; ++T) {
      const gfv_uint4* f = img + (T * 8 + nt) * 128 + lane;
      const gfv_f16x8 wh = __builtin_bit_cast(gfv_f16x8, f[0]);
      if (!LOWP) {
        const gfv_f16x8 wl = __builtin_bit_cast(gfv_f16x8, f[64]);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[T], acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[T], acc[nt], 0, 0, 0);
      }
      acc[nt] = gfv_mma_hh<LOWP == 2>(wh, xh[T], acc[nt]);
    }
  }
  // ---- LayerNorm backward of the row (tchain_kernel.h ln_stats / ln_bwd) ----
  float a, b;
  float sm = 0.f;
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) sm += (y[nt][0] + y[nt][1]) + (y[nt][2] + y[nt][3]);
  gfv_lane_xor16(sm, a, b);
  sm = a + b;
  gfv_lane_xor32(sm, a, b);
  const float mean = (a + b) * A.ln_inv_n;
  float qq = 0.f;
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    const float d0 = y[nt][0] - mean, d1 = y[nt][1] - mean, d2 = y[nt][2] - mean, d3 = y[nt][3] - mean;
    qq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  gfv_lane_xor16(qq, a, b);
  qq = a + b;
  gfv_lane_xor32(qq, a, b);
  const float rstd = rsqrtf(((a + b) - A.ln_npad * (mean * mean)) * A.ln_inv_n + 1e-5f);
  const float livef = live ? 1.0f : 0.0f;   // rows past M must not reach the (dgamma, dbeta) sums
  float vv[8][4], dgam[8][4], dbet[8][4];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    const float4 ga = *reinterpret_cast<const float4*>(B.fgamma + 16 * nt + 4 * g);
    const float gv[4] = {ga.x, ga.y, ga.z, ga.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = ((acc[nt][r] * inv) * invw) * livef;
      const float xhat = (y[nt][r] - mean) * rstd;
      dgam[nt][r] = v * xhat;
      dbet[nt][r] = v;
      vv[nt][r] = v * gv[r];
      s1 += vv[nt][r];
      s2 += vv[nt][r] * xhat;
    }
  }
  gfv_lane_xor16(s1, a, b);
  s1 = a + b;
  gfv_lane_xor32(s1, a, b);
  const float mm1 = (a + b) * A.ln_inv_n;
  gfv_lane_xor16(s2, a, b);
  s2 = a + b;
  gfv_lane_xor32(s2, a, b);
  const float mm2 = (a + b) * A.ln_inv_n;
  const float* resp = A.res[0];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    const int col = 16 * nt + 4 * g;
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = rstd * (vv[nt][r] - mm1 - ((y[nt][r] - mean) * rstd) * mm2);
    if (resp) {
      const float4 rr = *reinterpret_cast<const float4*>(resp + mr * A.res_ld[0] + col);
      o[0] += rr.x; o[1] += rr.y; o[2] += rr.z; o[3] += rr.w;
    }
    if (live) *reinterpret_cast<float4*>(A.out[0] + mr * A.out_ld[0] + col) = make_float4(o[0], o[1], o[2], o[3]);
  }
  // ---- (dgamma, dbeta): over the wave's 16 rows by DPP, over the four waves of a 64-row tile through LDS (the image is done) ----
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
    const int half = tid >> 8, j = tid & 255;   // tile 2 b + half; j: dgamma 0..127 | dbeta 128..255
    const int tile = 2 * blockIdx.x + half;
    if (tile < B.n_tiles && B.ln_partial) {
      const int w0 = 4 * half, which = j >> 7, c = j & 127;
      const float s = (red[((w0 + 0) * 2 + which) * 128 + c] + red[((w0 + 1) * 2 + which) * 128 + c]) +
                      (red[((w0 + 2) * 2 + which) * 128 + c] + red[((w0 + 3) * 2 + which) * 128 + c]);
      B.ln_partial[(size_t)tile * 256 + j] = s;
    }
  }
  (void)status;
}

// Two segmented-sum segments (gfv_seg_t.csr_rowptr: the per-side scatter of the factored EdgeBlock's adjoint, blocks.py:24-31 adjoint)
// in front of one Linear [256 -> 128]: row m of segment s = sum of the rows col_s[rowptr_s[m] .. rowptr_s[m+1]) of src_s, entries
// added in CSR order, two neighbour rows in flight per lane (the chain kernel's prologue, tchain_kernel.h load_segment); the
// assembled rows are written out for the weight-gradient launch (gfv_seg_t.save).
struct Lin1CsrArgs {
  const float* src[2];
  int src_ld[2];
  const int* rowptr[2];
  const int* col[2];
  float* save[2];
  const void* img;
  const float* wmax;
  float* out;
  int out_ld;
  int M;
};
template <int LOWP>
__global__ __launch_bounds__(512, 2) void lin1_csr_kernel(const Lin1CsrArgs A, int* status) {
  constexpr int KS = 8;
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  gfv_uint4* img = reinterpret_cast<gfv_uint4*>(lds_raw);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  {
    const gfv_uint4* src = reinterpret_cast<const gfv_uint4*>(A.img);
    gfv_uint4 t[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) t[u] = src[(size_t)u * 512 + tid];
#pragma unroll
    for (int u = 0; u < 16; ++u) img[u * 512 + tid] = t[u];
  }
  // XCD-aware order of the 128-row workgroups (consecutive row blocks share neighbour rows in the same L2).  (A persistent form -
  // one workgroup per CU walking a range of blocks with the image staged once, as lin1_kernel does - was measured in round 4:
  // 22.74 against 22.67 ms per step of 8 meshes, no gain: the gathers, not the image, are this kernel's time.)
  const int nwg = gridDim.x;
  const int wg = (nwg & 7) == 0 ? (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int m = wg * 128 + 16 * wave + li;
  const bool live = m < A.M;
  const int mc = live ? m : A.M - 1;
  float v[KS][8];
#pragma unroll
  for (int T = 0; T < KS; ++T)
#pragma unroll
    for (int e = 0; e < 8; ++e) v[T][e] = 0.f;
  // The two segments are walked TOGETHER, two neighbour rows of each in flight (four rows = 32 loads of 16 B per lane): walked
  // one after the other - the first form of this loop - a workgroup (alone on its CU: the image takes 128 KB of LDS) went through
  // four dependent memory round trips where this takes two.  Entries are added in CSR order per segment, as before.
  int kk[2], ee[2], cn0[2], cn1[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    kk[s] = A.rowptr[s][mc];
    ee[s] = live ? A.rowptr[s][mc + 1] : kk[s];
    cn0[s] = kk[s] < ee[s] ? A.col[s][kk[s]] : 0;
    cn1[s] = kk[s] + 1 < ee[s] ? A.col[s][kk[s] + 1] : cn0[s];
  }
  while (kk[0] < ee[0] || kk[1] < ee[1]) {
    float4 a0[2][8], a1[2][8];
    bool one[2], two[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int k = kk[s], end = ee[s];
      one[s] = k < end;
      two[s] = k + 1 < end;
      const int c0 = cn0[s], c1 = cn1[s];
      if (k + 2 < end) {
        cn0[s] = A.col[s][k + 2];
        cn1[s] = A.col[s][k + 3 < end ? k + 3 : k + 2];
      }
      const float* p0 = A.src[s] + (size_t)c0 * (size_t)A.src_ld[s] + 4 * g;
      const float* p1 = A.src[s] + (size_t)c1 * (size_t)A.src_ld[s] + 4 * g;
      // (a finished or empty segment issues no loads: its source table may have no rows at all)
      if (one[s]) {
#pragma unroll
        for (int t = 0; t < 8; ++t) a0[s][t] = *reinterpret_cast<const float4*>(p0 + 16 * t);
      }
      if (two[s]) {
#pragma unroll
        for (int t = 0; t < 8; ++t) a1[s][t] = *reinterpret_cast<const float4*>(p1 + 16 * t);
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        float* d = &v[4 * s + (t >> 1)][4 * (t & 1)];
        if (one[s]) { d[0] += a0[s][t].x; d[1] += a0[s][t].y; d[2] += a0[s][t].z; d[3] += a0[s][t].w; }
        if (two[s]) { d[0] += a1[s][t].x; d[1] += a1[s][t].y; d[2] += a1[s][t].z; d[3] += a1[s][t].w; }
      }
      kk[s] = min(kk[s] + 2, ee[s]);
    }
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (A.save[s] && live) {
#pragma unroll
      for (int T = 0; T < 4; ++T) {
        float* sp = A.save[s] + (size_t)m * 128 + 32 * T + 4 * g;
        *reinterpret_cast<float4*>(sp) = make_float4(v[4 * s + T][0], v[4 * s + T][1], v[4 * s + T][2], v[4 * s + T][3]);
        *reinterpret_cast<float4*>(sp + 16) = make_float4(v[4 * s + T][4], v[4 * s + T][5], v[4 * s + T][6], v[4 * s + T][7]);
      }
    }
  }
  float m0 = 0.f, m1 = 0.f;
#pragma unroll
  for (int T = 0; T < KS; ++T) {
    m0 = l1_max3_abs(m0, v[T][0], v[T][1]);
    m1 = l1_max3_abs(m1, v[T][2], v[T][3]);
    m0 = l1_max3_abs(m0, v[T][4], v[T][5]);
    m1 = l1_max3_abs(m1, v[T][6], v[T][7]);
  }
  const float sx = gfv_pow2_scale(l1_row_max4(fmaxf(m0, m1)));
  gfv_f16x8 xh[KS], xl[KS];
#pragma unroll
  for (int T = 0; T < KS; ++T) {
    float e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = v[T][i] * sx;
    gfv_uint4 hi, lo;
    gfv_split8_t<LOWP == 2>(e, hi, lo);
    xh[T] = __builtin_bit_cast(gfv_f16x8, hi);
    xl[T] = __builtin_bit_cast(gfv_f16x8, lo);
  }
  const float inv = 1.0f / sx, invw = 1.0f / gfv_pow2_scale(*A.wmax);
  __syncthreads();
#pragma unroll 2
  for (int nt = 0; nt < 8; ++nt) {
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int T = 0; T < KS; ++T) {
      const gfv_uint4* f = img + (T * 8 + nt) * 128 + lane;
      const gfv_f16x8 wh = __builtin_bit_cast(gfv_f16x8, f[0]);
      if (!LOWP) {
        const gfv_f16x8 wl = __builtin_bit_cast(gfv_f16x8, f[64]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[T], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[T], acc, 0, 0, 0);
      }
      acc = gfv_mma_hh<LOWP == 2>(wh, xh[T], acc);
    }
    if (live)
      *reinterpret_cast<float4*>(A.out + (size_t)m * A.out_ld + 16 * nt + 4 * g) =
          make_float4((acc[0] * inv) * invw, (acc[1] * inv) * invw, (acc[2] * inv) * invw, (acc[3] * inv) * invw);
  }
  (void)status;
}

inline bool al16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }
int l1_env(const char* n, int dflt) {
  const char* e = getenv(n);
  return e ? atoi(e) : dflt;
}

static int l1_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}
// The kernels' dynamic LDS beyond 64 KB is a per-DEVICE function attribute: set it once per (kernel, device) - a process that
// launches on a second GPU (torch.cuda.set_device(1) after device 0 was used) needs it there too.
static inline bool l1_dyn_lds(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_relaxed) & bit) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  done.fetch_or(bit);
  return true;
}
}  // namespace

// 1: launched; 0: not a launch of this family.  lowp: the reduced-precision product forms (one product per term; 1 fp16, 2 bf16).
// dry != 0: only tell whether the launch would be taken (the profiler prices it as this family's before it is issued)
int gfv_internal_lin1s_try(const gfv_rowtile_args_t* a, int lowp, hipStream_t stream, int dry);   // lin1s.hip: the small-tile form
int gfv_internal_lin1_try(const gfv_rowtile_args_t* a, int lowp, hipStream_t stream, int dry) {
  static const int on = l1_env("GFV_LIN1", 1);
  if (on && gfv_internal_lin1s_try(a, lowp, stream, dry)) return 1;   // short launches: column-owner small tiles (round 5)
  static const int min_m = l1_env("GFV_LIN1_MIN_M", 1024);
  if (!on || a->nlayers != 1 || a->M < min_m || (a->flags & (GFV_CHAIN_ROW_OWNER | GFV_CHAIN_COLUMN_OWNER))) return 0;
  const gfv_layer_t& L = a->layer[0];
  if (a->nseg == 2 && a->seg[0].csr_rowptr && a->seg[1].csr_rowptr) {
    static const int csr_on = l1_env("GFV_LIN1_CSR", 1);
    if (!csr_on || !L.Wh || !a->wmax || L.op != GFV_OP_NONE || L.save || L.aux || L.bias || L.bias2 || a->in_op != GFV_IN_NONE ||
        a->fin_op != GFV_FIN_PLAIN || L.K != 256 || L.N != 128)
      return 0;
    if (a->in_add || a->in_save || a->gscale || a->fin_presave || a->fin_stats || a->in_stats || a->dw_partial || a->gadd || a->padd ||
        a->in_aux || a->out_nores || a->ln_partial || a->res[0] || a->res[1] || a->res[2])
      return 0;
    if (!a->out[0] || a->out[1] || a->out[2] || (a->out_ld[0] & 3) || !al16(a->out[0])) return 0;
    Lin1CsrArgs B{};
    for (int i = 0; i < 2; ++i) {
      const gfv_seg_t& sg = a->seg[i];
      if (sg.width != 128 || !sg.idx || sg.csr_scale || (sg.ld & 3) || !al16(sg.ptr) || (sg.save && !al16(sg.save))) return 0;
      B.src[i] = sg.ptr; B.src_ld[i] = sg.ld; B.rowptr[i] = sg.csr_rowptr; B.col[i] = sg.idx; B.save[i] = sg.save;
    }
    B.img = L.Wh;
    B.wmax = a->wmax;
    B.out = a->out[0];
    B.out_ld = a->out_ld[0];
    B.M = a->M;
    if (dry) return 1;
    int* st = gfv_internal_status_ptr();
    const dim3 grid((a->M + 127) / 128), blk(512);
#define L1_CSR(LP)                                                                                               \
  do {                                                                                                           \
    static std::atomic<unsigned long long> done{0};                                                              \
    if (!l1_dyn_lds(reinterpret_cast<const void*>(&lin1_csr_kernel<LP>), 131072, done)) return 0;                \
    GFV_LAUNCH((lin1_csr_kernel<LP>), grid, blk, 131072, stream, B, st);                                 \
  } while (0)
    if (lowp == 2) L1_CSR(2);
    else if (lowp) L1_CSR(1);
    else L1_CSR(0);
#undef L1_CSR
    return 1;
  }
  if (a->fin_op == GFV_FIN_LNBWD) {
    // [M,256] x W^T -> LayerNorm backward (+ residual), per-tile (dgamma, dbeta) partials
    static const int lnb_on = l1_env("GFV_LIN1_LNBWD", 1);
    if (!lnb_on || !L.Wh || !a->wmax || L.op != GFV_OP_NONE || L.save || L.aux || L.bias || L.bias2 || a->in_op != GFV_IN_NONE) return 0;
    if (a->nseg != 2 || L.K != 256 || L.N != 128 || a->in_add || a->in_save || a->gscale || a->fin_presave || a->fin_stats || a->in_stats ||
        a->dw_partial || a->gadd || a->padd || a->in_aux || a->out_nores || !a->fin_aux || !a->fin_gamma || !a->ln_partial)
      return 0;
    for (int i = 0; i < 2; ++i) {
      const gfv_seg_t& sg = a->seg[i];
      if (sg.width != 128 || sg.idx || sg.csr_rowptr || sg.csr_scale || sg.save || (sg.ld & 3) || !al16(sg.ptr)) return 0;
    }
    if (!a->out[0] || a->out[1] || a->out[2] || (a->out_ld[0] & 3) || !al16(a->out[0]) || a->res[1] || a->res[2]) return 0;
    if (a->res[0] && ((a->res_ld[0] & 3) || !al16(a->res[0]))) return 0;
    if (!al16(a->fin_aux) || !al16(a->fin_gamma) || !al16(a->ln_partial)) return 0;
    Lin1LnbArgs B{};
    for (int i = 0; i < 2; ++i) { B.a.seg[i] = a->seg[i].ptr; B.a.seg_ld[i] = a->seg[i].ld; }
    B.a.img = L.Wh;
    B.a.wmax = a->wmax;
    B.a.res[0] = a->res[0]; B.a.res_ld[0] = a->res_ld[0];
    B.a.out[0] = a->out[0]; B.a.out_ld[0] = a->out_ld[0];
    B.a.M = a->M;
    {
      const int hs = gfv_hidden_size();
      const int h = (hs > 0 && hs < 128) ? hs : 128;
      B.a.ln_inv_n = 1.0f / (float)h;
      B.a.ln_npad = (float)(128 - h);
    }
    B.y = a->fin_aux;
    B.fgamma = a->fin_gamma;
    B.ln_partial = a->ln_partial;
    B.n_tiles = (a->M + 63) / 64;
    if (dry) return 1;
    int* st = gfv_internal_status_ptr();
    const dim3 grid((a->M + 127) / 128), blk(512);
#define L1_LNB(LP)                                                                                               \
  do {                                                                                                           \
    static std::atomic<unsigned long long> done{0};                                                              \
    if (!l1_dyn_lds(reinterpret_cast<const void*>(&lin1_lnbwd_kernel<LP>), 131072, done)) return 0;              \
    GFV_LAUNCH((lin1_lnbwd_kernel<LP>), grid, blk, 131072, stream, B, st);                               \
  } while (0)
    if (lowp == 2) L1_LNB(2);
    else if (lowp) L1_LNB(1);
    else L1_LNB(0);
#undef L1_LNB
    return 1;
  }
  const bool dgelu = L.op == GFV_OP_MUL_DGELU;
  if (!L.Wh || !a->wmax || (L.op != GFV_OP_NONE && !dgelu) || L.save) return 0;
  if (dgelu ? (!L.aux || !al16(L.aux) || L.bias || L.bias2 || a->in_op != GFV_IN_NONE) : (L.aux != nullptr)) return 0;
  if (a->in_op != GFV_IN_NONE && a->in_op != GFV_IN_GELU && a->in_op != GFV_IN_LN) return 0;
  if (a->in_op == GFV_IN_LN && (a->nseg != 1 || !a->in_gamma || !a->in_beta || !al16(a->in_gamma) || !al16(a->in_beta) || a->in_add || a->in_save))
    return 0;
  if (a->fin_op != GFV_FIN_PLAIN || a->fin_presave || a->fin_stats || a->in_stats || a->dw_partial || a->ln_partial || a->gadd || a->padd ||
      a->in_aux || a->out_nores)
    return 0;
  // gscale: slot 0 (the prologue result's rows) is written here; the chain kernel also fills slot 1 of a one-pass GELU' launch
  // without residual (the scale of its OUTPUT rows) - such a launch stays there
  if (a->gscale && (a->nseg != 1 || (dgelu && L.N == 128 && !a->res[0]))) return 0;
  if (a->nseg < 1 || a->nseg > 2 || L.K != 128 * a->nseg || (L.N != 128 && L.N != 256) || (a->nseg == 2 && L.N != 128)) return 0;
  for (int i = 0; i < a->nseg; ++i) {
    const gfv_seg_t& s = a->seg[i];
    if (s.width != 128 || s.idx || s.csr_rowptr || s.csr_scale || s.save || (s.ld & 3) || !al16(s.ptr)) return 0;
  }
  if (a->in_add && (a->nseg != 1 || !al16(a->in_add))) return 0;
  if (a->in_save && (a->nseg != 1 || !al16(a->in_save))) return 0;
  const int np = L.N / 128;
  for (int p = 0; p < 3; ++p) {
    if (p < np) {
      if (!a->out[p] || (a->out_ld[p] & 3) || !al16(a->out[p])) return 0;
      if (a->res[p] && ((a->res_ld[p] & 3) || !al16(a->res[p]))) return 0;
    } else if (a->out[p] || a->res[p]) {
      return 0;
    }
  }
  if (L.bias && !al16(L.bias)) return 0;
  if (L.bias2 && !al16(L.bias2)) return 0;
  Lin1Args A{};
  for (int i = 0; i < a->nseg; ++i) { A.seg[i] = a->seg[i].ptr; A.seg_ld[i] = a->seg[i].ld; }
  A.in_add = a->in_add;
  A.in_save = a->in_save;
  A.img = L.Wh;
  A.wmax = a->wmax;
  A.bias[0] = L.bias;
  A.bias[1] = np > 1 ? (L.bias2 ? L.bias2 : (L.bias ? L.bias + 128 : nullptr)) : nullptr;
  for (int p = 0; p < np; ++p) { A.res[p] = a->res[p]; A.res_ld[p] = a->res_ld[p]; A.out[p] = a->out[p]; A.out_ld[p] = a->out_ld[p]; }
  A.M = a->M;
  A.gamma = a->in_gamma;
  A.beta = a->in_beta;
  {
    const int hs = gfv_hidden_size();   // the launch's LayerNorm width (the calling thread's context; rowtile.hip has set it from args->hidden)
    const int h = (hs > 0 && hs < 128) ? hs : 128;
    A.ln_inv_n = 1.0f / (float)h;
    A.ln_npad = (float)(128 - h);
  }
  A.aux = L.aux;
  A.gscale = a->gscale;
  if (dry) return 1;
  int* st = gfv_internal_status_ptr();
  // at most GFV_LIN1_WGS_PER_CU workgroups per CU (default 2: the 64 KB forms fit two per CU, the 128 KB ones run them in turn)
  static const int per_cu = getenv("GFV_LIN1_WGS_PER_CU") ? atoi(getenv("GFV_LIN1_WGS_PER_CU")) : 2;
  const int nblk_all = (a->M + 127) / 128;
  const int cap = per_cu > 0 ? per_cu * l1_cus() : nblk_all;
  const dim3 grid(nblk_all < cap ? nblk_all : cap), blk(512);
  // instantiated: (in_op none | GELU | LayerNorm) without the GELU' epilogue, in_op none with it
  const int iop = a->in_op == GFV_IN_GELU ? 1 : (a->in_op == GFV_IN_LN ? 2 : 0);
#define L1_ONE(KS, NP, IOP, DG, LP)                                                                                             \
  do {                                                                                                                          \
    static std::atomic<unsigned long long> done{0};                                                                             \
    if (!l1_dyn_lds(reinterpret_cast<const void*>(&lin1_kernel<KS, NP, IOP, DG, LP>), (NP) * (KS) * 16384, done)) return 0;     \
    GFV_LAUNCH((lin1_kernel<KS, NP, IOP, DG, LP>), grid, blk, (size_t)(NP) * (KS) * 16384, stream, A, st);              \
  } while (0)
#define L1_FORM(KS, NP, LP)                                                                                                     \
  do {                                                                                                                          \
    if (dgelu) L1_ONE(KS, NP, 0, true, LP);                                                                                     \
    else if (iop == 0) L1_ONE(KS, NP, 0, false, LP);                                                                            \
    else if (iop == 1) L1_ONE(KS, NP, 1, false, LP);                                                                            \
    else L1_ONE(KS, NP, 2, false, LP);                                                                                          \
  } while (0)
#define L1_LAUNCH(KS, NP)                                                                                                       \
  do {                                                                                                                          \
    if (lowp == 2) L1_FORM(KS, NP, 2);                                                                                          \
    else if (lowp) L1_FORM(KS, NP, 1);                                                                                          \
    else L1_FORM(KS, NP, 0);                                                                                                    \
  } while (0)
  if (a->nseg == 2) L1_LAUNCH(8, 1);
  else if (np == 2) L1_LAUNCH(4, 2);
  else L1_LAUNCH(4, 1);
#undef L1_LAUNCH
#undef L1_FORM
#undef L1_ONE
  return 1;
}
