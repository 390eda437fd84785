// gfv-build-flags: -fno-slp-vectorize
// Column-owner SMALL-TILE form of the Transolver block's row-local forward chain (round 5; contract: include/gfv.h, gfv_trans_mlp_fwd):
//   fx1 = out_x W_out^T + b_out + fx_in;   z = LayerNorm(fx1; gamma, beta) W_pre^T + b_pre  [M, 256];   out = gelu(z) W_post^T + b_post + fx1
// (GraphTransolver.py:93-95,163-169) for SHORT launches.  transmlp.hip runs the chain on 128-row blocks with every layer's image staged
// in LDS: at 5 k rows that is 41 workgroups on 256 CUs, slower than the three single-layer launches it replaces (9 + 9 + 11 us on
// lin1s.hip).  Here, as in cfwd.hip: one 32-row tile per workgroup, 8 waves, wave w owns output columns 16 w .. 16 w + 15 of the
// 128-wide layers and 32 w .. 32 w + 31 of the 256-wide one; weight slices straight from L2 into registers a layer ahead;
// activations cross waves as MFMA B fragments in LDS; 4 barriers.
//   * LayerNorm sits in the MIDDLE of this chain: per-wave (mean, M2) pairs over 16 columns, Chan's combination across the eight waves
//     (the statistics of cfwd.hip's final LayerNorm, layout-agnostic width included), then the normalised rows go on as fragments
//     behind ONE power-of-two scale per launch taken from a bound - |LN| <= sqrt(127) max|gamma| + max|beta| - not from the rows
//     (a row maximum would need a second exchange behind the statistics): two or three bits of the 22 on typical rows.
//   * gelu(z) is split behind the fixed scale of every column-owner kernel (CT_SH; beyond 2^11 GFV_FLAG_CHAIN_RANGE is raised).
#include <cstdlib>

#include "tchain_kernel.h"
#include "../../include/gfv.h"

#include "gfv_limits.h"
int* gfv_internal_status_ptr();

namespace {

constexpr float CT_SH = 16.0f;
constexpr float CT_SH_INV = 1.0f / 16.0f;
constexpr float CT_SH_LIMIT = 2048.0f;

__device__ __forceinline__ void ct_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct CtLds {
  static constexpr int B0 = 0;                       // out_x fragments: 2 groups x 4 k-groups x 2 KB
  static constexpr int B1 = 16384;                   // LayerNorm output fragments
  static constexpr int B2 = 32768;                   // gelu(z) fragments: 2 groups x 8 k-groups x 2 KB
  static constexpr int SINV = 65536;                 // float [32]: 1 / row scale of the input rows
  static constexpr int LNP = SINV + 128;             // float2 [32][8]: per-wave (mean, M2) of a row
  static constexpr int BND = LNP + 32 * 8 * 8;       // float [8]: per-wave bound of |LayerNorm output|
  static constexpr int TOTAL = BND + 32;
};

struct CtArgs {
  const float* x; const float* res;
  const void* imgA; const void* imgB; const void* imgC;
  const float* bA; const float* bB; const float* bC;
  const float* gamma; const float* beta; const float* wmax;
  float* fx1; float* z; float* out;
  int M, hidden;
};

template <int LOWP>
__global__ __launch_bounds__(512, 2) void ctrans_fwd_kernel(const CtArgs A, int* status) {
  constexpr bool BF = LOWP == 2;
  constexpr int TG = 2;
  __shared__ __attribute__((aligned(16))) char lds[CtLds::TOTAL];
  char* b0 = lds + CtLds::B0;
  char* b1 = lds + CtLds::B1;
  char* b2 = lds + CtLds::B2;
  float* sinv = reinterpret_cast<float*>(lds + CtLds::SINV);
  float* lnp = reinterpret_cast<float*>(lds + CtLds::LNP);
  float* bnd = reinterpret_cast<float*>(lds + CtLds::BND);

  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int row0 = (int)blockIdx.x * 32;
  if (row0 >= A.M) return;
  const int ngt = min(TG, (A.M - row0 + 15) >> 4);
  const int c0 = 16 * w + 4 * g;   // this lane's columns of a 128-wide layer: c0 .. c0 + 3
  const float invw = 1.0f / gfv_pow2_scale(*A.wmax);

  // ---- to_out's weight slice: in flight beside the row loads ----
  gfv_f16x8 ah[4], al[4];
  {
    const gfv_f16x8* im = reinterpret_cast<const gfv_f16x8*>(A.imgA) + (size_t)w * 128 + lane;
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      ah[T] = im[T * 1024];
      al[T] = LOWP ? ah[T] : im[T * 1024 + 64];
    }
  }
  // ---- out_x rows -> row scale -> fragments (wave q loads group q) ----
  if (w < TG) {
    const int row = min(row0 + 16 * w + j, A.M - 1);
    const float* p = A.x + (size_t)row * 128 + 4 * g;
    float v[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float4 t = ld4(p + 16 * u);
      v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
    }
    float m0 = 0.f, m1 = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      m0 = max3_abs(m0, v[u][0], v[u][1]);
      m1 = max3_abs(m1, v[u][2], v[u][3]);
    }
    const float s = gfv_pow2_scale(row_max4(max3_abs(0.f, m0, m1)));
    if (g == 0) sinv[w * 16 + j] = 1.0f / s;
    gfv_uint4* dst = reinterpret_cast<gfv_uint4*>(b0 + (size_t)w * 4 * 2048) + lane;
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      float e[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { e[r] = v[2 * T][r] * s; e[4 + r] = v[2 * T + 1][r] * s; }
      gfv_uint4 hi, lo;
      gfv_split8_t<BF>(e, hi, lo);
      dst[(2 * T) * 64] = hi;
      if (!LOWP) dst[(2 * T + 1) * 64] = lo;
    }
  }
  // this wave's bias / affine columns, the residual rows, and its bound of the LayerNorm output
  const float4 bA = A.bA ? ld4(A.bA + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 gam = ld4(A.gamma + c0), bet = ld4(A.beta + c0);
  float4 rres[TG];
#pragma unroll
  for (int q = 0; q < TG; ++q) rres[q] = ld4(A.res + (size_t)min(row0 + 16 * q + j, A.M - 1) * 128 + c0);
  {
    // |LN[c]| <= sqrt(127) |gamma[c]| + |beta[c]|  (|xhat| <= sqrt(n - 1))
    const float b4 = fmaxf(fmaxf(fabsf(gam.x) * 11.27f + fabsf(bet.x), fabsf(gam.y) * 11.27f + fabsf(bet.y)),
                           fmaxf(fabsf(gam.z) * 11.27f + fabsf(bet.z), fabsf(gam.w) * 11.27f + fabsf(bet.w)));
    const float bw = gfv_wave_max(b4);
    if (lane == 0) bnd[w] = bw;
  }
  float mabs = 0.f;
  ct_barrier();

  // ---- to_out: b0 -> fx1 (saved, kept) ; LayerNorm statistics ----
  floatx4 acc[TG];
#pragma unroll
  for (int q = 0; q < TG; ++q) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int T = 0; T < 4; ++T)
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const gfv_f16x8* f = reinterpret_cast<const gfv_f16x8*>(b0 + (size_t)(q * 4 + T) * 2048) + lane;
      const gfv_f16x8 xh = f[0];
      if (!LOWP) {
        const gfv_f16x8 xl = f[64];
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[T], xh, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[T], xl, acc[q], 0, 0, 0);
      }
      acc[q] = gfv_mma_hh<BF>(ah[T], xh, acc[q]);
    }
  // linear_pre's slice (two n-tiles of its 256 columns): requested behind to_out's products
  gfv_f16x8 bh[2][4], bl[2][4];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int t16 = 2 * w + n;   // n-tile of the 256-wide layer: pass t16 >> 3, tile t16 & 7
    const gfv_f16x8* im = reinterpret_cast<const gfv_f16x8*>(A.imgB) + (size_t)((t16 >> 3) * 4 * 8 + (t16 & 7)) * 128 + lane;
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      bh[n][T] = im[T * 1024];
      bl[n][T] = LOWP ? bh[n][T] : im[T * 1024 + 64];
    }
  }
  float fx[TG][4];
#pragma unroll
  for (int q = 0; q < TG; ++q) {
    const int row = row0 + 16 * q + j;
    const bool live = q < ngt && row < A.M;
    const float si = sinv[q * 16 + j];
    fx[q][0] = (acc[q][0] * si) * invw + bA.x + rres[q].x; fx[q][1] = (acc[q][1] * si) * invw + bA.y + rres[q].y;
    fx[q][2] = (acc[q][2] * si) * invw + bA.z + rres[q].z; fx[q][3] = (acc[q][3] * si) * invw + bA.w + rres[q].w;
    if (live) st4(A.fx1 + (size_t)row * 128 + c0, fx[q]);
    const float mw = row_sum((fx[q][0] + fx[q][1]) + (fx[q][2] + fx[q][3])) * (1.0f / 16.0f);   // this wave's 16 columns
    const float d0 = fx[q][0] - mw, d1 = fx[q][1] - mw, d2 = fx[q][2] - mw, d3 = fx[q][3] - mw;
    const float m2 = row_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
    if (g == 0) *reinterpret_cast<float2*>(lnp + ((q * 16 + j) * 8 + w) * 2) = make_float2(mw, m2);
  }
  ct_barrier();
  // ---- LayerNorm ln_2 -> fragments of linear_pre's input ----
  float sB;
  {
    const float4 ba = *reinterpret_cast<const float4*>(bnd), bb = *reinterpret_cast<const float4*>(bnd + 4);
    sB = gfv_pow2_scale(fmaxf(fmaxf(fmaxf(ba.x, ba.y), fmaxf(ba.z, ba.w)), fmaxf(fmaxf(bb.x, bb.y), fmaxf(bb.z, bb.w))));
    const int hcols = (A.hidden > 0 && A.hidden < 128) ? A.hidden : 128;
    const float inv_h = 1.0f / (float)hcols, npad = (float)(128 - hcols);
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const float4* pp = reinterpret_cast<const float4*>(lnp + (q * 16 + j) * 16);
      const float4 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];   // (mean, M2) x 8 waves
      const float m128 = (((p0.x + p0.z) + (p1.x + p1.z)) + ((p2.x + p2.z) + (p3.x + p3.z))) * 0.125f;
      const float e0 = p0.x - m128, e1 = p0.z - m128, e2 = p1.x - m128, e3 = p1.z - m128, e4 = p2.x - m128, e5 = p2.z - m128,
                  e6 = p3.x - m128, e7 = p3.z - m128;
      const float m2a = (((p0.y + p0.w) + (p1.y + p1.w)) + ((p2.y + p2.w) + (p3.y + p3.w))) +
                        16.0f * (((e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3)) + ((e4 * e4 + e5 * e5) + (e6 * e6 + e7 * e7)));
      // (a narrower model: statistics over all 128 columns, whose padded ones are exactly zero, corrected - cfwd.hip)
      const float mean = hcols == 128 ? m128 : (m128 * 128.0f) * inv_h;
      const float dm = m128 - mean;
      const float m2 = hcols == 128 ? m2a : (m2a + 128.0f * dm * dm) - npad * (mean * mean);
      const float rstd = rsqrtf(m2 * inv_h + 1e-5f);
      const float l0 = (fx[q][0] - mean) * rstd * gam.x + bet.x, l1 = (fx[q][1] - mean) * rstd * gam.y + bet.y;
      const float l2 = (fx[q][2] - mean) * rstd * gam.z + bet.z, l3 = (fx[q][3] - mean) * rstd * gam.w + bet.w;
      unsigned h0, h1, lo0, lo1;
      gfv_split_pair_t<BF>(l0 * sB, l1 * sB, h0, lo0);
      gfv_split_pair_t<BF>(l2 * sB, l3 * sB, h1, lo1);
      char* dst = b1 + (size_t)(q * 4 + (w >> 1)) * 2048 + lane * 16 + (w & 1) * 8;   // half of k-group w >> 1
      *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
      if (!LOWP) *reinterpret_cast<uint2*>(dst + 1024) = make_uint2(lo0, lo1);
    }
  }
  ct_barrier();
  // ---- linear_pre: b1 -> z (saved); gelu(z) -> b2 (k-group w of linear_post's 256-deep input) ----
  floatx4 zc[TG][2];
#pragma unroll
  for (int q = 0; q < TG; ++q)
#pragma unroll
    for (int n = 0; n < 2; ++n) zc[q][n] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int T = 0; T < 4; ++T)
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const gfv_f16x8* f = reinterpret_cast<const gfv_f16x8*>(b1 + (size_t)(q * 4 + T) * 2048) + lane;
      const gfv_f16x8 xh = f[0];
      if (!LOWP) {
        const gfv_f16x8 xl = f[64];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          zc[q][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[n][T], xh, zc[q][n], 0, 0, 0);
          zc[q][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[n][T], xl, zc[q][n], 0, 0, 0);
        }
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) zc[q][n] = gfv_mma_hh<BF>(bh[n][T], xh, zc[q][n]);
    }
  // linear_post's slice (256 deep: 8 k-groups): requested behind linear_pre's products
  gfv_f16x8 ch[8], cl[8];
  {
    const gfv_f16x8* im = reinterpret_cast<const gfv_f16x8*>(A.imgC) + (size_t)w * 128 + lane;
#pragma unroll
    for (int T = 0; T < 8; ++T) {
      ch[T] = im[T * 1024];
      cl[T] = LOWP ? ch[T] : im[T * 1024 + 64];
    }
  }
  {
    const float isB = 1.0f / sB;
    const int cz = 32 * w + 4 * g;   // this lane's columns of the 256-wide layer: cz .. cz + 3 and cz + 16 .. cz + 19
    float4 bB[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) bB[n] = A.bB ? ld4(A.bB + cz + 16 * n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const int row = row0 + 16 * q + j;
      const bool live = q < ngt && row < A.M;
      float e8[8];
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        float zz[4] = {(zc[q][n][0] * isB) * invw + bB[n].x, (zc[q][n][1] * isB) * invw + bB[n].y,
                       (zc[q][n][2] * isB) * invw + bB[n].z, (zc[q][n][3] * isB) * invw + bB[n].w};
        if (live) st4(A.z + (size_t)row * 256 + cz + 16 * n, zz);
        const gfv_f2 g01 = gfv_gelu2(gfv_f2{zz[0], zz[1]}), g23 = gfv_gelu2(gfv_f2{zz[2], zz[3]});
        mabs = fmaxf(mabs, live ? max3_abs(max3_abs(0.f, g01.x, g01.y), g23.x, g23.y) : 0.f);
        e8[4 * n + 0] = g01.x * CT_SH; e8[4 * n + 1] = g01.y * CT_SH; e8[4 * n + 2] = g23.x * CT_SH; e8[4 * n + 3] = g23.y * CT_SH;
      }
      gfv_uint4 hi, lo;
      gfv_split8_t<BF>(e8, hi, lo);
      gfv_uint4* dst = reinterpret_cast<gfv_uint4*>(b2 + (size_t)(q * 8 + w) * 2048) + lane;
      dst[0] = hi;
      if (!LOWP) dst[64] = lo;
    }
  }
  const float4 bC = A.bC ? ld4(A.bC + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
  ct_barrier();
  // ---- linear_post + residual fx1 ----
#pragma unroll
  for (int q = 0; q < TG; ++q) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int T = 0; T < 8; ++T)
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const gfv_f16x8* f = reinterpret_cast<const gfv_f16x8*>(b2 + (size_t)(q * 8 + T) * 2048) + lane;
      const gfv_f16x8 xh = f[0];
      if (!LOWP) {
        const gfv_f16x8 xl = f[64];
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl[T], xh, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch[T], xl, acc[q], 0, 0, 0);
      }
      acc[q] = gfv_mma_hh<BF>(ch[T], xh, acc[q]);
    }
#pragma unroll
  for (int q = 0; q < TG; ++q) {
    const int row = row0 + 16 * q + j;
    if (q < ngt && row < A.M) {
      float o[4] = {(acc[q][0] * CT_SH_INV) * invw + bC.x + fx[q][0], (acc[q][1] * CT_SH_INV) * invw + bC.y + fx[q][1],
                    (acc[q][2] * CT_SH_INV) * invw + bC.z + fx[q][2], (acc[q][3] * CT_SH_INV) * invw + bC.w + fx[q][3]};
      st4(A.out + (size_t)row * 128 + c0, o);
    }
  }
  if (mabs > CT_SH_LIMIT) atomicOr(status, 2);   // GFV_FLAG_CHAIN_RANGE
}

// ---- backward: g = g_out (+ g_add) (-> g_sum); g_z = (g W_post) x gelu'(z); g_fx1 = LayerNorm-backward(g_z W_pre; fx1, gamma) + g;
// g_out_x = g_fx1 W_out; per-tile (dgamma, dbeta); the 16-row-group scales of g (include/gfv.h, gfv_trans_mlp_bwd) ----
// Scales: the rows of g carry their own power-of-two scales (the loader waves see whole rows; gscale gets each group's smallest, as the
// 128-row-block kernel writes it); g_z steps down from the tile's largest row by linear_post's guaranteed growth (row 1-norms of this
// wave's slice of the image, the workgroup's maximum through LDS - colchain_kernel.h); g_fx1 takes its scale from the bound of the
// LayerNorm backward (rstd max|v gamma| (2 + sqrt(127)) + max|g|, colchain_kernel.h P0b).  ln_partial: one row per 32 rows here.
struct CtLdsB {
  static constexpr int B0 = 0;                       // g fragments (2 groups x 4 k-groups x 2 KB); later g_fx1's
  static constexpr int B1 = 16384;                   // g_z fragments: 2 groups x 8 k-groups x 2 KB
  static constexpr int SINV = 49152;                 // float [32]
  static constexpr int LNP = SINV + 128;             // float2 [32][8]: per-wave (mean, M2) of an fx1 row
  static constexpr int PART = LNP + 2048;            // float2 [32][8]: per-wave (s1, s2) of a row
  static constexpr int SMAX = PART + 2048;           // float [8]: per-wave bound term of the LayerNorm backward
  static constexpr int NRM = SMAX + 32;              // float [8]: per-wave largest row 1-norm of linear_post^T's slice
  static constexpr int SMIN = NRM + 32;              // float [2]: smallest row scale of each group
  static constexpr int TOTAL = SMIN + 16;
};

struct CtBwdArgs {
  const float* g; const float* g_add; float* g_sum;
  const float* z; const float* fx1;
  const void* imgPt; const void* imgQt; const void* imgOt;
  const float* gamma; const float* wmax;
  float* g_z; float* g_fx1; float* g_out_x; float* ln_partial; float* gscale;
  int M, hidden;
};

template <int LOWP>
__global__ __launch_bounds__(512, 2) void ctrans_bwd_kernel(const CtBwdArgs A, int* status) {
  constexpr bool BF = LOWP == 2;
  constexpr int TG = 2;
  __shared__ __attribute__((aligned(16))) char lds[CtLdsB::TOTAL];
  char* b0 = lds + CtLdsB::B0;
  char* b1 = lds + CtLdsB::B1;
  float* sinv = reinterpret_cast<float*>(lds + CtLdsB::SINV);
  float* lnp = reinterpret_cast<float*>(lds + CtLdsB::LNP);
  float* part = reinterpret_cast<float*>(lds + CtLdsB::PART);
  float* smax = reinterpret_cast<float*>(lds + CtLdsB::SMAX);
  float* nrm = reinterpret_cast<float*>(lds + CtLdsB::NRM);
  float* smin = reinterpret_cast<float*>(lds + CtLdsB::SMIN);

  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int row0 = (int)blockIdx.x * 32;
  if (row0 >= A.M) return;
  const int ngt = min(TG, (A.M - row0 + 15) >> 4);
  const int c0 = 16 * w + 4 * g;    // this lane's columns of a 128-wide layer
  const int cz = 32 * w + 4 * g;    // ... of the 256-wide one: cz .. cz + 3 and cz + 16 .. cz + 19
  const float invw = 1.0f / gfv_pow2_scale(*A.wmax);
  float mabs = 0.f;

  // ---- linear_post^T's slice (two n-tiles of its 256 output columns) ----
  gfv_f16x8 ph[2][4], pl[2][4];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int t16 = 2 * w + n;
    const gfv_f16x8* im = reinterpret_cast<const gfv_f16x8*>(A.imgPt) + (size_t)((t16 >> 3) * 4 * 8 + (t16 & 7)) * 128 + lane;
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      ph[n][T] = im[T * 1024];
      pl[n][T] = LOWP ? ph[n][T] : im[T * 1024 + 64];
    }
  }
  // ---- g (+ g_add) rows -> g_sum, row scales (gscale), fragments (wave q loads group q) ----
  if (w < TG) {
    const int row = row0 + 16 * w + j;
    const bool live = w < ngt && row < A.M;
    const size_t mr = (size_t)min(row, A.M - 1);
    float v[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float4 t = ld4(A.g + mr * 128 + 16 * u + 4 * g);
      v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
    }
    if (A.g_add) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float4 t = ld4(A.g_add + mr * 128 + 16 * u + 4 * g);
        v[u][0] += t.x; v[u][1] += t.y; v[u][2] += t.z; v[u][3] += t.w;
      }
    }
    if (A.g_sum && live) {
#pragma unroll
      for (int u = 0; u < 8; ++u) st4(A.g_sum + mr * 128 + 16 * u + 4 * g, v[u]);
    }
    float m0 = 0.f, m1 = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      m0 = max3_abs(m0, v[u][0], v[u][1]);
      m1 = max3_abs(m1, v[u][2], v[u][3]);
    }
    const float s = gfv_pow2_scale(row_max4(max3_abs(0.f, m0, m1)));
    const float sg = gfv_row16_min(s);   // the group's smallest scale = its largest row
    if (g == 0) sinv[w * 16 + j] = 1.0f / s;
    if (lane == 0) {
      smin[w] = sg;
      if (A.gscale && w < ngt) A.gscale[(row0 >> 4) + w] = sg;
    }
    gfv_uint4* dst = reinterpret_cast<gfv_uint4*>(b0 + (size_t)w * 4 * 2048) + lane;
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      float e[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { e[r] = v[2 * T][r] * s; e[4 + r] = v[2 * T + 1][r] * s; }
      gfv_uint4 hi, lo;
      gfv_split8_t<BF>(e, hi, lo);
      dst[(2 * T) * 64] = hi;
      if (!LOWP) dst[(2 * T + 1) * 64] = lo;
    }
  }
  // ---- this wave's columns of the saved rows: z (GELU'), fx1 (LayerNorm statistics, xhat) ----
  float4 zq[TG][2], yq[TG];
#pragma unroll
  for (int q = 0; q < TG; ++q) {
    const size_t mr = (size_t)min(row0 + 16 * q + j, A.M - 1);
#pragma unroll
    for (int n = 0; n < 2; ++n) zq[q][n] = ld4(A.z + mr * 256 + cz + 16 * n);
    yq[q] = ld4(A.fx1 + mr * 128 + c0);
  }
  const float4 gam = ld4(A.gamma + c0);
  {
    // largest row 1-norm of the slice (both n-tiles): the growth bound of g -> g_z
    float nmax = 0.f;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      float acc = 0.f;
#pragma unroll
      for (int T = 0; T < 4; ++T) {
        if (LOWP == 2) {
          const gfv_bf16x8 hb = __builtin_bit_cast(gfv_bf16x8, ph[n][T]);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc += fabsf((float)hb[e]);
        } else if (LOWP == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc += fabsf((float)ph[n][T][e]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc += fabsf((float)ph[n][T][e] + (float)pl[n][T][e]);
        }
      }
      nmax = fmaxf(nmax, gfv_wave_max(row_sum(acc)));
    }
    if (lane == 0) nrm[w] = nmax;
  }
#pragma unroll
  for (int q = 0; q < TG; ++q) {
    const float mw = row_sum((yq[q].x + yq[q].y) + (yq[q].z + yq[q].w)) * (1.0f / 16.0f);
    const float d0 = yq[q].x - mw, d1 = yq[q].y - mw, d2 = yq[q].z - mw, d3 = yq[q].w - mw;
    const float m2 = row_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
    if (g == 0) *reinterpret_cast<float2*>(lnp + ((q * 16 + j) * 8 + w) * 2) = make_float2(mw, m2);
  }
  ct_barrier();

  // ---- g_z = (g W_post) x gelu'(z): b0 -> saved, fragments in b1 (k-group w of linear_pre^T's 256-deep input) ----
  const float s_tile = fminf(smin[0], smin[1]);
  float s_gz;
  {
    const float4 na = *reinterpret_cast<const float4*>(nrm), nb = *reinterpret_cast<const float4*>(nrm + 4);
    const float nP = fmaxf(fmaxf(fmaxf(na.x, na.y), fmaxf(na.z, na.w)), fmaxf(fmaxf(nb.x, nb.y), fmaxf(nb.z, nb.w)));
    s_gz = s_tile * (1.0f / gfv_pow2_ceil(1.13f * nP * invw));
  }
  floatx4 zc[TG][2];
#pragma unroll
  for (int q = 0; q < TG; ++q)
#pragma unroll
    for (int n = 0; n < 2; ++n) zc[q][n] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int T = 0; T < 4; ++T)
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const gfv_f16x8* f = reinterpret_cast<const gfv_f16x8*>(b0 + (size_t)(q * 4 + T) * 2048) + lane;
      const gfv_f16x8 xh = f[0];
      if (!LOWP) {
        const gfv_f16x8 xl = f[64];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          zc[q][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pl[n][T], xh, zc[q][n], 0, 0, 0);
          zc[q][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[n][T], xl, zc[q][n], 0, 0, 0);
        }
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) zc[q][n] = gfv_mma_hh<BF>(ph[n][T], xh, zc[q][n]);
    }
  // linear_pre^T's slice (256 deep: 8 k-groups, one n-tile): requested behind these products
  gfv_f16x8 qh[8], ql[8];
  {
    const gfv_f16x8* im = reinterpret_cast<const gfv_f16x8*>(A.imgQt) + (size_t)w * 128 + lane;
#pragma unroll
    for (int T = 0; T < 8; ++T) {
      qh[T] = im[T * 1024];
      ql[T] = LOWP ? qh[T] : im[T * 1024 + 64];
    }
  }
  // the residual branch's rows (g + g_add, this wave's columns): in flight through the next phase
  float4 rg[TG];
#pragma unroll
  for (int q = 0; q < TG; ++q) {
    const size_t mr = (size_t)min(row0 + 16 * q + j, A.M - 1);
    rg[q] = ld4(A.g + mr * 128 + c0);
    if (A.g_add) {
      const float4 t = ld4(A.g_add + mr * 128 + c0);
      rg[q].x += t.x; rg[q].y += t.y; rg[q].z += t.z; rg[q].w += t.w;
    }
  }
#pragma unroll
  for (int q = 0; q < TG; ++q) {
    const int row = row0 + 16 * q + j;
    const bool live = q < ngt && row < A.M;
    const float si = sinv[q * 16 + j];
    float e8[8];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const gfv_f2 d01 = gfv_dgelu2(gfv_f2{zq[q][n].x, zq[q][n].y}), d23 = gfv_dgelu2(gfv_f2{zq[q][n].z, zq[q][n].w});
      float v[4] = {((zc[q][n][0] * si) * invw) * d01.x, ((zc[q][n][1] * si) * invw) * d01.y,
                    ((zc[q][n][2] * si) * invw) * d23.x, ((zc[q][n][3] * si) * invw) * d23.y};
      if (live) st4(A.g_z + (size_t)row * 256 + cz + 16 * n, v);
#pragma unroll
      for (int r = 0; r < 4; ++r) e8[4 * n + r] = v[r] * s_gz;
    }
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e += 2) m = max3_abs(m, e8[e], e8[e + 1]);
    mabs = fmaxf(mabs, live ? m : 0.f);
    gfv_uint4 hi, lo;
    gfv_split8_t<BF>(e8, hi, lo);
    gfv_uint4* dst = reinterpret_cast<gfv_uint4*>(b1 + (size_t)(q * 8 + w) * 2048) + lane;
    dst[0] = hi;
    if (!LOWP) dst[64] = lo;
  }
  ct_barrier();

  // ---- v = g_z W_pre (b1, 8 k-groups); LayerNorm backward, first half ----
  floatx4 acc[TG];
#pragma unroll
  for (int q = 0; q < TG; ++q) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int T = 0; T < 8; ++T)
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const gfv_f16x8* f = reinterpret_cast<const gfv_f16x8*>(b1 + (size_t)(q * 8 + T) * 2048) + lane;
      const gfv_f16x8 xh = f[0];
      if (!LOWP) {
        const gfv_f16x8 xl = f[64];
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ql[T], xh, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qh[T], xl, acc[q], 0, 0, 0);
      }
      acc[q] = gfv_mma_hh<BF>(qh[T], xh, acc[q]);
    }
  // to_out^T's slice: requested behind these products
  gfv_f16x8 oh[4], ol[4];
  {
    const gfv_f16x8* im = reinterpret_cast<const gfv_f16x8*>(A.imgOt) + (size_t)w * 128 + lane;
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      oh[T] = im[T * 1024];
      ol[T] = LOWP ? oh[T] : im[T * 1024 + 64];
    }
  }
  float gg[TG][4], xh_[TG][4], rs[TG];
  {
    const int hcols = (A.hidden > 0 && A.hidden < 128) ? A.hidden : 128;
    const float inv_h = 1.0f / (float)hcols, npad = (float)(128 - hcols);
    const float isg = 1.0f / s_gz;
    float dgam[4] = {0.f, 0.f, 0.f, 0.f}, dbet[4] = {0.f, 0.f, 0.f, 0.f};
    float bmax = 0.f;
    const float gm[4] = {gam.x, gam.y, gam.z, gam.w};
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const int row = row0 + 16 * q + j;
      const float lf = (q < ngt && row < A.M) ? 1.0f : 0.0f;   // rows past M must not reach the sums over rows
      // the row's statistics: Chan's combination of the eight waves' (mean, M2) pairs (written before the first barrier)
      const float4* pp = reinterpret_cast<const float4*>(lnp + (q * 16 + j) * 16);
      const float4 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];
      const float m128 = (((p0.x + p0.z) + (p1.x + p1.z)) + ((p2.x + p2.z) + (p3.x + p3.z))) * 0.125f;
      const float e0 = p0.x - m128, e1 = p0.z - m128, e2 = p1.x - m128, e3 = p1.z - m128, e4 = p2.x - m128, e5 = p2.z - m128,
                  e6 = p3.x - m128, e7 = p3.z - m128;
      const float m2a = (((p0.y + p0.w) + (p1.y + p1.w)) + ((p2.y + p2.w) + (p3.y + p3.w))) +
                        16.0f * (((e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3)) + ((e4 * e4 + e5 * e5) + (e6 * e6 + e7 * e7)));
      const float mean = hcols == 128 ? m128 : (m128 * 128.0f) * inv_h;
      const float dm = m128 - mean;
      const float m2 = hcols == 128 ? m2a : (m2a + 128.0f * dm * dm) - npad * (mean * mean);
      const float rstd = rsqrtf(m2 * inv_h + 1e-5f);
      rs[q] = rstd;
      const float y[4] = {yq[q].x, yq[q].y, yq[q].z, yq[q].w};
      float s1 = 0.f, s2 = 0.f, am = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = ((acc[q][r] * isg) * invw) * lf;
        const float xx = (y[r] - mean) * rstd;
        dgam[r] += v * xx;
        dbet[r] += v;
        const float vg = v * gm[r];
        gg[q][r] = vg;
        xh_[q][r] = xx;
        s1 += vg;
        s2 += vg * xx;
        am = fmaxf(am, fabsf(vg));
      }
      s1 = row_sum(s1);
      s2 = row_sum(s2);
      if (g == 0) *reinterpret_cast<float2*>(part + ((q * 16 + j) * 8 + w) * 2) = make_float2(s1, s2);
      bmax = fmaxf(bmax, am * fabsf(rstd));
    }
    bmax = gfv_wave_max(bmax);
    if (lane == 0) smax[w] = bmax;
    // (dgamma, dbeta) of the tile: this wave owns its columns - a sum over the 16 lanes of a DPP row, no cross-wave step
    if (A.ln_partial) {
      float* lp = A.ln_partial + (size_t)blockIdx.x * 256 + c0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dgam[r] = gfv_row16_sum(dgam[r]);
        dbet[r] = gfv_row16_sum(dbet[r]);
      }
      if (j == 0) {
        st4(lp, dgam);
        st4(lp + 128, dbet);
      }
    }
  }
  ct_barrier();
  // ---- g_fx1 = LayerNorm-backward + g -> saved, fragments in b0 ----
  float s3;
  {
    const float4 ma = *reinterpret_cast<const float4*>(smax), mb = *reinterpret_cast<const float4*>(smax + 4);
    const float mx = fmaxf(fmaxf(fmaxf(ma.x, ma.y), fmaxf(ma.z, ma.w)), fmaxf(fmaxf(mb.x, mb.y), fmaxf(mb.z, mb.w)));
    // |g_fx1| <= rstd max|v gamma| (2 + sqrt(127)) + max|g|;  max|g| < 2^14 / s_tile
    s3 = gfv_pow2_scale(mx * 13.5f + 16384.0f / s_tile);
    const int hcols = (A.hidden > 0 && A.hidden < 128) ? A.hidden : 128;
    const float inv_h = 1.0f / (float)hcols;
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const int row = row0 + 16 * q + j;
      const bool live = q < ngt && row < A.M;
      const float4* pp = reinterpret_cast<const float4*>(part + (q * 16 + j) * 16);
      const float4 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];   // (s1, s2) x 8 waves
      const float m1 = (((p0.x + p0.z) + (p1.x + p1.z)) + ((p2.x + p2.z) + (p3.x + p3.z))) * inv_h;
      const float m2 = (((p0.y + p0.w) + (p1.y + p1.w)) + ((p2.y + p2.w) + (p3.y + p3.w))) * inv_h;
      const float rv[4] = {rg[q].x, rg[q].y, rg[q].z, rg[q].w};
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = rs[q] * (gg[q][r] - m1 - xh_[q][r] * m2) + rv[r];
      if (live) st4(A.g_fx1 + (size_t)row * 128 + c0, o);
      unsigned h0, h1, lo0, lo1;
      gfv_split_pair_t<BF>(o[0] * s3, o[1] * s3, h0, lo0);
      gfv_split_pair_t<BF>(o[2] * s3, o[3] * s3, h1, lo1);
      mabs = fmaxf(mabs, live ? max3_abs(max3_abs(0.f, o[0], o[1]), o[2], o[3]) * s3 : 0.f);
      char* dst = b0 + (size_t)(q * 4 + (w >> 1)) * 2048 + lane * 16 + (w & 1) * 8;
      *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
      if (!LOWP) *reinterpret_cast<uint2*>(dst + 1024) = make_uint2(lo0, lo1);
    }
  }
  ct_barrier();
  // ---- g_out_x = g_fx1 W_out ----
#pragma unroll
  for (int q = 0; q < TG; ++q) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int T = 0; T < 4; ++T)
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const gfv_f16x8* f = reinterpret_cast<const gfv_f16x8*>(b0 + (size_t)(q * 4 + T) * 2048) + lane;
      const gfv_f16x8 xh = f[0];
      if (!LOWP) {
        const gfv_f16x8 xl = f[64];
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ol[T], xh, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(oh[T], xl, acc[q], 0, 0, 0);
      }
      acc[q] = gfv_mma_hh<BF>(oh[T], xh, acc[q]);
    }
  {
    const float is3 = 1.0f / s3;
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const int row = row0 + 16 * q + j;
      if (q < ngt && row < A.M) {
        const float o[4] = {(acc[q][0] * is3) * invw, (acc[q][1] * is3) * invw, (acc[q][2] * is3) * invw, (acc[q][3] * is3) * invw};
        st4(A.g_out_x + (size_t)row * 128 + c0, o);
      }
    }
  }
  if (mabs > 60000.0f) atomicOr(status, 2);   // GFV_FLAG_CHAIN_RANGE: a fragment value beyond fp16's range
}


}  // namespace

extern "C" int gfv_hidden_size(void);

// rows of ln_partial a gfv_trans_mlp_bwd launch over M rows fills: one per 32 rows when the small-tile form takes it, one per 64 otherwise
extern "C" int gfv_trans_mlp_ln_rows(int32_t M) {
  if (M <= 0) return 0;
  return (gfv_internal_limit(GFV_LIM_CTRANS_ON) && M <= gfv_internal_limit(GFV_LIM_CTRANS_MAX_M)) ? (M + 31) / 32 : (M + 63) / 64;
}

int gfv_internal_ctrans_bwd_try(const gfv_trans_mlp_bwd_t* a, int form, hipStream_t stream) {
  if (!gfv_internal_limit(GFV_LIM_CTRANS_ON) || a->M > gfv_internal_limit(GFV_LIM_CTRANS_MAX_M) || !gfv_internal_status_ptr()) return 0;
  CtBwdArgs B{a->g, a->g_add, a->g_sum, a->z, a->fx1, a->img_post_t, a->img_pre_t, a->img_out_t, a->gamma, a->wmax,
              a->g_z, a->g_fx1, a->g_out_x, a->ln_partial, a->gscale, a->M, gfv_hidden_size()};
  int* st = gfv_internal_status_ptr();
  const dim3 grid((a->M + 31) / 32), blk(512);
  if (form == 3) GFV_LAUNCH((ctrans_bwd_kernel<2>), grid, blk, 0, stream, B, st);
  else if (form == 2) GFV_LAUNCH((ctrans_bwd_kernel<1>), grid, blk, 0, stream, B, st);
  else GFV_LAUNCH((ctrans_bwd_kernel<0>), grid, blk, 0, stream, B, st);
  return 1;
}

// 1: launched; 0: not this family's launch (too many rows, switched off).  The caller (transmlp.hip) has checked the arguments.
// form: gfv_f16split_enabled() of the calling thread (1 / 2 / 3)
int gfv_internal_ctrans_fwd_try(const gfv_trans_mlp_t* a, int form, hipStream_t stream) {
  if (!gfv_internal_limit(GFV_LIM_CTRANS_ON) || a->M > gfv_internal_limit(GFV_LIM_CTRANS_MAX_M) || !gfv_internal_status_ptr()) return 0;
  CtArgs B{a->x, a->res, a->img_out, a->img_pre, a->img_post, a->b_out, a->b_pre, a->b_post, a->gamma, a->beta, a->wmax,
           a->fx1, a->z, a->out, a->M, gfv_hidden_size()};
  int* st = gfv_internal_status_ptr();
  const dim3 grid((a->M + 31) / 32), blk(512);
  if (form == 3) GFV_LAUNCH((ctrans_fwd_kernel<2>), grid, blk, 0, stream, B, st);
  else if (form == 2) GFV_LAUNCH((ctrans_fwd_kernel<1>), grid, blk, 0, stream, B, st);
  else GFV_LAUNCH((ctrans_fwd_kernel<0>), grid, blk, 0, stream, B, st);
  return 1;
}
