// gfv-build-flags: -fno-slp-vectorize
// Column-owner SMALL-TILE forward of the 3-layer MLPs (round 5): the kernel family behind gfv_rowtile_chain (contract:
// include/gfv.h) for the forward of build_mlp (EPD.py:10-33: Linear GELU Linear GELU Linear LayerNorm) inside the NodeBlock /
// EdgeBlock (blocks.py:54,101-111) and the two encoders (EPD.py:92-119) when the launch is SHORT - a node-level launch of any
// mesh, every launch of a small mesh - in the split-fp16 product forms.
//
// Why: profiles/r05_latency_floor_before.txt.  The row-owner chain (tchain_kernel.h: a wave owns 16 rows and ALL 128 columns,
// the three layers' weights stream through LDS in twelve barrier-separated slices) takes 20 - 27 us for ONE 64-row tile however
// few tiles the launch has: a node-level launch of the 50 k-cell mesh (400 tiles on 256 CUs) takes 35 us, the same launch on a
// 5 k-cell mesh 25 us, on a 1 k-cell mesh 23 us.  One wave walks 288 MFMAs and ~2 400 vector instructions per tile behind
// twelve barriers; the launch's time is that serial walk, not bandwidth.  Here the walk is cut four ways instead:
//   * a workgroup = ONE tile of TG groups of 16 rows: 32 rows on 8 waves (wave w owns output COLUMNS 16 w .. 16 w + 15 of every
//     layer) or 64 rows on 4 waves (columns 32 w .. 32 w + 31): an eighth / a quarter of the MFMAs and of the epilogue arithmetic
//     per wave and tile row;
//   * its slice of a layer's weight image - the A operands W[32 w + i][k], (hi, lo) parts, 16 registers per k-group - comes
//     straight from L2 into registers, one layer ahead of its use: no weight staging, no slice barriers;
//   * activations cross waves between layers as MFMA B fragments in LDS ([group][k-group][part][lane] x 16 B; the 32 columns a
//     wave produces are exactly ONE k-group of the next layer: a 16-byte write per lane and part), one barrier per layer;
//   * LayerNorm over the four waves' partial (mean, M2) pairs (Chan's combination: two-pass accuracy, one exchange).
// 4 barriers per tile instead of 13; per wave 48 TG / 2 MFMAs per 128-deep layer.  Hidden activations are split behind the fixed
// scale CF_SH (colchain_kernel.h CC_SH: GELU outputs in [2^-4, 2^11] keep fp32 accuracy; beyond 2^11 GFV_FLAG_CHAIN_RANGE is
// raised), the layer INPUT rows behind their own power-of-two row scale as everywhere else.
// Price: every tile pulls the three images (160 - 224 KB) from L2 into registers.  Measured (profiles/r05_cfwd.txt, in-step
// averages): 5 k-row launches 11.9 us against 25 (NodeBlock), 17.5 against 27 (EdgeBlock, 10 k rows), 12.0 against 22 (encoder);
// 25 k-row NodeBlock launches 30.9 us against 35 for the row-owner chain; 75 k-row EdgeBlock launches 78 - 83 us against 80.
// What bounds the mid-size launches is NOT that weight stream (round 6: a persistent form with the slices resident in registers
// across tiles - profiles/tools/cfwdp - took 31.5 - 32.3 us where this kernel takes 29 - 30.6: the tile got 40 % shorter, and one
// workgroup per CU instead of two doubled the rounds) but the latency of a tile's dependent chain times the number of rounds
// (797 tiles on 2 x 256 slots).  32-row tiles on 8 waves at every size up to GFV_CFWD_MAX_M = 250 000 rows (beyond - the edge-level launches of 8 meshes per
// GPU - the row-owner chain, whose 64 rows share one weight stream through LDS), the encoders' narrow inputs and the decoder
// included (through round 5 those stopped at 16 384 rows; over eight mesh sizes from 8 k to 40 k nodes the step is 0.4 - 1.2 %
// faster with them here at every size, profiles/r06_dispatch_sweep.txt).  (A 64-row form on 4 waves existed through round 5:
// slower at every size, removed.)
#include <cstdlib>

#include "tchain_kernel.h"

#include "gfv_limits.h"
int* gfv_internal_status_ptr();

namespace {

constexpr float CF_SH = 16.0f;
constexpr float CF_SH_INV = 1.0f / 16.0f;
constexpr float CF_SH_LIMIT = 2048.0f;

__device__ __forceinline__ void cf_barrier() {
  // LDS only: global loads (next layer's weights, the residual rows) and stores stay in flight across it
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int KT0, int TG, int NW>
struct CfLds {
  static constexpr int KTB = KT0 > 4 ? KT0 : 4;
  static constexpr int B0 = 0;                          // the input fragments; later the third layer's input
  static constexpr int B1 = TG * KTB * 2048;            // the second layer's input
  static constexpr int SINV = B1 + TG * 8192;           // float [TG * 16]: 1 / row scale of the input rows
  static constexpr int LNP = SINV + TG * 64;            // float2 [TG * 16][NW]: per-wave (mean, M2) of a row
  static constexpr int TOTAL = LNP + TG * 16 * NW * 8;
};

// KT0: k-groups of the first layer (K / 32, zero padded); N0: 16-column pieces of segment 0 (the rest from segment 1);
// PADD: gathered first-layer addend (factored EdgeBlock); LOWP: 0 three products, 1 / 2 the single-product forms (fp16 / bf16);
// RAGIN: ONE narrow segment (width <= 32, any row stride: the encoders' raw inputs), loaded element by element
// NW: waves per workgroup - 4 (a wave owns NT = 2 n-tiles = 32 columns) or 8 (one n-tile = 16 columns: twice the waves in
// flight per tile, half the instruction stream per wave; the columns a wave produces are then HALF a k-group of the next layer)
// FINLN = false: no LayerNorm and a NARROW last layer (the decoder, EPD.py:199-219: 128 -> 128 -> 128 -> 3): the wave that owns
// columns 0 .. 15 runs the last layer alone and stores the N <= 16 valid columns (any output row stride)
template <int KT0, int N0, int TG, bool PADD, int LOWP, bool RAGIN, int NW = 4, bool FINLN = true>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 2) void cfwd_kernel(const gfv_rowtile_args_t A, int* status) {
  static_assert(TG == 2 || TG == 4, "one loader wave per group");
  static_assert(NW == 4 || NW == 8, "32 or 16 columns per wave");
  constexpr int NT = 8 / NW;   // n-tiles per wave
  using LY = CfLds<KT0, TG, NW>;
  constexpr bool BF = LOWP == 2;
  __shared__ __attribute__((aligned(16))) char lds[LY::TOTAL];
  char* b0 = lds + LY::B0;
  char* b1 = lds + LY::B1;
  float* sinv = reinterpret_cast<float*>(lds + LY::SINV);
  float* lnp = reinterpret_cast<float*>(lds + LY::LNP);

  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int tile = PADD ? gfv_xcd_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int row0 = tile * (16 * TG);
  if (row0 >= A.M) return;
  const int ngt = min(TG, (A.M - row0 + 15) >> 4);   // live groups of this tile
  const int c0 = 16 * NT * w + 4 * g;                // this lane's columns: c0 .. c0 + 3 (and, NT = 2, c0 + 16 .. + 19)
  const float invw = 1.0f / gfv_pow2_scale(*A.wmax);

  // ---- first layer's weight slice: in flight beside the row loads ----
  gfv_f16x8 wh[NT][KT0 > 4 ? KT0 : 4], wl[NT][KT0 > 4 ? KT0 : 4];
  {
    const gfv_f16x8* im = reinterpret_cast<const gfv_f16x8*>(A.layer[0].Wh) + (size_t)(NT * w) * 128 + lane;
#pragma unroll
    for (int T = 0; T < KT0; ++T)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        wh[n][T] = im[T * 1024 + n * 128];
        if (!LOWP) wl[n][T] = im[T * 1024 + n * 128 + 64];
      }
  }
  // gathered addend rows of the first pre-activation: index, then row - two round trips, started now
  float4 ps[PADD ? TG : 1][NT], pr[PADD ? TG : 1][NT];
  if (PADD) {
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const int row = min(row0 + 16 * q + j, A.M - 1);
      const float* s = A.padd + (size_t)A.padd_s[row] * A.padd_ld + c0;
      const float* r = A.padd + (size_t)A.padd_r[row] * A.padd_ld + 128 + c0;
#pragma unroll
      for (int n = 0; n < NT; ++n) { ps[q][n] = ld4(s + 16 * n); pr[q][n] = ld4(r + 16 * n); }
    }
  }
  // ---- input rows -> row scale -> fragments (wave q loads group q) ----
  if (w < TG) {
    const int row = min(row0 + 16 * w + j, A.M - 1);
    float v[2 * KT0][4];
    if (RAGIN) {
      const int width = A.seg[0].width;
      const float* rp = A.seg[0].ptr + (size_t)row * A.seg[0].ld;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int col = 16 * u + 4 * g + e;
          v[u][e] = col < width ? rp[col] : 0.f;
        }
    } else {
      const float* p0 = A.seg[0].ptr + (size_t)row * A.seg[0].ld + 4 * g;
      const float* p1 = p0;
      if (N0 < 2 * KT0) p1 = A.seg[1].ptr + (size_t)row * A.seg[1].ld + 4 * g;
#pragma unroll
      for (int u = 0; u < 2 * KT0; ++u) {
        const float4 t = ld4(u < N0 ? p0 + 16 * u : p1 + 16 * (u - N0));
        v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
      }
    }
    float m0 = 0.f, m1 = 0.f;
#pragma unroll
    for (int u = 0; u < 2 * KT0; ++u) {
      m0 = max3_abs(m0, v[u][0], v[u][1]);
      m1 = max3_abs(m1, v[u][2], v[u][3]);
    }
    const float s = gfv_pow2_scale(row_max4(max3_abs(0.f, m0, m1)));
    if (g == 0) sinv[w * 16 + j] = 1.0f / s;
    gfv_uint4* dst = reinterpret_cast<gfv_uint4*>(b0 + (size_t)w * KT0 * 2048) + lane;
#pragma unroll
    for (int T = 0; T < KT0; ++T) {
      float e[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { e[r] = v[2 * T][r] * s; e[4 + r] = v[2 * T + 1][r] * s; }
      gfv_uint4 hi, lo;
      gfv_split8_t<BF>(e, hi, lo);
      dst[(2 * T) * 64] = hi;
      if (!LOWP) dst[(2 * T + 1) * 64] = lo;
    }
  }
  // this wave's bias columns of the three layers, LayerNorm affine
  float4 bias[3][NT];
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const float* bp = (FINLN || l < 2) ? A.layer[l].bias : nullptr;
#pragma unroll
    for (int n = 0; n < NT; ++n) bias[l][n] = bp ? ld4(bp + c0 + 16 * n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float mabs = 0.f;
  cf_barrier();

  floatx4 acc[TG][NT];
  // one layer's products: acc[q][n] = sum_T W[n-tile 2 w + n][T] x frag[q][T]
  auto mma = [&](const char* xbuf, auto ktc) {
    constexpr int KT = decltype(ktc)::value;
#pragma unroll
    for (int q = 0; q < TG; ++q)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[q][n] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int T = 0; T < KT; ++T) {
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        const gfv_f16x8* f = reinterpret_cast<const gfv_f16x8*>(xbuf + (size_t)(q * KT + T) * 2048) + lane;
        const gfv_f16x8 xh = f[0];
        if (!LOWP) {
          const gfv_f16x8 xl = f[64];
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            acc[q][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[n][T], xh, acc[q][n], 0, 0, 0);
            acc[q][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[n][T], xl, acc[q][n], 0, 0, 0);
          }
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[q][n] = gfv_mma_hh<BF>(wh[n][T], xh, acc[q][n]);
      }
    }
  };
  auto load_w = [&](const void* image) {   // a 128-deep layer's slice
    const gfv_f16x8* im = reinterpret_cast<const gfv_f16x8*>(image) + (size_t)(NT * w) * 128 + lane;
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        wh[n][T] = im[T * 1024 + n * 128];
        if (!LOWP) wl[n][T] = im[T * 1024 + n * 128 + 64];
      }
  };
  // hidden-layer epilogue: v = acc / scales + bias (+ addend) -> saved; gelu(v) -> the next layer's fragments (k-group w)
  auto hidden = [&](int layer, char* xout, float* save) {
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const int row = row0 + 16 * q + j;
      const bool live = q < ngt && row < A.M;
      const float si = layer == 0 ? sinv[q * 16 + j] : CF_SH_INV;
      float a[4 * NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float4 b = bias[layer][n];
        float v[4] = {(acc[q][n][0] * si) * invw + b.x, (acc[q][n][1] * si) * invw + b.y, (acc[q][n][2] * si) * invw + b.z,
                      (acc[q][n][3] * si) * invw + b.w};
        if (PADD && layer == 0) {
          v[0] += ps[PADD ? q : 0][n].x + pr[PADD ? q : 0][n].x; v[1] += ps[PADD ? q : 0][n].y + pr[PADD ? q : 0][n].y;
          v[2] += ps[PADD ? q : 0][n].z + pr[PADD ? q : 0][n].z; v[3] += ps[PADD ? q : 0][n].w + pr[PADD ? q : 0][n].w;
        }
        if (save && live) st4(save + (size_t)row * 128 + c0 + 16 * n, v);
        const gfv_f2 g01 = gfv_gelu2(gfv_f2{v[0], v[1]}), g23 = gfv_gelu2(gfv_f2{v[2], v[3]});
        a[4 * n + 0] = g01.x; a[4 * n + 1] = g01.y; a[4 * n + 2] = g23.x; a[4 * n + 3] = g23.y;
      }
      float m = 0.f;
#pragma unroll
      for (int e = 0; e < 4 * NT; e += 2) m = max3_abs(m, a[e], a[e + 1]);
      mabs = fmaxf(mabs, live ? m : 0.f);
      if constexpr (NT == 2) {
        // the wave's 32 columns are ONE k-group of the next layer: 16 bytes per lane and part
        float e8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) e8[e] = a[e] * CF_SH;
        gfv_uint4 hi, lo;
        gfv_split8_t<BF>(e8, hi, lo);
        gfv_uint4* dst = reinterpret_cast<gfv_uint4*>(xout + (size_t)(q * 4 + w) * 2048) + lane;
        dst[0] = hi;
        if (!LOWP) dst[64] = lo;
      } else {
        // 16 columns: half a k-group (k-group w >> 1, slots 4 (w & 1) .. + 3): 8 bytes per lane and part
        unsigned h0, h1, l0, l1;
        gfv_split_pair_t<BF>(a[0] * CF_SH, a[1] * CF_SH, h0, l0);
        gfv_split_pair_t<BF>(a[2] * CF_SH, a[3] * CF_SH, h1, l1);
        char* dst = xout + (size_t)(q * 4 + (w >> 1)) * 2048 + lane * 16 + (w & 1) * 8;
        *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
        if (!LOWP) *reinterpret_cast<uint2*>(dst + 1024) = make_uint2(l0, l1);
      }
    }
  };

  // ---- layer 0: b0 -> b1 ----
  mma(b0, std::integral_constant<int, KT0>{});
  load_w(A.layer[1].Wh);
  hidden(0, b1, A.layer[0].save);
  cf_barrier();
  // ---- layer 1: b1 -> b0 ----
  mma(b1, std::integral_constant<int, 4>{});
  load_w(A.layer[2].Wh);
  hidden(1, b0, A.layer[1].save);
  // LayerNorm affine and the residual rows: requested ahead of the last layer
  float4 gam[NT], bet[NT];
  if constexpr (FINLN) {
#pragma unroll
    for (int n = 0; n < NT; ++n) { gam[n] = ld4(A.fin_gamma + c0 + 16 * n); bet[n] = ld4(A.fin_beta + c0 + 16 * n); }
  }
  float4 rres[TG][NT];
  if (A.res[0]) {
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const float* rp = A.res[0] + (size_t)min(row0 + 16 * q + j, A.M - 1) * A.res_ld[0] + c0;
#pragma unroll
      for (int n = 0; n < NT; ++n) rres[q][n] = ld4(rp + 16 * n);
    }
  }
  cf_barrier();
  if constexpr (!FINLN) {
    // ---- narrow last layer (N <= 16 columns): wave 0's first n-tile; bias read element-wise (its array has N entries) ----
    if (w == 0) {
      mma(b0, std::integral_constant<int, 4>{});
      const int ncol = A.layer[2].N;
      const float* bp = A.layer[2].bias;
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        const int row = row0 + 16 * q + j;
        if (q < ngt && row < A.M) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int col = 4 * g + r;
            if (col < ncol) A.out[0][(size_t)row * A.out_ld[0] + col] = (acc[q][0][r] * CF_SH_INV) * invw + (bp ? bp[col] : 0.f);
          }
        }
      }
    }
    if (mabs > CF_SH_LIMIT) atomicOr(status, 2);
    return;
  }
  // ---- layer 2: b0 -> y; LayerNorm ----
  mma(b0, std::integral_constant<int, 4>{});
  // LayerNorm width: a narrower model runs zero padded to 128 columns, and WHERE its h real columns sit depends on the tensor
  // (node latents 0 .. h - 1, the two halves of an edge latent at 0 and 64: FVMmodel/padding.py) - so the statistics are taken
  // over all 128 columns, whose padded ones are exactly zero, and corrected: mean_h = sum / h,
  // sum_real (y - mean_h)^2 = M2_128 + 128 (mean_128 - mean_h)^2 - (128 - h) mean_h^2   (tchain_kernel.h ln_stats: the same sums)
  const int hcols = (A.hidden > 0 && A.hidden < 128) ? A.hidden : 128;
  float y[TG][4 * NT];
#pragma unroll
  for (int q = 0; q < TG; ++q) {
    const int row = row0 + 16 * q + j;
    const bool live = q < ngt && row < A.M;
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const float4 b = bias[2][n];
      float v[4] = {(acc[q][n][0] * CF_SH_INV) * invw + b.x, (acc[q][n][1] * CF_SH_INV) * invw + b.y,
                    (acc[q][n][2] * CF_SH_INV) * invw + b.z, (acc[q][n][3] * CF_SH_INV) * invw + b.w};
      if (A.fin_presave && live) st4(A.fin_presave + (size_t)row * 128 + c0 + 16 * n, v);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        y[q][4 * n + r] = v[r];
        s += v[r];
      }
    }
    const float mw = row_sum(s) * (1.0f / (16.0f * NT));   // this wave's 16 NT columns
    float m2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4 * NT; ++e) {
      const float d = y[q][e] - mw;
      m2 += d * d;
    }
    m2 = row_sum(m2);
    if (g == 0) *reinterpret_cast<float2*>(lnp + ((q * 16 + j) * NW + w) * 2) = make_float2(mw, m2);
  }
  cf_barrier();
  {
    const float inv_h = 1.0f / (float)hcols, npad = (float)(128 - hcols);
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const int row = row0 + 16 * q + j;
      const bool live = q < ngt && row < A.M;
      const float4* pp = reinterpret_cast<const float4*>(lnp + (q * 16 + j) * 2 * NW);
      float m128, m2a;   // mean and sum of squared deviations over all 128 columns (Chan's combination of the waves' pairs)
      if constexpr (NW == 4) {
        const float4 p0 = pp[0], p1 = pp[1];   // (mean, M2) of waves 0, 1 | 2, 3
        m128 = ((p0.x + p0.z) + (p1.x + p1.z)) * 0.25f;
        const float e0 = p0.x - m128, e1 = p0.z - m128, e2 = p1.x - m128, e3 = p1.z - m128;
        m2a = ((p0.y + p0.w) + (p1.y + p1.w)) + 32.0f * ((e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3));
      } else {
        const float4 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];
        m128 = (((p0.x + p0.z) + (p1.x + p1.z)) + ((p2.x + p2.z) + (p3.x + p3.z))) * 0.125f;
        const float e0 = p0.x - m128, e1 = p0.z - m128, e2 = p1.x - m128, e3 = p1.z - m128, e4 = p2.x - m128, e5 = p2.z - m128,
                    e6 = p3.x - m128, e7 = p3.z - m128;
        m2a = (((p0.y + p0.w) + (p1.y + p1.w)) + ((p2.y + p2.w) + (p3.y + p3.w))) +
              16.0f * (((e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3)) + ((e4 * e4 + e5 * e5) + (e6 * e6 + e7 * e7)));
      }
      const float mean = hcols == 128 ? m128 : (m128 * 128.0f) * inv_h;
      const float dm = m128 - mean;
      const float m2 = hcols == 128 ? m2a : (m2a + 128.0f * dm * dm) - npad * (mean * mean);
      const float rstd = rsqrtf(m2 * inv_h + 1e-5f);   // nn.LayerNorm eps (EPD.py:32)
      if (A.fin_stats && live && w == 0 && g == 0) *reinterpret_cast<float2*>(A.fin_stats + 2 * (size_t)row) = make_float2(mean, rstd);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float4 ga = gam[n], be = bet[n];
        float o[4] = {(y[q][4 * n + 0] - mean) * rstd * ga.x + be.x, (y[q][4 * n + 1] - mean) * rstd * ga.y + be.y,
                      (y[q][4 * n + 2] - mean) * rstd * ga.z + be.z, (y[q][4 * n + 3] - mean) * rstd * ga.w + be.w};
        if (live) {
          if (A.out_nores) st4(A.out_nores + (size_t)row * 128 + c0 + 16 * n, o);
          if (A.res[0]) { o[0] += rres[q][n].x; o[1] += rres[q][n].y; o[2] += rres[q][n].z; o[3] += rres[q][n].w; }
          st4(A.out[0] + (size_t)row * A.out_ld[0] + c0 + 16 * n, o);
        }
      }
    }
  }
  if (mabs > CF_SH_LIMIT) atomicOr(status, 2);   // GFV_FLAG_CHAIN_RANGE
}

inline bool cf_al16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

template <int KT0, int N0, bool PADD, bool RAGIN, bool FINLN = true>
void cf_launch(const gfv_rowtile_args_t& a, int lowp, hipStream_t stream) {
  int* st = gfv_internal_status_ptr();
  // 32-row tiles on 8 waves of 16 columns at every size.  (The 64-row form - 4 waves of 32 columns, selectable by row count through
  // round 5 - measured WORSE at every size it was tried at with the final kernels: -1.8 % / -0.6 % / -0.6 % of the step at 45 k / 51 k
  // / 75 k edge rows, 35.6 against 30.9 us at 25 k node rows (profiles/r05_thresholds.txt, r05_cfwd.txt); its instantiations were
  // removed in round 6.)
  constexpr int tg = 2;
  const int tiles = (a.M + 16 * tg - 1) / (16 * tg);
  const dim3 grid(PADD ? gfv_xcd_grid(tiles) : tiles), blk(512);
  if (lowp == 2) GFV_LAUNCH((cfwd_kernel<KT0, N0, 2, PADD, 2, RAGIN, 8, FINLN>), grid, blk, 0, stream, a, st);
  else if (lowp) GFV_LAUNCH((cfwd_kernel<KT0, N0, 2, PADD, 1, RAGIN, 8, FINLN>), grid, blk, 0, stream, a, st);
  else GFV_LAUNCH((cfwd_kernel<KT0, N0, 2, PADD, 0, RAGIN, 8, FINLN>), grid, blk, 0, stream, a, st);
}

}  // namespace

// 1: launched; 0: not a launch of this family.  lowp: 0 three products, 1 / 2 the single-product forms.  dry != 0: only tell
// whether the launch would be taken.  `args` carries `hidden` (the launcher of rowtile.hip fills it in).
int gfv_internal_cfwd_try(const gfv_rowtile_args_t* a, int lowp, hipStream_t stream, int dry) {
  // (the dispatch limits: gfv_limits.h - the tests move them to reach both tile heights at any size)
  if (!gfv_internal_status_ptr()) return 0;   // (the kernels raise their range flag there)
  const int on = gfv_internal_limit(GFV_LIM_CFWD_ON);
  const int max_m = gfv_internal_limit(GFV_LIM_CFWD_MAX_M);
  if (!on || a->nlayers != 3 || a->M > max_m || a->M < 1 || (a->flags & (GFV_CHAIN_ROW_OWNER | GFV_CHAIN_COLUMN_OWNER))) return 0;
  // the decoder's shape: no LayerNorm, a last layer of <= 16 columns, nothing else around it
  const bool dec = a->fin_op == GFV_FIN_PLAIN && a->layer[2].N >= 1 && a->layer[2].N <= 16 && !a->res[0] && !a->out_nores && !a->fin_presave &&
                   !a->fin_stats && !a->padd && a->nseg == 1 && a->layer[0].K == 128;
  if (a->in_op != GFV_IN_NONE || !a->wmax) return 0;
  if (!dec && (a->fin_op != GFV_FIN_LN || !a->fin_gamma || !a->fin_beta)) return 0;
  if (a->in_add || a->gadd || a->in_save || a->in_aux || a->ln_partial || a->gscale || a->dw_partial || a->in_stats || a->fin_aux) return 0;
  for (int l = 0; l < 3; ++l) {
    const gfv_layer_t& L = a->layer[l];
    if (!L.Wh || (L.N != 128 && !(dec && l == 2)) || L.aux || L.bias2 || (L.bias && !cf_al16(L.bias) && !(dec && l == 2))) return 0;
    if (L.op != (l < 2 ? GFV_OP_BIAS_GELU : GFV_OP_NONE)) return 0;
    if (l > 0 && L.K != 128) return 0;
    if (L.save && !cf_al16(L.save)) return 0;
  }
  if (a->layer[2].save) return 0;
  if (!a->out[0] || a->out[1] || a->out[2] || (!dec && ((a->out_ld[0] & 3) || !cf_al16(a->out[0]))) || a->res[1] || a->res[2]) return 0;
  if (a->res[0] && ((a->res_ld[0] & 3) || !cf_al16(a->res[0]))) return 0;
  if ((a->out_nores && !cf_al16(a->out_nores)) || (a->fin_presave && !cf_al16(a->fin_presave)) ||
      (!dec && (!cf_al16(a->fin_gamma) || !cf_al16(a->fin_beta))))
    return 0;
  if (a->fin_stats && (reinterpret_cast<size_t>(a->fin_stats) & 7)) return 0;
  for (int i = 0; i < a->nseg; ++i)
    if (a->seg[i].idx || a->seg[i].csr_rowptr || a->seg[i].csr_scale || a->seg[i].save) return 0;
  const int K0 = a->layer[0].K;
  int shape = -1;   // 0: [64 | 128] (NodeBlock), 1: [128] + gathered addend (factored EdgeBlock), 2: [128] plain, 3: one narrow ragged segment
  auto plain = [&](int i, int width) {
    const gfv_seg_t& s = a->seg[i];
    return s.width == width && (s.ld & 3) == 0 && cf_al16(s.ptr);
  };
  if (a->padd) {
    if (a->nseg == 1 && K0 == 128 && plain(0, 128) && a->padd_s && a->padd_r && a->padd_ld >= 256 && (a->padd_ld & 3) == 0 && cf_al16(a->padd))
      shape = 1;
  } else if (a->nseg == 2 && K0 == 192 && plain(0, 64) && plain(1, 128)) {
    shape = 0;
  } else if (dec) {
    if (plain(0, 128)) shape = 4;
  } else if (a->nseg == 1 && K0 == 128 && plain(0, 128)) {
    shape = 2;
  } else if (a->nseg == 1 && K0 <= 32 && a->seg[0].width == K0) {
    shape = 3;
  }
  if (shape < 0) return 0;
  if (dry) return 1;
  if (shape == 0) cf_launch<6, 4, false, false>(*a, lowp, stream);
  else if (shape == 1) cf_launch<4, 8, true, false>(*a, lowp, stream);
  else if (shape == 2) cf_launch<4, 8, false, false>(*a, lowp, stream);
  else if (shape == 4) cf_launch<4, 8, false, false, false>(*a, lowp, stream);
  else cf_launch<1, 2, false, true>(*a, lowp, stream);
  return 1;
}
