// gfv-build-flags: -fno-slp-vectorize
// PERSISTENT column-owner small-tile forward of the 3-layer LayerNorm MLPs (round 6; VERDICT r5 item 2, DESIGN r5 9.3).
//
// csrc/cfwd.hip runs ONE 32-row tile per workgroup and pulls the three layers' weight images (160 - 224 KB) from L2 into registers
// for every tile: at 25 k node rows that is 797 tiles x 224 KB = 180 MB of L2 -> register traffic against 85 MB of activations, and
// a CU fetches 25 - 35 B / clock from L2 (profiles/r05_launch_floor.txt) - the node-level launches sat at 1.75 x their HBM time
// waiting for weight slices.  Here a workgroup is PERSISTENT: min(CUs, tiles) workgroups of 8 waves, wave w owns output columns
// 16 w .. 16 w + 15 of every layer and keeps its slices of ALL THREE layers in registers for the whole launch (hi + lo parts:
// 8 registers per k-group: 112 for the NodeBlock's 192-deep first layer, 96 otherwise - affordable at 2 waves per SIMD), and walks
// the tiles wg, wg + grid, ...: one weight fetch per CU instead of one per tile.  Per tile the arithmetic, the fragment layouts,
// the scales and the barrier structure are cfwd.hip's (8 waves x 16 columns on a 32-row tile), so the two families agree bit for
// bit (tests/test_cfwd_gpu.py runs every shape on both).  The loader waves request the NEXT tile's input rows before the current
// tile's LayerNorm, so their latency hides behind it.
// Shapes: 0 NodeBlock [64 | 128] -> 192-deep first layer; 1 factored EdgeBlock (128 + gathered addend, XCD-aware tile ranges);
// 2 plain 128.  No narrow / ragged inputs, no decoder (those launches are short: cfwd.hip).
#include <cstdlib>

#include "tchain_kernel.h"

#include "gfv_limits.h"
int* gfv_internal_status_ptr();

namespace {

constexpr float CP_SH = 16.0f;            // = cfwd.hip CF_SH
constexpr float CP_SH_INV = 1.0f / 16.0f;
constexpr float CP_SH_LIMIT = 2048.0f;
constexpr int TG = 2, NW = 8;

__device__ __forceinline__ void cp_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int KT0>
struct CpLds {
  static constexpr int KTB = KT0 > 4 ? KT0 : 4;
  static constexpr int B0 = 0;
  static constexpr int B1 = TG * KTB * 2048;
  static constexpr int SINV = B1 + TG * 8192;
  static constexpr int LNP = SINV + TG * 64;
  static constexpr int TOTAL = LNP + TG * 16 * NW * 8;
};

template <int KT0, int N0, bool PADD, int LOWP>
__global__ __launch_bounds__(64 * NW, 2) void cfwdp_kernel(const gfv_rowtile_args_t A, int* status, int ntiles) {
  using LY = CpLds<KT0>;
  constexpr bool BF = LOWP == 2;
  __shared__ __attribute__((aligned(16))) char lds[LY::TOTAL];
  char* b0 = lds + LY::B0;
  char* b1 = lds + LY::B1;
  float* sinv = reinterpret_cast<float*>(lds + LY::SINV);
  float* lnp = reinterpret_cast<float*>(lds + LY::LNP);

  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int c0 = 16 * w + 4 * g;   // this lane's columns c0 .. c0 + 3
  const float invw = 1.0f / gfv_pow2_scale(*A.wmax);

  // tiles of this workgroup: a contiguous range per workgroup with PADD (the gathered addend rows of neighbouring edges are the
  // same node rows: they stay in this XCD's L2 - the XCD-aware mapping of cfwd.hip, range by range), else strided
  int t_first, t_step, t_end;
  if (PADD) {
    const int wg = gfv_xcd_tile(blockIdx.x, gridDim.x);
    const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    t_first = wg * per;
    t_end = min(ntiles, t_first + per);
    t_step = 1;
  } else {
    t_first = blockIdx.x;
    t_end = ntiles;
    t_step = gridDim.x;
  }
  if (t_first >= t_end) return;

  // ---- the wave's weight slices of all three layers: resident for the whole launch ----
  gfv_f16x8 wh0[KT0], wl0[KT0], wh1[4], wl1[4], wh2[4], wl2[4];
  {
    const gfv_f16x8* im0 = reinterpret_cast<const gfv_f16x8*>(A.layer[0].Wh) + (size_t)w * 128 + lane;
#pragma unroll
    for (int T = 0; T < KT0; ++T) {
      wh0[T] = im0[T * 1024];
      if (!LOWP) wl0[T] = im0[T * 1024 + 64];
    }
    const gfv_f16x8* im1 = reinterpret_cast<const gfv_f16x8*>(A.layer[1].Wh) + (size_t)w * 128 + lane;
    const gfv_f16x8* im2 = reinterpret_cast<const gfv_f16x8*>(A.layer[2].Wh) + (size_t)w * 128 + lane;
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      wh1[T] = im1[T * 1024];
      wh2[T] = im2[T * 1024];
      if (!LOWP) { wl1[T] = im1[T * 1024 + 64]; wl2[T] = im2[T * 1024 + 64]; }
    }
  }
  float4 bias[3], gam, bet;
#pragma unroll
  for (int l = 0; l < 3; ++l) bias[l] = A.layer[l].bias ? ld4(A.layer[l].bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
  gam = ld4(A.fin_gamma + c0);
  bet = ld4(A.fin_beta + c0);
  const int hcols = (A.hidden > 0 && A.hidden < 128) ? A.hidden : 128;
  float mabs = 0.f;

  // input rows of a tile (loader waves w < TG: wave q loads group q), requested one tile ahead
  float v[2 * KT0][4];
  auto request_rows = [&](int tile) {
    const int row = min(tile * (16 * TG) + 16 * w + j, A.M - 1);
    const float* p0 = A.seg[0].ptr + (size_t)row * A.seg[0].ld + 4 * g;
    const float* p1 = p0;
    if (N0 < 2 * KT0) p1 = A.seg[1].ptr + (size_t)row * A.seg[1].ld + 4 * g;
#pragma unroll
    for (int u = 0; u < 2 * KT0; ++u) {
      const float4 t = ld4(u < N0 ? p0 + 16 * u : p1 + 16 * (u - N0));
      v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
    }
  };
  if (w < TG) request_rows(t_first);

  floatx4 acc[TG];
  auto mma = [&](const char* xbuf, const gfv_f16x8* wh, const gfv_f16x8* wl, auto ktc) {
    constexpr int KT = decltype(ktc)::value;
#pragma unroll
    for (int q = 0; q < TG; ++q) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int T = 0; T < KT; ++T) {
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        const gfv_f16x8* f = reinterpret_cast<const gfv_f16x8*>(xbuf + (size_t)(q * KT + T) * 2048) + lane;
        const gfv_f16x8 xh = f[0];
        if (!LOWP) {
          const gfv_f16x8 xl = f[64];
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[T], xh, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[T], xl, acc[q], 0, 0, 0);
        }
        acc[q] = gfv_mma_hh<BF>(wh[T], xh, acc[q]);
      }
    }
  };

  for (int tile = t_first; tile < t_end; tile += t_step) {
    const int row0 = tile * (16 * TG);
    const int ngt = min(TG, (A.M - row0 + 15) >> 4);
    // gathered addend rows of the first pre-activation: index, then row - two round trips, started now
    float4 ps[PADD ? TG : 1], pr[PADD ? TG : 1];
    if (PADD) {
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        const int row = min(row0 + 16 * q + j, A.M - 1);
        ps[q] = ld4(A.padd + (size_t)A.padd_s[row] * A.padd_ld + c0);
        pr[q] = ld4(A.padd + (size_t)A.padd_r[row] * A.padd_ld + 128 + c0);
      }
    }
    // ---- input rows -> row scale -> fragments ----
    if (w < TG) {
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
    cp_barrier();

    // hidden-layer epilogue: v = acc / scales + bias (+ addend) -> saved; gelu(v) -> the next layer's fragments (half of k-group w >> 1)
    auto hidden = [&](int layer, char* xout, float* save) {
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        const int row = row0 + 16 * q + j;
        const bool live = q < ngt && row < A.M;
        const float si = layer == 0 ? sinv[q * 16 + j] : CP_SH_INV;
        const float4 b = bias[layer];
        float z[4] = {(acc[q][0] * si) * invw + b.x, (acc[q][1] * si) * invw + b.y, (acc[q][2] * si) * invw + b.z,
                      (acc[q][3] * si) * invw + b.w};
        if (PADD && layer == 0) {
          z[0] += ps[PADD ? q : 0].x + pr[PADD ? q : 0].x; z[1] += ps[PADD ? q : 0].y + pr[PADD ? q : 0].y;
          z[2] += ps[PADD ? q : 0].z + pr[PADD ? q : 0].z; z[3] += ps[PADD ? q : 0].w + pr[PADD ? q : 0].w;
        }
        if (save && live) st4(save + (size_t)row * 128 + c0, z);
        const gfv_f2 g01 = gfv_gelu2(gfv_f2{z[0], z[1]}), g23 = gfv_gelu2(gfv_f2{z[2], z[3]});
        const float a[4] = {g01.x, g01.y, g23.x, g23.y};
        float m = max3_abs(max3_abs(0.f, a[0], a[1]), a[2], a[3]);
        mabs = fmaxf(mabs, live ? m : 0.f);
        unsigned h0, h1, l0, l1;
        gfv_split_pair_t<BF>(a[0] * CP_SH, a[1] * CP_SH, h0, l0);
        gfv_split_pair_t<BF>(a[2] * CP_SH, a[3] * CP_SH, h1, l1);
        char* dst = xout + (size_t)(q * 4 + (w >> 1)) * 2048 + lane * 16 + (w & 1) * 8;
        *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
        if (!LOWP) *reinterpret_cast<uint2*>(dst + 1024) = make_uint2(l0, l1);
      }
    };

    // ---- layer 0: b0 -> b1 ----
    mma(b0, wh0, wl0, std::integral_constant<int, KT0>{});
    hidden(0, b1, A.layer[0].save);
    cp_barrier();
    // ---- layer 1: b1 -> b0 ----
    mma(b1, wh1, wl1, std::integral_constant<int, 4>{});
    hidden(1, b0, A.layer[1].save);
    // the residual rows: requested ahead of the last layer
    float4 rres[TG];
    if (A.res[0]) {
#pragma unroll
      for (int q = 0; q < TG; ++q) rres[q] = ld4(A.res[0] + (size_t)min(row0 + 16 * q + j, A.M - 1) * A.res_ld[0] + c0);
    }
    cp_barrier();
    // ---- layer 2: b0 -> y; LayerNorm (statistics over all 128 columns, corrected for a narrower model: cfwd.hip) ----
    mma(b0, wh2, wl2, std::integral_constant<int, 4>{});
    // the NEXT tile's input rows: in flight behind the LayerNorm (b0 is free again behind the barrier below)
    const int tnext = tile + t_step;
    if (w < TG && tnext < t_end) request_rows(tnext);
    float y[TG][4];
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const int row = row0 + 16 * q + j;
      const bool live = q < ngt && row < A.M;
      const float4 b = bias[2];
      float z[4] = {(acc[q][0] * CP_SH_INV) * invw + b.x, (acc[q][1] * CP_SH_INV) * invw + b.y, (acc[q][2] * CP_SH_INV) * invw + b.z,
                    (acc[q][3] * CP_SH_INV) * invw + b.w};
      if (A.fin_presave && live) st4(A.fin_presave + (size_t)row * 128 + c0, z);
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        y[q][r] = z[r];
        s += z[r];
      }
      const float mw = row_sum(s) * (1.0f / 16.0f);
      float m2 = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = y[q][e] - mw;
        m2 += d * d;
      }
      m2 = row_sum(m2);
      if (g == 0) *reinterpret_cast<float2*>(lnp + ((q * 16 + j) * NW + w) * 2) = make_float2(mw, m2);
    }
    cp_barrier();
    {
      const float inv_h = 1.0f / (float)hcols, npad = (float)(128 - hcols);
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        const int row = row0 + 16 * q + j;
        const bool live = q < ngt && row < A.M;
        const float4* pp = reinterpret_cast<const float4*>(lnp + (q * 16 + j) * 2 * NW);
        const float4 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];
        const float m128 = (((p0.x + p0.z) + (p1.x + p1.z)) + ((p2.x + p2.z) + (p3.x + p3.z))) * 0.125f;
        const float e0 = p0.x - m128, e1 = p0.z - m128, e2 = p1.x - m128, e3 = p1.z - m128, e4 = p2.x - m128, e5 = p2.z - m128,
                    e6 = p3.x - m128, e7 = p3.z - m128;
        const float m2a = (((p0.y + p0.w) + (p1.y + p1.w)) + ((p2.y + p2.w) + (p3.y + p3.w))) +
                          16.0f * (((e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3)) + ((e4 * e4 + e5 * e5) + (e6 * e6 + e7 * e7)));
        const float mean = hcols == 128 ? m128 : (m128 * 128.0f) * inv_h;
        const float dm = m128 - mean;
        const float m2 = hcols == 128 ? m2a : (m2a + 128.0f * dm * dm) - npad * (mean * mean);
        const float rstd = rsqrtf(m2 * inv_h + 1e-5f);   // nn.LayerNorm eps (EPD.py:32)
        if (A.fin_stats && live && w == 0 && g == 0) *reinterpret_cast<float2*>(A.fin_stats + 2 * (size_t)row) = make_float2(mean, rstd);
        float o[4] = {(y[q][0] - mean) * rstd * gam.x + bet.x, (y[q][1] - mean) * rstd * gam.y + bet.y,
                      (y[q][2] - mean) * rstd * gam.z + bet.z, (y[q][3] - mean) * rstd * gam.w + bet.w};
        if (live) {
          if (A.out_nores) st4(A.out_nores + (size_t)row * 128 + c0, o);
          if (A.res[0]) { o[0] += rres[q].x; o[1] += rres[q].y; o[2] += rres[q].z; o[3] += rres[q].w; }
          st4(A.out[0] + (size_t)row * A.out_ld[0] + c0, o);
        }
      }
    }
    // (no barrier at the end of the iteration: the loader waves rewrite b0 / sinv only behind the barrier above, which every wave
    // passes after its last read of b0 (layer 2) and of sinv (layer 0); lnp is rewritten three barriers from here)
  }
  if (mabs > CP_SH_LIMIT) atomicOr(status, 2);   // GFV_FLAG_CHAIN_RANGE
}

inline bool cp_al16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

template <int KT0, int N0, bool PADD>
void cp_launch(const gfv_rowtile_args_t& a, int lowp, int wgs, hipStream_t stream) {
  int* st = gfv_internal_status_ptr();
  const int tiles = (a.M + 16 * TG - 1) / (16 * TG);
  int grid = tiles < wgs ? tiles : wgs;
  if (PADD) grid = gfv_xcd_grid(grid);
  if (lowp == 2) GFV_LAUNCH((cfwdp_kernel<KT0, N0, PADD, 2>), dim3(grid), dim3(64 * NW), 0, stream, a, st, tiles);
  else if (lowp) GFV_LAUNCH((cfwdp_kernel<KT0, N0, PADD, 1>), dim3(grid), dim3(64 * NW), 0, stream, a, st, tiles);
  else GFV_LAUNCH((cfwdp_kernel<KT0, N0, PADD, 0>), dim3(grid), dim3(64 * NW), 0, stream, a, st, tiles);
}

}  // namespace

// 1: launched; 0: not a launch of this family.  Same contract as gfv_internal_cfwd_try (cfwd.hip), which rowtile.hip asks next.
int gfv_internal_cfwdp_try(const gfv_rowtile_args_t* a, int lowp, hipStream_t stream, int dry) {
  if (!gfv_internal_status_ptr()) return 0;
  const int min_m = gfv_internal_limit(GFV_LIM_CFWDP_MIN_M), max_m = gfv_internal_limit(GFV_LIM_CFWDP_MAX_M);
  if (a->nlayers != 3 || a->M < min_m || a->M > max_m || (a->flags & (GFV_CHAIN_ROW_OWNER | GFV_CHAIN_COLUMN_OWNER))) return 0;
  if (a->in_op != GFV_IN_NONE || !a->wmax || a->fin_op != GFV_FIN_LN || !a->fin_gamma || !a->fin_beta) return 0;
  if (a->in_add || a->gadd || a->in_save || a->in_aux || a->ln_partial || a->gscale || a->dw_partial || a->in_stats || a->fin_aux) return 0;
  for (int l = 0; l < 3; ++l) {
    const gfv_layer_t& L = a->layer[l];
    if (!L.Wh || L.N != 128 || L.aux || L.bias2 || (L.bias && !cp_al16(L.bias))) return 0;
    if (L.op != (l < 2 ? GFV_OP_BIAS_GELU : GFV_OP_NONE)) return 0;
    if (l > 0 && L.K != 128) return 0;
    if (L.save && !cp_al16(L.save)) return 0;
  }
  if (a->layer[2].save) return 0;
  if (!a->out[0] || a->out[1] || a->out[2] || (a->out_ld[0] & 3) || !cp_al16(a->out[0]) || a->res[1] || a->res[2]) return 0;
  if (a->res[0] && ((a->res_ld[0] & 3) || !cp_al16(a->res[0]))) return 0;
  if ((a->out_nores && !cp_al16(a->out_nores)) || (a->fin_presave && !cp_al16(a->fin_presave)) || !cp_al16(a->fin_gamma) || !cp_al16(a->fin_beta))
    return 0;
  if (a->fin_stats && (reinterpret_cast<size_t>(a->fin_stats) & 7)) return 0;
  for (int i = 0; i < a->nseg; ++i)
    if (a->seg[i].idx || a->seg[i].csr_rowptr || a->seg[i].csr_scale || a->seg[i].save) return 0;
  const int K0 = a->layer[0].K;
  auto plain = [&](int i, int width) {
    const gfv_seg_t& s = a->seg[i];
    return s.width == width && (s.ld & 3) == 0 && cp_al16(s.ptr);
  };
  int shape = -1;
  if (a->padd) {
    if (a->nseg == 1 && K0 == 128 && plain(0, 128) && a->padd_s && a->padd_r && a->padd_ld >= 256 && (a->padd_ld & 3) == 0 && cp_al16(a->padd))
      shape = 1;
  } else if (a->nseg == 2 && K0 == 192 && plain(0, 64) && plain(1, 128)) {
    shape = 0;
  } else if (a->nseg == 1 && K0 == 128 && plain(0, 128)) {
    shape = 2;
  }
  if (shape < 0) return 0;
  if (shape == 1 && !gfv_internal_limit(GFV_LIM_CFWDP_EDGE)) return 0;
  if (dry) return 1;
  const int wgs = gfv_internal_limit(GFV_LIM_CFWDP_WGS);
  if (shape == 0) cp_launch<6, 4, false>(*a, lowp, wgs, stream);
  else if (shape == 1) cp_launch<4, 8, true>(*a, lowp, wgs, stream);
  else cp_launch<4, 8, false>(*a, lowp, wgs, stream);
  return 1;
}
