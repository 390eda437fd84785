// Column-owner persistent GEMM chain (gfx950): the second kernel family behind gfv_rowtile_chain (contract: include/gfv.h)
// for the big 3-layer MLP launches of the GnBlocks (EPD.py:10-33 build_mlp inside blocks.py EdgeBlock / NodeBlock) in
// the split-fp16 product form.
//
// tchain_kernel.h gives every wave 16 ROWS and streams all three layers' weight images (192 KB) through LDS for every
// 64-row tile: 12 slice barriers per tile, each behind an L2 round trip, and the tile's life is latency, not work
// (profiles/r02_sq_counters.txt).  Here the roles are swapped:
//
//   * ONE workgroup of 8 waves per CU, persistent over a contiguous range of 16-row groups.  Wave w owns output COLUMNS
//     16 w .. 16 w + 15 of every 128-wide layer and keeps ITS slice of all three weight images - the A operands
//     W[16 w + i][k] of v_mfma_f32_16x16x32_f16, hi and lo parts, 32 VGPRs per 128-deep layer - in registers for the whole
//     launch.  Weights are read once per workgroup (256 x 192 KB per launch instead of one 192 KB stream per 64 rows).
//   * LDS holds only activations, already in MFMA B-fragment form ([group][k-group T][part][lane] x 16 B: what
//     to_halves() of the row-owner kernel builds in registers).  A tile is up to TG groups of 16 rows; per layer every
//     wave reads all of the tile's fragments (one conflict-free ds_read_b128 per fragment), runs 12 MFMAs per group
//     against its resident weights, applies the element ops to its 16 columns and writes its 8-byte share of the next
//     layer's fragments (the columns a wave produces are exactly half a k-group of the next layer: T' = w >> 1,
//     slots 4 (w & 1) .. + 3).  One barrier per layer, four per tile of 128 rows.
//   * Rows enter through "loader" roles: wave w < (groups in the tile) loads the 16 full rows of group w one tile ahead
//     (global -> registers, no wait until the tile is consumed), takes the row's power-of-two scale (exact), splits
//     and parks the fragments.  Hidden activations (GELU outputs) are split after a FIXED power-of-two scale CC_SH: a row scale would
//     need the row maximum over all eight waves (a second barrier per layer), and the split has 2^16 of slack - a hidden
//     row with max |a| in [2^-4, 2^11] keeps every product at fp32 accuracy; beyond 2^11 the status flag
//     GFV_FLAG_CHAIN_RANGE is raised (the values still convert up to 4095).
//   * LayerNorm statistics of a row are spread over the eight waves: each leaves (mean, M2) of its 16 columns in LDS, after
//     the barrier every lane combines the eight pairs (Chan's parallel form of the two-pass variance).
//
// Element-op semantics, argument struct and saved tensors are those of tchain_kernel.h (same launches, same results to
// rounding: the summation order inside a dot product differs, and hidden activations carry the fixed scale).
#pragma once
#include "tchain_kernel.h"

namespace {

constexpr int CC_W = 8;                  // waves per workgroup
constexpr float CC_SH = 16.0f;           // fixed scale of hidden activations ahead of the fp16 split
constexpr float CC_SH_INV = 1.0f / 16.0f;
constexpr float CC_SH_LIMIT = 2048.0f;   // |a| beyond this raises GFV_FLAG_CHAIN_RANGE

// LDS carve (bytes).  XIN: the tile's input fragments (KT0 k-groups), later the second hidden layer's (4 k-groups);
// XMID: the first hidden layer's fragments, later each wave's stash of its last-layer values (TG x 1 KB per wave).
template <int KT0, int TG>
struct CcLds {
  static constexpr int XIN = 0;
  static constexpr int XMID = XIN + TG * KT0 * 2048;
  static constexpr int SINV = XMID + TG * 8192;          // float [TG][16]: 1 / row scale of the input rows
  static constexpr int IDXS = SINV + TG * 64;            // int   [TG][16]: gather rows of the first-layer addend (sender)
  static constexpr int IDXR = IDXS + TG * 64;            //                                            (receiver)
  static constexpr int LNP = IDXR + TG * 64;             // float2 [TG][16][8]: (mean, M2) of a row's 16 columns per wave
  static constexpr int TOTAL = LNP + TG * 16 * 8 * 8;
};

__device__ __forceinline__ void cc_barrier() {
  // LDS only: the tile-ahead global loads and the epilogue stores stay in flight across it (a __syncthreads() drains vmcnt)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct CcCtx {
  int w, lane, j, g, col0;   // wave, lane, row inside a group, lane group, first of this lane's 4 columns
  int M;
  int row0;                  // first row of the tile
  int ngt;                   // live groups of the tile
  float invw;                // 1 / weight scale
  float mabs;                // running max |hidden activation| (range flag)
};

// one pair of groups against this wave's resident weights: acc_q += W[16 w + i][k] x_q[row][k] over KT k-groups
template <int KT, bool LOWP>
__device__ __forceinline__ void cc_mma_pair(const char* xbuf, int pair, const gfv_f16x8 (&wh)[KT], const gfv_f16x8 (&wl)[KT],
                                            int lane, floatx4& a0, floatx4& a1) {
  const gfv_f16x8* f0 = reinterpret_cast<const gfv_f16x8*>(xbuf + (size_t)(2 * pair) * KT * 2048) + lane;
  const gfv_f16x8* f1 = f0 + KT * 128;
  a0 = floatx4{0.f, 0.f, 0.f, 0.f};
  a1 = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int T = 0; T < KT; ++T) {
    const gfv_f16x8 xh0 = f0[(2 * T) * 64], xh1 = f1[(2 * T) * 64];
    if (!LOWP) {
      const gfv_f16x8 xl0 = f0[(2 * T + 1) * 64], xl1 = f1[(2 * T + 1) * 64];
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[T], xh0, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[T], xh1, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[T], xl0, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[T], xl1, a1, 0, 0, 0);
    }
    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[T], xh0, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[T], xh1, a1, 0, 0, 0);
  }
}

// this lane's 4 values of one row -> its 8-byte share of the next layer's fragments (k-group w >> 1, half w & 1)
__device__ __forceinline__ void cc_put_frag(char* xbuf, int q, const CcCtx& c, const float (&a)[4], float scale) {
  unsigned h0, h1, l0, l1;
  gfv_split_pair(a[0] * scale, a[1] * scale, h0, l0);
  gfv_split_pair(a[2] * scale, a[3] * scale, h1, l1);
  char* dst = xbuf + (size_t)((q * 4 + (c.w >> 1)) * 2) * 1024 + c.lane * 16 + (c.w & 1) * 8;
  *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(dst + 1024) = make_uint2(l0, l1);
}

struct CcAdd {   // prefetched first-layer addend rows of one pair of groups (factored EdgeBlock: (W1a nb)[s] + (W1b nb)[r])
  float4 s0, r0, s1, r1;
};
__device__ __forceinline__ CcAdd cc_padd_load(const gfv_rowtile_args_t& A, const CcCtx& c, const int* idxs, const int* idxr,
                                              int pair) {
  CcAdd p;
  const int q0 = 2 * pair, q1 = 2 * pair + 1;
  const float* base = A.padd + c.col0;
  p.s0 = ld4(base + (size_t)idxs[q0 * 16 + c.j] * A.padd_ld);
  p.r0 = ld4(base + (size_t)idxr[q0 * 16 + c.j] * A.padd_ld + 128);
  p.s1 = ld4(base + (size_t)idxs[q1 * 16 + c.j] * A.padd_ld);
  p.r1 = ld4(base + (size_t)idxr[q1 * 16 + c.j] * A.padd_ld + 128);
  return p;
}

// hidden-layer epilogue of one group (forward form, GFV_OP_BIAS_GELU): v = acc / scales + bias (+ addend) - handed back for the
// save, which the caller issues after the math of both groups of a pair (a predicated store ends a basic block: the MFMAs
// of the next pair and this arithmetic are to stay in one); a = gelu(v) -> fragments of the next layer
template <int L, bool PADD>
__device__ __forceinline__ void cc_hidden_fwd(CcCtx& c, int q, const floatx4& acc, const float4& bias, const float* sinv,
                                              const float4& ps, const float4& pr, char* xout, float (&v)[4]) {
  if (L == 0) {
    const float si = sinv[q * 16 + c.j];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (acc[r] * si) * c.invw;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (acc[r] * CC_SH_INV) * c.invw;
  }
  v[0] += bias.x; v[1] += bias.y; v[2] += bias.z; v[3] += bias.w;
  if (L == 0 && PADD) {
    v[0] += ps.x + pr.x; v[1] += ps.y + pr.y; v[2] += ps.z + pr.z; v[3] += ps.w + pr.w;
  }
  float a[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) a[r] = gfv_gelu(v[r]);
  const float mq = max3_abs(max3_abs(0.f, a[0], a[1]), a[2], a[3]);
  c.mabs = fmaxf(c.mabs, q < c.ngt ? mq : 0.f);   // (the dead groups of a partial tile hold whatever LDS held)
  cc_put_frag(xout, q, c, a, CC_SH);
}
__device__ __forceinline__ void cc_save_pair(float* save, const CcCtx& c, int p, const float (&v0)[4], const float (&v1)[4]) {
  const int r0 = c.row0 + 32 * p + c.j, r1 = r0 + 16;
  if (save && 2 * p < c.ngt && r0 < c.M) st4(save + (size_t)r0 * 128 + c.col0, v0);
  if (save && 2 * p + 1 < c.ngt && r1 < c.M) st4(save + (size_t)r1 * 128 + c.col0, v1);
}

// the tile-ahead input rows of a loader wave: 2 KT0 pieces of 16 columns (float4 per lane), concatenated segments
template <int KT0>
struct CcPre {
  float4 v[2 * KT0];
  int is, ir;   // gather rows of the first-layer addend for this lane's row
};

// N0: 16-column pieces of segment 0 (the rest of the 2 KT0 pieces come from segment 1) - compile-time, so that the loads
// are one straight run (a run-time segment lookup per piece compiled into a branch per load)
template <int KT0, int N0>
__device__ __forceinline__ void cc_prefetch(const gfv_rowtile_args_t& A, const CcCtx& c, int row0, CcPre<KT0>& pre) {
  // (unconditional loads from a clamped row: a register array filled under a branch is parked in scratch by the compiler)
  const int row = min(row0 + 16 * c.w + c.j, A.M - 1);
  const int* i0 = A.seg[0].idx;
  const float* p0 = A.seg[0].ptr + (size_t)(i0 ? i0[row] : row) * A.seg[0].ld + 4 * c.g;
  const float* p1 = p0;
  if (N0 < 2 * KT0) {
    const int* i1 = A.seg[1].idx;
    p1 = A.seg[1].ptr + (size_t)(i1 ? i1[row] : row) * A.seg[1].ld + 4 * c.g;
  }
#pragma unroll
  for (int u = 0; u < 2 * KT0; ++u) pre.v[u] = ld4(u < N0 ? p0 + 16 * u : p1 + 16 * (u - N0));
  pre.is = A.padd ? A.padd_s[row] : 0;
  pre.ir = A.padd ? A.padd_r[row] : 0;
}

// loader: the prefetched 16 rows -> row scale, fragments, gather rows in LDS (dead groups: zero gather rows)
template <int KT0>
__device__ __forceinline__ void cc_park_input(const CcCtx& c, const CcPre<KT0>& pre, char* xin, float* sinv, int* idxs, int* idxr) {
  float m0 = 0.f, m1 = 0.f;
#pragma unroll
  for (int u = 0; u < 2 * KT0; ++u) {
    m0 = max3_abs(m0, pre.v[u].x, pre.v[u].y);
    m1 = max3_abs(m1, pre.v[u].z, pre.v[u].w);
  }
  const float s = gfv_pow2_scale(row_max4(max3_abs(0.f, m0, m1)));
  const bool livegrp = c.w < c.ngt;
  if (c.g == 0) {
    sinv[c.w * 16 + c.j] = 1.0f / s;
    idxs[c.w * 16 + c.j] = livegrp ? pre.is : 0;
    idxr[c.w * 16 + c.j] = livegrp ? pre.ir : 0;
  }
  if (livegrp) {
    gfv_uint4* dst = reinterpret_cast<gfv_uint4*>(xin + (size_t)c.w * KT0 * 2048) + c.lane;
#pragma unroll
    for (int T = 0; T < KT0; ++T) {
      const float e[8] = {pre.v[2 * T].x * s,     pre.v[2 * T].y * s,     pre.v[2 * T].z * s,     pre.v[2 * T].w * s,
                          pre.v[2 * T + 1].x * s, pre.v[2 * T + 1].y * s, pre.v[2 * T + 1].z * s, pre.v[2 * T + 1].w * s};
      gfv_uint4 hi, lo;
      gfv_split8(e, hi, lo);
      dst[(2 * T) * 64] = hi;
      dst[(2 * T + 1) * 64] = lo;
    }
  }
}

// Forward form: 3 layers (bias + GELU, bias + GELU, bias), LayerNorm, optional residual; segments of 32-multiples wide
// (plain or row-gathered), optional gathered first-layer addend.  KT0 = k-groups of the first layer (K / 32), N0 = 16-column
// pieces of the first segment,
// TG = groups of 16 rows per tile (<= 8: one loader wave per group).
template <int KT0, int N0, int TG, bool PADD, bool LOWP>
__global__ __launch_bounds__(64 * CC_W, 2) void colchain_fwd_kernel(const gfv_rowtile_args_t A, int* status) {
  static_assert(TG <= CC_W && (TG & 1) == 0, "one loader wave per group, groups in pairs");
  using LY = CcLds<KT0, TG>;
  __shared__ __attribute__((aligned(16))) char lds[LY::TOTAL];
  char* xin = lds + LY::XIN;
  char* xmid = lds + LY::XMID;
  float* sinv = reinterpret_cast<float*>(lds + LY::SINV);
  int* idxs = reinterpret_cast<int*>(lds + LY::IDXS);
  int* idxr = reinterpret_cast<int*>(lds + LY::IDXR);
  float* lnp = reinterpret_cast<float*>(lds + LY::LNP);

  CcCtx c;
  c.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  c.lane = threadIdx.x & 63;
  c.j = c.lane & 15;
  c.g = c.lane >> 4;
  c.col0 = 16 * c.w + 4 * c.g;
  c.M = A.M;
  c.mabs = 0.f;
  c.invw = 1.0f / gfv_pow2_scale(*A.wmax);

  // this workgroup's groups: a contiguous range, XCD-aware (neighbouring ranges gather the same rows: one L2)
  const int nwg = gridDim.x;
  const int wg = (nwg & 7) == 0 ? (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int NG = (A.M + 15) >> 4;
  const int g_beg = (int)((long)NG * wg / nwg), g_end = (int)((long)NG * (wg + 1) / nwg);
  if (g_beg >= g_end) return;

  // tile-ahead loads of the first tile go out before anything else
  CcPre<KT0> pre;
  cc_prefetch<KT0, N0>(A, c, 16 * g_beg, pre);

  // resident weights: this wave's n-tile of every layer's image ([pass][T][nt][part][lane] x 16 B, include/gfv.h)
  gfv_f16x8 wh0[KT0], wl0[KT0], wh1[4], wl1[4], wh2[4], wl2[4];
  {
    const gfv_f16x8* i0 = reinterpret_cast<const gfv_f16x8*>(A.layer[0].Wh) + (size_t)c.w * 128 + c.lane;
    const gfv_f16x8* i1 = reinterpret_cast<const gfv_f16x8*>(A.layer[1].Wh) + (size_t)c.w * 128 + c.lane;
    const gfv_f16x8* i2 = reinterpret_cast<const gfv_f16x8*>(A.layer[2].Wh) + (size_t)c.w * 128 + c.lane;
#pragma unroll
    for (int T = 0; T < KT0; ++T) { wh0[T] = i0[T * 1024]; wl0[T] = i0[T * 1024 + 64]; }
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      wh1[T] = i1[T * 1024]; wl1[T] = i1[T * 1024 + 64];
      wh2[T] = i2[T * 1024]; wl2[T] = i2[T * 1024 + 64];
    }
  }
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 b0 = A.layer[0].bias ? ld4(A.layer[0].bias + c.col0) : zero4;
  const float4 b1 = A.layer[1].bias ? ld4(A.layer[1].bias + c.col0) : zero4;
  const float4 b2 = A.layer[2].bias ? ld4(A.layer[2].bias + c.col0) : zero4;
  const float4 gam = ld4(A.fin_gamma + c.col0);
  const float4 bet = ld4(A.fin_beta + c.col0);
  const float hsc = CC_SH_INV;

  for (int t0 = g_beg; t0 < g_end; t0 += TG) {
    c.row0 = 16 * t0;
    c.ngt = min(TG, g_end - t0);
    const int np = (c.ngt + 1) >> 1;
    // ---- P0: the prefetched rows become the tile's input fragments ----
    if (c.w < TG) cc_park_input<KT0>(c, pre, xin, sinv, idxs, idxr);
    cc_barrier();
    // next tile's rows: in flight through the whole tile (after the last tile: every lane re-reads row M - 1, a few cached
    // lines - the loads stay unconditional, a register array filled under a branch is parked in scratch by the compiler)
    cc_prefetch<KT0, N0>(A, c, t0 + TG < g_end ? 16 * (t0 + TG) : A.M, pre);

    // ---- P1: layer 0, xin -> xmid ----
    {
      floatx4 a0, a1;
      cc_mma_pair<KT0, LOWP>(xin, 0, wh0, wl0, c.lane, a0, a1);
      CcAdd pn;
      pn.s0 = pn.r0 = pn.s1 = pn.r1 = zero4;
      if (PADD) pn = cc_padd_load(A, c, idxs, idxr, 0);
      for (int p = 0; p < np; ++p) {
        const int pnext = min(p + 1, TG / 2 - 1);
        const CcAdd pc = pn;
        if (PADD) pn = cc_padd_load(A, c, idxs, idxr, pnext);
        floatx4 n0, n1;
        cc_mma_pair<KT0, LOWP>(xin, pnext, wh0, wl0, c.lane, n0, n1);
        float v0[4], v1[4];
        cc_hidden_fwd<0, PADD>(c, 2 * p, a0, b0, sinv, pc.s0, pc.r0, xmid, v0);
        cc_hidden_fwd<0, PADD>(c, 2 * p + 1, a1, b0, sinv, pc.s1, pc.r1, xmid, v1);
        cc_save_pair(A.layer[0].save, c, p, v0, v1);
        a0 = n0; a1 = n1;
      }
    }
    cc_barrier();
    // ---- P2: layer 1, xmid -> xin ----
    {
      floatx4 a0, a1;
      cc_mma_pair<4, LOWP>(xmid, 0, wh1, wl1, c.lane, a0, a1);
      for (int p = 0; p < np; ++p) {
        const int pnext = min(p + 1, TG / 2 - 1);
        floatx4 n0, n1;
        cc_mma_pair<4, LOWP>(xmid, pnext, wh1, wl1, c.lane, n0, n1);
        float v0[4], v1[4];
        cc_hidden_fwd<1, false>(c, 2 * p, a0, b1, sinv, zero4, zero4, xin, v0);
        cc_hidden_fwd<1, false>(c, 2 * p + 1, a1, b1, sinv, zero4, zero4, xin, v1);
        cc_save_pair(A.layer[1].save, c, p, v0, v1);
        a0 = n0; a1 = n1;
      }
    }
    cc_barrier();
    // ---- P3: layer 2, xin -> values; LayerNorm partials; the values wait in this wave's stash (xmid is free) ----
    float4* stash = reinterpret_cast<float4*>(xmid + (size_t)c.w * TG * 1024) + c.lane;
    {
      floatx4 a0, a1;
      cc_mma_pair<4, LOWP>(xin, 0, wh2, wl2, c.lane, a0, a1);
      for (int p = 0; p < np; ++p) {
        const int pnext = min(p + 1, TG / 2 - 1);
        floatx4 n0, n1;
        cc_mma_pair<4, LOWP>(xin, pnext, wh2, wl2, c.lane, n0, n1);
        float y0[4], y1[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int q = 2 * p + h;
          const floatx4& acc = h ? a1 : a0;
          float (&y)[4] = h ? y1 : y0;
#pragma unroll
          for (int r = 0; r < 4; ++r) y[r] = (acc[r] * hsc) * c.invw;
          y[0] += b2.x; y[1] += b2.y; y[2] += b2.z; y[3] += b2.w;
          const float mw = row_sum((y[0] + y[1]) + (y[2] + y[3])) * 0.0625f;
          const float d0 = y[0] - mw, d1 = y[1] - mw, d2 = y[2] - mw, d3 = y[3] - mw;
          const float m2 = row_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
          if (c.g == 0) *reinterpret_cast<float2*>(lnp + ((q * 16 + c.j) * 8 + c.w) * 2) = make_float2(mw, m2);
          stash[q * 64] = make_float4(y[0], y[1], y[2], y[3]);
        }
        cc_save_pair(A.fin_presave, c, p, y0, y1);
        a0 = n0; a1 = n1;
      }
    }
    cc_barrier();   // (also the write-after-read guard of xin for the next tile's P0)
    {
      // ---- P4: LayerNorm over the eight waves' partials, affine, residual, stores ----
      for (int q = 0; q < c.ngt; ++q) {
        const int row = c.row0 + 16 * q + c.j;
        const bool live = row < c.M;
        const int rc = live ? row : c.M - 1;
        float4 rv = zero4;
        if (A.res[0]) rv = ld4(A.res[0] + (size_t)rc * A.res_ld[0] + c.col0);
        const float4* pp = reinterpret_cast<const float4*>(lnp + (q * 16 + c.j) * 16);
        const float4 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];   // (mean, M2) x 8 waves
        const float mean = (((p0.x + p0.z) + (p1.x + p1.z)) + ((p2.x + p2.z) + (p3.x + p3.z))) * 0.125f;
        const float e0 = p0.x - mean, e1 = p0.z - mean, e2 = p1.x - mean, e3 = p1.z - mean, e4 = p2.x - mean,
                    e5 = p2.z - mean, e6 = p3.x - mean, e7 = p3.z - mean;
        const float m2 = (((p0.y + p0.w) + (p1.y + p1.w)) + ((p2.y + p2.w) + (p3.y + p3.w))) +
                         16.0f * (((e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3)) + ((e4 * e4 + e5 * e5) + (e6 * e6 + e7 * e7)));
        const float rstd = rsqrtf(m2 * 0.0078125f + 1e-5f);   // nn.LayerNorm eps (EPD.py:32)
        const float4 yv = stash[q * 64];
        float o[4] = {(yv.x - mean) * rstd * gam.x + bet.x, (yv.y - mean) * rstd * gam.y + bet.y,
                      (yv.z - mean) * rstd * gam.z + bet.z, (yv.w - mean) * rstd * gam.w + bet.w};
        if (live) {
          if (A.out_nores) st4(A.out_nores + (size_t)row * 128 + c.col0, o);
          o[0] += rv.x; o[1] += rv.y; o[2] += rv.z; o[3] += rv.w;
          st4(A.out[0] + (size_t)row * A.out_ld[0] + c.col0, o);
        }
      }
    }
    // (no barrier here: the next tile's P0 writes xin / sinv / idx, last read before the P3 / P1 barriers; xmid's stash is
    // rewritten as fragments only after the next tile's first barrier)
  }
  if (c.mabs > CC_SH_LIMIT) atomicOr(status, 2);
}

}  // namespace
