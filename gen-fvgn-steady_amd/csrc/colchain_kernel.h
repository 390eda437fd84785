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
#ifndef GFV_CC_SPLIT
#define GFV_CC_SPLIT 0
#endif
constexpr bool SPLIT = GFV_CC_SPLIT != 0;   // fragment reads | epilogue of the previous pair | MFMAs (else reads + MFMAs | epilogue)

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

// -DGFV_CC_TIMING: per wave, cycles spent in each phase and at each barrier, written through fin_presave's ... no:
// through `status` + 64 (a debug build takes a bigger status buffer; scratch experiments only)
#ifdef GFV_CC_TIMING
#define CT_DECL long long ct_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long ct_prev_ = clock64();
#define CT(k) do { __builtin_amdgcn_s_waitcnt(0xc07f); const long long now_ = clock64(); ct_[k] += now_ - ct_prev_; ct_prev_ = now_; } while (0)
#else
#define CT_DECL
#define CT(k)
#endif
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
template <int KT, bool LOWP, bool FENCE = false>
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
    // (FENCE: the scheduler may not hoist the next k-group's four fragment reads above this one's MFMAs - at a budget of 128
    // registers sixteen fragments in flight are 64 of them)
    if (FENCE) __builtin_amdgcn_sched_barrier(0);
  }
}

// the same in two steps, so that the epilogue arithmetic of the previous pair sits between the fragment reads and the MFMAs
// that consume them (the LDS latency of 16 reads is ~ 1 k cycles for a wave that has nothing else to issue)
template <int KT, bool LOWP>
struct CcFrags {
  gfv_f16x8 h0[KT], h1[KT], l0[LOWP ? 1 : KT], l1[LOWP ? 1 : KT];
};
template <int KT, bool LOWP>
__device__ __forceinline__ void cc_frag_load(const char* xbuf, int pair, int lane, CcFrags<KT, LOWP>& f) {
  const gfv_f16x8* f0 = reinterpret_cast<const gfv_f16x8*>(xbuf + (size_t)(2 * pair) * KT * 2048) + lane;
  const gfv_f16x8* f1 = f0 + KT * 128;
#pragma unroll
  for (int T = 0; T < KT; ++T) {
    f.h0[T] = f0[(2 * T) * 64];
    f.h1[T] = f1[(2 * T) * 64];
    if (!LOWP) {
      f.l0[T] = f0[(2 * T + 1) * 64];
      f.l1[T] = f1[(2 * T + 1) * 64];
    }
  }
}
template <int KT, bool LOWP>
__device__ __forceinline__ void cc_mma_frags(const CcFrags<KT, LOWP>& f, const gfv_f16x8 (&wh)[KT], const gfv_f16x8 (&wl)[KT],
                                             floatx4& a0, floatx4& a1) {
  a0 = floatx4{0.f, 0.f, 0.f, 0.f};
  a1 = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int T = 0; T < KT; ++T) {
    if (!LOWP) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[T], f.h0[T], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[T], f.h1[T], a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[T], f.l0[T], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[T], f.l1[T], a1, 0, 0, 0);
    }
    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[T], f.h0[T], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[T], f.h1[T], a1, 0, 0, 0);
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
// LITE: the high-occupancy form.  The kernels of this family are bound by instruction issue per wave, not by memory or the
// matrix pipe (profiles/tools/colchain_ablate.py: with every output stream but one removed a launch still takes 58 % of its
// time), so waves per SIMD are what buys throughput: a wave keeps only the CURRENT layer's weight slice in registers (reloaded
// from L2 per phase, 4 KB per wave, issued ahead of the barrier that starts the phase), rows are loaded where they are consumed
// (the second workgroup of the CU covers the latency), the register budget is 128 and two workgroups of 8 waves share a CU.
template <int KT0, int N0, int TG, bool PADD, bool LOWP, bool LITE>
__global__ __launch_bounds__(64 * CC_W, LITE ? 4 : 2) void colchain_fwd_kernel(const gfv_rowtile_args_t A, int* status) {
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
  if (!LITE) cc_prefetch<KT0, N0>(A, c, 16 * g_beg, pre);

  // this wave's n-tile of every layer's image ([pass][T][nt][part][lane] x 16 B, include/gfv.h): resident for the whole launch,
  // or (LITE) one layer at a time in (wh0, wl0) / (wh1, wl1)
  gfv_f16x8 wh0[KT0], wl0[KT0], wh1[4], wl1[4], wh2[4], wl2[4];
  const gfv_f16x8* i0 = reinterpret_cast<const gfv_f16x8*>(A.layer[0].Wh) + (size_t)c.w * 128 + c.lane;
  const gfv_f16x8* i1 = reinterpret_cast<const gfv_f16x8*>(A.layer[1].Wh) + (size_t)c.w * 128 + c.lane;
  const gfv_f16x8* i2 = reinterpret_cast<const gfv_f16x8*>(A.layer[2].Wh) + (size_t)c.w * 128 + c.lane;
  if (!LITE) {
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

  CT_DECL
  for (int t0 = g_beg; t0 < g_end; t0 += TG) {
    c.row0 = 16 * t0;
    c.ngt = min(TG, g_end - t0);
    const int np = (c.ngt + 1) >> 1;
    // ---- P0: the prefetched rows become the tile's input fragments ----
    if (LITE) {
#pragma unroll
      for (int T = 0; T < KT0; ++T) { wh0[T] = i0[T * 1024]; wl0[T] = i0[T * 1024 + 64]; }   // layer 0's slice: lands behind the barrier
      if (c.w < TG) {
        cc_prefetch<KT0, N0>(A, c, c.row0, pre);
        cc_park_input<KT0>(c, pre, xin, sinv, idxs, idxr);
      }
    } else if (c.w < TG) {
      cc_park_input<KT0>(c, pre, xin, sinv, idxs, idxr);
    }
    CT(0);
    cc_barrier();
    CT(1);
    // next tile's rows: in flight through the whole tile (after the last tile: every lane re-reads row M - 1, a few cached
    // lines - the loads stay unconditional, a register array filled under a branch is parked in scratch by the compiler)
    if (!LITE) cc_prefetch<KT0, N0>(A, c, t0 + TG < g_end ? 16 * (t0 + TG) : A.M, pre);

    // ---- P1: layer 0, xin -> xmid ----
    {
      floatx4 a0, a1;
      cc_mma_pair<KT0, LOWP, LITE>(xin, 0, wh0, wl0, c.lane, a0, a1);
      CcAdd pn;
      pn.s0 = pn.r0 = pn.s1 = pn.r1 = zero4;
      if (PADD && !LITE) pn = cc_padd_load(A, c, idxs, idxr, 0);
      for (int p = 0; p < np; ++p) {
        const int pnext = min(p + 1, TG / 2 - 1);
        // (LITE: the addend rows of THIS pair, loaded ahead of the next pair's MFMAs; else one pair ahead)
        const CcAdd pc = (LITE && PADD) ? cc_padd_load(A, c, idxs, idxr, p) : pn;
        if (PADD && !LITE) pn = cc_padd_load(A, c, idxs, idxr, pnext);
        floatx4 n0, n1;
        cc_mma_pair<KT0, LOWP, LITE>(xin, pnext, wh0, wl0, c.lane, n0, n1);

        float v0[4], v1[4];
        cc_hidden_fwd<0, PADD>(c, 2 * p, a0, b0, sinv, pc.s0, pc.r0, xmid, v0);
        cc_hidden_fwd<0, PADD>(c, 2 * p + 1, a1, b0, sinv, pc.s1, pc.r1, xmid, v1);
        cc_save_pair(A.layer[0].save, c, p, v0, v1);
        a0 = n0; a1 = n1;
      }
    }
    if (LITE) {   // layer 1's slice goes out ahead of the barrier
#pragma unroll
      for (int T = 0; T < 4; ++T) { wh1[T] = i1[T * 1024]; wl1[T] = i1[T * 1024 + 64]; }
    }
    CT(2);
    cc_barrier();
    CT(3);
    // ---- P2: layer 1, xmid -> xin ----
    {
      floatx4 a0, a1;
      cc_mma_pair<4, LOWP, LITE>(xmid, 0, wh1, wl1, c.lane, a0, a1);
      for (int p = 0; p < np; ++p) {
        const int pnext = min(p + 1, TG / 2 - 1);
        floatx4 n0, n1;
        float v0[4], v1[4];
        if (SPLIT && !LITE) {
          CcFrags<4, LOWP> fr;
          cc_frag_load<4, LOWP>(xmid, pnext, c.lane, fr);
          __builtin_amdgcn_sched_barrier(0);
          cc_hidden_fwd<1, false>(c, 2 * p, a0, b1, sinv, zero4, zero4, xin, v0);
          cc_hidden_fwd<1, false>(c, 2 * p + 1, a1, b1, sinv, zero4, zero4, xin, v1);
          __builtin_amdgcn_sched_barrier(0);
          cc_mma_frags<4, LOWP>(fr, wh1, wl1, n0, n1);
        } else {
          cc_mma_pair<4, LOWP, LITE>(xmid, pnext, wh1, wl1, c.lane, n0, n1);
          cc_hidden_fwd<1, false>(c, 2 * p, a0, b1, sinv, zero4, zero4, xin, v0);
          cc_hidden_fwd<1, false>(c, 2 * p + 1, a1, b1, sinv, zero4, zero4, xin, v1);
        }
        cc_save_pair(A.layer[1].save, c, p, v0, v1);
        a0 = n0; a1 = n1;
      }
    }
    if (LITE) {
#pragma unroll
      for (int T = 0; T < 4; ++T) { wh2[T] = i2[T * 1024]; wl2[T] = i2[T * 1024 + 64]; }
    }
    CT(4);
    cc_barrier();
    CT(5);
    // ---- P3: layer 2, xin -> values; LayerNorm partials; the values wait in this wave's stash (xmid is free) ----
    float4* stash = reinterpret_cast<float4*>(xmid + (size_t)c.w * TG * 1024) + c.lane;
    {
      floatx4 a0, a1;
      cc_mma_pair<4, LOWP, LITE>(xin, 0, wh2, wl2, c.lane, a0, a1);
      for (int p = 0; p < np; ++p) {
        const int pnext = min(p + 1, TG / 2 - 1);
        floatx4 n0, n1;
        cc_mma_pair<4, LOWP, LITE>(xin, pnext, wh2, wl2, c.lane, n0, n1);
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
    // the residual rows of the whole tile go out ahead of the barrier and ahead of P4's stores (the memory counter is in order:
    // a load waited for behind stores waits for their acknowledgement too)
    float4 rres[LITE ? 1 : TG];
    if (!LITE && A.res[0]) {
#pragma unroll
      for (int q = 0; q < TG; ++q)
        rres[q] = ld4(A.res[0] + (size_t)min(c.row0 + 16 * q + c.j, c.M - 1) * A.res_ld[0] + c.col0);
    }
    CT(6);
    cc_barrier();   // (also the write-after-read guard of xin for the next tile's P0)
    CT(7);
    {
      // ---- P4: LayerNorm over the eight waves' partials, affine, residual, stores ----
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        if (q >= c.ngt) break;
        const int row = c.row0 + 16 * q + c.j;
        const bool live = row < c.M;
        const int rc = live ? row : c.M - 1;
        float4 rv = zero4;
        if (A.res[0]) rv = LITE ? ld4(A.res[0] + (size_t)rc * A.res_ld[0] + c.col0) : rres[LITE ? 0 : q];
        const float4* pp = reinterpret_cast<const float4*>(lnp + (q * 16 + c.j) * 16);
        const float4 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];   // (mean, M2) x 8 waves
        const float mean = (((p0.x + p0.z) + (p1.x + p1.z)) + ((p2.x + p2.z) + (p3.x + p3.z))) * 0.125f;
        const float e0 = p0.x - mean, e1 = p0.z - mean, e2 = p1.x - mean, e3 = p1.z - mean, e4 = p2.x - mean,
                    e5 = p2.z - mean, e6 = p3.x - mean, e7 = p3.z - mean;
        const float m2 = (((p0.y + p0.w) + (p1.y + p1.w)) + ((p2.y + p2.w) + (p3.y + p3.w))) +
                         16.0f * (((e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3)) + ((e4 * e4 + e5 * e5) + (e6 * e6 + e7 * e7)));
        const float rstd = rsqrtf(m2 * 0.0078125f + 1e-5f);   // nn.LayerNorm eps (EPD.py:32)
        if (A.fin_stats && live && c.w == 0 && c.g == 0) *reinterpret_cast<float2*>(A.fin_stats + 2 * (size_t)row) = make_float2(mean, rstd);
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
#ifdef GFV_CC_TIMING
  CT(8);
  if (c.lane == 0 && A.in_aux) {
    long long* dbg = reinterpret_cast<long long*>(const_cast<float*>(A.in_aux)) + ((size_t)blockIdx.x * CC_W + c.w) * 12;
    for (int k = 0; k < 12; ++k) dbg[k] = ct_[k];
  }
#endif
  if (c.mabs > CC_SH_LIMIT) atomicOr(status, 2);
}


// =====================================================================================================================
// Backward form: LayerNorm backward -> [W3^T, x gelu'(z2)] -> [W2^T, x gelu'(z1)] -> [W1^T] (+ residual), with the weight
// gradients of the forward's third and second Linear, their bias gradients and the LayerNorm's (dgamma, dbeta) accumulated on
// the way (include/gfv.h, gfv_rowtile_args_t.dw_partial).  Replaces, for the big MLP launches, the dX chain of
// tchain_kernel.h AND two of the three tiles of dw_multi_h_kernel (dw.hip) AND their reads of what the chain wrote:
// per row the pair moved 4 x 512 B in + 4 x 512 B out (chain) and 4 x 512 B in (the two tiles); this kernel reads
// dy, y3, z2, z1 and writes gz1 and the input gradient - g3 and gz2 never leave the CU.
//
// Everything is column-owner (a wave owns columns 16 w .. 16 w + 15 of every row of the tile):
//   P0   dy (+ gathered / plain addends), y3 and the row statistics of the forward (in_stats) -> xhat, gg = dy gamma,
//        (dgamma, dbeta) into lane-private sums, the row sums of the LayerNorm backward as per-wave partials in LDS
//   P0b  (barrier) g3 = rstd (gg - m1 - xhat m2) -> fragments with ONE power-of-two scale per tile (the weight gradient
//        contracts over rows: a per-row scale would not factor out of its sums)
//   P3   (barrier) chain layer 0; epilogue: z2 -> gelu'(z2) for gz2, gelu(z2) = a2 for the weight gradient; both -> fragments
//   dW3  (barrier) D[n][k] += sum_rows g3[row][n] a2[row][k]: both operands are the transposes of what the fragments hold
//        (lane = row there, lane = column here) - ds_read_b64_tr_b16 reads them out of the SAME fragment buffers: each 16-lane
//        group fetches the 8-byte pieces of 4 rows x 16 columns and the hardware transposes them; a wave owns n-tile w
//        and walks the 8 k-tiles; contraction over 32 rows (two groups) per MFMA, hi / lo split as in the chain
//   P2   (barrier) chain layer 1; epilogue: z1 -> gz1 (stored: the node-level scatter reads it), a1 -> fragments
//   P1   (barrier) chain layer 2 -> input gradient (+ residual) stored;  dW2 from (gz2, a1)
// Three fragment buffers (g3 | gz1, gz2, a2 | a1); tiles of TG = 4 groups (64 rows): 96 KB + 4 KB of partials.
// Scales: g3 fragments carry s3 = 2^k with s3 max|g3| <= 2^15 over the tile (from a bound taken before the barrier; an
// order of magnitude of slack costs nothing: the split keeps 2^13 of headroom below the maximum); gz2 / gz1 step down from it by
// the layers' guaranteed growth bounds (row 1-norms of the weight images, taken at kernel start); a2 / a1 carry the fixed CC_SH.  The weight-gradient
// accumulators live in units of the current tile's scale and are rescaled (exactly: powers of two) when it changes; the
// scale may not rise more than 2^20 above the smallest one seen (rows that small add nothing to sums dominated by rows a
// million times larger).
constexpr int CB_TG = 4;
struct CbLds {
  static constexpr int BUF = CB_TG * 8192;
  static constexpr int B0 = 0, B1 = BUF, B2 = 2 * BUF;
  static constexpr int PART = 3 * BUF;                        // float2 [TG][16][8]: per-wave partial (s1, s2) of a row
  static constexpr int SMAX = PART + CB_TG * 16 * 8 * 8;      // float [8]: per-wave bound of max |g3| over the tile
  static constexpr int TOTAL = SMAX + 64;
};

// transposed operand of the weight gradient: columns 16 ct .. 16 ct + 15 of the two groups (q0, q0 + 1) of a fragment buffer,
// lane (i = column, g') slots e = 4 h + r <- row 4 g' + r of group q0 + h.  One ds_read_b64_tr_b16 per (group, part).
__device__ __forceinline__ void cb_tr_operand(const char* xbuf, int q0, int ct, int lane, gfv_f16x8& hi, gfv_f16x8& lo) {
  typedef short s4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s4* lds_s4;
  const int s = lane & 15, gp = lane >> 4;
  // source piece of this lane: row 4 g' + (s >> 2), column quad s & 3 of the tile = producer lane group g = s & 3
  const int off = ((q0 * 4 + (ct >> 1)) * 2) * 1024 + (16 * (s & 3) + 4 * gp + (s >> 2)) * 16 + (ct & 1) * 8;
  const char* p = xbuf + off;
  const s4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p));
  const s4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p + 1024));
  const s4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p + 8192));
  const s4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p + 8192 + 1024));
  typedef short s8 __attribute__((ext_vector_type(8)));
  const s8 h = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
  const s8 l = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
  hi = __builtin_bit_cast(gfv_f16x8, h);
  lo = __builtin_bit_cast(gfv_f16x8, l);
}

// one tile's contribution to a fused weight gradient: acc[kt] += G^T (n-tile w) x A (k-tile kt) over the tile's row pairs;
// accb += G^T x ones (the bias gradient: every column of the result is the column sum of G)
template <bool LOWP>
__device__ __forceinline__ void cb_dw_tile(const char* gbuf, const char* abuf, int npairs, int w, int lane, floatx4 (&acc)[8],
                                           floatx4& accb) {
  const _Float16 one = (_Float16)1.0f;
  const gfv_f16x8 ones = {one, one, one, one, one, one, one, one};
  for (int pr = 0; pr < npairs; ++pr) {
    gfv_f16x8 gh, gl;
    cb_tr_operand(gbuf, 2 * pr, w, lane, gh, gl);
    if (!LOWP) accb = __builtin_amdgcn_mfma_f32_16x16x32_f16(gl, ones, accb, 0, 0, 0);
    accb = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh, ones, accb, 0, 0, 0);
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      gfv_f16x8 ah, al;
      cb_tr_operand(abuf, 2 * pr, kt, lane, ah, al);
      if (!LOWP) {
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gl, ah, acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh, al, acc[kt], 0, 0, 0);
      }
      acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh, ah, acc[kt], 0, 0, 0);
      if ((kt & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (four k-tiles' operands in flight, not eight: the register budget)
    }
  }
}

// chain-layer epilogue of one group in the backward form (GFV_OP_MUL_DGELU): v = acc / scales x gelu'(z) -> fragments with
// the scale `sg`; a = gelu(z) -> fragments with CC_SH (the weight gradient's other operand); v is handed back for the save
__device__ __forceinline__ void cb_hidden_bwd(CcCtx& c, int q, const floatx4& acc, float inv_in, const float4& z, float sg,
                                              char* gout, char* aout, float (&v)[4]) {
  const float zz[4] = {z.x, z.y, z.z, z.w};
  float a[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    // gelu and gelu' share the erfc evaluation (gfv_common.h)
    const gfv_erfc_t e = gfv_erfc_half(zz[r]);
    const float cdf = zz[r] >= 0.0f ? 1.0f - e.y : e.y;
    const float dg = fmaf(zz[r] * 0.39894228040143267794f, e.e, cdf);
    a[r] = fmaf(-fabsf(zz[r]), e.y, fmaxf(zz[r], 0.0f));
    v[r] = ((acc[r] * inv_in) * c.invw) * dg;
  }
  const float mq = max3_abs(max3_abs(0.f, v[0], v[1]), v[2], v[3]) * sg;
  const float ma = max3_abs(max3_abs(0.f, a[0], a[1]), a[2], a[3]) * (CC_SH * (1.0f / 32.0f));   // (a against CC_SH_LIMIT x 32 = 65536)
  c.mabs = fmaxf(c.mabs, q < c.ngt ? fmaxf(mq, ma) : 0.f);
  cc_put_frag(gout, q, c, v, sg);
  cc_put_frag(aout, q, c, a, CC_SH);
}

// ---- buffer addressing ---------------------------------------------------------------------------------------------
// Every [M, 128] array of a launch is addressed through a buffer descriptor (4 SGPRs, built from the kernel arguments) and
// ONE 32-bit byte offset per row group that all of them share (row x 512 + first column x 4): no 64-bit address per access
// (flat addressing cost the first version of this kernel ~140 address computations and two registers per live pointer,
// which is what made it spill).  The bounds check of the descriptor does the predication: a store whose offset lies
// beyond the array - rows past M, the dead groups of a partial tile get such an offset - is dropped, an absent array
// (NULL: zero records) reads as zeros and swallows its stores.
typedef int cb_i32x4 __attribute__((ext_vector_type(4)));
typedef int cb_i32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t cb_rsrc;
constexpr int CB_OFF_DEAD = 0x7ffffff0;
__device__ __forceinline__ cb_rsrc cb_buf(const void* p, size_t bytes) {
  const size_t n = p ? (bytes < 0x7fffffe0ull ? bytes : 0x7fffffe0ull) : 0;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)n, 0x00020000);
}
// (the loaded vector is cast to floats as a WHOLE: hipcc 7.2 narrows the load to one dword when the four lanes of the integer
// vector are bit-cast one by one - found by the parity test, profiles/tools/tr)
typedef float cb_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 cb_ld4(cb_rsrc r, int off) {
  const floatx4 f = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  return make_float4(f[0], f[1], f[2], f[3]);
}
__device__ __forceinline__ float2 cb_ld2(cb_rsrc r, int off) {
  const cb_f32x2 f = __builtin_bit_cast(cb_f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
  return make_float2(f[0], f[1]);
}
__device__ __forceinline__ void cb_st4(cb_rsrc r, int off, const float (&v)[4]) {
  const floatx4 f = {v[0], v[1], v[2], v[3]};
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cb_i32x4, f), r, off, 0, 0);
}

// the rows a tile's LayerNorm backward starts from, loaded one tile ahead (issued behind the previous tile's second chain
// layer, consumed in P0): this wave's 16 columns of dy, y3, the gathered addend, and the rows' forward statistics
struct CbIn {
  float4 dy[CB_TG], yv[CB_TG], ga[CB_TG];
  float2 st[CB_TG];
};
struct CbBufs {
  cb_rsrc dy, y3, stats, z2, z1, gadd, gidx, save1, res, out;   // (the rarely used ones are built where they are used: SGPR budget)
};
// byte offset of this lane's 16 bytes of row (row0 + 16 q + j) in an [M, 128] array, clamped to the last row
__device__ __forceinline__ int cb_off(const CcCtx& c, int row0, int q) { return min(row0 + 16 * q + c.j, c.M - 1) * 512 + c.col0 * 4; }

template <bool GADD>
__device__ __forceinline__ void cb_load_gidx(const CbBufs& B, const CcCtx& c, int row0, int (&gidx)[CB_TG]) {
  if (GADD) {
#pragma unroll
    for (int q = 0; q < CB_TG; ++q) gidx[q] = __builtin_amdgcn_raw_buffer_load_b32(B.gidx, min(row0 + 16 * q + c.j, c.M - 1) * 4, 0, 0);
  }
}
__device__ __forceinline__ void cb_load_inputs(const CbBufs& B, const CcCtx& c, int row0, CbIn& in) {
#pragma unroll
  for (int q = 0; q < CB_TG; ++q) {
    const int off = cb_off(c, row0, q);
    in.dy[q] = cb_ld4(B.dy, off);
    in.yv[q] = cb_ld4(B.y3, off);
    in.st[q] = cb_ld2(B.stats, min(row0 + 16 * q + c.j, c.M - 1) * 8);
  }
}
template <bool GADD>
__device__ __forceinline__ void cb_load_gathers(const CbBufs& B, const CcCtx& c, const int (&gidx)[CB_TG], CbIn& in) {
  if (GADD) {   // [gadd[s] (64) | gadd[r] (64)]: this wave's 16 columns lie in one half
#pragma unroll
    for (int q = 0; q < CB_TG; ++q) in.ga[q] = cb_ld4(B.gadd, gidx[q] * 256 + (c.col0 & 63) * 4);
  }
}

// GADD: the gathered addend [gadd[s] | gadd[r]] of the incoming gradient exists; DW1: the forward's first Linear is 128 deep
// and its input rows are given (dw_in): its weight gradient is fused as well (a fourth fragment buffer, 36 more registers)
// OUT2: the last chain layer is 192 wide (NodeBlock: W1^T with the rows for x first, then the 64 for the neighbour mean,
// blocks.py:54): waves 0..3 own a second n-tile and write out[1] ([M, 64], no residual)
// NOOUT: the MLP's input needs no gradient (the encoders, EPD.py:92-119): a two-layer launch whose out[0] receives gz1 (what the
// first Linear's weight-gradient launch reads); the third chain phase is only the weight gradient of the second Linear
template <bool LOWP, bool GADD, bool DW1, bool OUT2 = false, bool NOOUT = false>
__global__ __launch_bounds__(64 * CC_W, 2) void colchain_bwd_kernel(const gfv_rowtile_args_t A, int* status) {
  constexpr int TG = CB_TG;
  __shared__ __attribute__((aligned(16))) char lds[CbLds::TOTAL + (DW1 ? CbLds::BUF : 0)];
  char* b0 = lds + CbLds::B0;
  char* b1 = lds + CbLds::B1;
  char* b2 = lds + CbLds::B2;
  char* b3 = lds + CbLds::TOTAL;   // (DW1) fragments of the first Linear's input rows
  float* part = reinterpret_cast<float*>(lds + CbLds::PART);
  float* smax = reinterpret_cast<float*>(lds + CbLds::SMAX);

  CcCtx c;
  c.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  c.lane = threadIdx.x & 63;
  c.j = c.lane & 15;
  c.g = c.lane >> 4;
  c.col0 = 16 * c.w + 4 * c.g;
  c.M = A.M;
  c.mabs = 0.f;
  c.invw = 1.0f / gfv_pow2_scale(*A.wmax);

  const int nwg = gridDim.x;
  const int wg = (nwg & 7) == 0 ? (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int NG = (A.M + 15) >> 4;
  const int g_beg = (int)((long)NG * wg / nwg), g_end = (int)((long)NG * (wg + 1) / nwg);

  const size_t rows128 = (size_t)A.M * 512;
  CbBufs B;
  B.dy = cb_buf(A.seg[0].ptr, rows128);
  B.y3 = cb_buf(A.in_aux, rows128);
  B.stats = cb_buf(A.in_stats, (size_t)A.M * 8);
  B.z2 = cb_buf(A.layer[0].aux, rows128);
  B.z1 = cb_buf(A.layer[1].aux, rows128);
  B.gadd = cb_buf(A.gadd, 0x7fffffe0ull);   // (its row count is not an argument; the gather rows come from the index arrays)
  B.gidx = cb_buf(c.w < 4 ? A.gadd_s : A.gadd_r, (size_t)A.M * 4);
  B.save1 = cb_buf(NOOUT ? A.out[0] : A.layer[1].save, rows128);
  B.res = cb_buf(NOOUT ? nullptr : A.res[0], rows128);
  B.out = cb_buf(NOOUT ? nullptr : A.out[0], rows128);
  const cb_rsrc out2 = cb_buf(OUT2 ? A.out[1] : nullptr, (size_t)A.M * 256);

  // this wave's n-tile of the three transposed layers' images.  The weight-gradient accumulators take 72 (DW1: 108) registers
  // for the whole launch, so the chain's weights are NOT resident here: every tile fetches each layer's slice (4 KB per wave,
  // L2 hits) one phase ahead of its use - 12 KB per wave and 64-row tile next to 230 KB of activation traffic - and the
  // LayerNorm-backward phase, where the register pressure peaks, holds none of them.
  const cb_rsrc w0 = cb_buf(A.layer[0].Wh, 65536), w1 = cb_buf(A.layer[1].Wh, 65536),
                w2 = cb_buf(NOOUT ? nullptr : A.layer[2].Wh, OUT2 ? 131072 : 65536);
  const int woff = (c.w * 128 + c.lane) * 16;   // + T * 16384 (+ 1024: the lo part)
  // fused weight gradients: D[n = 16 w + 4 g + r][k = 16 kt + j] in lane (j, g) of acc[kt][r]
  floatx4 dw3[8], dw2[8], dw1[DW1 ? 8 : 1], db3 = floatx4{0.f, 0.f, 0.f, 0.f}, db2 = db3, db1 = db3;
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) dw3[kt] = dw2[kt] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < (DW1 ? 8 : 1); ++kt) dw1[kt] = floatx4{0.f, 0.f, 0.f, 0.f};
  float dgam[4] = {0.f, 0.f, 0.f, 0.f}, dbet[4] = {0.f, 0.f, 0.f, 0.f};
  float sacc = 0.f;          // scale the g3-side accumulators are in (0: nothing accumulated yet); gz2 side: / 32, gz1 side: / 1024
  float scap = 3.0e38f;      // 2^20 x the smallest tile scale so far

  // How much a chain layer can enlarge its input rows: |(g W^T)[n]| <= max|g| sum_k |W^T[n][k]|, x gelu' <= 1.13.  The largest
  // row 1-norm of each of the first two layers' images (this wave's 16 rows from its fragments, the workgroup's maximum through
  // LDS; in units of the weight scale) gives the power of two by which the next layer's fragment scale steps down from this
  // one's: a guaranteed bound, so the gradient fragments cannot overflow - and for weights as the reference initialises them
  // (trunc_normal 0.02: row norms ~ 2) it costs two bits where a fixed allowance for the worst case would cost seven.
  float step1, step2;   // s2 = s3 * step1, s1 = s2 * step2
  {
    float l1[2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const cb_rsrc wb = l == 0 ? w0 : w1;
      float acc = 0.f;
#pragma unroll
      for (int T = 0; T < 4; ++T) {
        const gfv_f16x8 h = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(wb, woff + T * 16384, 0, 0));
        const gfv_f16x8 lo = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(wb, woff + T * 16384 + 1024, 0, 0));
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += fabsf((float)h[e] + (float)lo[e]);
      }
      l1[l] = gfv_wave_max(row_sum(acc));   // row_sum: the 4 lane groups that share a weight row; then the 16 rows
    }
    if (c.lane == 0) { smax[c.w] = l1[0]; part[c.w] = l1[1]; }
    cc_barrier();
    float m0 = 0.f, m1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { m0 = fmaxf(m0, smax[k]); m1 = fmaxf(m1, part[k]); }
    cc_barrier();
    // step = 2^-ceil(log2(1.13 * L1 / ws)): an exact power of two at most 1 / growth
    const float g0 = 1.13f * m0 * c.invw, g1 = 1.13f * m1 * c.invw;
    step1 = 1.0f / gfv_pow2_ceil(g0);
    step2 = 1.0f / gfv_pow2_ceil(g1);
  }

  // (DW1: the 36 registers of the third accumulator set leave no room for rows in flight a tile ahead - they are loaded in P0)
  constexpr bool PRE = !DW1;
  int gidx[TG] = {0, 0, 0, 0};
  CbIn in;
  if (g_beg < g_end) {
    cb_load_gidx<GADD>(B, c, 16 * g_beg, gidx);
    if (PRE) {
      cb_load_inputs(B, c, 16 * g_beg, in);
      cb_load_gathers<GADD>(B, c, gidx, in);
    }
  }
  CT_DECL
  for (int t0 = g_beg; t0 < g_end; t0 += TG) {
    c.row0 = 16 * t0;
    c.ngt = min(TG, g_end - t0);
    const int np = (c.ngt + 1) >> 1;
    const int next_row0 = t0 + TG < g_end ? 16 * (t0 + TG) : c.M;   // (behind the last tile: every lane re-reads row M - 1)
    int offL[TG], offS[TG];   // this tile's byte offsets: loads (clamped rows), stores (dead rows / groups: out of bounds)
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const int row = c.row0 + 16 * q + c.j;
      offL[q] = cb_off(c, c.row0, q);
      offS[q] = (q < c.ngt && row < c.M) ? row * 512 + c.col0 * 4 : CB_OFF_DEAD;
    }
    // ---- P0: LayerNorm backward, first half (the rows were loaded one tile ahead) ----
    float gg[TG][4], xh[TG][4], rs[TG];
    {
      float bmax = 0.f;
      if (!PRE) {
        cb_load_inputs(B, c, c.row0, in);
        cb_load_gathers<GADD>(B, c, gidx, in);
      }
      if (A.in_add) {
        const cb_rsrc ia = cb_buf(A.in_add, rows128);
#pragma unroll
        for (int q = 0; q < TG; ++q) {
          const float4 t = cb_ld4(ia, offL[q]);
          in.dy[q].x += t.x; in.dy[q].y += t.y; in.dy[q].z += t.z; in.dy[q].w += t.w;
        }
      }
      const float4 gam = ld4(A.in_gamma + c.col0);
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        const float lf = offS[q] != CB_OFF_DEAD ? 1.0f : 0.0f;   // rows past M / dead groups must not reach any sum over rows
        float d[4] = {in.dy[q].x, in.dy[q].y, in.dy[q].z, in.dy[q].w};
        if (GADD) { d[0] += in.ga[q].x; d[1] += in.ga[q].y; d[2] += in.ga[q].z; d[3] += in.ga[q].w; }
        const float y[4] = {in.yv[q].x, in.yv[q].y, in.yv[q].z, in.yv[q].w};
        const float gm[4] = {gam.x, gam.y, gam.z, gam.w};
        const float mean = in.st[q].x, rstd = in.st[q].y;
        rs[q] = rstd;
        float s1 = 0.f, s2 = 0.f, am = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          d[r] *= lf;
          xh[q][r] = (y[r] - mean) * rstd;
          dgam[r] += d[r] * xh[q][r];
          dbet[r] += d[r];
          gg[q][r] = d[r] * gm[r];
          s1 += gg[q][r];
          s2 += gg[q][r] * xh[q][r];
          am = fmaxf(am, fabsf(gg[q][r]));
        }
        s1 = row_sum(s1);
        s2 = row_sum(s2);
        if (c.g == 0) *reinterpret_cast<float2*>(part + ((q * 16 + c.j) * 8 + c.w) * 2) = make_float2(s1, s2);
        bmax = fmaxf(bmax, am * fabsf(rstd) * lf);
      }
      bmax = gfv_wave_max(bmax);
      if (c.lane == 0) smax[c.w] = bmax;
    }
    gfv_f16x8 wh[4], wl[4];   // the current chain layer's slice
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      wh[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w0, woff + T * 16384, 0, 0));
      wl[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w0, woff + T * 16384 + 1024, 0, 0));
    }
    // the saved pre-activations the first chain layer's epilogue needs: in flight through P0b
    float4 zq[TG];
#pragma unroll
    for (int q = 0; q < TG; ++q) zq[q] = cb_ld4(B.z2, offL[q]);
    CT(0);
    cc_barrier();
    CT(1);
    // ---- P0b: g3 and its fragments ----
    float s3;
    {
      const float4 ma = *reinterpret_cast<const float4*>(smax), mb = *reinterpret_cast<const float4*>(smax + 4);
      const float mx = fmaxf(fmaxf(fmaxf(ma.x, ma.y), fmaxf(ma.z, ma.w)), fmaxf(fmaxf(mb.x, mb.y), fmaxf(mb.z, mb.w)));
      // |g3| <= rstd (|gg| + |m1| + |xhat| |m2|) with |m1| <= max|gg|, |m2| <= max|gg| mean|xhat| <= max|gg| (mean xhat^2 = 1)
      // and |xhat| <= sqrt(127): rstd max|gg| (2 + 11.3) - the bound, not the maximum, sets the scale
      s3 = fminf(gfv_pow2_scale(mx * 13.5f) * 2.0f, scap);
      scap = fminf(scap, s3 * 1048576.0f);
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        const float4* pp = reinterpret_cast<const float4*>(part + (q * 16 + c.j) * 16);
        const float4 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];   // (s1, s2) x 8 waves
        const float m1 = (((p0.x + p0.z) + (p1.x + p1.z)) + ((p2.x + p2.z) + (p3.x + p3.z))) * 0.0078125f;
        const float m2 = (((p0.y + p0.w) + (p1.y + p1.w)) + ((p2.y + p2.w) + (p3.y + p3.w))) * 0.0078125f;
        float g3[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) g3[r] = rs[q] * (gg[q][r] - m1 - xh[q][r] * m2);
        if (A.in_save) cb_st4(cb_buf(A.in_save, rows128), offS[q], g3);
        cc_put_frag(b0, q, c, g3, s3);
      }
    }
    // the accumulators move to this tile's units
    if (sacc != 0.f && sacc != s3) {
      const float ratio = s3 / sacc;
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) { dw3[kt] *= ratio; dw2[kt] *= ratio; }
#pragma unroll
      for (int kt = 0; kt < (DW1 ? 8 : 1); ++kt) dw1[kt] *= ratio;
      db3 *= ratio;
      db2 *= ratio;
      db1 *= ratio;
    }
    sacc = s3;
    const float s2s = s3 * step1, s1s = s2s * step2;
    // per-16-row scales of the rows this launch leaves for a weight-gradient launch of its own (gfv_rowtile_args_t.gscale):
    // the tile's fragment scales, a quarter of them (a slab scale s wants s max|v| <= 2^14, the fragments allow 2^16)
    if (A.gscale && c.w == 0 && c.lane < c.ngt) {
      const size_t grp = (size_t)(c.row0 >> 4) + c.lane;
      A.gscale[grp] = s3 * 0.25f;
      A.gscale[(size_t)A.gscale_ld + grp] = s2s * 0.25f;
      A.gscale[2 * (size_t)A.gscale_ld + grp] = s1s * 0.25f;
    }
    CT(2);
    cc_barrier();
    CT(3);
    // ---- P3: chain layer 0 (b0 -> gz2 in b1, a2 in b2) ----
    cb_load_gidx<GADD>(B, c, next_row0, gidx);   // the next tile's gather rows: the gathers go out behind P1
    {
      const float inv_in = 1.0f / s3;
      floatx4 a0, a1;
      cc_mma_pair<4, LOWP, true>(b0, 0, wh, wl, c.lane, a0, a1);
#pragma unroll
      for (int p = 0; p < TG / 2; ++p) {
        if (p >= np) break;
        floatx4 n0 = a0, n1 = a1;
        if (p + 1 < TG / 2) cc_mma_pair<4, LOWP, true>(b0, p + 1, wh, wl, c.lane, n0, n1);
        float v0[4], v1[4];
        cb_hidden_bwd(c, 2 * p, a0, inv_in, zq[2 * p], s2s, b1, b2, v0);
        cb_hidden_bwd(c, 2 * p + 1, a1, inv_in, zq[2 * p + 1], s2s, b1, b2, v1);
        if (A.layer[0].save) {
          const cb_rsrc s0 = cb_buf(A.layer[0].save, rows128);
          cb_st4(s0, offS[2 * p], v0);
          cb_st4(s0, offS[2 * p + 1], v1);
        }
        a0 = n0; a1 = n1;
      }
    }
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      wh[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w1, woff + T * 16384, 0, 0));
      wl[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w1, woff + T * 16384 + 1024, 0, 0));
    }
    // the second chain layer's saved pre-activations (and the first Linear's input rows): in flight through dW3
#pragma unroll
    for (int q = 0; q < TG; ++q) zq[q] = cb_ld4(B.z1, offL[q]);
    CT(4);
    cc_barrier();
    CT(5);
    // ---- dW3 += g3^T a2 ----
    float4 ev[DW1 ? TG : 1];   // (DW1) the first Linear's input rows: in flight through this phase, fragments at its end
    if (DW1) {
#pragma unroll
      for (int q = 0; q < TG; ++q) ev[q] = cb_ld4(cb_buf(A.dw_in, rows128), offL[q]);
    }
    cb_dw_tile<LOWP>(b0, b2, np, c.w, c.lane, dw3, db3);
    if (DW1) {
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        const float e4[4] = {ev[q].x, ev[q].y, ev[q].z, ev[q].w};
        const float me = max3_abs(max3_abs(0.f, e4[0], e4[1]), e4[2], e4[3]) * (CC_SH * (1.0f / 32.0f));
        c.mabs = fmaxf(c.mabs, q < c.ngt ? me : 0.f);
        cc_put_frag(b3, q, c, e4, CC_SH);
      }
    }
    CT(6);
    cc_barrier();
    CT(7);
    // ---- P2: chain layer 1 (b1 -> gz1 in b0, a1 in b2) ----
    {
      const float inv_in = 1.0f / s2s;
      floatx4 a0, a1;
      cc_mma_pair<4, LOWP, true>(b1, 0, wh, wl, c.lane, a0, a1);
#pragma unroll
      for (int p = 0; p < TG / 2; ++p) {
        if (p >= np) break;
        floatx4 n0 = a0, n1 = a1;
        if (p + 1 < TG / 2) cc_mma_pair<4, LOWP, true>(b1, p + 1, wh, wl, c.lane, n0, n1);
        float v0[4], v1[4];
        cb_hidden_bwd(c, 2 * p, a0, inv_in, zq[2 * p], s1s, b0, b2, v0);
        cb_hidden_bwd(c, 2 * p + 1, a1, inv_in, zq[2 * p + 1], s1s, b0, b2, v1);
        cb_st4(B.save1, offS[2 * p], v0);
        cb_st4(B.save1, offS[2 * p + 1], v1);
        a0 = n0; a1 = n1;
      }
    }
    if constexpr (!NOOUT) {
#pragma unroll
      for (int T = 0; T < 4; ++T) {
        wh[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w2, woff + T * 16384, 0, 0));
        wl[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w2, woff + T * 16384 + 1024, 0, 0));
      }
    }
    // this tile's residual rows go out here: in flight through the barrier
    float4 rr[NOOUT ? 1 : TG];
    if constexpr (!NOOUT) {
#pragma unroll
      for (int q = 0; q < TG; ++q) rr[q] = cb_ld4(B.res, offL[q]);
    }
    CT(8);
    cc_barrier();
    CT(9);
    // ---- P1: chain layer 2 (b0 -> the input gradient) and dW2 += gz2^T a1 (and dW1 += gz1^T x) ----
    if constexpr (!NOOUT) {
      const float inv_in = 1.0f / s1s;
      floatx4 a0, a1;
      cc_mma_pair<4, LOWP, true>(b0, 0, wh, wl, c.lane, a0, a1);
#pragma unroll
      for (int p = 0; p < TG / 2; ++p) {
        if (p >= np) break;
        floatx4 n0 = a0, n1 = a1;
        if (p + 1 < TG / 2) cc_mma_pair<4, LOWP, true>(b0, p + 1, wh, wl, c.lane, n0, n1);
        float o0[4], o1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o0[r] = (a0[r] * inv_in) * c.invw;
          o1[r] = (a1[r] * inv_in) * c.invw;
        }
        o0[0] += rr[2 * p].x; o0[1] += rr[2 * p].y; o0[2] += rr[2 * p].z; o0[3] += rr[2 * p].w;
        o1[0] += rr[2 * p + 1].x; o1[1] += rr[2 * p + 1].y; o1[2] += rr[2 * p + 1].z; o1[3] += rr[2 * p + 1].w;
        cb_st4(B.out, offS[2 * p], o0);
        cb_st4(B.out, offS[2 * p + 1], o1);
        a0 = n0; a1 = n1;
      }
      if constexpr (OUT2) {
        if (c.w < 4) {   // (wave-uniform) output columns 128 + 16 w ..: the image's second pass, n-tile w
          gfv_f16x8 xh2[4], xl2[4];
#pragma unroll
          for (int T = 0; T < 4; ++T) {
            xh2[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w2, 65536 + woff + T * 16384, 0, 0));
            xl2[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w2, 65536 + woff + T * 16384 + 1024, 0, 0));
          }
#pragma unroll
          for (int p = 0; p < TG / 2; ++p) {
            if (p >= np) break;
            floatx4 e0, e1;
            cc_mma_pair<4, LOWP, true>(b0, p, xh2, xl2, c.lane, e0, e1);
            float o0[4], o1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              o0[r] = (e0[r] * inv_in) * c.invw;
              o1[r] = (e1[r] * inv_in) * c.invw;
            }
            const int ra = c.row0 + 32 * p + c.j, rb = ra + 16;
            cb_st4(out2, (2 * p < c.ngt && ra < c.M) ? ra * 256 + c.col0 * 4 : CB_OFF_DEAD, o0);
            cb_st4(out2, (2 * p + 1 < c.ngt && rb < c.M) ? rb * 256 + c.col0 * 4 : CB_OFF_DEAD, o1);
          }
        }
      }
    }
    CT(10);
    // the next tile's rows (and its gathered addend rows): in flight through the last weight gradients.  (Issued any earlier
    // they sit in 56 registers beside a chain phase, and the kernel spills: a scratch reload waits for every load in flight.)
    if (PRE) {
      cb_load_inputs(B, c, next_row0, in);
      cb_load_gathers<GADD>(B, c, gidx, in);
    }
    cb_dw_tile<LOWP>(b1, b2, np, c.w, c.lane, dw2, db2);
    if constexpr (DW1) cb_dw_tile<LOWP>(b0, b3, np, c.w, c.lane, dw1, db1);
    CT(11);
    // (the next tile's P0 writes only `part` / `smax`, last read in P0b; its P0b writes b0 behind the barrier that follows P0)
  }

#ifdef GFV_CC_TIMING
  if (c.lane == 0 && A.fin_aux) {
    long long* dbg = reinterpret_cast<long long*>(const_cast<float*>(A.fin_aux)) + ((size_t)blockIdx.x * CC_W + c.w) * 12;
    for (int kk = 0; kk < 12; ++kk) dbg[kk] = ct_[kk];
  }
#endif
  // ---- the workgroup's partial block: [dW3 | db3 | dW2 | db2 | dgamma | dbeta | dW1 | db1] (include/gfv.h) ----
  if (A.dw_partial) {
    float* blk = A.dw_partial + (size_t)blockIdx.x * A.dw_partial_stride;
    const float is = sacc != 0.f ? 1.0f / sacc : 0.f;
    const float r2 = 1.0f / step1, r1 = r2 / step2;   // the gz2 / gz1 sides are in units of sacc * step1 (* step2)
    const float u3 = is * CC_SH_INV, u2 = (is * r2) * CC_SH_INV, u1 = (is * r1) * CC_SH_INV;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = 16 * c.w + 4 * c.g + r, k = 16 * kt + c.j;
        blk[n * 128 + k] = dw3[kt][r] * u3;
        blk[16384 + 128 + n * 128 + k] = dw2[kt][r] * u2;
        if (DW1) blk[2 * 16384 + 512 + n * 128 + k] = dw1[kt][r] * u1;
      }
    if (c.j == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        blk[16384 + 16 * c.w + 4 * c.g + r] = db3[r] * is;
        blk[2 * 16384 + 128 + 16 * c.w + 4 * c.g + r] = db2[r] * (is * r2);
        if (DW1) blk[3 * 16384 + 512 + 16 * c.w + 4 * c.g + r] = db1[r] * (is * r1);
      }
    }
    // (dgamma, dbeta): lane-private sums over the rows j and the groups this lane saw -> sum over the 16 lanes of a DPP row
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float dg = gfv_row16_sum(dgam[r]), db = gfv_row16_sum(dbet[r]);
      if (c.j == 0) {
        blk[2 * 16384 + 256 + c.col0 + r] = dg;
        blk[2 * 16384 + 384 + c.col0 + r] = db;
      }
    }
  }
  if (c.mabs > 60000.0f) atomicOr(status, 2);
}


// =====================================================================================================================
// Forward form, second generation: the structure that made the backward kernel fast, applied to the forward chain.
// Against colchain_fwd_kernel above: no loader roles (every wave loads ITS 16 columns of the tile's input rows, the row
// maxima for the row scales are combined through LDS behind the barrier the phase needs anyway), buffer addressing (one
// 32-bit offset per row group for every [M, 128] array, bounds-checked stores instead of exec-masked ones), the layers'
// weight slices fetched one phase ahead instead of resident, the gathered first-layer addend and the next tile's rows in
// flight a phase / a tile ahead, the last layer's values kept in registers through the LayerNorm barrier.  Tiles of
// TG = 4 groups (64 rows), five barriers per tile; WPS = waves per SIMD the register allocation aims at (2: one workgroup
// per CU; 4: two, 128 registers).
//   P0   (rows loaded a tile ahead) per-wave row maxima -> LDS;  the addend gathers and layer 0's slice go out
//   P0b  row scale from the eight partial maxima; input fragments
//   P1   layer 0 -> z1 (+ bias + addend) saved, gelu -> fragments;  P2  layer 1 likewise;  P3  layer 2 -> y3 saved,
//        (mean, M2) partials of the LayerNorm -> LDS, values stay in registers
//   P4   LayerNorm from the eight partials, statistics / pre-residual / output rows stored; next tile's rows requested
struct CfLds {
  static constexpr int XIN = 0;
  static constexpr int XMID = CB_TG * 8192;
  static constexpr int RMAX = 2 * CB_TG * 8192;                 // float [TG][16][8]: per-wave max |x| of a row's 16 columns
  static constexpr int LNP = RMAX + CB_TG * 16 * 8 * 4;        // float2 [TG][16][8]
  static constexpr int TOTAL = LNP + CB_TG * 16 * 8 * 8;
};

template <bool PADD, bool LOWP, int WPS>
__global__ __launch_bounds__(64 * CC_W, WPS) void colchain_fwd2_kernel(const gfv_rowtile_args_t A, int* status) {
  constexpr int TG = CB_TG;
  __shared__ __attribute__((aligned(16))) char lds[CfLds::TOTAL];
  char* xin = lds + CfLds::XIN;
  char* xmid = lds + CfLds::XMID;
  float* rmax = reinterpret_cast<float*>(lds + CfLds::RMAX);
  float* lnp = reinterpret_cast<float*>(lds + CfLds::LNP);

  CcCtx c;
  c.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  c.lane = threadIdx.x & 63;
  c.j = c.lane & 15;
  c.g = c.lane >> 4;
  c.col0 = 16 * c.w + 4 * c.g;
  c.M = A.M;
  c.mabs = 0.f;
  c.invw = 1.0f / gfv_pow2_scale(*A.wmax);

  const int nwg = gridDim.x;
  const int wg = (nwg & 7) == 0 ? (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int NG = (A.M + 15) >> 4;
  const int g_beg = (int)((long)NG * wg / nwg), g_end = (int)((long)NG * (wg + 1) / nwg);

  const size_t rows128 = (size_t)A.M * 512;
  const cb_rsrc bx = cb_buf(A.seg[0].ptr, rows128), bz1 = cb_buf(A.layer[0].save, rows128), bz2 = cb_buf(A.layer[1].save, rows128),
                by3 = cb_buf(A.fin_presave, rows128), bnr = cb_buf(A.out_nores, rows128), bres = cb_buf(A.res[0], rows128),
                bout = cb_buf(A.out[0], rows128), bst = cb_buf(A.fin_stats, (size_t)A.M * 8),
                bpad = cb_buf(A.padd, 0x7fffffe0ull), bis = cb_buf(A.padd_s, (size_t)A.M * 4), bir = cb_buf(A.padd_r, (size_t)A.M * 4);
  const cb_rsrc w0 = cb_buf(A.layer[0].Wh, 65536), w1 = cb_buf(A.layer[1].Wh, 65536), w2 = cb_buf(A.layer[2].Wh, 65536);
  const int woff = (c.w * 128 + c.lane) * 16;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 bi0 = A.layer[0].bias ? ld4(A.layer[0].bias + c.col0) : zero4;
  const float4 bi1 = A.layer[1].bias ? ld4(A.layer[1].bias + c.col0) : zero4;
  const float4 bi2 = A.layer[2].bias ? ld4(A.layer[2].bias + c.col0) : zero4;
  const int padld4 = A.padd_ld * 4;

  // the first tile's rows and gather rows
  float4 xv[TG];
  int is[TG], ir[TG];
#pragma unroll
  for (int q = 0; q < TG; ++q) {
    xv[q] = cb_ld4(bx, cb_off(c, 16 * g_beg, q));
    if (PADD) {
      const int ro = min(16 * g_beg + 16 * q + c.j, c.M - 1) * 4;
      is[q] = __builtin_amdgcn_raw_buffer_load_b32(bis, ro, 0, 0);
      ir[q] = __builtin_amdgcn_raw_buffer_load_b32(bir, ro, 0, 0);
    }
  }

  for (int t0 = g_beg; t0 < g_end; t0 += TG) {
    c.row0 = 16 * t0;
    c.ngt = min(TG, g_end - t0);
    const int np = (c.ngt + 1) >> 1;
    const int next_row0 = t0 + TG < g_end ? 16 * (t0 + TG) : c.M;
    int offL[TG], offS[TG];
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const int row = c.row0 + 16 * q + c.j;
      offL[q] = cb_off(c, c.row0, q);
      offS[q] = (q < c.ngt && row < c.M) ? row * 512 + c.col0 * 4 : CB_OFF_DEAD;
    }
    // ---- P0: this wave's share of the row maxima; the addend rows and layer 0's slice go out ----
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const float m = row_max4(max3_abs(max3_abs(0.f, xv[q].x, xv[q].y), xv[q].z, xv[q].w));
      if (c.g == 0) rmax[(q * 16 + c.j) * 8 + c.w] = m;
    }
    float4 ps[PADD ? TG : 1], pr[PADD ? TG : 1];
    if (PADD) {
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        ps[q] = cb_ld4(bpad, is[q] * padld4 + c.col0 * 4);
        pr[q] = cb_ld4(bpad, ir[q] * padld4 + 512 + c.col0 * 4);
      }
    }
    gfv_f16x8 wh[4], wl[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      wh[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w0, woff + T * 16384, 0, 0));
      wl[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w0, woff + T * 16384 + 1024, 0, 0));
    }
    cc_barrier();
    // ---- P0b: row scales, input fragments ----
    float sinv[TG];
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const float4* pp = reinterpret_cast<const float4*>(rmax + (q * 16 + c.j) * 8);
      const float4 a = pp[0], b = pp[1];
      const float sc = gfv_pow2_scale(fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w))));
      sinv[q] = 1.0f / sc;
      const float x4[4] = {xv[q].x, xv[q].y, xv[q].z, xv[q].w};
      cc_put_frag(xin, q, c, x4, sc);
    }
    cc_barrier();
    // ---- P1: layer 0, xin -> xmid ----
    {
      floatx4 a0, a1;
      cc_mma_pair<4, LOWP, true>(xin, 0, wh, wl, c.lane, a0, a1);
#pragma unroll
      for (int p = 0; p < TG / 2; ++p) {
        if (p >= np) break;
        floatx4 n0 = a0, n1 = a1;
        if (p + 1 < TG / 2) cc_mma_pair<4, LOWP, true>(xin, p + 1, wh, wl, c.lane, n0, n1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int q = 2 * p + h;
          const floatx4& acc = h ? a1 : a0;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (acc[r] * sinv[q]) * c.invw;
          v[0] += bi0.x; v[1] += bi0.y; v[2] += bi0.z; v[3] += bi0.w;
          if (PADD) {
            v[0] += ps[q].x + pr[q].x; v[1] += ps[q].y + pr[q].y; v[2] += ps[q].z + pr[q].z; v[3] += ps[q].w + pr[q].w;
          }
          cb_st4(bz1, offS[q], v);
          float a[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) a[r] = gfv_gelu(v[r]);
          const float mq = max3_abs(max3_abs(0.f, a[0], a[1]), a[2], a[3]);
          c.mabs = fmaxf(c.mabs, q < c.ngt ? mq : 0.f);
          cc_put_frag(xmid, q, c, a, CC_SH);
        }
        a0 = n0; a1 = n1;
      }
    }
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      wh[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w1, woff + T * 16384, 0, 0));
      wl[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w1, woff + T * 16384 + 1024, 0, 0));
    }
    cc_barrier();
    // ---- P2: layer 1, xmid -> xin ----
    {
      floatx4 a0, a1;
      cc_mma_pair<4, LOWP, true>(xmid, 0, wh, wl, c.lane, a0, a1);
#pragma unroll
      for (int p = 0; p < TG / 2; ++p) {
        if (p >= np) break;
        floatx4 n0 = a0, n1 = a1;
        if (p + 1 < TG / 2) cc_mma_pair<4, LOWP, true>(xmid, p + 1, wh, wl, c.lane, n0, n1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int q = 2 * p + h;
          const floatx4& acc = h ? a1 : a0;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (acc[r] * CC_SH_INV) * c.invw;
          v[0] += bi1.x; v[1] += bi1.y; v[2] += bi1.z; v[3] += bi1.w;
          cb_st4(bz2, offS[q], v);
          float a[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) a[r] = gfv_gelu(v[r]);
          const float mq = max3_abs(max3_abs(0.f, a[0], a[1]), a[2], a[3]);
          c.mabs = fmaxf(c.mabs, q < c.ngt ? mq : 0.f);
          cc_put_frag(xin, q, c, a, CC_SH);
        }
        a0 = n0; a1 = n1;
      }
    }
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      wh[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w2, woff + T * 16384, 0, 0));
      wl[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w2, woff + T * 16384 + 1024, 0, 0));
    }
    // the residual rows: in flight through P3
    float4 rr[TG];
#pragma unroll
    for (int q = 0; q < TG; ++q) rr[q] = cb_ld4(bres, offL[q]);
    cc_barrier();
    // ---- P3: layer 2, xin -> y3 (saved), LayerNorm partials; the values stay in registers ----
    float y[TG][4];
    {
      floatx4 a0, a1;
      cc_mma_pair<4, LOWP, true>(xin, 0, wh, wl, c.lane, a0, a1);
#pragma unroll
      for (int p = 0; p < TG / 2; ++p) {
        floatx4 n0 = a0, n1 = a1;
        if (p + 1 < TG / 2) cc_mma_pair<4, LOWP, true>(xin, p + 1, wh, wl, c.lane, n0, n1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int q = 2 * p + h;
          const floatx4& acc = h ? a1 : a0;
#pragma unroll
          for (int r = 0; r < 4; ++r) y[q][r] = (acc[r] * CC_SH_INV) * c.invw;
          y[q][0] += bi2.x; y[q][1] += bi2.y; y[q][2] += bi2.z; y[q][3] += bi2.w;
          cb_st4(by3, offS[q], y[q]);
          const float mw = row_sum((y[q][0] + y[q][1]) + (y[q][2] + y[q][3])) * 0.0625f;
          const float d0 = y[q][0] - mw, d1 = y[q][1] - mw, d2 = y[q][2] - mw, d3 = y[q][3] - mw;
          const float m2 = row_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
          if (c.g == 0) *reinterpret_cast<float2*>(lnp + ((q * 16 + c.j) * 8 + c.w) * 2) = make_float2(mw, m2);
        }
        a0 = n0; a1 = n1;
      }
    }
    // the next tile's rows and gather rows: requested here, in flight through P4 and the next tile's P0
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      xv[q] = cb_ld4(bx, cb_off(c, next_row0, q));
      if (PADD) {
        const int ro = min(next_row0 + 16 * q + c.j, c.M - 1) * 4;
        is[q] = __builtin_amdgcn_raw_buffer_load_b32(bis, ro, 0, 0);
        ir[q] = __builtin_amdgcn_raw_buffer_load_b32(bir, ro, 0, 0);
      }
    }
    cc_barrier();
    // ---- P4: LayerNorm over the eight waves' partials, affine, residual, stores ----
    {
      const float4 gam = ld4(A.fin_gamma + c.col0), bet = ld4(A.fin_beta + c.col0);
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        const float4* pp = reinterpret_cast<const float4*>(lnp + (q * 16 + c.j) * 16);
        const float4 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];   // (mean, M2) x 8 waves
        const float mean = (((p0.x + p0.z) + (p1.x + p1.z)) + ((p2.x + p2.z) + (p3.x + p3.z))) * 0.125f;
        const float e0 = p0.x - mean, e1 = p0.z - mean, e2 = p1.x - mean, e3 = p1.z - mean, e4 = p2.x - mean,
                    e5 = p2.z - mean, e6 = p3.x - mean, e7 = p3.z - mean;
        const float m2 = (((p0.y + p0.w) + (p1.y + p1.w)) + ((p2.y + p2.w) + (p3.y + p3.w))) +
                         16.0f * (((e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3)) + ((e4 * e4 + e5 * e5) + (e6 * e6 + e7 * e7)));
        const float rstd = rsqrtf(m2 * 0.0078125f + 1e-5f);   // nn.LayerNorm eps (EPD.py:32)
        if (c.w == 0 && c.g == 0) {
          const int row = c.row0 + 16 * q + c.j;
          const cb_f32x2 st = {mean, rstd};
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(cb_i32x2, st), bst, (q < c.ngt && row < c.M) ? row * 8 : CB_OFF_DEAD, 0, 0);
        }
        float o[4] = {(y[q][0] - mean) * rstd * gam.x + bet.x, (y[q][1] - mean) * rstd * gam.y + bet.y,
                      (y[q][2] - mean) * rstd * gam.z + bet.z, (y[q][3] - mean) * rstd * gam.w + bet.w};
        cb_st4(bnr, offS[q], o);
        o[0] += rr[q].x; o[1] += rr[q].y; o[2] += rr[q].z; o[3] += rr[q].w;
        cb_st4(bout, offS[q], o);
      }
    }
    // (the next tile's P0 writes only rmax, last read in P0b; xin / xmid / lnp are rewritten behind its barriers)
  }
  if (c.mabs > CC_SH_LIMIT) atomicOr(status, 2);
}

}  // namespace
