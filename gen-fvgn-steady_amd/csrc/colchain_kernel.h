// Column-owner persistent backward chain (gfx950): the kernel family behind gfv_rowtile_chain (contract: include/gfv.h) for
// the backward of the big 3-layer MLP launches (EPD.py:10-33 build_mlp inside blocks.py EdgeBlock / NodeBlock, the encoders)
// in the split-fp16 product form, with the weight gradients fused (gfv_rowtile_args_t.dw_partial).
//
// tchain_kernel.h gives every wave 16 ROWS and streams the layers' weight images through LDS; the 128 x 128 output of a
// weight gradient G^T A is then spread over rows no wave owns.  Here the roles are swapped:
//
//   * ONE workgroup of 8 waves per CU, persistent over a contiguous range of 16-row groups.  Wave w owns output COLUMNS
//     16 w .. 16 w + 15 of every 128-wide layer; its slice of a layer's weight image - the A operands W[16 w + i][k] of
//     v_mfma_f32_16x16x32_f16, hi and lo parts, 32 VGPRs per 128-deep layer - is fetched per tile one phase ahead of its use
//     (4 KB per wave, L2 hits), and ITS n-tile of every fused weight gradient lives in registers for the whole launch.
//   * LDS holds only activations, already in MFMA B-fragment form ([group][k-group T][part][lane] x 16 B: what
//     to_halves() of the row-owner kernel builds in registers).  A tile is TG groups of 16 rows; per layer every
//     wave reads all of the tile's fragments (one conflict-free ds_read_b128 per fragment), runs 12 MFMAs per group
//     against its weight slice, applies the element ops to its 16 columns and writes its 8-byte share of the next
//     layer's fragments (the columns a wave produces are exactly half a k-group of the next layer: T' = w >> 1,
//     slots 4 (w & 1) .. + 3).  One barrier per phase.
//   * Hidden activations (GELU outputs) are split after a FIXED power-of-two scale CC_SH: a row scale would
//     need the row maximum over all eight waves (a second barrier per layer), and the split has 2^16 of slack - a hidden
//     row with max |a| in [2^-4, 2^11] keeps every product at fp32 accuracy; beyond 2^11 the status flag
//     GFV_FLAG_CHAIN_RANGE is raised (the values still convert up to 4095).
//
// (Round 3 also had two generations of a column-owner FORWARD here; parity-green and never faster than the row-owner chain -
// DESIGN.md 5 - they were removed in round 4.)
#pragma once
#include "tchain_kernel.h"

namespace {

// (Round 4's ablation switches - GFV_ABL: no MFMA / no fragment reads / no erfc; GFV_DW24, GFV_SKEW_WAVES, GFV_FENCE_MASK - are
// gone with round 5: their results are recorded in profiles/r04_colchain_phases.txt, the kept forms are the code below.)
constexpr int CC_W = 8;                  // waves per workgroup
constexpr float CC_SH = 16.0f;           // fixed scale of hidden activations ahead of the fp16 split
constexpr float CC_SH_INV = 1.0f / 16.0f;
constexpr float CC_SH_LIMIT = 2048.0f;   // |a| beyond this raises GFV_FLAG_CHAIN_RANGE
// -DGFV_CC_TIMING: per wave, cycles spent in each phase and at each barrier, written through fin_presave's ... no:
// through `status` + 64 (a debug build takes a bigger status buffer; scratch experiments only)
#ifdef GFV_CC_TIMING
#define CT_DECL long long ct_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long ct_prev_ = clock64();
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
template <int KT, int LOWP, bool FENCE = false>   // LOWP: 0 three products, 1 / 2 the single-product forms (fp16 / bf16)
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
    a0 = gfv_mma_hh<LOWP == 2>(wh[T], xh0, a0);
    a1 = gfv_mma_hh<LOWP == 2>(wh[T], xh1, a1);
    // (FENCE: the scheduler may not hoist the next k-group's four fragment reads above this one's MFMAs - at a budget of 128
    // registers sixteen fragments in flight are 64 of them)
    if (FENCE) __builtin_amdgcn_sched_barrier(0);
  }
}

// this lane's 4 values of one row -> its 8-byte share of the next layer's fragments (k-group w >> 1, half w & 1)
template <bool BF>   // BF: the bf16 single-product form (the high parts in bf16; the low slots are written but never read)
__device__ __forceinline__ void cc_put_frag(char* xbuf, int q, const CcCtx& c, const float (&a)[4], float scale) {
  unsigned h0, h1, l0, l1;
  const gfv_f2 s01 = gfv_f2{a[0], a[1]} * gfv_splat2(scale), s23 = gfv_f2{a[2], a[3]} * gfv_splat2(scale);
  gfv_split_pair_t<BF>(s01.x, s01.y, h0, l0);
  gfv_split_pair_t<BF>(s23.x, s23.y, h1, l1);
  char* dst = xbuf + (size_t)((q * 4 + (c.w >> 1)) * 2) * 1024 + c.lane * 16 + (c.w & 1) * 8;
  *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(dst + 1024) = make_uint2(l0, l1);
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
//
// RC (gfv_rowtile_args_t.rc_Wh: recompute instead of re-read).  The forward then saves only z1 (and the row statistics); this
// kernel rebuilds what the LayerNorm backward and the GELU' factors need, on a matrix pipe that sat at 13 %:
//   R1   a1 = gelu(z1) (rows loaded a tile ahead) -> fragments (a FOURTH buffer: a1 lives until the weight-gradient phase)
//   R2   (barrier) z2 = W2 a1 + b2 against the FORWARD image of W2; a2 = gelu(z2) -> fragments, gelu'(z2) parked in LDS (in the
//        bytes of the gz2 buffer this lane will overwrite in P3)
//   R3   (barrier) y3 = W3 a2 + b3 -> registers; then P0 .. P3 as above (P3's epilogue multiplies by the parked gelu'(z2))
//   dW   (barrier) dW3 += g3^T a2 AND dW2 += gz2^T a1 in one phase (both operand pairs exist by then), which frees the a1
//        buffer two barriers ahead of the next tile's R1 - no barrier between tiles
//   P2   (barrier) chain layer 1 with gelu'(z1) from z1 re-read (an L2 hit: this tile's rows were read in R1);  P1 chain layer 2
// Per row the launch reads dy, z1 (+ residual) and writes gz1 and the input gradient: y3 and z2 are neither written by the
// forward nor read here (2 x 512 B per row less in each direction); 7 barriers and 7 matrix phases per tile instead of 6 and 5.
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

// one tile's contribution to a fused weight gradient: D += G^T x A over the tile's row pairs; accb += G^T x ones (the bias
// gradient: every column of the result is the column sum of G).
// A wave owns a 2 x 4 block of the 8 x 8 output tiles - n-tiles 2 (w >> 1) + {0, 1}, k-tiles 4 (w & 1) + {0 .. 3}; acc[4 nn + kk] -
// and the bias gradient of n-tile 2 (w >> 1) + (w & 1).  (The first form gave a wave one n-tile and all eight k-tiles: 36
// transposed LDS reads per row pair where this takes 24 - every wave read ALL of A - and the weight-gradient phases are
// LDS-read time: DESIGN.md 5.)
__device__ __forceinline__ int cb_dw_ntile(int w, int i) { return 2 * (w >> 1) + (i >> 2); }
__device__ __forceinline__ int cb_dw_ktile(int w, int i) { return 4 * (w & 1) + (i & 3); }
__device__ __forceinline__ int cb_dw_btile(int w) { return 2 * (w >> 1) + (w & 1); }
template <int LOWP>
__device__ __forceinline__ void cb_dw_tile(const char* gbuf, const char* abuf, int npairs, int w, int lane, floatx4 (&acc)[8],
                                           floatx4& accb) {
  const gfv_f16x8 ones = gfv_frag_ones<LOWP == 2>();
  const int nt0 = 2 * (w >> 1), kt0 = 4 * (w & 1);
  for (int pr = 0; pr < npairs; ++pr) {
    gfv_f16x8 gh[2], gl[2];
    cb_tr_operand(gbuf, 2 * pr, nt0, lane, gh[0], gl[0]);
    cb_tr_operand(gbuf, 2 * pr, nt0 + 1, lane, gh[1], gl[1]);
    {
      const bool odd = (w & 1) != 0;   // (wave-uniform)
      const gfv_f16x8 bh = odd ? gh[1] : gh[0], bl = odd ? gl[1] : gl[0];
      if (!LOWP) accb = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ones, accb, 0, 0, 0);
      accb = gfv_mma_hh<LOWP == 2>(bh, ones, accb);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      gfv_f16x8 ah, al;
      cb_tr_operand(abuf, 2 * pr, kt0 + kk, lane, ah, al);
#pragma unroll
      for (int nn = 0; nn < 2; ++nn) {
        if (!LOWP) {
          acc[4 * nn + kk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gl[nn], ah, acc[4 * nn + kk], 0, 0, 0);
          acc[4 * nn + kk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh[nn], al, acc[4 * nn + kk], 0, 0, 0);
        }
        acc[4 * nn + kk] = gfv_mma_hh<LOWP == 2>(gh[nn], ah, acc[4 * nn + kk]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // (one row pair's operands in flight: the register budget)
  }
}

// chain-layer epilogue of one group in the backward form (GFV_OP_MUL_DGELU): v = acc / scales x gelu'(z) -> fragments with
// the scale `sg`; a = gelu(z) -> fragments with CC_SH (the weight gradient's other operand); v is handed back for the save
template <bool BF>
__device__ __forceinline__ void cb_hidden_bwd(CcCtx& c, int q, const floatx4& acc, float inv_in, const float4& z, float sg,
                                              char* gout, char* aout, float (&v)[4]) {
  // two values per instruction (packed fp32, gfv_common.h): the same operations in the same order as the scalar form
  const gfv_f2 z01 = {z.x, z.y}, z23 = {z.z, z.w};
  gfv_f2 a01, a23, d01, d23;
  gfv_gelu_dgelu2(z01, a01, d01);   // gelu and gelu' share the erfc evaluation
  gfv_gelu_dgelu2(z23, a23, d23);
  const gfv_f2 ki = gfv_splat2(inv_in), kw = gfv_splat2(c.invw);
  const gfv_f2 v01 = ((gfv_f2{acc[0], acc[1]} * ki) * kw) * d01, v23 = ((gfv_f2{acc[2], acc[3]} * ki) * kw) * d23;
  v[0] = v01.x; v[1] = v01.y; v[2] = v23.x; v[3] = v23.y;
  const float a[4] = {a01.x, a01.y, a23.x, a23.y};
  const float mq = max3_abs(max3_abs(0.f, v[0], v[1]), v[2], v[3]) * sg;
  const float ma = max3_abs(max3_abs(0.f, a[0], a[1]), a[2], a[3]) * (60000.0f / CC_SH_LIMIT);   // (|a| beyond CC_SH_LIMIT raises the flag: mabs is compared with 60000)
  c.mabs = fmaxf(c.mabs, q < c.ngt ? fmaxf(mq, ma) : 0.f);
  cc_put_frag<BF>(gout, q, c, v, sg);
  cc_put_frag<BF>(aout, q, c, a, CC_SH);
}

// (RC) this lane's 16 bytes of a fragment buffer - the two 8-byte slots cc_put_frag(xbuf, q, ...) will write - as a parking place
// for four floats of its own: the lane reads them back right before it writes the fragments there, no other lane touches them
__device__ __forceinline__ char* cb_own_slot(char* xbuf, int q, const CcCtx& c) {
  return xbuf + (size_t)((q * 4 + (c.w >> 1)) * 2) * 1024 + c.lane * 16 + (c.w & 1) * 8;
}
__device__ __forceinline__ void cb_park4(char* xbuf, int q, const CcCtx& c, const float (&v)[4]) {
  char* d = cb_own_slot(xbuf, q, c);
  *reinterpret_cast<float2*>(d) = make_float2(v[0], v[1]);
  *reinterpret_cast<float2*>(d + 1024) = make_float2(v[2], v[3]);
}
__device__ __forceinline__ void cb_unpark4(char* xbuf, int q, const CcCtx& c, float (&v)[4]) {
  const char* d = cb_own_slot(xbuf, q, c);
  const float2 a = *reinterpret_cast<const float2*>(d), b = *reinterpret_cast<const float2*>(d + 1024);
  v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}

// the same with the GELU' factor given (RC: kept from the recompute phase, or taken from z alone) and no activation output
template <bool BF>
__device__ __forceinline__ void cb_hidden_bwd_dg(CcCtx& c, int q, const floatx4& acc, float inv_in, const float (&dg)[4], float sg,
                                                 char* gout, float (&v)[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = ((acc[r] * inv_in) * c.invw) * dg[r];
  const float mq = max3_abs(max3_abs(0.f, v[0], v[1]), v[2], v[3]) * sg;
  c.mabs = fmaxf(c.mabs, q < c.ngt ? mq : 0.f);
  cc_put_frag<BF>(gout, q, c, v, sg);
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
__device__ __forceinline__ void cb_st4v(cb_rsrc r, int off, const floatx4& f) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cb_i32x4, f), r, off, 0, 0);
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
// (RC) the incoming gradient rows and the forward's row statistics only: y3 is recomputed
__device__ __forceinline__ void cb_load_dy_stats(const CbBufs& B, const CcCtx& c, int row0, CbIn& in) {
#pragma unroll
  for (int q = 0; q < CB_TG; ++q) {
    in.dy[q] = cb_ld4(B.dy, cb_off(c, row0, q));
    in.st[q] = cb_ld2(B.stats, min(row0 + 16 * q + c.j, c.M - 1) * 8);
  }
}
__device__ __forceinline__ void cb_load_rows(cb_rsrc r, const CcCtx& c, int row0, float4 (&v)[CB_TG]) {
#pragma unroll
  for (int q = 0; q < CB_TG; ++q) v[q] = cb_ld4(r, cb_off(c, row0, q));
}
template <bool GADD>
__device__ __forceinline__ void cb_load_gathers(const CbBufs& B, const CcCtx& c, const int (&gidx)[CB_TG], CbIn& in) {
  if (GADD) {   // [gadd[s] (64) | gadd[r] (64)]: this wave's 16 columns lie in one half
#pragma unroll
    for (int q = 0; q < CB_TG; ++q) in.ga[q] = cb_ld4(B.gadd, gidx[q] * 256 + (c.col0 & 63) * 4);
  }
}

// GADD: the gathered addend [gadd[s] | gadd[r]] of the incoming gradient exists.
// (DW1 - the first Linear's weight gradient fused as well, per tile in rounds 3 - 4 (18 - 43 spilled registers), as a trailing pass
// of every workgroup in round 5 (3 spills, parity-green, TIME-NEUTRAL: 3.71 - 3.76 against 3.72 - 3.74 ms, cavity +2.5 %:
// profiles/r05_ab_dw1_trailing.txt) - was removed in round 6 with its six instantiations: that weight gradient stays a one-tile
// launch of the side queue.)
// OUT2: the last chain layer is 192 wide (NodeBlock: W1^T with the rows for x first, then the 64 for the neighbour mean,
// blocks.py:54): waves 0..3 own a second n-tile and write out[1] ([M, 64], no residual)
// NOOUT: the MLP's input needs no gradient (the encoders, EPD.py:92-119): a two-layer launch whose out[0] receives gz1 (what the
// first Linear's weight-gradient launch reads); the third chain phase is only the weight gradient of the second Linear
// RC: z2 and y3 are recomputed from z1 (rc_Wh / rc_bias: the forward's second and third Linear) instead of read
template <int LOWP, bool GADD, bool OUT2 = false, bool NOOUT = false, bool RC = false>
__global__ __launch_bounds__(64 * CC_W, 2) void colchain_bwd_kernel(const gfv_rowtile_args_t A, int* status) {
  constexpr int TG = CB_TG;
  __shared__ __attribute__((aligned(16))) char lds[CbLds::TOTAL + (RC ? CbLds::BUF : 0)];
  char* b0 = lds + CbLds::B0;
  char* b1 = lds + CbLds::B1;
  char* b2 = lds + CbLds::B2;
  char* b3 = lds + CbLds::TOTAL;   // (RC) a1 = gelu(z1)
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
  B.y3 = cb_buf(RC ? nullptr : A.in_aux, rows128);
  B.stats = cb_buf(A.in_stats, (size_t)A.M * 8);
  B.z2 = cb_buf(RC ? nullptr : A.layer[0].aux, rows128);
  B.z1 = cb_buf(A.layer[1].aux, rows128);
  B.gadd = cb_buf(A.gadd, 0x7fffffe0ull);   // (its row count is not an argument; the gather rows come from the index arrays)
  B.gidx = cb_buf(c.w < 4 ? A.gadd_s : A.gadd_r, (size_t)A.M * 4);
  B.save1 = cb_buf(NOOUT ? A.out[0] : A.layer[1].save, rows128);
  B.res = cb_buf(NOOUT ? nullptr : A.res[0], rows128);
  B.out = cb_buf(NOOUT ? nullptr : A.out[0], rows128);
  const cb_rsrc out2 = cb_buf(OUT2 ? A.out[1] : nullptr, (size_t)A.M * 256);

  // this wave's n-tile of the three transposed layers' images.  The weight-gradient accumulators take 72 registers
  // for the whole launch, so the chain's weights are NOT resident here: every tile fetches each layer's slice (4 KB per wave,
  // L2 hits) one phase ahead of its use - 12 KB per wave and 64-row tile next to 230 KB of activation traffic - and the
  // LayerNorm-backward phase, where the register pressure peaks, holds none of them.
  const cb_rsrc w0 = cb_buf(A.layer[0].Wh, 65536), w1 = cb_buf(A.layer[1].Wh, 65536),
                w2 = cb_buf(NOOUT ? nullptr : A.layer[2].Wh, OUT2 ? 131072 : 65536);
  // (RC) the FORWARD images of the second and third Linear, for the recompute phases
  const cb_rsrc rw2 = cb_buf(RC ? A.rc_Wh[0] : nullptr, 65536), rw3 = cb_buf(RC ? A.rc_Wh[1] : nullptr, 65536);
  const int woff = (c.w * 128 + c.lane) * 16;   // + T * 16384 (+ 1024: the lo part)
  // fused weight gradients: D[n = 16 w + 4 g + r][k = 16 kt + j] in lane (j, g) of acc[kt][r]
  floatx4 dw3[8], dw2[8], db3 = floatx4{0.f, 0.f, 0.f, 0.f}, db2 = db3;
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) dw3[kt] = dw2[kt] = floatx4{0.f, 0.f, 0.f, 0.f};
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
        if (LOWP == 2) {   // (the bf16 form's image: bf16 high parts, zero low parts)
          const gfv_bf16x8 hb = __builtin_bit_cast(gfv_bf16x8, h);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc += fabsf((float)hb[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc += fabsf((float)h[e] + (float)lo[e]);
        }
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

  constexpr bool PRE = true;
  int gidx[TG] = {0, 0, 0, 0};
  CbIn in;
  float4 zn[RC ? TG : 1];   // (RC) the next tile's z1 rows
  if (g_beg < g_end) {
    cb_load_gidx<GADD>(B, c, 16 * g_beg, gidx);
    if constexpr (RC) {
      cb_load_rows(B.z1, c, 16 * g_beg, zn);
      cb_load_dy_stats(B, c, 16 * g_beg, in);
      cb_load_gathers<GADD>(B, c, gidx, in);
    } else if (PRE) {
      cb_load_inputs(B, c, 16 * g_beg, in);
      cb_load_gathers<GADD>(B, c, gidx, in);
    }
  }
  // The second wave of each SIMD (waves 4 .. 7) runs a chain phase as MMA(0) EPI(0) MMA(1) EPI(1) where the first runs
  // MMA(0) MMA(1) EPI(0) EPI(1): its epilogue (vector arithmetic, stores) then falls beside the other wave's matrix instructions
  // instead of both queueing for the matrix pipe and then for the vector unit.  Same operations per wave, bit-identical results;
  // 42.1 k -> 40.3 k cycles per tile (most of it in P1, whose epilogue is stores).
  const bool CB_LATE = c.w >= 4;
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
    if constexpr (RC) {
      gfv_f16x8 rh[4], rl[4];   // the forward slice of the phase that follows
#pragma unroll
      for (int T = 0; T < 4; ++T) {
        rh[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(rw2, woff + T * 16384, 0, 0));
        rl[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(rw2, woff + T * 16384 + 1024, 0, 0));
      }
      const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
      // ---- R1: a1 = gelu(z1) -> b3 (free since the previous tile's weight-gradient phase, two barriers back) ----
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        const float z4[4] = {zn[q].x, zn[q].y, zn[q].z, zn[q].w};
        float a[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = gfv_gelu(z4[r]);
        const float ma = max3_abs(max3_abs(0.f, a[0], a[1]), a[2], a[3]) * (60000.0f / CC_SH_LIMIT);
        c.mabs = fmaxf(c.mabs, q < c.ngt ? ma : 0.f);
        cc_put_frag<LOWP == 2>(b3, q, c, a, CC_SH);
      }
      CT(12);
      cc_barrier();
      CT(13);
      // ---- R2: z2 = W2 a1 + b2;  a2 = gelu(z2) -> b2, gelu'(z2) kept ----
#pragma unroll
      for (int q = 0; q < TG; ++q) in.yv[q] = zero4;   // (pairs past the tile's end: zeros, not whatever LDS held)
      {
        // (the bias rows are fetched where they are used - an L1 hit per tile - through a descriptor and a 32-bit offset: as a
        // plain load the compiler hoists them out of the tile loop, eight registers alive through every phase, or keeps their
        // 64-bit addresses in registers and spills those; `tvar` makes the offset look loop-variant)
        int tvar = 0;
        asm volatile("" : "+v"(tvar));
        const float4 bi2 = cb_ld4(cb_buf(A.rc_bias[0], 512), c.col0 * 4 + tvar);   // (NULL: zero records, reads as zeros)
        floatx4 a0, a1;
        cc_mma_pair<4, LOWP, true>(b3, 0, rh, rl, c.lane, a0, a1);
#pragma unroll
        for (int p = 0; p < TG / 2; ++p) {
          if (p >= np) break;
          floatx4 n0 = a0, n1 = a1;
          if (p + 1 < TG / 2) cc_mma_pair<4, LOWP, true>(b3, p + 1, rh, rl, c.lane, n0, n1);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int q = 2 * p + h;
            const floatx4& acc = h ? a1 : a0;
            const float bb[4] = {bi2.x, bi2.y, bi2.z, bi2.w};
            float a[4], dg[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float z = (acc[r] * CC_SH_INV) * c.invw + bb[r];
              const gfv_erfc_t e = gfv_erfc_half(z);   // gelu and gelu' share the erfc evaluation (gfv_common.h)
              const float cdf = z >= 0.0f ? 1.0f - e.y : e.y;
              dg[r] = fmaf(z * 0.39894228040143267794f, e.e, cdf);
              a[r] = fmaf(-fabsf(z), e.y, fmaxf(z, 0.0f));
            }
            // gelu'(z2) waits for P3 in b1 (free since the previous tile's second chain layer): in the very bytes this lane
            // will write its share of the gz2 fragments to - 16 registers less from here to P3, where the kernel peaks
            cb_park4(b1, q, c, dg);
            const float ma = max3_abs(max3_abs(0.f, a[0], a[1]), a[2], a[3]) * (60000.0f / CC_SH_LIMIT);
            c.mabs = fmaxf(c.mabs, q < c.ngt ? ma : 0.f);
            cc_put_frag<LOWP == 2>(b2, q, c, a, CC_SH);
          }
          a0 = n0; a1 = n1;
        }
      }
#pragma unroll
      for (int T = 0; T < 4; ++T) {
        rh[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(rw3, woff + T * 16384, 0, 0));
        rl[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(rw3, woff + T * 16384 + 1024, 0, 0));
      }
      CT(14);
      cc_barrier();
      CT(15);
      // ---- R3: y3 = W3 a2 + b3 -> registers (what the launch would otherwise read from in_aux) ----
      {
        int tvar = 0;
        asm volatile("" : "+v"(tvar));
        const float4 bi3 = cb_ld4(cb_buf(A.rc_bias[1], 512), c.col0 * 4 + tvar);
        floatx4 a0, a1;
        cc_mma_pair<4, LOWP, true>(b2, 0, rh, rl, c.lane, a0, a1);
#pragma unroll
        for (int p = 0; p < TG / 2; ++p) {
          if (p >= np) break;
          floatx4 n0 = a0, n1 = a1;
          if (p + 1 < TG / 2) cc_mma_pair<4, LOWP, true>(b2, p + 1, rh, rl, c.lane, n0, n1);
          in.yv[2 * p] = make_float4((a0[0] * CC_SH_INV) * c.invw + bi3.x, (a0[1] * CC_SH_INV) * c.invw + bi3.y,
                                     (a0[2] * CC_SH_INV) * c.invw + bi3.z, (a0[3] * CC_SH_INV) * c.invw + bi3.w);
          in.yv[2 * p + 1] = make_float4((a1[0] * CC_SH_INV) * c.invw + bi3.x, (a1[1] * CC_SH_INV) * c.invw + bi3.y,
                                         (a1[2] * CC_SH_INV) * c.invw + bi3.z, (a1[3] * CC_SH_INV) * c.invw + bi3.w);
          a0 = n0; a1 = n1;
        }
      }
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
      float* part_j = part + (c.j * 8 + c.w) * 2;   // (one address + immediate offsets: four separate addresses were spilled)
      const float4 gam = RC ? cb_ld4(cb_buf(A.in_gamma, 512), c.col0 * 4) : ld4(A.in_gamma + c.col0);
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
        // (value pairs on the packed-fp32 instructions, gfv_common.h; every value sees the operations of the scalar form)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = 2 * h;
          const gfv_f2 dd = gfv_f2{d[r], d[r + 1]} * gfv_splat2(lf);
          const gfv_f2 xx = (gfv_f2{y[r], y[r + 1]} - gfv_splat2(mean)) * gfv_splat2(rstd);
          const gfv_f2 dx = dd * xx;
          const gfv_f2 g2 = dd * gfv_f2{gm[r], gm[r + 1]};
          const gfv_f2 gx = g2 * xx;
          const gfv_f2 ng = gfv_f2{dgam[r], dgam[r + 1]} + dx, nb = gfv_f2{dbet[r], dbet[r + 1]} + dd;
          dgam[r] = ng.x; dgam[r + 1] = ng.y;
          dbet[r] = nb.x; dbet[r + 1] = nb.y;
          xh[q][r] = xx.x; xh[q][r + 1] = xx.y;
          gg[q][r] = g2.x; gg[q][r + 1] = g2.y;
          s1 += g2.x; s1 += g2.y;
          s2 += gx.x; s2 += gx.y;
          am = max3_abs(am, g2.x, g2.y);
        }
        s1 = row_sum(s1);
        s2 = row_sum(s2);
        if (c.g == 0) *reinterpret_cast<float2*>(part_j + q * 256) = make_float2(s1, s2);   // part[((q * 16 + j) * 8 + w) * 2]
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
    if constexpr (!RC) {
#pragma unroll
      for (int q = 0; q < TG; ++q) zq[q] = cb_ld4(B.z2, offL[q]);
    }
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
        for (int h = 0; h < 2; ++h) {
          const int r = 2 * h;
          const gfv_f2 t = gfv_splat2(rs[q]) * ((gfv_f2{gg[q][r], gg[q][r + 1]} - gfv_splat2(m1)) - gfv_f2{xh[q][r], xh[q][r + 1]} * gfv_splat2(m2));
          g3[r] = t.x; g3[r + 1] = t.y;
        }
        if (A.in_save) cb_st4(cb_buf(A.in_save, rows128), offS[q], g3);
        cc_put_frag<LOWP == 2>(b0, q, c, g3, s3);
      }
    }
    // the accumulators move to this tile's units
    if (sacc != 0.f && sacc != s3) {
      const float ratio = s3 / sacc;
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) { dw3[kt] *= ratio; dw2[kt] *= ratio; }
      db3 *= ratio;
      db2 *= ratio;
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
        float v0[4], v1[4];
        if (!CB_LATE && p + 1 < TG / 2) cc_mma_pair<4, LOWP, true>(b0, p + 1, wh, wl, c.lane, n0, n1);
        if constexpr (RC) {   // gelu'(z2) was parked in b1 by R2, a2 is in b2 already
          float d0[4], d1[4];
          cb_unpark4(b1, 2 * p, c, d0);
          cb_unpark4(b1, 2 * p + 1, c, d1);
          cb_hidden_bwd_dg<LOWP == 2>(c, 2 * p, a0, inv_in, d0, s2s, b1, v0);
          cb_hidden_bwd_dg<LOWP == 2>(c, 2 * p + 1, a1, inv_in, d1, s2s, b1, v1);
        } else {
          cb_hidden_bwd<LOWP == 2>(c, 2 * p, a0, inv_in, zq[2 * p], s2s, b1, b2, v0);
          cb_hidden_bwd<LOWP == 2>(c, 2 * p + 1, a1, inv_in, zq[2 * p + 1], s2s, b1, b2, v1);
        }
        if (A.layer[0].save) {
          const cb_rsrc s0 = cb_buf(A.layer[0].save, rows128);
          cb_st4(s0, offS[2 * p], v0);
          cb_st4(s0, offS[2 * p + 1], v1);
        }
        if (CB_LATE && p + 1 < TG / 2) cc_mma_pair<4, LOWP, true>(b0, p + 1, wh, wl, c.lane, n0, n1);
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
    // (RC) the next tile's z1 rows: in flight through the weight gradients and the last two chain layers
    if constexpr (RC) cb_load_rows(B.z1, c, next_row0, zn);
    cb_dw_tile<LOWP>(b0, b2, np, c.w, c.lane, dw3, db3);
    if constexpr (RC) cb_dw_tile<LOWP>(b1, b3, np, c.w, c.lane, dw2, db2);   // gz2 and a1 both exist: frees b3 for the next tile's R1
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
        float v0[4], v1[4];
        if (!CB_LATE && p + 1 < TG / 2) cc_mma_pair<4, LOWP, true>(b1, p + 1, wh, wl, c.lane, n0, n1);
        if constexpr (RC) {   // a1 is in b3 since R1; only gelu'(z1) is needed here
          const float d0[4] = {gfv_dgelu(zq[2 * p].x), gfv_dgelu(zq[2 * p].y), gfv_dgelu(zq[2 * p].z), gfv_dgelu(zq[2 * p].w)};
          const float d1[4] = {gfv_dgelu(zq[2 * p + 1].x), gfv_dgelu(zq[2 * p + 1].y), gfv_dgelu(zq[2 * p + 1].z),
                               gfv_dgelu(zq[2 * p + 1].w)};
          cb_hidden_bwd_dg<LOWP == 2>(c, 2 * p, a0, inv_in, d0, s1s, b0, v0);
          cb_hidden_bwd_dg<LOWP == 2>(c, 2 * p + 1, a1, inv_in, d1, s1s, b0, v1);
        } else {
          cb_hidden_bwd<LOWP == 2>(c, 2 * p, a0, inv_in, zq[2 * p], s1s, b0, b2, v0);
          cb_hidden_bwd<LOWP == 2>(c, 2 * p + 1, a1, inv_in, zq[2 * p + 1], s1s, b0, b2, v1);
        }
        cb_st4(B.save1, offS[2 * p], v0);
        cb_st4(B.save1, offS[2 * p + 1], v1);
        if (CB_LATE && p + 1 < TG / 2) cc_mma_pair<4, LOWP, true>(b1, p + 1, wh, wl, c.lane, n0, n1);
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
        if (!CB_LATE && p + 1 < TG / 2) cc_mma_pair<4, LOWP, true>(b0, p + 1, wh, wl, c.lane, n0, n1);
        // (whole accumulator vectors: hipcc turns these into packed-fp32 instructions, two values each)
        const floatx4 o0 = (a0 * inv_in) * c.invw + floatx4{rr[2 * p].x, rr[2 * p].y, rr[2 * p].z, rr[2 * p].w};
        const floatx4 o1 = (a1 * inv_in) * c.invw + floatx4{rr[2 * p + 1].x, rr[2 * p + 1].y, rr[2 * p + 1].z, rr[2 * p + 1].w};
        cb_st4v(B.out, offS[2 * p], o0);
        cb_st4v(B.out, offS[2 * p + 1], o1);
        if (CB_LATE && p + 1 < TG / 2) cc_mma_pair<4, LOWP, true>(b0, p + 1, wh, wl, c.lane, n0, n1);
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
            const floatx4 o0 = (e0 * inv_in) * c.invw, o1 = (e1 * inv_in) * c.invw;
            const int ra = c.row0 + 32 * p + c.j, rb = ra + 16;
            cb_st4v(out2, (2 * p < c.ngt && ra < c.M) ? ra * 256 + c.col0 * 4 : CB_OFF_DEAD, o0);
            cb_st4v(out2, (2 * p + 1 < c.ngt && rb < c.M) ? rb * 256 + c.col0 * 4 : CB_OFF_DEAD, o1);
          }
        }
      }
    }
    CT(10);
    // the next tile's rows (and its gathered addend rows): in flight through the last weight gradients.  (Issued any earlier
    // they sit in 56 registers beside a chain phase, and the kernel spills: a scratch reload waits for every load in flight.)
    if constexpr (RC) {
      // the next tile's gradient rows, statistics and gathered addend: in flight through R1 .. R3.  (Issued at the start of P1
      // they sit beside the last chain layer's weight slice, residual rows and accumulators: 254 registers, the kernel spills.)
      cb_load_dy_stats(B, c, next_row0, in);
      cb_load_gathers<GADD>(B, c, gidx, in);
    } else {
      if (PRE) {
        cb_load_inputs(B, c, next_row0, in);
        cb_load_gathers<GADD>(B, c, gidx, in);
      }
      cb_dw_tile<LOWP>(b1, b2, np, c.w, c.lane, dw2, db2);
    }
    CT(11);
    // (the next tile's P0 writes only `part` / `smax`, last read in P0b; its P0b writes b0 behind the barrier that follows P0)
  }

#ifdef GFV_CC_TIMING
  if (c.lane == 0 && A.fin_aux) {
    long long* dbg = reinterpret_cast<long long*>(const_cast<float*>(A.fin_aux)) + ((size_t)blockIdx.x * CC_W + c.w) * 16;
    for (int kk = 0; kk < 16; ++kk) dbg[kk] = ct_[kk];
  }
#endif
  // ---- the workgroup's partial block: [dW3 | db3 | dW2 | db2 | dgamma | dbeta | dW1 | db1] (include/gfv.h) ----
  if (A.dw_partial) {
    float* blk = A.dw_partial + (size_t)blockIdx.x * A.dw_partial_stride;
    const float is = sacc != 0.f ? 1.0f / sacc : 0.f;
    const float r2 = 1.0f / step1, r1 = r2 / step2;   // the gz2 / gz1 sides are in units of sacc * step1 (* step2)
    const float u3 = is * CC_SH_INV, u2 = (is * r2) * CC_SH_INV;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = 16 * cb_dw_ntile(c.w, kt) + 4 * c.g + r, k = 16 * cb_dw_ktile(c.w, kt) + c.j;
        blk[n * 128 + k] = dw3[kt][r] * u3;
        blk[16384 + 128 + n * 128 + k] = dw2[kt][r] * u2;
      }
    if (c.j == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nb = 16 * cb_dw_btile(c.w) + 4 * c.g + r;
        blk[16384 + nb] = db3[r] * is;
        blk[2 * 16384 + 128 + nb] = db2[r] * (is * r2);
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


}  // namespace
