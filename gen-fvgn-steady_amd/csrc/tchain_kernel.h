// Register-resident fused GEMM chain on fp32 MFMA (gfx950): the fast path behind gfv_rowtile_chain - the kernel template,
// instantiated by tchain.hip and tchain_fwd.hip (two translation units: they are compiled with different vectoriser settings)
// (contract: include/gfv.h).  Used when every segment width is a multiple of 32 and every layer width a
// multiple of 128; rowtile.hip keeps the generic shapes.
//
// Idea: compute the TRANSPOSED product.  For v_mfma_f32_16x16x4_f32 the A operand is the weight tile
// (A[i][k] = W[16nt+i][k]) and the B operand the activations (B[k][j] = X[row j][k]), so the accumulator of lane
// (j = lane&15, g = lane>>4) holds out[row j][16nt + 4g + r], r = 0..3.  With the contraction index of MFMA step
// (t, s) chosen as k = 16t + 4g + s, the B operand of the NEXT layer for lane (j, g) at step (t, s) is exactly
// accumulator register (nt = t, r = s) of this layer: activations never leave the registers between the layers of the
// chain - no LDS round trip, no transposes, element ops (bias, GELU, GELU', LayerNorm, LayerNorm backward,
// residuals) act on the accumulators in place, and inputs / saved tensors / outputs move as float4 per lane straight
// between global memory and registers (one row per lane, the 4 g-lanes of a row cover 64 contiguous bytes).
//
// LDS holds only the weight stream: 32-wide k slices [128 n][32 k] (16 KB), double buffered, one barrier per slice.
// The image is written linearly (thread tid -> bytes 16*tid + 4096*p) with the 16-B chunk index XOR-swizzled on the
// SOURCE side (chunk c of row n sits in slot c ^ ((n>>1)&7)), which makes every ds_read_b128 lane group hit 16
// distinct slots: the A fragments of four consecutive MFMAs come from one conflict-free ds_read_b128.
//
// H instantiations (gfv_layer_t.Wh given): the same chain with every fp32 product split into fp16 parts on the f16 MFMA
// pipe - v_mfma_f32_16x16x32_f16 has the SAME accumulator layout, and its operand lane (j, g) holds 8 k-slots whose
// assignment to actual k is free as long as A and B agree: slot e of lane group g in 32-group T is k = 32T + 16(e>>2) +
// 4g + (e&3), i.e. exactly the accumulator registers (nt = 2T, 2T+1; r = 0..3) this lane already owns, so the
// register-resident chaining carries over.  x = hi + lo per operand (activations: split in registers after an exact
// per-row power-of-two scaling; weights: pre-split image, one global scale), acc += w_lo x_hi + w_hi x_lo + w_hi x_hi:
// 3 MFMAs of 16 cycles per (n-tile, 32 k) instead of 8 of 32, with the error of the f32 MFMA.  The LDS slice is the
// image's (pass, T) block copied linearly: [nt][part][lane] x 16 B, conflict-free ds_read_b128 without a swizzle.
#pragma once
#include <stdlib.h>
#include "gfv_common.h"
#include "gfv_prof.h"
#include "gfv_split.h"
#include "../../include/gfv.h"

#ifdef GFV_TIMING
// phase timing build (scratch experiments only): per wave 10 int64 counters written through ln_partial
#define TS_DECL long long ts_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long t_prev_ = clock64(); const long long t_start_ = t_prev_;
#define TS(k) do { const long long now_ = clock64(); ts_[k] += now_ - t_prev_; t_prev_ = now_; } while (0)
#define TS_WAIT() __builtin_amdgcn_s_waitcnt(0)
#else
#define TS_DECL
#define TS(k)
#define TS_WAIT()
#endif

namespace {

constexpr int WK = 32;
constexpr int WS_FLOATS = 128 * WK;  // one weight slice
constexpr int PAR_GAMMA = 640, PAR_BETA = 768, PAR_FLOATS = 896;

struct WBlk {
  const float* w;  // &W[128*pass][koff]
  int ldw;
  int nsl;    // 32-wide slices in the block
  int nrows;  // valid output rows of the block (128, or the 64-wide tail of a 192-wide last layer)
  int kvalid; // valid k columns of the block (RAG instantiation: first-layer K of 12 / 15 / 3)
  int rag;    // block needs element-wise weight loads (k not a multiple of 32, or rows not 16-B aligned)
};

template <bool H>
__device__ __forceinline__ WBlk w_block(const gfv_rowtile_args_t& A, int layer, int pass, int chunk) {
  const gfv_layer_t& L = A.layer[layer];
  int koff = 0, width = 128;
  if (layer == 0) {
    for (int i = 0; i < chunk; ++i) koff += A.seg[i].width;
    width = A.seg[chunk].width;
  }
  WBlk b;
  if (H) {  // image slices of 4096 floats (16 KB), [pass][T]; the block starts at T = koff / 32
    const int nT = (L.K + 31) >> 5;
    b.w = reinterpret_cast<const float*>(L.Wh) + ((size_t)pass * nT + (koff >> 5)) * 4096;
    b.ldw = 0;
    b.nsl = (width + WK - 1) / WK;
    b.nrows = min(128, L.N - 128 * pass);
    b.kvalid = width;
    b.rag = 0;
    return b;
  }
  b.ldw = L.ldw ? L.ldw : L.K;
  b.w = L.W + (size_t)(128 * pass) * b.ldw + koff;
  b.nsl = (width + WK - 1) / WK;
  b.nrows = min(128, L.N - 128 * pass);
  b.kvalid = width;
  b.rag = ((width % WK != 0) || (b.ldw & 3) || ((reinterpret_cast<size_t>(b.w) & 15) != 0)) ? 1 : 0;
  return b;
}

// Weight prefetch registers are four NAMED native vectors and the load is unconditional (pointer picked with a
// ternary): an array of HIP float4 filled under a branch is parked in scratch by the compiler with a vmcnt wait right
// behind the load, which serialises the whole weight stream on the L2 latency.
struct WRegs {
  floatx4 a, b, c, d;
};
__device__ __forceinline__ WRegs w_load(const float* w, int ldw, int nrows, int wrow, int wc) {
  WRegs r;  // rows past the block's last valid row re-read that row (their products are never stored)
  const float* p0 = w + wc;
  const int last = nrows - 1;
  r.a = *reinterpret_cast<const floatx4*>(p0 + (size_t)min(wrow, last) * ldw);
  r.b = *reinterpret_cast<const floatx4*>(p0 + (size_t)min(wrow + 32, last) * ldw);
  r.c = *reinterpret_cast<const floatx4*>(p0 + (size_t)min(wrow + 64, last) * ldw);
  r.d = *reinterpret_cast<const floatx4*>(p0 + (size_t)min(wrow + 96, last) * ldw);
  return r;
}
// element-wise form for ragged blocks: k >= kvalid reads as zero, no alignment assumed
__device__ __forceinline__ WRegs w_load_ragged(const float* w, int ldw, int nrows, int k0, int kvalid, int wrow, int wc) {
  WRegs r;
  const int last = nrows - 1;
  floatx4 v[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float* rp = w + (size_t)min(wrow + 32 * p, last) * ldw;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + wc + e;
      v[p][e] = k < kvalid ? rp[k] : 0.f;
    }
  }
  r.a = v[0]; r.b = v[1]; r.c = v[2]; r.d = v[3];
  return r;
}
__device__ __forceinline__ void w_store(float* Wb, int tid, const WRegs& r) {
  *reinterpret_cast<floatx4*>(Wb + 4 * tid) = r.a;
  *reinterpret_cast<floatx4*>(Wb + 4 * tid + 1024) = r.b;
  *reinterpret_cast<floatx4*>(Wb + 4 * tid + 2048) = r.c;
  *reinterpret_cast<floatx4*>(Wb + 4 * tid + 3072) = r.d;
}

// (cross-lane helpers: gfv_common.h - DPP within a 16-lane row, v_permlane16/32_swap across rows, no LDS crossbar)
// sum over the 4 lanes (g = 0..3) that share a row
__device__ __forceinline__ float row_sum(float v) {
  float a, b;
  gfv_lane_xor16(v, a, b);
  v = a + b;
  gfv_lane_xor32(v, a, b);
  return a + b;
}
__device__ __forceinline__ float row_max4(float v) {
  float a, b;
  gfv_lane_xor16(v, a, b);
  v = fmaxf(a, b);
  gfv_lane_xor32(v, a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
// rows saved for the backward (pre-activations, the pre-LayerNorm values): written once, read a millisecond later
// (non-temporal stores were measured in round 3: 4.07 - 4.11 against 4.06 ms, no gain)
__device__ __forceinline__ void st4_save(float* p, const float (&v)[4]) {
  st4(p, v);
}

// (Round 3 also had a 128-byte-run form of every row access here - lane pairs exchanging half rows by DPP, `GFV_RUN128`:
// 3.99 against 3.89 ms per step, the extra moves cost what the wider runs gave, profiles/r03_ab_run128.txt - removed in round 5.)

// LayerNorm statistics of one row spread over 4 lanes x 8 x 4 registers
// LnW: the LayerNorm width.  A model of hidden_size h < 128 runs zero-padded to 128 columns (gfv_set_hidden_size): the
// statistics are those of the h real columns - the padded zeros add nothing to the sum, and (0 - mean)^2 each to the sum of
// squared deviations, which is taken out again (npad = 128 - h).  h = 128: inv_n = 1/128, npad = 0 - the same arithmetic as
// before (q - 0 * mean * mean).
struct LnW {
  float inv_n, npad;
};
__device__ __forceinline__ LnW ln_width(int cols) {
  const int n = (cols > 0 && cols < 128) ? cols : 128;
  return LnW{1.0f / (float)n, (float)(128 - n)};
}
__device__ __forceinline__ void ln_stats(const float (&v)[8][4], float& mean, float& rstd, const LnW w) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) s += (v[t][0] + v[t][1]) + (v[t][2] + v[t][3]);
  mean = row_sum(s) * w.inv_n;
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float d0 = v[t][0] - mean, d1 = v[t][1] - mean, d2 = v[t][2] - mean, d3 = v[t][3] - mean;
    q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  rstd = rsqrtf((row_sum(q) - w.npad * (mean * mean)) * w.inv_n + 1e-5f);  // nn.LayerNorm eps (EPD.py:32)
}

// v <- LayerNorm(v) * gamma + beta (gamma / beta at columns 16t + 4g + r)
__device__ __forceinline__ void ln_apply(float (&v)[8][4], const float* gamma, const float* beta, int g, const LnW w,
                                         float* stats_row = nullptr) {
  float mean, rstd;
  ln_stats(v, mean, rstd, w);
  // (fin_stats: the row's (mean, 1 / std) for a backward launch that does not recompute them, include/gfv.h)
  if (stats_row && g == 0) *reinterpret_cast<float2*>(stats_row) = make_float2(mean, rstd);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float4 ga = ld4(gamma + 16 * t + 4 * g), be = ld4(beta + 16 * t + 4 * g);
    v[t][0] = (v[t][0] - mean) * rstd * ga.x + be.x;
    v[t][1] = (v[t][1] - mean) * rstd * ga.y + be.y;
    v[t][2] = (v[t][2] - mean) * rstd * ga.z + be.z;
    v[t][3] = (v[t][3] - mean) * rstd * ga.w + be.w;
  }
}

// LayerNorm backward of one row: y = LN input (pre-normalisation), go = grad wrt LN output (in v, replaced by the
// grad wrt the LN input); accumulates this lane's 32 columns of dgamma / dbeta.
template <bool FIRST>   // FIRST: (dgam, dbet) are assigned, not added to (no zero-initialised accumulators for the first tile)
__device__ __forceinline__ void ln_bwd(float (&v)[8][4], const float (&y)[8][4], const float* gamma, int g,
                                       float (&dgam)[8][4], float (&dbet)[8][4], const LnW w) {
  float mean, rstd;
  ln_stats(y, mean, rstd, w);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float4 ga = ld4(gamma + 16 * t + 4 * g);
    const float gv[4] = {ga.x, ga.y, ga.z, ga.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float xh = (y[t][r] - mean) * rstd;
      if (FIRST) {
        dgam[t][r] = v[t][r] * xh;
        dbet[t][r] = v[t][r];
      } else {
        dgam[t][r] += v[t][r] * xh;
        dbet[t][r] += v[t][r];
      }
      v[t][r] *= gv[r];  // gg
      s1 += v[t][r];
      s2 += v[t][r] * xh;
    }
  }
  const float m1 = row_sum(s1) * w.inv_n, m2 = row_sum(s2) * w.inv_n;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[t][r] = rstd * (v[t][r] - m1 - ((y[t][r] - mean) * rstd) * m2);
}

// sum over the 16 lanes of a DPP row (= the 16 rows of the wave's tile at fixed g), total in lane 15 of the row: four
// v_add_f32 with a row_shr operand (zeros shifted in) instead of four ds_bpermute round trips through the LDS crossbar
__device__ __forceinline__ float row16_sum_to_last(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, true));  // row_shr:1
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x112, 0xf, 0xf, true));  // row_shr:2
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xf, 0xf, true));  // row_shr:4
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xf, 0xf, true));  // row_shr:8
  return x;
}

// fold the lane-private (dgamma, dbeta) sums over the 16 rows of the wave and park them in LDS: red[wave][2][128]
__device__ __forceinline__ void ln_park(float (&dgam)[8][4], float (&dbet)[8][4], float* red, int wave, int li, int g) {
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dgam[t][r] = row16_sum_to_last(dgam[t][r]);
      dbet[t][r] = row16_sum_to_last(dbet[t][r]);
    }
  if (li == 15) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      st4(red + (wave * 2 + 0) * 128 + 16 * t + 4 * g, dgam[t]);
      st4(red + (wave * 2 + 1) * 128 + 16 * t + 4 * g, dbet[t]);
    }
  }
}

// LayerNorm backward of ONE row per lane (T = 1) with the (dgamma, dbeta) fold in the same pass: the lane's 32 products
// dy * xhat and its 32 dy go through the row fold and into LDS four at a time - holding all 64 through the second half of
// ln_bwd and handing them to ln_park afterwards keeps 64 more registers alive in the most register-hungry phase of the
// LayerNorm-backward instantiation.
__device__ __forceinline__ void ln_bwd_park(float (&v)[8][4], const float (&y)[8][4], const float* gamma, int g, float* red,
                                            int wave, int li, const LnW w) {
  float mean, rstd;
  ln_stats(y, mean, rstd, w);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float4 ga = ld4(gamma + 16 * t + 4 * g);
    const float gv[4] = {ga.x, ga.y, ga.z, ga.w};
    float dg[4], db[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float xh = (y[t][r] - mean) * rstd;
      dg[r] = row16_sum_to_last(v[t][r] * xh);
      db[r] = row16_sum_to_last(v[t][r]);
      v[t][r] *= gv[r];  // gg
      s1 += v[t][r];
      s2 += v[t][r] * xh;
    }
    if (li == 15) {
      st4(red + (wave * 2 + 0) * 128 + 16 * t + 4 * g, dg);
      st4(red + (wave * 2 + 1) * 128 + 16 * t + 4 * g, db);
    }
  }
  const float m1 = row_sum(s1) * w.inv_n, m2 = row_sum(s2) * w.inv_n;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[t][r] = rstd * (v[t][r] - m1 - ((y[t][r] - mean) * rstd) * m2);
}

// one 32-wide k slice: acc[tt][nt] += W[16nt + i][k] * act[tt][k], k = 16t + 4g + s for t in {2sl, 2sl+1}
template <int T>
__device__ __forceinline__ void mma_slice(floatx4 (&acc)[T][8], const float (&act)[T][8][4], int t0, const float* Wb, int off0) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float4 w[8];
    const float* wp = Wb + (h ? (off0 ^ 16) : off0);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) w[nt] = *reinterpret_cast<const float4*>(wp + 512 * nt);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int tt = 0; tt < T; ++tt) acc[tt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt].x, act[tt][t0 + h][0], acc[tt][nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int tt = 0; tt < T; ++tt) acc[tt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt].y, act[tt][t0 + h][1], acc[tt][nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int tt = 0; tt < T; ++tt) acc[tt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt].z, act[tt][t0 + h][2], acc[tt][nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int tt = 0; tt < T; ++tt) acc[tt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt].w, act[tt][t0 + h][3], acc[tt][nt], 0, 0, 0);
  }
}

// H: a slice is 16 KB contiguous in the image, copied linearly by the NW waves of the workgroup (16 / NW chunks of 16 B
// per thread: 4 for the 4-wave workgroup, 2 / 1 for the 8- / 16-wave ones that share one weight stream over more rows)
template <int NW>
__device__ __forceinline__ WRegs w_load_h(const float* slice, int tid) {
  WRegs r;
  r.a = *reinterpret_cast<const floatx4*>(slice + 4 * tid);
  if (NW <= 8) r.b = *reinterpret_cast<const floatx4*>(slice + 4 * tid + 256 * NW);
  if (NW <= 4) {
    r.c = *reinterpret_cast<const floatx4*>(slice + 4 * tid + 2048);
    r.d = *reinterpret_cast<const floatx4*>(slice + 4 * tid + 3072);
  }
  return r;
}
template <int NW>
__device__ __forceinline__ void w_store_n(float* Wb, int tid, const WRegs& r) {
  *reinterpret_cast<floatx4*>(Wb + 4 * tid) = r.a;
  if (NW <= 8) *reinterpret_cast<floatx4*>(Wb + 4 * tid + 256 * NW) = r.b;
  if (NW <= 4) {
    *reinterpret_cast<floatx4*>(Wb + 4 * tid + 2048) = r.c;
    *reinterpret_cast<floatx4*>(Wb + 4 * tid + 3072) = r.d;
  }
}

// H: one 32-wide k group on the f16 pipe; LDS slice image [nt 8][part 2][lane 64] x 16 B; n-tiles >= 4 are skipped for
// the 64-wide tail chunk of a 192-wide last layer (nth = number of 4-tile halves with valid rows)
// lowp (gfv_set_f16split(2)): the hi x hi term only.  BF (gfv_set_f16split(3); its own instantiations, tchain_bf16.hip): the
// hi x hi term on bf16 operands (v_mfma_f32_16x16x32_bf16)
template <bool BF>
__device__ __forceinline__ void mma_slice_h(floatx4 (&acc)[8], const gfv_f16x8& xh, const gfv_f16x8& xl, const float* Wb,
                                            int lane, int nth, bool lowp) {
  const gfv_f16x8* wp = reinterpret_cast<const gfv_f16x8*>(Wb) + lane;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h < nth) {
      gfv_f16x8 w0[4], w1[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) w0[n] = wp[((4 * h + n) * 2 + 0) * 64];
      if constexpr (BF) {
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[4 * h + n] = gfv_mma_hh<true>(w0[n], xh, acc[4 * h + n]);
        continue;
      }
      if (!lowp) {   // (uniform branch)
#pragma unroll
        for (int n = 0; n < 4; ++n) w1[n] = wp[((4 * h + n) * 2 + 1) * 64];
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[4 * h + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[n], xh, acc[4 * h + n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[4 * h + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[n], xl, acc[4 * h + n], 0, 0, 0);
      }
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[4 * h + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[n], xh, acc[4 * h + n], 0, 0, 0);
    }
  }
}

// H: power-of-two scale of one row (max over the lane's 32 values and the 4 lanes of the row)
// (v_max3_f32 with |.| source modifiers, two independent chains: 16 instructions for the 32 values.  `fmaxf(m, fabsf(x))`
// costs hipcc 2.6 per value - IEEE mode makes it canonicalise every operand with a v_max_f32 x, x first.)
__device__ __forceinline__ float max3_abs(float m, float a, float b) {
  float r;
  asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float row_scale(const float (&v)[8][4]) {
  float m0 = 0.f, m1 = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    m0 = max3_abs(m0, v[t][0], v[t][1]);
    m1 = max3_abs(m1, v[t][2], v[t][3]);
  }
  return gfv_pow2_scale(row_max4(max3_abs(0.f, m0, m1)));
}
// H: the scale of a group of 16 rows (= this wave's rows) is the smallest of its rows' scales; lane 0 leaves it for the
// weight-gradient kernel (gfv_rowtile_args_t.gscale)
__device__ __forceinline__ void group_scale_out(float* dst, float s, int lane) {
  s = gfv_row16_min(s);
  if (lane == 0) *dst = s;
}
// H: fp32 activations -> B-operand fragments of the four 32-groups: slots e = 0..3 <- act[2T][.], 4..7 <- act[2T+1][.]
template <bool BF>   // BF: high parts in bf16, no low parts
__device__ __forceinline__ void to_halves(const float (&v)[8][4], float sc, gfv_f16x8 (&xh)[4], gfv_f16x8 (&xl)[4]) {
#pragma unroll
  for (int T32 = 0; T32 < 4; ++T32) {
    const float e[8] = {v[2 * T32][0] * sc,     v[2 * T32][1] * sc,     v[2 * T32][2] * sc,     v[2 * T32][3] * sc,
                        v[2 * T32 + 1][0] * sc, v[2 * T32 + 1][1] * sc, v[2 * T32 + 1][2] * sc, v[2 * T32 + 1][3] * sc};
    gfv_uint4 hi, lo;
    gfv_split8_t<BF>(e, hi, lo);
    xh[T32] = __builtin_bit_cast(gfv_f16x8, hi);
    xl[T32] = __builtin_bit_cast(gfv_f16x8, lo);
  }
}

// ---- input segment -> activation registers (gather / concat piece / prologue element ops) -------------------------
template <int T, int LNM, bool RAG, bool CSR>
__device__ __forceinline__ void load_segment(const gfv_rowtile_args_t& A, int si, int rowbase, int g, const float* gam,
                                             const float* bet, float (&act)[T][8][4], float (&dgam)[8][4],
                                             float (&dbet)[8][4], int in_op, float* red, int wave, int li, const LnW lnw) {
  const gfv_seg_t& s = A.seg[si];
  const int nt_valid = s.width >> 4;
  const bool first = (si == 0);
#pragma unroll
  for (int tt = 0; tt < T; ++tt) {
    const int m = rowbase + 16 * tt;
    const bool live = m < A.M;
    const int mc = live ? m : A.M - 1;
    const bool csr = CSR && s.csr_rowptr != nullptr;
    const size_t srow = (s.idx && !csr) ? (size_t)s.idx[mc] : (size_t)mc;
    const float* rp = s.ptr + srow * (size_t)s.ld + 4 * g;
    if (csr) {
      // the segment row is a segmented sum: scale[m] * sum_{k in [rowptr[m], rowptr[m+1])} src[idx[k], :] - the neighbour
      // aggregation of the GnBlock (blocks.py:25-51,84-99) and the per-side scatter of the factored EdgeBlock's adjoint,
      // formed right here instead of by a launch of its own that writes the sums out and a chain launch that reads them
      // back.  Two neighbour rows in flight per lane, entries added in CSR order (fixed: deterministic).
      const int beg = s.csr_rowptr[mc], end = live ? s.csr_rowptr[mc + 1] : beg;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) act[tt][t][r] = 0.f;
      // (the column indices run one iteration ahead of the rows they address: one dependent round trip per iteration, not two)
      int cn0 = beg < end ? s.idx[beg] : 0, cn1 = beg + 1 < end ? s.idx[beg + 1] : cn0;
      for (int k = beg; k < end; k += 2) {
        const bool two = (k + 1 < end);
        const int c0 = cn0, c1 = cn1;
        if (k + 2 < end) {
          cn0 = s.idx[k + 2];
          cn1 = s.idx[k + 3 < end ? k + 3 : k + 2];
        }
        const float* p0 = s.ptr + (size_t)c0 * (size_t)s.ld + 4 * g;
        const float* p1 = s.ptr + (size_t)c1 * (size_t)s.ld + 4 * g;
        // (unconditional loads - a register array filled under a branch is parked in scratch by the compiler; 16-column
        // groups past the segment's width re-read group 0 and are not added)
        float4 v0[8], v1[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int tv = t < nt_valid ? t : 0;
          v0[t] = ld4(p0 + 16 * tv);
          v1[t] = ld4(p1 + 16 * tv);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (t < nt_valid) {
            act[tt][t][0] += v0[t].x; act[tt][t][1] += v0[t].y; act[tt][t][2] += v0[t].z; act[tt][t][3] += v0[t].w;
            if (two) { act[tt][t][0] += v1[t].x; act[tt][t][1] += v1[t].y; act[tt][t][2] += v1[t].z; act[tt][t][3] += v1[t].w; }
          }
      }
      if (s.csr_scale) {
        const float sc = s.csr_scale[mc];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) act[tt][t][r] *= sc;
      }
    } else if (RAG && ((s.width & 31) || (s.ld & 3))) {
      // ragged segment (encoder inputs of 12 / 15 columns, the decoder's 3-wide gradient): element-wise, zero padded
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = 16 * t + 4 * g + r;
          float v = 0.f;
          if (col < s.width) {
            v = rp[16 * t + r];
            if (first && A.in_add) v += A.in_add[srow * (size_t)s.ld + col];
          }
          act[tt][t][r] = v;
        }
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < nt_valid) v = ld4(rp + 16 * t);
        act[tt][t][0] = v.x; act[tt][t][1] = v.y; act[tt][t][2] = v.z; act[tt][t][3] = v.w;
      }
    }
    if (first && A.in_add && !(RAG && ((s.width & 31) || (s.ld & 3)))) {
      const float* ap = A.in_add + srow * (size_t)s.ld + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (t < nt_valid) {
          const float4 v = ld4(ap + 16 * t);
          act[tt][t][0] += v.x; act[tt][t][1] += v.y; act[tt][t][2] += v.z; act[tt][t][3] += v.w;
        }
    }
    if (first && A.gadd) {
      const float* gs = A.gadd + (size_t)A.gadd_s[mc] * 64 + 4 * g;
      const float* gr = A.gadd + (size_t)A.gadd_r[mc] * 64 + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float4 v = ld4((t < 4 ? gs : gr) + 16 * (t & 3));
        act[tt][t][0] += v.x; act[tt][t][1] += v.y; act[tt][t][2] += v.z; act[tt][t][3] += v.w;
      }
    }
    if (in_op == GFV_IN_GELU) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) act[tt][t][r] = gfv_gelu(act[tt][t][r]);
    } else if (in_op == GFV_IN_LN) {
      ln_apply(act[tt], gam, bet, g, lnw);
    } else if (LNM == 1 && in_op == GFV_IN_LNBWD) {
      float y[8][4];
      const float* yp = A.in_aux + (size_t)mc * 128 + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float4 v = ld4(yp + 16 * t);
        y[t][0] = v.x; y[t][1] = v.y; y[t][2] = v.z; y[t][3] = v.w;
      }
      // rows past M (clamped re-reads of row M - 1) must not reach the (dgamma, dbeta) sums: their incoming gradient is
      // multiplied by 0 - everything ln_bwd derives from it is then 0 as well.  (A multiplication, not `if (!live) act = 0`:
      // hipcc turns the conditional assignment of a register array into a copy of all 32 registers plus 32 more moves
      // under the exec mask, on every tile - 9 % of this instantiation's VALU instructions together with the zeroing
      // that used to follow every segment load.)
      const float livef = live ? 1.0f : 0.0f;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) act[tt][t][r] *= livef;
      if (T == 1) ln_bwd_park(act[tt], y, gam, g, red, wave, li, lnw);   // (the fold lands in LDS right here)
      else if (tt == 0) ln_bwd<true>(act[tt], y, gam, g, dgam, dbet, lnw);
      else ln_bwd<false>(act[tt], y, gam, g, dgam, dbet, lnw);
    }
    // (rows past M keep the values of row M - 1 from here on: nothing of theirs is stored, and no other sum runs over rows)
    if (first && A.in_save && live) {
      float* sp = A.in_save + (size_t)m * 128 + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) st4(sp + 16 * t, act[tt][t]);
    }
    if (CSR && s.save && live) {   // the assembled rows of THIS segment (the weight-gradient launch reads them)
      float* sp = s.save + (size_t)m * 128 + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) st4(sp + 16 * t, act[tt][t]);
    }
  }
}

// LNM: 0 = no LayerNorm backward, 1 = GFV_IN_LNBWD prologue, 2 = GFV_FIN_LNBWD epilogue (the (dgamma, dbeta)
// accumulators exist only in those instantiations)
// RAG: also takes ragged shapes (first-layer K / segment widths that are not multiples of 32, unaligned rows, a last
// layer narrower than 64): element-wise loads / stores on those pieces only
// H: products on the f16 MFMA pipe from the layers' split-fp16 weight images (T = 1 only)
// NW: waves per workgroup (4; 8 exists in the H form for experiments): a workgroup owns 16 NW rows and ONE weight stream.
// Per 64-row tile the three layers' images are 192 KB from L2 next to 256 KB of activations; an 8-wave workgroup (one per
// CU at 2 waves / SIMD, the occupancy of two 4-wave ones) moves half the weight bytes and runs half the barrier rounds
// per row - and was slower (see the launcher): 8 waves marching in lockstep through the slice barriers hide less latency
// than two independent groups of 4.  The LayerNorm partials stay per 64-row tile (same sums, same order) either way.
// IOP: 0 = the element ops are read from the arguments at run time; 1 / 2 = every layer before the last has
// GFV_OP_BIAS_GELU / GFV_OP_MUL_DGELU and the prologue op is none (LNM = 1: the LayerNorm backward) - what all but a
// handful of the model's launches are.  With the ops known at compile time the epilogues have no branches to merge:
// hipcc joins the arms of a run-time `switch (op)` over a 32-register activation array with 100 - 150 register moves per
// layer (a tenth of the kernel's VALU instructions).
template <int T, int LNM, bool RAG, bool H, int NW = 4, bool CSR = false, int IOP = 0, bool BF = false>
#ifndef GFV_CHAIN_WAVES   // waves per SIMD the register allocation aims at (a translation unit may set its own)
#define GFV_CHAIN_WAVES(H, LNM, RAG) 2
#endif
__global__ __launch_bounds__(64 * NW, GFV_CHAIN_WAVES(H, LNM, RAG)) void tchain_kernel(const gfv_rowtile_args_t A) {
  static_assert(!CSR || (LNM == 0 && !RAG && T == 1), "segmented-sum segments exist in the plain instantiation");
  static_assert(!H || T == 1, "the f16 form is instantiated for 16 rows per wave");
  static_assert(NW == 4 || (H && T == 1 && NW == 8), "the wide workgroup exists in the f16 form only");
  constexpr int NT = 64 * NW;
  __shared__ __attribute__((aligned(16))) float lds[2 * WS_FLOATS + 256 * NW + PAR_FLOATS];
  float* red = lds + 2 * WS_FLOATS;
  float* par = red + 256 * NW;  // bias of layer l at 128 l (N_l floats), LayerNorm gamma / beta: read from LDS in the epilogues
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  // XCD-aware tile order (gfv_common.h) where the prologue gathers neighbour rows; measured (profiles/tools/ab.sh, one box):
  // segmented-sum instantiation 0.773 -> 0.757 ms / step, seg_gather_sum 0.232 -> 0.227; the streaming instantiations do
  // not gain (the LayerNorm-backward one loses 1 %) and keep the plain order
  const int tile = CSR ? gfv_xcd_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  if (tile * (16 * NW * T) >= A.M) return;
  const int rowbase = tile * (16 * NW * T) + wave * (16 * T) + li;
  const int rowgroup = tile * NW + wave;   // H (T = 1): index of this wave's 16 rows
  // weight staging: thread -> row (tid>>3) + 32p, LDS slot tid&7, source chunk slot ^ ((row>>1)&7)
  const int wrow = tid >> 3;
  const int wc = 4 * ((tid & 7) ^ ((tid >> 4) & 7));
  // fragment reads: row 16nt + li, chunk (4h + g) ^ ((li>>1)&7)
  const int off0 = li * 32 + 4 * (g ^ ((li >> 1) & 7));

  float act[T][8][4];
  floatx4 acc[T][8];
  constexpr bool lnb_in = (LNM == 1), lnb_fin = (LNM == 2);
  gfv_f16x8 xh[4], xl[4];                                    // H: the activations as (hi, lo) B fragments
  float sx = 1.f;                                            // H: this row's current power-of-two scale
  const float ws = H ? gfv_pow2_scale(*A.wmax) : 1.f;        // H: the images' weight scale
  const bool lowp = H && A.product_form != 0;
  const LnW lnw = ln_width(A.hidden);                          // LayerNorm width (0 / 128: every column)                       // H: reduced-precision form (hi x hi products only)

  // gather rows of the factored first-layer addend: index round trip issued first thing, used in the first epilogue
  const float* pad_s[T];
  const float* pad_r[T];
  if (A.padd) {
#pragma unroll
    for (int tt = 0; tt < T; ++tt) {
      const int mc = min(rowbase + 16 * tt, A.M - 1);
      pad_s[tt] = A.padd + (size_t)A.padd_s[mc] * A.padd_ld + 4 * g;
      pad_r[tt] = A.padd + (size_t)A.padd_r[mc] * A.padd_ld + 128 + 4 * g;
    }
  }
  // small parameter vectors -> LDS: one round trip at kernel start (overlapping the first weight slice) instead of a
  // synchronous global load in every epilogue; visible after the first barrier
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    if (l < A.nlayers) {
      const float* bp = A.layer[l].bias;
      const float* b2 = A.layer[l].bias2;   // columns >= 128 of a row-stacked last layer
      const int nl = A.layer[l].N;
      for (int c = tid; c < nl; c += NT) par[128 * l + c] = (b2 && c >= 128) ? b2[c - 128] : (bp ? bp[c] : 0.f);
    }
  }
  {
    const bool in_ln = (A.in_op == GFV_IN_LN || A.in_op == GFV_IN_LNBWD);
    const float* gp = in_ln ? A.in_gamma : A.fin_gamma;
    const float* bp = (A.in_op == GFV_IN_LN) ? A.in_beta : A.fin_beta;
    if (in_ln || A.fin_op != GFV_FIN_PLAIN) {
      if (tid < 128) par[PAR_GAMMA + tid] = gp[tid];
      else if (tid < 256 && (A.in_op == GFV_IN_LN || A.fin_op == GFV_FIN_LN)) par[PAR_BETA + tid - 128] = bp[tid - 128];
    }
  }
  int wbuf = 0;
  TS_DECL
  WBlk cur = w_block<H>(A, 0, 0, 0);
  // weight pipeline: slice j+1 is loaded to registers while slice j feeds the MFMAs, then parked in the other LDS
  // buffer (prefetch distance 2 with a second register set was measured: no gain, +33 VGPRs)
  WRegs wr0 = H ? w_load_h<NW>(cur.w, tid)
                 : ((RAG && cur.rag) ? w_load_ragged(cur.w, cur.ldw, cur.nrows, 0, cur.kvalid, wrow, wc)
                                     : w_load(cur.w, cur.ldw, cur.nrows, wrow, wc));
  w_store_n<NW>(lds, tid, wr0);
  __syncthreads();
  TS(0);

  for (int layer = 0; layer < A.nlayers; ++layer) {
    const gfv_layer_t& L = A.layer[layer];
    const bool last = (layer == A.nlayers - 1);
    const int npass = last ? (L.N + 127) / 128 : 1;
    const int nchunk = (layer == 0) ? A.nseg : 1;
    for (int pass = 0; pass < npass; ++pass) {
#pragma unroll
      for (int tt = 0; tt < T; ++tt)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) acc[tt][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
      for (int chunk = 0; chunk < nchunk; ++chunk) {
        // next block in the flat (layer, pass, chunk) order
        int nl_ = layer, np_ = pass, nc_ = chunk + 1;
        bool have_next = true;
        if (nc_ >= nchunk) {
          nc_ = 0;
          np_ = pass + 1;
          if (np_ >= npass) {
            np_ = 0;
            nl_ = layer + 1;
            if (nl_ >= A.nlayers) have_next = false;
          }
        }
        WBlk nxt = cur;
        if (have_next) nxt = w_block<H>(A, nl_, np_, nc_);
        if (layer == 0 && (nchunk > 1 || pass == 0)) {
          // (dgamma, dbeta) accumulators live only here: parked in LDS before the MFMA loop needs the registers
          float dgam[8][4], dbet[8][4];
          if (lnb_in) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) dgam[t][r] = dbet[t][r] = 0.f;
          }
          load_segment<T, LNM, RAG, CSR>(A, chunk, rowbase, g, par + PAR_GAMMA, par + PAR_BETA, act, dgam, dbet,
                                         IOP != 0 ? (LNM == 1 ? (int)GFV_IN_LNBWD : (int)GFV_IN_NONE) : A.in_op, red, wave, li, lnw);
          if (lnb_in && T > 1) ln_park(dgam, dbet, red, wave, li, g);
          if (H) {
            // every segment gets its own row scale; the accumulator follows (exact: powers of two)
            float sn = row_scale(act[0]);
            if (A.gscale && chunk == 0 && pass == 0) group_scale_out(A.gscale + rowgroup, sn, lane);
            if (chunk > 0) {
              // a segment 2^40 below what the accumulator already holds cannot be resolved next to it anyway: its
              // scale is capped so that the ratio stays finite
              sn = fminf(sn, sx * 1.099511627776e12f);
              const float ratio = sn / sx;
#pragma unroll
              for (int nt = 0; nt < 8; ++nt) acc[0][nt] *= ratio;
            }
            sx = sn;
            to_halves<BF>(act[0], sx, xh, xl);
          }
          TS_WAIT();
          TS(1);
        }
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
          if (sl < cur.nsl) {
            // prefetch the next slice (after the very last one: a harmless reload, nxt == cur there)
            const bool more = (sl + 1 < cur.nsl);
            const float* wsrc = more ? cur.w + WK * (sl + 1) : nxt.w;
            const int wld = more ? cur.ldw : nxt.ldw;
            const int wnr = more ? cur.nrows : nxt.nrows;
            if (H) {
              wr0 = w_load_h<NW>(more ? cur.w + 4096 * (sl + 1) : nxt.w, tid);
            } else if (RAG && (more ? cur.rag : nxt.rag)) {
              wr0 = w_load_ragged(more ? cur.w : nxt.w, wld, wnr, more ? WK * (sl + 1) : 0, more ? cur.kvalid : nxt.kvalid,
                                  wrow, wc);
            } else {
              wr0 = w_load(wsrc, wld, wnr, wrow, wc);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ABOVE the MFMAs (the scheduler sinks it otherwise)
            TS(2);
            if (H) mma_slice_h<BF>(acc[0], xh[sl], xl[sl], lds + wbuf * WS_FLOATS, lane, (cur.nrows + 63) >> 6, lowp);
            else mma_slice<T>(acc, act, 2 * sl, lds + wbuf * WS_FLOATS, off0);
            TS(3);
            __builtin_amdgcn_sched_barrier(0);
            w_store_n<NW>(lds + (wbuf ^ 1) * WS_FLOATS, tid, wr0);
            TS(4);
            __syncthreads();
            TS(5);
            wbuf ^= 1;
          }
        }
        cur = nxt;
      }

      const float invx = H ? 1.0f / sx : 1.0f, invw = H ? 1.0f / ws : 1.0f;   // H: undo the operand scales (exact)
      if (!last) {
        // ---- intermediate epilogue: accumulators -> next layer's activations, in registers ----
        const int lop = IOP != 0 ? IOP : L.op;
#pragma unroll
        for (int tt = 0; tt < T; ++tt) {
          const int m = rowbase + 16 * tt;
          const bool live = m < A.M;
          const size_t mrow = (size_t)(live ? m : A.M - 1) * 128 + 4 * g;
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) {
            float v[4] = {acc[tt][nt][0], acc[tt][nt][1], acc[tt][nt][2], acc[tt][nt][3]};
            if (H) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = (v[r] * invx) * invw;
            }
            if (lop == GFV_OP_MUL_DGELU) {
              const float4 z = ld4(L.aux + mrow + 16 * nt);
              v[0] *= gfv_dgelu(z.x); v[1] *= gfv_dgelu(z.y); v[2] *= gfv_dgelu(z.z); v[3] *= gfv_dgelu(z.w);
              if (L.save && live) st4_save(L.save + mrow + 16 * nt, v);
            } else {
              {
                const float4 b = ld4(par + 128 * layer + 16 * nt + 4 * g);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
              }
              if (layer == 0 && A.padd) {
                // first layer factored through the nodes: + (W1a x)[s] + (W1b x)[r]
                const float4 pa = ld4(pad_s[tt] + 16 * nt), pb = ld4(pad_r[tt] + 16 * nt);
                v[0] += pa.x + pb.x; v[1] += pa.y + pb.y; v[2] += pa.z + pb.z; v[3] += pa.w + pb.w;
              }
              if (lop == GFV_OP_BIAS_GELU) {
                if (L.save && live) st4_save(L.save + mrow + 16 * nt, v);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gfv_gelu(v[r]);
              }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) act[tt][nt][r] = v[r];
          }
        }
        if (H) {
          sx = row_scale(act[0]);
          if (A.gscale && lop == GFV_OP_MUL_DGELU) group_scale_out(A.gscale + (size_t)(layer + 1) * A.gscale_ld + rowgroup, sx, lane);
          to_halves<BF>(act[0], sx, xh, xl);
        }
        TS_WAIT();
        TS(6);
      } else {
        // ---- final epilogue for output chunk `pass` ----
        float* out = pass == 0 ? A.out[0] : (pass == 1 ? A.out[1] : A.out[2]);
        const float* res = pass == 0 ? A.res[0] : (pass == 1 ? A.res[1] : A.res[2]);
        const int old = pass == 0 ? A.out_ld[0] : (pass == 1 ? A.out_ld[1] : A.out_ld[2]);
        const int rld = pass == 0 ? A.res_ld[0] : (pass == 1 ? A.res_ld[1] : A.res_ld[2]);
        const int ncols = min(128, L.N - 128 * pass);
        const int ntv = (ncols + 15) >> 4;  // 16-column groups of this chunk with valid columns (8; 4 for N = 192; 1 for N = 3)
        const bool rag_out = RAG && ((ncols & 15) || (old & 3));
        float dgam[8][4], dbet[8][4];
        if (lnb_fin) {
#pragma unroll
          for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) dgam[t][r] = dbet[t][r] = 0.f;
        }
#pragma unroll
        for (int tt = 0; tt < T; ++tt) {
          const int m = rowbase + 16 * tt;
          const bool live = m < A.M;
          const size_t mc = (size_t)(live ? m : A.M - 1);
          float v[8][4];
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[nt][r] = H ? (acc[tt][nt][r] * invx) * invw : acc[tt][nt][r];
            if (nt < ntv) {
              const float4 b = ld4(par + 128 * layer + 128 * pass + 16 * nt + 4 * g);
              v[nt][0] += b.x; v[nt][1] += b.y; v[nt][2] += b.z; v[nt][3] += b.w;
            }
            if (L.op == GFV_OP_MUL_DGELU && nt < ntv) {
              const float4 z = ld4(L.aux + mc * (size_t)L.N + 128 * pass + 16 * nt + 4 * g);
              v[nt][0] *= gfv_dgelu(z.x); v[nt][1] *= gfv_dgelu(z.y); v[nt][2] *= gfv_dgelu(z.z); v[nt][3] *= gfv_dgelu(z.w);
            }
          }
          if (A.fin_op == GFV_FIN_LN) {
            if (A.fin_presave && live) {
#pragma unroll
              for (int nt = 0; nt < 8; ++nt) st4_save(A.fin_presave + mc * 128 + 16 * nt + 4 * g, v[nt]);
            }
            ln_apply(v, par + PAR_GAMMA, par + PAR_BETA, g, lnw, (A.fin_stats && live) ? A.fin_stats + 2 * mc : nullptr);
          } else if (lnb_fin) {
            float y[8][4];
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
              const float4 yy = ld4(A.fin_aux + mc * 128 + 16 * nt + 4 * g);
              y[nt][0] = yy.x; y[nt][1] = yy.y; y[nt][2] = yy.z; y[nt][3] = yy.w;
              const float livef = live ? 1.0f : 0.0f;   // (rows past M: see the prologue form)
              v[nt][0] *= livef; v[nt][1] *= livef; v[nt][2] *= livef; v[nt][3] *= livef;
            }
            if (tt == 0) ln_bwd<true>(v, y, par + PAR_GAMMA, g, dgam, dbet, lnw);
            else ln_bwd<false>(v, y, par + PAR_GAMMA, g, dgam, dbet, lnw);
          }
          if (H && A.gscale && npass == 1 && A.nlayers <= 2 && L.op == GFV_OP_MUL_DGELU && !res) {
            float rs = row_scale(v);             // (all lanes take part in the row / group reductions)
            if (!live) rs = 8.5070592e37f;       // 2^126: a dead row never lowers the group's scale
            group_scale_out(A.gscale + (size_t)A.nlayers * A.gscale_ld + rowgroup, rs, lane);
          }
          if (live) {
            if (pass == 0 && A.out_nores) {
#pragma unroll
              for (int nt = 0; nt < 8; ++nt) st4(A.out_nores + mc * 128 + 16 * nt + 4 * g, v[nt]);
            }
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
              if (nt < ntv) {
                if (rag_out) {
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    const int col = 16 * nt + 4 * g + r;
                    if (col < ncols) out[mc * (size_t)old + col] = v[nt][r] + (res ? res[mc * (size_t)rld + col] : 0.f);
                  }
                } else {
                  if (res) {
                    const float4 rv = ld4(res + mc * (size_t)rld + 16 * nt + 4 * g);
                    v[nt][0] += rv.x; v[nt][1] += rv.y; v[nt][2] += rv.z; v[nt][3] += rv.w;
                  }
                  st4(out + mc * (size_t)old + 16 * nt + 4 * g, v[nt]);
                }
              }
            }
          }
        }
        if (lnb_fin) ln_park(dgam, dbet, red, wave, li, g);
        TS_WAIT();
        TS(7);
      }
    }
  }

#ifdef GFV_TIMING
  if (A.ln_partial && LNM == 0) {
    ts_[8] = clock64() - t_start_;
    ts_[9] = t_start_;
    if (lane == 0)
      for (int k = 0; k < 10; ++k) reinterpret_cast<long long*>(A.ln_partial)[((size_t)blockIdx.x * 4 + wave) * 10 + k] = ts_[k];
    return;
  }
#endif
  if (A.ln_partial) {
    __syncthreads();
    // ln_partial rows are indexed by 64-row tile: every group of 4 waves folds its own (same sums, same order, whatever
    // the workgroup width); a T-tile workgroup owns T consecutive rows
    const int q = tid >> 8, t = tid & 255;
    const float* rq = red + q * 1024;
    const float s = rq[t] + rq[256 + t] + rq[512 + t] + rq[768 + t];
    const size_t tile64 = (size_t)tile * (NW / 4) * T + q * T;
    if (tile64 < (size_t)((A.M + 63) / 64)) A.ln_partial[tile64 * 256 + t] = s;
#pragma unroll
    for (int x = 1; x < T; ++x)
      if (tile64 + x < (size_t)((A.M + 63) / 64)) A.ln_partial[(tile64 + x) * 256 + t] = 0.f;
  }
}

}  // namespace

