// Row-tile fused GEMM chain on fp32 MFMA (gfx950).  See include/gfv.h for the contract.
//
// One workgroup = 4 waves = a tile of 64 rows; wave w owns rows 16w..16w+15 of the tile for the whole chain, so
// the activation tile in LDS is wave-private (no workgroup barrier on it) and LayerNorm / row ops are wave-local.
// Only the weight stream is shared: W is streamed from L2 through a double-buffered LDS stage in 32-wide k
// slices ([128 n][32 k], row stride 36 floats), one barrier per slice, next slice prefetched to registers while
// the current one feeds the MFMAs.
//
// MFMA: v_mfma_f32_16x16x4_f32, exact fp32.  Each wave computes 16 rows x 128 columns = 8 accumulator tiles.
// Operand trick: the k index inside a 16-wide k step is permuted consistently for A and B (lane group q takes
// k = 4q..4q+3), so every lane fetches its four A (and four B) values of four consecutive MFMAs with ONE
// ds_read_b128 from row-major [row][k] / [n][k] images - nn.Linear's [out,in] weight layout is read as stored.
#include <atomic>
#include <stdio.h>
#include <stdlib.h>
#include "gfv_common.h"
#include "gfv_prof.h"
#include "../../include/gfv.h"

namespace {

constexpr int BM = 64;
constexpr int LDX = 132;   // activation tile row stride (floats): 128 + 4 -> rows start 4 banks apart
constexpr int WK = 32;     // k slice of the weight stage
constexpr int LDW = 36;    // weight stage row stride
constexpr int LDO = 132;   // output staging row stride
constexpr int XS_FLOATS = BM * LDX;
constexpr int WS_FLOATS = 128 * LDW;
constexpr int OS_FLOATS = 4 * 4 * LDO;  // per wave 4 rows
constexpr int LDS_FLOATS = XS_FLOATS + 2 * WS_FLOATS + OS_FLOATS;

struct Ctx {
  int tid, wave, lane, nl, q, rr, c4;
  int row0;  // first global row of the tile
  int M;
};

__device__ __forceinline__ void wave_lds_sync() {
  // LDS traffic of one wave is processed in order; make prior ds ops complete and stop compiler reordering.
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
}

// ---- weight slice: global -> registers -> LDS ------------------------------------------------------------
struct WSlice {
  const float* W;  // base of the layer's weight
  int ldw;         // = K of the layer
  int n0;          // first output row of this pass
  int N;           // valid output rows of the layer
  int kcol;        // first weight column of the slice
  int kvalid;      // number of valid columns from kcol (may be <= 0 .. WK)
  int vec;         // float4 loads allowed
};

__device__ __forceinline__ void wslice_load(const WSlice& s, const Ctx& c, float4 (&reg)[4]) {
  const int c4 = c.tid & 7;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int n = (c.tid >> 3) + 32 * p;
    const int gn = s.n0 + n;
    const int k = 4 * c4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gn < s.N) {
      const float* src = s.W + (size_t)gn * s.ldw + s.kcol + k;
      if (s.vec) {
        if (k < s.kvalid) v = *reinterpret_cast<const float4*>(src);
      } else {
        if (k + 0 < s.kvalid) v.x = src[0];
        if (k + 1 < s.kvalid) v.y = src[1];
        if (k + 2 < s.kvalid) v.z = src[2];
        if (k + 3 < s.kvalid) v.w = src[3];
      }
    }
    reg[p] = v;
  }
}

__device__ __forceinline__ void wslice_store(float* Wb, const Ctx& c, const float4 (&reg)[4]) {
  const int c4 = c.tid & 7;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int n = (c.tid >> 3) + 32 * p;
    *reinterpret_cast<float4*>(&Wb[n * LDW + 4 * c4]) = reg[p];
  }
}

// one 32-wide k slice: acc[t] += X[wave rows][kx0..kx0+KSTEPS*16) * Wb^T, statically unrolled (FAST kernel)
template <int KSTEPS>
__device__ __forceinline__ void mma_slice_full(floatx4 (&acc)[8], const float* Xs, const float* Wb, const Ctx& c, int kx0) {
  const float* xrow = Xs + (c.wave * 16 + c.nl) * LDX + kx0 + 4 * c.q;
  const float* wrow = Wb + c.nl * LDW + 4 * c.q;
  float4 a[KSTEPS], b[KSTEPS][8];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    a[ks] = *reinterpret_cast<const float4*>(xrow + 16 * ks);
#pragma unroll
    for (int t = 0; t < 8; ++t) b[ks][t] = *reinterpret_cast<const float4*>(wrow + t * 16 * LDW + 16 * ks);
  }
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks].x, b[ks][t].x, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks].y, b[ks][t].y, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks].z, b[ks][t].z, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks].w, b[ks][t].w, acc[t], 0, 0, 0);
  }
}

// generic slice (narrow outputs / 16-wide tails): only used by the non-FAST kernel
__device__ __forceinline__ void mma_slice_generic(floatx4 (&acc)[8], const float* Xs, const float* Wb, const Ctx& c, int kx0,
                                                  int ksteps, int ntiles) {
  const float* xrow = Xs + (c.wave * 16 + c.nl) * LDX + kx0 + 4 * c.q;
  const float* wrow = Wb + c.nl * LDW + 4 * c.q;
  for (int ks = 0; ks < ksteps; ++ks) {
    const float4 a = *reinterpret_cast<const float4*>(xrow + 16 * ks);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t < ntiles) {
        const float4 b = *reinterpret_cast<const float4*>(wrow + t * 16 * LDW + 16 * ks);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[t], 0, 0, 0);
      }
    }
  }
}

// ---- row helpers in the "coalesced" layout: a row of 128 floats = 32 lanes x float4 --------------------------
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float f4_sum(float4 a) { return (a.x + a.y) + (a.z + a.w); }

// (LayerNorm over the h real columns of a zero-padded 128-column row: gfv_set_hidden_size; the launcher passes h in the
// kernel arguments' pad_ field, as for the register-resident chain - tchain_kernel.h LnW)
struct LnW {
  float inv_n, npad;
};
__device__ __forceinline__ LnW ln_width(int cols) {
  const int n = (cols > 0 && cols < 128) ? cols : 128;
  return LnW{1.0f / (float)n, (float)(128 - n)};
}
__device__ __forceinline__ void row_stats(const float4 v, float& mean, float& rstd, const LnW w) {
  mean = gfv_half_sum(f4_sum(v)) * w.inv_n;
  const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
  const float var = (gfv_half_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) - w.npad * (mean * mean)) * w.inv_n;
  rstd = rsqrtf(var + 1e-5f);  // nn.LayerNorm eps (EPD.py:32)
}

__device__ __forceinline__ float4 row_layernorm(const float4 v, const float4 g, const float4 b, const LnW w) {
  float mean, rstd;
  row_stats(v, mean, rstd, w);
  return make_float4((v.x - mean) * rstd * g.x + b.x, (v.y - mean) * rstd * g.y + b.y,
                     (v.z - mean) * rstd * g.z + b.z, (v.w - mean) * rstd * g.w + b.w);
}

// LayerNorm backward for one row: y = LN input row, go = grad wrt LN output; returns grad wrt LN input and
// accumulates the lane's 4 columns of dgamma / dbeta.
__device__ __forceinline__ float4 row_layernorm_bwd(const float4 y, const float4 go, const float4 g, float4& dgam,
                                                    float4& dbet, const LnW w) {
  float mean, rstd;
  row_stats(y, mean, rstd, w);
  const float4 xh = make_float4((y.x - mean) * rstd, (y.y - mean) * rstd, (y.z - mean) * rstd, (y.w - mean) * rstd);
  const float4 gg = make_float4(go.x * g.x, go.y * g.y, go.z * g.z, go.w * g.w);
  const float m1 = gfv_half_sum(f4_sum(gg)) * w.inv_n;
  const float m2 = gfv_half_sum((gg.x * xh.x + gg.y * xh.y) + (gg.z * xh.z + gg.w * xh.w)) * w.inv_n;
  dgam.x += go.x * xh.x; dgam.y += go.y * xh.y; dgam.z += go.z * xh.z; dgam.w += go.w * xh.w;
  dbet.x += go.x; dbet.y += go.y; dbet.z += go.z; dbet.w += go.w;
  return make_float4(rstd * (gg.x - m1 - xh.x * m2), rstd * (gg.y - m1 - xh.y * m2), rstd * (gg.z - m1 - xh.z * m2),
                     rstd * (gg.w - m1 - xh.w * m2));
}

__device__ __forceinline__ float4 load_row4(const float* base, size_t row, int ld, int col, int width, bool vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* p = base + row * (size_t)ld + col;
  if (vec) {
    if (col < width) v = *reinterpret_cast<const float4*>(p);
  } else {
    if (col + 0 < width) v.x = p[0];
    if (col + 1 < width) v.y = p[1];
    if (col + 2 < width) v.z = p[2];
    if (col + 3 < width) v.w = p[3];
  }
  return v;
}

__device__ __forceinline__ void store_row4(float* base, size_t row, int ld, int col, int width, bool vec, float4 v) {
  float* p = base + row * (size_t)ld + col;
  if (vec) {
    if (col < width) *reinterpret_cast<float4*>(p) = v;
  } else {
    if (col + 0 < width) p[0] = v.x;
    if (col + 1 < width) p[1] = v.y;
    if (col + 2 < width) p[2] = v.z;
    if (col + 3 < width) p[3] = v.w;
  }
}

// ---- stage one input segment (<=128 columns) of the tile into the wave's rows of Xs ---------------------------
__device__ void stage_input(const gfv_rowtile_args_t& A, int si, float* Xs, const Ctx& c, float4& dgam, float4& dbet) {
  const gfv_seg_t& s = A.seg[si];
  const bool vec = ((s.width & 3) == 0) && ((s.ld & 3) == 0);
  const int col = 4 * c.c4;
  float4 gam = make_float4(1.f, 1.f, 1.f, 1.f), bet = make_float4(0.f, 0.f, 0.f, 0.f);
  if (A.in_op == GFV_IN_LN || A.in_op == GFV_IN_LNBWD) {
    gam = *reinterpret_cast<const float4*>(A.in_gamma + col);
    if (A.in_op == GFV_IN_LN) bet = *reinterpret_cast<const float4*>(A.in_beta + col);
  }
#pragma unroll 2
  for (int p = 0; p < 8; ++p) {
    const int r = c.wave * 16 + c.rr + 2 * p;
    const int m = c.row0 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < c.M) {
      const size_t srow = s.idx ? (size_t)s.idx[m] : (size_t)m;
      v = load_row4(s.ptr, srow, s.ld, col, s.width, vec);
      if (si == 0 && A.in_add) v = f4_add(v, load_row4(A.in_add, srow, s.ld, col, s.width, vec));
      if (si == 0 && A.gadd) {
        const int node = (col < 64) ? A.gadd_s[m] : A.gadd_r[m];
        v = f4_add(v, *reinterpret_cast<const float4*>(A.gadd + (size_t)node * 64 + (col & 63)));
      }
    }
    if (A.in_op == GFV_IN_GELU) {
      v = make_float4(gfv_gelu(v.x), gfv_gelu(v.y), gfv_gelu(v.z), gfv_gelu(v.w));
    } else if (A.in_op == GFV_IN_LN) {
      v = row_layernorm(v, gam, bet, ln_width(A.hidden));
    } else if (A.in_op == GFV_IN_LNBWD) {
      float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < c.M) y = *reinterpret_cast<const float4*>(A.in_aux + (size_t)m * 128 + col);
      v = row_layernorm_bwd(y, v, gam, dgam, dbet, ln_width(A.hidden));
    }
    if (m >= c.M) v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (si == 0 && A.in_save && m < c.M) *reinterpret_cast<float4*>(A.in_save + (size_t)m * 128 + col) = v;
    *reinterpret_cast<float4*>(&Xs[r * LDX + col]) = v;
  }
}

// write the accumulators of the wave's 16x128 slab into its rows of Xs
__device__ __forceinline__ void acc_to_xs(const floatx4 (&acc)[8], float* Xs, const Ctx& c) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) Xs[(c.wave * 16 + 4 * c.q + reg) * LDX + 16 * t + c.nl] = acc[t][reg];
  }
}

// intermediate epilogue: element op in the coalesced layout, result stays in Xs as the next layer's input
__device__ void mid_epilogue(const gfv_rowtile_args_t& A, int layer, const gfv_layer_t& L, float* Xs, const Ctx& c) {
  const int col = 4 * c.c4;
  float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
  if (L.bias && L.op != GFV_OP_MUL_DGELU) bias = *reinterpret_cast<const float4*>(L.bias + col);
#pragma unroll 2
  for (int p = 0; p < 8; ++p) {
    const int r = c.wave * 16 + c.rr + 2 * p;
    const int m = c.row0 + r;
    float4 v = f4_add(*reinterpret_cast<const float4*>(&Xs[r * LDX + col]), bias);
    if (layer == 0 && A.padd && m < c.M) {
      v = f4_add(v, *reinterpret_cast<const float4*>(A.padd + (size_t)A.padd_s[m] * A.padd_ld + col));
      v = f4_add(v, *reinterpret_cast<const float4*>(A.padd + (size_t)A.padd_r[m] * A.padd_ld + 128 + col));
    }
    if (L.op == GFV_OP_BIAS_GELU) {
      if (L.save && m < c.M) *reinterpret_cast<float4*>(L.save + (size_t)m * 128 + col) = v;
      v = make_float4(gfv_gelu(v.x), gfv_gelu(v.y), gfv_gelu(v.z), gfv_gelu(v.w));
    } else if (L.op == GFV_OP_MUL_DGELU) {
      float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < c.M) z = *reinterpret_cast<const float4*>(L.aux + (size_t)m * 128 + col);
      v = make_float4(v.x * gfv_dgelu(z.x), v.y * gfv_dgelu(z.y), v.z * gfv_dgelu(z.z), v.w * gfv_dgelu(z.w));
      if (L.save && m < c.M) *reinterpret_cast<float4*>(L.save + (size_t)m * 128 + col) = v;
    }
    *reinterpret_cast<float4*>(&Xs[r * LDX + col]) = v;
  }
}

// final epilogue for one 128-wide output chunk: 4 rounds of 4 rows through the wave's staging buffer
__device__ void final_epilogue(const gfv_rowtile_args_t& A, const gfv_layer_t& L, int chunk, const floatx4 (&acc)[8],
                               float* Os, const Ctx& c, float4& dgam, float4& dbet) {
  float* os = Os + c.wave * 4 * LDO;
  const int ncol = L.N - 128 * chunk;
  const int width = ncol < 128 ? ncol : 128;
  const int old = A.out_ld[chunk];
  const bool ovec = ((width & 3) == 0) && ((old & 3) == 0);
  const int col = 4 * c.c4;
  float* out = A.out[chunk];
  const float* res = A.res[chunk];
  const int rld = A.res_ld[chunk];
  float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
  if (L.bias) bias = load_row4(L.bias, 0, 0, 128 * chunk + col, L.N, (L.N & 3) == 0);
  float4 gam = make_float4(1.f, 1.f, 1.f, 1.f), bet = make_float4(0.f, 0.f, 0.f, 0.f);
  if (A.fin_op != GFV_FIN_PLAIN) {
    gam = *reinterpret_cast<const float4*>(A.fin_gamma + col);
    if (A.fin_op == GFV_FIN_LN) bet = *reinterpret_cast<const float4*>(A.fin_beta + col);
  }
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    // rows {reg, 4+reg, 8+reg, 12+reg} of the wave's slab -> staging rows 0..3 (index q)
#pragma unroll
    for (int t = 0; t < 8; ++t) os[c.q * LDO + 16 * t + c.nl] = acc[t][reg];
    wave_lds_sync();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int sr = c.rr + 2 * p;  // staging row = q index
      const int r = c.wave * 16 + 4 * sr + reg;
      const int m = c.row0 + r;
      float4 v = f4_add(*reinterpret_cast<const float4*>(&os[sr * LDO + col]), bias);
      const bool live = m < c.M;
      if (L.op == GFV_OP_MUL_DGELU) {
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) z = load_row4(L.aux, (size_t)m, L.N, 128 * chunk + col, L.N, true);
        v = make_float4(v.x * gfv_dgelu(z.x), v.y * gfv_dgelu(z.y), v.z * gfv_dgelu(z.z), v.w * gfv_dgelu(z.w));
      }
      if (A.fin_op == GFV_FIN_LN) {
        if (A.fin_presave && live) *reinterpret_cast<float4*>(A.fin_presave + (size_t)m * 128 + col) = v;
        v = row_layernorm(v, gam, bet, ln_width(A.hidden));
      } else if (A.fin_op == GFV_FIN_LNBWD) {
        float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) y = *reinterpret_cast<const float4*>(A.fin_aux + (size_t)m * 128 + col);
        if (!live) v = make_float4(0.f, 0.f, 0.f, 0.f);
        v = row_layernorm_bwd(y, v, gam, dgam, dbet, ln_width(A.hidden));
      }
      if (live) {
        if (chunk == 0 && A.out_nores) *reinterpret_cast<float4*>(A.out_nores + (size_t)m * 128 + col) = v;
        if (res) v = f4_add(v, load_row4(res, (size_t)m, rld, col, width, ovec && ((rld & 3) == 0)));
        store_row4(out, (size_t)m, old, col, width, ovec, v);
      }
    }
    wave_lds_sync();
  }
}

// FAST: every layer width is a multiple of 128 and every input segment a multiple of 32 columns with 16-byte
// aligned rows -> statically unrolled MFMA slices, accumulators never move between registers.
template <bool FAST>
__global__ __launch_bounds__(256, 2) void rowtile_chain_kernel(const gfv_rowtile_args_t A) {
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  float* Xs = lds;
  float* Wb0 = lds + XS_FLOATS;
  float* Wb1 = Wb0 + WS_FLOATS;
  float* Os = Wb1 + WS_FLOATS;

  Ctx c;
  c.tid = threadIdx.x;
  c.wave = c.tid >> 6;
  c.lane = c.tid & 63;
  c.nl = c.lane & 15;
  c.q = c.lane >> 4;
  c.rr = c.lane >> 5;
  c.c4 = c.lane & 31;
  c.row0 = blockIdx.x * BM;
  c.M = A.M;

  float4 dgam = make_float4(0.f, 0.f, 0.f, 0.f), dbet = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 wreg[4];
  int wbuf = 0;

  // flat weight-slice sequence: (layer, pass, input chunk, k slice)
  auto make_slice = [&](int layer, int pass, int chunk, int ks) {
    WSlice s;
    const gfv_layer_t& L = A.layer[layer];
    s.W = L.W;
    s.ldw = L.ldw ? L.ldw : L.K;
    s.n0 = 128 * pass;
    s.N = L.N;
    int koff = 0, width = 128;
    if (layer == 0) {
      for (int i = 0; i < chunk; ++i) koff += A.seg[i].width;
      width = A.seg[chunk].width;
    }
    s.kcol = koff + ks;
    s.kvalid = width - ks;
    s.vec = ((s.ldw & 3) == 0) && ((koff & 3) == 0) && ((width & 3) == 0) && ((reinterpret_cast<size_t>(L.W) & 15) == 0);
    return s;
  };

  {
    WSlice s0 = make_slice(0, 0, 0, 0);
    wslice_load(s0, c, wreg);
    wslice_store(Wb0, c, wreg);
  }
  __syncthreads();

  for (int layer = 0; layer < A.nlayers; ++layer) {
    const gfv_layer_t& L = A.layer[layer];
    const bool last = (layer == A.nlayers - 1);
    const int npass = last ? (L.N + 127) / 128 : 1;
    const int nchunk = (layer == 0) ? A.nseg : 1;
    for (int pass = 0; pass < npass; ++pass) {
      floatx4 acc[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
      const int nrem = L.N - 128 * pass;
      const int ntiles = nrem >= 128 ? 8 : (nrem + 15) / 16;
      for (int chunk = 0; chunk < nchunk; ++chunk) {
        int width = 128;
        if (layer == 0) {
          width = A.seg[chunk].width;
          if (nchunk > 1 || pass == 0) {
            wave_lds_sync();
            stage_input(A, chunk, Xs, c, dgam, dbet);
            wave_lds_sync();
          }
        }
        const int kpad = (width + 15) & ~15;
        for (int ks = 0; ks < kpad; ks += WK) {
          // locate the next slice in the flat sequence
          int nl_ = layer, np_ = pass, nc_ = chunk, nk_ = ks + WK;
          bool have_next = true;
          if (nk_ >= kpad) {
            nk_ = 0;
            nc_ = chunk + 1;
            if (nc_ >= nchunk) {
              nc_ = 0;
              np_ = pass + 1;
              if (np_ >= npass) {
                np_ = 0;
                nl_ = layer + 1;
                if (nl_ >= A.nlayers) have_next = false;
              }
            }
          }
          if (have_next) {
            WSlice sn = make_slice(nl_, np_, nc_, nk_);
            wslice_load(sn, c, wreg);
          }
          if constexpr (FAST) {
            mma_slice_full<2>(acc, Xs, wbuf ? Wb1 : Wb0, c, ks);
          } else {
            const int ksteps = (kpad - ks) >= WK ? 2 : 1;
            mma_slice_generic(acc, Xs, wbuf ? Wb1 : Wb0, c, ks, ksteps, ntiles);
          }
          if (have_next) wslice_store(wbuf ? Wb0 : Wb1, c, wreg);
          __syncthreads();
          wbuf ^= 1;
        }
      }
      if (!last) {
        wave_lds_sync();
        acc_to_xs(acc, Xs, c);
        wave_lds_sync();
        mid_epilogue(A, layer, L, Xs, c);
        wave_lds_sync();
      } else {
        final_epilogue(A, L, pass, acc, Os, c, dgam, dbet);
      }
    }
  }

  if (A.ln_partial) {
    // (dgamma, dbeta) of the tile: lane halves -> waves -> global partial row
    dgam.x += __shfl_xor(dgam.x, 32, 64); dgam.y += __shfl_xor(dgam.y, 32, 64);
    dgam.z += __shfl_xor(dgam.z, 32, 64); dgam.w += __shfl_xor(dgam.w, 32, 64);
    dbet.x += __shfl_xor(dbet.x, 32, 64); dbet.y += __shfl_xor(dbet.y, 32, 64);
    dbet.z += __shfl_xor(dbet.z, 32, 64); dbet.w += __shfl_xor(dbet.w, 32, 64);
    __syncthreads();
    float* red = Xs;  // [4 waves][2][128]
    if (c.lane < 32) {
      *reinterpret_cast<float4*>(&red[(c.wave * 2 + 0) * 128 + 4 * c.c4]) = dgam;
      *reinterpret_cast<float4*>(&red[(c.wave * 2 + 1) * 128 + 4 * c.c4]) = dbet;
    }
    __syncthreads();
    const int j = c.tid;  // 0..255 = [2][128]
    const float s = red[j] + red[256 + j] + red[512 + j] + red[768 + j];
    A.ln_partial[(size_t)blockIdx.x * 256 + j] = s;
  }
}

}  // namespace

int gfv_internal_tchain_launch(const gfv_rowtile_args_t* args, int ragged, int f16, hipStream_t stream);  // tchain.hip

static int tchain_mode() {
  // GFV_TCHAIN: 0 = LDS row-tile kernel for everything, otherwise the register-resident chain (default)
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("GFV_TCHAIN");
    mode = e ? atoi(e) : 64;
  }
  return mode;
}

// GFV_F16SPLIT (or gfv_set_f16split): 0 = every GEMM product on the fp32 MFMA, even when a launch carries split-fp16
// weight images; 1 (default) = split-fp16 products; 2 = the reduced-precision form, ONE fp16 x fp16 product with fp32
// accumulation (the high parts only); shared with dw.hip
// State: a PROCESS-WIDE default (atomic; gfv_set_f16split) - PyTorch runs the backward of an autograd node on its device
// worker thread, and a form chosen on the user's thread must reach the launches issued there - and a per-thread OVERRIDE
// (gfv_set_f16split_thread; -1 = none) for two host threads driving two models in different forms; a launch may also carry its
// own, gfv_rowtile_args_t.product_form / .hidden
static std::atomic<int> g_f16_default{-1};
static thread_local int g_f16split = -1;   // this thread's override
static int f16_default() {
  int d = g_f16_default.load(std::memory_order_relaxed);
  if (d < 0) {
    const char* e = getenv("GFV_F16SPLIT");
    d = e ? (atoi(e) == 2 ? 2 : (atoi(e) != 0)) : 1;
    int expect = -1;
    if (!g_f16_default.compare_exchange_strong(expect, d)) d = expect;
  }
  return d;
}
extern "C" int gfv_f16split_enabled(void) { return g_f16split >= 0 ? g_f16split : f16_default(); }
extern "C" int gfv_set_f16split(int32_t on) {
  g_f16_default.store(on == 2 ? 2 : (on ? 1 : 0));
  g_f16split = -1;   // (the caller asked for the process-wide form: its own override, if any, would hide it)
  return GFV_OK;
}
extern "C" int gfv_set_f16split_thread(int32_t on) {
  if (on < -1 || on > 2) return GFV_ERR_ARG;
  g_f16split = on;
  return GFV_OK;
}
static int f16_mode() { return gfv_f16split_enabled(); }

// hidden_size of the model the following launches belong to (utils/get_param.py:69; default 128).  A model of h < 128 runs
// zero-padded to 128 columns (FVMmodel/padding.py): what changes in the kernels is the LayerNorm width (statistics over the h
// real columns) and the attention scale (dim_head = h / 8).  Host-side state read by the launchers (chain: passed in the
// kernel arguments, also of the generic row-tile kernel; weight gradient: DwLaunch; slice attention: its argument struct) -
// set it before the launches of a model, gfv.engine.Engine does on every forward / backward.
static thread_local int g_hidden = 128;
extern "C" int gfv_hidden_size(void) { return g_hidden; }
extern "C" int gfv_set_hidden_size(int32_t h) {
  if (h < 16 || h > 128 || (h & 15)) return GFV_ERR_ARG;
  g_hidden = h;   // (host state only: safe inside a stream capture)
  return GFV_OK;
}

static thread_local int g_last_path = -1;
extern "C" int gfv_rowtile_last_path(void) { return g_last_path; }

extern "C" int gfv_rowtile_tiles(int32_t M) { return (M + BM - 1) / BM; }

static int rowtile_chain_impl(const gfv_rowtile_args_t* args, void* stream);
extern "C" int gfv_rowtile_chain(const gfv_rowtile_args_t* args, void* stream) {
  if (!args) return GFV_ERR_ARG;
  // the launch's own product form / hidden size (0: the calling thread's context)
  if (args->product_form < 0 || args->product_form > 3 || (args->hidden != 0 && (args->hidden < 16 || args->hidden > 128 || (args->hidden & 15))))
    return GFV_ERR_ARG;
  if (args->product_form == 0 && args->hidden == 0) return rowtile_chain_impl(args, stream);
  const int f0 = g_f16split, h0 = g_hidden;   // (f0: this thread's override or -1; restored below)
  if (args->product_form) g_f16split = args->product_form - 1;   // 1 fp32 MFMA, 2 split-fp16, 3 reduced precision
  if (args->hidden) g_hidden = args->hidden;
  const int rc = rowtile_chain_impl(args, stream);
  g_f16split = f0;
  g_hidden = h0;
  return rc;
}
int gfv_internal_lin1_try(const gfv_rowtile_args_t* a, int lowp, hipStream_t stream, int dry);   // lin1.hip
static int rowtile_chain_impl(const gfv_rowtile_args_t* args, void* stream) {
  if (!args || args->M < 0 || args->nlayers < 1 || args->nlayers > 3 || args->nseg < 1 || args->nseg > 3) return GFV_ERR_ARG;
  if (args->M == 0) return GFV_OK;
  for (int i = 0; i < args->nseg; ++i)
    if (args->seg[i].width < 1 || args->seg[i].width > 128) return GFV_ERR_ARG;
  for (int l = 0; l + 1 < args->nlayers; ++l)
    if (args->layer[l].N != 128 || args->layer[l + 1].K != 128) return GFV_ERR_ARG;
  int k0 = 0;
  for (int i = 0; i < args->nseg; ++i) k0 += args->seg[i].width;
  if (k0 != args->layer[0].K) return GFV_ERR_ARG;
  const gfv_layer_t& last = args->layer[args->nlayers - 1];
  if (last.N > 384 || last.N < 1) return GFV_ERR_ARG;
  if ((args->rc_Wh[0] || args->rc_Wh[1]) && !args->dw_partial) return GFV_ERR_ARG;   // (recompute: the fused column-owner backward only)
  if (args->padd && (args->nlayers < 2 || args->padd_ld < 256 || (args->padd_ld & 3) || !args->padd_s || !args->padd_r)) return GFV_ERR_ARG;
  for (int l = 0; l < args->nlayers; ++l)
    if (args->layer[l].ldw != 0 && args->layer[l].ldw < args->layer[l].K) return GFV_ERR_ARG;
  if (last.N > 128 && (last.N & 15)) return GFV_ERR_ARG;
  if ((args->fin_op != GFV_FIN_PLAIN || args->in_op == GFV_IN_LN || args->in_op == GFV_IN_LNBWD) &&
      (args->fin_op != GFV_FIN_PLAIN ? last.N != 128 : false))
    return GFV_ERR_ARG;
  if ((args->in_op == GFV_IN_LN || args->in_op == GFV_IN_LNBWD) && (args->nseg != 1 || args->seg[0].width != 128))
    return GFV_ERR_ARG;
  const int tiles = (args->M + BM - 1) / BM;
  bool fast = true;
  for (int i = 0; i < args->nseg; ++i)
    fast = fast && (args->seg[i].width % 32 == 0) && (args->seg[i].ld % 4 == 0);
  for (int l = 0; l < args->nlayers; ++l) fast = fast && (args->layer[l].N % 128 == 0) && (args->layer[l].K % 4 == 0) && (args->layer[l].ldw % 4 == 0);
  // the register-resident chain also takes a last layer whose final 128-chunk is 64 wide (NodeBlock dX: 128 + 64)
  bool fast_t = true;
  for (int i = 0; i < args->nseg; ++i) fast_t = fast_t && (args->seg[i].width % 32 == 0) && (args->seg[i].ld % 4 == 0);
  for (int l = 0; l < args->nlayers; ++l) {
    const bool lastl = (l == args->nlayers - 1);
    fast_t = fast_t && (args->layer[l].K % 4 == 0) && (args->layer[l].ldw % 4 == 0) && (args->layer[l].N % (lastl ? 64 : 128) == 0);
  }
  if (args->fin_op != GFV_FIN_PLAIN) fast_t = fast_t && last.N == 128;
  for (int c = 0; c < 3; ++c) {
    if (args->out[c]) { fast = fast && (args->out_ld[c] % 4 == 0); fast_t = fast_t && (args->out_ld[c] % 4 == 0); }
    if (args->res[c]) { fast = fast && (args->res_ld[c] % 4 == 0); fast_t = fast_t && (args->res_ld[c] % 4 == 0); }
  }
  // ragged shapes the register-resident chain also takes (its RAG instantiation): any segment width / row stride, any
  // first-layer K, a last layer of any width <= 128 (decoder N = 3); no LayerNorm backward, no DGELU on a ragged last
  // layer, inner layers 128 wide
  bool rag_t = !fast_t && args->in_op != GFV_IN_LNBWD && args->fin_op != GFV_FIN_LNBWD;
  for (int l = 0; l + 1 < args->nlayers; ++l) rag_t = rag_t && args->layer[l].N == 128;
  if (last.N > 128) rag_t = rag_t && (last.N % 64 == 0);
  if (last.N % 16) rag_t = rag_t && last.op != GFV_OP_MUL_DGELU && args->fin_op == GFV_FIN_PLAIN && !args->out_nores;
  if (args->in_op == GFV_IN_LN) rag_t = rag_t && args->seg[0].width == 128 && (args->seg[0].ld % 4 == 0);
  if (args->gadd) rag_t = rag_t && args->seg[0].width == 128 && (args->seg[0].ld % 4 == 0);
  for (int l = 1; l < args->nlayers; ++l) rag_t = rag_t && (args->layer[l].K == 128);
  // segmented-sum segments / per-segment saves: the plain instantiation of the register-resident chain only
  {
    bool csr = false;
    for (int i = 0; i < args->nseg; ++i) {
      const gfv_seg_t& sg = args->seg[i];
      csr = csr || sg.csr_rowptr != nullptr || sg.save != nullptr;
      if (sg.csr_rowptr && (!sg.idx || (i == 0 && args->in_add))) return GFV_ERR_ARG;
    }
    if (csr && (!fast_t || tchain_mode() == 0 || args->in_op == GFV_IN_LNBWD || args->fin_op == GFV_FIN_LNBWD))
      return GFV_ERR_ARG;
  }
  // split-fp16 form: every layer carries a weight image; first-layer segments start at 32-k slice boundaries
  bool f16 = (fast_t || rag_t) && tchain_mode() != 0 && f16_mode() != 0 && args->wmax != nullptr;
  for (int l = 0; l < args->nlayers; ++l) f16 = f16 && args->layer[l].Wh != nullptr;
  for (int i = 0; i + 1 < args->nseg; ++i) f16 = f16 && (args->seg[i].width % 32 == 0);
  // a layer may come without fp32 weights (a row-stacked virtual layer that exists as an image only): split form or nothing
  for (int l = 0; l < args->nlayers; ++l)
    if ((!args->layer[l].W || args->layer[l].bias2) && !f16) return GFV_ERR_ARG;
  static const bool dbg = getenv("GFV_ROWTILE_DEBUG") != nullptr;
  if (dbg && !fast_t && !rag_t) {
    fprintf(stderr, "[gfv] generic rowtile: M=%d nseg=%d widths=%d,%d,%d ld0=%d nlayers=%d K0=%d Nlast=%d out_ld=%d in_op=%d fin_op=%d\n",
            args->M, args->nseg, args->seg[0].width, args->nseg > 1 ? args->seg[1].width : 0,
            args->nseg > 2 ? args->seg[2].width : 0, args->seg[0].ld, args->nlayers, args->layer[0].K, last.N,
            args->out_ld[0], args->in_op, args->fin_op);
  }
  void* tok = nullptr;
  if (gfv_prof_enabled()) {
    // algorithmic work: 2*M*K*N flops per layer; bytes: every input row read once, every output/saved row written once
    double fl = 0, by = 0;
    for (int l = 0; l < args->nlayers; ++l) {
      fl += 2.0 * args->M * (double)args->layer[l].K * args->layer[l].N;
      by += 4.0 * ((double)args->layer[l].K * args->layer[l].N + args->layer[l].N);
      if (args->layer[l].save) by += 4.0 * args->M * args->layer[l].N;
      if (args->layer[l].aux) by += 4.0 * args->M * args->layer[l].N;
    }
    by += 4.0 * args->M * (double)args->layer[0].K + 4.0 * args->M * (double)last.N;
    for (int i = 0; i < args->nseg; ++i) if (args->seg[i].idx) by += 4.0 * args->M;
    if (args->in_aux || args->fin_aux) by += 4.0 * args->M * 128.0;
    if (args->in_save) by += 4.0 * args->M * 128.0;
    if (args->fin_presave) by += 4.0 * args->M * 128.0;
    if (args->out_nores) by += 4.0 * args->M * 128.0;
    if (args->padd) by += 4.0 * args->M * 258.0;
    for (int c = 0; c < 3; ++c) if (args->res[c]) by += 4.0 * args->M * 128.0;
    const int lnm = args->in_op == GFV_IN_LNBWD ? 1 : (args->fin_op == GFV_FIN_LNBWD ? 2 : 0);
    int kind = (fast_t && tchain_mode() != 0) ? GFV_K_TCHAIN0 + lnm : ((rag_t && tchain_mode() != 0) ? GFV_K_TCHAIN_RAG : GFV_K_ROWTILE);
    for (int i = 0; i < args->nseg; ++i)
      if (args->seg[i].csr_rowptr || args->seg[i].save) kind = GFV_K_TCHAIN_CSR;
    if (fast_t && f16 && args->nlayers == 1 && gfv_internal_lin1_try(args, f16_mode() == 2 ? 1 : 0, (hipStream_t)stream, 1))
      kind = GFV_K_LIN1;   // the lean single-layer kernel (same algorithmic work as the chain launch it stands in for)
    if (args->dw_partial) {
      // dX chain with fused weight gradients (column-owner backward family): + the two (three) weight-gradient GEMMs, the
      // forward's row statistics read, the first Linear's input rows read, the per-workgroup blocks written
      kind = GFV_K_COLCHAIN_BWD;
      const int nfused = (args->dw_in ? 3 : 2) + (args->rc_Wh[0] ? 2 : 0);   // (+ the two recomputed forward layers)
      fl += nfused * 2.0 * args->M * 128.0 * 128.0;
      by += 8.0 * args->M + (args->dw_in ? 4.0 * args->M * 128.0 : 0.0) + 4.0 * (double)gfv_rowtile_dw_partials() * (double)args->dw_partial_stride;
    }
    tok = gfv_prof_begin(kind, fl, by, (hipStream_t)stream);
  }
  g_last_path = (tchain_mode() != 0 && (fast_t || rag_t)) ? ((fast_t ? 1 : 2) + (f16 ? 4 : 0)) : 0;
  if (args->dw_partial && !(fast_t && tchain_mode() != 0 && f16)) return GFV_ERR_ARG;   // (fused weight gradients: ask gfv_rowtile_fuses_dw first)
  if (fast_t && f16 && args->nlayers == 1 && gfv_internal_lin1_try(args, f16_mode() == 2 ? 1 : 0, (hipStream_t)stream, 0)) {
    g_last_path += 32;   // the lean single-layer kernel (lin1.hip)
  } else if (fast_t && tchain_mode() != 0) {
    const int took = gfv_internal_tchain_launch(args, 0, f16 ? 1 : 0, (hipStream_t)stream);   // 1: the column-owner family, 2: with fused dW
    if (args->dw_partial && took != 2) return GFV_ERR_ARG;
    g_last_path += 8 * took;
  }
  else if (rag_t && tchain_mode() != 0)
    gfv_internal_tchain_launch(args, 1, f16 ? 1 : 0, (hipStream_t)stream);
  else {
    if (args->fin_stats) return GFV_ERR_ARG;   // (the generic row-tile kernel does not write them)
    gfv_rowtile_args_t local = *args;
    local.hidden = g_hidden;   // LayerNorm width (gfv_set_hidden_size)
    if (fast) hipLaunchKernelGGL(rowtile_chain_kernel<true>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, local);
    else hipLaunchKernelGGL(rowtile_chain_kernel<false>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, local);
  }
  gfv_prof_end(tok, (hipStream_t)stream);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}
