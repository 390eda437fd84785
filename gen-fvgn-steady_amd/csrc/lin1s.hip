// gfv-build-flags: -fno-slp-vectorize
// Column-owner SMALL-TILE form of the single-layer launches (round 5): what lin1.hip does for launches of tens of thousands of
// rows, for SHORT launches - every single-layer launch of a small mesh (28 per step: the EdgeBlock's node projection and its
// adjoint, the eight Linear launches of a Transolver block, GraphTransolver.py:51-62,93-95,163-169).
//
// Why (profiles/r05_latency_floor_before.txt): lin1_kernel gives a wave 16 rows and ALL output columns - 96 to 384 MFMAs and a
// whole 64 - 128 KB image read out of LDS per wave - behind a full image staging; its launch takes 9 - 15 us however few rows it
// has.  Here, as in cfwd.hip: a workgroup = 4 waves = one tile of TG groups of 16 rows (TG = 2; 4 for the LayerNorm-backward
// epilogue, whose (dgamma, dbeta) partials are per 64-row tile); the first TG waves load, transform and split the tile's rows
// (the prologues of lin1.hip unchanged: in_add / in_save, GELU, LayerNorm, two segmented sums, the group scales for the
// weight-gradient launch) into MFMA B fragments in LDS; then wave w owns output columns 32 w .. 32 w + 31 of every 128-column
// pass, takes its slice of the image straight from L2 into registers and runs 48 MFMAs per group pair and 128-deep pass; bias /
// residual / GELU' / LayerNorm-backward epilogues on its own columns.  One barrier (two with the LayerNorm backward).
// Every tile pulls the whole image from L2 (64 - 128 KB): above GFV_LIN1S_MAX_M rows (default 16 384) lin1.hip keeps the launch.
#include <atomic>
#include <cstdlib>

#include "tchain_kernel.h"
#include "gfv_limits.h"

int* gfv_internal_status_ptr();
extern "C" int gfv_hidden_size(void);

namespace {

__device__ __forceinline__ void l1s_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct Lin1sArgs {
  const float* seg[2];
  int seg_ld[2];
  const int* rowptr[2];   // CSR: segment s of row m = sum of the rows col[rowptr[m] .. rowptr[m + 1]) of seg[s]
  const int* col[2];
  float* save[2];         // CSR: the assembled rows [M, 128]
  const float* in_add;
  float* in_save;
  const void* img;        // [pass][T][nt][hi 64 lanes | lo 64 lanes] x 16 B (gfv_weight_images)
  const float* wmax;
  const float* bias[2];
  const float* res[2];
  int res_ld[2];
  float* out[2];
  int out_ld[2];
  int M;
  const float* gamma;     // IN_OP 2 (LayerNorm prologue) / EPI 2 (LayerNorm backward: the LayerNorm's weight)
  const float* beta;
  float ln_inv_n, ln_npad;
  const float* aux;       // EPI 1: the saved pre-activations [M, 128 NP]; EPI 2: the LayerNorm's input rows [M, 128]
  float* gscale;
  float* ln_partial;      // EPI 2: [n_tiles, 2, 128]
};

template <int KS, int TG>
struct L1sLds {
  static constexpr int FRAG = 0;                       // [TG][KS][2][64] x 16 B
  static constexpr int SINV = TG * KS * 2048;          // float [TG * 16]
  static constexpr int STAT = SINV + TG * 64;          // float2 [TG * 16]: (mean, rstd) of the LayerNorm input rows (EPI 2)
  static constexpr int PART = STAT + TG * 128;         // float2 [TG * 16][4]: per-wave (s1, s2) of a row (EPI 2)
  static constexpr int TOTAL = PART + TG * 16 * 4 * 8;
};

// KS: 32-wide k-groups (4 / 8); NP: 128-column output passes; IN_OP: 0 none, 1 GELU, 2 LayerNorm; EPI: 0 bias / residual,
// 1 x GELU'(aux), 2 LayerNorm backward (+ residual, per-tile partials); CSR: two segmented-sum segments; LOWP: product form
// (CSR with KS = 4: ONE segmented-sum segment - the GnBlock's neighbour sum in front of the EdgeBlock's node projection)
template <int KS, int NP, int IN_OP, int EPI, bool CSR, int LOWP, int TG>
__global__ __launch_bounds__(256, 2) void lin1s_kernel(const Lin1sArgs A, int* status) {
  constexpr int NSEG = KS / 4;   // 128-wide segments of the input
  using LY = L1sLds<KS, TG>;
  constexpr bool BF = LOWP == 2;
  __shared__ __attribute__((aligned(16))) char lds[LY::TOTAL];
  float* sinv = reinterpret_cast<float*>(lds + LY::SINV);
  float* stat = reinterpret_cast<float*>(lds + LY::STAT);
  float* part = reinterpret_cast<float*>(lds + LY::PART);
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int tile = CSR ? gfv_xcd_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int row0 = tile * (16 * TG);
  if (row0 >= A.M) return;
  const int ngt = min(TG, (A.M - row0 + 15) >> 4);
  const int c0 = 32 * w + 4 * g;
  const float invw = 1.0f / gfv_pow2_scale(*A.wmax);

  // ---- the first pass's weight slice: in flight beside the row loads ----
  gfv_f16x8 wh[2][KS], wl[2][KS];
  auto load_w = [&](int p) {
    const gfv_f16x8* im = reinterpret_cast<const gfv_f16x8*>(A.img) + (size_t)p * KS * 1024 + (size_t)(2 * w) * 128 + lane;
#pragma unroll
    for (int T = 0; T < KS; ++T)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        wh[n][T] = im[T * 1024 + n * 128];
        if (!LOWP) wl[n][T] = im[T * 1024 + n * 128 + 64];
      }
  };
  // (behind the gathers in the segmented-sum form: its loader has four neighbour rows per segment pair in flight - 128 registers)
  if (!CSR) load_w(0);

  // ---- loader waves: rows -> prologue -> row scale -> fragments ----
  if (w < TG) {
    const int m = row0 + 16 * w + j;
    const bool live = m < A.M;
    const size_t mr = (size_t)(live ? m : A.M - 1);
    float v[KS][8];
    if (CSR) {
      // two segmented sums, walked together, two neighbour rows of each in flight (lin1.hip lin1_csr_kernel; CSR order per segment)
#pragma unroll
      for (int T = 0; T < KS; ++T)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[T][e] = 0.f;
      const int mc = (int)mr;
      int kk[2], ee[2], cn0[2], cn1[2];
      kk[1] = ee[1] = 0;
#pragma unroll
      for (int s = 0; s < NSEG; ++s) {
        kk[s] = A.rowptr[s][mc];
        ee[s] = live ? A.rowptr[s][mc + 1] : kk[s];
        cn0[s] = kk[s] < ee[s] ? A.col[s][kk[s]] : 0;
        cn1[s] = kk[s] + 1 < ee[s] ? A.col[s][kk[s] + 1] : cn0[s];
      }
      while (kk[0] < ee[0] || kk[1] < ee[1]) {
        float4 a0[NSEG][8], a1[NSEG][8];
        bool one[NSEG], two[NSEG];
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
          const int k = kk[s], end = ee[s];
          one[s] = k < end;
          two[s] = k + 1 < end;
          const int cc0 = cn0[s], cc1 = cn1[s];
          if (k + 2 < end) {
            cn0[s] = A.col[s][k + 2];
            cn1[s] = A.col[s][k + 3 < end ? k + 3 : k + 2];
          }
          const float* p0 = A.seg[s] + (size_t)cc0 * (size_t)A.seg_ld[s] + 4 * g;
          const float* p1 = A.seg[s] + (size_t)cc1 * (size_t)A.seg_ld[s] + 4 * g;
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
        for (int s = 0; s < NSEG; ++s) {
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            float* d = &v[(4 * s + (t >> 1)) % KS][4 * (t & 1)];
            if (one[s]) { d[0] += a0[s][t].x; d[1] += a0[s][t].y; d[2] += a0[s][t].z; d[3] += a0[s][t].w; }
            if (two[s]) { d[0] += a1[s][t].x; d[1] += a1[s][t].y; d[2] += a1[s][t].z; d[3] += a1[s][t].w; }
          }
          kk[s] = min(kk[s] + 2, ee[s]);
        }
      }
#pragma unroll
      for (int s = 0; s < NSEG; ++s) {
        if (A.save[s] && live) {
#pragma unroll
          for (int T = 0; T < 4; ++T) {
            float* sp = A.save[s] + (size_t)m * 128 + 32 * T + 4 * g;
            const float (&r)[8] = v[(4 * s + T) % KS];
            *reinterpret_cast<float4*>(sp) = make_float4(r[0], r[1], r[2], r[3]);
            *reinterpret_cast<float4*>(sp + 16) = make_float4(r[4], r[5], r[6], r[7]);
          }
        }
      }
    } else {
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
    }
    if (IN_OP == 1) {
#pragma unroll
      for (int T = 0; T < KS; ++T)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[T][e] = gfv_gelu(v[T][e]);
    }
    if (IN_OP == 2) {   // LayerNorm of the row (lin1.hip / tchain_kernel.h ln_stats: the same sums in the same order)
      float sm = 0.f;
#pragma unroll
      for (int T = 0; T < KS; ++T) {
        sm += (v[T][0] + v[T][1]) + (v[T][2] + v[T][3]);
        sm += (v[T][4] + v[T][5]) + (v[T][6] + v[T][7]);
      }
      const float mean = row_sum(sm) * A.ln_inv_n;
      float qq = 0.f;
#pragma unroll
      for (int T = 0; T < KS; ++T) {
        const float d0 = v[T][0] - mean, d1 = v[T][1] - mean, d2 = v[T][2] - mean, d3 = v[T][3] - mean;
        qq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        const float d4 = v[T][4] - mean, d5 = v[T][5] - mean, d6 = v[T][6] - mean, d7 = v[T][7] - mean;
        qq += (d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7);
      }
      const float rstd = rsqrtf((row_sum(qq) - A.ln_npad * (mean * mean)) * A.ln_inv_n + 1e-5f);
#pragma unroll
      for (int T = 0; T < KS; ++T) {
        const float4 ga0 = ld4(A.gamma + 32 * T + 4 * g), ga1 = ld4(A.gamma + 32 * T + 16 + 4 * g);
        const float4 be0 = ld4(A.beta + 32 * T + 4 * g), be1 = ld4(A.beta + 32 * T + 16 + 4 * g);
        v[T][0] = (v[T][0] - mean) * rstd * ga0.x + be0.x; v[T][1] = (v[T][1] - mean) * rstd * ga0.y + be0.y;
        v[T][2] = (v[T][2] - mean) * rstd * ga0.z + be0.z; v[T][3] = (v[T][3] - mean) * rstd * ga0.w + be0.w;
        v[T][4] = (v[T][4] - mean) * rstd * ga1.x + be1.x; v[T][5] = (v[T][5] - mean) * rstd * ga1.y + be1.y;
        v[T][6] = (v[T][6] - mean) * rstd * ga1.z + be1.z; v[T][7] = (v[T][7] - mean) * rstd * ga1.w + be1.w;
      }
    }
    float m0 = 0.f, m1 = 0.f;
#pragma unroll
    for (int T = 0; T < KS; ++T) {
      m0 = max3_abs(m0, v[T][0], v[T][1]);
      m1 = max3_abs(m1, v[T][2], v[T][3]);
      m0 = max3_abs(m0, v[T][4], v[T][5]);
      m1 = max3_abs(m1, v[T][6], v[T][7]);
    }
    const float sx = gfv_pow2_scale(row_max4(fmaxf(m0, m1)));
    if (A.gscale) {   // the group's scale = the smallest of its 16 rows' (tchain_kernel.h group_scale_out)
      const float sg = gfv_row16_min(sx);
      if (lane == 0) A.gscale[(size_t)(row0 >> 4) + w] = sg;
    }
    if (g == 0) sinv[w * 16 + j] = 1.0f / sx;
    gfv_uint4* dst = reinterpret_cast<gfv_uint4*>(lds + LY::FRAG + (size_t)w * KS * 2048) + lane;
#pragma unroll
    for (int T = 0; T < KS; ++T) {
      float e[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) e[i] = v[T][i] * sx;
      gfv_uint4 hi, lo;
      gfv_split8_t<BF>(e, hi, lo);
      dst[(2 * T) * 64] = hi;
      if (!LOWP) dst[(2 * T + 1) * 64] = lo;
    }
    if (EPI == 2) {
      // statistics of the LayerNorm's input row (the epilogue's waves own columns, not rows): mean, 1 / std
      float y[8][4];
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const float4 t = ld4(A.aux + mr * 128 + 16 * nt + 4 * g);
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
      if (g == 0) *reinterpret_cast<float2*>(stat + 2 * (w * 16 + j)) = make_float2(mean, rstd);
    }
  }
  if (CSR) load_w(0);
  l1s_barrier();

  // ---- products and epilogues: wave w, columns 32 w .. 32 w + 31 of every pass ----
  float dgam[2][4], dbet[2][4];   // (EPI 2) lane-private column sums over the tile's rows
  if (EPI == 2) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) dgam[n][r] = dbet[n][r] = 0.f;
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    floatx4 acc[TG][2];
#pragma unroll
    for (int q = 0; q < TG; ++q)
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[q][n] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int T = 0; T < KS; ++T) {
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        const gfv_f16x8* f = reinterpret_cast<const gfv_f16x8*>(lds + LY::FRAG + (size_t)(q * KS + T) * 2048) + lane;
        const gfv_f16x8 xh = f[0];
        if (!LOWP) {
          const gfv_f16x8 xl = f[64];
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            acc[q][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[n][T], xh, acc[q][n], 0, 0, 0);
            acc[q][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[n][T], xl, acc[q][n], 0, 0, 0);
          }
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[q][n] = gfv_mma_hh<BF>(wh[n][T], xh, acc[q][n]);
      }
    }
    if (p + 1 < NP) load_w(p + 1);   // the next pass's slice: in flight through this pass's epilogue
    const float* bp = A.bias[p];
    const float* resp = A.res[p];
    float* outp = A.out[p];
    if (EPI != 2) {
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        const int row = row0 + 16 * q + j;
        const bool live = q < ngt && row < A.M;
        const size_t rc = (size_t)min(row, A.M - 1);
        const float inv = sinv[q * 16 + j];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int col = c0 + 16 * n;
          float4 o = make_float4((acc[q][n][0] * inv) * invw, (acc[q][n][1] * inv) * invw, (acc[q][n][2] * inv) * invw,
                                 (acc[q][n][3] * inv) * invw);
          if (bp) {
            const float4 b = ld4(bp + col);
            o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
          }
          if (EPI == 1) {
            const float4 z = ld4(A.aux + rc * (128 * NP) + 128 * p + col);
            o.x *= gfv_dgelu(z.x); o.y *= gfv_dgelu(z.y); o.z *= gfv_dgelu(z.z); o.w *= gfv_dgelu(z.w);
          }
          if (resp) {
            const float4 r = ld4(resp + rc * A.res_ld[p] + col);
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
          }
          if (live) *reinterpret_cast<float4*>(outp + rc * A.out_ld[p] + col) = o;
        }
      }
    } else {
      // ---- LayerNorm backward of the rows (tchain_kernel.h ln_bwd): v = acc, xhat from the loader's statistics; the row sums
      // s1 = sum v gamma, s2 = sum v gamma xhat over the four waves through LDS ----
      float vv[TG][2][4], xh_[TG][2][4];
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        const int row = row0 + 16 * q + j;
        const bool live = q < ngt && row < A.M;
        const size_t rc = (size_t)min(row, A.M - 1);
        const float inv = sinv[q * 16 + j];
        const float2 st = *reinterpret_cast<const float2*>(stat + 2 * (q * 16 + j));
        const float livef = live ? 1.0f : 0.0f;   // rows past M must not reach the (dgamma, dbeta) sums
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int col = c0 + 16 * n;
          const float4 yv = ld4(A.aux + rc * 128 + col);
          const float4 ga = ld4(A.gamma + col);
          const float yy[4] = {yv.x, yv.y, yv.z, yv.w}, gm[4] = {ga.x, ga.y, ga.z, ga.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = ((acc[q][n][r] * inv) * invw) * livef;
            const float xhat = (yy[r] - st.x) * st.y;
            dgam[n][r] += v * xhat;
            dbet[n][r] += v;
            vv[q][n][r] = v * gm[r];
            xh_[q][n][r] = xhat;
            s1 += vv[q][n][r];
            s2 += vv[q][n][r] * xhat;
          }
        }
        s1 = row_sum(s1);
        s2 = row_sum(s2);
        if (g == 0) *reinterpret_cast<float2*>(part + ((q * 16 + j) * 4 + w) * 2) = make_float2(s1, s2);
      }
      l1s_barrier();
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        const int row = row0 + 16 * q + j;
        const bool live = q < ngt && row < A.M;
        const size_t rc = (size_t)min(row, A.M - 1);
        const float4* pp = reinterpret_cast<const float4*>(part + (q * 16 + j) * 8);
        const float4 p0 = pp[0], p1 = pp[1];
        const float mm1 = ((p0.x + p0.z) + (p1.x + p1.z)) * A.ln_inv_n, mm2 = ((p0.y + p0.w) + (p1.y + p1.w)) * A.ln_inv_n;
        const float rstd = stat[2 * (q * 16 + j) + 1];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int col = c0 + 16 * n;
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = rstd * (vv[q][n][r] - mm1 - xh_[q][n][r] * mm2);
          if (resp) {
            const float4 rr = ld4(resp + rc * A.res_ld[0] + col);
            o[0] += rr.x; o[1] += rr.y; o[2] += rr.z; o[3] += rr.w;
          }
          if (live) *reinterpret_cast<float4*>(outp + rc * A.out_ld[0] + col) = make_float4(o[0], o[1], o[2], o[3]);
        }
      }
      // (dgamma, dbeta) of this 64-row tile: over the 16 rows of a lane group by DPP; this wave owns its 32 columns outright
      if (A.ln_partial) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float dg = gfv_row16_sum(dgam[n][r]), db = gfv_row16_sum(dbet[n][r]);
            if (j == 0) {
              A.ln_partial[(size_t)tile * 256 + c0 + 16 * n + r] = dg;
              A.ln_partial[(size_t)tile * 256 + 128 + c0 + 16 * n + r] = db;
            }
          }
      }
    }
  }
  (void)status;
}

inline bool l1s_al16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

}  // namespace

// 1: launched; 0: not a launch of this family (lin1.hip asks here first).  Same contract as gfv_internal_lin1_try.
int gfv_internal_lin1s_try(const gfv_rowtile_args_t* a, int lowp, hipStream_t stream, int dry) {
  const int on = gfv_internal_limit(GFV_LIM_LIN1S_ON);
  const int max_m = gfv_internal_limit(GFV_LIM_LIN1S_MAX_M);
  if (!on || a->nlayers != 1 || a->M > max_m || a->M < 1 || (a->flags & (GFV_CHAIN_ROW_OWNER | GFV_CHAIN_COLUMN_OWNER))) return 0;
  const gfv_layer_t& L = a->layer[0];
  if (!L.Wh || !a->wmax || L.save || L.bias2 && !l1s_al16(L.bias2)) return 0;
  if (a->fin_presave || a->fin_stats || a->in_stats || a->dw_partial || a->gadd || a->padd || a->in_aux || a->out_nores) return 0;
  Lin1sArgs B{};
  B.img = L.Wh;
  B.wmax = a->wmax;
  B.M = a->M;
  {
    const int hs = gfv_hidden_size();
    const int h = (hs > 0 && hs < 128) ? hs : 128;
    B.ln_inv_n = 1.0f / (float)h;
    B.ln_npad = (float)(128 - h);
  }
  int* st = gfv_internal_status_ptr();
  const bool csr = (a->nseg == 2 && a->seg[0].csr_rowptr && a->seg[1].csr_rowptr) || (a->nseg == 1 && a->seg[0].csr_rowptr);
  int epi = 0, iop = 0, ks = 0, np = 0, tg = 2;
  if (csr) {
    // two segments in front of [256 -> 128] (the per-side scatter of the factored EdgeBlock's adjoint), or one in front of
    // [128 -> 256] (the neighbour sum in front of its node projection, two row-stacked blocks)
    const bool one = a->nseg == 1;
    if (L.op != GFV_OP_NONE || L.aux || L.bias || a->in_op != GFV_IN_NONE || a->fin_op != GFV_FIN_PLAIN ||
        L.K != (one ? 128 : 256) || L.N != (one ? 256 : 128))
      return 0;
    if (a->in_add || a->in_save || a->gscale || a->ln_partial || a->res[0] || a->res[1] || a->res[2]) return 0;
    if (!a->out[0] || (one ? !a->out[1] : a->out[1] != nullptr) || a->out[2] || (a->out_ld[0] & 3) || !l1s_al16(a->out[0])) return 0;
    if (one && ((a->out_ld[1] & 3) || !l1s_al16(a->out[1]))) return 0;
    for (int i = 0; i < a->nseg; ++i) {
      const gfv_seg_t& sg = a->seg[i];
      if (sg.width != 128 || !sg.idx || sg.csr_scale || (sg.ld & 3) || !l1s_al16(sg.ptr) || (sg.save && !l1s_al16(sg.save))) return 0;
      B.seg[i] = sg.ptr; B.seg_ld[i] = sg.ld; B.rowptr[i] = sg.csr_rowptr; B.col[i] = sg.idx; B.save[i] = sg.save;
    }
    B.out[0] = a->out[0]; B.out_ld[0] = a->out_ld[0];
    if (one) { B.out[1] = a->out[1]; B.out_ld[1] = a->out_ld[1]; }
    ks = one ? 4 : 8; np = one ? 2 : 1;
  } else {
    if (a->nseg < 1 || a->nseg > 2 || L.K != 128 * a->nseg || (L.N != 128 && L.N != 256) || (a->nseg == 2 && L.N != 128)) return 0;
    for (int i = 0; i < a->nseg; ++i) {
      const gfv_seg_t& s = a->seg[i];
      if (s.width != 128 || s.idx || s.csr_rowptr || s.csr_scale || s.save || (s.ld & 3) || !l1s_al16(s.ptr)) return 0;
      B.seg[i] = s.ptr; B.seg_ld[i] = s.ld;
    }
    if (a->fin_op == GFV_FIN_LNBWD) {
      if (L.op != GFV_OP_NONE || L.aux || L.bias || a->in_op != GFV_IN_NONE || a->nseg != 2 || L.N != 128) return 0;
      if (a->in_add || a->in_save || a->gscale || !a->fin_aux || !a->fin_gamma || !a->ln_partial) return 0;
      if (!l1s_al16(a->fin_aux) || !l1s_al16(a->fin_gamma) || !l1s_al16(a->ln_partial)) return 0;
      B.aux = a->fin_aux; B.gamma = a->fin_gamma; B.ln_partial = a->ln_partial;
      epi = 2; tg = 4;
    } else {
      if (a->fin_op != GFV_FIN_PLAIN || a->ln_partial) return 0;
      const bool dgelu = L.op == GFV_OP_MUL_DGELU;
      if (L.op != GFV_OP_NONE && !dgelu) return 0;
      if (dgelu ? (!L.aux || !l1s_al16(L.aux) || L.bias || L.bias2 || a->in_op != GFV_IN_NONE) : (L.aux != nullptr)) return 0;
      if (a->in_op != GFV_IN_NONE && a->in_op != GFV_IN_GELU && a->in_op != GFV_IN_LN) return 0;
      if (a->in_op == GFV_IN_LN && (a->nseg != 1 || !a->in_gamma || !a->in_beta || !l1s_al16(a->in_gamma) || !l1s_al16(a->in_beta) || a->in_add || a->in_save))
        return 0;
      if (a->gscale && (a->nseg != 1 || (dgelu && L.N == 128 && !a->res[0]))) return 0;   // (lin1.hip: the chain kernel also fills slot 1 there)
      if (a->in_add && (a->nseg != 1 || !l1s_al16(a->in_add))) return 0;
      if (a->in_save && (a->nseg != 1 || !l1s_al16(a->in_save))) return 0;
      B.in_add = a->in_add; B.in_save = a->in_save; B.gscale = a->gscale;
      B.gamma = a->in_gamma; B.beta = a->in_beta; B.aux = L.aux;
      epi = dgelu ? 1 : 0;
      iop = a->in_op == GFV_IN_GELU ? 1 : (a->in_op == GFV_IN_LN ? 2 : 0);
    }
    ks = 4 * a->nseg; np = L.N / 128;
    for (int p = 0; p < 3; ++p) {
      if (p < np) {
        if (!a->out[p] || (a->out_ld[p] & 3) || !l1s_al16(a->out[p])) return 0;
        if (a->res[p] && ((a->res_ld[p] & 3) || !l1s_al16(a->res[p]))) return 0;
        B.res[p] = a->res[p]; B.res_ld[p] = a->res_ld[p]; B.out[p] = a->out[p]; B.out_ld[p] = a->out_ld[p];
      } else if (a->out[p] || a->res[p]) {
        return 0;
      }
    }
    if (L.bias && !l1s_al16(L.bias)) return 0;
    B.bias[0] = L.bias;
    B.bias[1] = np > 1 ? (L.bias2 ? L.bias2 : (L.bias ? L.bias + 128 : nullptr)) : nullptr;
  }
  if (epi == 1 && ks == 8) return 0;   // (no GELU' epilogue behind a 256-deep input in this family)
  if (dry) return 1;
  const int tiles = (a->M + 16 * tg - 1) / (16 * tg);
  const dim3 grid(csr ? gfv_xcd_grid(tiles) : tiles), blk(256);
#define L1S_ONE(KS, NP, IOP, EPI, CSR, LP, TG) GFV_LAUNCH((lin1s_kernel<KS, NP, IOP, EPI, CSR, LP, TG>), grid, blk, 0, stream, B, st)
#define L1S_FORM(LP)                                                                 \
  do {                                                                               \
    if (csr && ks == 4) L1S_ONE(4, 2, 0, 0, true, LP, 2);                            \
    else if (csr) L1S_ONE(8, 1, 0, 0, true, LP, 2);                                  \
    else if (epi == 2) L1S_ONE(8, 1, 0, 2, false, LP, 4);                            \
    else if (epi == 1 && ks == 4 && np == 2) L1S_ONE(4, 2, 0, 1, false, LP, 2);      \
    else if (epi == 1 && ks == 4 && np == 1) L1S_ONE(4, 1, 0, 1, false, LP, 2);      \
    else if (ks == 8 && iop == 0) L1S_ONE(8, 1, 0, 0, false, LP, 2);                 \
    else if (ks == 8 && iop == 1) L1S_ONE(8, 1, 1, 0, false, LP, 2);                 \
    else if (np == 2 && iop == 0) L1S_ONE(4, 2, 0, 0, false, LP, 2);                 \
    else if (np == 2 && iop == 2) L1S_ONE(4, 2, 2, 0, false, LP, 2);                 \
    else if (np == 2 && iop == 1) L1S_ONE(4, 2, 1, 0, false, LP, 2);                 \
    else if (iop == 0) L1S_ONE(4, 1, 0, 0, false, LP, 2);                            \
    else if (iop == 1) L1S_ONE(4, 1, 1, 0, false, LP, 2);                            \
    else L1S_ONE(4, 1, 2, 0, false, LP, 2);                                          \
  } while (0)
  if (lowp == 2) L1S_FORM(2);
  else if (lowp) L1S_FORM(1);
  else L1S_FORM(0);
#undef L1S_FORM
#undef L1S_ONE
  return 1;
}
