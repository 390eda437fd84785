// gfv-build-flags: -fno-slp-vectorize
// Column-owner SMALL-TILE backward of the 3-layer MLPs (round 5): the kernel family behind gfv_rowtile_chain (contract:
// include/gfv.h) for the dX chain of build_mlp (EPD.py:10-33: Linear GELU Linear GELU Linear LayerNorm) inside the NodeBlock /
// EdgeBlock (blocks.py:54,101-111) and the two encoders (EPD.py:92-119) when the launch is SHORT, in the split-fp16 product forms:
//   LayerNorm backward -> [W3^T, x gelu'(z2)] -> [W2^T, x gelu'(z1)] -> [W1^T] (+ residual)
// with g3, gz2, gz1 left in HBM for the weight-gradient launch of the side queue (dw.hip) and the per-tile (dgamma, dbeta) sums in
// ln_partial.
//
// Why: profiles/r05_timeline_cavity.txt.  On a 5 k-cell mesh every backward launch of the persistent column-owner kernel
// (colchain_kernel.h: 64-row tiles, 8 waves at 253 registers, the fused weight gradients' 72 accumulator registers, a 136 KB
// partial block per workgroup) takes 25 - 30 us for ONE tile per workgroup, and that kernel needs a whole CU to itself - it waits
// for the side queue's workgroups to leave.  The row-owner chain (tchain_kernel.h) takes 22 - 41 us for the same launch.  This is
// the backward counterpart of cfwd.hip: a workgroup = ONE tile of 32 rows (TG = 2 groups of 16) on 8 waves,
// wave w owns output COLUMNS 16 w .. 16 w + 15 of every layer; the layers' weight slices come straight from L2 into registers a
// phase ahead; activations cross waves as MFMA B fragments in LDS; 4 barriers per tile; at most 128 registers, 34 KB of LDS: two
// tiles (or a tile and the side queue's workgroups) share a CU.  No fused weight gradients - a 32-row tile would leave a 136 KB
// partial block per 32 rows - so the three weight gradients of the MLP run as ONE launch on the side queue, as they did before
// round 3 fused them for the big launches.
//
// The arithmetic is colchain_bwd_kernel's, phase by phase (P0 / P0b / P3 / P2 / P1 there), including its scales: the g3 fragments
// carry ONE power-of-two scale per tile taken from a bound before the barrier, gz2 / gz1 step down from it by the layers'
// guaranteed growth bounds (row 1-norms of the weight images).  The LayerNorm width is the launch's (gfv_rowtile_args_t.hidden):
// a narrower model's padded columns carry gamma = 0 and add nothing to the row sums.
// ln_partial: one row [dgamma 128 | dbeta 128] per 32 ROWS of the launch - gfv_rowtile_ln_rows(M)
// rows; every other kernel family writes one per 64 rows (include/gfv.h; gfv_rowtile_last_path() & 128 tells).
#include <cstdlib>

#include "colchain_kernel.h"

#include "gfv_limits.h"
int* gfv_internal_status_ptr();

namespace {

template <int TG>
struct CwLds {
  static constexpr int BUF = TG * 8192;
  static constexpr int B0 = 0, B1 = BUF;
  static constexpr int PART = 2 * BUF;                     // float2 [TG][16][8]: per-wave partial (s1, s2) of a row
  static constexpr int SMAX = PART + TG * 16 * 8 * 8;      // float [8]: per-wave bound of max |gg| rstd over the tile
  static constexpr int NRM = SMAX + 32;                    // float [2][8]: per-wave largest row 1-norm of the first two images
  static constexpr int TOTAL = NRM + 64;
};

// this wave's slice of a 128-deep transposed image: n-tile `nt` of the 128-column pass at byte offset `pass`, 4 k-groups x (hi, lo)
template <int LOWP>
__device__ __forceinline__ void cw_load_w(cb_rsrc wb, int nt, int lane, gfv_f16x8 (&wh)[4], gfv_f16x8 (&wl)[4], int pass = 0) {
  const int woff = pass + (nt * 128 + lane) * 16;
#pragma unroll
  for (int T = 0; T < 4; ++T) {
    wh[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(wb, woff + T * 16384, 0, 0));
    if (!LOWP) wl[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(wb, woff + T * 16384 + 1024, 0, 0));
    else wl[T] = wh[T];
  }
}
// largest row 1-norm of the slice over the wave's 16 rows (colchain_kernel.h: the growth bound of a chain layer)
template <int LOWP>
__device__ __forceinline__ float cw_slice_norm(const gfv_f16x8 (&wh)[4], const gfv_f16x8 (&wl)[4]) {
  float acc = 0.f;
#pragma unroll
  for (int T = 0; T < 4; ++T) {
    if (LOWP == 2) {
      const gfv_bf16x8 hb = __builtin_bit_cast(gfv_bf16x8, wh[T]);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += fabsf((float)hb[e]);
    } else if (LOWP == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += fabsf((float)wh[T][e]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += fabsf((float)wh[T][e] + (float)wl[T][e]);
    }
  }
  return gfv_wave_max(row_sum(acc));
}

// GADD: the gathered addend [gadd[s] | gadd[r]] of the incoming gradient; OUT2: a 192-wide last layer (NodeBlock: out[1] [M, 64]
// from the image's second pass, waves 0 .. 3); NOOUT: two layers, out[0] receives gz1 (the encoders)
template <int TG, int LOWP, bool GADD, bool OUT2, bool NOOUT>
__global__ __launch_bounds__(512, 4) void cbwd_kernel(const gfv_rowtile_args_t A, int* status) {
  static_assert(TG == 2, "32-row tiles: the only form that is built and tested");
  using LY = CwLds<TG>;
  constexpr bool BF = LOWP == 2;
  constexpr int NP = TG / 2;
  __shared__ __attribute__((aligned(16))) char lds[LY::TOTAL];
  char* b0 = lds + LY::B0;
  char* b1 = lds + LY::B1;
  float* part = reinterpret_cast<float*>(lds + LY::PART);
  float* smax = reinterpret_cast<float*>(lds + LY::SMAX);
  float* nrm = reinterpret_cast<float*>(lds + LY::NRM);

  CcCtx c;
  c.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  c.lane = threadIdx.x & 63;
  c.j = c.lane & 15;
  c.g = c.lane >> 4;
  c.col0 = 16 * c.w + 4 * c.g;
  c.M = A.M;
  c.mabs = 0.f;
  c.row0 = (int)blockIdx.x * (16 * TG);
  if (c.row0 >= A.M) return;
  c.ngt = min(TG, (A.M - c.row0 + 15) >> 4);
  const int np = (c.ngt + 1) >> 1;

  const size_t rows128 = (size_t)A.M * 512;
  const cb_rsrc bdy = cb_buf(A.seg[0].ptr, rows128), by3 = cb_buf(A.in_aux, rows128), bst = cb_buf(A.in_stats, (size_t)A.M * 8),
                bz2 = cb_buf(A.layer[0].aux, rows128), bz1 = cb_buf(A.layer[1].aux, rows128);
  const cb_rsrc w0 = cb_buf(A.layer[0].Wh, 65536), w1 = cb_buf(A.layer[1].Wh, 65536),
                w2 = cb_buf(NOOUT ? nullptr : A.layer[2].Wh, OUT2 ? 131072 : 65536);

  int offL[TG], offS[TG];   // byte offsets of this lane's 16 bytes: loads (clamped rows), stores (dead rows: out of bounds)
#pragma unroll
  for (int q = 0; q < TG; ++q) {
    const int row = c.row0 + 16 * q + c.j;
    offL[q] = min(row, c.M - 1) * 512 + c.col0 * 4;
    offS[q] = (q < c.ngt && row < c.M) ? row * 512 + c.col0 * 4 : CB_OFF_DEAD;
  }
  // ---- everything the first two phases read goes out now ----
  int gidx[GADD ? TG : 1];
  if (GADD) {
    const cb_rsrc bgi = cb_buf(c.w < 4 ? A.gadd_s : A.gadd_r, (size_t)A.M * 4);
#pragma unroll
    for (int q = 0; q < TG; ++q) gidx[q] = __builtin_amdgcn_raw_buffer_load_b32(bgi, min(c.row0 + 16 * q + c.j, c.M - 1) * 4, 0, 0);
  }
  float4 dy[TG], yv[TG];
  float2 st[TG];
#pragma unroll
  for (int q = 0; q < TG; ++q) {
    dy[q] = cb_ld4(bdy, offL[q]);
    yv[q] = cb_ld4(by3, offL[q]);
    st[q] = cb_ld2(bst, min(c.row0 + 16 * q + c.j, c.M - 1) * 8);
  }
  gfv_f16x8 wh[4], wl[4];
  cw_load_w<LOWP>(w0, c.w, c.lane, wh, wl);
  if (GADD) {
    const cb_rsrc bga = cb_buf(A.gadd, 0x7fffffe0ull);
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const float4 t = cb_ld4(bga, gidx[q] * 256 + (c.col0 & 63) * 4);
      dy[q].x += t.x; dy[q].y += t.y; dy[q].z += t.z; dy[q].w += t.w;
    }
  }
  if (A.in_add) {
    const cb_rsrc ia = cb_buf(A.in_add, rows128);
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const float4 t = cb_ld4(ia, offL[q]);
      dy[q].x += t.x; dy[q].y += t.y; dy[q].z += t.z; dy[q].w += t.w;
    }
  }
  float4 zq[TG];
#pragma unroll
  for (int q = 0; q < TG; ++q) zq[q] = cb_ld4(bz2, offL[q]);
  c.invw = 1.0f / gfv_pow2_scale(*A.wmax);
  const float4 gam = ld4(A.in_gamma + c.col0);
  const float inv_n = ln_width(A.hidden).inv_n;

  // ---- P0: LayerNorm backward, first half: gg = dy gamma, xhat, the row sums as per-wave partials, (dgamma, dbeta) ----
  float gg[TG][4], xh[TG][4], rs[TG];
  {
    float dgam[4] = {0.f, 0.f, 0.f, 0.f}, dbet[4] = {0.f, 0.f, 0.f, 0.f};
    float bmax = 0.f;
    float* part_j = part + (c.j * 8 + c.w) * 2;
    const float gm[4] = {gam.x, gam.y, gam.z, gam.w};
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const float lf = offS[q] != CB_OFF_DEAD ? 1.0f : 0.0f;   // rows past M / dead groups must not reach any sum over rows
      const float d[4] = {dy[q].x, dy[q].y, dy[q].z, dy[q].w};
      const float y[4] = {yv[q].x, yv[q].y, yv[q].z, yv[q].w};
      const float mean = st[q].x, rstd = st[q].y;
      rs[q] = rstd;
      float s1 = 0.f, s2 = 0.f, am = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = 2 * h;
        const gfv_f2 dd = gfv_f2{d[r], d[r + 1]} * gfv_splat2(lf);
        const gfv_f2 xx = (gfv_f2{y[r], y[r + 1]} - gfv_splat2(mean)) * gfv_splat2(rstd);
        const gfv_f2 dx = dd * xx;
        const gfv_f2 g2 = dd * gfv_f2{gm[r], gm[r + 1]};
        const gfv_f2 gx = g2 * xx;
        dgam[r] += dx.x; dgam[r + 1] += dx.y;
        dbet[r] += dd.x; dbet[r + 1] += dd.y;
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
    const float n0 = cw_slice_norm<LOWP>(wh, wl);
    if (c.lane == 0) { smax[c.w] = bmax; nrm[c.w] = n0; }
    // (dgamma, dbeta) of the tile: this wave owns its columns - a sum over the 16 lanes of a DPP row, no cross-wave step
    if (A.ln_partial) {
      float* lp = A.ln_partial + (size_t)blockIdx.x * 256 + c.col0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dgam[r] = gfv_row16_sum(dgam[r]);
        dbet[r] = gfv_row16_sum(dbet[r]);
      }
      if (c.j == 0) {
        st4(lp, dgam);
        st4(lp + 128, dbet);
      }
    }
  }
  cc_barrier();
  // ---- P0b: g3 and its fragments ----
  // (every load goes out a phase before its use: the second layer's slice and its saved pre-activations now)
  gfv_f16x8 xh1[4], xl1[4];
  cw_load_w<LOWP>(w1, c.w, c.lane, xh1, xl1);
  float s3, s2s;
  {
    const float4 ma = *reinterpret_cast<const float4*>(smax), mb = *reinterpret_cast<const float4*>(smax + 4);
    const float mx = fmaxf(fmaxf(fmaxf(ma.x, ma.y), fmaxf(ma.z, ma.w)), fmaxf(fmaxf(mb.x, mb.y), fmaxf(mb.z, mb.w)));
    // |g3| <= rstd (|gg| + |m1| + |xhat| |m2|) <= rstd max|gg| (2 + sqrt(127)): the bound, not the maximum, sets the scale
    s3 = gfv_pow2_scale(mx * 13.5f) * 2.0f;
    const float4 na = *reinterpret_cast<const float4*>(nrm), nb = *reinterpret_cast<const float4*>(nrm + 4);
    const float m0 = fmaxf(fmaxf(fmaxf(na.x, na.y), fmaxf(na.z, na.w)), fmaxf(fmaxf(nb.x, nb.y), fmaxf(nb.z, nb.w)));
    s2s = s3 * (1.0f / gfv_pow2_ceil(1.13f * m0 * c.invw));
    const cb_rsrc bsv = cb_buf(A.in_save, rows128);
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const float4* pp = reinterpret_cast<const float4*>(part + (q * 16 + c.j) * 16);
      const float4 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];   // (s1, s2) x 8 waves
      const float m1 = (((p0.x + p0.z) + (p1.x + p1.z)) + ((p2.x + p2.z) + (p3.x + p3.z))) * inv_n;
      const float m2 = (((p0.y + p0.w) + (p1.y + p1.w)) + ((p2.y + p2.w) + (p3.y + p3.w))) * inv_n;
      float g3[4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = 2 * h;
        const gfv_f2 t = gfv_splat2(rs[q]) * ((gfv_f2{gg[q][r], gg[q][r + 1]} - gfv_splat2(m1)) - gfv_f2{xh[q][r], xh[q][r + 1]} * gfv_splat2(m2));
        g3[r] = t.x; g3[r + 1] = t.y;
      }
      cb_st4(bsv, offS[q], g3);   // (no in_save: zero records, the stores are dropped)
      const float mq = max3_abs(max3_abs(0.f, g3[0], g3[1]), g3[2], g3[3]) * s3;
      c.mabs = fmaxf(c.mabs, offS[q] != CB_OFF_DEAD ? mq : 0.f);
      cc_put_frag<BF>(b0, q, c, g3, s3);
    }
  }
  float4 zq1[TG];
#pragma unroll
  for (int q = 0; q < TG; ++q) zq1[q] = cb_ld4(bz1, offL[q]);
  cc_barrier();
  // ---- P3: chain layer 0 (b0 -> gz2 in b1) ----
  float4 rr[NOOUT ? 1 : TG];
  {
    const float inv_in = 1.0f / s3;
    const cb_rsrc bs0 = cb_buf(A.layer[0].save, rows128);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (p >= np) break;
      floatx4 a0, a1;
      cc_mma_pair<4, LOWP, true>(b0, p, wh, wl, c.lane, a0, a1);
      float v0[4], v1[4];
      const float4 za = zq[2 * p], zb = zq[2 * p + 1];
      const float d0[4] = {gfv_dgelu(za.x), gfv_dgelu(za.y), gfv_dgelu(za.z), gfv_dgelu(za.w)};
      const float d1[4] = {gfv_dgelu(zb.x), gfv_dgelu(zb.y), gfv_dgelu(zb.z), gfv_dgelu(zb.w)};
      cb_hidden_bwd_dg<BF>(c, 2 * p, a0, inv_in, d0, s2s, b1, v0);
      cb_hidden_bwd_dg<BF>(c, 2 * p + 1, a1, inv_in, d1, s2s, b1, v1);
      cb_st4(bs0, offS[2 * p], v0);
      cb_st4(bs0, offS[2 * p + 1], v1);
    }
  }
  {
    const float n1 = cw_slice_norm<LOWP>(xh1, xl1);
    if (c.lane == 0) nrm[8 + c.w] = n1;
  }
  if constexpr (!NOOUT) {
    // the last layer's slice (into the registers the first layer's products have left) and the residual rows
    cw_load_w<LOWP>(w2, c.w, c.lane, wh, wl);
    const cb_rsrc bres = cb_buf(A.res[0], rows128);
#pragma unroll
    for (int q = 0; q < TG; ++q) rr[q] = cb_ld4(bres, offL[q]);
  }
  cc_barrier();
  // ---- P2: chain layer 1 (b1 -> gz1 in b0) ----
  float s1s;
  {
    const float4 na = *reinterpret_cast<const float4*>(nrm + 8), nb = *reinterpret_cast<const float4*>(nrm + 12);
    const float m1 = fmaxf(fmaxf(fmaxf(na.x, na.y), fmaxf(na.z, na.w)), fmaxf(fmaxf(nb.x, nb.y), fmaxf(nb.z, nb.w)));
    s1s = s2s * (1.0f / gfv_pow2_ceil(1.13f * m1 * c.invw));
    // per-16-row scales of the rows this launch leaves for the weight-gradient launch (gfv_rowtile_args_t.gscale): the tile's
    // fragment scales, a quarter of them (a slab scale s wants s max|v| <= 2^14, the fragments allow 2^16)
    if (A.gscale && c.w == 0 && c.lane < c.ngt) {
      const size_t grp = (size_t)(c.row0 >> 4) + c.lane;
      A.gscale[grp] = s3 * 0.25f;
      A.gscale[(size_t)A.gscale_ld + grp] = s2s * 0.25f;
      A.gscale[2 * (size_t)A.gscale_ld + grp] = s1s * 0.25f;
    }
    const float inv_in = 1.0f / s2s;
    const cb_rsrc bs1 = cb_buf(NOOUT ? A.out[0] : A.layer[1].save, rows128);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (p >= np) break;
      floatx4 a0, a1;
      cc_mma_pair<4, LOWP, true>(b1, p, xh1, xl1, c.lane, a0, a1);
      if (OUT2 && p == NP - 1 && c.w < 4) cw_load_w<LOWP>(w2, c.w, c.lane, xh1, xl1, 65536);   // (wave-uniform) the image's second pass, n-tile w
      float v0[4], v1[4];
      const float4 za = zq1[2 * p], zb = zq1[2 * p + 1];
      const float d0[4] = {gfv_dgelu(za.x), gfv_dgelu(za.y), gfv_dgelu(za.z), gfv_dgelu(za.w)};
      const float d1[4] = {gfv_dgelu(zb.x), gfv_dgelu(zb.y), gfv_dgelu(zb.z), gfv_dgelu(zb.w)};
      cb_hidden_bwd_dg<BF>(c, 2 * p, a0, inv_in, d0, s1s, b0, v0);
      cb_hidden_bwd_dg<BF>(c, 2 * p + 1, a1, inv_in, d1, s1s, b0, v1);
      cb_st4(bs1, offS[2 * p], v0);
      cb_st4(bs1, offS[2 * p + 1], v1);
    }
  }
  if constexpr (!NOOUT) {
    const cb_rsrc bout = cb_buf(A.out[0], rows128);
    cc_barrier();
    // ---- P1: chain layer 2 (b0 -> the input gradient) ----
    const float inv_in = 1.0f / s1s;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (p >= np) break;
      floatx4 a0, a1;
      cc_mma_pair<4, LOWP, true>(b0, p, wh, wl, c.lane, a0, a1);
      const floatx4 o0 = (a0 * inv_in) * c.invw + floatx4{rr[2 * p].x, rr[2 * p].y, rr[2 * p].z, rr[2 * p].w};
      const floatx4 o1 = (a1 * inv_in) * c.invw + floatx4{rr[2 * p + 1].x, rr[2 * p + 1].y, rr[2 * p + 1].z, rr[2 * p + 1].w};
      cb_st4v(bout, offS[2 * p], o0);
      cb_st4v(bout, offS[2 * p + 1], o1);
    }
    if constexpr (OUT2) {
      if (c.w < 4) {
        const cb_rsrc out2 = cb_buf(A.out[1], (size_t)A.M * 256);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          if (p >= np) break;
          floatx4 e0, e1;
          cc_mma_pair<4, LOWP, true>(b0, p, xh1, xl1, c.lane, e0, e1);
          const floatx4 o0 = (e0 * inv_in) * c.invw, o1 = (e1 * inv_in) * c.invw;
          const int ra = c.row0 + 32 * p + c.j, rb = ra + 16;
          cb_st4v(out2, (2 * p < c.ngt && ra < c.M) ? ra * 256 + c.col0 * 4 : CB_OFF_DEAD, o0);
          cb_st4v(out2, (2 * p + 1 < c.ngt && rb < c.M) ? rb * 256 + c.col0 * 4 : CB_OFF_DEAD, o1);
        }
      }
    }
  }
  if (c.mabs > 60000.0f) atomicOr(status, 2);   // GFV_FLAG_CHAIN_RANGE
}

inline bool cw_al16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

// 32-row tiles only.  (A 64-row instantiation was timed - slower at every size, profiles/r05_cfwd.txt run 20 / 21 - and then failed a
// model-level check that reached it through the environment, 1.4e-3 on the gradients of a 50 k-cell mesh: it had no kernel test of
// its own.  It is gone; the 32-row form is tested up to 41 003 rows and through the whole model at every size.)
template <bool GADD, bool OUT2, bool NOOUT>
void cw_launch(const gfv_rowtile_args_t& a, int lowp, hipStream_t stream) {
  constexpr int TG = 2;
  int* st = gfv_internal_status_ptr();
  const dim3 grid((a.M + 16 * TG - 1) / (16 * TG)), blk(512);
  if (lowp == 2) GFV_LAUNCH((cbwd_kernel<TG, 2, GADD, OUT2, NOOUT>), grid, blk, 0, stream, a, st);
  else if (lowp) GFV_LAUNCH((cbwd_kernel<TG, 1, GADD, OUT2, NOOUT>), grid, blk, 0, stream, a, st);
  else GFV_LAUNCH((cbwd_kernel<TG, 0, GADD, OUT2, NOOUT>), grid, blk, 0, stream, a, st);
}

}  // namespace

// rows of gfv_rowtile_args_t.ln_partial a caller provides for a launch over M rows (one per 32 rows: what this family fills;
// the other families fill the first ceil(M / 64))
extern "C" int gfv_rowtile_ln_rows(int32_t M) { return (M + 31) / 32; }

// 1: launched; 0: not a launch of this family.  lowp: 0 three products, 1 / 2 the single-product forms.  dry != 0: only tell
// whether the launch would be taken.  `a` carries `hidden` (the launcher of rowtile.hip fills it in).
int gfv_internal_cbwd_try(const gfv_rowtile_args_t* a, int lowp, hipStream_t stream, int dry) {
  // (the dispatch limits: gfv_limits.h - environment once, gfv_set_limit afterwards: the tests move the row limit to reach the
  // kernel at sizes the default leaves to the persistent backward)
  const int on = gfv_internal_limit(GFV_LIM_CBWD_ON);
  const int max_m = gfv_internal_limit(GFV_LIM_CBWD_MAX_M);
  if (!gfv_internal_status_ptr()) return 0;   // (the kernels raise their range flag there)
  if (!on || a->M > max_m || a->M < 1 || (a->flags & (GFV_CHAIN_ROW_OWNER | GFV_CHAIN_COLUMN_OWNER))) return 0;
  const bool noout = a->nlayers == 2;
  if ((a->nlayers != 3 && !noout) || a->in_op != GFV_IN_LNBWD || a->fin_op != GFV_FIN_PLAIN || a->nseg != 1) return 0;
  if (a->dw_partial || a->dw_in || a->rc_Wh[0] || a->rc_Wh[1] || a->padd || a->fin_stats || a->fin_presave || a->out_nores || a->fin_aux) return 0;
  if (!a->in_stats || (reinterpret_cast<size_t>(a->in_stats) & 7) || !a->in_aux || !cw_al16(a->in_aux) || !a->in_gamma || !cw_al16(a->in_gamma) || !a->wmax)
    return 0;
  const gfv_seg_t& s = a->seg[0];
  if (s.width != 128 || s.ld != 128 || s.idx || s.csr_rowptr || s.csr_scale || s.save || !cw_al16(s.ptr)) return 0;
  if (a->M > (1 << 22)) return 0;
  for (int l = 0; l < a->nlayers; ++l) {
    const gfv_layer_t& L = a->layer[l];
    if (!L.Wh || L.K != 128 || L.bias || L.bias2 || (L.save && !cw_al16(L.save))) return 0;
    if (l < 2 && (L.N != 128 || L.op != GFV_OP_MUL_DGELU || !L.aux || !cw_al16(L.aux))) return 0;
    if (l == 2 && ((L.N != 128 && L.N != 192) || L.op != GFV_OP_NONE || L.aux || L.save)) return 0;
  }
  if (noout && a->layer[1].save) return 0;   // (out[0] receives gz1)
  const bool out2 = !noout && a->layer[2].N == 192;
  if (!a->out[0] || a->out_ld[0] != 128 || !cw_al16(a->out[0]) || a->out[2] || a->res[1] || a->res[2]) return 0;
  if (out2 ? (!a->out[1] || a->out_ld[1] != 64 || !cw_al16(a->out[1])) : a->out[1] != nullptr) return 0;
  if (a->res[0] && (noout || a->res_ld[0] != 128 || !cw_al16(a->res[0]))) return 0;
  if (a->in_add && !cw_al16(a->in_add)) return 0;
  if (a->in_save && !cw_al16(a->in_save)) return 0;
  if (a->gadd && (!cw_al16(a->gadd) || !a->gadd_s || !a->gadd_r || out2 || noout)) return 0;
  if (a->ln_partial && !cw_al16(a->ln_partial)) return 0;
  if (dry) return 1;
  if (noout) cw_launch<false, false, true>(*a, lowp, stream);
  else if (out2) cw_launch<false, true, false>(*a, lowp, stream);
  else if (a->gadd) cw_launch<true, false, false>(*a, lowp, stream);
  else cw_launch<false, false, false>(*a, lowp, stream);
  return 1;
}
