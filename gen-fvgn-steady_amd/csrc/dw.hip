// Weight gradient of a Linear layer on fp32 MFMA (gfx950):  dW[n,k] = sum_m G[m,n] * A[m,k],  db[n] = sum_m G[m,n].
// Contract: include/gfv.h (gfv_linear_dw).
//
// Both operands have the contraction index m as their row index in memory, so tiles of 32 rows of G and A are
// staged row-major in LDS ([m][128], row stride 144 floats: the four k-groups of an MFMA operand fetch land on
// disjoint bank quarters) and each MFMA operand is one ds_read_b32 per lane.  A workgroup owns a 512-row slab of
// m and the full 128 x (<=128) output block of one input segment; its 4 waves own the 64x64 quadrants (16
// accumulator tiles each).  Slab partials go to a workspace and are summed by gfv_reduce_partials in a fixed
// order (no float atomics -> deterministic).
#include "gfv_common.h"
#include "gfv_prof.h"
#include "../../include/gfv.h"

namespace {

constexpr int RCH = 512;  // rows of m per workgroup
constexpr int SUB = 32;   // rows per staged sub-tile
constexpr int LDT = 144;

struct DwArgs {
  const float* G;
  int ldg, n_out;
  gfv_seg_t seg[3];
  int nseg;
  const float* in_add;
  int a_op;  // 0 none, 1 gelu, 2 layernorm(gamma,beta)
  const float* a_gamma;
  const float* a_beta;
  int M, Ktot;
  float* ws_dw;  // [chunks][n_out*Ktot]
  float* ws_db;  // [chunks][n_out] or NULL
};

__device__ __forceinline__ float4 ld4(const float* base, size_t row, int ld, int col, int width, bool vec) {
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

__global__ __launch_bounds__(256, 2) void linear_dw_kernel(const DwArgs A) {
  __shared__ __attribute__((aligned(16))) float Gs[SUB * LDT];
  __shared__ __attribute__((aligned(16))) float As[SUB * LDT];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nl = lane & 15, q = lane >> 4;
  const int wn = wave >> 1, wk = wave & 1;
  const int chunk = blockIdx.x, si = blockIdx.y;
  const gfv_seg_t seg = A.seg[si];
  int koff = 0;
  for (int i = 0; i < si; ++i) koff += A.seg[i].width;
  const int kpad = (seg.width + 15) & ~15;
  const int npad = (A.n_out + 15) & ~15;
  const bool gvec = ((A.ldg & 3) == 0) && ((A.n_out & 3) == 0);
  const bool avec = ((seg.ld & 3) == 0) && ((seg.width & 3) == 0);
  const int c4 = tid & 31, rg = tid >> 5;  // staging: row group 0..7, float4 column
  const int col = 4 * c4;
  float4 gam = make_float4(1.f, 1.f, 1.f, 1.f), bet = make_float4(0.f, 0.f, 0.f, 0.f);
  if (A.a_op == 2) {
    gam = *reinterpret_cast<const float4*>(A.a_gamma + col);
    bet = *reinterpret_cast<const float4*>(A.a_beta + col);
  }

  floatx4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  float4 dbacc = make_float4(0.f, 0.f, 0.f, 0.f);

  const int m_beg = chunk * RCH;
  const int m_end = min(m_beg + RCH, A.M);
  float4 greg[4], areg[4];

  auto load_sub = [&](int m0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int m = m0 + rg + 8 * p;
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f), a = g;
      if (m < m_end) {
        g = ld4(A.G, (size_t)m, A.ldg, col, A.n_out, gvec);
        const size_t srow = seg.idx ? (size_t)seg.idx[m] : (size_t)m;
        a = ld4(seg.ptr, srow, seg.ld, col, seg.width, avec);
        if (si == 0 && A.in_add) {
          const float4 b = ld4(A.in_add, srow, seg.ld, col, seg.width, avec);
          a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
      }
      if (A.a_op == 1) {
        a = make_float4(gfv_gelu(a.x), gfv_gelu(a.y), gfv_gelu(a.z), gfv_gelu(a.w));
      } else if (A.a_op == 2) {
        // row LayerNorm (eps 1e-5); the 32 lanes that share a row are an aligned half wave
        const float mean = gfv_half_sum((a.x + a.y) + (a.z + a.w)) * (1.0f / 128.0f);
        const float dx = a.x - mean, dy = a.y - mean, dz = a.z - mean, dw = a.w - mean;
        const float var = gfv_half_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.0f / 128.0f);
        const float rstd = rsqrtf(var + 1e-5f);
        a = make_float4(dx * rstd * gam.x + bet.x, dy * rstd * gam.y + bet.y, dz * rstd * gam.z + bet.z,
                        dw * rstd * gam.w + bet.w);
      }
      if (m >= m_end) a = make_float4(0.f, 0.f, 0.f, 0.f);
      greg[p] = g;
      areg[p] = a;
    }
  };

  if (m_beg < m_end) load_sub(m_beg);
  for (int m0 = m_beg; m0 < m_end; m0 += SUB) {
    __syncthreads();  // previous sub-tile fully consumed
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      *reinterpret_cast<float4*>(&Gs[(rg + 8 * p) * LDT + col]) = greg[p];
      *reinterpret_cast<float4*>(&As[(rg + 8 * p) * LDT + col]) = areg[p];
      dbacc.x += greg[p].x; dbacc.y += greg[p].y; dbacc.z += greg[p].z; dbacc.w += greg[p].w;
    }
    __syncthreads();
    if (m0 + SUB < m_end) load_sub(m0 + SUB);
#pragma unroll 2
    for (int ms = 0; ms < SUB / 4; ++ms) {
      const float* grow = &Gs[(4 * ms + q) * LDT + 64 * wn + nl];
      const float* arow = &As[(4 * ms + q) * LDT + 64 * wk + nl];
      float g[4], a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] = grow[16 * i];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = arow[16 * j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (64 * wn + 16 * i < npad && 64 * wk + 16 * j < kpad)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[i], a[j], acc[i][j], 0, 0, 0);
        }
    }
  }

  float* ws = A.ws_dw + (size_t)chunk * A.n_out * A.Ktot;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int n = 64 * wn + 16 * i + 4 * q + reg;
        const int k = 64 * wk + 16 * j + nl;
        if (n < A.n_out && k < seg.width) ws[(size_t)n * A.Ktot + koff + k] = acc[i][j][reg];
      }

  if (si == 0 && A.ws_db) {
    __syncthreads();
    *reinterpret_cast<float4*>(&Gs[rg * LDT + col]) = dbacc;
    __syncthreads();
    if (tid < A.n_out) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) s += Gs[r * LDT + tid];
      A.ws_db[(size_t)chunk * A.n_out + tid] = s;
    }
  }
}

}  // namespace

extern "C" int gfv_dw_chunks(int32_t M) { return (M + RCH - 1) / RCH; }

extern "C" size_t gfv_linear_dw_workspace_floats(int32_t M, int32_t n_out, int32_t K) {
  const size_t ch = (size_t)((M + RCH - 1) / RCH);
  return ch * (size_t)n_out * (size_t)K + ch * (size_t)n_out;
}

extern "C" int gfv_reduce_partials(const float*, int32_t, int32_t, float*, int32_t, void*);

extern "C" int gfv_linear_dw_ex(const float* G, int32_t ldg, int32_t n_out, const gfv_seg_t* segs, int32_t nseg,
                                const float* in_add, int32_t a_op, const float* a_gamma, const float* a_beta, int32_t M,
                                float* dW, int32_t ld_dw_unused, float* db, float* workspace, int32_t accumulate,
                                void* stream) {
  (void)ld_dw_unused;
  if (nseg < 1 || nseg > 3 || n_out < 1 || n_out > 128 || M < 0) return GFV_ERR_ARG;
  DwArgs a;
  a.G = G; a.ldg = ldg; a.n_out = n_out; a.nseg = nseg; a.in_add = in_add; a.a_op = a_op;
  a.a_gamma = a_gamma; a.a_beta = a_beta; a.M = M;
  int K = 0;
  for (int i = 0; i < nseg; ++i) {
    if (segs[i].width < 1 || segs[i].width > 128) return GFV_ERR_ARG;
    a.seg[i] = segs[i];
    K += segs[i].width;
  }
  if (a_op == 2 && (nseg != 1 || segs[0].width != 128)) return GFV_ERR_ARG;
  a.Ktot = K;
  const int chunks = (M + RCH - 1) / RCH;
  if (chunks == 0) return GFV_OK;
  a.ws_dw = workspace;
  a.ws_db = db ? workspace + (size_t)chunks * n_out * K : nullptr;
  void* tok = gfv_prof_begin(GFV_K_DW, 2.0 * M * (double)n_out * K,
                             4.0 * M * ((double)n_out + K) + 4.0 * (double)chunks * n_out * K, (hipStream_t)stream);
  hipLaunchKernelGGL(linear_dw_kernel, dim3(chunks, nseg), dim3(256), 0, (hipStream_t)stream, a);
  gfv_prof_end(tok, (hipStream_t)stream);
  GFV_CHECK_LAUNCH();
  int rc = gfv_reduce_partials(a.ws_dw, chunks, n_out * K, dW, accumulate, stream);
  if (rc) return rc;
  if (db) rc = gfv_reduce_partials(a.ws_db, chunks, n_out, db, accumulate, stream);
  return rc;
}

extern "C" int gfv_linear_dw(const float* G, int32_t ldg, int32_t n_out, const gfv_seg_t* segs, int32_t nseg,
                             const float* in_add, int32_t a_gelu, int32_t M, float* dW, float* db, float* workspace,
                             int32_t accumulate, void* stream) {
  return gfv_linear_dw_ex(G, ldg, n_out, segs, nseg, in_add, a_gelu ? 1 : 0, nullptr, nullptr, M, dW, 0, db,
                          workspace, accumulate, stream);
}
