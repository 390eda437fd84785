// Finite-volume discretisation kernels (gfx950): WLSQ gradient reconstruction, node->face / node->cell
// interpolation, conserved-form flux assembly, per-graph residual norms, cell->node smoothing, and the adjoint of
// each.  Contract: include/gfv.h.  Reference: FVMmodel/FVdiscretization/{FVscheme,FVgrad,FVInterpolation}.py,
// formulas restated in SURVEY.md 8(a-16).
//
// All of this is HBM/L2-bound index work on 7-channel fields: every kernel is a CSR-ordered gather (node-, face- or
// cell-parallel) with no atomics, so results are deterministic; plans (CSR tables, permuted moment vectors,
// normalised 5x5 matrices) are built once per mesh batch.
#include "gfv_common.h"
#include "gfv_prof.h"
#include "../../include/gfv.h"

namespace {

enum { NT_NORMAL = 0, NT_INFLOW = 1, NT_OUTFLOW = 2, NT_WALL = 3, NT_PRESS = 4, NT_INWALL = 5 };

// ---- phi = [uvp_new | uv_hat | uv_old] from the decoder output ---------------------------------------------------
// uvp_new = BC(10*tanh(dec/10))  (importer.py:187-189), uv_hat per integrator (importer.py:192-201)
__global__ __launch_bounds__(256) void phi_fwd_kernel(const float* __restrict__ dec, const float* __restrict__ y,
                                                      const int* __restrict__ node_type, const float* __restrict__ uv_old,
                                                      float* __restrict__ phi, int N, int mode) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int nt = node_type[i];
  float u = tanhf(dec[3 * i] / 10.f) * 10.f, v = tanhf(dec[3 * i + 1] / 10.f) * 10.f, p = tanhf(dec[3 * i + 2] / 10.f) * 10.f;
  if (nt == NT_WALL || nt == NT_INFLOW || nt == NT_PRESS || nt == NT_INWALL) { u = y[2 * i]; v = y[2 * i + 1]; }
  if (nt == NT_PRESS) p = 0.f;
  const float uo = uv_old[2 * i], vo = uv_old[2 * i + 1];
  float uh, vh;
  if (mode == 0) { uh = uo; vh = vo; }
  else if (mode == 1) { uh = u; vh = v; }
  else { uh = (uo + u) / 2.0f; vh = (vo + v) / 2.0f; }
  float4* o = reinterpret_cast<float4*>(phi + (size_t)i * 8);
  o[0] = make_float4(u, v, p, uh);
  o[1] = make_float4(vh, uo, vo, 0.f);
}

__global__ __launch_bounds__(256) void phi_bwd_kernel(const float* __restrict__ gphi, const float* __restrict__ dec,
                                                      const int* __restrict__ node_type, float* __restrict__ gdec, int N,
                                                      int mode) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float4 a = *reinterpret_cast<const float4*>(gphi + (size_t)i * 8);
  const float4 b = *reinterpret_cast<const float4*>(gphi + (size_t)i * 8 + 4);
  const float ch = (mode == 0) ? 0.f : (mode == 1 ? 1.f : 0.5f);
  float gu = a.x + ch * a.w, gv = a.y + ch * b.x, gp = a.z;
  const int nt = node_type[i];
  if (nt == NT_WALL || nt == NT_INFLOW || nt == NT_PRESS || nt == NT_INWALL) { gu = 0.f; gv = 0.f; }
  if (nt == NT_PRESS) gp = 0.f;
  const float t0 = tanhf(dec[3 * i] / 10.f), t1 = tanhf(dec[3 * i + 1] / 10.f), t2 = tanhf(dec[3 * i + 2] / 10.f);
  gdec[3 * i] = gu * (1.f - t0 * t0);
  gdec[3 * i + 1] = gv * (1.f - t1 * t1);
  gdec[3 * i + 2] = gp * (1.f - t2 * t2);
}

// ---- M x M LU with partial pivoting (LAPACK getf2 order), solve and transpose-solve; M = Taylor terms of the WLSQ
// reconstruction order: 2 (1st), 5 (2nd, the default), 9 (3rd), 14 (4th) (FVorder.py:23-72) ------------------------------
//
// The factorisation and the two triangular solves run in DOUBLE.  The moment matrices of stretched cells are badly
// conditioned (boundary-layer nodes of mesh_example/airfoil_L=1: cond(A_n) = 4e5), where ANY fp32 elimination returns
// rounding noise of size cond * 2^-24 - the reference's LAPACK call, an FMA build of the same loop and this kernel's former
// fp32 loop land 0.009, 0.17 and 1.7 away from the exact gradient 1833 at such a node (profiles/r03_wlsq_conditioning.txt).
// Solving the system exactly (right-hand side accumulated in double from the fp32 data, A normalised in double) puts the
// result in the middle of that noise ball: its distance to the reference's value is the reference's own rounding error.
typedef double lu_t;
template <int M>
struct LU {
  lu_t a[M][M];
  int piv[M];
};

template <int M>
__device__ __forceinline__ void lu_factor(LU<M>& m) {
#pragma unroll
  for (int k = 0; k < M; ++k) {
    int p = k;
    lu_t mx = fabs(m.a[k][k]);
#pragma unroll
    for (int r = k + 1; r < M; ++r) {
      const lu_t v = fabs(m.a[r][k]);
      if (v > mx) { mx = v; p = r; }
    }
    m.piv[k] = p;
    // row swap with static register indices (a[p][c] with a run-time p would put the matrix in scratch memory)
#pragma unroll
    for (int r = k + 1; r < M; ++r) {
      const bool sw = (p == r);
#pragma unroll
      for (int c = 0; c < M; ++c) {
        const lu_t tk = m.a[k][c], tr = m.a[r][c];
        m.a[k][c] = sw ? tr : tk;
        m.a[r][c] = sw ? tk : tr;
      }
    }
    const lu_t inv = 1.0 / m.a[k][k];
#pragma unroll
    for (int r = k + 1; r < M; ++r) {
      m.a[r][k] *= inv;
#pragma unroll
      for (int c = k + 1; c < M; ++c) m.a[r][c] -= m.a[r][k] * m.a[k][c];
    }
  }
}

template <int M>
__device__ __forceinline__ void lu_solve(const LU<M>& m, lu_t (&b)[M]) {
#pragma unroll
  for (int k = 0; k < M; ++k) {
    const int p = m.piv[k];
#pragma unroll
    for (int r = k + 1; r < M; ++r) {
      const bool sw = (p == r);
      const lu_t tk = b[k], tr = b[r];
      b[k] = sw ? tr : tk;
      b[r] = sw ? tk : tr;
    }
  }
#pragma unroll
  for (int k = 0; k < M; ++k)
#pragma unroll
    for (int r = k + 1; r < M; ++r) b[r] -= m.a[r][k] * b[k];
#pragma unroll
  for (int k = M - 1; k >= 0; --k) {
    b[k] /= m.a[k][k];
#pragma unroll
    for (int r = 0; r < k; ++r) b[r] -= m.a[r][k] * b[k];
  }
}

// solve A^T x = b with A = P^T L U
template <int M>
__device__ __forceinline__ void lu_solve_t(const LU<M>& m, lu_t (&b)[M]) {
#pragma unroll
  for (int k = 0; k < M; ++k) {  // U^T w = b
#pragma unroll
    for (int r = 0; r < k; ++r) b[k] -= m.a[r][k] * b[r];
    b[k] /= m.a[k][k];
  }
#pragma unroll
  for (int k = M - 1; k >= 0; --k)  // L^T mu = w (unit diagonal)
#pragma unroll
    for (int r = k + 1; r < M; ++r) b[k] -= m.a[r][k] * b[r];
#pragma unroll
  for (int k = M - 1; k >= 0; --k) {  // x = P^T mu
    const int p = m.piv[k];
#pragma unroll
    for (int r = k + 1; r < M; ++r) {
      const bool sw = (p == r);
      const lu_t tk = b[k], tr = b[r];
      b[k] = sw ? tr : tk;
      b[r] = sw ? tk : tr;
    }
  }
}

// A_n = A / (row norm + 1e-8) (FVgrad.py:335-336), formed in double from the moment matrix as the reference stores it
template <int M>
__device__ __forceinline__ void load_An(const float* A, const float* rn, int i, LU<M>& m) {
  const float* p = A + (size_t)i * (M * M);
#pragma unroll
  for (int r = 0; r < M; ++r) {
    const lu_t inv = 1.0 / (lu_t)rn[(size_t)i * M + r];
#pragma unroll
    for (int c = 0; c < M; ++c) m.a[r][c] = (lu_t)p[M * r + c] * inv;
  }
}

// WLSQ forward (FVgrad.py:295-359): lane (node i, channel c), 8 lanes per node
template <int M>
__global__ __launch_bounds__(256) void wlsq_fwd_kernel(const float* __restrict__ phi, const int* __restrict__ rowptr,
                                                       const int* __restrict__ outn, const float* __restrict__ Bp,
                                                       const float* __restrict__ An, const float* __restrict__ rn,
                                                       float* __restrict__ grad, float* __restrict__ full, int N) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int i = t >> 3, c = t & 7;
  if (i >= N) return;
  lu_t rhs[M];
#pragma unroll
  for (int j = 0; j < M; ++j) rhs[j] = 0.0;
  if (c < 7) {
    const lu_t pi = (lu_t)phi[(size_t)i * 8 + c];
    const int beg = rowptr[i], end = rowptr[i + 1];
    // 4 stencil entries per trip: index, neighbour value and moment vector of all four are requested before the first
    // dependent multiply (the loop was one memory round trip per entry); the accumulation order stays entry by entry
    int k = beg;
    for (; k + 4 <= end; k += 4) {
      lu_t dv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) dv[u] = (lu_t)phi[(size_t)outn[k + u] * 8 + c] - pi;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* b = Bp + (size_t)(k + u) * M;
#pragma unroll
        for (int j = 0; j < M; ++j) rhs[j] += (lu_t)b[j] * dv[u];
      }
    }
    for (; k < end; ++k) {
      const lu_t d = (lu_t)phi[(size_t)outn[k] * 8 + c] - pi;
      const float* b = Bp + (size_t)k * M;
#pragma unroll
      for (int j = 0; j < M; ++j) rhs[j] += (lu_t)b[j] * d;
    }
    LU<M> m;
    load_An(An, rn, i, m);
#pragma unroll
    for (int j = 0; j < M; ++j) rhs[j] = rhs[j] / (lu_t)rn[(size_t)i * M + j];
    lu_factor(m);
    lu_solve(m, rhs);
    grad[(size_t)i * 16 + 2 * c] = (float)rhs[0];
    grad[(size_t)i * 16 + 2 * c + 1] = (float)rhs[1];
    if (full) {
#pragma unroll
      for (int j = 0; j < M; ++j) full[((size_t)i * 8 + c) * M + j] = (float)rhs[j];
    }
  } else {
    grad[(size_t)i * 16 + 14] = 0.f;
    grad[(size_t)i * 16 + 15] = 0.f;
  }
}

// WLSQ backward, stage 1: g_rhs[i,c,:] = (A_n^-T [g_grad[i,c,0:2],0,...]) / rn
template <int M>
__global__ __launch_bounds__(256) void wlsq_bwd_solve_kernel(const float* __restrict__ ggrad, const float* __restrict__ gfull,
                                                             const float* __restrict__ An, const float* __restrict__ rn,
                                                             float* __restrict__ grhs, int N, int nch) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int i = t >> 3, c = t & 7;
  if (i >= N) return;
  lu_t b[M];
#pragma unroll
  for (int j = 0; j < M; ++j) b[j] = 0.0;
  if (c < nch) {  // fused path: channels 5,6 (uv_old) carry no gradient
    if (gfull) {
#pragma unroll
      for (int j = 0; j < M; ++j) b[j] = (lu_t)gfull[((size_t)i * 8 + c) * M + j];
    } else {
      b[0] = (lu_t)ggrad[(size_t)i * 16 + 2 * c];
      b[1] = (lu_t)ggrad[(size_t)i * 16 + 2 * c + 1];
    }
    LU<M> m;
    load_An(An, rn, i, m);
    lu_factor(m);
    lu_solve_t(m, b);
#pragma unroll
    for (int j = 0; j < M; ++j) b[j] = b[j] / (lu_t)rn[(size_t)i * M + j];
  }
#pragma unroll
  for (int j = 0; j < M; ++j) grhs[((size_t)i * 8 + c) * M + j] = (float)b[j];
}

// dot product of two M-vectors, paired like the 5-term form this kernel started with
template <int M>
__device__ __forceinline__ float dotM(const float* b, const float* g) {
  float s = 0.f;
  if (M == 5) {
    s = (b[0] * g[0] + b[1] * g[1]) + (b[2] * g[2] + b[3] * g[3]) + b[4] * g[4];
  } else {
#pragma unroll
    for (int j = 0; j < M; ++j) s += b[j] * g[j];
  }
  return s;
}

// stage 2: g_phi[j,c] += sum_{d: out_d = j} B_d . g_rhs[in_d,c,:]  -  sumB[j] . g_rhs[j,c,:]
template <int M>
__global__ __launch_bounds__(256) void wlsq_bwd_gather_kernel(const float* __restrict__ grhs, const int* __restrict__ rowptr_o,
                                                              const int* __restrict__ inn, const float* __restrict__ Bo,
                                                              const float* __restrict__ sumB, float* __restrict__ gphi,
                                                              int N, int nch) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int j = t >> 3, c = t & 7;
  if (j >= N || c >= nch) return;
  float s = 0.f;
  const int beg = rowptr_o[j], end = rowptr_o[j + 1];
  int k = beg;
  for (; k + 4 <= end; k += 4) {   // 4 entries per trip (loads of all four in flight), summed entry by entry
    float t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = dotM<M>(Bo + (size_t)(k + u) * M, grhs + ((size_t)inn[k + u] * 8 + c) * M);
    s += t[0]; s += t[1]; s += t[2]; s += t[3];
  }
  for (; k < end; ++k) s += dotM<M>(Bo + (size_t)k * M, grhs + ((size_t)inn[k] * 8 + c) * M);
  s -= dotM<M>(sumB + (size_t)j * M, grhs + ((size_t)j * 8 + c) * M);
  gphi[(size_t)j * 8 + c] += s;
}

// ---- node -> face (FVInterpolation.py:111-185) + face BC (FVscheme.py:32-48) ------------------------------------------
// Ff[f] = { phi_f[0..4], grad_f[c][a] (c<5, a<2) at 5+2c+a, pad }
__global__ __launch_bounds__(256) void face_fwd_kernel(const float* __restrict__ phi, const float* __restrict__ grad,
                                                       const int* __restrict__ es, const int* __restrict__ er,
                                                       const float* __restrict__ pos, const float* __restrict__ fpos,
                                                       const int* __restrict__ ftype, const float* __restrict__ y,
                                                       float* __restrict__ Ff, int E) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= E) return;
  const int s = es[f], r = er[f];
  const float fx = fpos[2 * f], fy = fpos[2 * f + 1];
  const float rsx = fx - pos[2 * s], rsy = fy - pos[2 * s + 1];
  const float rrx = fx - pos[2 * r], rry = fy - pos[2 * r + 1];
  float out[16];
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    const float gsx = grad[(size_t)s * 16 + 2 * c], gsy = grad[(size_t)s * 16 + 2 * c + 1];
    const float grx = grad[(size_t)r * 16 + 2 * c], gry = grad[(size_t)r * 16 + 2 * c + 1];
    const float vs = phi[(size_t)s * 8 + c] + (rsx * gsx + rsy * gsy);
    const float vr = phi[(size_t)r * 8 + c] + (rrx * grx + rry * gry);
    out[c] = (vs + vr) / 2.0f;
    out[5 + 2 * c] = (gsx + grx) / 2.0f;
    out[5 + 2 * c + 1] = (gsy + gry) / 2.0f;
  }
  out[15] = 0.f;
  const int ft = ftype[f];
  if (ft == NT_INFLOW) {
    const float yu = (y[2 * s] + y[2 * r]) / 2.f, yv = (y[2 * s + 1] + y[2 * r + 1]) / 2.f;
    out[0] = yu; out[1] = yv; out[3] = yu; out[4] = yv;
  } else if (ft == NT_WALL) {
    out[0] = 0.f; out[1] = 0.f; out[3] = 0.f; out[4] = 0.f;
  }
  float4* o = reinterpret_cast<float4*>(Ff + (size_t)f * 16);
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = make_float4(out[4 * j], out[4 * j + 1], out[4 * j + 2], out[4 * j + 3]);
}

// ---- cells: node->cell mean (FVInterpolation.py:36-109), fluxes and residuals (FVscheme.py:145-250) --------------------
struct CellArgs {
  const float* phi;      // [N,8]
  const float* grad;     // [N,16]
  const float* Ff;       // [E,16]
  const float* pos;      // [N,2]
  const int* crow;       // [C+1]
  const int* kface;      // [Sg]
  const int* knode;      // [Sg]
  const float* kS;       // [Sg,2] surface vectors
  const int* ftype;      // [E]
  const float* centroid; // [C,2]
  const float* area;     // [C]
  const int* cbatch;     // [C]
  const float* theta;    // [B,9]
  const float* dt;       // [B]
  const float* uvp_dim;  // [B,3]
  const float* sigma;    // [B,3]
  float* phic;           // [C,8]
  float* cres;           // [C,4] = (div, Rx, Ry, lp2)
  float* uvp_cell;       // [C,3] dimensional cell output
  int C;
  int mode;              // 0 conserved form (FVscheme.py:50-274), 1 non-conserved form (:276-511, hessian None)
  float* gradc;          // mode 1: [C,16] cell means of the node gradients of channels 0..4 (saved for the adjoint)
};

__global__ __launch_bounds__(256) void cell_fwd_kernel(const CellArgs A) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= A.C) return;
  const int b = A.cbatch[c];
  const float* th = A.theta + (size_t)b * 9;
  const float th0 = th[0], th2 = th[2], th3 = th[3], th4 = th[4], th5 = th[5];
  const float cx = A.centroid[2 * c], cy = A.centroid[2 * c + 1];
  const float area = A.area[c];
  float pc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float div = 0.f, Jx = 0.f, Jy = 0.f, lp2 = 0.f;
  float gcs[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // mode 1: sum of node gradients, channels 0..4
  float vx = 0.f, vy = 0.f;                                            // mode 1: sum_faces grad(uv_hat)_f . S
  const int beg = A.crow[c], end = A.crow[c + 1];
  for (int k = beg; k < end; ++k) {
    const int n = A.knode[k], f = A.kface[k];
    const float rx = cx - A.pos[2 * n], ry = cy - A.pos[2 * n + 1];
#pragma unroll
    for (int ch = 0; ch < 7; ++ch)
      pc[ch] += A.phi[(size_t)n * 8 + ch] + (rx * A.grad[(size_t)n * 16 + 2 * ch] + ry * A.grad[(size_t)n * 16 + 2 * ch + 1]);
    const float Sx = A.kS[2 * k], Sy = A.kS[2 * k + 1];
    const float4* fp = reinterpret_cast<const float4*>(A.Ff + (size_t)f * 16);
    const float4 f0 = fp[0], f1 = fp[1], f2 = fp[2], f3 = fp[3];
    // f0 = (u, v, p, uh)  f1 = (vh, gu_x, gu_y, gv_x)  f2 = (gv_y, gp_x, gp_y, guh_x)  f3 = (guh_y, gvh_x, gvh_y, pad)
    const float u = f0.x, v = f0.y, p = f0.z, uh = f0.w, vh = f1.x;
    if (A.mode == 0) {
      div += u * Sx + v * Sy;
      const float m00 = (uh * uh) * th2 + p * th3 - f2.w * th4;
      const float m01 = (uh * vh) * th2 + 0.f * th3 - f3.x * th4;
      const float m10 = (vh * uh) * th2 + 0.f * th3 - f3.y * th4;
      const float m11 = (vh * vh) * th2 + p * th3 - f3.z * th4;
      Jx += m00 * Sx + m01 * Sy;
      Jy += m10 * Sx + m11 * Sy;
    } else {
#pragma unroll
      for (int j = 0; j < 10; ++j) gcs[j] += A.grad[(size_t)n * 16 + j];
      vx += f2.w * Sx + f3.x * Sy;   // divergence-form diffusion (FVscheme.py:458-469)
      vy += f3.y * Sx + f3.z * Sy;
    }
    if (A.ftype[f] == NT_OUTFLOW) {
      const float l0 = th4 * (f1.y * Sx + f1.z * Sy) - p * Sx;
      const float l1 = th4 * (f1.w * Sx + f2.x * Sy) - p * Sy;
      lp2 += l0 * l0 + l1 * l1;
    }
  }
  const float cnt = fmaxf((float)(end - beg), 1.f);
#pragma unroll
  for (int ch = 0; ch < 7; ++ch) pc[ch] = pc[ch] / cnt;
  const float dtb = A.dt[b];
  const float ux = ((pc[0] - pc[5]) / dtb) * area, uy = ((pc[1] - pc[6]) / dtb) * area;
  const float src = th5 * area;
  float Rx, Ry;
  if (A.mode == 0) {
    Rx = th0 * ux + (Jx - src);
    Ry = th0 * uy + (Jy - src);
  } else {
#pragma unroll
    for (int j = 0; j < 10; ++j) gcs[j] = gcs[j] / cnt;
    div = (gcs[0] + gcs[3]) * area;                                      // (du/dx + dv/dy) area, FVscheme.py:405-408
    const float cvx = (gcs[6] * pc[3] + gcs[7] * pc[4]) * area;          // (u_hat . grad) u_hat, :447-450
    const float cvy = (gcs[8] * pc[3] + gcs[9] * pc[4]) * area;
    const float gpx = gcs[4] * area, gpy = gcs[5] * area;                // grad p, :454
    Rx = th0 * ux + th2 * cvx + th3 * gpx - th4 * vx - src;              // :472-478
    Ry = th0 * uy + th2 * cvy + th3 * gpy - th4 * vy - src;
    float4* og = reinterpret_cast<float4*>(A.gradc + (size_t)c * 16);
    og[0] = make_float4(gcs[0], gcs[1], gcs[2], gcs[3]);
    og[1] = make_float4(gcs[4], gcs[5], gcs[6], gcs[7]);
    og[2] = make_float4(gcs[8], gcs[9], 0.f, 0.f);
    og[3] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4* o = reinterpret_cast<float4*>(A.phic + (size_t)c * 8);
  o[0] = make_float4(pc[0], pc[1], pc[2], pc[3]);
  o[1] = make_float4(pc[4], pc[5], pc[6], 0.f);
  *reinterpret_cast<float4*>(A.cres + (size_t)c * 4) = make_float4(div, Rx, Ry, lp2);
  if (A.uvp_cell) {
    const float* ud = A.uvp_dim + (size_t)b * 3;
    const float* sg = A.sigma + (size_t)b * 3;
    A.uvp_cell[3 * c] = pc[0] * ud[0] * sg[0];
    A.uvp_cell[3 * c + 1] = pc[1] * ud[1] * sg[1];
    A.uvp_cell[3 * c + 2] = pc[2] * ud[2] * sg[2];
  }
}

// per-graph sums of squares -> the four residual losses (FVscheme.py:158-164,184-188,243-250)
__global__ __launch_bounds__(1024) void graph_loss_kernel(const float* __restrict__ cres, const int* __restrict__ gcell_ptr,
                                                         const float* __restrict__ theta, const float* __restrict__ sigma,
                                                         float* __restrict__ sums, float* __restrict__ losses) {
  __shared__ float red[4][1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  // one workgroup per graph: 8 loads in flight per thread (the loop was one memory round trip per cell), summed in order
  const int c_end = gcell_ptr[b + 1];
  int c = gcell_ptr[b] + tid;
  for (; c + 7 * 1024 < c_end; c += 8 * 1024) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(cres + (size_t)(c + 1024 * u) * 4);
#pragma unroll
    for (int u = 0; u < 8; ++u) { s0 += v[u].x * v[u].x; s1 += v[u].y * v[u].y; s2 += v[u].z * v[u].z; s3 += v[u].w; }
  }
  for (; c < c_end; c += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(cres + (size_t)c * 4);
    s0 += v.x * v.x; s1 += v.y * v.y; s2 += v.z * v.z; s3 += v.w;
  }
  red[0][tid] = s0; red[1][tid] = s1; red[2][tid] = s2; red[3][tid] = s3;
  __syncthreads();
  for (int o = 512; o >= 1; o >>= 1) {
    if (tid < o) {
#pragma unroll
      for (int j = 0; j < 4; ++j) red[j][tid] += red[j][tid + o];
    }
    __syncthreads();
  }
  if (tid == 0) {
    const float S0 = red[0][0], S1 = red[1][0], S2 = red[2][0], S3 = red[3][0];
    sums[4 * b] = S0; sums[4 * b + 1] = S1; sums[4 * b + 2] = S2; sums[4 * b + 3] = S3;
    losses[4 * b] = sqrtf(S0) * theta[(size_t)b * 9 + 1];
    losses[4 * b + 1] = sqrtf(S1) * sigma[(size_t)b * 3];
    losses[4 * b + 2] = sqrtf(S2) * sigma[(size_t)b * 3 + 1];
    losses[4 * b + 3] = sqrtf(S3);
  }
}

// cell -> node inverse-distance smoothing (FVInterpolation.py:218-265), Dirichlet overwrite (importer.py:223) and
// re-dimensionalisation (importer.py:228-229)
__global__ __launch_bounds__(256) void cell_to_node_kernel(const float* __restrict__ phic, const int* __restrict__ nrow,
                                                           const int* __restrict__ ncell, const float* __restrict__ pos,
                                                           const float* __restrict__ centroid, const int* __restrict__ node_type,
                                                           const float* __restrict__ y, const int* __restrict__ nbatch,
                                                           const float* __restrict__ uvp_dim, const float* __restrict__ sigma,
                                                           const float* __restrict__ phi, int smooth, float* __restrict__ out,
                                                           int N) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  float u, v, p;
  // smooth: bit 0 = inverse-distance smoothing (else the node field itself), bit 1 = RAW: the value the reference's stand-alone
  // Intergrator returns (FVscheme.py:253-262,718-724) - before the Dirichlet overwrite, not re-dimensionalised
  if (smooth & 1) {
    const float px = pos[2 * i], py = pos[2 * i + 1];
    float su = 0.f, sv = 0.f, sp = 0.f, sw = 0.f;
    for (int k = nrow[i]; k < nrow[i + 1]; ++k) {
      const int c = ncell[k];
      const float dx = px - centroid[2 * c], dy = py - centroid[2 * c + 1];
      const float w = 1.0f / sqrtf(dx * dx + dy * dy);
      su += phic[(size_t)c * 8] * w; sv += phic[(size_t)c * 8 + 1] * w; sp += phic[(size_t)c * 8 + 2] * w; sw += w;
    }
    u = su / sw; v = sv / sw; p = sp / sw;
  } else {
    u = phi[(size_t)i * 8]; v = phi[(size_t)i * 8 + 1]; p = phi[(size_t)i * 8 + 2];
  }
  if (smooth & 2) {
    out[3 * i] = u; out[3 * i + 1] = v; out[3 * i + 2] = p;
    return;
  }
  const int nt = node_type[i];
  if (nt == NT_WALL || nt == NT_INFLOW || nt == NT_PRESS || nt == NT_INWALL) { u = y[2 * i]; v = y[2 * i + 1]; }
  if (nt == NT_PRESS) p = 0.f;
  const int b = nbatch[i];
  out[3 * i] = u * uvp_dim[3 * b] * sigma[3 * b];
  out[3 * i + 1] = v * uvp_dim[3 * b + 1] * sigma[3 * b + 1];
  out[3 * i + 2] = p * uvp_dim[3 * b + 2] * sigma[3 * b + 2];
}

// ---- backward ---------------------------------------------------------------------------------------------------
// per cell: gc = (g_div, g_Rx, g_Ry, coef_lp) from the loss gradients
__global__ __launch_bounds__(256) void cell_bwd_kernel(const float* __restrict__ cres, const float* __restrict__ sums,
                                                       const float* __restrict__ gloss, const int* __restrict__ cbatch,
                                                       const float* __restrict__ theta, const float* __restrict__ sigma,
                                                       float* __restrict__ gc, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int b = cbatch[c];
  const float4 r = *reinterpret_cast<const float4*>(cres + (size_t)c * 4);
  const float S0 = sums[4 * b], S1 = sums[4 * b + 1], S2 = sums[4 * b + 2], S3 = sums[4 * b + 3];
  const float g0 = S0 > 0.f ? gloss[4 * b] * theta[(size_t)b * 9 + 1] * r.x / sqrtf(S0) : 0.f;
  const float g1 = S1 > 0.f ? gloss[4 * b + 1] * sigma[(size_t)b * 3] * r.y / sqrtf(S1) : 0.f;
  const float g2 = S2 > 0.f ? gloss[4 * b + 2] * sigma[(size_t)b * 3 + 1] * r.z / sqrtf(S2) : 0.f;
  const float g3 = S3 > 0.f ? gloss[4 * b + 3] / sqrtf(S3) : 0.f;
  *reinterpret_cast<float4*>(gc + (size_t)c * 4) = make_float4(g0, g1, g2, g3);
}

// per face: gradient of everything the adjacent cells did with the face values -> gFf [E,16]
__global__ __launch_bounds__(256) void face_bwd_kernel(const float* __restrict__ Ff, const float* __restrict__ gc,
                                                       const int* __restrict__ frow, const int* __restrict__ fk,
                                                       const int* __restrict__ kcell, const float* __restrict__ kS,
                                                       const int* __restrict__ ftype, const int* __restrict__ cbatch,
                                                       const float* __restrict__ theta, float* __restrict__ gFf, int E,
                                                       int mode) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= E) return;
  const float4* fp = reinterpret_cast<const float4*>(Ff + (size_t)f * 16);
  const float4 f0 = fp[0], f1 = fp[1], f2 = fp[2];
  const float p = f0.z, uh = f0.w, vh = f1.x;
  const int ft = ftype[f];
  float g[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) g[j] = 0.f;
  for (int q = frow[f]; q < frow[f + 1]; ++q) {
    const int k = fk[q];
    const int c = kcell[k];
    const float* th = theta + (size_t)cbatch[c] * 9;
    const float th2 = th[2], th3 = th[3], th4 = th[4];
    const float Sx = kS[2 * k], Sy = kS[2 * k + 1];
    const float4 gcv = *reinterpret_cast<const float4*>(gc + (size_t)c * 4);
    const float gd = gcv.x, gx = gcv.y, gy = gcv.z;
    if (mode == 0) {  // face fluxes of the conserved form; the non-conserved form takes these terms from cell gradients
      g[0] += gd * Sx;
      g[1] += gd * Sy;
      const float uS = uh * Sx + vh * Sy, gU = gx * uh + gy * vh, gS = gx * Sx + gy * Sy;
      g[3] += th2 * (gx * uS + gU * Sx);
      g[4] += th2 * (gy * uS + gU * Sy);
      g[2] += th3 * gS;
    }
    g[5 + 6] -= th4 * gx * Sx;  // d/d grad(u_hat)_x
    g[5 + 7] -= th4 * gx * Sy;
    g[5 + 8] -= th4 * gy * Sx;
    g[5 + 9] -= th4 * gy * Sy;
    if (ft == NT_OUTFLOW) {
      const float l0 = th4 * (f1.y * Sx + f1.z * Sy) - p * Sx;
      const float l1 = th4 * (f1.w * Sx + f2.x * Sy) - p * Sy;
      const float gl0 = gcv.w * l0, gl1 = gcv.w * l1;
      g[5 + 0] += th4 * gl0 * Sx;
      g[5 + 1] += th4 * gl0 * Sy;
      g[5 + 2] += th4 * gl1 * Sx;
      g[5 + 3] += th4 * gl1 * Sy;
      g[2] -= gl0 * Sx + gl1 * Sy;
    }
  }
  if (ft == NT_INFLOW || ft == NT_WALL) { g[0] = 0.f; g[1] = 0.f; g[3] = 0.f; g[4] = 0.f; }
  float4* o = reinterpret_cast<float4*>(gFf + (size_t)f * 16);
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = make_float4(g[4 * j], g[4 * j + 1], g[4 * j + 2], g[4 * j + 3]);
}

// per node: adjoint of node->face and node->cell interpolation -> g_phi [N,8], g_grad [N,16]
struct NodeBwdArgs {
  const float* gFf;      // [E,16]
  const float* gc;       // [C,4]
  const int* nfrow;      // [N+1] node -> incident faces
  const int* nfcol2;     // [2E]  2*face + side
  const int* nrow;       // [N+1] node -> incident (cell, node) incidences
  const int* ncell;      // [Sg]  cell of each incidence
  const float* pos;      // [N,2]
  const float* fpos;     // [E,2]
  const float* centroid; // [C,2]
  const int* crow;       // [C+1] (for the incidence count of a cell)
  const float* area;
  const int* cbatch;
  const float* theta;
  const float* dt;
  float* gphi;           // [N,8]
  float* ggrad;          // [N,16]
  int N;
  int mode;              // 1: non-conserved form
  const float* gradc;    // mode 1: [C,16] saved cell gradients
  const float* phic;     // mode 1: [C,8] saved cell values
};

__global__ __launch_bounds__(256) void node_bwd_kernel(const NodeBwdArgs A) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= A.N) return;
  const float px = A.pos[2 * i], py = A.pos[2 * i + 1];
  float gp[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  float gg[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) gg[j] = 0.f;
  for (int k = A.nfrow[i]; k < A.nfrow[i + 1]; ++k) {
    const int f = A.nfcol2[k] >> 1;
    const float rx = A.fpos[2 * f] - px, ry = A.fpos[2 * f + 1] - py;
    const float4* gpn = reinterpret_cast<const float4*>(A.gFf + (size_t)f * 16);
    const float4 a0 = gpn[0], a1 = gpn[1], a2 = gpn[2], a3 = gpn[3];
    const float v[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const float h = 0.5f * v[c];
      gp[c] += h;
      gg[2 * c] += h * rx + 0.5f * v[5 + 2 * c];
      gg[2 * c + 1] += h * ry + 0.5f * v[5 + 2 * c + 1];
    }
  }
  for (int k = A.nrow[i]; k < A.nrow[i + 1]; ++k) {
    const int c = A.ncell[k];
    const int b = A.cbatch[c];
    const float cnt = fmaxf((float)(A.crow[c + 1] - A.crow[c]), 1.f);
    const float coef = A.theta[(size_t)b * 9] * A.area[c] / A.dt[b] / cnt;
    const float gx = A.gc[(size_t)c * 4 + 1] * coef, gy = A.gc[(size_t)c * 4 + 2] * coef;
    const float rx = A.centroid[2 * c] - px, ry = A.centroid[2 * c + 1] - py;
    gp[0] += gx; gp[1] += gy;
    gg[0] += gx * rx; gg[1] += gx * ry;
    gg[2] += gy * rx; gg[3] += gy * ry;
    if (A.mode == 1) {
      // adjoint of the gradient-based terms of the non-conserved form: every node of the cell carries 1/cnt of the cell
      // mean of the node gradients, and of the cell value uv_hat that multiplies them in the convection term
      const float* th = A.theta + (size_t)b * 9;
      const float w = A.area[c] / cnt;
      const float gd = A.gc[(size_t)c * 4] * w, gRx = A.gc[(size_t)c * 4 + 1] * w, gRy = A.gc[(size_t)c * 4 + 2] * w;
      const float* gcv = A.gradc + (size_t)c * 16;
      const float uhc = A.phic[(size_t)c * 8 + 3], vhc = A.phic[(size_t)c * 8 + 4];
      const float th2 = th[2], th3 = th[3];
      gg[0] += gd;                          // d div / d (du/dx)
      gg[3] += gd;                          // d div / d (dv/dy)
      gg[4] += th3 * gRx; gg[5] += th3 * gRy;
      gg[6] += th2 * gRx * uhc; gg[7] += th2 * gRx * vhc;
      gg[8] += th2 * gRy * uhc; gg[9] += th2 * gRy * vhc;
      const float a3 = th2 * (gRx * gcv[6] + gRy * gcv[8]);   // d / d u_hat(cell), per node share
      const float a4 = th2 * (gRx * gcv[7] + gRy * gcv[9]);   // d / d v_hat(cell)
      gp[3] += a3; gp[4] += a4;
      gg[6] += a3 * rx; gg[7] += a3 * ry;
      gg[8] += a4 * rx; gg[9] += a4 * ry;
    }
  }
  float4* o = reinterpret_cast<float4*>(A.gphi + (size_t)i * 8);
  o[0] = make_float4(gp[0], gp[1], gp[2], gp[3]);
  o[1] = make_float4(gp[4], 0.f, 0.f, 0.f);
  float4* og = reinterpret_cast<float4*>(A.ggrad + (size_t)i * 16);
  og[0] = make_float4(gg[0], gg[1], gg[2], gg[3]);
  og[1] = make_float4(gg[4], gg[5], gg[6], gg[7]);
  og[2] = make_float4(gg[8], gg[9], 0.f, 0.f);
  og[3] = make_float4(0.f, 0.f, 0.f, 0.f);
}


// ==== round 6: fewer launches in the finite-volume section (each of these kernels lasted 3 - 10 us, most of it the ~4.5 us a
// launch of any size costs: profiles/r05_latency_floor.txt) ==========================================================
//
// (1) forward tail = graph_loss + train_loss + cell_to_node in ONE launch of 1024-thread workgroups: workgroups [0, B) form the
// residual norms of graph b exactly as graph_loss_kernel does; the one of them that finishes LAST also forms the training loss
// and its gradient with train_loss_kernel's arithmetic (same strided partial sums, same fold); workgroups >= B smooth 1024
// nodes each (cell_to_node_kernel's arithmetic).  The three have no data dependence on each other except losses -> train loss.
__global__ __launch_bounds__(1024) void fvm_tail_kernel(const float* __restrict__ cres, const int* __restrict__ gcell_ptr,
                                                       const float* __restrict__ theta, const float* __restrict__ sigma,
                                                       float* __restrict__ sums, float* __restrict__ losses, int B,
                                                       const float* __restrict__ hyper, float* __restrict__ loss,
                                                       float* __restrict__ gloss, int* __restrict__ counter,
                                                       const float* __restrict__ phic, const int* __restrict__ nrow,
                                                       const int* __restrict__ ncell, const float* __restrict__ pos,
                                                       const float* __restrict__ centroid, const int* __restrict__ node_type,
                                                       const float* __restrict__ y, const int* __restrict__ nbatch,
                                                       const float* __restrict__ uvp_dim, const float* __restrict__ phi, int smooth,
                                                       float* __restrict__ out, int N) {
  __shared__ float red[4][1024];
  __shared__ int last;
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= B) {
    const int i = ((int)blockIdx.x - B) * 1024 + tid;
    if (i >= N) return;
    float u, v, p;
    if (smooth & 1) {
      const float px = pos[2 * i], py = pos[2 * i + 1];
      float su = 0.f, sv = 0.f, sp = 0.f, sw = 0.f;
      for (int k = nrow[i]; k < nrow[i + 1]; ++k) {
        const int c = ncell[k];
        const float dx = px - centroid[2 * c], dy = py - centroid[2 * c + 1];
        const float w = 1.0f / sqrtf(dx * dx + dy * dy);
        su += phic[(size_t)c * 8] * w; sv += phic[(size_t)c * 8 + 1] * w; sp += phic[(size_t)c * 8 + 2] * w; sw += w;
      }
      u = su / sw; v = sv / sw; p = sp / sw;
    } else {
      u = phi[(size_t)i * 8]; v = phi[(size_t)i * 8 + 1]; p = phi[(size_t)i * 8 + 2];
    }
    if (smooth & 2) {
      out[3 * i] = u; out[3 * i + 1] = v; out[3 * i + 2] = p;
      return;
    }
    const int nt = node_type[i];
    if (nt == NT_WALL || nt == NT_INFLOW || nt == NT_PRESS || nt == NT_INWALL) { u = y[2 * i]; v = y[2 * i + 1]; }
    if (nt == NT_PRESS) p = 0.f;
    const int b = nbatch[i];
    out[3 * i] = u * uvp_dim[3 * b] * sigma[3 * b];
    out[3 * i + 1] = v * uvp_dim[3 * b + 1] * sigma[3 * b + 1];
    out[3 * i + 2] = p * uvp_dim[3 * b + 2] * sigma[3 * b + 2];
    return;
  }
  const int b = blockIdx.x;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const int c_end = gcell_ptr[b + 1];
  int c = gcell_ptr[b] + tid;
  for (; c + 7 * 1024 < c_end; c += 8 * 1024) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(cres + (size_t)(c + 1024 * u) * 4);
#pragma unroll
    for (int u = 0; u < 8; ++u) { s0 += v[u].x * v[u].x; s1 += v[u].y * v[u].y; s2 += v[u].z * v[u].z; s3 += v[u].w; }
  }
  for (; c < c_end; c += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(cres + (size_t)c * 4);
    s0 += v.x * v.x; s1 += v.y * v.y; s2 += v.z * v.z; s3 += v.w;
  }
  red[0][tid] = s0; red[1][tid] = s1; red[2][tid] = s2; red[3][tid] = s3;
  __syncthreads();
  for (int o = 512; o >= 1; o >>= 1) {
    if (tid < o) {
#pragma unroll
      for (int j = 0; j < 4; ++j) red[j][tid] += red[j][tid + o];
    }
    __syncthreads();
  }
  if (tid == 0) {
    const float S0 = red[0][0], S1 = red[1][0], S2 = red[2][0], S3 = red[3][0];
    sums[4 * b] = S0; sums[4 * b + 1] = S1; sums[4 * b + 2] = S2; sums[4 * b + 3] = S3;
    losses[4 * b] = sqrtf(S0) * theta[(size_t)b * 9 + 1];
    losses[4 * b + 1] = sqrtf(S1) * sigma[(size_t)b * 3];
    losses[4 * b + 2] = sqrtf(S2) * sigma[(size_t)b * 3 + 1];
    losses[4 * b + 3] = sqrtf(S3);
    last = 0;
    if (hyper) {
      __threadfence();
      last = atomicAdd(counter, 1) == B - 1;
    }
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  // train_loss_kernel (csrc/misc.hip) by the first 64 threads of the workgroup that saw every graph's losses written
  const float wc = hyper[5], wm = hyper[6], wp = hyper[7];
  float s = 0.f;
  if (tid < 64) {
    for (int g = tid; g < B; g += 64) {
      const float l0 = __hip_atomic_load(losses + 4 * g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float l1 = __hip_atomic_load(losses + 4 * g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float l2 = __hip_atomic_load(losses + 4 * g + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float l3 = __hip_atomic_load(losses + 4 * g + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float tot = wp * l3 + wc * l0 + wm * l1 + wm * l2;
      s += logf(tot);
      const float inv = 1.0f / (tot * (float)B);
      gloss[4 * g] = wc * inv; gloss[4 * g + 1] = wm * inv; gloss[4 * g + 2] = wm * inv; gloss[4 * g + 3] = wp * inv;
    }
    red[0][tid] = s;
  }
  __syncthreads();
  if (tid == 0) {
    float a = 0.f;
    for (int i = 0; i < 64; ++i) a += red[0][i];
    *loss = a / (float)B;
    *counter = 0;
  }
}

// (2) backward, conserved form: three launches instead of six.
// gc of a cell = cell_bwd_kernel's expressions, formed where it is used instead of being written by a launch of its own
__device__ __forceinline__ float4 fvm_gc(const float* __restrict__ cres, const float* __restrict__ sums, const float* __restrict__ gloss,
                                         const float* __restrict__ theta, const float* __restrict__ sigma, int c, int b) {
  const float4 r = *reinterpret_cast<const float4*>(cres + (size_t)c * 4);
  const float S0 = sums[4 * b], S1 = sums[4 * b + 1], S2 = sums[4 * b + 2], S3 = sums[4 * b + 3];
  const float g0 = S0 > 0.f ? gloss[4 * b] * theta[(size_t)b * 9 + 1] * r.x / sqrtf(S0) : 0.f;
  const float g1 = S1 > 0.f ? gloss[4 * b + 1] * sigma[(size_t)b * 3] * r.y / sqrtf(S1) : 0.f;
  const float g2 = S2 > 0.f ? gloss[4 * b + 2] * sigma[(size_t)b * 3 + 1] * r.z / sqrtf(S2) : 0.f;
  const float g3 = S3 > 0.f ? gloss[4 * b + 3] / sqrtf(S3) : 0.f;
  return make_float4(g0, g1, g2, g3);
}

// face_bwd_kernel (mode 0) with gc formed on the fly
__global__ __launch_bounds__(256) void face_bwd2_kernel(const float* __restrict__ Ff, const float* __restrict__ cres,
                                                        const float* __restrict__ sums, const float* __restrict__ gloss,
                                                        const float* __restrict__ sigma, const int* __restrict__ frow,
                                                        const int* __restrict__ fk, const int* __restrict__ kcell,
                                                        const float* __restrict__ kS, const int* __restrict__ ftype,
                                                        const int* __restrict__ cbatch, const float* __restrict__ theta,
                                                        float* __restrict__ gFf, int E) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= E) return;
  const float4* fp = reinterpret_cast<const float4*>(Ff + (size_t)f * 16);
  const float4 f0 = fp[0], f1 = fp[1], f2 = fp[2];
  const float p = f0.z, uh = f0.w, vh = f1.x;
  const int ft = ftype[f];
  float g[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) g[j] = 0.f;
  for (int q = frow[f]; q < frow[f + 1]; ++q) {
    const int k = fk[q];
    const int c = kcell[k];
    const int b = cbatch[c];
    const float* th = theta + (size_t)b * 9;
    const float th2 = th[2], th3 = th[3], th4 = th[4];
    const float Sx = kS[2 * k], Sy = kS[2 * k + 1];
    const float4 gcv = fvm_gc(cres, sums, gloss, theta, sigma, c, b);
    const float gd = gcv.x, gx = gcv.y, gy = gcv.z;
    g[0] += gd * Sx;
    g[1] += gd * Sy;
    const float uS = uh * Sx + vh * Sy, gU = gx * uh + gy * vh, gS = gx * Sx + gy * Sy;
    g[3] += th2 * (gx * uS + gU * Sx);
    g[4] += th2 * (gy * uS + gU * Sy);
    g[2] += th3 * gS;
    g[5 + 6] -= th4 * gx * Sx;
    g[5 + 7] -= th4 * gx * Sy;
    g[5 + 8] -= th4 * gy * Sx;
    g[5 + 9] -= th4 * gy * Sy;
    if (ft == NT_OUTFLOW) {
      const float l0 = th4 * (f1.y * Sx + f1.z * Sy) - p * Sx;
      const float l1 = th4 * (f1.w * Sx + f2.x * Sy) - p * Sy;
      const float gl0 = gcv.w * l0, gl1 = gcv.w * l1;
      g[5 + 0] += th4 * gl0 * Sx;
      g[5 + 1] += th4 * gl0 * Sy;
      g[5 + 2] += th4 * gl1 * Sx;
      g[5 + 3] += th4 * gl1 * Sy;
      g[2] -= gl0 * Sx + gl1 * Sy;
    }
  }
  if (ft == NT_INFLOW || ft == NT_WALL) { g[0] = 0.f; g[1] = 0.f; g[3] = 0.f; g[4] = 0.f; }
  float4* o = reinterpret_cast<float4*>(gFf + (size_t)f * 16);
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = make_float4(g[4 * j], g[4 * j + 1], g[4 * j + 2], g[4 * j + 3]);
}

// node_bwd_kernel (mode 0) + wlsq_bwd_solve_kernel: lane (node i, channel c), 8 lanes per node.  Channel c's share of the node
// adjoint - g_phi[i,c] and g_grad[i,c,0:2] - is a sum over the node's faces and cells that involves channel c only, formed in
// node_bwd_kernel's order; the lane then solves its own transposed system: g_grad never goes to memory.
template <int M>
__global__ __launch_bounds__(256) void node_bwd_solve_kernel(const NodeBwdArgs A, const float* __restrict__ cres,
                                                             const float* __restrict__ sums, const float* __restrict__ gloss,
                                                             const float* __restrict__ sigma, const float* __restrict__ An,
                                                             const float* __restrict__ rn, float* __restrict__ grhs) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int i = t >> 3, c = t & 7;
  if (i >= A.N) return;
  lu_t b[M];
#pragma unroll
  for (int j = 0; j < M; ++j) b[j] = 0.0;
  float gp = 0.f;
  if (c < 5) {
    const float px = A.pos[2 * i], py = A.pos[2 * i + 1];
    float g0 = 0.f, g1 = 0.f;
    for (int k = A.nfrow[i]; k < A.nfrow[i + 1]; ++k) {
      const int f = A.nfcol2[k] >> 1;
      const float rx = A.fpos[2 * f] - px, ry = A.fpos[2 * f + 1] - py;
      const float* gf = A.gFf + (size_t)f * 16;
      const float h = 0.5f * gf[c];
      gp += h;
      g0 += h * rx + 0.5f * gf[5 + 2 * c];
      g1 += h * ry + 0.5f * gf[5 + 2 * c + 1];
    }
    if (c < 2) {
      for (int k = A.nrow[i]; k < A.nrow[i + 1]; ++k) {
        const int cc = A.ncell[k];
        const int bb = A.cbatch[cc];
        const float cnt = fmaxf((float)(A.crow[cc + 1] - A.crow[cc]), 1.f);
        const float coef = A.theta[(size_t)bb * 9] * A.area[cc] / A.dt[bb] / cnt;
        const float4 gcv = fvm_gc(cres, sums, gloss, A.theta, sigma, cc, bb);
        const float gxy = (c == 0 ? gcv.y : gcv.z) * coef;
        const float rx = A.centroid[2 * cc] - px, ry = A.centroid[2 * cc + 1] - py;
        gp += gxy;
        g0 += gxy * rx;
        g1 += gxy * ry;
      }
    }
    b[0] = (lu_t)g0;
    b[1] = (lu_t)g1;
    LU<M> m;
    load_An(An, rn, i, m);
    lu_factor(m);
    lu_solve_t(m, b);
#pragma unroll
    for (int j = 0; j < M; ++j) b[j] = b[j] / (lu_t)rn[(size_t)i * M + j];
  }
  A.gphi[(size_t)i * 8 + c] = gp;
#pragma unroll
  for (int j = 0; j < M; ++j) grhs[((size_t)i * 8 + c) * M + j] = (float)b[j];
}

// wlsq_bwd_gather_kernel + phi_bwd_kernel: the eight lanes of node j finish g_phi[j, 0:5], lanes 0..2 turn it into the gradient
// of the decoder output (g_phi is not written back)
template <int M>
__global__ __launch_bounds__(256) void wlsq_gather_phi_bwd_kernel(const float* __restrict__ grhs, const int* __restrict__ rowptr_o,
                                                                  const int* __restrict__ inn, const float* __restrict__ Bo,
                                                                  const float* __restrict__ sumB, const float* __restrict__ gphi,
                                                                  const float* __restrict__ dec, const int* __restrict__ node_type,
                                                                  float* __restrict__ gdec, int N, int mode) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int j = t >> 3, c = t & 7;
  float v = 0.f;
  if (j < N && c < 5) {
    float s = 0.f;
    const int beg = rowptr_o[j], end = rowptr_o[j + 1];
    int k = beg;
    for (; k + 4 <= end; k += 4) {
      float tt[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) tt[u] = dotM<M>(Bo + (size_t)(k + u) * M, grhs + ((size_t)inn[k + u] * 8 + c) * M);
      s += tt[0]; s += tt[1]; s += tt[2]; s += tt[3];
    }
    for (; k < end; ++k) s += dotM<M>(Bo + (size_t)k * M, grhs + ((size_t)inn[k] * 8 + c) * M);
    s -= dotM<M>(sumB + (size_t)j * M, grhs + ((size_t)j * 8 + c) * M);
    v = gphi[(size_t)j * 8 + c] + s;
  }
  // (every lane of the wave takes part in the shuffles; 8-lane groups are aligned inside the 64-lane wave)
  const int lane = threadIdx.x & 63, base = lane & ~7;
  const float v3 = __shfl(v, base + 3), v4 = __shfl(v, base + 4);
  if (j >= N || c >= 3) return;
  const float ch = (mode == 0) ? 0.f : (mode == 1 ? 1.f : 0.5f);
  float g = c == 0 ? v + ch * v3 : (c == 1 ? v + ch * v4 : v);
  const int nt = node_type[j];
  if (c < 2 && (nt == NT_WALL || nt == NT_INFLOW || nt == NT_PRESS || nt == NT_INWALL)) g = 0.f;
  if (c == 2 && nt == NT_PRESS) g = 0.f;
  const float th = tanhf(dec[3 * j + c] / 10.f);
  gdec[3 * j + c] = g * (1.f - th * th);
}

// ---- WLSQ moment matrices on the device (row f2; Load_mesh.py:247-272 calc_WLSQ_A_B_normal_matrix -> FVgrad.py:183-232
// compute_normal_matrix -> FVorder.py:7-86 moments_order), float64 like the host preprocessing -------------------------
// One thread per receiving node walks its directed stencil entries in CSR order (fixed order: deterministic):
//   d = pos[out_k] - pos[i];  t = Taylor monomials of d up to the order (MM = 2 / 5 / 9 / 14 terms);  w = 1 / |d|
//   A[i] += w t t^T   [MM, MM];   B[entry_k] = w t   (written to the entry's ORIGINAL position: edge order, not CSR order)
template <int MM>
__device__ __forceinline__ void taylor_terms(double x, double y, double (&t)[MM]) {
  t[0] = x; t[1] = y;
  if (MM >= 5) { t[2] = 0.5 * x * x; t[3] = 0.5 * y * y; t[4] = x * y; }
  if (MM >= 9) { t[5] = x * x * x / 6.0; t[6] = y * y * y / 6.0; t[7] = 0.5 * x * x * y; t[8] = 0.5 * y * y * x; }
  if (MM >= 14) {
    t[9] = x * x * x * x / 24.0; t[10] = x * x * x * y / 6.0; t[11] = 0.25 * x * x * y * y; t[12] = x * y * y * y / 6.0;
    t[13] = y * y * y * y / 24.0;
  }
}

template <int MM>
__global__ __launch_bounds__(128) void wlsq_moments_kernel(const double* __restrict__ pos, const int* __restrict__ rowptr,
                                                           const int* __restrict__ outn, const int* __restrict__ entry,
                                                           double* __restrict__ A, double* __restrict__ B, int N) {
  const int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= N) return;
  const double px = pos[2 * i], py = pos[2 * i + 1];
  double a[MM * (MM + 1) / 2];   // upper triangle of the symmetric sum
#pragma unroll
  for (int q = 0; q < MM * (MM + 1) / 2; ++q) a[q] = 0.0;
  for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
    const int o = outn[k];
    const double dx = pos[2 * o] - px, dy = pos[2 * o + 1] - py;
    const double w = 1.0 / sqrt(dx * dx + dy * dy);
    double t[MM];
    taylor_terms<MM>(dx, dy, t);
    double* b = B + (size_t)entry[k] * MM;
    int q = 0;
#pragma unroll
    for (int r = 0; r < MM; ++r) {
      b[r] = w * t[r];
#pragma unroll
      for (int c = r; c < MM; ++c) a[q++] += (w * t[r]) * t[c];
    }
  }
  double* Ai = A + (size_t)i * MM * MM;
  int q = 0;
#pragma unroll
  for (int r = 0; r < MM; ++r)
#pragma unroll
    for (int c = r; c < MM; ++c) {
      Ai[r * MM + c] = a[q];
      Ai[c * MM + r] = a[q];
      ++q;
    }
}

// ---- stand-alone 2nd-order interpolation operators (FVInterpolation.py:36-109, 111-185, 218-265) --------------------------
// One generic gather: out[r, c] = sum_{k in row r} w_k (phi[col_k, c] + (tgt_r - src_{col_k}) . grad[col_k, c, :]) / W_r
//   mode 0: w_k = 1, W_r = number of entries (clamped at 1)            node -> cell mean, node -> face average
//   mode 1: w_k = 1 / |tgt_r - src_{col_k}|, W_r = sum_k w_k             cell -> node inverse-distance weights
// Any channel count C; one thread per (row, channel).  The adjoint runs over the transposed incidence (rows of the
// SOURCE points: trow / tidx list, per source, the forward entries (r) that read it), so it needs no atomics either.
__global__ __launch_bounds__(256) void interp2_fwd_kernel(const float* __restrict__ phi, const float* __restrict__ grad,
                                                          const float* __restrict__ srcpos, const float* __restrict__ tgtpos,
                                                          const int* __restrict__ rowptr, const int* __restrict__ col,
                                                          int mode, float* __restrict__ out, float* __restrict__ wsum,
                                                          int R, int C) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)R * C) return;
  const int r = (int)(i / C), c = (int)(i % C);
  const float tx = tgtpos[2 * r], ty = tgtpos[2 * r + 1];
  const int beg = rowptr[r], end = rowptr[r + 1];
  float acc = 0.f, W = 0.f;
  for (int k = beg; k < end; ++k) {
    const int s = col[k];
    const float dx = tx - srcpos[2 * s], dy = ty - srcpos[2 * s + 1];
    float v = phi[(size_t)s * C + c];
    if (grad) v += dx * grad[((size_t)s * C + c) * 2] + dy * grad[((size_t)s * C + c) * 2 + 1];
    const float w = mode == 1 ? 1.0f / sqrtf(dx * dx + dy * dy) : 1.0f;
    acc += w * v;
    W += w;
  }
  if (mode == 0) W = fmaxf(W, 1.0f);
  out[i] = acc / W;
  if (wsum && c == 0) wsum[r] = W;
}

__global__ __launch_bounds__(256) void interp2_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ wsum,
                                                          const float* __restrict__ srcpos, const float* __restrict__ tgtpos,
                                                          const int* __restrict__ trow, const int* __restrict__ tidx,
                                                          int mode, float* __restrict__ gphi, float* __restrict__ ggrad,
                                                          int S, int C) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)S * C) return;
  const int s = (int)(i / C), c = (int)(i % C);
  const float sx = srcpos[2 * s], sy = srcpos[2 * s + 1];
  float gp = 0.f, gx = 0.f, gy = 0.f;
  for (int k = trow[s]; k < trow[s + 1]; ++k) {
    const int r = tidx[k];
    const float dx = tgtpos[2 * r] - sx, dy = tgtpos[2 * r + 1] - sy;
    const float w = mode == 1 ? 1.0f / sqrtf(dx * dx + dy * dy) : 1.0f;
    const float g = gout[(size_t)r * C + c] * (w / wsum[r]);
    gp += g;
    gx += g * dx;
    gy += g * dy;
  }
  gphi[i] = gp;
  if (ggrad) {
    ggrad[2 * i] = gx;
    ggrad[2 * i + 1] = gy;
  }
}

}  // namespace

#define LAUNCH1D(kernel, n, stream, ...)                                                                    \
  do {                                                                                                      \
    if ((n) > 0) {                                                                                          \
      GFV_LAUNCH(kernel, dim3(gfv_div_up((n), 256)), dim3(256), 0, (hipStream_t)(stream), __VA_ARGS__); \
      GFV_CHECK_LAUNCH();                                                                                   \
    }                                                                                                       \
  } while (0)

extern "C" int gfv_phi_fwd(const float* dec, const float* y, const int32_t* node_type, const float* uv_old, float* phi,
                           int32_t N, int32_t mode, void* stream) {
  GfvProfScope ps_(GFV_K_FVM, 0, 56.0 * N, stream);
  LAUNCH1D(phi_fwd_kernel, N, stream, dec, y, node_type, uv_old, phi, N, mode);
  return GFV_OK;
}

extern "C" int gfv_phi_bwd(const float* gphi, const float* dec, const int32_t* node_type, float* gdec, int32_t N,
                           int32_t mode, void* stream) {
  GfvProfScope ps_(GFV_K_FVM, 0, 60.0 * N, stream);
  LAUNCH1D(phi_bwd_kernel, N, stream, gphi, dec, node_type, gdec, N, mode);
  return GFV_OK;
}

// terms = Taylor terms of the reconstruction order (2 / 5 / 9 / 14)
#define WLSQ_DISPATCH(terms, CALL)        \
  switch (terms) {                        \
    case 2: { constexpr int MM = 2; CALL; break; }   \
    case 5: { constexpr int MM = 5; CALL; break; }   \
    case 9: { constexpr int MM = 9; CALL; break; }   \
    case 14: { constexpr int MM = 14; CALL; break; } \
    default: return GFV_ERR_ARG;          \
  }

extern "C" int gfv_wlsq_fwd_ex(const float* phi, const int32_t* rowptr, const int32_t* outn, const float* Bp,
                               const float* An, const float* rn, float* grad, float* full, int32_t N, int32_t terms,
                               void* stream) {
  // stencil entry = index + moment vector, per node phi + A + rn + grad
  GfvProfScope ps_(GFV_K_FVM, 0, (4.0 + 4.0 * terms) * gfv_prof_size_S() + (32.0 + 4.0 * terms * terms + 4.0 * terms + 64.0) * N, stream);
  WLSQ_DISPATCH(terms, LAUNCH1D(wlsq_fwd_kernel<MM>, (long)N * 8, stream, phi, rowptr, outn, Bp, An, rn, grad, full, N));
  return GFV_OK;
}

extern "C" int gfv_wlsq_bwd_ex(const float* ggrad, const float* gfull, const float* An, const float* rn,
                               const int32_t* rowptr_o, const int32_t* inn, const float* Bo, const float* sumB,
                               float* grhs_ws, float* gphi, int32_t N, int32_t terms, void* stream) {
  GfvProfScope ps_(GFV_K_FVM, 0, (4.0 + 4.0 * terms) * gfv_prof_size_S() + (64.0 + 4.0 * terms * terms + 4.0 * terms + 2 * 32.0 * terms + 32.0) * N, stream);
  if ((ggrad == nullptr) == (gfull == nullptr)) return GFV_ERR_ARG;
  const int nch = gfull ? 7 : 5;
  WLSQ_DISPATCH(terms, LAUNCH1D(wlsq_bwd_solve_kernel<MM>, (long)N * 8, stream, ggrad, gfull, An, rn, grhs_ws, N, nch));
  WLSQ_DISPATCH(terms, LAUNCH1D(wlsq_bwd_gather_kernel<MM>, (long)N * 8, stream, grhs_ws, rowptr_o, inn, Bo, sumB, gphi, N, nch));
  return GFV_OK;
}

extern "C" int gfv_wlsq_fwd(const float* phi, const int32_t* rowptr, const int32_t* outn, const float* Bp, const float* An,
                            const float* rn, float* grad, int32_t N, void* stream) {
  return gfv_wlsq_fwd_ex(phi, rowptr, outn, Bp, An, rn, grad, nullptr, N, 5, stream);
}

extern "C" int gfv_wlsq_fwd_full(const float* phi, const int32_t* rowptr, const int32_t* outn, const float* Bp,
                                 const float* An, const float* rn, float* grad, float* full5, int32_t N, void* stream) {
  return gfv_wlsq_fwd_ex(phi, rowptr, outn, Bp, An, rn, grad, full5, N, 5, stream);
}

extern "C" int gfv_wlsq_bwd_full(const float* g5, const float* An, const float* rn, const int32_t* rowptr_o,
                                 const int32_t* inn, const float* Bo, const float* sumB, float* grhs_ws, float* gphi,
                                 int32_t N, void* stream) {
  return gfv_wlsq_bwd_ex(nullptr, g5, An, rn, rowptr_o, inn, Bo, sumB, grhs_ws, gphi, N, 5, stream);
}

extern "C" int gfv_wlsq_bwd(const float* ggrad, const float* An, const float* rn, const int32_t* rowptr_o,
                            const int32_t* inn, const float* Bo, const float* sumB, float* grhs_ws, float* gphi, int32_t N,
                            void* stream) {
  return gfv_wlsq_bwd_ex(ggrad, nullptr, An, rn, rowptr_o, inn, Bo, sumB, grhs_ws, gphi, N, 5, stream);
}

extern "C" int gfv_face_fwd(const float* phi, const float* grad, const int32_t* es, const int32_t* er, const float* pos,
                            const float* fpos, const int32_t* ftype, const float* y, float* Ff, int32_t E, void* stream) {
  GfvProfScope ps_(GFV_K_FVM, 0, (8.0 + 8.0 + 4.0 + 64.0) * E, stream);   // indices, face centre, type, Ff out (node rows are cache hits)
  LAUNCH1D(face_fwd_kernel, E, stream, phi, grad, es, er, pos, fpos, ftype, y, Ff, E);
  return GFV_OK;
}

extern "C" int gfv_cell_fwd_ex(const float* phi, const float* grad, const float* Ff, const float* pos, const int32_t* crow,
                               const int32_t* kface, const int32_t* knode, const float* kS, const int32_t* ftype,
                               const float* centroid, const float* area, const int32_t* cbatch, const float* theta,
                               const float* dt, const float* uvp_dim, const float* sigma, float* phic, float* cres,
                               float* uvp_cell, int32_t C, int32_t non_conserved, float* gradc, void* stream) {
  GfvProfScope ps_(GFV_K_FVM, 0, 16.0 * gfv_prof_size_Sigma() + (4.0 + 8.0 + 4.0 + 4.0 + 32.0 + 16.0 + 12.0) * C, stream);
  if (non_conserved && !gradc) return GFV_ERR_ARG;
  CellArgs a{phi, grad, Ff, pos, crow, kface, knode, kS, ftype, centroid, area, cbatch, theta, dt, uvp_dim, sigma,
             phic, cres, uvp_cell, C, non_conserved ? 1 : 0, gradc};
  LAUNCH1D(cell_fwd_kernel, C, stream, a);
  return GFV_OK;
}

extern "C" int gfv_cell_fwd(const float* phi, const float* grad, const float* Ff, const float* pos, const int32_t* crow,
                            const int32_t* kface, const int32_t* knode, const float* kS, const int32_t* ftype,
                            const float* centroid, const float* area, const int32_t* cbatch, const float* theta,
                            const float* dt, const float* uvp_dim, const float* sigma, float* phic, float* cres,
                            float* uvp_cell, int32_t C, void* stream) {
  return gfv_cell_fwd_ex(phi, grad, Ff, pos, crow, kface, knode, kS, ftype, centroid, area, cbatch, theta, dt, uvp_dim, sigma,
                         phic, cres, uvp_cell, C, 0, nullptr, stream);
}

extern "C" int gfv_graph_loss(const float* cres, const int32_t* gcell_ptr, const float* theta, const float* sigma,
                              float* sums, float* losses, int32_t B, void* stream) {
  GfvProfScope ps_(GFV_K_FVM, 0, 16.0 * gfv_prof_size_Sigma() / 3.0, stream);
  if (B <= 0) return GFV_OK;
  GFV_LAUNCH(graph_loss_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, cres, gcell_ptr, theta, sigma, sums,
                     losses);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_cell_to_node(const float* phic, const int32_t* nrow, const int32_t* ncell, const float* pos,
                                const float* centroid, const int32_t* node_type, const float* y, const int32_t* nbatch,
                                const float* uvp_dim, const float* sigma, const float* phi, int32_t smooth, float* out,
                                int32_t N, void* stream) {
  GfvProfScope ps_(GFV_K_FVM, 0, 4.0 * gfv_prof_size_Sigma() + (4.0 + 8.0 + 4.0 + 8.0 + 32.0 + 12.0) * N, stream);
  LAUNCH1D(cell_to_node_kernel, N, stream, phic, nrow, ncell, pos, centroid, node_type, y, nbatch, uvp_dim, sigma, phi,
           smooth, out, N);
  return GFV_OK;
}

extern "C" int gfv_fvm_bwd_ex(const float* cres, const float* sums, const float* gloss, const float* Ff,
                              const int32_t* cbatch, const float* theta, const float* sigma, const float* dt,
                              const int32_t* frow, const int32_t* fk, const int32_t* kcell, const float* kS,
                              const int32_t* ftype, const int32_t* nfrow, const int32_t* nfcol2, const int32_t* nrow,
                              const int32_t* ncell, const int32_t* crow, const float* pos, const float* fpos,
                              const float* centroid, const float* area, float* gc_ws, float* gFf_ws, float* gphi,
                              float* ggrad, int32_t N, int32_t E, int32_t C, int32_t non_conserved, const float* gradc,
                              const float* phic, void* stream) {
  GfvProfScope ps_(GFV_K_FVM, 0, 40.0 * gfv_prof_size_Sigma() + 32.0 * C + 140.0 * E + 120.0 * N, stream);
  if (non_conserved && (!gradc || !phic)) return GFV_ERR_ARG;
  const int mode = non_conserved ? 1 : 0;
  LAUNCH1D(cell_bwd_kernel, C, stream, cres, sums, gloss, cbatch, theta, sigma, gc_ws, C);
  LAUNCH1D(face_bwd_kernel, E, stream, Ff, gc_ws, frow, fk, kcell, kS, ftype, cbatch, theta, gFf_ws, E, mode);
  NodeBwdArgs a{gFf_ws, gc_ws, nfrow, nfcol2, nrow, ncell, pos, fpos, centroid, crow, area, cbatch, theta, dt, gphi, ggrad, N,
                mode, gradc, phic};
  LAUNCH1D(node_bwd_kernel, N, stream, a);
  return GFV_OK;
}

extern "C" int gfv_fvm_bwd(const float* cres, const float* sums, const float* gloss, const float* Ff, const int32_t* cbatch,
                           const float* theta, const float* sigma, const float* dt, const int32_t* frow, const int32_t* fk,
                           const int32_t* kcell, const float* kS, const int32_t* ftype, const int32_t* nfrow,
                           const int32_t* nfcol2, const int32_t* nrow, const int32_t* ncell, const int32_t* crow,
                           const float* pos, const float* fpos, const float* centroid, const float* area, float* gc_ws,
                           float* gFf_ws, float* gphi, float* ggrad, int32_t N, int32_t E, int32_t C, void* stream) {
  return gfv_fvm_bwd_ex(cres, sums, gloss, Ff, cbatch, theta, sigma, dt, frow, fk, kcell, kS, ftype, nfrow, nfcol2, nrow, ncell,
                        crow, pos, fpos, centroid, area, gc_ws, gFf_ws, gphi, ggrad, N, E, C, 0, nullptr, nullptr, stream);
}

extern "C" int gfv_interp2_fwd(const float* phi, const float* grad, const float* srcpos, const float* tgtpos,
                               const int32_t* rowptr, const int32_t* col, int32_t mode, float* out, float* wsum, int32_t R,
                               int32_t C, void* stream) {
  if (R < 0 || C < 1 || (mode != 0 && mode != 1) || !wsum) return GFV_ERR_ARG;
  GfvProfScope ps_(GFV_K_FVM, 0, 0, stream);
  LAUNCH1D(interp2_fwd_kernel, (long)R * C, stream, phi, grad, srcpos, tgtpos, rowptr, col, mode, out, wsum, R, C);
  return GFV_OK;
}

extern "C" int gfv_interp2_bwd(const float* gout, const float* wsum, const float* srcpos, const float* tgtpos,
                               const int32_t* trow, const int32_t* tidx, int32_t mode, float* gphi, float* ggrad, int32_t S,
                               int32_t C, void* stream) {
  if (S < 0 || C < 1 || (mode != 0 && mode != 1)) return GFV_ERR_ARG;
  GfvProfScope ps_(GFV_K_FVM, 0, 0, stream);
  LAUNCH1D(interp2_bwd_kernel, (long)S * C, stream, gout, wsum, srcpos, tgtpos, trow, tidx, mode, gphi, ggrad, S, C);
  return GFV_OK;
}

extern "C" int gfv_wlsq_moments(const double* pos, const int32_t* rowptr, const int32_t* outn, const int32_t* entry, double* A,
                                double* B, int32_t N, int32_t terms, void* stream) {
  if (N < 0) return GFV_ERR_ARG;
  if (N == 0) return GFV_OK;
  switch (terms) {
    case 2: GFV_LAUNCH(wlsq_moments_kernel<2>, dim3(gfv_div_up(N, 128)), dim3(128), 0, (hipStream_t)stream, pos, rowptr, outn, entry, A, B, N); break;
    case 5: GFV_LAUNCH(wlsq_moments_kernel<5>, dim3(gfv_div_up(N, 128)), dim3(128), 0, (hipStream_t)stream, pos, rowptr, outn, entry, A, B, N); break;
    case 9: GFV_LAUNCH(wlsq_moments_kernel<9>, dim3(gfv_div_up(N, 128)), dim3(128), 0, (hipStream_t)stream, pos, rowptr, outn, entry, A, B, N); break;
    case 14: GFV_LAUNCH(wlsq_moments_kernel<14>, dim3(gfv_div_up(N, 128)), dim3(128), 0, (hipStream_t)stream, pos, rowptr, outn, entry, A, B, N); break;
    default: return GFV_ERR_ARG;
  }
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

// ---- round 6: fused forward tail / three-launch backward (kernels above) ----------------------------------------------
extern "C" int gfv_fvm_fwd_tail(const gfv_fvm_mesh_t* m, const float* cres, const float* phic, const float* phi, float* sums,
                                float* losses, float* uvp_node, const float* hyper, float* loss, float* gloss, int32_t* counter,
                                void* stream) {
  if (!m || m->B <= 0 || !cres || !sums || !losses) return GFV_ERR_ARG;
  if (hyper && (!loss || !gloss || !counter)) return GFV_ERR_ARG;
  if (uvp_node && (!phic || !phi)) return GFV_ERR_ARG;
  GfvProfScope ps_(GFV_K_FVM, 0, 16.0 * gfv_prof_size_Sigma() / 3.0 + (uvp_node ? 4.0 * gfv_prof_size_Sigma() + 68.0 * m->N : 0.0), stream);
  const int nodes = uvp_node ? m->N : 0;
  GFV_LAUNCH(fvm_tail_kernel, dim3(m->B + gfv_div_up(nodes, 1024)), dim3(1024), 0, (hipStream_t)stream, cres, m->gcell_ptr, m->theta,
             m->sigma, sums, losses, m->B, hyper, loss, gloss, counter, phic, m->nrow, m->ncell, m->pos, m->centroid, m->node_type,
             m->y, m->batch, m->uvp_dim, phi, m->smooth, uvp_node, nodes);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_fvm_bwd_fused(const gfv_fvm_mesh_t* m, const float* cres, const float* sums, const float* gloss, const float* Ff,
                                 const float* dec, float* gFf_ws, float* gphi_ws, float* grhs_ws, float* gdec, void* stream) {
  if (!m || !cres || !sums || !gloss || !Ff || !dec || !gFf_ws || !gphi_ws || !grhs_ws || !gdec) return GFV_ERR_ARG;
  const int N = m->N, E = m->E, terms = m->terms;
  GfvProfScope ps_(GFV_K_FVM, 0, 40.0 * gfv_prof_size_Sigma() + 32.0 * m->C + 140.0 * E + 120.0 * N + (4.0 + 4.0 * terms) * gfv_prof_size_S()
                                 + (64.0 + 4.0 * terms * terms + 4.0 * terms + 2 * 32.0 * terms + 32.0) * N + 60.0 * N, stream);
  LAUNCH1D(face_bwd2_kernel, E, stream, Ff, cres, sums, gloss, m->sigma, m->frow, m->fk, m->kcell, m->kS, m->ftype, m->cbatch, m->theta,
           gFf_ws, E);
  NodeBwdArgs a{gFf_ws, nullptr, m->nfrow, m->nfcol2, m->nrow, m->ncell, m->pos, m->fpos, m->centroid, m->crow, m->area, m->cbatch, m->theta,
                m->dt, gphi_ws, nullptr, N, 0, nullptr, nullptr};
  WLSQ_DISPATCH(terms, LAUNCH1D(node_bwd_solve_kernel<MM>, (long)N * 8, stream, a, cres, sums, gloss, m->sigma, m->An, m->rn, grhs_ws));
  WLSQ_DISPATCH(terms, LAUNCH1D(wlsq_gather_phi_bwd_kernel<MM>, (long)N * 8, stream, grhs_ws, m->xo_rowptr, m->xo_in, m->xo_B, m->sumB,
                                gphi_ws, dec, m->node_type, gdec, N, m->mode));
  return GFV_OK;
}

