// Split-fp16 weight images for the f16-MFMA form of the fused GEMM chain (contract: include/gfv.h, gfv_weight_images).
#include "gfv_common.h"
#include "gfv_prof.h"
#include "gfv_split.h"
#include "../../include/gfv.h"
#include <mutex>
#include <unordered_map>

namespace {

// The form tag of a set of images (round 5).  An image holds fp16 (hi, lo) parts - forms 0 / 1 / 2 - or bf16 high parts and no low
// parts - form 3 -, and nothing in its bytes says which; a launch in the other class of form would multiply garbage without any
// error.  The images of one build share the device scalar they were scaled with (`wmax`), and every launch that uses them names
// that scalar again: so the library remembers, per wmax address, the class its images were last built in (host state, updated
// by gfv_weight_images - also while a command list is being recorded), and the entry points that take images refuse a launch whose
// product form is of the other class (GFV_ERR_ARG).  An address the library has not seen (images a caller laid out itself) is
// not checked.
std::mutex g_tag_mu;
std::unordered_map<const void*, int> g_tag;   // wmax address -> 0 fp16 parts, 1 bf16 high parts

__global__ __launch_bounds__(256) void wabsmax_kernel(const gfv_wimg_desc_t* __restrict__ descs, float* __restrict__ wmax) {
  const gfv_wimg_desc_t d = descs[blockIdx.y];
  float m = 0.f;
  if (d.ldw == d.K && ((d.N * d.K) & 3) == 0 && (reinterpret_cast<size_t>(d.W) & 15) == 0) {
    // a whole parameter matrix: contiguous, float4
    const int total4 = (d.N * d.K) >> 2;
    const float4* w4 = reinterpret_cast<const float4*>(d.W);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total4; i += gridDim.x * 256) {
      const float4 v = w4[i];
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
  } else {
    for (int n = blockIdx.x * 4 + (threadIdx.x >> 6); n < d.N; n += gridDim.x * 4)
      for (int k = threadIdx.x & 63; k < d.K; k += 64) m = fmaxf(m, fabsf(d.W[(size_t)n * d.ldw + k]));
  }
  m = gfv_wave_max(m);
  // ONE atomic per workgroup (four per workgroup - 800 on one address - were most of this launch's 12.6 us: round 5)
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  // non-negative floats order like their bit patterns
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    if (m > 0.f) atomicMax(reinterpret_cast<unsigned*>(wmax), __float_as_uint(m));
  }
}

// one thread = one (pass, T, nt, lane) fragment, both parts (bf: the bf16 single-product form - the high part in bf16, no low part)
__global__ __launch_bounds__(256) void wimg_kernel(const gfv_wimg_desc_t* __restrict__ descs, const float* __restrict__ wmax, int bf) {
  const gfv_wimg_desc_t d = descs[blockIdx.y];
  const int nT = (d.K + 31) >> 5, npass = (d.N + 127) >> 7;
  const long f = (long)blockIdx.x * 256 + threadIdx.x;
  if (f >= (long)npass * nT * 512) return;
  const float ws = gfv_pow2_scale(*wmax);
  const int lane = (int)(f & 63), nt = (int)((f >> 6) & 7);
  const int rest = (int)(f >> 9);
  const int T = rest % nT, pass = rest / nT;
  const int n = 128 * pass + 16 * nt + (lane & 15), g = lane >> 4;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 32 * T + 16 * (e >> 2) + 4 * g + (e & 3);
    v[e] = (n < d.N && k < d.K) ? d.W[(size_t)n * d.ldw + k] * ws : 0.f;
  }
  gfv_uint4 hi, lo;
  if (bf) gfv_split8_t<true>(v, hi, lo);   // (uniform)
  else gfv_split8(v, hi, lo);
  gfv_uint4* img = reinterpret_cast<gfv_uint4*>(d.img) + ((size_t)(pass * nT + T) * 8 + nt) * 128 + lane;
  img[0] = hi;
  img[64] = lo;
}

}  // namespace

extern "C" size_t gfv_weight_image_bytes(int32_t N, int32_t K) {
  return (size_t)((N + 127) / 128) * ((K + 31) / 32) * 16384;
}

extern "C" int gfv_weight_absmax(const gfv_wimg_desc_t* descs_dev, int32_t n_desc, float* wmax, void* stream) {
  GfvProfScope ps_(GFV_K_WIMG, 0, 4.0 * 1181539.0, stream);
  if (!descs_dev || !wmax || n_desc < 0) return GFV_ERR_ARG;
  if (gfv_memset_rec(wmax, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) return GFV_ERR_LAUNCH;   // (recordable: gfv_launch.h)
  if (n_desc == 0) return GFV_OK;
  GFV_LAUNCH(wabsmax_kernel, dim3(4, n_desc), dim3(256), 0, (hipStream_t)stream, descs_dev, wmax);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

extern "C" int gfv_f16split_enabled(void);   // rowtile.hip
extern "C" int gfv_weight_images(const gfv_wimg_desc_t* descs_dev, int32_t n_desc, int64_t max_frags, const float* wmax,
                                 void* stream) {
  GfvProfScope ps_(GFV_K_WIMG, 0, 32.0 * (double)max_frags * n_desc, stream);
  if (!descs_dev || !wmax || n_desc < 0 || max_frags < 0) return GFV_ERR_ARG;
  if (n_desc == 0 || max_frags == 0) return GFV_OK;
  const int bf = gfv_f16split_enabled() == 3 ? 1 : 0;
  {
    std::lock_guard<std::mutex> lk(g_tag_mu);
    g_tag[wmax] = bf;
  }
  GFV_LAUNCH(wimg_kernel, dim3((unsigned)((max_frags + 255) / 256), n_desc), dim3(256), 0, (hipStream_t)stream,
                     descs_dev, wmax, bf);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

// -1: images the library did not build; 0: fp16 parts (forms 0 - 2); 1: bf16 high parts (form 3)
extern "C" int gfv_weight_images_form(const float* wmax) {
  std::lock_guard<std::mutex> lk(g_tag_mu);
  const auto it = g_tag.find(wmax);
  return it == g_tag.end() ? -1 : it->second;
}
// may a launch of the calling thread's product form use the images that were scaled with `wmax`?
bool gfv_internal_wimg_form_ok(const float* wmax) {
  const int tag = gfv_weight_images_form(wmax);
  return tag < 0 || tag == (gfv_f16split_enabled() == 3 ? 1 : 0);
}
