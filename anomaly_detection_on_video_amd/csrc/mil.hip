// MIL top-k magnitude selection of the MGFN head as wavefront-shuffle primitives (fwd + bwd).
// Restates magnitude_selection_and_score_prediction,
// /root/reference/src/models/mgfn/modeling_mgfn.py:302-374:
//   mag[b,t]  = mean_c || features[b*ncrops+c, t, :] ||_2            (:314-315)
//   sc[b,t]   = mean_c scores[b*ncrops+c, t]                          (:318)
//   idx       = topk(mag * keep, k, dim=1).indices                    (:345-346)
//   sel[c*n+v, j, :] = features[v*ncrops+c, idx[v,j], :]              (:349-355, crop-major cat)
//   score[v]  = mean_j sc[v, idx[v,j]]                                (:359-362)
#include <algorithm>

#include "common.h"

namespace advhip {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// sum of squares of one F-float row, all 64 lanes cooperate, result in every lane
__device__ __forceinline__ float row_sumsq(const float* __restrict__ p, int F, int lane) {
  float s = 0.f;
  if ((F & 3) == 0) {
    const float4* p4 = reinterpret_cast<const float4*>(p);
    for (int i = lane; i < F / 4; i += 64) {
      const float4 v = p4[i];
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
  } else {
    for (int i = lane; i < F; i += 64) s += p[i] * p[i];
  }
  return wave_sum(s);
}

// one wavefront per (b, t)
__global__ void mil_magnitude_kernel(const float* __restrict__ feat, const float* __restrict__ scores,
                                     float* __restrict__ mag, float* __restrict__ sc, int bs, int ncrops, int T,
                                     int F) {
  const int lane = threadIdx.x & 63;
  const long long w = blockIdx.x * (long long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (w >= (long long)bs * T) return;
  const int b = (int)(w / T), t = (int)(w % T);
  float m = 0.f, s = 0.f;
  for (int c = 0; c < ncrops; ++c) {
    const size_t row = (size_t)(b * ncrops + c) * T + t;
    m += sqrtf(row_sumsq(feat + row * F, F, lane));
    s += scores[row];
  }
  if (lane == 0) {
    mag[w] = m / (float)ncrops;
    sc[w] = s / (float)ncrops;
  }
}

__global__ void mil_magnitude_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ d_mag,
                                         const float* __restrict__ d_sc, float* __restrict__ d_feat,
                                         float* __restrict__ d_scores, int bs, int ncrops, int T, int F) {
  const int lane = threadIdx.x & 63;
  const long long w = blockIdx.x * (long long)(blockDim.x >> 6) + (threadIdx.x >> 6);  // (b*ncrops+c, t)
  if (w >= (long long)bs * ncrops * T) return;
  const int r = (int)(w / T), t = (int)(w % T);
  const int b = r / ncrops;
  const float* p = feat + (size_t)w * F;
  const float nrm = sqrtf(row_sumsq(p, F, lane));
  const float g = d_mag ? d_mag[(size_t)b * T + t] / (float)ncrops : 0.f;
  const float coef = nrm > 0.f ? g / nrm : 0.f;
  if (d_mag) {
    float* q = d_feat + (size_t)w * F;
    for (int i = lane; i < F; i += 64) q[i] += coef * p[i];
  }
  if (lane == 0 && d_sc) d_scores[w] += d_sc[(size_t)b * T + t] / (float)ncrops;
}

// ---- arg-max order of torch.topk: NaN is the largest value, then the larger value, then (ties) the lower index.
constexpr int MIL_NONE = 0x7fffffff;  // "no candidate yet"
__device__ __forceinline__ bool mil_better(float ov, int oi, float bv, int bi) {
  if (oi == MIL_NONE) return false;
  if (bi == MIL_NONE) return true;
  const bool o_nan = ov != ov, b_nan = bv != bv;
  if (o_nan != b_nan) return o_nan;
  if (!o_nan && ov != bv) return ov > bv;
  return oi < bi;
}

// One workgroup of W wavefronts per video, any T: k rounds of arg-max.  A round scans the row with the <= 15 indices already
// chosen excluded (kept in LDS, read as a broadcast: k <= 16, so the list replaces a per-element "taken" bitmap and T has no
// bound but int32), reduces within each wave by xor-shuffle and across the W waves through LDS in wave order.  W = 1 for the
// training shapes (T = 32 segments), 4 / 16 for whole-video validation (T = n_clips, /root/reference/src/runner.py:42-50 ->
// modeling_mgfn.py:345-346 -- torch.topk has no length limit and neither does this).
template <int W>
__global__ __launch_bounds__(64 * W) void mil_topk_kernel(const float* __restrict__ mag, const float* __restrict__ keep,
                                                          const float* __restrict__ sc, long long* __restrict__ idx,
                                                          float* __restrict__ score, int T, int k) {
  __shared__ int chosen[16];
  __shared__ float w_best[W];
  __shared__ int w_bi[W];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t v = blockIdx.x;
  const float* m = mag + v * T;
  const float* kp = keep ? keep + v * T : nullptr;
  float ssum = 0.f;
  for (int j = 0; j < k; ++j) {
    float best = -INFINITY;
    int bi = MIL_NONE;
    for (int t = threadIdx.x; t < T; t += 64 * W) {
      bool skip = false;
      for (int q = 0; q < j; ++q) skip |= chosen[q] == t;
      if (skip) continue;
      const float val = kp ? m[t] * kp[t] : m[t];
      if (mil_better(val, t, best, bi)) { best = val; bi = t; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (mil_better(ob, oi, best, bi)) { best = ob; bi = oi; }
    }
    if (W > 1) {
      if (lane == 0) { w_best[wave] = best; w_bi[wave] = bi; }
      __syncthreads();
      best = w_best[0];
      bi = w_bi[0];
#pragma unroll
      for (int w = 1; w < W; ++w)
        if (mil_better(w_best[w], w_bi[w], best, bi)) { best = w_best[w]; bi = w_bi[w]; }
    }
    ssum += sc[v * T + bi];
    if (threadIdx.x == 0) {
      chosen[j] = bi;
      idx[v * k + j] = bi;
    }
    __syncthreads();  // chosen[j] is visible to the next round's scan; w_best / w_bi may be rewritten
  }
  if (threadIdx.x == 0) score[v] = ssum / (float)k;
}

// gather: one block per (video v, crop c, j); copies F floats
__global__ void mil_gather_kernel(const float* __restrict__ feat, const long long* __restrict__ idx,
                                  float* __restrict__ sel, int n, int ncrops, int T, int F, int k) {
  const int j = blockIdx.x % k;
  const int c = (blockIdx.x / k) % ncrops;
  const int v = blockIdx.x / (k * ncrops);
  const int t = (int)idx[(size_t)v * k + j];
  const float* src = feat + ((size_t)(v * ncrops + c) * T + t) * F;
  float* dst = sel + ((size_t)(c * n + v) * k + j) * F;
  if ((F & 3) == 0) {
    for (int i = threadIdx.x; i < F / 4; i += blockDim.x)
      reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
  } else {
    for (int i = threadIdx.x; i < F; i += blockDim.x) dst[i] = src[i];
  }
}

__global__ void mil_scatter_kernel(const long long* __restrict__ idx, const float* __restrict__ d_sel,
                                   const float* __restrict__ d_score, float* __restrict__ d_feat,
                                   float* __restrict__ d_sc, int n, int ncrops, int T, int F, int k) {
  const int j = blockIdx.x % k;
  const int c = (blockIdx.x / k) % ncrops;
  const int v = blockIdx.x / (k * ncrops);
  const int t = (int)idx[(size_t)v * k + j];
  if (d_sel) {
    float* dst = d_feat + ((size_t)(v * ncrops + c) * T + t) * F;
    const float* src = d_sel + ((size_t)(c * n + v) * k + j) * F;
    // the k indices of one video are distinct, so no two blocks touch the same row
    for (int i = threadIdx.x; i < F; i += blockDim.x) dst[i] += src[i];
  }
  if (c == 0 && threadIdx.x == 0 && d_score) d_sc[(size_t)v * T + t] += d_score[v] / (float)k;
}

}  // namespace advhip

using namespace advhip;

extern "C" int advhip_mil_magnitude_f32(const float* features, const float* scores, float* mag, float* sc, int32_t bs,
                                        int32_t ncrops, int32_t T, int32_t F, void* stream) {
  ADVHIP_REQUIRE(features && scores && mag && sc, "mil_magnitude: null pointer");
  ADVHIP_REQUIRE(bs > 0 && ncrops > 0 && T > 0 && F > 0, "mil_magnitude: bad shape");
  const long long waves = (long long)bs * T;
  hipLaunchKernelGGL(mil_magnitude_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, features,
                     scores, mag, sc, bs, ncrops, T, F);
  return check_launch("mil_magnitude");
}

extern "C" int advhip_mil_magnitude_bwd_f32(const float* features, const float* d_mag, const float* d_sc,
                                            float* d_features, float* d_scores, int32_t bs, int32_t ncrops, int32_t T,
                                            int32_t F, void* stream) {
  ADVHIP_REQUIRE(features && (d_mag == nullptr || d_features) && (d_sc == nullptr || d_scores), "mil_magnitude_bwd: null pointer");
  ADVHIP_REQUIRE(bs > 0 && ncrops > 0 && T > 0 && F > 0, "mil_magnitude_bwd: bad shape");
  const long long waves = (long long)bs * ncrops * T;
  hipLaunchKernelGGL(mil_magnitude_bwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     features, d_mag, d_sc, d_features, d_scores, bs, ncrops, T, F);
  return check_launch("mil_magnitude_bwd");
}

extern "C" int advhip_mil_topk_select_f32(const float* mag, const float* keep, const float* sc, const float* features,
                                          int64_t* idx, float* sel, float* score, int32_t n, int32_t ncrops, int32_t T,
                                          int32_t F, int32_t k, void* stream) {
  ADVHIP_REQUIRE(mag && sc && features && idx && sel && score, "mil_topk_select: null pointer");
  ADVHIP_REQUIRE(n > 0 && ncrops > 0 && F > 0, "mil_topk_select: bad shape");
  ADVHIP_REQUIRE(k > 0 && k <= 16 && k <= T, "mil_topk_select: need 0 < k <= min(16, T) (k=%d T=%d)", k, T);
  long long* idx_ll = reinterpret_cast<long long*>(idx);
  if (T <= 2048)
    hipLaunchKernelGGL(mil_topk_kernel<1>, dim3(n), dim3(64), 0, (hipStream_t)stream, mag, keep, sc, idx_ll, score, T, k);
  else if (T <= 32768)
    hipLaunchKernelGGL(mil_topk_kernel<4>, dim3(n), dim3(256), 0, (hipStream_t)stream, mag, keep, sc, idx_ll, score, T, k);
  else
    hipLaunchKernelGGL(mil_topk_kernel<16>, dim3(n), dim3(1024), 0, (hipStream_t)stream, mag, keep, sc, idx_ll, score, T, k);
  if (int rc = check_launch("mil_topk")) return rc;
  hipLaunchKernelGGL(mil_gather_kernel, dim3(n * ncrops * k), dim3(256), 0, (hipStream_t)stream, features,
                     reinterpret_cast<const long long*>(idx), sel, n, ncrops, T, F, k);
  return check_launch("mil_gather");
}

extern "C" int advhip_mil_topk_select_bwd_f32(const int64_t* idx, const float* d_sel, const float* d_score,
                                              float* d_features, float* d_sc, int32_t n, int32_t ncrops, int32_t T,
                                              int32_t F, int32_t k, void* stream) {
  ADVHIP_REQUIRE(idx && (d_sel == nullptr || d_features) && (d_score == nullptr || d_sc), "mil_topk_select_bwd: null pointer");
  ADVHIP_REQUIRE(n > 0 && ncrops > 0 && F > 0 && k > 0 && k <= 16 && k <= T, "mil_topk_select_bwd: bad shape");
  hipLaunchKernelGGL(mil_scatter_kernel, dim3(n * ncrops * k), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(idx), d_sel, d_score, d_features, d_sc, n, ncrops, T, F, k);
  return check_launch("mil_scatter");
}
