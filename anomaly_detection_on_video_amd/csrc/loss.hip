// Fused MGFN loss reductions, forward and backward.
//   smooth  = 8e-4 * sum_{b,t>=1} (s[b,t]-s[b,t-1])^2            /root/reference/src/loss/base.py:16-18
//   sparse  = 8e-3 * || s[:bs/2].flatten() ||_2                 base.py:30-31, modeling_mgfn.py:409
//   bce     = BCELoss(cat(nor,abn scores), cat(nor,abn labels))  /root/reference/src/loss/mgfn.py:23-27
//   la/ln   = L1 norm over F of the selected features            mgfn.py:29-30
//   con     = mean_r clamp(200 - ||la_r - ln_r + 1e-6||_2, 0)^2  mgfn.py:28-32, base.py:42-47
//   con_a/n = mean_r ||l[sep+r] - l[r] + 1e-6||_2^2              mgfn.py:33-42
//   mgfn    = bce + 1e-3 * (1e-3*con + con_a + con_n)            mgfn.py:44-45
//   total   = mgfn + smooth + sparse                             modeling_mgfn.py:418
// Small tensors (bs*T ~ 1e3, R*k ~ 5e2): one block does the scalar reductions deterministically.
#include <algorithm>

#include "common.h"

namespace advhip {

constexpr float kLambda1 = 8e-4f, kLambda2 = 8e-3f, kAlpha = 1e-3f, kMargin = 200.f, kPdEps = 1e-6f;

__device__ __forceinline__ float wave_sum_l(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// block-wide sum (blockDim.x == 256), result valid in every thread
__device__ float block_sum(float v, float* red) {
  v = wave_sum_l(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// one wavefront per (row, j): L1 norm over F
__global__ void l1norm_kernel(const float* __restrict__ a, const float* __restrict__ n, float* __restrict__ ws, int Rk,
                              int F) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (w >= 2 * Rk) return;
  const float* p = (w < Rk ? a + (size_t)w * F : n + (size_t)(w - Rk) * F);
  float s = 0.f;
  for (int i = lane; i < F; i += 64) s += fabsf(p[i]);
  s = wave_sum_l(s);
  if (lane == 0) ws[w] = s;
}

__device__ __forceinline__ float bce_term(float p, float y) {
  // torch.nn.BCELoss clamps the logs at -100
  const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);
  return -(y * lp + (1.f - y) * lq);
}

__global__ __launch_bounds__(256) void loss_fwd_kernel(const float* __restrict__ scores,
                                                       const float* __restrict__ abn, const float* __restrict__ nor,
                                                       const float* __restrict__ yl_abn,
                                                       const float* __restrict__ yl_nor, const float* __restrict__ ws,
                                                       float* __restrict__ out, int bs, int T, int R, int k) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int n = bs / 2;
  float v = 0.f;
  for (int i = tid; i < bs * T; i += 256) {
    const int t = i % T;
    if (t > 0) { const float d = scores[i] - scores[i - 1]; v += d * d; }
  }
  const float smooth = kLambda1 * block_sum(v, red);
  v = 0.f;
  for (int i = tid; i < n * T; i += 256) v += scores[i] * scores[i];
  const float sparse = kLambda2 * sqrtf(block_sum(v, red));
  v = 0.f;
  for (int i = tid; i < 2 * n; i += 256) v += (i < n) ? bce_term(nor[i], yl_nor[i]) : bce_term(abn[i - n], yl_abn[i - n]);
  const float bce = block_sum(v, red) / (float)(2 * n);
  const float* la = ws;
  const float* ln = ws + (size_t)R * k;
  v = 0.f;
  for (int r = tid; r < R; r += 256) {
    float d2 = 0.f;
    for (int j = 0; j < k; ++j) { const float d = la[r * k + j] - ln[r * k + j] + kPdEps; d2 += d * d; }
    const float c = fmaxf(kMargin - sqrtf(d2), 0.f);
    v += c * c;
  }
  const float con = block_sum(v, red) / (float)R;
  const int sep = R / 2;
  float va = 0.f, vn = 0.f;
  for (int r = tid; r < sep; r += 256) {
    for (int j = 0; j < k; ++j) {
      const float da = la[(sep + r) * k + j] - la[r * k + j] + kPdEps;
      const float dn = ln[(sep + r) * k + j] - ln[r * k + j] + kPdEps;
      va += da * da;
      vn += dn * dn;
    }
  }
  const float con_a = block_sum(va, red) / (float)sep;
  const float con_n = block_sum(vn, red) / (float)sep;
  if (tid == 0) {
    const float mgfn = bce + kAlpha * (kAlpha * con + con_a + con_n);
    out[0] = mgfn + smooth + sparse;
    out[1] = bce; out[2] = con; out[3] = con_a; out[4] = con_n;
    out[5] = smooth; out[6] = sparse; out[7] = mgfn;
  }
}

__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ d_loss,
                                                       const float* __restrict__ scores,
                                                       const float* __restrict__ abn, const float* __restrict__ nor,
                                                       const float* __restrict__ yl_abn,
                                                       const float* __restrict__ yl_nor, float* __restrict__ ws,
                                                       float* __restrict__ d_scores, float* __restrict__ d_abn,
                                                       float* __restrict__ d_nor, int bs, int T, int R, int k) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int n = bs / 2;
  const float g = d_loss[0];
  float v = 0.f;
  for (int i = tid; i < n * T; i += 256) v += scores[i] * scores[i];
  const float nrm = sqrtf(block_sum(v, red));
  for (int i = tid; i < bs * T; i += 256) {
    const int t = i % T;
    float d = 0.f;
    if (t > 0) d += scores[i] - scores[i - 1];
    if (t < T - 1) d -= scores[i + 1] - scores[i];
    float gs = 2.f * kLambda1 * d;
    if (i < n * T && nrm > 0.f) gs += kLambda2 * scores[i] / nrm;
    d_scores[i] = g * gs;
  }
  for (int i = tid; i < 2 * n; i += 256) {
    const float p = (i < n) ? nor[i] : abn[i - n];
    const float y = (i < n) ? yl_nor[i] : yl_abn[i - n];
    // torch's binary_cross_entropy backward: (p - y) / max((1-p) p, 1e-12) / N
    const float gp = g * (p - y) / fmaxf((1.f - p) * p, 1e-12f) / (float)(2 * n);
    if (i < n) d_nor[i] = gp; else d_abn[i - n] = gp;
  }
  const float* la = ws;
  const float* ln = ws + (size_t)R * k;
  float* dla = ws + (size_t)2 * R * k;
  float* dln = ws + (size_t)3 * R * k;
  const int sep = R / 2;
  const float w_con = g * kAlpha * kAlpha / (float)R;
  const float w_same = g * kAlpha / (float)sep;
  for (int r = tid; r < R; r += 256) {
    float d2 = 0.f;
    for (int j = 0; j < k; ++j) { const float d = la[r * k + j] - ln[r * k + j] + kPdEps; d2 += d * d; }
    const float dist = sqrtf(d2);
    const float c = fmaxf(kMargin - dist, 0.f);
    const float coef = dist > 0.f ? -2.f * c / dist * w_con : 0.f;
    for (int j = 0; j < k; ++j) {
      const float d = la[r * k + j] - ln[r * k + j] + kPdEps;
      float ga = coef * d, gn = -coef * d;
      // same-class clustering terms: row r pairs with r+sep (r < sep) or r-sep (r >= sep)
      if (r < sep) {
        ga -= 2.f * w_same * (la[(sep + r) * k + j] - la[r * k + j] + kPdEps);
        gn -= 2.f * w_same * (ln[(sep + r) * k + j] - ln[r * k + j] + kPdEps);
      } else if (r - sep < sep) {
        ga += 2.f * w_same * (la[r * k + j] - la[(r - sep) * k + j] + kPdEps);
        gn += 2.f * w_same * (ln[r * k + j] - ln[(r - sep) * k + j] + kPdEps);
      }
      dla[r * k + j] = ga;
      dln[r * k + j] = gn;
    }
  }
}

// d_feat[r,j,f] = d_l1[r,j] * sign(feat[r,j,f])
__global__ void l1norm_bwd_kernel(const float* __restrict__ a, const float* __restrict__ nf,
                                  const float* __restrict__ ws, float* __restrict__ d_a, float* __restrict__ d_n,
                                  int Rk, int F) {
  const long long total = (long long)2 * Rk * F;
  const float* dla = ws + (size_t)2 * Rk;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const bool first = i < (long long)Rk * F;
    const long long e = first ? i : i - (long long)Rk * F;
    const float x = first ? a[e] : nf[e];
    const float gl = dla[(first ? 0 : Rk) + e / F];
    const float sg = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
    (first ? d_a : d_n)[e] = gl * sg;
  }
}

}  // namespace advhip

using namespace advhip;

extern "C" int64_t advhip_mgfn_loss_ws_floats(int32_t n, int32_t ncrops, int32_t k) {
  return (int64_t)4 * n * ncrops * k;
}

static int loss_shape_ok(int bs, int T, int ncrops, int k, int F) {
  ADVHIP_REQUIRE(bs >= 2 && bs % 2 == 0 && T > 0 && ncrops > 0 && k > 0 && F > 0, "mgfn_loss: bad shape");
  ADVHIP_REQUIRE(((bs / 2) * ncrops) % 2 == 0,
                 "mgfn_loss: (bs/2)*ncrops=%d must be even (the reference splits the selected rows in halves, "
                 "src/loss/mgfn.py:25)", (bs / 2) * ncrops);
  return ADVHIP_OK;
}

extern "C" int advhip_mgfn_loss_fwd_f32(const float* scores, const float* abn_score, const float* nor_score,
                                        const float* a_feat, const float* n_feat, const float* abn_labels,
                                        const float* nor_labels, float* ws, float* out, int32_t bs, int32_t T,
                                        int32_t ncrops, int32_t k, int32_t F, void* stream) {
  ADVHIP_REQUIRE(scores && abn_score && nor_score && a_feat && n_feat && abn_labels && nor_labels && ws && out,
                 "mgfn_loss_fwd: null pointer");
  if (int rc = loss_shape_ok(bs, T, ncrops, k, F)) return rc;
  const int R = (bs / 2) * ncrops, Rk = R * k;
  hipLaunchKernelGGL(l1norm_kernel, dim3((2 * Rk + 3) / 4), dim3(256), 0, (hipStream_t)stream, a_feat, n_feat, ws, Rk, F);
  if (int rc = check_launch("l1norm")) return rc;
  hipLaunchKernelGGL(loss_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scores, abn_score, nor_score, abn_labels,
                     nor_labels, ws, out, bs, T, R, k);
  return check_launch("loss_fwd");
}

extern "C" int advhip_mgfn_loss_bwd_f32(const float* d_loss, const float* scores, const float* abn_score,
                                        const float* nor_score, const float* a_feat, const float* n_feat,
                                        const float* abn_labels, const float* nor_labels, const float* ws,
                                        float* d_scores, float* d_abn_score, float* d_nor_score, float* d_a_feat,
                                        float* d_n_feat, int32_t bs, int32_t T, int32_t ncrops, int32_t k, int32_t F,
                                        void* stream) {
  ADVHIP_REQUIRE(d_loss && scores && abn_score && nor_score && a_feat && n_feat && abn_labels && nor_labels && ws &&
                     d_scores && d_abn_score && d_nor_score && d_a_feat && d_n_feat,
                 "mgfn_loss_bwd: null pointer");
  if (int rc = loss_shape_ok(bs, T, ncrops, k, F)) return rc;
  const int R = (bs / 2) * ncrops, Rk = R * k;
  // the upper half of ws (written here) holds d(la), d(ln); the lower half keeps the fwd L1 norms
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d_loss, scores, abn_score, nor_score,
                     abn_labels, nor_labels, const_cast<float*>(ws), d_scores, d_abn_score, d_nor_score, bs, T, R, k);
  if (int rc = check_launch("loss_bwd")) return rc;
  const long long total = (long long)2 * Rk * F;
  const int grid = (int)std::min<long long>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(l1norm_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a_feat, n_feat, ws, d_a_feat,
                     d_n_feat, Rk, F);
  return check_launch("l1norm_bwd");
}
