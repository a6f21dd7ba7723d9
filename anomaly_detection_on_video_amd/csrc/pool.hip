// Pooling kernels of the I3D backbone (HBM-bound, NCDHW fp32).
//   maxpool3d      : nn.MaxPool3d, padding 0, floor mode   (/root/reference/src/i3d.py:212-217, 306, 309)
//   global_avgpool : nn.AdaptiveAvgPool3d((1,1,1))         (/root/reference/src/i3d.py:244, 314)
#include <algorithm>
#include <cmath>

#include "common.h"

namespace advhip {

// One thread per output element, consecutive threads along W (coalesced stores; the window
// reads of neighbouring threads overlap and are served by L1/L2).
// `ypad` = extra elements between consecutive samples of y (0 when dense; y may be a channel slice of a wider tensor);
// `per_sample` = C*To*Ho*Wo.
// (pt, ph, pw): implicit -inf padding on both sides of each axis (nn.MaxPool3d's padding).
__global__ void maxpool3d_kernel(const float* __restrict__ x, float* __restrict__ y, int T, int H, int W, int To,
                                 int Ho, int Wo, int kt, int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw,
                                 long long total, long long per_sample, long long ypad) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int wo = (int)(i % Wo);
    long long r = i / Wo;
    const int ho = (int)(r % Ho);
    r /= Ho;
    const int to = (int)(r % To);
    const long long bc = r / To;
    const int t0 = to * st - pt, h0 = ho * sh - ph, w0 = wo * sw - pw;
    const float* p = x + bc * T * (long long)H * W;
    float m = -INFINITY;
    for (int a = 0; a < kt; ++a)
      for (int b = 0; b < kh; ++b)
        for (int c = 0; c < kw; ++c) {
          const int t = t0 + a, h = h0 + b, w = w0 + c;
          if ((unsigned)t >= (unsigned)T || (unsigned)h >= (unsigned)H || (unsigned)w >= (unsigned)W) continue;
          const float v = p[((long long)t * H + h) * W + w];
          // torch's max pooling propagates NaN
          m = (v > m || v != v) ? v : m;
        }
    y[i + (i / per_sample) * ypad] = m;
  }
}

// Specialisation for the stem pool k(2,3,3) s(2,2,2) on rows of <= 128 floats (W even): one
// wavefront per output row.  Each of the 2x3 input rows is read once as a coalesced float2 per lane
// (columns 2l, 2l+1); the third column of a window comes from the next lane by shuffle, so every
// input element is loaded by at most two waves (vs 2.25 scattered dword loads per element before).
__global__ __launch_bounds__(256) void maxpool3d_233_kernel(const float* __restrict__ x, float* __restrict__ y, int T,
                                                            int H, int W, int To, int Ho, int Wo, long long rows,
                                                            long long rows_per_sample, long long ypad) {
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * (long long)(blockDim.x >> 6) + (threadIdx.x >> 6);  // (bc, to, ho)
  if (row >= rows) return;
  const int ho = (int)(row % Ho);
  const long long r = row / Ho;
  const int to = (int)(r % To);
  const long long bc = r / To;
  const float* base = x + ((bc * T + 2 * to) * H + 2 * ho) * (long long)W;
  const bool ld = 2 * lane + 1 < W;
  float m = -INFINITY;
  bool isnan_ = false;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      float2 v = make_float2(-INFINITY, -INFINITY);
      if (ld) v = *reinterpret_cast<const float2*>(base + ((long long)a * H + b) * W + 2 * lane);
      const float nx = __shfl_down(v.x, 1, 64);  // column 2l+2
      const float w3 = fmaxf(fmaxf(v.x, v.y), nx);
      isnan_ |= (v.x != v.x) | (v.y != v.y) | (nx != nx);
      m = fmaxf(m, w3);
    }
  if (lane < Wo) y[row * Wo + lane + (row / rows_per_sample) * ypad] = isnan_ ? NAN : m;  // torch's max pooling propagates NaN
}

// Temporal-only pooling (kh = kw = 1, sh = sw = 1): y[bc,to,p] = max_a x[bc, to*st + a, p]; four
// independent outputs per thread keep more loads in flight.
__global__ void maxpool3d_t_kernel(const float* __restrict__ x, float* __restrict__ y, int T, int HW, int To, int kt,
                                   int st, long long total) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i0 = blockIdx.x * (long long)blockDim.x + threadIdx.x; i0 < total; i0 += 4 * stride) {
    float m[4];
    long long idx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = i0 + u * stride;
      idx[u] = i;
      m[u] = -INFINITY;
      if (i < total) {
        const int p = (int)(i % HW);
        const long long r = i / HW;
        const int to = (int)(r % To);
        const long long bc = r / To;
        const float* q = x + ((bc * T + (long long)to * st) * HW) + p;
        for (int a = 0; a < kt; ++a) {
          const float v = q[(long long)a * HW];
          m[u] = (v > m[u] || v != v) ? v : m[u];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (idx[u] < total) y[idx[u]] = m[u];
  }
}

// One wavefront per row.  Rows of <= 128 positions (the I3D head: 2 x 7 x 7 = 98) are added exactly as the conv kernel's fused
// mean epilogue adds them (conv_igemm.hip, EPI_AVG): positions 0..63 and 64..127 each with the xor butterfly, chunk 0 + chunk 1,
// divided by n -- conv + this launch and the fused launch agree bit for bit.  Longer rows: lanes stride the row first.
__global__ void global_avgpool_kernel(const float* __restrict__ x, float* __restrict__ y, long long rows, int n) {
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * (long long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = x + row * n;
  if (n <= 128) {
    float s0 = lane < n ? p[lane] : 0.f, s1 = 64 + lane < n ? p[64 + lane] : 0.f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      s0 += __shfl_xor(s0, off, 64);
      s1 += __shfl_xor(s1, off, 64);
    }
    if (lane == 0) y[row] = (s0 + s1) / (float)n;
    return;
  }
  float s = 0.f;
  for (int i = lane; i < n; i += 64) s += p[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) y[row] = s / (float)n;
}

}  // namespace advhip

using namespace advhip;

extern "C" int advhip_maxpool3d_f32(const float* x, float* y, int32_t B, int32_t C, int32_t T, int32_t H, int32_t W,
                                    int32_t kt, int32_t kh, int32_t kw, int32_t st, int32_t sh, int32_t sw,
                                    void* stream) {
  return advhip_maxpool3d_strided_f32(x, y, 0, B, C, T, H, W, kt, kh, kw, st, sh, sw, stream);
}

extern "C" int advhip_maxpool3d_padded_f32(const float* x, float* y, int32_t B, int32_t C, int32_t T, int32_t H, int32_t W,
                                           int32_t kt, int32_t kh, int32_t kw, int32_t st, int32_t sh, int32_t sw, int32_t pt,
                                           int32_t ph, int32_t pw, void* stream) {
  ADVHIP_REQUIRE(x && y, "maxpool3d_padded: null pointer");
  ADVHIP_REQUIRE(B > 0 && C > 0 && kt > 0 && kh > 0 && kw > 0 && st > 0 && sh > 0 && sw > 0 && pt >= 0 && ph >= 0 && pw >= 0 &&
                     2 * pt <= kt && 2 * ph <= kh && 2 * pw <= kw && T + 2 * pt >= kt && H + 2 * ph >= kh && W + 2 * pw >= kw,
                 "maxpool3d_padded: bad shape (T=%d H=%d W=%d k=%d,%d,%d s=%d,%d,%d p=%d,%d,%d; padding at most half the window)", T, H, W,
                 kt, kh, kw, st, sh, sw, pt, ph, pw);
  const int To = (T + 2 * pt - kt) / st + 1, Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
  const long long total = (long long)B * C * To * Ho * Wo;
  const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(maxpool3d_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, T, H, W, To, Ho, Wo, kt, kh, kw, st, sh,
                     sw, pt, ph, pw, total, total, 0ll);
  return check_launch("maxpool3d_padded");
}

extern "C" int advhip_maxpool3d_strided_f32(const float* x, float* y, int64_t y_batch_stride, int32_t B, int32_t C, int32_t T,
                                            int32_t H, int32_t W, int32_t kt, int32_t kh, int32_t kw, int32_t st, int32_t sh,
                                            int32_t sw, void* stream) {
  ADVHIP_REQUIRE(x && y, "maxpool3d: null pointer");
  ADVHIP_REQUIRE(B > 0 && C > 0 && T >= kt && H >= kh && W >= kw && kt > 0 && kh > 0 && kw > 0 && st > 0 && sh > 0 && sw > 0,
                 "maxpool3d: bad shape (B=%d C=%d T=%d H=%d W=%d k=%d,%d,%d s=%d,%d,%d)", B, C, T, H, W, kt, kh, kw, st, sh, sw);
  const int To = (T - kt) / st + 1, Ho = (H - kh) / sh + 1, Wo = (W - kw) / sw + 1;
  const long long total = (long long)B * C * To * Ho * Wo;
  const long long per_sample = (long long)C * To * Ho * Wo;
  ADVHIP_REQUIRE(y_batch_stride == 0 || y_batch_stride >= per_sample, "maxpool3d: y batch stride %lld < one sample (%lld)",
                 (long long)y_batch_stride, per_sample);
  const long long ypad = y_batch_stride > 0 ? y_batch_stride - per_sample : 0;
  if (kt == 2 && kh == 3 && kw == 3 && st == 2 && sh == 2 && sw == 2 && W % 2 == 0 && W <= 128 && Wo < 64) {
    const long long rows = (long long)B * C * To * Ho;
    const long long blocks = (rows + 3) / 4;
    ADVHIP_REQUIRE(blocks < (1ll << 31), "maxpool3d: too many rows");
    hipLaunchKernelGGL(maxpool3d_233_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, T, H, W, To,
                       Ho, Wo, rows, (long long)C * To * Ho, ypad);
    return check_launch("maxpool3d_233");
  }
  if (kh == 1 && kw == 1 && sh == 1 && sw == 1 && ypad == 0) {
    const int grid = (int)std::min<long long>((total + 1023) / 1024, 256 * 16);
    hipLaunchKernelGGL(maxpool3d_t_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, T, H * W, To, kt, st, total);
    return check_launch("maxpool3d_t");
  }
  const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(maxpool3d_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, T, H, W, To, Ho, Wo, kt, kh,
                     kw, st, sh, sw, 0, 0, 0, total, per_sample, ypad);
  return check_launch("maxpool3d");
}

extern "C" int advhip_global_avgpool_f32(const float* x, float* y, int64_t rows, int32_t n, void* stream) {
  ADVHIP_REQUIRE(x && y && rows > 0 && n > 0, "global_avgpool: bad arguments");
  const int wpb = 4;  // waves per block
  const long long grid = (rows + wpb - 1) / wpb;
  ADVHIP_REQUIRE(grid < (1ll << 31), "global_avgpool: too many rows");
  hipLaunchKernelGGL(global_avgpool_kernel, dim3((unsigned)grid), dim3(64 * wpb), 0, (hipStream_t)stream, x, y,
                     (long long)rows, n);
  return check_launch("global_avgpool");
}
