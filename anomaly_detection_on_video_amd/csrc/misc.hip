// On-device analogues of the numpy post-processing either side of the hot path.
//   segment_features : /root/reference/extract_features.py:171-183
//   add_magnitude    : /root/reference/src/dataset.py:121-124
#include <algorithm>

#include "common.h"

namespace advhip {

// np.linspace(0, n, seg+1, dtype=int): float64 i*step truncated; last element is exactly n.
__device__ __forceinline__ int linspace_int(int i, int n, int seg) {
  if (i >= seg) return n;
  const double step = (double)n / (double)seg;
  return (int)((double)i * step);
}

// one thread per output element (crop, s, c); rows of a bucket are added in order in fp32 and
// divided by the count, as np.mean over axis 0 of a float32 array does
__global__ void segment_kernel(const float* __restrict__ f, float* __restrict__ out, int n, int ncrops, int C, int seg) {
  const long long total = (long long)ncrops * seg * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int s = (int)((i / C) % seg);
    const int crop = (int)(i / ((long long)C * seg));
    const int r0 = linspace_int(s, n, seg), r1 = linspace_int(s + 1, n, seg);
    float v;
    if (r0 != r1) {
      float acc = 0.f;
      for (int r = r0; r < r1; ++r) acc += f[((size_t)r * ncrops + crop) * C + c];
      v = acc / (float)(r1 - r0);
    } else {
      v = f[((size_t)r0 * ncrops + crop) * C + c];
    }
    out[i] = v;
  }
}

// one wavefront per row: copy C floats and append the L2 norm
__global__ void add_magnitude_kernel(const float* __restrict__ f, float* __restrict__ out, long long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * (long long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = f + row * C;
  float* q = out + row * (C + 1);
  float s = 0.f;
  for (int i = lane; i < C; i += 64) {
    const float v = p[i];
    q[i] = v;
    s += v * v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) q[C] = sqrtf(s);
}

// uint8 frames (N, T, C, H, W) -> fp32 (N, C, T, H, W), y = (x - mean) / std: the reference's
// PILToTensor().float() + GroupNormalize + the (B,10,16,3,H,W)->(B,10,3,16,H,W) permute
// (/root/reference/src/dataset.py:175-183, extract_features.py:83) in one pass, so only uint8 crosses
// PCIe.  Each thread converts 4 consecutive pixels of a row (one 32-bit load, one 16-byte store).
__global__ void normalize_permute_u8_kernel(const uint8_t* __restrict__ x, float* __restrict__ y, int T, int C,
                                            long long HW, float mean, float stdv, long long total4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4;
       i += (long long)gridDim.x * blockDim.x) {
    const long long hw4 = HW / 4;
    const long long p4 = i % hw4;            // output order: (n, c, t, hw)
    long long r = i / hw4;
    const int t = (int)(r % T);
    r /= T;
    const int c = (int)(r % C);
    const long long n = r / C;
    const uint32_t v = *reinterpret_cast<const uint32_t*>(x + (((n * T + t) * C + c) * HW + p4 * 4));
    float4 o;
    o.x = ((float)(v & 0xff) - mean) / stdv;
    o.y = ((float)((v >> 8) & 0xff) - mean) / stdv;
    o.z = ((float)((v >> 16) & 0xff) - mean) / stdv;
    o.w = ((float)(v >> 24) - mean) / stdv;
    reinterpret_cast<float4*>(y)[i] = o;
  }
}

// TenCrop + float + normalise + LoopPad + permute in one pass (src/gtransforms.py:20-73, 115-132; src/dataset.py:175-195;
// extract_features.py:83): resized uint8 frames (F, H, W, C) -- the HWC bytes a decoder / PIL hands over -- to the backbone's
// input (n_clips * 10, C, fpc, cs, cs) fp32.  Crop j < 5 of a frame = its (top_j, left_j) window: top-left, top-right,
// bottom-left, bottom-right, centre (torchvision five_crop order; centre offsets are Python-rounded halves, computed by the
// caller); crops 5..9 = the same five windows of the horizontally flipped frame: pixel (y, x) = frame[top + y][W - 1 - (left + x)].
// Frame t of clip c = frames[c * fpc + t % len_c], len_c = min(fpc, F - c * fpc) (LoopPad: a short last clip repeats itself).
// One wave per output row (clip, crop, c, t, y): the row decode (five divisions) happens once per 224 outputs, the lanes run
// along x with 16-byte stores (the first form decoded every float4 separately and indexed a by-value crop table: 165 us for
// 40 crop-clips, VALU-bound at 2.3 TB/s).
template <int VW>
__global__ __launch_bounds__(256) void tencrop_normalize_u8_kernel(const uint8_t* __restrict__ x, float* __restrict__ y, int F, int H, int W,
                                                                   int C, int fpc, int cs, int ctop, int cleft, float mean, float stdv,
                                                                   long long rows) {
  const int lane = threadIdx.x & 63;
  const int csv = cs / VW;
  for (long long r0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r0 < rows; r0 += (long long)gridDim.x * 4) {
    long long r = r0;  // output order: (clip, crop, c, t, y, x)
    const int yo = (int)(r % cs);
    r /= cs;
    const int t = (int)(r % fpc);
    r /= fpc;
    const int c = (int)(r % C);
    r /= C;
    const int crop = (int)(r % 10);
    const int clip = (int)(r / 10);
    const int len = min(fpc, F - clip * fpc);
    const int f = clip * fpc + t % len;
    const int j = crop % 5;
    const int top = j == 4 ? ctop : ((j >> 1) ? H - cs : 0), left = j == 4 ? cleft : ((j & 1) ? W - cs : 0);
    const uint8_t* row = x + (((long long)f * H + top + yo) * W) * C + c;
    float* out = y + r0 * cs;
    for (int q = lane; q < csv; q += 64) {
      float v[VW];
#pragma unroll
      for (int e = 0; e < VW; ++e) {
        const int xo = q * VW + e;
        const int sx = crop < 5 ? left + xo : W - 1 - (left + xo);
        v[e] = ((float)row[(long long)sx * C] - mean) / stdv;
      }
      if constexpr (VW == 4) reinterpret_cast<float4*>(out)[q] = make_float4(v[0], v[1], v[2], v[3]);
      else out[q] = v[0];
    }
  }
}

// The same pass writing COLUMN-PARITY PLANES (the operand of the stem's 16-byte gather, advhip_conv3d_s2w_*): crop-clips
// [first, first + count), xs[(clip-crop, c, t, y)][par][2 + j] = normalised pixel of crop column 2 j + par, zero in the
// padding columns.  One wave per (row, both planes); the arithmetic per pixel is the pass above's, so the values are its values.
__global__ __launch_bounds__(256) void tencrop_normalize_planes_u8_kernel(const uint8_t* __restrict__ x, float* __restrict__ xs, int F, int H,
                                                                          int W, int C, int fpc, int cs, int ctop, int cleft, float mean,
                                                                          float stdv, long long first, long long rows, int WP) {
  const int lane = threadIdx.x & 63;
  for (long long r0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r0 < rows; r0 += (long long)gridDim.x * 4) {
    long long r = r0;  // (clip-crop - first, c, t, y)
    const int yo = (int)(r % cs);
    r /= cs;
    const int t = (int)(r % fpc);
    r /= fpc;
    const int c = (int)(r % C);
    r = r / C + first;
    const int crop = (int)(r % 10);
    const int clip = (int)(r / 10);
    const int len = min(fpc, F - clip * fpc);
    const int f = clip * fpc + t % len;
    const int j5 = crop % 5;
    const int top = j5 == 4 ? ctop : ((j5 >> 1) ? H - cs : 0), left = j5 == 4 ? cleft : ((j5 & 1) ? W - cs : 0);
    const uint8_t* row = x + (((long long)f * H + top + yo) * W) * C + c;
    float* out = xs + r0 * 2 * WP;
    for (int q = lane; q < 2 * WP; q += 64) {
      const int par = q >= WP, idx = q - par * WP, xo = 2 * (idx - 2) + par;
      float v = 0.f;
      if (idx >= 2 && xo < cs) {
        const int sx = crop < 5 ? left + xo : W - 1 - (left + xo);
        v = ((float)row[(long long)sx * C] - mean) / stdv;
      }
      out[q] = v;
    }
  }
}

}  // namespace advhip

using namespace advhip;

extern "C" int advhip_tencrop_normalize_planes_u8(const uint8_t* frames, float* xs, int32_t F, int32_t H, int32_t W, int32_t C,
                                                  int32_t frames_per_clip, int32_t crop, int64_t first_crop_clip, int64_t count, float mean,
                                                  float stdv, void* stream) {
  ADVHIP_REQUIRE(frames && xs && F > 0 && C > 0 && frames_per_clip > 0 && crop > 0 && crop % 2 == 0, "tencrop_normalize_planes_u8: bad arguments");
  ADVHIP_REQUIRE(H >= crop && W >= crop, "tencrop_normalize_planes_u8: frames (%d x %d) smaller than the %d crop", H, W, crop);
  ADVHIP_REQUIRE(stdv != 0.f, "tencrop_normalize_planes_u8: std must be non-zero");
  const long long n_clips = (F + frames_per_clip - 1) / frames_per_clip;
  ADVHIP_REQUIRE(first_crop_clip >= 0 && count > 0 && first_crop_clip + count <= n_clips * 10,
                 "tencrop_normalize_planes_u8: crop-clips [%lld, %lld) outside the video's %lld", (long long)first_crop_clip,
                 (long long)(first_crop_clip + count), n_clips * 10);
  auto half_even = [](int d) { return (d % 2 == 0) ? d / 2 : ((d / 2) % 2 == 0 ? d / 2 : d / 2 + 1); };
  const int ctop = half_even(H - crop), cleft = half_even(W - crop);
  const long long rows = (long long)count * C * frames_per_clip * crop;
  const int grid = (int)std::min<long long>((rows + 3) / 4, 256 * 256);
  hipLaunchKernelGGL(tencrop_normalize_planes_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, frames, xs, F, H, W, C, frames_per_clip,
                     crop, ctop, cleft, mean, stdv, (long long)first_crop_clip, rows, crop / 2 + 4);
  return check_launch("tencrop_normalize_planes_u8");
}

extern "C" int advhip_tencrop_normalize_u8(const uint8_t* frames, float* y, int32_t F, int32_t H, int32_t W, int32_t C,
                                           int32_t frames_per_clip, int32_t crop, float mean, float stdv, void* stream) {
  ADVHIP_REQUIRE(frames && y && F > 0 && C > 0 && frames_per_clip > 0 && crop > 0, "tencrop_normalize_u8: bad arguments");
  ADVHIP_REQUIRE(H >= crop && W >= crop, "tencrop_normalize_u8: frames (%d x %d) smaller than the %d crop", H, W, crop);
  ADVHIP_REQUIRE(stdv != 0.f, "tencrop_normalize_u8: std must be non-zero");
  // torchvision center_crop: int(round((H - crop) / 2.0)) with Python's round-half-to-even
  auto half_even = [](int d) { return (d % 2 == 0) ? d / 2 : ((d / 2) % 2 == 0 ? d / 2 : d / 2 + 1); };
  const int ctop = half_even(H - crop), cleft = half_even(W - crop);
  const long long n_clips = (F + frames_per_clip - 1) / frames_per_clip;
  const bool vec = crop % 4 == 0 && ((uintptr_t)y & 15) == 0;
  const long long rows = n_clips * 10 * C * frames_per_clip * (long long)crop;
  const int grid = (int)std::min<long long>((rows + 3) / 4, 256 * 256);
  if (vec) hipLaunchKernelGGL(tencrop_normalize_u8_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, frames, y, F, H, W, C,
                              frames_per_clip, crop, ctop, cleft, mean, stdv, rows);
  else hipLaunchKernelGGL(tencrop_normalize_u8_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, frames, y, F, H, W, C,
                          frames_per_clip, crop, ctop, cleft, mean, stdv, rows);
  return check_launch("tencrop_normalize_u8");
}

extern "C" int advhip_normalize_permute_u8(const uint8_t* x, float* y, int64_t N, int32_t T, int32_t C, int32_t H,
                                           int32_t W, float mean, float stdv, void* stream) {
  ADVHIP_REQUIRE(x && y && N > 0 && T > 0 && C > 0 && H > 0 && W > 0, "normalize_permute_u8: bad arguments");
  ADVHIP_REQUIRE(((long long)H * W) % 4 == 0, "normalize_permute_u8: H*W=%lld must be a multiple of 4", (long long)H * W);
  ADVHIP_REQUIRE(stdv != 0.f, "normalize_permute_u8: std must be non-zero");
  const long long total4 = (long long)N * T * C * H * W / 4;
  const int grid = (int)std::min<long long>((total4 + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(normalize_permute_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, T, C,
                     (long long)H * W, mean, stdv, total4);
  return check_launch("normalize_permute_u8");
}

extern "C" int advhip_segment_features_f32(const float* feats, float* out, int32_t n_clips, int32_t ncrops, int32_t C,
                                           int32_t seg, void* stream) {
  ADVHIP_REQUIRE(feats && out && n_clips > 0 && ncrops > 0 && C > 0 && seg > 0, "segment_features: bad arguments");
  const long long total = (long long)ncrops * seg * C;
  const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(segment_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, feats, out, n_clips, ncrops, C, seg);
  return check_launch("segment_features");
}

extern "C" int advhip_add_magnitude_f32(const float* feats, float* out, int64_t rows, int32_t C, void* stream) {
  ADVHIP_REQUIRE(feats && out && rows > 0 && C > 0, "add_magnitude: bad arguments");
  const long long grid = (rows + 3) / 4;
  ADVHIP_REQUIRE(grid < (1ll << 31), "add_magnitude: too many rows");
  hipLaunchKernelGGL(add_magnitude_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, feats, out,
                     (long long)rows, C);
  return check_launch("add_magnitude");
}
