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

}  // namespace advhip

using namespace advhip;

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
