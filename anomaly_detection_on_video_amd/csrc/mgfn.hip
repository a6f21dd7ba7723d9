// Fused element-wise / reduction kernels of the MGFN scorer body on (C, N) activations (channels outermost, N = B*T
// positions contiguous), forward and backward -- the work between the GEMM-shaped layers that the torch formulation
// spends ~10 passes over the activation on:
//   chan_layernorm   MGFNLayerNorm over channels, (x - mean) / (sqrt(var_biased) + eps) * g + b
//                    (/root/reference/src/models/mgfn/modeling_mgfn.py:36-46)
//   dwconv_t         FocusAttention.rel_pos: per-head depth-wise temporal Conv1d(heads, heads, k, padding=k/2, groups=heads)
//                    applied to "b (c h) n -> (b c) h n" (modeling_mgfn.py:169-171, 176-178): channel c uses filter c % heads
// All reductions are deterministic (fixed-order partial sums, no atomics).
#include <algorithm>

#include "common.h"

namespace advhip {

// A block = LN_COLS positions x LN_GROUPS channel groups (512 threads): lanes run along n (128-byte row pieces), the channel
// groups split C and combine through LDS.  N = 10240 positions give 320 blocks (64-position blocks: 160 for 256 CUs) and
// the channel loops are unrolled eight-fold -- these kernels live on bytes in flight (round 2: 42 / 68 us fwd / bwd at
// 1024 x 10240 with 64 positions x 4 waves, ~2 TB/s).
#ifndef ADV_LN_COLS
#define ADV_LN_COLS 32
#endif
constexpr int LN_COLS = ADV_LN_COLS, LN_GROUPS = 512 / LN_COLS, LN_THREADS = LN_COLS * LN_GROUPS, LN_UNROLL = 8;

// per-position mean and 1 / (std_biased + eps) of x[:, n] in one read: sums of d = x - x[0, n] and d^2 (the shift keeps
// s2 / C - (s1 / C)^2 free of cancellation: d is of the order of the spread, whatever the mean)
__device__ __forceinline__ void ln_stats(const float* __restrict__ x, int Cc, long long N, long long n, bool ok, int p, int grp,
                                         float (*part)[LN_GROUPS][LN_COLS], float eps, float& mean, float& r) {
  const float x0 = ok ? x[n] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  if (ok) {
    int c = grp;
    for (; c + (LN_UNROLL - 1) * LN_GROUPS < Cc; c += LN_UNROLL * LN_GROUPS) {
      float v[LN_UNROLL];
#pragma unroll
      for (int u = 0; u < LN_UNROLL; ++u) v[u] = x[(long long)(c + u * LN_GROUPS) * N + n];
#pragma unroll
      for (int u = 0; u < LN_UNROLL; ++u) {
        const float d = v[u] - x0;
        s1 += d;
        s2 += d * d;
      }
    }
    for (; c < Cc; c += LN_GROUPS) {
      const float d = x[(long long)c * N + n] - x0;
      s1 += d;
      s2 += d * d;
    }
  }
  part[0][grp][p] = s1;
  part[1][grp][p] = s2;
  __syncthreads();
  float t1 = 0.f, t2 = 0.f;
#pragma unroll
  for (int gI = 0; gI < LN_GROUPS; ++gI) {  // fixed order: deterministic
    t1 += part[0][gI][p];
    t2 += part[1][gI][p];
  }
  const float m1 = t1 / (float)Cc;
  mean = x0 + m1;
  const float var = fmaxf(t2 / (float)Cc - m1 * m1, 0.f);
  r = 1.f / (sqrtf(var) + eps);
}

// mu[n] = mean_c x[c, n], rs[n] = 1 / (sqrt(var_biased_c x[c, n]) + eps) alone (the LayerNorm folded into a GEMM epilogue)
__global__ __launch_bounds__(LN_THREADS) void chan_stats_kernel(const float* __restrict__ x, float* __restrict__ mu, float* __restrict__ rs,
                                                                int Cc, long long N, float eps) {
  __shared__ float part[2][LN_GROUPS][LN_COLS];
  const int p = threadIdx.x % LN_COLS, grp = threadIdx.x / LN_COLS;
  const long long n = blockIdx.x * (long long)LN_COLS + p;
  const bool ok = n < N;
  float mean, r;
  ln_stats(x, Cc, N, n, ok, p, grp, part, eps, mean, r);
  if (ok && grp == 0) { mu[n] = mean; rs[n] = r; }
}

// y = (x - mu) * rs * g + b;  mu / rs per position are outputs too (saved for the backward pass)
__global__ __launch_bounds__(LN_THREADS) void chan_layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                        const float* __restrict__ b, float* __restrict__ y,
                                                                        float* __restrict__ mu, float* __restrict__ rs, int Cc, long long N,
                                                                        float eps) {
  __shared__ float part[2][LN_GROUPS][LN_COLS];
  const int p = threadIdx.x % LN_COLS, grp = threadIdx.x / LN_COLS;
  const long long n = blockIdx.x * (long long)LN_COLS + p;
  const bool ok = n < N;
  float mean, r;
  ln_stats(x, Cc, N, n, ok, p, grp, part, eps, mean, r);
  if (!ok) return;
  if (grp == 0) { mu[n] = mean; rs[n] = r; }
  int c = grp;
  for (; c + (LN_UNROLL - 1) * LN_GROUPS < Cc; c += LN_UNROLL * LN_GROUPS) {
    float v[LN_UNROLL];
#pragma unroll
    for (int u = 0; u < LN_UNROLL; ++u) v[u] = x[(long long)(c + u * LN_GROUPS) * N + n];
#pragma unroll
    for (int u = 0; u < LN_UNROLL; ++u) {
      const int cc = c + u * LN_GROUPS;
      y[(long long)cc * N + n] = (v[u] - mean) * r * g[cc] + b[cc];
    }
  }
  for (; c < Cc; c += LN_GROUPS) y[(long long)c * N + n] = (x[(long long)c * N + n] - mean) * r * g[c] + b[c];
}

// With xc = x - mu, r = rs, sigma = 1/r - eps, dyg = dy * g:
//   dx = r * (dyg - mean_c(dyg)) - r^2 / sigma * mean_c(dyg * xc) * xc
//   dg[c] = sum_n dy * xc * r,  db[c] = sum_n dy        (written as per-block partial sums pg / pb [blocks][C])
__global__ __launch_bounds__(LN_THREADS) void chan_layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                        const float* __restrict__ g, const float* __restrict__ mu,
                                                                        const float* __restrict__ rs, float* __restrict__ dx,
                                                                        float* __restrict__ pg, float* __restrict__ pb, int Cc, long long N,
                                                                        float eps, const float* __restrict__ add, long long prow) {
  // `add` (nullable, dx's shape): dx = LayerNorm backward + add -- the gradient of the block's skip connection
  // (y = f(LN(x)) + x: dL/dx = LN'(...) + dL/dy) without autograd's separate add over the activation.
  // `prow`: row pitch of the partial-sum matrices (C when pg / pb are separate, 2C when they are the halves of one matrix)
  __shared__ float part[2][LN_GROUPS][LN_COLS];
  const int p = threadIdx.x % LN_COLS, grp = threadIdx.x / LN_COLS;
  const long long n = blockIdx.x * (long long)LN_COLS + p;
  const bool ok = n < N;
  const float mean = ok ? mu[n] : 0.f, r = ok ? rs[n] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  // every thread walks its channels (also past-the-end positions, with zeros: the per-channel sums need whole half-waves)
  for (int c0 = grp; c0 < Cc; c0 += LN_UNROLL * LN_GROUPS) {
    float d[LN_UNROLL], xc[LN_UNROLL];
#pragma unroll
    for (int u = 0; u < LN_UNROLL; ++u) {
      const int c = c0 + u * LN_GROUPS;
      const bool in = ok && c < Cc;
      d[u] = in ? dy[(long long)c * N + n] : 0.f;
      xc[u] = in ? x[(long long)c * N + n] - mean : 0.f;
    }
#pragma unroll
    for (int u = 0; u < LN_UNROLL; ++u) {
      const int c = c0 + u * LN_GROUPS;
      if (c >= Cc) break;  // (uniform over the 32 lanes of a channel group)
      const float dyg = d[u] * g[c];
      s1 += dyg;
      s2 += dyg * xc[u];
      // per-channel sums over the block's 32 positions: a half-wave reduction in fixed order
      float a = d[u] * xc[u] * r, bsum = d[u];
#pragma unroll
      for (int off = LN_COLS / 2; off > 0; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        bsum += __shfl_xor(bsum, off, 64);
      }
      if (p == 0) {
        pg[(long long)blockIdx.x * prow + c] = a;
        pb[(long long)blockIdx.x * prow + c] = bsum;
      }
    }
  }
  part[0][grp][p] = s1;
  part[1][grp][p] = s2;
  __syncthreads();
  if (!ok) return;
  float t1 = 0.f, t2 = 0.f;
#pragma unroll
  for (int gI = 0; gI < LN_GROUPS; ++gI) {
    t1 += part[0][gI][p];
    t2 += part[1][gI][p];
  }
  const float m1 = t1 / (float)Cc, m2 = t2 / (float)Cc;
  const float sigma = 1.f / r - eps;
  const float k2 = sigma > 0.f ? r * r / sigma * m2 : 0.f;
  int c = grp;
  for (; c + (LN_UNROLL - 1) * LN_GROUPS < Cc; c += LN_UNROLL * LN_GROUPS) {
    float d[LN_UNROLL], xv[LN_UNROLL];
#pragma unroll
    for (int u = 0; u < LN_UNROLL; ++u) {
      const long long o = (long long)(c + u * LN_GROUPS) * N + n;
      d[u] = dy[o];
      xv[u] = x[o];
    }
    float av[LN_UNROLL];
#pragma unroll
    for (int u = 0; u < LN_UNROLL; ++u) av[u] = add ? add[(long long)(c + u * LN_GROUPS) * N + n] : 0.f;
#pragma unroll
    for (int u = 0; u < LN_UNROLL; ++u) {
      const int cc = c + u * LN_GROUPS;
      dx[(long long)cc * N + n] = r * (d[u] * g[cc] - m1) - k2 * (xv[u] - mean) + av[u];
    }
  }
  for (; c < Cc; c += LN_GROUPS) {
    const long long o = (long long)c * N + n;
    dx[o] = r * (dy[o] * g[c] - m1) - k2 * (x[o] - mean) + (add ? add[o] : 0.f);
  }
}

// The same backward for the shapes the scorer has (C = 64 / 128 / 1024, N a multiple of 4), re-blocked around what bounds it.  The
// kernel above gives every lane ONE position and a 16th of the channels: a dword per lane and access (the texture addresser takes
// 16 cycles per 256-byte wave-instruction: ~17 us per 512-KB block whatever HBM does), ten cross-lane shuffles per channel and
// thread for the per-channel sums, and dy / x read twice.  Here a lane owns FOUR consecutive positions (16-byte accesses: a
// quarter of the wave-instructions) and a 64th of the channels, with every value resident in registers between the statistics
// and the output (CPT = C / 64 float4 pairs per thread: 128 VGPRs at C = 1 024; a block has its CU to itself at N / 32 = 320
// blocks on 256 CUs), the per-channel sums fold the lane's four positions first (three shuffle steps over 8 lanes instead of five
// over 32), the per-position sums fold across the wave's 8 channel groups by shuffles and across the 8 waves through LDS.
// Same formulas; the sums associate differently from the kernel above (1e-7 relative).  Stage-2 launch (1 024 x 10 240, + skip
// gradient): profiles/r05_mgfn_passes.md.
constexpr int LV_POS = 32, LV_PL = 8, LV_GROUPS = 64, LV_THREADS = LV_PL * LV_GROUPS;  // 8 position quads x 64 channel groups
template <int CPT>
__global__ __launch_bounds__(LV_THREADS) void chan_layernorm_bwd_v4_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                           const float* __restrict__ g, const float* __restrict__ mu,
                                                                           const float* __restrict__ rs, float* __restrict__ dx,
                                                                           float* __restrict__ pg, float* __restrict__ pb, long long N,
                                                                           float eps, const float* __restrict__ add, long long prow) {
  constexpr int Cc = CPT * LV_GROUPS;
  __shared__ float part[2][LV_THREADS / 64][LV_POS];
  const int pl = threadIdx.x % LV_PL, grp = threadIdx.x / LV_PL, wave = threadIdx.x >> 6;
  const long long n = blockIdx.x * (long long)LV_POS + 4 * pl;  // N % 4 == 0: a quad is inside the tensor or past it
  const bool ok = n < N;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 mean = zero4, r = zero4, d[CPT], xc[CPT];
#pragma unroll
  for (int u = 0; u < CPT; ++u) d[u] = xc[u] = zero4;
  if (ok) {  // (a branch, not a select between addresses: the loads stay 16-byte global loads)
    mean = *reinterpret_cast<const float4*>(mu + n);
    r = *reinterpret_cast<const float4*>(rs + n);
    const float* dyp = dy + (long long)grp * N + n;
    const float* xp = x + (long long)grp * N + n;
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      d[u] = *reinterpret_cast<const float4*>(dyp + (long long)u * LV_GROUPS * N);
      xc[u] = *reinterpret_cast<const float4*>(xp + (long long)u * LV_GROUPS * N);
    }
  }
  float4 s1 = zero4, s2 = zero4;
#pragma unroll
  for (int u = 0; u < CPT; ++u) {
    const int c = grp + u * LV_GROUPS;
    const float gc = g[c];
    if (ok) { xc[u].x -= mean.x; xc[u].y -= mean.y; xc[u].z -= mean.z; xc[u].w -= mean.w; }
    const float4 dyg = make_float4(d[u].x * gc, d[u].y * gc, d[u].z * gc, d[u].w * gc);
    s1.x += dyg.x; s1.y += dyg.y; s1.z += dyg.z; s1.w += dyg.w;
    s2.x += dyg.x * xc[u].x; s2.y += dyg.y * xc[u].y; s2.z += dyg.z * xc[u].z; s2.w += dyg.w * xc[u].w;
    // dg / db of channel c over the block's 32 positions: the lane's four, then the 8 lanes of the channel group
    float a = ((d[u].x * xc[u].x * r.x + d[u].y * xc[u].y * r.y) + d[u].z * xc[u].z * r.z) + d[u].w * xc[u].w * r.w;
    float bsum = ((d[u].x + d[u].y) + d[u].z) + d[u].w;
#pragma unroll
    for (int off = LV_PL / 2; off > 0; off >>= 1) {
      a += __shfl_xor(a, off, 64);
      bsum += __shfl_xor(bsum, off, 64);
    }
    if (pl == 0) {
      pg[(long long)blockIdx.x * prow + c] = a;
      pb[(long long)blockIdx.x * prow + c] = bsum;
    }
  }
  // per-position sums over the channels: the wave's 8 channel groups by shuffles (lanes 8 apart), the 8 waves through LDS
#pragma unroll
  for (int off = LV_PL; off < 64; off <<= 1) {
    s1.x += __shfl_xor(s1.x, off, 64); s1.y += __shfl_xor(s1.y, off, 64); s1.z += __shfl_xor(s1.z, off, 64); s1.w += __shfl_xor(s1.w, off, 64);
    s2.x += __shfl_xor(s2.x, off, 64); s2.y += __shfl_xor(s2.y, off, 64); s2.z += __shfl_xor(s2.z, off, 64); s2.w += __shfl_xor(s2.w, off, 64);
  }
  if ((threadIdx.x & 63) < LV_PL) {
    *reinterpret_cast<float4*>(&part[0][wave][4 * pl]) = s1;
    *reinterpret_cast<float4*>(&part[1][wave][4 * pl]) = s2;
  }
  __syncthreads();
  if (!ok) return;
  float4 t1 = zero4, t2 = zero4;
#pragma unroll
  for (int w = 0; w < LV_THREADS / 64; ++w) {
    const float4 a1 = *reinterpret_cast<const float4*>(&part[0][w][4 * pl]), a2 = *reinterpret_cast<const float4*>(&part[1][w][4 * pl]);
    t1.x += a1.x; t1.y += a1.y; t1.z += a1.z; t1.w += a1.w;
    t2.x += a2.x; t2.y += a2.y; t2.z += a2.z; t2.w += a2.w;
  }
  const float inv = 1.f / (float)Cc;
  const float4 m1 = make_float4(t1.x * inv, t1.y * inv, t1.z * inv, t1.w * inv);
  float4 k2;
  {
    const float sx = 1.f / r.x - eps, sy = 1.f / r.y - eps, sz = 1.f / r.z - eps, sw = 1.f / r.w - eps;
    k2.x = sx > 0.f ? r.x * r.x / sx * (t2.x * inv) : 0.f;
    k2.y = sy > 0.f ? r.y * r.y / sy * (t2.y * inv) : 0.f;
    k2.z = sz > 0.f ? r.z * r.z / sz * (t2.z * inv) : 0.f;
    k2.w = sw > 0.f ? r.w * r.w / sw * (t2.w * inv) : 0.f;
  }
  constexpr int AB = CPT < 2 ? CPT : 2;  // `add` quads fetched per batch
#pragma unroll
  for (int u0 = 0; u0 < CPT; u0 += AB) {
    float4 av[AB];
#pragma unroll
    for (int u = 0; u < AB; ++u) {
      av[u] = zero4;
      if (add != nullptr) av[u] = *reinterpret_cast<const float4*>(add + (long long)(grp + (u0 + u) * LV_GROUPS) * N + n);
    }
#pragma unroll
    for (int u = 0; u < AB; ++u) {
      const int cc = grp + (u0 + u) * LV_GROUPS;
      const float gc = g[cc];
      const float4 dd = d[u0 + u], xx = xc[u0 + u];
      *reinterpret_cast<float4*>(dx + (long long)cc * N + n) =
          make_float4(r.x * (dd.x * gc - m1.x) - k2.x * xx.x + av[u].x, r.y * (dd.y * gc - m1.y) - k2.y * xx.y + av[u].y,
                      r.z * (dd.z * gc - m1.z) - k2.z * xx.z + av[u].z, r.w * (dd.w * gc - m1.w) - k2.w * xx.w + av[u].w);
    }
  }
}

// forward in the same blocking: y = (x - mu) * rs * g + b with the one-pass shifted statistics of ln_stats, x resident in registers
template <int CPT>
__global__ __launch_bounds__(LV_THREADS) void chan_layernorm_fwd_v4_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                                           float* __restrict__ y, float* __restrict__ mu, float* __restrict__ rs, long long N,
                                                                           float eps) {
  constexpr int Cc = CPT * LV_GROUPS;
  __shared__ float part[2][LV_THREADS / 64][LV_POS];
  const int pl = threadIdx.x % LV_PL, grp = threadIdx.x / LV_PL, wave = threadIdx.x >> 6;
  const long long n = blockIdx.x * (long long)LV_POS + 4 * pl;
  const bool ok = n < N;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 v[CPT], x0 = zero4;
#pragma unroll
  for (int u = 0; u < CPT; ++u) v[u] = zero4;
  if (ok) {
    x0 = *reinterpret_cast<const float4*>(x + n);
    const float* xp = x + (long long)grp * N + n;
#pragma unroll
    for (int u = 0; u < CPT; ++u) v[u] = *reinterpret_cast<const float4*>(xp + (long long)u * LV_GROUPS * N);
  }
  float4 s1 = zero4, s2 = zero4;
#pragma unroll
  for (int u = 0; u < CPT; ++u) {
    const float4 d = make_float4(v[u].x - x0.x, v[u].y - x0.y, v[u].z - x0.z, v[u].w - x0.w);
    s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
    s2.x += d.x * d.x; s2.y += d.y * d.y; s2.z += d.z * d.z; s2.w += d.w * d.w;
  }
#pragma unroll
  for (int off = LV_PL; off < 64; off <<= 1) {
    s1.x += __shfl_xor(s1.x, off, 64); s1.y += __shfl_xor(s1.y, off, 64); s1.z += __shfl_xor(s1.z, off, 64); s1.w += __shfl_xor(s1.w, off, 64);
    s2.x += __shfl_xor(s2.x, off, 64); s2.y += __shfl_xor(s2.y, off, 64); s2.z += __shfl_xor(s2.z, off, 64); s2.w += __shfl_xor(s2.w, off, 64);
  }
  if ((threadIdx.x & 63) < LV_PL) {
    *reinterpret_cast<float4*>(&part[0][wave][4 * pl]) = s1;
    *reinterpret_cast<float4*>(&part[1][wave][4 * pl]) = s2;
  }
  __syncthreads();
  if (!ok) return;
  float4 t1 = zero4, t2 = zero4;
#pragma unroll
  for (int w = 0; w < LV_THREADS / 64; ++w) {
    const float4 a1 = *reinterpret_cast<const float4*>(&part[0][w][4 * pl]), a2 = *reinterpret_cast<const float4*>(&part[1][w][4 * pl]);
    t1.x += a1.x; t1.y += a1.y; t1.z += a1.z; t1.w += a1.w;
    t2.x += a2.x; t2.y += a2.y; t2.z += a2.z; t2.w += a2.w;
  }
  const float inv = 1.f / (float)Cc;
  const float4 m1 = make_float4(t1.x * inv, t1.y * inv, t1.z * inv, t1.w * inv);
  const float4 mean = make_float4(x0.x + m1.x, x0.y + m1.y, x0.z + m1.z, x0.w + m1.w);
  const float4 r = make_float4(1.f / (sqrtf(fmaxf(t2.x * inv - m1.x * m1.x, 0.f)) + eps), 1.f / (sqrtf(fmaxf(t2.y * inv - m1.y * m1.y, 0.f)) + eps),
                               1.f / (sqrtf(fmaxf(t2.z * inv - m1.z * m1.z, 0.f)) + eps), 1.f / (sqrtf(fmaxf(t2.w * inv - m1.w * m1.w, 0.f)) + eps));
  if (grp == 0) {
    *reinterpret_cast<float4*>(mu + n) = mean;
    *reinterpret_cast<float4*>(rs + n) = r;
  }
  float* yp = y + (long long)grp * N + n;
#pragma unroll
  for (int u = 0; u < CPT; ++u) {
    const int c = grp + u * LV_GROUPS;
    const float gc = g[c], bc = b[c];
    *reinterpret_cast<float4*>(yp + (long long)u * LV_GROUPS * N) =
        make_float4((v[u].x - mean.x) * r.x * gc + bc, (v[u].y - mean.y) * r.y * gc + bc, (v[u].z - mean.z) * r.z * gc + bc, (v[u].w - mean.w) * r.w * gc + bc);
  }
}

static bool launch_ln_fwd_v4(const float* x, const float* g, const float* b, float* y, float* mu, float* rs, int C, long long N, float eps, hipStream_t st) {
  if (N % 4 != 0 || (((uintptr_t)x | (uintptr_t)y | (uintptr_t)mu | (uintptr_t)rs) & 15) != 0) return false;
  const dim3 grid((unsigned)((N + LV_POS - 1) / LV_POS)), block(LV_THREADS);
  switch (C) {
    case 1 * LV_GROUPS: hipLaunchKernelGGL(chan_layernorm_fwd_v4_kernel<1>, grid, block, 0, st, x, g, b, y, mu, rs, N, eps); return true;
    case 2 * LV_GROUPS: hipLaunchKernelGGL(chan_layernorm_fwd_v4_kernel<2>, grid, block, 0, st, x, g, b, y, mu, rs, N, eps); return true;
    case 16 * LV_GROUPS: hipLaunchKernelGGL(chan_layernorm_fwd_v4_kernel<16>, grid, block, 0, st, x, g, b, y, mu, rs, N, eps); return true;
    default: return false;
  }
}

static bool launch_ln_bwd_v4(const float* dy, const float* x, const float* g, const float* mu, const float* rs, float* dx, float* pg, float* pb, int C,
                             long long N, float eps, const float* add, long long prow, hipStream_t st) {
  static_assert(LV_POS == LN_COLS, "the partial-sum matrices have one row per LN_COLS positions (advhip_chan_layernorm_bwd_partial_rows)");
  if (N % 4 != 0 || (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)mu | (uintptr_t)rs | (uintptr_t)add) & 15) != 0) return false;
  const dim3 grid((unsigned)((N + LV_POS - 1) / LV_POS)), block(LV_THREADS);
  switch (C) {
    case 1 * LV_GROUPS: hipLaunchKernelGGL(chan_layernorm_bwd_v4_kernel<1>, grid, block, 0, st, dy, x, g, mu, rs, dx, pg, pb, N, eps, add, prow); return true;
    case 2 * LV_GROUPS: hipLaunchKernelGGL(chan_layernorm_bwd_v4_kernel<2>, grid, block, 0, st, dy, x, g, mu, rs, dx, pg, pb, N, eps, add, prow); return true;
    case 16 * LV_GROUPS: hipLaunchKernelGGL(chan_layernorm_bwd_v4_kernel<16>, grid, block, 0, st, dy, x, g, mu, rs, dx, pg, pb, N, eps, add, prow); return true;
    default: return false;
  }
}

// out[c, b, t] = bias[c % H] + sum_j w[c % H][j] * v[c, b, t + j - K/2]   (zero outside [0, T)); one thread per element
template <int K>
__global__ __launch_bounds__(256) void dwconv_t_fwd_kernel(const float* __restrict__ v, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out, int H, int T,
                                                           long long rows_per_c, long long total) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int t = (int)(i % T);
    const int h = (int)((i / (rows_per_c * T)) % H);
    float acc = bias[h];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int tt = t + j - K / 2;
      if (tt >= 0 && tt < T) acc += w[h * K + j] * v[i + j - K / 2];
    }
    out[i] = acc;
  }
}

// dv[c, b, t] = sum_j w[h][j] * dout[c, b, t - j + K/2];  per block (one channel c = c_idx * H + h, a chunk of its B*T elements):
// partial[c_idx * chunks + chunk][h][0..K-1] = sum dout[c,b,t] * v[c,b,t+j-K/2],  [..][h][K] = sum dout
template <int K>
__global__ __launch_bounds__(256) void dwconv_t_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ v,
                                                           const float* __restrict__ w, float* __restrict__ dv,
                                                           float* __restrict__ partial, int H, int T, long long per_c, int chunks) {
  __shared__ float red[4][K + 1];
  const int c = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  const int h = c % H;
  const long long lo = (per_c * chunk) / chunks, hi = (per_c * (chunk + 1)) / chunks;  // whole rows: per_c / chunks is a multiple of T
  float acc[K + 1];
#pragma unroll
  for (int j = 0; j <= K; ++j) acc[j] = 0.f;
  for (long long e = lo + threadIdx.x; e < hi; e += 256) {
    const long long i = (long long)c * per_c + e;
    const int t = (int)(e % T);
    const float d = dout[i];
    float g = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int tv = t + j - K / 2;   // v index this output touched with tap j
      if (tv >= 0 && tv < T) acc[j] += d * v[i + j - K / 2];
      const int td = t - j + K / 2;   // output that touched v[t] with tap j
      if (td >= 0 && td < T) g += w[h * K + j] * dout[i - j + K / 2];
    }
    acc[K] += d;
    dv[i] = g;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j <= K; ++j) {
    float a = acc[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if (lane == 0) red[wv][j] = a;
  }
  __syncthreads();
  // partial[(c / H) * chunks + chunk][h][K + 1]: the rows of one head's column block are what the caller adds up (advhip_colsum_f32)
  if (threadIdx.x <= K)
    partial[(((long long)(c / H) * chunks + chunk) * H + h) * (K + 1) + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// The same two kernels with a thread owning FOUR consecutive t of a row (T % 4 == 0, 16-byte accesses, one 64-bit division per
// quad instead of three per element): the window v[t - K/2 .. t + 3 + K/2] is the quad, its left and its right neighbour quad
// (zero outside the row).  Same sums per output; the per-block partial sums of the backward associate differently (1e-7).
__device__ __forceinline__ void dw_window(const float* __restrict__ p, long long e, int t, int T, float (&win)[12]) {
  const float4 c = *reinterpret_cast<const float4*>(p + e);
  float4 l = make_float4(0.f, 0.f, 0.f, 0.f), r = l;
  if (t > 0) l = *reinterpret_cast<const float4*>(p + e - 4);
  if (t + 4 < T) r = *reinterpret_cast<const float4*>(p + e + 4);
  win[0] = l.x; win[1] = l.y; win[2] = l.z; win[3] = l.w;
  win[4] = c.x; win[5] = c.y; win[6] = c.z; win[7] = c.w;
  win[8] = r.x; win[9] = r.y; win[10] = r.z; win[11] = r.w;
}

template <int K>
__global__ __launch_bounds__(256) void dwconv_t_fwd_v4_kernel(const float* __restrict__ v, const float* __restrict__ w, const float* __restrict__ bias,
                                                              float* __restrict__ out, int H, int T, long long rows_per_c, long long total4) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const long long e = i * 4;
    const int t = (int)(e % T);
    const int h = (int)((e / (rows_per_c * T)) % H);
    float win[12], wk[K];
    dw_window(v, e, t, T, win);
#pragma unroll
    for (int j = 0; j < K; ++j) wk[j] = w[h * K + j];
    const float b = bias[h];
    float o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float acc = b;
#pragma unroll
      for (int j = 0; j < K; ++j) acc += wk[j] * win[4 + q + j - K / 2];  // (zero outside the row: the same sum as the scalar kernel's taps in range)
      o[q] = acc;
    }
    *reinterpret_cast<float4*>(out + e) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

template <int K>
__global__ __launch_bounds__(256) void dwconv_t_bwd_v4_kernel(const float* __restrict__ dout, const float* __restrict__ v, const float* __restrict__ w,
                                                              float* __restrict__ dv, float* __restrict__ partial, int H, int T, long long per_c,
                                                              int chunks) {
  __shared__ float red[4][K + 1];
  const int c = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  const int h = c % H;
  const long long lo = (per_c * chunk) / chunks, hi = (per_c * (chunk + 1)) / chunks;  // whole rows: multiples of T, hence of 4
  float acc[K + 1], wk[K];
#pragma unroll
  for (int j = 0; j <= K; ++j) acc[j] = 0.f;
#pragma unroll
  for (int j = 0; j < K; ++j) wk[j] = w[h * K + j];
  for (long long e = lo + 4ll * threadIdx.x; e < hi; e += 1024) {
    const long long i = (long long)c * per_c + e;
    const int t = (int)(e % T);
    float dw[12], vw[12], g[4];
    dw_window(dout, i, t, T, dw);
    dw_window(v, i, t, T, vw);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float gq = 0.f;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        acc[j] += dw[4 + q] * vw[4 + q + j - K / 2];  // tap j of output t + q touched v[t + q + j - K/2]
        gq += wk[j] * dw[4 + q - j + K / 2];          // v[t + q] was touched with tap j by output t + q - j + K/2
      }
      acc[K] += dw[4 + q];
      g[q] = gq;
    }
    *reinterpret_cast<float4*>(dv + i) = make_float4(g[0], g[1], g[2], g[3]);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j <= K; ++j) {
    float a = acc[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if (lane == 0) red[wv][j] = a;
  }
  __syncthreads();
  if (threadIdx.x <= K)
    partial[(((long long)(c / H) * chunks + chunk) * H + h) * (K + 1) + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// The rows a k = 3, padding 1 temporal conv contracts with, for its weight gradient dW = dY . U^T:
//   U[(c * 3 + j), r, t] = x[c, r, t + j - 1]  (0 outside [0, T)), x (C, rows, T) -> U (3 C, rows, T), tap-minor like the weights.
// One thread per float4 of x's row: reads it (+ one neighbour on each side) and writes the three shifted copies.
__global__ __launch_bounds__(256) void unfold3_kernel(const float* __restrict__ x, float* __restrict__ u, int T, long long per_c,
                                                      long long total4) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const long long e = i * 4;               // first element of this float4 (T % 4 == 0: a float4 never straddles a row)
    const long long c = e / per_c, w = e - c * per_c;
    const int t = (int)(w % T);
    const float4 v = *reinterpret_cast<const float4*>(x + e);
    const float l = t > 0 ? x[e - 1] : 0.f, r = t + 4 < T ? x[e + 4] : 0.f;
    float* o = u + (c * 3) * per_c + w;
    *reinterpret_cast<float4*>(o) = make_float4(l, v.x, v.y, v.z);                 // tap 0: x[t - 1]
    *reinterpret_cast<float4*>(o + per_c) = v;                                     // tap 1
    *reinterpret_cast<float4*>(o + 2 * per_c) = make_float4(v.y, v.z, v.w, r);     // tap 2: x[t + 1]
  }
}

// ---- BatchNorm1d over the rows of a (C, N) activation (FocusAttention.norm, modeling_mgfn.py:162, 174) -------------------
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();  // red may still be read from a previous call
  if (lane == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) t += red[i];
  return t;
}

// training mode: batch statistics per channel (one block per channel row); y = (x - mean) * rstd * gamma + beta;
// mean / biased var are returned (running-statistics update and backward)
__global__ __launch_bounds__(256) void bn_rows_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y,
                                                          float* __restrict__ mean_out, float* __restrict__ var_out, long long N,
                                                          float eps, float* __restrict__ run_mean, float* __restrict__ run_var,
                                                          float momentum) {
  // run_mean / run_var (nullable, together): nn.BatchNorm1d's running statistics, updated in place as torch does
  // (running = (1 - momentum) * running + momentum * batch; the variance with the unbiased N / (N - 1) factor)
  __shared__ float red[4];
  const int c = blockIdx.x;
  const float* xr = x + (long long)c * N;
  float s = 0.f;
  for (long long i = threadIdx.x; i < N; i += 256) s += xr[i];
  const float mean = block_sum_256(s, red) / (float)N;
  float q = 0.f;
  for (long long i = threadIdx.x; i < N; i += 256) {
    const float d = xr[i] - mean;
    q += d * d;
  }
  const float var = block_sum_256(q, red) / (float)N;
  const float sc = gamma[c] * rsqrtf(var + eps), sh = beta[c] - mean * sc;
  float* yr = y + (long long)c * N;
  for (long long i = threadIdx.x; i < N; i += 256) yr[i] = xr[i] * sc + sh;
  if (threadIdx.x == 0) {
    mean_out[c] = mean;
    var_out[c] = var;
    if (run_mean != nullptr) {
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * (var * ((float)N / (float)(N > 1 ? N - 1 : 1)));
    }
  }
}

// The same forward for rows of up to NT x 4 x BQ floats (N % 4 == 0): the thread's share of the row -- BQ float4 -- stays in
// registers between the mean, the variance and the output: ONE read of x (the kernel above: three).  NT = 1 024 threads for the
// narrow layers (C = 128 rows are 128 blocks on 256 CUs: the block's own width is what parallelism there is).
template <int NT, int BQ>
__global__ __launch_bounds__(NT) void bn_rows_fwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ var_out,
                                                              long long N, float eps, float* __restrict__ run_mean, float* __restrict__ run_var,
                                                              float momentum) {
  __shared__ float red[NT / 64];
  const int c = blockIdx.x;
  const float* xr = x + (long long)c * N;
  const long long nq = N / 4;
  float4 v[BQ];
#pragma unroll
  for (int u = 0; u < BQ; ++u) {
    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    const long long q = threadIdx.x + (long long)NT * u;
    if (q < nq) v[u] = *reinterpret_cast<const float4*>(xr + 4 * q);
  }
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < BQ; ++u) s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
  const float mean = block_sum<NT>(s, red) / (float)N;
  float q2 = 0.f;
#pragma unroll
  for (int u = 0; u < BQ; ++u) {
    if (threadIdx.x + (long long)NT * u < nq) {
      const float d0 = v[u].x - mean, d1 = v[u].y - mean, d2 = v[u].z - mean, d3 = v[u].w - mean;
      q2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
  const float var = block_sum<NT>(q2, red) / (float)N;
  const float sc = gamma[c] * rsqrtf(var + eps), sh = beta[c] - mean * sc;
  float* yr = y + (long long)c * N;
#pragma unroll
  for (int u = 0; u < BQ; ++u) {
    const long long q = threadIdx.x + (long long)NT * u;
    if (q < nq) *reinterpret_cast<float4*>(yr + 4 * q) = make_float4(v[u].x * sc + sh, v[u].y * sc + sh, v[u].z * sc + sh, v[u].w * sc + sh);
  }
  if (threadIdx.x == 0) {
    mean_out[c] = mean;
    var_out[c] = var;
    if (run_mean != nullptr) {
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * (var * ((float)N / (float)(N > 1 ? N - 1 : 1)));
    }
  }
}

static void launch_bn_rows_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* var, int C, long long N, float eps,
                               float* run_mean, float* run_var, float momentum, hipStream_t st) {
  const bool vec = N % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
  const dim3 grid((unsigned)C);
  if (vec && C < 512 && N <= 1024 * 4 * 4)
    hipLaunchKernelGGL((bn_rows_fwd_reg_kernel<1024, 4>), grid, dim3(1024), 0, st, x, gamma, beta, y, mean, var, N, eps, run_mean, run_var, momentum);
  else if (vec && N <= 256 * 4 * 4)
    hipLaunchKernelGGL((bn_rows_fwd_reg_kernel<256, 4>), grid, dim3(256), 0, st, x, gamma, beta, y, mean, var, N, eps, run_mean, run_var, momentum);
  else if (vec && N <= 256 * 4 * 12)
    hipLaunchKernelGGL((bn_rows_fwd_reg_kernel<256, 12>), grid, dim3(256), 0, st, x, gamma, beta, y, mean, var, N, eps, run_mean, run_var, momentum);
  else
    hipLaunchKernelGGL(bn_rows_fwd_kernel, grid, dim3(256), 0, st, x, gamma, beta, y, mean, var, N, eps, run_mean, run_var, momentum);
}

// dx = gamma * rstd * (dy - mean(dy) - xhat * mean(dy * xhat)),  dgamma = sum dy * xhat,  dbeta = sum dy
__global__ __launch_bounds__(256) void bn_rows_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ gamma, const float* __restrict__ mean,
                                                          const float* __restrict__ var, float* __restrict__ dx,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, long long N, float eps,
                                                          const float* __restrict__ add) {
  // `add` (nullable, dx's shape): dx = BatchNorm backward + add (the skip connection's gradient, as in chan_layernorm_bwd)
  __shared__ float red[4];
  const int c = blockIdx.x;
  const float* xr = x + (long long)c * N;
  const float* dr = dy + (long long)c * N;
  const float mu = mean[c], rstd = rsqrtf(var[c] + eps);
  float s1 = 0.f, s2 = 0.f;
  for (long long i = threadIdx.x; i < N; i += 256) {
    const float d = dr[i];
    s1 += d;
    s2 += d * (xr[i] - mu) * rstd;
  }
  const float sum_dy = block_sum_256(s1, red);
  const float sum_dyx = block_sum_256(s2, red);
  const float k = gamma[c] * rstd, m1 = sum_dy / (float)N, m2 = sum_dyx / (float)N;
  float* o = dx + (long long)c * N;
  const float* ar = add ? add + (long long)c * N : nullptr;
  for (long long i = threadIdx.x; i < N; i += 256) o[i] = k * (dr[i] - m1 - (xr[i] - mu) * rstd * m2) + (ar ? ar[i] : 0.f);
  if (threadIdx.x == 0) { dgamma[c] = sum_dyx; dbeta[c] = sum_dy; }
}

// dst[c] = sum over r (in row order) of src[r][c] for a (rows, cols) matrix of per-block partial sums (LayerNorm dg | db, the head's
// four parameter gradients, the depth-wise conv's filter gradients): 32 columns x 8 row groups per block, each thread walks its
// rows eight loads at a time, the row groups are added in fixed order through LDS.  torch's generic reduction takes 10-20 us on
// these few-hundred-row matrices; 26 of them per training step.
// `period` > 0: the columns are [cols / period groups][period] and the sums leave de-interleaved -- element j < period - 1 of group h to
// dst[h * (period - 1) + j], the last element of every group to dst[(cols / period) * (period - 1) + h] (the depth-wise conv's per-head
// (filter taps | bias) sums -> the filter gradient, then the bias gradient, each contiguous)
__device__ __forceinline__ void colsum_body(const float* __restrict__ src, float* __restrict__ dst, long long rows, int cols, int block, int period) {
  __shared__ float part[8][33];
  const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int c = block * 32 + cl;
  float acc = 0.f;
  if (c < cols) {
    long long r = grp;
    for (; r + 56 < rows; r += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(r + 8 * u) * cols + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; r < rows; r += 8) acc += src[r * cols + c];
  }
  part[grp][cl] = acc;
  __syncthreads();
  if (grp == 0 && c < cols) {
    float t = part[0][cl];
#pragma unroll
    for (int gI = 1; gI < 8; ++gI) t += part[gI][cl];
    if (period > 0) {
      const int h = c / period, j = c - h * period;
      dst[j < period - 1 ? h * (period - 1) + j : (cols / period) * (period - 1) + h] = t;
    } else {
      dst[c] = t;
    }
  }
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ src, float* __restrict__ dst, long long rows, int cols) {
  colsum_body(src, dst, rows, cols, (int)blockIdx.x, 0);
}

// Many such matrices in one launch (items in the kernel arguments): every partial-sum matrix a backward pass left behind, at its end.
constexpr int COLSUM_GROUP_MAX = 64;
struct ColsumGroupItem {
  const float* src;
  float* dst;
  long long rows;
  int cols, block_begin, period, pad;
};
struct ColsumGroupArgs {
  ColsumGroupItem it[COLSUM_GROUP_MAX];
  int n;
};
__global__ __launch_bounds__(256) void colsum_group_kernel(const ColsumGroupArgs ga) {
  int i = 0;
  while (i + 1 < ga.n && (int)blockIdx.x >= ga.it[i + 1].block_begin) ++i;
  const ColsumGroupItem& t = ga.it[i];
  colsum_body(t.src, t.dst, t.rows, t.cols, (int)blockIdx.x - t.block_begin, t.period);
}

// bn_rows_bwd_kernel with the thread's share of dy and x resident in registers (rows of up to NT x 4 x BQ floats, N % 4 == 0): one
// read of each instead of two; NT = 1 024 for the narrow layers, as in the forward.
template <int NT, int BQ>
__global__ __launch_bounds__(NT) void bn_rows_bwd_reg_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ mean, const float* __restrict__ var, float* __restrict__ dx,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, long long N, float eps,
                                                             const float* __restrict__ add) {
  __shared__ float red[NT / 64];
  const int c = blockIdx.x;
  const float* xr = x + (long long)c * N;
  const float* dr = dy + (long long)c * N;
  const long long nq = N / 4;
  const float mu = mean[c], rstd = rsqrtf(var[c] + eps);
  float4 d[BQ], h[BQ];
#pragma unroll
  for (int u = 0; u < BQ; ++u) {
    d[u] = h[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    const long long q = threadIdx.x + (long long)NT * u;
    if (q < nq) {
      d[u] = *reinterpret_cast<const float4*>(dr + 4 * q);
      h[u] = *reinterpret_cast<const float4*>(xr + 4 * q);
    }
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int u = 0; u < BQ; ++u) {
    if (threadIdx.x + (long long)NT * u < nq) {
      h[u] = make_float4((h[u].x - mu) * rstd, (h[u].y - mu) * rstd, (h[u].z - mu) * rstd, (h[u].w - mu) * rstd);
      s1 += (d[u].x + d[u].y) + (d[u].z + d[u].w);
      s2 += (d[u].x * h[u].x + d[u].y * h[u].y) + (d[u].z * h[u].z + d[u].w * h[u].w);
    }
  }
  const float sum_dy = block_sum<NT>(s1, red);
  const float sum_dyx = block_sum<NT>(s2, red);
  const float k = gamma[c] * rstd, m1 = sum_dy / (float)N, m2 = sum_dyx / (float)N;
  float* o = dx + (long long)c * N;
#pragma unroll
  for (int u = 0; u < BQ; ++u) {
    const long long q = threadIdx.x + (long long)NT * u;
    if (q < nq) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      if (add != nullptr) a = *reinterpret_cast<const float4*>(add + (long long)c * N + 4 * q);
      *reinterpret_cast<float4*>(o + 4 * q) = make_float4(k * (d[u].x - m1 - h[u].x * m2) + a.x, k * (d[u].y - m1 - h[u].y * m2) + a.y,
                                                          k * (d[u].z - m1 - h[u].z * m2) + a.z, k * (d[u].w - m1 - h[u].w * m2) + a.w);
    }
  }
  if (threadIdx.x == 0) {
    dgamma[c] = sum_dyx;
    dbeta[c] = sum_dy;
  }
}

static void launch_bn_rows_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* var, float* dx, float* dgamma,
                               float* dbeta, int C, long long N, float eps, const float* add, hipStream_t st) {
  const bool vec = N % 4 == 0 && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)add) & 15) == 0;
  const dim3 grid((unsigned)C);
  if (vec && C < 512 && N <= 1024 * 4 * 4)
    hipLaunchKernelGGL((bn_rows_bwd_reg_kernel<1024, 4>), grid, dim3(1024), 0, st, dy, x, gamma, mean, var, dx, dgamma, dbeta, N, eps, add);
  else if (vec && N <= 256 * 4 * 4)
    hipLaunchKernelGGL((bn_rows_bwd_reg_kernel<256, 4>), grid, dim3(256), 0, st, dy, x, gamma, mean, var, dx, dgamma, dbeta, N, eps, add);
  else if (vec && N <= 256 * 4 * 12)
    hipLaunchKernelGGL((bn_rows_bwd_reg_kernel<256, 12>), grid, dim3(256), 0, st, dy, x, gamma, mean, var, dx, dgamma, dbeta, N, eps, add);
  else
    hipLaunchKernelGGL(bn_rows_bwd_kernel, grid, dim3(256), 0, st, dy, x, gamma, mean, var, dx, dgamma, dbeta, N, eps, add);
}

// ---- the scorer's head on the body's layout (modeling_mgfn.py:387-389: permute -> nn.LayerNorm(C) -> Linear(C, 1) -> sigmoid) ----
// y (C, N) [the body's layout] -> xn (N, C) = (y - mean_c) * rsqrt(var_c + eps) * g + b  [the layout the MIL head reads rows of],
// score[n] = sigmoid(xn[n, :] . w + b0).  One pass: the (C, N) -> (N, C) transposition goes through a 32 x 128 LDS tile, so both
// the reads (along n) and the writes (along c) are coalesced; torch: a 42-MB permute copy, LayerNorm, a GEMV, a sigmoid.
constexpr int HD_CHUNK = 128;
__global__ __launch_bounds__(LN_THREADS) void head_ln_fc_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                    const float* __restrict__ b, const float* __restrict__ w, const float* __restrict__ b0,
                                                                    float* __restrict__ xn, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                    float* __restrict__ score, int Cc, long long N, float eps) {
  __shared__ float part[2][LN_GROUPS][LN_COLS];
  __shared__ float tile[LN_COLS][HD_CHUNK + 1];
  const int p = threadIdx.x % LN_COLS, grp = threadIdx.x / LN_COLS;
  const long long n0 = blockIdx.x * (long long)LN_COLS, n = n0 + p;
  const bool ok = n < N;
  // statistics in one read, shifted by the first channel's value (as ln_stats), nn.LayerNorm's rsqrt(var + eps)
  const float x0 = ok ? x[n] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  if (ok)
    for (int c = grp; c < Cc; c += LN_GROUPS) {
      const float d = x[(long long)c * N + n] - x0;
      s1 += d;
      s2 += d * d;
    }
  part[0][grp][p] = s1;
  part[1][grp][p] = s2;
  __syncthreads();
  float t1 = 0.f, t2 = 0.f;
#pragma unroll
  for (int gI = 0; gI < LN_GROUPS; ++gI) { t1 += part[0][gI][p]; t2 += part[1][gI][p]; }
  const float m1 = t1 / (float)Cc, mean = x0 + m1;
  const float r = rsqrtf(fmaxf(t2 / (float)Cc - m1 * m1, 0.f) + eps);
  if (ok && grp == 0) { mean_out[n] = mean; rstd_out[n] = r; }
  float dot = 0.f;
  for (int c0 = 0; c0 < Cc; c0 += HD_CHUNK) {
#pragma unroll
    for (int u = 0; u < HD_CHUNK / LN_GROUPS; ++u) {
      const int cl = grp + LN_GROUPS * u, c = c0 + cl;
      float v = 0.f;
      if (ok && c < Cc) {
        v = (x[(long long)c * N + n] - mean) * r * g[c] + b[c];
        dot += v * w[c];
      }
      tile[p][cl] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < LN_COLS * HD_CHUNK; e += LN_THREADS) {
      const int row = e / HD_CHUNK, col = e % HD_CHUNK;
      if (n0 + row < N && c0 + col < Cc) xn[(n0 + row) * Cc + c0 + col] = tile[row][col];
    }
    __syncthreads();
  }
  part[0][grp][p] = dot;
  __syncthreads();
  if (ok && grp == 0) {
    float z = b0[0];
#pragma unroll
    for (int gI = 0; gI < LN_GROUPS; ++gI) z += part[0][gI][p];
    score[n] = 1.f / (1.f + expf(-z));
  }
}

// backward: dy[n][c] = dxn[n][c] + dscore[n] s (1 - s) w[c];  dx = r (dy g - mean_c(dy g) - xhat mean_c(dy g xhat));
// per-block partial sums [blocks][3 C + 1]: dg = sum dy xhat | db = sum dy | dw = sum dlogit xn | db0 = sum dlogit
__global__ __launch_bounds__(LN_THREADS) void head_ln_fc_bwd_kernel(const float* __restrict__ dxn, const float* __restrict__ dscore,
                                                                    const float* __restrict__ x, const float* __restrict__ g,
                                                                    const float* __restrict__ b, const float* __restrict__ w,
                                                                    const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                                    const float* __restrict__ score, float* __restrict__ dx,
                                                                    float* __restrict__ partial, int Cc, long long N) {
  __shared__ float part[2][LN_GROUPS][LN_COLS];
  __shared__ float tile[LN_COLS][HD_CHUNK + 1];
  const int p = threadIdx.x % LN_COLS, grp = threadIdx.x / LN_COLS;
  const long long n0 = blockIdx.x * (long long)LN_COLS, n = n0 + p;
  const bool ok = n < N;
  const float mean = ok ? mean_in[n] : 0.f, r = ok ? rstd_in[n] : 0.f;
  float dl = 0.f;
  if (ok) {
    const float sg = score[n];
    dl = (dscore ? dscore[n] : 0.f) * sg * (1.f - sg);
  }
  float* prow = partial + (long long)blockIdx.x * (3 * Cc + 1);
  auto load_tile = [&](int c0) {
    for (int e = threadIdx.x; e < LN_COLS * HD_CHUNK; e += LN_THREADS) {
      const int row = e / HD_CHUNK, col = e % HD_CHUNK;
      tile[row][col] = (dxn && n0 + row < N && c0 + col < Cc) ? dxn[(n0 + row) * Cc + c0 + col] : 0.f;
    }
  };
  float s1 = 0.f, s2 = 0.f;
  for (int c0 = 0; c0 < Cc; c0 += HD_CHUNK) {
    load_tile(c0);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < HD_CHUNK / LN_GROUPS; ++u) {
      const int cl = grp + LN_GROUPS * u, c = c0 + cl;
      if (c >= Cc) break;  // (uniform over the 32 lanes of a channel group)
      float dy = 0.f, xh = 0.f, dwv = 0.f;
      if (ok) {
        xh = (x[(long long)c * N + n] - mean) * r;
        dy = tile[p][cl] + dl * w[c];
        dwv = dl * (xh * g[c] + b[c]);
      }
      const float dyg = dy * g[c];
      s1 += dyg;
      s2 += dyg * xh;
      float a = dy * xh, bs = dy, cw = dwv;  // per-channel sums over the block's 32 positions: a half-wave reduction
#pragma unroll
      for (int off = LN_COLS / 2; off > 0; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        bs += __shfl_xor(bs, off, 64);
        cw += __shfl_xor(cw, off, 64);
      }
      if (p == 0) {
        prow[c] = a;
        prow[Cc + c] = bs;
        prow[2 * Cc + c] = cw;
      }
    }
    __syncthreads();
  }
  if (grp == 0) {  // db0: sum of dlogit over the block's positions
    float t = dl;
#pragma unroll
    for (int off = LN_COLS / 2; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    if (p == 0) prow[3 * Cc] = t;
  }
  part[0][grp][p] = s1;
  part[1][grp][p] = s2;
  __syncthreads();
  float t1 = 0.f, t2 = 0.f;
#pragma unroll
  for (int gI = 0; gI < LN_GROUPS; ++gI) { t1 += part[0][gI][p]; t2 += part[1][gI][p]; }
  const float m1 = t1 / (float)Cc, m2 = t2 / (float)Cc;
  for (int c0 = 0; c0 < Cc; c0 += HD_CHUNK) {
    __syncthreads();
    load_tile(c0);
    __syncthreads();
    if (!ok) continue;
#pragma unroll
    for (int u = 0; u < HD_CHUNK / LN_GROUPS; ++u) {
      const int cl = grp + LN_GROUPS * u, c = c0 + cl;
      if (c >= Cc) break;
      const long long o = (long long)c * N + n;
      const float xh = (x[o] - mean) * r;
      const float dy = tile[p][cl] + dl * w[c];
      dx[o] = r * (dy * g[c] - m1 - xh * m2);
    }
  }
}

// ---- the head, re-blocked like chan_layernorm_bwd_v4_kernel (a lane owns four consecutive positions and C / 64 channels, every
// value of y resident in registers: ONE read of y in 16-byte accesses; the (C, N) <-> (N, C) transposition through a 32 x 64 LDS tile
// per 64-channel chunk, rows written / read as 16-byte pieces).  C = 64 * CPT, N % 4 == 0, 16-byte aligned operands.
constexpr int HV_PAD = LV_GROUPS + 4;  // tile row pitch in floats (16-byte aligned rows)

// the lane's four per-position partial sums, over the wave's 8 channel groups (lanes 8 apart) and the block's 8 waves
__device__ __forceinline__ float4 lv_position_sum(float4 v, float (*part)[LV_POS], int pl, int wave) {
#pragma unroll
  for (int off = LV_PL; off < 64; off <<= 1) {
    v.x += __shfl_xor(v.x, off, 64); v.y += __shfl_xor(v.y, off, 64); v.z += __shfl_xor(v.z, off, 64); v.w += __shfl_xor(v.w, off, 64);
  }
  __syncthreads();  // (part may still be read from a previous call)
  if ((threadIdx.x & 63) < LV_PL) *reinterpret_cast<float4*>(&part[wave][4 * pl]) = v;
  __syncthreads();
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int w = 0; w < LV_THREADS / 64; ++w) {
    const float4 a = *reinterpret_cast<const float4*>(&part[w][4 * pl]);
    t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
  }
  return t;
}

template <int CPT>
__global__ __launch_bounds__(LV_THREADS) void head_ln_fc_fwd_v4_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                                       const float* __restrict__ w, const float* __restrict__ b0, float* __restrict__ xn,
                                                                       float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                       float* __restrict__ score, long long N, float eps) {
  constexpr int Cc = CPT * LV_GROUPS;
  __shared__ float part[LV_THREADS / 64][LV_POS];
  __shared__ __attribute__((aligned(16))) float tile[LV_POS][HV_PAD];
  const int pl = threadIdx.x % LV_PL, grp = threadIdx.x / LV_PL, wave = threadIdx.x >> 6;
  const long long n0 = blockIdx.x * (long long)LV_POS, n = n0 + 4 * pl;
  const bool ok = n < N;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 v[CPT], x0 = zero4;
#pragma unroll
  for (int u = 0; u < CPT; ++u) v[u] = zero4;
  if (ok) {
    x0 = *reinterpret_cast<const float4*>(x + n);  // channel 0: the shift of the one-pass statistics (as ln_stats)
    const float* xp = x + (long long)grp * N + n;
#pragma unroll
    for (int u = 0; u < CPT; ++u) v[u] = *reinterpret_cast<const float4*>(xp + (long long)u * LV_GROUPS * N);
  }
  float4 s1 = zero4, s2 = zero4;
#pragma unroll
  for (int u = 0; u < CPT; ++u) {
    const float4 d = make_float4(v[u].x - x0.x, v[u].y - x0.y, v[u].z - x0.z, v[u].w - x0.w);
    s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
    s2.x += d.x * d.x; s2.y += d.y * d.y; s2.z += d.z * d.z; s2.w += d.w * d.w;
  }
  const float4 t1 = lv_position_sum(s1, part, pl, wave), t2 = lv_position_sum(s2, part, pl, wave);
  const float inv = 1.f / (float)Cc;
  const float4 m1 = make_float4(t1.x * inv, t1.y * inv, t1.z * inv, t1.w * inv);
  const float4 mean = make_float4(x0.x + m1.x, x0.y + m1.y, x0.z + m1.z, x0.w + m1.w);
  const float4 r = make_float4(rsqrtf(fmaxf(t2.x * inv - m1.x * m1.x, 0.f) + eps), rsqrtf(fmaxf(t2.y * inv - m1.y * m1.y, 0.f) + eps),
                               rsqrtf(fmaxf(t2.z * inv - m1.z * m1.z, 0.f) + eps), rsqrtf(fmaxf(t2.w * inv - m1.w * m1.w, 0.f) + eps));
  if (ok && grp == 0) {
    *reinterpret_cast<float4*>(mean_out + n) = mean;
    *reinterpret_cast<float4*>(rstd_out + n) = r;
  }
  float4 dot = zero4;
  const int orow = threadIdx.x / 16, ocol = (threadIdx.x % 16) * 4;  // this thread's 16-byte piece of the 32 x 64 tile on the way out
#pragma unroll
  for (int u = 0; u < CPT; ++u) {
    const int c = grp + u * LV_GROUPS;
    const float gc = g[c], bc = b[c], wc = w[c];
    const float4 o = make_float4((v[u].x - mean.x) * r.x * gc + bc, (v[u].y - mean.y) * r.y * gc + bc, (v[u].z - mean.z) * r.z * gc + bc,
                                 (v[u].w - mean.w) * r.w * gc + bc);
    if (ok) { dot.x += o.x * wc; dot.y += o.y * wc; dot.z += o.z * wc; dot.w += o.w * wc; }
    __syncthreads();  // (the previous chunk's readers are done)
    tile[4 * pl + 0][grp] = o.x;
    tile[4 * pl + 1][grp] = o.y;
    tile[4 * pl + 2][grp] = o.z;
    tile[4 * pl + 3][grp] = o.w;
    __syncthreads();
    if (n0 + orow < N) *reinterpret_cast<float4*>(xn + (n0 + orow) * Cc + u * LV_GROUPS + ocol) = *reinterpret_cast<const float4*>(&tile[orow][ocol]);
  }
  const float4 z = lv_position_sum(dot, part, pl, wave);
  if (ok && grp == 0) {
    const float bb = b0[0];
    *reinterpret_cast<float4*>(score + n) = make_float4(1.f / (1.f + expf(-(z.x + bb))), 1.f / (1.f + expf(-(z.y + bb))), 1.f / (1.f + expf(-(z.z + bb))),
                                                        1.f / (1.f + expf(-(z.w + bb))));
  }
}

template <int CPT>
__global__ __launch_bounds__(LV_THREADS) void head_ln_fc_bwd_v4_kernel(const float* __restrict__ dxn, const float* __restrict__ dscore,
                                                                       const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                                       const float* __restrict__ w, const float* __restrict__ mean_in,
                                                                       const float* __restrict__ rstd_in, const float* __restrict__ score,
                                                                       float* __restrict__ dx, float* __restrict__ partial, long long N) {
  constexpr int Cc = CPT * LV_GROUPS;
  __shared__ float part[LV_THREADS / 64][LV_POS];
  __shared__ __attribute__((aligned(16))) float tile[LV_POS][HV_PAD];
  const int pl = threadIdx.x % LV_PL, grp = threadIdx.x / LV_PL, wave = threadIdx.x >> 6;
  const long long n0 = blockIdx.x * (long long)LV_POS, n = n0 + 4 * pl;
  const bool ok = n < N;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 mean = zero4, r = zero4, dl = zero4, xh[CPT], dy[CPT];
#pragma unroll
  for (int u = 0; u < CPT; ++u) xh[u] = dy[u] = zero4;
  if (ok) {
    mean = *reinterpret_cast<const float4*>(mean_in + n);
    r = *reinterpret_cast<const float4*>(rstd_in + n);
    if (dscore != nullptr) {
      const float4 sg = *reinterpret_cast<const float4*>(score + n), ds = *reinterpret_cast<const float4*>(dscore + n);
      dl = make_float4(ds.x * sg.x * (1.f - sg.x), ds.y * sg.y * (1.f - sg.y), ds.z * sg.z * (1.f - sg.z), ds.w * sg.w * (1.f - sg.w));
    }
    const float* xp = x + (long long)grp * N + n;
#pragma unroll
    for (int u = 0; u < CPT; ++u) xh[u] = *reinterpret_cast<const float4*>(xp + (long long)u * LV_GROUPS * N);
  }
  float* prow = partial + (long long)blockIdx.x * (3 * Cc + 1);
  const int irow = threadIdx.x / 16, icol = (threadIdx.x % 16) * 4;  // this thread's 16-byte piece of the 32 x 64 tile on the way in
  float4 s1 = zero4, s2 = zero4;
#pragma unroll
  for (int u = 0; u < CPT; ++u) {
    const int c = grp + u * LV_GROUPS;
    const float gc = g[c], bc = b[c], wc = w[c];
    float4 din = zero4;
    if (dxn != nullptr) {  // d_xn (N, C) -> the lane's (channel, four positions) through the tile
      __syncthreads();
      float4 piece = zero4;
      if (n0 + irow < N) piece = *reinterpret_cast<const float4*>(dxn + (n0 + irow) * Cc + u * LV_GROUPS + icol);
      *reinterpret_cast<float4*>(&tile[irow][icol]) = piece;
      __syncthreads();
      din = make_float4(tile[4 * pl + 0][grp], tile[4 * pl + 1][grp], tile[4 * pl + 2][grp], tile[4 * pl + 3][grp]);
    }
    float4 h = zero4, d = zero4, cw4 = zero4;
    if (ok) {
      h = make_float4((xh[u].x - mean.x) * r.x, (xh[u].y - mean.y) * r.y, (xh[u].z - mean.z) * r.z, (xh[u].w - mean.w) * r.w);
      d = make_float4(din.x + dl.x * wc, din.y + dl.y * wc, din.z + dl.z * wc, din.w + dl.w * wc);
      cw4 = make_float4(dl.x * (h.x * gc + bc), dl.y * (h.y * gc + bc), dl.z * (h.z * gc + bc), dl.w * (h.w * gc + bc));
    }
    xh[u] = h;
    dy[u] = d;
    s1.x += d.x * gc; s1.y += d.y * gc; s1.z += d.z * gc; s1.w += d.w * gc;
    s2.x += d.x * gc * h.x; s2.y += d.y * gc * h.y; s2.z += d.z * gc * h.z; s2.w += d.w * gc * h.w;
    float a = ((d.x * h.x + d.y * h.y) + d.z * h.z) + d.w * h.w, bs = ((d.x + d.y) + d.z) + d.w, cw = ((cw4.x + cw4.y) + cw4.z) + cw4.w;
#pragma unroll
    for (int off = LV_PL / 2; off > 0; off >>= 1) {
      a += __shfl_xor(a, off, 64);
      bs += __shfl_xor(bs, off, 64);
      cw += __shfl_xor(cw, off, 64);
    }
    if (pl == 0) {
      prow[c] = a;
      prow[Cc + c] = bs;
      prow[2 * Cc + c] = cw;
    }
  }
  if (grp == 0) {  // db0: the block's sum of dlogit
    float t = ((dl.x + dl.y) + dl.z) + dl.w;
#pragma unroll
    for (int off = LV_PL / 2; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    if (pl == 0) prow[3 * Cc] = t;
  }
  const float4 t1 = lv_position_sum(s1, part, pl, wave), t2 = lv_position_sum(s2, part, pl, wave);
  if (!ok) return;
  const float inv = 1.f / (float)Cc;
  const float4 m1 = make_float4(t1.x * inv, t1.y * inv, t1.z * inv, t1.w * inv), m2 = make_float4(t2.x * inv, t2.y * inv, t2.z * inv, t2.w * inv);
  float* dxp = dx + (long long)grp * N + n;
#pragma unroll
  for (int u = 0; u < CPT; ++u) {
    const float gc = g[grp + u * LV_GROUPS];
    *reinterpret_cast<float4*>(dxp + (long long)u * LV_GROUPS * N) =
        make_float4(r.x * (dy[u].x * gc - m1.x - xh[u].x * m2.x), r.y * (dy[u].y * gc - m1.y - xh[u].y * m2.y),
                    r.z * (dy[u].z * gc - m1.z - xh[u].z * m2.z), r.w * (dy[u].w * gc - m1.w - xh[u].w * m2.w));
  }
}

// ---- GlanceAttention core on (C, B, T) activations, T = 32, dim_head = 64 (modeling_mgfn.py:107-123) ------------------------
// qkv (3 * inner, B, T) [q rows, then k rows, then v rows; head h = rows h*64 .. h*64+63 of each], one workgroup per (b, h):
//   sim[i][j] = scale * sum_d q[d][i] k[d][j];  p = softmax_j(sim);  out[d][i] = sum_j v[d][j] p[i][j]
// p (B, heads, T, T) is kept for the backward pass.  The whole (b, h) problem (24 KB) lives in LDS; 0.26 MFLOP per workgroup:
// latency, not arithmetic -- what matters is that torch's five launches per block (scale, bmm, softmax, bmm, copy) are one.
constexpr int GA_T = 32, GA_D = 64;
__global__ __launch_bounds__(256) void glance_attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ p_out,
                                                              int inner, int heads, long long N, float scale) {
  __shared__ float q[GA_D][GA_T + 1], k[GA_D][GA_T + 1], v[GA_D][GA_T + 1], p[GA_T][GA_T + 1];
  const int b = blockIdx.x / heads, h = blockIdx.x % heads, tid = threadIdx.x;
  const long long col0 = (long long)b * GA_T;
  for (int e = tid; e < GA_D * GA_T; e += 256) {
    const int d = e / GA_T, t = e % GA_T;
    const long long o = (long long)(h * GA_D + d) * N + col0 + t;
    q[d][t] = qkv[o];
    k[d][t] = qkv[o + (long long)inner * N];
    v[d][t] = qkv[o + 2ll * inner * N];
  }
  __syncthreads();
  // sim: thread -> row i = tid / 8, columns j = (tid % 8) * 4 .. + 3
  const int i = tid >> 3, j0 = (tid & 7) * 4;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int d = 0; d < GA_D; ++d) {
    const float qi = q[d][i];
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += qi * k[d][j0 + e];
  }
  float mx = -3.4e38f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { s[e] *= scale; mx = fmaxf(mx, s[e]); }
  // the 8 threads of a row are consecutive lanes: reduce over them
#pragma unroll
  for (int off = 4; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { s[e] = expf(s[e] - mx); sum += s[e]; }
#pragma unroll
  for (int off = 4; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  const float inv = 1.f / sum;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float pv = s[e] * inv;
    p[i][j0 + e] = pv;
    p_out[((long long)blockIdx.x * GA_T + i) * GA_T + j0 + e] = pv;
  }
  __syncthreads();
  // out[d][i2]: thread -> d = tid / 4, i2 = (tid % 4) * 8 .. + 7
  const int d = tid >> 2, i0 = (tid & 3) * 8;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  for (int j = 0; j < GA_T; ++j) {
    const float vj = v[d][j];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] += vj * p[i0 + e][j];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) out[(long long)(h * GA_D + d) * N + col0 + i0 + e] = o[e];
}

// backward: dqkv (3 * inner, B, T) from dout (inner, B, T), qkv and p
__global__ __launch_bounds__(256) void glance_attn_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ qkv,
                                                              const float* __restrict__ p_in, float* __restrict__ dqkv, int inner, int heads,
                                                              long long N, float scale) {
  __shared__ float q[GA_D][GA_T + 1], k[GA_D][GA_T + 1], v[GA_D][GA_T + 1], g[GA_D][GA_T + 1], p[GA_T][GA_T + 1], ds[GA_T][GA_T + 1];
  const int b = blockIdx.x / heads, h = blockIdx.x % heads, tid = threadIdx.x;
  const long long col0 = (long long)b * GA_T;
  for (int e = tid; e < GA_D * GA_T; e += 256) {
    const int d = e / GA_T, t = e % GA_T;
    const long long o = (long long)(h * GA_D + d) * N + col0 + t;
    q[d][t] = qkv[o];
    k[d][t] = qkv[o + (long long)inner * N];
    v[d][t] = qkv[o + 2ll * inner * N];
    g[d][t] = dout[o];
  }
  for (int e = tid; e < GA_T * GA_T; e += 256) p[e / GA_T][e % GA_T] = p_in[(long long)blockIdx.x * GA_T * GA_T + e];
  __syncthreads();
  // dp[i][j] = sum_d g[d][i] v[d][j];  dsim = p * (dp - sum_j dp p) * scale
  const int i = tid >> 3, j0 = (tid & 7) * 4;
  float dp[4] = {0.f, 0.f, 0.f, 0.f};
  for (int d = 0; d < GA_D; ++d) {
    const float gi = g[d][i];
#pragma unroll
    for (int e = 0; e < 4; ++e) dp[e] += gi * v[d][j0 + e];
  }
  float dot = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) dot += dp[e] * p[i][j0 + e];
#pragma unroll
  for (int off = 4; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
#pragma unroll
  for (int e = 0; e < 4; ++e) ds[i][j0 + e] = p[i][j0 + e] * (dp[e] - dot) * scale;
  __syncthreads();
  // dv[d][j] = sum_i g[d][i] p[i][j];  dq[d][i] = sum_j ds[i][j] k[d][j];  dk[d][j] = sum_i ds[i][j] q[d][i]
  const int d = tid >> 2, t0 = (tid & 3) * 8;
  float dv[8], dq[8], dk[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) dv[e] = dq[e] = dk[e] = 0.f;
  for (int r = 0; r < GA_T; ++r) {
    const float gr = g[d][r], kr = k[d][r], qr = q[d][r];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      dv[e] += gr * p[r][t0 + e];    // i = r, j = t0 + e
      dq[e] += ds[t0 + e][r] * kr;   // i = t0 + e, j = r
      dk[e] += ds[r][t0 + e] * qr;   // i = r, j = t0 + e
    }
  }
  const long long o = (long long)(h * GA_D + d) * N + col0 + t0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    dqkv[o + e] = dq[e];
    dqkv[o + (long long)inner * N + e] = dk[e];
    dqkv[o + 2ll * inner * N + e] = dv[e];
  }
}

// ---- GlanceAttention core for ANY T (dim_head = 64): the validation pass scores a whole video, T = n_clips
// (/root/reference/src/runner.py:42-50 feeds modeling_mgfn.py:107-123 with T = 50..500+).  One workgroup per (32-query tile,
// sequence, head); key / value tiles of 32 clips run through LDS with an online softmax (running row maximum m, running sum l,
// the accumulator rescaled by exp(m_old - m_new) per tile): T x T never exists in memory.  lse[i] = m + log(l) (nullable) is
// what the backward pass needs instead of the T x T softmax.  Same thread -> element maps as the T = 32 kernel above.
constexpr float GA_NEG = -3.0e38f;  // "minus infinity" that stays finite under subtraction

__device__ __forceinline__ void ga_load_tile(float (*dst)[GA_T + 1], const float* __restrict__ src, long long row0, long long N, long long col0,
                                             int t_base, int T, int tid) {
  for (int e = tid; e < GA_D * GA_T; e += 256) {
    const int d = e / GA_T, t = e % GA_T, tt = t_base + t;
    dst[d][t] = tt < T ? src[(row0 + d) * N + col0 + tt] : 0.f;
  }
}

// (LDS images with 16-byte aligned rows, pitch GA_P: the inner loops read k and the transposed softmax tile as float4 -- a third
// of the LDS instructions of the one-dword form, which is what a workgroup's ~10 sequential key tiles at T = 290 spend their time on)
constexpr int GA_P = GA_T + 4;
__device__ __forceinline__ void ga_load_tile_p(float (*dst)[GA_P], const float* __restrict__ src, long long row0, long long N, long long col0,
                                               int t_base, int T, int tid) {
  for (int e = tid; e < GA_D * GA_T; e += 256) {
    const int d = e / GA_T, t = e % GA_T, tt = t_base + t;
    dst[d][t] = tt < T ? src[(row0 + d) * N + col0 + tt] : 0.f;
  }
}

__global__ __launch_bounds__(256) void glance_attn_fwd_anyt_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ lse,
                                                                   int inner, int heads, int T, long long N, float scale, int bh0) {
  __shared__ __attribute__((aligned(16))) float q[GA_D][GA_P], k[GA_D][GA_P], v[GA_D][GA_P], pT[GA_T][GA_P];  // pT[j][i] = p[i][j]
  __shared__ float alpha_s[GA_T], l_s[GA_T];
  const int bh = bh0 + blockIdx.y, b = bh / heads, h = bh % heads, tid = threadIdx.x;
  const int i_base = blockIdx.x * GA_T;
  const long long col0 = (long long)b * T, row0 = (long long)h * GA_D;
  ga_load_tile_p(q, qkv, row0, N, col0, i_base, T, tid);
  const int i = tid >> 3, j0 = (tid & 7) * 4;   // sim: row i, columns j0 .. j0 + 3
  const int d2 = tid >> 2, i0 = (tid & 3) * 8;  // out: channel d2, queries i0 .. i0 + 7
  float m = GA_NEG, l = 0.f, o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  for (int jb = 0; jb < T; jb += GA_T) {
    __syncthreads();  // (the previous tile's readers of k, v, pT, alpha_s are done; first pass: q is complete)
    ga_load_tile_p(k, qkv, inner + row0, N, col0, jb, T, tid);
    ga_load_tile_p(v, qkv, 2ll * inner + row0, N, col0, jb, T, tid);
    __syncthreads();
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < GA_D; ++d) {
      const float qi = q[d][i];
      const float4 kv = *reinterpret_cast<const float4*>(&k[d][j0]);
      s[0] += qi * kv.x; s[1] += qi * kv.y; s[2] += qi * kv.z; s[3] += qi * kv.w;
    }
    float mx = GA_NEG;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s[e] = (jb + j0 + e < T) ? s[e] * scale : GA_NEG;
      mx = fmaxf(mx, s[e]);
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const float m_new = fmaxf(m, mx);  // (finite from the first tile on: key jb is always a real one)
    const float a = expf(m - m_new);
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s[e] = (jb + j0 + e < T) ? expf(s[e] - m_new) : 0.f;
      sum += s[e];
      pT[j0 + e][i] = s[e];
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    l = l * a + sum;
    m = m_new;
    if ((tid & 7) == 0) alpha_s[i] = a;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] *= alpha_s[i0 + e];
    for (int j = 0; j < GA_T; ++j) {
      const float vj = v[d2][j];
      const float4 p0 = *reinterpret_cast<const float4*>(&pT[j][i0]), p1 = *reinterpret_cast<const float4*>(&pT[j][i0 + 4]);
      o[0] += vj * p0.x; o[1] += vj * p0.y; o[2] += vj * p0.z; o[3] += vj * p0.w;
      o[4] += vj * p1.x; o[5] += vj * p1.y; o[6] += vj * p1.z; o[7] += vj * p1.w;
    }
  }
  if ((tid & 7) == 0) {
    l_s[i] = l;
    if (lse != nullptr && i_base + i < T) lse[(long long)bh * T + i_base + i] = m + logf(l);
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (i_base + i0 + e < T) out[(row0 + d2) * N + col0 + i_base + i0 + e] = o[e] / l_s[i0 + e];
}

// ---- the same forward on the matrix pipe (v_mfma_f32_16x16x4_f32, exact fp32 products) for whole-video lengths -----------------
// A validation pass scores a video with T = its clip count (runner.py:42-50): at T in the thousands the T x T products are the
// only part of GlanceAttention that grows quadratically (10 crops x 8192^2 x 64 x 4 FLOP = 172 GFLOP at T = 8192).  One workgroup
// per (64-query tile, sequence, head), four waves of 16 queries each; key / value tiles of 64 clips.  Everything is computed
// TRANSPOSED so that no operand ever changes lanes:
//   S^T[key][query] = sum_d K[d][key] Q[d][query]      A = K^T from LDS (kt[key][d]: one ds_read_b128 = the operands of 4 k-steps),
//                                                       B = this wave's Q columns, 16 registers loaded once
//   O^T[d][query]  += sum_key V[d][key] P^T[key][query] A = V from LDS (vs[d][key], b128 as above), B = P^T = exp(S^T - m) AS IT SITS in
//                                                       the accumulators of the first product (a contraction index may be permuted
//                                                       as long as A and B agree: k-step (jk, e) of lane group lg is key 16 jk + 4 lg + e
//                                                       on both sides)
// A query is a COLUMN (lane & 15): its running maximum / sum need the 16 values of a lane and two xor-shuffles (lanes 16 / 32 apart).
// The next tile's K / V rows are fetched into registers while the current tile is multiplied.  Same recurrence, masks and lse as
// glance_attn_fwd_anyt_kernel; the summation order inside a dot product differs (MFMA chains of 4), so the two agree to rounding.
constexpr int GM_Q = 64, GM_K = 64, GM_KP = GA_D + 4, GM_VP = GM_K + 4;  // LDS pitches: 272-byte rows -> conflict-free b128 fragment reads
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void glance_attn_fwd_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ lse,
                                                                   int inner, int heads, int T, long long N, float scale, int bh0) {
  __shared__ __attribute__((aligned(16))) float kt[GM_K][GM_KP];  // kt[key][d]
  __shared__ __attribute__((aligned(16))) float vs[GA_D][GM_VP];  // vs[d][key]
  const int bh = bh0 + blockIdx.y, b = bh / heads, h = bh % heads, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const long long col0 = (long long)b * T, row0 = (long long)h * GA_D;
  const float* __restrict__ qp = qkv + row0 * N + col0;
  const float* __restrict__ kp = qkv + ((long long)inner + row0) * N + col0;
  const float* __restrict__ vp = qkv + (2ll * inner + row0) * N + col0;
  const int qi = blockIdx.x * GM_Q + wave * 16 + li;  // this lane's query (a column of every fragment)
  const bool q_ok = qi < T;
  // B operand of the first product: qreg[c][e] = Q[d = 16 c + 4 lg + e][qi]
  float qreg[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) qreg[c][e] = q_ok ? qp[(long long)(16 * c + 4 * lg + e) * N + qi] : 0.f;
  // staging map of the K / V tiles: thread = key `lane`, d group `wave`: K rows d = 16 i + 4 wave + e (b128 LDS writes kt[key][d .. d+3]),
  // V rows d = 16 i + 4 wave + e as well (dword LDS writes vs[d][key]); global reads run along the keys: 256 contiguous bytes per wave
  float kst[4][4], vst[4][4];
  const int wrow = __builtin_amdgcn_readfirstlane(4 * wave);
  const float* __restrict__ kpl = kp + (long long)wrow * N + lane;  // this thread's column of the wave's first row; the rest is wave-uniform
  const float* __restrict__ vpl = vp + (long long)wrow * N + lane;
  auto fetch = [&](int jb) __attribute__((always_inline)) {
    if (jb + lane < T) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const long long o = (long long)(16 * i + e) * N + jb;
          kst[i][e] = kpl[o];
          vst[i][e] = vpl[o];
        }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) kst[i][e] = vst[i][e] = 0.f;
    }
  };
  auto stash = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(&kt[lane][16 * i + 4 * wave]) = make_float4(kst[i][0], kst[i][1], kst[i][2], kst[i][3]);
#pragma unroll
      for (int e = 0; e < 4; ++e) vs[16 * i + 4 * wave + e][lane] = vst[i][e];
    }
  };
  f32x4 o[4];  // O^T fragment jd: rows d = 16 jd + 4 lg + r, column qi
#pragma unroll
  for (int jd = 0; jd < 4; ++jd) o[jd] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = GA_NEG, l = 0.f;
  fetch(0);
  for (int jb = 0; jb < T; jb += GM_K) {
    __syncthreads();  // the previous tile's fragment reads are done
    stash();
    __syncthreads();
    if (jb + GM_K < T) fetch(jb + GM_K);  // in flight under this tile's products
    f32x4 s[4];  // S^T fragment jk: rows key = jb + 16 jk + 4 lg + r, column qi
#pragma unroll
    for (int jk = 0; jk < 4; ++jk) s[jk] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (consecutive MFMAs go to DIFFERENT accumulators: four independent chains keep the pipe issuing back to back)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 a[4];
#pragma unroll
      for (int jk = 0; jk < 4; ++jk) a[jk] = *reinterpret_cast<const f32x4*>(&kt[16 * jk + li][16 * c + 4 * lg]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int jk = 0; jk < 4; ++jk) s[jk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jk][e], qreg[c][e], s[jk], 0, 0, 0);
    }
    float mx = GA_NEG;
#pragma unroll
    for (int jk = 0; jk < 4; ++jk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = (jb + 16 * jk + 4 * lg + r < T) ? s[jk][r] * scale : GA_NEG;
        s[jk][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx);  // (finite from the first tile on: key jb is always a real one)
    const float alpha = expf(m - m_new);
    float sum = 0.f;
#pragma unroll
    for (int jk = 0; jk < 4; ++jk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = (jb + 16 * jk + 4 * lg + r < T) ? expf(s[jk][r] - m_new) : 0.f;
        s[jk][r] = pv;
        sum += pv;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    l = l * alpha + sum;
    m = m_new;
#pragma unroll
    for (int jd = 0; jd < 4; ++jd) o[jd] *= alpha;
#pragma unroll
    for (int jk = 0; jk < 4; ++jk) {
      f32x4 a[4];
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) a[jd] = *reinterpret_cast<const f32x4*>(&vs[16 * jd + li][16 * jk + 4 * lg]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int jd = 0; jd < 4; ++jd) o[jd] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jd][e], s[jk][e], o[jd], 0, 0, 0);
    }
  }
  if (!q_ok) return;
  const float inv = 1.f / l;
  float* __restrict__ op = out + row0 * N + col0 + qi;
#pragma unroll
  for (int jd = 0; jd < 4; ++jd)
#pragma unroll
    for (int r = 0; r < 4; ++r) op[(long long)(16 * jd + 4 * lg + r) * N] = o[jd][r] * inv;
  if (lse != nullptr && lg == 0) lse[(long long)bh * T + qi] = m + logf(l);
}

// backward for any T from dout, qkv, out and lse: p[i][j] = exp(scale q_i.k_j - lse[i]) is recomputed tile by tile;
//   D[i] = sum_d dout[d][i] out[d][i];  dp = dout^T v;  ds = p (dp - D) scale;
//   blockIdx.z = 0: dq[d][i] = sum_j ds[i][j] k[d][j] for one query tile (loop over the key tiles);
//   blockIdx.z = 1: dk[d][j] = sum_i ds[i][j] q[d][i], dv[d][j] = sum_i dout[d][i] p[i][j] for one key tile (loop over the query tiles).
// Every output element is written by exactly one thread, sums in tile order: deterministic.
__global__ __launch_bounds__(256) void glance_attn_bwd_anyt_kernel(const float* __restrict__ dout, const float* __restrict__ qkv,
                                                                   const float* __restrict__ out, const float* __restrict__ lse,
                                                                   float* __restrict__ dqkv, int inner, int heads, int T, long long N, float scale, int bh0) {
  __shared__ float q[GA_D][GA_T + 1], k[GA_D][GA_T + 1], v[GA_D][GA_T + 1], g[GA_D][GA_T + 1], go[GA_D][GA_T + 1];
  __shared__ float p[GA_T][GA_T + 1], ds[GA_T][GA_T + 1], D_s[GA_T], L_s[GA_T];
  const int bh = bh0 + blockIdx.y, b = bh / heads, h = bh % heads, tid = threadIdx.x;
  const bool role_q = blockIdx.z == 0;
  const int own = blockIdx.x * GA_T;  // the query tile (role_q) or the key tile this workgroup owns
  const long long col0 = (long long)b * T, row0 = (long long)h * GA_D;
  const int i = tid >> 3, j0 = (tid & 7) * 4;
  const int d2 = tid >> 2, t0 = (tid & 3) * 8;
  float acc0[8], acc1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc0[e] = acc1[e] = 0.f;
  if (!role_q) {
    ga_load_tile(k, qkv, inner + row0, N, col0, own, T, tid);
    ga_load_tile(v, qkv, 2ll * inner + row0, N, col0, own, T, tid);
  }
  for (int ob = 0; ob < T; ob += GA_T) {
    const int ib = role_q ? own : ob, jb = role_q ? ob : own;
    __syncthreads();
    if (role_q) {
      ga_load_tile(k, qkv, inner + row0, N, col0, jb, T, tid);
      ga_load_tile(v, qkv, 2ll * inner + row0, N, col0, jb, T, tid);
    }
    if (!role_q || ob == 0) {  // the query-side tiles: q, dout, dout * out, lse
      ga_load_tile(q, qkv, row0, N, col0, ib, T, tid);
      for (int e = tid; e < GA_D * GA_T; e += 256) {
        const int d = e / GA_T, t = e % GA_T, tt = ib + t;
        const long long o = (row0 + d) * N + col0 + tt;
        const float gv = tt < T ? dout[o] : 0.f;
        g[d][t] = gv;
        go[d][t] = tt < T ? gv * out[o] : 0.f;
      }
      if (tid < GA_T) L_s[tid] = ib + tid < T ? lse[(long long)bh * T + ib + tid] : 0.f;
    }
    __syncthreads();
    if ((!role_q || ob == 0) && tid < GA_T) {
      float dsum = 0.f;
      for (int d = 0; d < GA_D; ++d) dsum += go[d][tid];
      D_s[tid] = dsum;
    }
    float s[4] = {0.f, 0.f, 0.f, 0.f}, dp[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < GA_D; ++d) {
      const float qi = q[d][i], gi = g[d][i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[e] += qi * k[d][j0 + e];
        dp[e] += gi * v[d][j0 + e];
      }
    }
    __syncthreads();  // D_s
    const float Li = L_s[i], Di = D_s[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float pv = (jb + j0 + e < T && ib + i < T) ? expf(s[e] * scale - Li) : 0.f;
      p[i][j0 + e] = pv;
      ds[i][j0 + e] = pv * (dp[e] - Di) * scale;
    }
    __syncthreads();
    if (role_q) {
      for (int r = 0; r < GA_T; ++r) {
        const float kr = k[d2][r];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc0[e] += ds[t0 + e][r] * kr;  // dq: i = t0 + e, j = r
      }
    } else {
      for (int r = 0; r < GA_T; ++r) {
        const float gr = g[d2][r], qr = q[d2][r];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          acc0[e] += ds[r][t0 + e] * qr;  // dk: i = r, j = t0 + e
          acc1[e] += gr * p[r][t0 + e];   // dv
        }
      }
    }
  }
  const long long o = (row0 + d2) * N + col0 + own + t0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (own + t0 + e >= T) break;
    if (role_q) {
      dqkv[o + e] = acc0[e];
    } else {
      dqkv[o + (long long)inner * N + e] = acc0[e];
      dqkv[o + 2ll * inner * N + e] = acc1[e];
    }
  }
}


// ---- a per-input-channel affine map folded into the 1x1 layer that follows it ------------------------------------------------
// W (O, C), the map x -> mul[c] * x[c] + add[c] in front of it:  W (mul . x + add) = (W diag(mul)) x + W add.
//   Wf[o][c] = W[o][c] * mul[c];  bias_f[o] = bias[o] + sum_c W[o][c] * add[c];  rowsum[o] = sum_c Wf[o][c] (nullable)
// Users: eval-mode BatchNorm1d in front of FocusAttention.to_v (modeling_mgfn.py:162, 173-174; mul / add from advhip_bn_fold_f32) and
// the channel LayerNorm in front of MGFNFeedForward.in_conv at inference (mul = g, add = b, rowsum = the mean term's coefficient).
// One wave per output row, lanes along c, xor-butterfly sums: operand-build time only (once per set of weights).
__global__ __launch_bounds__(64) void fold_affine_kernel(const float* __restrict__ W, const float* __restrict__ mul, const float* __restrict__ add,
                                                         const float* __restrict__ bias, float* __restrict__ Wf, float* __restrict__ bias_f,
                                                         float* __restrict__ rowsum, int Cc) {
  const int o = blockIdx.x, lane = threadIdx.x;
  float sa = 0.f, sw = 0.f;
  for (int c = lane; c < Cc; c += 64) {
    const float w = W[(size_t)o * Cc + c];
    const float wf = w * mul[c];
    Wf[(size_t)o * Cc + c] = wf;
    sw += wf;
    if (add != nullptr) sa += w * add[c];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sa += __shfl_xor(sa, off, 64);
    sw += __shfl_xor(sw, off, 64);
  }
  if (lane == 0) {
    bias_f[o] = (bias != nullptr ? bias[o] : 0.f) + sa;
    if (rowsum != nullptr) rowsum[o] = sw;
  }
}

// ---- every GEMM weight of a training step re-packed in ONE launch -------------------------------------------------------
// A differentiated forward needs the packed image of each (just updated) parameter: 42 forward operands and 8 transposed-conv
// operands at the benchmarked size, 5 us launches each.  Block t takes 32 x 32 output tile t of the item whose tile range holds
// it: mode 0 = pack_weight_kernel's LDS transpose, mode 1 = pack_weight_dx_kernel's flipped gather.
__global__ __launch_bounds__(256) void pack_multi_kernel(const advhip_pack_item* __restrict__ items, int n_items, int n_tiles) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    int lo = 0, hi = n_items - 1;  // the last item whose first tile is <= t
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (items[mid].tile_begin <= t) lo = mid; else hi = mid - 1;
    }
    const advhip_pack_item it = items[lo];
    const int lt = t - it.tile_begin;
    if (it.mode == 0) {
      const int K = it.Cin * it.k, Kpad = (K + 31) / 32 * 32, tiles_k = Kpad / 32;
      const int k0 = (lt % tiles_k) * 32, c0 = (lt / tiles_k) * 32;
#pragma unroll
      for (int r = ty; r < 32; r += 8) {
        const int co = c0 + r, k = k0 + tx;
        tile[r][tx] = (co < it.Cout && k < K) ? it.src[(size_t)co * K + k] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, co = c0 + tx;
        if (co < it.Cout) it.dst[(size_t)k * it.Cout + co] = tile[tx][r];
      }
      __syncthreads();
    } else {
      const int rows = (it.Cout * it.k + 31) / 32 * 32, tiles_r = rows / 32;
      const int r0 = (lt % tiles_r) * 32, c0 = (lt / tiles_r) * 32;
#pragma unroll
      for (int r = ty; r < 32; r += 8) {
        const int row = r0 + r, c = c0 + tx;
        const int o = row / it.k, j = row - o * it.k;
        if (c < it.Cin) it.dst[(size_t)row * it.Cin + c] = (o < it.Cout) ? it.src[((size_t)o * it.Cin + c) * it.k + (it.k - 1 - j)] : 0.f;
      }
    }
  }
}

// ---- MGFNFeatureAmplifier's combine step (modeling_mgfn.py:81-93) ------------------------------------------------------------
// tokens = Conv1d_k3(features) + mag_ratio * Conv1d_k3(magnitude).  The 2048 -> 64 conv arrives as its three per-tap products
// z[j] = W_j X (one GEMM on the input as stored); this launch finishes it -- the shifted add over the taps (zero outside [0, T)),
// the bias -- and adds the whole 1 -> 64 magnitude conv, which is three multiply-adds per output.  torch: pad, slices, adds,
// a second unfold + GEMM + bias for the magnitude, scale, add: ~20 launches forward, ~20 backward.
__global__ __launch_bounds__(256) void amp_combine_fwd_kernel(const float* __restrict__ z, const float* __restrict__ bias, const float* __restrict__ mag,
                                                              long long mag_stride, const float* __restrict__ wm, const float* __restrict__ bm,
                                                              float ratio, float* __restrict__ y, int O, long long rows, int T) {
  const long long n = rows * T, total = (long long)O * n;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int o = (int)(i / n);
    const long long p = i - (long long)o * n;
    const int t = (int)(p % T);
    float acc = bias[o] + z[((long long)O + o) * n + p];                  // tap 1: the position itself
    float m = wm[o * 3 + 1] * mag[p * mag_stride];
    if (t > 0) {
      acc += z[(long long)o * n + p - 1];                                  // tap 0 reads t - 1
      m += wm[o * 3] * mag[(p - 1) * mag_stride];
    }
    if (t + 1 < T) {
      acc += z[((long long)2 * O + o) * n + p + 1];                        // tap 2 reads t + 1
      m += wm[o * 3 + 2] * mag[(p + 1) * mag_stride];
    }
    y[i] = acc + ratio * (m + bm[o]);
  }
}

// backward: dz[j][o][p] = dy[o][p - j + 1] inside the row, else 0; one block per output channel o also reduces, in a fixed order,
// d_bias[o] = sum dy, d_bm[o] = ratio * sum dy, d_wm[o][j] = ratio * sum dy[o][p] * mag[p + j - 1]
constexpr int AMPB_THREADS = 1024;  // one block per output channel (64 of them): the block's width is the parallelism there is
__global__ __launch_bounds__(AMPB_THREADS) void amp_combine_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ mag, long long mag_stride,
                                                              float ratio, float* __restrict__ dz, float* __restrict__ d_bias,
                                                              float* __restrict__ d_wm, float* __restrict__ d_bm, int O, long long rows, int T) {
  __shared__ float red[4][AMPB_THREADS];
  const int o = blockIdx.x;
  const long long n = rows * T;
  const float* g = dy + (long long)o * n;
  float s = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (long long p = threadIdx.x; p < n; p += AMPB_THREADS) {
    const int t = (int)(p % T);
    const float v = g[p];
    dz[((long long)O + o) * n + p] = v;
    dz[(long long)o * n + p] = t + 1 < T ? g[p + 1] : 0.f;                 // z[0][p] fed output p + 1
    dz[((long long)2 * O + o) * n + p] = t > 0 ? g[p - 1] : 0.f;           // z[2][p] fed output p - 1
    s += v;
    s1 += v * mag[p * mag_stride];
    if (t > 0) s0 += v * mag[(p - 1) * mag_stride];
    if (t + 1 < T) s2 += v * mag[(p + 1) * mag_stride];
  }
  red[0][threadIdx.x] = s; red[1][threadIdx.x] = s0; red[2][threadIdx.x] = s1; red[3][threadIdx.x] = s2;
  __syncthreads();
  for (int w = AMPB_THREADS / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
#pragma unroll
      for (int q = 0; q < 4; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    d_bias[o] = red[0][0];
    d_bm[o] = ratio * red[0][0];
    d_wm[o * 3] = ratio * red[1][0];
    d_wm[o * 3 + 1] = ratio * red[2][0];
    d_wm[o * 3 + 2] = ratio * red[3][0];
  }
}
}  // namespace advhip

using namespace advhip;

extern "C" int advhip_bn_rows_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* var,
                                      int32_t C, int64_t N, float eps, void* stream) {
  ADVHIP_REQUIRE(x && gamma && beta && y && mean && var && C > 0 && N > 0, "bn_rows_fwd: bad arguments");
  launch_bn_rows_fwd(x, gamma, beta, y, mean, var, C, (long long)N, eps, nullptr, nullptr, 0.f, (hipStream_t)stream);
  return check_launch("bn_rows_fwd");
}

extern "C" int advhip_bn_rows_fwd_running_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* var,
                                              float* running_mean, float* running_var, float momentum, int32_t C, int64_t N, float eps,
                                              void* stream) {
  ADVHIP_REQUIRE(x && gamma && beta && y && mean && var && C > 0 && N > 0, "bn_rows_fwd_running: bad arguments");
  ADVHIP_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_rows_fwd_running: running_mean and running_var come together");
  launch_bn_rows_fwd(x, gamma, beta, y, mean, var, C, (long long)N, eps, running_mean, running_var, momentum, (hipStream_t)stream);
  return check_launch("bn_rows_fwd_running");
}

extern "C" int advhip_bn_rows_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* var,
                                      float* dx, float* dgamma, float* dbeta, int32_t C, int64_t N, float eps, void* stream) {
  ADVHIP_REQUIRE(dy && x && gamma && mean && var && dx && dgamma && dbeta && C > 0 && N > 0, "bn_rows_bwd: bad arguments");
  launch_bn_rows_bwd(dy, x, gamma, mean, var, dx, dgamma, dbeta, C, (long long)N, eps, nullptr, (hipStream_t)stream);
  return check_launch("bn_rows_bwd");
}

extern "C" int advhip_bn_rows_bwd_add_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* var,
                                          const float* add, float* dx, float* dgamma, float* dbeta, int32_t C, int64_t N, float eps,
                                          void* stream) {
  ADVHIP_REQUIRE(dy && x && gamma && mean && var && dx && dgamma && dbeta && C > 0 && N > 0, "bn_rows_bwd_add: bad arguments");
  launch_bn_rows_bwd(dy, x, gamma, mean, var, dx, dgamma, dbeta, C, (long long)N, eps, add, (hipStream_t)stream);
  return check_launch("bn_rows_bwd_add");
}

extern "C" int advhip_chan_stats_f32(const float* x, float* mu, float* rs, int32_t C, int64_t N, float eps, void* stream) {
  ADVHIP_REQUIRE(x && mu && rs && C > 0 && N > 0, "chan_stats: bad arguments");
  const long long blocks = (N + LN_COLS - 1) / LN_COLS;
  ADVHIP_REQUIRE(blocks < (1ll << 31), "chan_stats: too many positions");
  hipLaunchKernelGGL(chan_stats_kernel, dim3((unsigned)blocks), dim3(LN_THREADS), 0, (hipStream_t)stream, x, mu, rs, C, (long long)N, eps);
  return check_launch("chan_stats");
}

extern "C" int advhip_chan_layernorm_fwd_f32(const float* x, const float* g, const float* b, float* y, float* mu, float* rs,
                                             int32_t C, int64_t N, float eps, void* stream) {
  ADVHIP_REQUIRE(x && g && b && y && mu && rs && C > 0 && N > 0, "chan_layernorm_fwd: bad arguments");
  const long long blocks = (N + LN_COLS - 1) / LN_COLS;
  ADVHIP_REQUIRE(blocks < (1ll << 31), "chan_layernorm_fwd: too many positions");
  if (!launch_ln_fwd_v4(x, g, b, y, mu, rs, C, (long long)N, eps, (hipStream_t)stream))
    hipLaunchKernelGGL(chan_layernorm_fwd_kernel, dim3((unsigned)blocks), dim3(LN_THREADS), 0, (hipStream_t)stream, x, g, b, y, mu, rs, C,
                       (long long)N, eps);
  return check_launch("chan_layernorm_fwd");
}

extern "C" int64_t advhip_chan_layernorm_bwd_partial_rows(int64_t N) { return (N + LN_COLS - 1) / LN_COLS; }

extern "C" int advhip_chan_layernorm_bwd_f32(const float* dy, const float* x, const float* g, const float* mu, const float* rs,
                                             float* dx, float* dg_partial, float* db_partial, int32_t C, int64_t N, float eps,
                                             void* stream) {
  ADVHIP_REQUIRE(dy && x && g && mu && rs && dx && dg_partial && db_partial && C > 0 && N > 0, "chan_layernorm_bwd: bad arguments");
  const long long blocks = (N + LN_COLS - 1) / LN_COLS;
  ADVHIP_REQUIRE(blocks < (1ll << 31), "chan_layernorm_bwd: too many positions");
  if (!launch_ln_bwd_v4(dy, x, g, mu, rs, dx, dg_partial, db_partial, C, (long long)N, eps, nullptr, (long long)C, (hipStream_t)stream))
    hipLaunchKernelGGL(chan_layernorm_bwd_kernel, dim3((unsigned)blocks), dim3(LN_THREADS), 0, (hipStream_t)stream, dy, x, g, mu, rs, dx,
                       dg_partial, db_partial, C, (long long)N, eps, (const float*)nullptr, (long long)C);
  return check_launch("chan_layernorm_bwd");
}

extern "C" int advhip_chan_layernorm_bwd_add_f32(const float* dy, const float* x, const float* g, const float* mu, const float* rs,
                                                 const float* add, float* dx, float* dgb_partial, int32_t C, int64_t N, float eps,
                                                 void* stream) {
  ADVHIP_REQUIRE(dy && x && g && mu && rs && dx && dgb_partial && C > 0 && N > 0, "chan_layernorm_bwd_add: bad arguments");
  const long long blocks = (N + LN_COLS - 1) / LN_COLS;
  ADVHIP_REQUIRE(blocks < (1ll << 31), "chan_layernorm_bwd_add: too many positions");
  if (!launch_ln_bwd_v4(dy, x, g, mu, rs, dx, dgb_partial, dgb_partial + C, C, (long long)N, eps, add, 2ll * C, (hipStream_t)stream))
    hipLaunchKernelGGL(chan_layernorm_bwd_kernel, dim3((unsigned)blocks), dim3(LN_THREADS), 0, (hipStream_t)stream, dy, x, g, mu, rs, dx,
                       dgb_partial, dgb_partial + C, C, (long long)N, eps, add, 2ll * C);
  return check_launch("chan_layernorm_bwd_add");
}

extern "C" int advhip_unfold3_f32(const float* x, float* u, int32_t C, int64_t rows, int32_t T, void* stream) {
  ADVHIP_REQUIRE(x && u && C > 0 && rows > 0 && T > 0, "unfold3: bad arguments");
  ADVHIP_REQUIRE(T % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)u & 15) == 0, "unfold3: T=%d must be a multiple of 4 and the tensors 16-byte aligned", T);
  const long long per_c = (long long)rows * T, total4 = (long long)C * per_c / 4;
  const int grid = (int)std::min<long long>((total4 + 255) / 256, 256 * 64);
  hipLaunchKernelGGL(unfold3_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, u, T, per_c, total4);
  return check_launch("unfold3");
}

extern "C" int advhip_dwconv_t_fwd_f32(const float* v, const float* w, const float* bias, float* out, int32_t C, int32_t H,
                                       int64_t rows, int32_t T, int32_t K, void* stream) {
  ADVHIP_REQUIRE(v && w && bias && out && C > 0 && H > 0 && rows > 0 && T > 0 && C % H == 0, "dwconv_t_fwd: bad arguments");
  ADVHIP_REQUIRE(K == 5 || K == 3, "dwconv_t: kernel size %d (3 and 5 are instantiated)", K);
  const long long total = (long long)C * rows * T;
  const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 32);
  if (T % 4 == 0 && (((uintptr_t)v | (uintptr_t)out) & 15) == 0) {
    const int grid4 = (int)std::min<long long>((total / 4 + 255) / 256, 256 * 32);
    if (K == 5) hipLaunchKernelGGL(dwconv_t_fwd_v4_kernel<5>, dim3(grid4), dim3(256), 0, (hipStream_t)stream, v, w, bias, out, H, T, (long long)rows, total / 4);
    else hipLaunchKernelGGL(dwconv_t_fwd_v4_kernel<3>, dim3(grid4), dim3(256), 0, (hipStream_t)stream, v, w, bias, out, H, T, (long long)rows, total / 4);
    return check_launch("dwconv_t_fwd");
  }
  if (K == 5) hipLaunchKernelGGL(dwconv_t_fwd_kernel<5>, dim3(grid), dim3(256), 0, (hipStream_t)stream, v, w, bias, out, H, T, (long long)rows, total);
  else hipLaunchKernelGGL(dwconv_t_fwd_kernel<3>, dim3(grid), dim3(256), 0, (hipStream_t)stream, v, w, bias, out, H, T, (long long)rows, total);
  return check_launch("dwconv_t_fwd");
}

extern "C" int32_t advhip_dwconv_t_bwd_chunks(int32_t C, int64_t rows) {
  // blocks per channel: enough blocks for the chip, whole rows per block
  int chunks = 1;
  while ((long long)C * chunks < 2048 && rows % (chunks * 2) == 0) chunks *= 2;
  return chunks;
}

extern "C" int advhip_dwconv_t_bwd_f32(const float* dout, const float* v, const float* w, float* dv, float* partial, int32_t C,
                                       int32_t H, int64_t rows, int32_t T, int32_t K, void* stream) {
  ADVHIP_REQUIRE(dout && v && w && dv && partial && C > 0 && H > 0 && rows > 0 && T > 0 && C % H == 0, "dwconv_t_bwd: bad arguments");
  ADVHIP_REQUIRE(K == 5 || K == 3, "dwconv_t: kernel size %d (3 and 5 are instantiated)", K);
  const int chunks = advhip_dwconv_t_bwd_chunks(C, rows);
  const long long per_c = (long long)rows * T;
  const dim3 grid((unsigned)((long long)C * chunks));
  if (T % 4 == 0 && (((uintptr_t)v | (uintptr_t)dout | (uintptr_t)dv) & 15) == 0) {
    if (K == 5) hipLaunchKernelGGL(dwconv_t_bwd_v4_kernel<5>, grid, dim3(256), 0, (hipStream_t)stream, dout, v, w, dv, partial, H, T, per_c, chunks);
    else hipLaunchKernelGGL(dwconv_t_bwd_v4_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, dout, v, w, dv, partial, H, T, per_c, chunks);
    return check_launch("dwconv_t_bwd");
  }
  if (K == 5) hipLaunchKernelGGL(dwconv_t_bwd_kernel<5>, grid, dim3(256), 0, (hipStream_t)stream, dout, v, w, dv, partial, H, T, per_c, chunks);
  else hipLaunchKernelGGL(dwconv_t_bwd_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, dout, v, w, dv, partial, H, T, per_c, chunks);
  return check_launch("dwconv_t_bwd");
}

extern "C" int advhip_glance_attention_fwd_f32(const float* qkv, float* out, float* p, int32_t heads, int64_t B, int32_t T, int32_t dim_head,
                                               float scale, void* stream) {
  ADVHIP_REQUIRE(qkv && out && p && heads > 0 && B > 0, "glance_attention_fwd: bad arguments");
  ADVHIP_REQUIRE(T == GA_T && dim_head == GA_D, "glance_attention: T = %d / dim_head = %d (the kernel is built for %d / %d)", T, dim_head, GA_T, GA_D);
  ADVHIP_REQUIRE(B * heads < (1ll << 31), "glance_attention: too many (sequence, head) pairs");
  hipLaunchKernelGGL(glance_attn_fwd_kernel, dim3((unsigned)(B * heads)), dim3(256), 0, (hipStream_t)stream, qkv, out, p, heads * GA_D, heads,
                     (long long)B * T, scale);
  return check_launch("glance_attention_fwd");
}

extern "C" int advhip_glance_attention_bwd_f32(const float* dout, const float* qkv, const float* p, float* dqkv, int32_t heads, int64_t B,
                                               int32_t T, int32_t dim_head, float scale, void* stream) {
  ADVHIP_REQUIRE(dout && qkv && p && dqkv && heads > 0 && B > 0, "glance_attention_bwd: bad arguments");
  ADVHIP_REQUIRE(T == GA_T && dim_head == GA_D, "glance_attention: T = %d / dim_head = %d (the kernel is built for %d / %d)", T, dim_head, GA_T, GA_D);
  ADVHIP_REQUIRE(B * heads < (1ll << 31), "glance_attention: too many (sequence, head) pairs");
  hipLaunchKernelGGL(glance_attn_bwd_kernel, dim3((unsigned)(B * heads)), dim3(256), 0, (hipStream_t)stream, dout, qkv, p, dqkv, heads * GA_D, heads,
                     (long long)B * T, scale);
  return check_launch("glance_attention_bwd");
}

constexpr int GLANCE_MFMA_MIN_T = 256;
extern "C" int advhip_glance_attention_fwd_anyt_f32(const float* qkv, float* out, float* lse, int32_t heads, int64_t B, int32_t T, int32_t dim_head,
                                                    float scale, void* stream) {
  ADVHIP_REQUIRE(qkv && out && heads > 0 && B > 0 && T > 0, "glance_attention_fwd_anyt: bad arguments");
  ADVHIP_REQUIRE(dim_head == GA_D, "glance_attention_anyt: dim_head = %d (the kernel is built for %d)", dim_head, GA_D);
  ADVHIP_REQUIRE(B * heads < (1ll << 31), "glance_attention_anyt: too many (sequence, head) pairs");
  // the matrix-pipe form from GLANCE_MFMA_MIN_T clips on (below it a pass is a few dozen workgroups of a handful of key tiles: the
  // 32-query tiles of the vector kernel give the chip more of them).  A compile-time constant: the library reads no environment
  // (the A/B of profiles/r06_studies.md section 4 was a study build)
  const bool mfma = T >= GLANCE_MFMA_MIN_T;
  const unsigned tiles = mfma ? (unsigned)((T + GM_Q - 1) / GM_Q) : (unsigned)((T + GA_T - 1) / GA_T);
  for (long long bh0 = 0; bh0 < B * heads; bh0 += 32768) {  // (grid.y is 16 bits wide)
    const unsigned ny = (unsigned)std::min<long long>(32768, B * heads - bh0);
    if (mfma)
      hipLaunchKernelGGL(glance_attn_fwd_mfma_kernel, dim3(tiles, ny), dim3(256), 0, (hipStream_t)stream, qkv, out, lse, heads * GA_D, heads, T,
                         (long long)B * T, scale, (int)bh0);
    else
      hipLaunchKernelGGL(glance_attn_fwd_anyt_kernel, dim3(tiles, ny), dim3(256), 0, (hipStream_t)stream, qkv, out, lse, heads * GA_D, heads, T,
                         (long long)B * T, scale, (int)bh0);
  }
  return check_launch("glance_attention_fwd_anyt");
}

extern "C" int advhip_glance_attention_bwd_anyt_f32(const float* dout, const float* qkv, const float* out, const float* lse, float* dqkv, int32_t heads,
                                                    int64_t B, int32_t T, int32_t dim_head, float scale, void* stream) {
  ADVHIP_REQUIRE(dout && qkv && out && lse && dqkv && heads > 0 && B > 0 && T > 0, "glance_attention_bwd_anyt: bad arguments");
  ADVHIP_REQUIRE(dim_head == GA_D, "glance_attention_anyt: dim_head = %d (the kernel is built for %d)", dim_head, GA_D);
  ADVHIP_REQUIRE(B * heads < (1ll << 31), "glance_attention_anyt: too many (sequence, head) pairs");
  const unsigned tiles = (unsigned)((T + GA_T - 1) / GA_T);
  for (long long bh0 = 0; bh0 < B * heads; bh0 += 32768) {
    const unsigned ny = (unsigned)std::min<long long>(32768, B * heads - bh0);
    hipLaunchKernelGGL(glance_attn_bwd_anyt_kernel, dim3(tiles, ny, 2), dim3(256), 0, (hipStream_t)stream, dout, qkv, out, lse, dqkv, heads * GA_D, heads,
                       T, (long long)B * T, scale, (int)bh0);
  }
  return check_launch("glance_attention_bwd_anyt");
}

extern "C" int advhip_fold_affine_f32(const float* W, const float* mul, const float* add, const float* bias, float* Wf, float* bias_f, float* rowsum,
                                      int32_t O, int32_t C, void* stream) {
  ADVHIP_REQUIRE(W && mul && Wf && bias_f && O > 0 && C > 0, "fold_affine: bad arguments");
  hipLaunchKernelGGL(fold_affine_kernel, dim3((unsigned)O), dim3(64), 0, (hipStream_t)stream, W, mul, add, bias, Wf, bias_f, rowsum, C);
  return check_launch("fold_affine");
}

extern "C" int64_t advhip_head_ln_fc_partial_rows(int64_t N) { return (N + LN_COLS - 1) / LN_COLS; }

extern "C" int advhip_head_ln_fc_fwd_f32(const float* y, const float* ln_g, const float* ln_b, const float* fc_w, const float* fc_b, float* xn,
                                         float* mean, float* rstd, float* score, int32_t C, int64_t N, float eps, void* stream) {
  ADVHIP_REQUIRE(y && ln_g && ln_b && fc_w && fc_b && xn && mean && rstd && score && C > 0 && N > 0, "head_ln_fc_fwd: bad arguments");
  const long long blocks = (N + LN_COLS - 1) / LN_COLS;
  ADVHIP_REQUIRE(blocks < (1ll << 31), "head_ln_fc_fwd: too many positions");
  const bool v4 = C == 16 * LV_GROUPS && N % 4 == 0 &&
                  (((uintptr_t)y | (uintptr_t)xn | (uintptr_t)mean | (uintptr_t)rstd | (uintptr_t)score) & 15) == 0;
  if (v4)
    hipLaunchKernelGGL(head_ln_fc_fwd_v4_kernel<16>, dim3((unsigned)blocks), dim3(LV_THREADS), 0, (hipStream_t)stream, y, ln_g, ln_b, fc_w, fc_b, xn, mean,
                       rstd, score, (long long)N, eps);
  else
    hipLaunchKernelGGL(head_ln_fc_fwd_kernel, dim3((unsigned)blocks), dim3(LN_THREADS), 0, (hipStream_t)stream, y, ln_g, ln_b, fc_w, fc_b, xn, mean, rstd,
                       score, C, (long long)N, eps);
  return check_launch("head_ln_fc_fwd");
}

extern "C" int advhip_head_ln_fc_bwd_f32(const float* d_xn, const float* d_score, const float* y, const float* ln_g, const float* ln_b,
                                         const float* fc_w, const float* mean, const float* rstd, const float* score, float* dy, float* partial,
                                         int32_t C, int64_t N, void* stream) {
  ADVHIP_REQUIRE(y && ln_g && ln_b && fc_w && mean && rstd && score && dy && partial && C > 0 && N > 0, "head_ln_fc_bwd: bad arguments");
  const long long blocks = (N + LN_COLS - 1) / LN_COLS;
  ADVHIP_REQUIRE(blocks < (1ll << 31), "head_ln_fc_bwd: too many positions");
  const bool v4 = C == 16 * LV_GROUPS && N % 4 == 0 &&
                  (((uintptr_t)y | (uintptr_t)d_xn | (uintptr_t)d_score | (uintptr_t)mean | (uintptr_t)rstd | (uintptr_t)score | (uintptr_t)dy) & 15) == 0;
  if (v4)
    hipLaunchKernelGGL(head_ln_fc_bwd_v4_kernel<16>, dim3((unsigned)blocks), dim3(LV_THREADS), 0, (hipStream_t)stream, d_xn, d_score, y, ln_g, ln_b, fc_w,
                       mean, rstd, score, dy, partial, (long long)N);
  else
    hipLaunchKernelGGL(head_ln_fc_bwd_kernel, dim3((unsigned)blocks), dim3(LN_THREADS), 0, (hipStream_t)stream, d_xn, d_score, y, ln_g, ln_b, fc_w, mean,
                       rstd, score, dy, partial, C, (long long)N);
  return check_launch("head_ln_fc_bwd");
}

extern "C" int advhip_colsum_f32(const float* src, float* dst, int64_t rows, int32_t cols, void* stream) {
  ADVHIP_REQUIRE(src && dst && rows > 0 && cols > 0, "colsum: bad arguments");
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((cols + 31) / 32)), dim3(256), 0, (hipStream_t)stream, src, dst, (long long)rows, cols);
  return check_launch("colsum");
}

extern "C" int64_t advhip_pack_item_tiles(int32_t Cout, int32_t Cin, int32_t k, int32_t mode) {
  if (Cout <= 0 || Cin <= 0 || k <= 0 || k > 10 || (mode != 0 && mode != 1)) return -1;
  const int64_t K = (int64_t)Cin * k, rows = (int64_t)Cout * k;
  return mode == 0 ? ((K + 31) / 32) * ((Cout + 31) / 32) : ((rows + 31) / 32) * ((Cin + 31) / 32);
}

extern "C" int advhip_pack_weights_multi_f32(const advhip_pack_item* items_dev, int32_t n_items, int32_t n_tiles, void* stream) {
  ADVHIP_REQUIRE(items_dev && n_items > 0 && n_tiles > 0, "pack_weights_multi: bad arguments");
  hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)std::min(n_tiles, 256 * 64)), dim3(256), 0, (hipStream_t)stream, items_dev, n_items, n_tiles);
  return check_launch("pack_weights_multi");
}

extern "C" int advhip_colsum_group_f32(const advhip_colsum_item* items, int32_t n_items, void* stream) {
  ADVHIP_REQUIRE(items && n_items > 0, "colsum_group: bad arguments");
  for (int base = 0; base < n_items; base += COLSUM_GROUP_MAX) {
    ColsumGroupArgs ga;
    ga.n = std::min(COLSUM_GROUP_MAX, n_items - base);
    long long blocks = 0;
    for (int i = 0; i < ga.n; ++i) {
      const advhip_colsum_item& s = items[base + i];
      ADVHIP_REQUIRE(s.src && s.dst && s.rows > 0 && s.cols > 0 && s.period >= 0 && (s.period == 0 || (s.period >= 2 && s.cols % s.period == 0)),
                     "colsum_group: item %d: bad arguments", base + i);
      ga.it[i] = ColsumGroupItem{s.src, s.dst, (long long)s.rows, s.cols, (int)blocks, s.period, 0};
      blocks += (s.cols + 31) / 32;
      ADVHIP_REQUIRE(blocks < (1ll << 31), "colsum_group: too many columns");
    }
    for (int i = ga.n; i < COLSUM_GROUP_MAX; ++i) ga.it[i] = ColsumGroupItem{nullptr, nullptr, 0, 0, 0x7FFFFFFF, 0, 0};
    hipLaunchKernelGGL(colsum_group_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ga);
    if (int rc = check_launch("colsum_group")) return rc;
  }
  return ADVHIP_OK;
}

// ---- Adam with L2-in-gradient weight decay over many parameter tensors in one launch (torch.optim.Adam's update rule,
// /root/reference/src/runner.py:53-59; torch/optim/adam.py) -------------------------------------------------------------------
//   g' = g + wd * p;  m = m + (1 - b1) (g' - m);  v = b2 v + (1 - b2) g'^2;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// t = *step (a device-side counter the caller has already incremented for this step; fp32 as torch's capturable Adam keeps it).
// The item table travels in the kernel arguments (a launch inside a captured HIP graph replays it as is); a workgroup finds its
// item by a scalar scan and updates ADAM_PER_BLOCK consecutive elements of it, 16 bytes per lane and access.
constexpr int ADAM_MAX_ITEMS = 80, ADAM_PER_BLOCK = 4096;
struct AdamItem {
  float* p;
  const float* g;
  float* m;
  float* v;
  const float* step;
  int n;
  int block_begin;
};
struct AdamArgs {
  AdamItem it[ADAM_MAX_ITEMS];
  int n;
  double lr, b1, b2;  // (doubles: 1 - beta and lr / (1 - beta^t) are formed in double and rounded once, as torch does in Python)
  float eps, wd;
};
__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamArgs aa) {
  int i = 0;
  while (i + 1 < aa.n && (int)blockIdx.x >= aa.it[i + 1].block_begin) ++i;
  const AdamItem& t = aa.it[i];
  const int base = ((int)blockIdx.x - t.block_begin) * ADAM_PER_BLOCK;
  const double st = (double)*t.step;
  const float step_size = (float)(aa.lr / (1.0 - pow(aa.b1, st)));
  const float bc2_sqrt = (float)sqrt(1.0 - pow(aa.b2, st));
  const float omb1 = (float)(1.0 - aa.b1), b2 = (float)aa.b2, omb2 = (float)(1.0 - aa.b2), eps = aa.eps, wd = aa.wd;
  auto upd = [&](float& p, float g, float& m, float& v) {
    g = g + wd * p;
    m = m + omb1 * (g - m);
    v = b2 * v + omb2 * g * g;
    p = p - step_size * m / (sqrtf(v) / bc2_sqrt + eps);
  };
  const bool vec = ((((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v) & 15) == 0);
#pragma unroll
  for (int u = 0; u < ADAM_PER_BLOCK / 1024; ++u) {
    const int e = base + u * 1024 + (int)threadIdx.x * 4;
    if (e >= t.n) break;
    if (vec && e + 4 <= t.n) {
      float4 p = *reinterpret_cast<const float4*>(t.p + e), m = *reinterpret_cast<const float4*>(t.m + e), v = *reinterpret_cast<const float4*>(t.v + e);
      const float4 g = *reinterpret_cast<const float4*>(t.g + e);
      upd(p.x, g.x, m.x, v.x); upd(p.y, g.y, m.y, v.y); upd(p.z, g.z, m.z, v.z); upd(p.w, g.w, m.w, v.w);
      *reinterpret_cast<float4*>(t.p + e) = p;
      *reinterpret_cast<float4*>(t.m + e) = m;
      *reinterpret_cast<float4*>(t.v + e) = v;
    } else {
      for (int k = e; k < e + 4 && k < t.n; ++k) {
        float p = t.p[k], m = t.m[k], v = t.v[k];
        upd(p, t.g[k], m, v);
        t.p[k] = p; t.m[k] = m; t.v[k] = v;
      }
    }
  }
}

extern "C" int advhip_adam_multi_f32(const advhip_adam_item* items, int32_t n_items, double lr, double beta1, double beta2, double eps,
                                     double weight_decay, void* stream) {
  ADVHIP_REQUIRE(items && n_items > 0, "adam_multi: bad arguments");
  ADVHIP_REQUIRE(lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0. && weight_decay >= 0.,
                 "adam_multi: bad hyper-parameters (lr %g, betas %g %g, eps %g, weight_decay %g)", lr, beta1, beta2, eps, weight_decay);
  for (int base = 0; base < n_items; base += ADAM_MAX_ITEMS) {
    AdamArgs aa;
    aa.n = std::min(ADAM_MAX_ITEMS, n_items - base);
    aa.lr = lr; aa.b1 = beta1; aa.b2 = beta2; aa.eps = (float)eps; aa.wd = (float)weight_decay;
    long long blocks = 0;
    for (int i = 0; i < aa.n; ++i) {
      const advhip_adam_item& s = items[base + i];
      ADVHIP_REQUIRE(s.param && s.grad && s.exp_avg && s.exp_avg_sq && s.step && s.n > 0 && s.n < (1ll << 31), "adam_multi: item %d: bad arguments", base + i);
      aa.it[i] = AdamItem{s.param, s.grad, s.exp_avg, s.exp_avg_sq, s.step, (int)s.n, (int)blocks};
      blocks += (s.n + ADAM_PER_BLOCK - 1) / ADAM_PER_BLOCK;
      ADVHIP_REQUIRE(blocks < (1ll << 31), "adam_multi: too many elements");
    }
    for (int i = aa.n; i < ADAM_MAX_ITEMS; ++i) aa.it[i] = AdamItem{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0x7FFFFFFF};
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, aa);
    if (int rc = check_launch("adam_multi")) return rc;
  }
  return ADVHIP_OK;
}

extern "C" int advhip_amp_combine_fwd_f32(const float* z, const float* bias, const float* mag, int64_t mag_stride, const float* wm, const float* bm,
                                          float ratio, float* y, int32_t O, int64_t rows, int32_t T, void* stream) {
  ADVHIP_REQUIRE(z && bias && mag && wm && bm && y && O > 0 && rows > 0 && T > 0 && mag_stride > 0, "amp_combine_fwd: bad arguments");
  const long long total = (long long)O * rows * T;
  hipLaunchKernelGGL(amp_combine_fwd_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 256 * 32)), dim3(256), 0, (hipStream_t)stream, z, bias, mag,
                     (long long)mag_stride, wm, bm, ratio, y, O, (long long)rows, T);
  return check_launch("amp_combine_fwd");
}

extern "C" int advhip_amp_combine_bwd_f32(const float* dy, const float* mag, int64_t mag_stride, float ratio, float* dz, float* d_bias, float* d_wm,
                                          float* d_bm, int32_t O, int64_t rows, int32_t T, void* stream) {
  ADVHIP_REQUIRE(dy && mag && dz && d_bias && d_wm && d_bm && O > 0 && rows > 0 && T > 0 && mag_stride > 0, "amp_combine_bwd: bad arguments");
  hipLaunchKernelGGL(amp_combine_bwd_kernel, dim3((unsigned)O), dim3(AMPB_THREADS), 0, (hipStream_t)stream, dy, mag, (long long)mag_stride, ratio, dz, d_bias, d_wm,
                     d_bm, O, (long long)rows, T);
  return check_launch("amp_combine_bwd");
}
