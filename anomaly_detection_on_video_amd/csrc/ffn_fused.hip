// One launch for a whole `x = x + FFN(LN(x))` step of a NARROW MGFN block (64 or 128 channels, hidden = 4 x), forward and backward.
// Restates MGFNFeedForward + the block's residual add, /root/reference/src/models/mgfn/modeling_mgfn.py:36-64, 147, 205:
//   xh = (x - mean_c x) / (sqrt(var_c x) + eps) * g + b          MGFNLayerNorm over the channels of every position (:36-46)
//   h  = GELU(W1 xh + b1),  y = W2 h + b2 + x                    Conv1d(C, 4C, 1) -> GELU -> Conv1d(4C, C, 1) (:53-64), + x (:147, :205)
// As three launches (LayerNorm; GEMM + GELU; GEMM + bias + residual) such a step is 37 us (C = 64) / 52 us (C = 128) of a training step
// for 0.67 / 2.7 GFLOP: every launch sits at its latency floor (profiles/r06_studies.md section 7).  Here one wave owns 16 positions
// for the whole chain and nothing but the saved tensors goes through memory:
//   * activations are (C, N) with the N positions contiguous; lane (li = lane & 15, lg = lane >> 4) holds position p0 + li and the
//     channels c = 16 cc + 4 lg + e (cc < C / 16, e < 4) of it: the four lg-lanes of a position hold all its channels, so the
//     LayerNorm statistics are an in-lane sum and two xor-shuffles (lanes 16 / 32 apart);
//   * both GEMMs run TRANSPOSED on v_mfma_f32_16x16x4_f32 (D[row][col]: row = output channel, col = position), so that a lane's
//     values are always "some channels of MY position": the B operand of the first product is the lane's xh registers as they
//     are (k-step (cc, e) of lane group lg contracts channel 16 cc + 4 lg + e on both sides -- a contraction index may be
//     permuted), its result H^T sits in the accumulators exactly as the second product's B operand wants it (k-step (f, r)
//     contracts hidden channel 16 f + 4 lg + r), and the second product's result is again "channels c = 16 fc + 4 lg + r of my
//     position" -- the distribution x arrived in, which is what the residual add and the LayerNorm backward need;
//   * the A operands (weights, rows = output channel, k contiguous) are streamed through LDS in chunks of CH hidden channels, double
//     buffered, read as ds_read_b128 = the operands of four k-steps; row pitches of K + 4 floats keep those reads conflict-free.
// The backward kernel is the same machine on the transposed operands: dZ = (W2^T dY) * GELU'(z) [GELU' saved by the forward],
// dXh = W1^T dZ, then the LayerNorm backward + the skip connection's gradient in registers.
//
// OPT-IN (ADV_MGFN_FUSED_FFN=1; mgfn_ops.FUSED_FFN): correct (tests/test_hip_mgfn.py::test_fused_narrow_ffn_block_..., the training-step
// tests pass with it on) but not faster.  Inside the graph-replayed training step: C = 64: 26.5 us forward + 29.0 backward against 37 + 29 for
// the launches it replaces; C = 128: 74 + 99 us against 52 + 46.  640 waves on 1 024 SIMDs run their 512 / 2 048 MFMAs as one serial chain
// each with a barrier and an LDS hand-over per chunk, about 3x the chain's own issue time; the three-launch form spreads the same work over
// 640 ... 1 280 workgroups of the tuned GEMM kernel.  Kept as the tested end point of that study (profiles/r06_studies.md section 7).
#include "common.h"

namespace advhip {

constexpr int FF_POS = 64;  // positions per workgroup: 4 waves x 16

// the next chunk of both weight matrices into registers (A1: rows [j*CH, (j+1)*CH) of a [HID][C] matrix, contiguous;
// A2: columns [j*CH, (j+1)*CH) of a [C][HID] matrix), and from the registers into an LDS buffer
template <int C, int CH>
struct FfChunk {
  static constexpr int P1 = C + 4, P2 = CH + 4;            // LDS row pitches (floats)
  static constexpr int N4 = C * CH / 4 / 256;               // float4 per thread and matrix
  static_assert(C * CH / 4 % 256 == 0, "chunk size");
  float4 r1[N4], r2[N4];
  __device__ __forceinline__ void fetch(const float* __restrict__ A1, const float* __restrict__ A2, int HID, int j, int tid) {
#pragma unroll
    for (int i = 0; i < N4; ++i) {
      const int q = tid + 256 * i;
      r1[i] = *reinterpret_cast<const float4*>(A1 + (size_t)j * CH * C + (size_t)q * 4);
      const int row = q / (CH / 4), c4 = q % (CH / 4);
      r2[i] = *reinterpret_cast<const float4*>(A2 + (size_t)row * HID + (size_t)j * CH + c4 * 4);
    }
  }
  __device__ __forceinline__ void stash(float* __restrict__ s1, float* __restrict__ s2, int tid) const {
#pragma unroll
    for (int i = 0; i < N4; ++i) {
      const int q = tid + 256 * i;
      *reinterpret_cast<float4*>(s1 + (q / (C / 4)) * P1 + (q % (C / 4)) * 4) = r1[i];
      *reinterpret_cast<float4*>(s2 + (q / (CH / 4)) * P2 + (q % (CH / 4)) * 4) = r2[i];
    }
  }
};

__device__ __forceinline__ float ff_sum4(float v) {  // over the four lane groups of a position
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// acc1[f] (f < CH/16): rows = hidden 16 f + 4 lg + r of the chunk, col = my position; B operand = bq[cc][e]
template <int C, int CH>
__device__ __forceinline__ void ff_gemm_first(const float* __restrict__ s1, const float (&bq)[C / 16][4], f32x4 (&acc)[CH / 16], int li, int lg) {
  constexpr int P1 = C + 4, NF = CH / 16, SP = NF >= 4 ? 1 : 4 / NF;  // SP partial sums per fragment: >= 4 independent MFMA chains in flight
  f32x4 part[NF][SP];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int p = 0; p < SP; ++p) part[f][p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int cc = 0; cc < C / 16; ++cc) {
    f32x4 a[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) a[f] = *reinterpret_cast<const f32x4*>(s1 + (16 * f + li) * P1 + 16 * cc + 4 * lg);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int f = 0; f < NF; ++f) part[f][cc % SP] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f][e], bq[cc][e], part[f][cc % SP], 0, 0, 0);
  }
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    acc[f] = part[f][0];
#pragma unroll
    for (int p = 1; p < SP; ++p) acc[f] += part[f][p];
  }
}

// out[fc] (fc < C/16) += A2 chunk (rows = channel 16 fc + li, k = hidden of the chunk) x hv (B operand as it sits in acc1's layout)
template <int C, int CH>
__device__ __forceinline__ void ff_gemm_second(const float* __restrict__ s2, const f32x4 (&hv)[CH / 16], f32x4 (&out)[C / 16], int li, int lg) {
  constexpr int P2 = CH + 4, NF = CH / 16;
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    f32x4 a[C / 16];
#pragma unroll
    for (int fc = 0; fc < C / 16; ++fc) a[fc] = *reinterpret_cast<const f32x4*>(s2 + (16 * fc + li) * P2 + 16 * f + 4 * lg);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int fc = 0; fc < C / 16; ++fc) out[fc] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[fc][r], hv[f][r], out[fc], 0, 0, 0);
  }
}

template <int C, int CH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void ffn_block_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b, float eps,
                                                            const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                                                            const float* __restrict__ b2, float* __restrict__ xh, float* __restrict__ mu,
                                                            float* __restrict__ rs, float* __restrict__ h, float* __restrict__ z, float* __restrict__ y,
                                                            long long N) {
  constexpr int HID = 4 * C, NCH = HID / CH, NF = CH / 16, P1 = C + 4, P2 = CH + 4;
  __shared__ __attribute__((aligned(16))) float s1[2][CH * P1];
  __shared__ __attribute__((aligned(16))) float s2[2][C * P2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const long long pos = (long long)blockIdx.x * FF_POS + wave * 16 + li;  // (N % 64 == 0: every lane has a position)
  FfChunk<C, CH> st;
  st.fetch(W1, W2, HID, 0, tid);
  // this lane's channels of its position, LayerNorm in registers
  float xq[C / 16][4];
#pragma unroll
  for (int cc = 0; cc < C / 16; ++cc)
#pragma unroll
    for (int e = 0; e < 4; ++e) xq[cc][e] = x[(long long)(16 * cc + 4 * lg + e) * N + pos];
  float s = 0.f;
#pragma unroll
  for (int cc = 0; cc < C / 16; ++cc)
#pragma unroll
    for (int e = 0; e < 4; ++e) s += xq[cc][e];
  const float mean = ff_sum4(s) / (float)C;
  float v = 0.f;
#pragma unroll
  for (int cc = 0; cc < C / 16; ++cc)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = xq[cc][e] - mean;
      v += d * d;
    }
  const float rinv = 1.f / (sqrtf(ff_sum4(v) / (float)C) + eps);
  if (lg == 0) { mu[pos] = mean; rs[pos] = rinv; }
#pragma unroll
  for (int cc = 0; cc < C / 16; ++cc) {
    const float4 g4 = *reinterpret_cast<const float4*>(g + 16 * cc + 4 * lg), b4 = *reinterpret_cast<const float4*>(b + 16 * cc + 4 * lg);
    const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xq[cc][e] = (xq[cc][e] - mean) * rinv * gg[e] + bb[e];
      xh[(long long)(16 * cc + 4 * lg + e) * N + pos] = xq[cc][e];
    }
  }
  f32x4 yacc[C / 16];
#pragma unroll
  for (int fc = 0; fc < C / 16; ++fc) yacc[fc] = f32x4{0.f, 0.f, 0.f, 0.f};
  st.stash(s1[0], s2[0], tid);
  __syncthreads();
  for (int j = 0; j < NCH; ++j) {
    const int buf = j & 1;
    if (j + 1 < NCH) st.fetch(W1, W2, HID, j + 1, tid);  // in flight under this chunk's products
    f32x4 acc[NF];
    float4 bias[NF];  // (fetched ahead of the product they are added to)
#pragma unroll
    for (int f = 0; f < NF; ++f) bias[f] = *reinterpret_cast<const float4*>(b1 + j * CH + 16 * f + 4 * lg);
    ff_gemm_first<C, CH>(s1[buf], xq, acc, li, lg);
    // bias, GELU and GELU' (one erf for both, as the GEMM epilogue's code 3 does), the saved tensors
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int hid0 = j * CH + 16 * f + 4 * lg;
      const float bb[4] = {bias[f].x, bias[f].y, bias[f].z, bias[f].w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pre = acc[f][r] + bb[r];
        const float cdf = 0.5f * (1.f + erff(pre * 0.70710678118654752440f));
        const float hval = pre * cdf;
        const long long o = (long long)(hid0 + r) * N + pos;
        h[o] = hval;
        z[o] = cdf + pre * expf(-0.5f * pre * pre) * 0.39894228040143267794f;
        acc[f][r] = hval;
      }
    }
    ff_gemm_second<C, CH>(s2[buf], acc, yacc, li, lg);
    if (j + 1 < NCH) st.stash(s1[buf ^ 1], s2[buf ^ 1], tid);  // (the other buffer: last read during chunk j - 1, before the previous barrier)
    __syncthreads();
  }
#pragma unroll
  for (int fc = 0; fc < C / 16; ++fc) {
    const float4 bv = *reinterpret_cast<const float4*>(b2 + 16 * fc + 4 * lg);
    const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long o = (long long)(16 * fc + 4 * lg + r) * N + pos;
      y[o] = yacc[fc][r] + bb[r] + x[o];
    }
  }
}

// backward: W2p = the forward's packed out_conv operand [HID][C], W1p = the packed in_conv operand [C][HID] (advhip_conv3d_pack_weight_f32 /
// advhip_pack_weights_multi_f32 mode 0); z = GELU'(pre-activation) saved by the forward.
//   dz = (W2^T dy) * z            (saved: the weight gradient dW1 = dz xh^T contracts with it)
//   dxh = W1^T dz
//   dx = r (dxh g - mean_c(dxh g)) - r^2 / sigma * mean_c(dxh g xc) xc + dy,   xc = x - mu, r = rs, sigma = 1 / r - eps     (chan_layernorm_bwd's formula)
//   dgb[block][0..C) = sum over the block's positions of dxh xc r,  dgb[block][C..2C) = sum of dxh     (partial sums, one row per workgroup)
template <int C, int CH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void ffn_block_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ mu, const float* __restrict__ rs, float eps,
                                                            const float* __restrict__ z, const float* __restrict__ W2p, const float* __restrict__ W1p,
                                                            float* __restrict__ dz, float* __restrict__ dx, float* __restrict__ dgb, long long N) {
  constexpr int HID = 4 * C, NCH = HID / CH, NF = CH / 16, P1 = C + 4, P2 = CH + 4;
  __shared__ __attribute__((aligned(16))) float s1[2][CH * P1];
  __shared__ __attribute__((aligned(16))) float s2[2][C * P2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const long long pos = (long long)blockIdx.x * FF_POS + wave * 16 + li;
  FfChunk<C, CH> st;
  st.fetch(W2p, W1p, HID, 0, tid);
  float dq[C / 16][4];
#pragma unroll
  for (int cc = 0; cc < C / 16; ++cc)
#pragma unroll
    for (int e = 0; e < 4; ++e) dq[cc][e] = dy[(long long)(16 * cc + 4 * lg + e) * N + pos];
  f32x4 xacc[C / 16];
#pragma unroll
  for (int fc = 0; fc < C / 16; ++fc) xacc[fc] = f32x4{0.f, 0.f, 0.f, 0.f};
  st.stash(s1[0], s2[0], tid);
  __syncthreads();
  for (int j = 0; j < NCH; ++j) {
    const int buf = j & 1;
    if (j + 1 < NCH) st.fetch(W2p, W1p, HID, j + 1, tid);
    f32x4 acc[NF];
    float zq[NF][4];  // (in flight under the product they multiply: a load issued after it would expose its latency once per chunk)
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) zq[f][r] = z[(long long)(j * CH + 16 * f + 4 * lg + r) * N + pos];
    ff_gemm_first<C, CH>(s1[buf], dq, acc, li, lg);
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = acc[f][r] * zq[f][r];
        dz[(long long)(j * CH + 16 * f + 4 * lg + r) * N + pos] = d;
        acc[f][r] = d;
      }
    ff_gemm_second<C, CH>(s2[buf], acc, xacc, li, lg);
    if (j + 1 < NCH) st.stash(s1[buf ^ 1], s2[buf ^ 1], tid);
    __syncthreads();
  }
  // LayerNorm backward on the lane's channels (xacc[fc][r] = dxh of channel 16 fc + 4 lg + r: the distribution dq / x came in)
  const float mean = mu[pos], r = rs[pos];
  const float sigma = 1.f / r - eps;
  float m1 = 0.f, m2 = 0.f;
  float xc[C / 16][4], dg_[C / 16][4];
#pragma unroll
  for (int fc = 0; fc < C / 16; ++fc) {
    const float4 g4 = *reinterpret_cast<const float4*>(g + 16 * fc + 4 * lg);
    const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xc[fc][e] = x[(long long)(16 * fc + 4 * lg + e) * N + pos] - mean;
      dg_[fc][e] = xacc[fc][e] * gg[e];
      m1 += dg_[fc][e];
      m2 += dg_[fc][e] * xc[fc][e];
    }
  }
  m1 = ff_sum4(m1) / (float)C;
  m2 = ff_sum4(m2) / (float)C;
  const float k2 = r * r / sigma * m2;
#pragma unroll
  for (int fc = 0; fc < C / 16; ++fc)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      dx[(long long)(16 * fc + 4 * lg + e) * N + pos] = r * (dg_[fc][e] - m1) - k2 * xc[fc][e] + dq[fc][e];
  // partial sums of dg / db over the workgroup's 64 positions: 16 positions of a wave by xor-shuffles, the four waves through LDS
  __syncthreads();  // (s1 is free: every wave is past its last fragment read)
  float* red = &s1[0][0];  // [4 waves][2 C]
#pragma unroll
  for (int fc = 0; fc < C / 16; ++fc)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = xacc[fc][e] * xc[fc][e] * r, bsum = xacc[fc][e];
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        bsum += __shfl_xor(bsum, off, 64);
      }
      if (li == 0) {
        red[wave * 2 * C + 16 * fc + 4 * lg + e] = a;
        red[wave * 2 * C + C + 16 * fc + 4 * lg + e] = bsum;
      }
    }
  __syncthreads();
  for (int i = tid; i < 2 * C; i += 256) dgb[(long long)blockIdx.x * 2 * C + i] = (red[i] + red[2 * C + i]) + (red[4 * C + i] + red[6 * C + i]);
}

}  // namespace advhip

using namespace advhip;

extern "C" int64_t advhip_ffn_block_partial_rows(int64_t N) { return N / FF_POS; }

extern "C" int advhip_ffn_block_fwd_f32(const float* x, const float* ln_g, const float* ln_b, float eps, const float* w1, const float* b1,
                                        const float* w2, const float* b2, float* xh, float* mu, float* rs, float* h, float* z, float* y, int32_t C,
                                        int64_t N, void* stream) {
  ADVHIP_REQUIRE(x && ln_g && ln_b && w1 && b1 && w2 && b2 && xh && mu && rs && h && z && y, "ffn_block_fwd: null pointer");
  ADVHIP_REQUIRE((C == 64 || C == 128) && N > 0 && N % FF_POS == 0 && N * 4 * C < (1ll << 31),
                 "ffn_block_fwd: C = 64 or 128 channels and a multiple of 64 positions (C=%d, N=%lld)", C, (long long)N);
  const dim3 grid((unsigned)(N / FF_POS));
  if (C == 64)
    hipLaunchKernelGGL((ffn_block_fwd_kernel<64, 64>), grid, dim3(256), 0, (hipStream_t)stream, x, ln_g, ln_b, eps, w1, b1, w2, b2, xh, mu, rs, h, z, y, (long long)N);
  else
    hipLaunchKernelGGL((ffn_block_fwd_kernel<128, 32>), grid, dim3(256), 0, (hipStream_t)stream, x, ln_g, ln_b, eps, w1, b1, w2, b2, xh, mu, rs, h, z, y, (long long)N);
  return check_launch("ffn_block_fwd");
}

extern "C" int advhip_ffn_block_bwd_f32(const float* dy, const float* x, const float* ln_g, const float* mu, const float* rs, float eps, const float* z,
                                        const float* w2_packed, const float* w1_packed, float* dz, float* dx, float* dgb_partial, int32_t C, int64_t N,
                                        void* stream) {
  ADVHIP_REQUIRE(dy && x && ln_g && mu && rs && z && w2_packed && w1_packed && dz && dx && dgb_partial, "ffn_block_bwd: null pointer");
  ADVHIP_REQUIRE((C == 64 || C == 128) && N > 0 && N % FF_POS == 0 && N * 4 * C < (1ll << 31),
                 "ffn_block_bwd: C = 64 or 128 channels and a multiple of 64 positions (C=%d, N=%lld)", C, (long long)N);
  const dim3 grid((unsigned)(N / FF_POS));
  if (C == 64)
    hipLaunchKernelGGL((ffn_block_bwd_kernel<64, 64>), grid, dim3(256), 0, (hipStream_t)stream, dy, x, ln_g, mu, rs, eps, z, w2_packed, w1_packed, dz, dx, dgb_partial, (long long)N);
  else
    hipLaunchKernelGGL((ffn_block_bwd_kernel<128, 32>), grid, dim3(256), 0, (hipStream_t)stream, dy, x, ln_g, mu, rs, eps, z, w2_packed, w1_packed, dz, dx, dgb_partial, (long long)N);
  return check_launch("ffn_block_bwd");
}
