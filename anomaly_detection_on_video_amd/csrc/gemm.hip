// Batched strided fp32 GEMM on the f32-input MFMA (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate) with a
// fused epilogue, for the GEMM-shaped work of the path that is not a convolution over an NCDHW activation:
//   * NonLocalBlock (src/i3d.py:124-195): theta^T.phi (bmm, :171-173), g.p^T (bmm, :178)
//   * MGFN scorer (src/models/mgfn/modeling_mgfn.py): the 1x1 / k=3 Conv1d, Linear and attention einsum contractions and
//     their backward products (weights^T.dY, dY.X^T), with bias / GELU / channel-LayerNorm-fold / residual epilogues.
//
//   C[b, m, n] = epi( alpha * sum_k A[b, m, k] * B[b, k, n] )
//   A[b, m, k] at A + b*sAb + m*sAm + k*sAk,  B[b, k, n] at B + b*sBb + k*sBk + n*sBn,  C likewise (elements).
//
// Each operand is staged through LDS as [k][m] / [k][n] (the layout of the conv kernels: one ds_read_b128 per lane yields
// the operands of four row-interleaved 16-row fragments).  A template flag says which of an operand's two strides is 1:
//   row-contiguous (sAm == 1 / sBn == 1): threads run along m (n), one dword per thread per k-row -- coalesced rows;
//   k-contiguous   (sAk == 1 / sBk == 1): a thread loads 4 consecutive k of one row as one 16-byte vector (the next k-tile
//                  takes the next 16 bytes of the same cache line) and scatters them down its LDS column.
// Everything is bounds-checked (M, N, K arbitrary; out-of-range elements read as 0 and are not stored).
#include <algorithm>

#include "common.h"

namespace advhip {

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K, batch;
  long long sAb, sAm, sAk, sBb, sBk, sBn, sCb, sCm, sCn;
  float alpha;
  // epilogue:  v = alpha*acc;  LN fold: v = v*rs[col] - u[row]*(mu[col]*rs[col]) (see advhip.h);  v += bias;  v = act(v);
  //            v += beta * R;  C = v        (row = m index, col = n index of C)
  const float* bias_m;  // [M] or null
  const float* bias_n;  // [N] or null
  const float* ln_u;    // [M] or null: row vector of the LayerNorm fold
  const float* ln_mu;   // [batch? no: N] column means
  const float* ln_rs;   // [N] column reciprocal (std + eps)
  const float* R;       // residual, C's strides, or null
  float beta;
  int act;              // 0 none, 1 relu, 2 gelu (erf)
  int tiles_m, tiles_n;
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

template <int BM, int BN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void bgemm_f32_kernel(const GemmArgs g) {
  constexpr int BK = 16;
  constexpr int FM = BM / 32, FN = BN / 32;
  static_assert((FM == 2 || FM == 4) && (FN == 2 || FN == 4), "wave tile 32 or 64 wide");
  __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = g.tiles_m * g.tiles_n;
  const int b = blockIdx.x / ntiles;
  const int L = xcd_remap((int)(blockIdx.x - b * ntiles), ntiles);
  const int tile_m = L / g.tiles_n, tile_n = L - tile_m * g.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const float* __restrict__ A = g.A + (long long)b * g.sAb;
  const float* __restrict__ B = g.B + (long long)b * g.sBb;

  // ---- operand staging ----------------------------------------------------------------------------------------
  constexpr int RA = A_KC ? BM * BK / 4 / 256 : BM * BK / 256;  // vectors (A_KC) or dwords per thread per k-tile
  constexpr int RB = B_KC ? BN * BK / 4 / 256 : BN * BK / 256;
  static_assert(RA >= 1 && RB >= 1, "tile too small for 256 threads");
  float ra[A_KC ? RA * 4 : RA], rb[B_KC ? RB * 4 : RB];
  const bool vecA = A_KC && (g.sAm % 4 == 0) && (g.K % 4 == 0) && (((size_t)A & 15) == 0);
  const bool vecB = B_KC && (g.sBn % 4 == 0) && (g.K % 4 == 0) && (((size_t)B & 15) == 0);

  auto load = [&](int k0) {
    if constexpr (A_KC) {
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        const int idx = tid + 256 * j;
        const int ml = idx % BM, kq = idx / BM;  // 4 consecutive k (kq*4 ..) of row ml
        const int m = m0 + ml, k = k0 + kq * 4;
        const float* p = A + (long long)m * g.sAm + k;
        if (m < g.M && vecA && k + 3 < g.K) {
          const float4 t = *reinterpret_cast<const float4*>(p);
          ra[4 * j] = t.x; ra[4 * j + 1] = t.y; ra[4 * j + 2] = t.z; ra[4 * j + 3] = t.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) ra[4 * j + e] = (m < g.M && k + e < g.K) ? p[e] : 0.f;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        const int idx = tid + 256 * j;
        const int ml = idx % BM, kr = idx / BM;
        const int m = m0 + ml, k = k0 + kr;
        ra[j] = (m < g.M && k < g.K) ? A[(long long)m * g.sAm + (long long)k * g.sAk] : 0.f;
      }
    }
    if constexpr (B_KC) {
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const int idx = tid + 256 * j;
        const int nl = idx % BN, kq = idx / BN;
        const int n = n0 + nl, k = k0 + kq * 4;
        const float* p = B + (long long)n * g.sBn + k;
        if (n < g.N && vecB && k + 3 < g.K) {
          const float4 t = *reinterpret_cast<const float4*>(p);
          rb[4 * j] = t.x; rb[4 * j + 1] = t.y; rb[4 * j + 2] = t.z; rb[4 * j + 3] = t.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) rb[4 * j + e] = (n < g.N && k + e < g.K) ? p[e] : 0.f;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const int idx = tid + 256 * j;
        const int nl = idx % BN, kr = idx / BN;
        const int n = n0 + nl, k = k0 + kr;
        rb[j] = (n < g.N && k < g.K) ? B[(long long)k * g.sBk + (long long)n * g.sBn] : 0.f;
      }
    }
  };
  auto store = [&](int buf) {
    if constexpr (A_KC) {
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        const int idx = tid + 256 * j;
        const int ml = idx % BM, kq = idx / BM;
#pragma unroll
        for (int e = 0; e < 4; ++e) As[buf][kq * 4 + e][ml] = ra[4 * j + e];
      }
    } else {
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        const int idx = tid + 256 * j;
        As[buf][idx / BM][idx % BM] = ra[j];
      }
    }
    if constexpr (B_KC) {
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const int idx = tid + 256 * j;
        const int nl = idx % BN, kq = idx / BN;
#pragma unroll
        for (int e = 0; e < 4; ++e) Bs[buf][kq * 4 + e][nl] = rb[4 * j + e];
      }
    } else {
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const int idx = tid + 256 * j;
        Bs[buf][idx / BN][idx % BN] = rb[j];
      }
    }
  };

  // ---- MFMA loop (fragment scheme of conv_igemm.hip: row r of fragment j = offset FM*r + j) ---------------------------
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  const int a_col = wm * (BM / 2) + FM * li;
  const int b_col = wn * (BN / 2) + FN * li;
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (g.K + BK - 1) / BK;
  load(0);
  store(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load((kt + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      float av[FM], bv[FN];
      if constexpr (FM == 4) {
        const float4 t = *reinterpret_cast<const float4*>(&As[cur][4 * ks + lg][a_col]);
        av[0] = t.x; av[1] = t.y; av[2] = t.z; av[3] = t.w;
      } else {
        const float2 t = *reinterpret_cast<const float2*>(&As[cur][4 * ks + lg][a_col]);
        av[0] = t.x; av[1] = t.y;
      }
      if constexpr (FN == 4) {
        const float4 t = *reinterpret_cast<const float4*>(&Bs[cur][4 * ks + lg][b_col]);
        bv[0] = t.x; bv[1] = t.y; bv[2] = t.z; bv[3] = t.w;
      } else {
        const float2 t = *reinterpret_cast<const float2*>(&Bs[cur][4 * ks + lg][b_col]);
        bv[0] = t.x; bv[1] = t.y;
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: acc[jm][jn][r] of lane (li, lg): m = m0 + wm*BM/2 + FM*(4*lg + r) + jm,  n = n0 + wn*BN/2 + FN*li + jn ----
  float* __restrict__ C = g.C + (long long)b * g.sCb;
  const float* __restrict__ R = g.R ? g.R + (long long)b * g.sCb : nullptr;
  const int m_lane = m0 + wm * (BM / 2) + FM * 4 * lg;
  const int n_lane = n0 + wn * (BN / 2) + FN * li;
#pragma unroll
  for (int jn = 0; jn < FN; ++jn) {
    const int n = n_lane + jn;
    if (n >= g.N) continue;
    const float bn_ = g.bias_n ? g.bias_n[n] : 0.f;
    float rs = 1.f, mrs = 0.f;
    if (g.ln_u) { rs = g.ln_rs[(long long)b * g.N + n]; mrs = g.ln_mu[(long long)b * g.N + n] * rs; }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int jm = 0; jm < FM; ++jm) {
        const int m = m_lane + FM * r + jm;
        if (m >= g.M) continue;
        float v = g.alpha * acc[jm][jn][r];
        if (g.ln_u) v = v * rs - g.ln_u[m] * mrs;
        v += bn_;
        if (g.bias_m) v += g.bias_m[m];
        if (g.act == 1) v = fmaxf(v, 0.f);
        else if (g.act == 2) v = gelu_erf(v);
        const long long o = (long long)m * g.sCm + (long long)n * g.sCn;
        if (R) v += g.beta * R[o];
        C[o] = v;
      }
  }
}

// Row softmax: y[r, :] = softmax(x[r, :] * scale) over n contiguous elements; one wavefront per row (n <= 16384).
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, long long rows, int n,
                                                           float scale) {
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * (long long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = x + row * n;
  float* q = y + row * n;
  float mx = -INFINITY;
  for (int i = lane; i < n; i += 64) mx = fmaxf(mx, p[i] * scale);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  float s = 0.f;
  for (int i = lane; i < n; i += 64) s += expf(p[i] * scale - mx);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  const float inv = 1.f / s;
  for (int i = lane; i < n; i += 64) q[i] = expf(p[i] * scale - mx) * inv;
}

}  // namespace advhip

using namespace advhip;

extern "C" int advhip_bgemm_f32(const advhip_gemm_desc* d, const float* A, const float* B, float* C, void* stream) {
  ADVHIP_REQUIRE(d && A && B && C, "bgemm: null pointer");
  ADVHIP_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->batch > 0, "bgemm: non-positive extent (M=%d N=%d K=%d batch=%d)", d->M, d->N, d->K, d->batch);
  ADVHIP_REQUIRE((d->sAm == 1 || d->sAk == 1) && (d->sBn == 1 || d->sBk == 1), "bgemm: each operand needs one unit stride");
  ADVHIP_REQUIRE(d->act >= 0 && d->act <= 2, "bgemm: unknown activation %d", d->act);
  ADVHIP_REQUIRE(!d->ln_u == !d->ln_mu && !d->ln_u == !d->ln_rs, "bgemm: the LayerNorm fold needs ln_u, ln_mu and ln_rs together");
  GemmArgs g;
  g.A = A; g.B = B; g.C = C;
  g.M = d->M; g.N = d->N; g.K = d->K; g.batch = d->batch;
  g.sAb = d->sAb; g.sAm = d->sAm; g.sAk = d->sAk;
  g.sBb = d->sBb; g.sBk = d->sBk; g.sBn = d->sBn;
  g.sCb = d->sCb; g.sCm = d->sCm; g.sCn = d->sCn;
  g.alpha = d->alpha; g.beta = d->beta; g.act = d->act;
  g.bias_m = d->bias_m; g.bias_n = d->bias_n; g.ln_u = d->ln_u; g.ln_mu = d->ln_mu; g.ln_rs = d->ln_rs; g.R = d->residual;
  // tile: 128 x 64 when there are plenty of rows, else 64 x 64 (more workgroups for the 256 CUs)
  const bool a_kc = d->sAm != 1, b_kc = d->sBn != 1;
  const long long t64 = (long long)((d->M + 63) / 64) * ((d->N + 63) / 64) * d->batch;
  const bool big = t64 >= 4096 && d->M >= 128;
  const int BM = big ? 128 : 64;
  g.tiles_m = (d->M + BM - 1) / BM;
  g.tiles_n = (d->N + 63) / 64;
  const long long blocks = (long long)g.tiles_m * g.tiles_n * d->batch;
  ADVHIP_REQUIRE(blocks < (1ll << 31), "bgemm: too many tiles");
  const dim3 grid((unsigned)blocks);
  hipStream_t st = (hipStream_t)stream;
#define ADVHIP_BGEMM(BM_)                                                                                      \
  if (a_kc && b_kc) hipLaunchKernelGGL((bgemm_f32_kernel<BM_, 64, true, true>), grid, dim3(256), 0, st, g);      \
  else if (a_kc) hipLaunchKernelGGL((bgemm_f32_kernel<BM_, 64, true, false>), grid, dim3(256), 0, st, g);        \
  else if (b_kc) hipLaunchKernelGGL((bgemm_f32_kernel<BM_, 64, false, true>), grid, dim3(256), 0, st, g);        \
  else hipLaunchKernelGGL((bgemm_f32_kernel<BM_, 64, false, false>), grid, dim3(256), 0, st, g);
  if (big) { ADVHIP_BGEMM(128) } else { ADVHIP_BGEMM(64) }
#undef ADVHIP_BGEMM
  return check_launch("bgemm");
}

extern "C" int advhip_softmax_rows_f32(const float* x, float* y, int64_t rows, int32_t n, float scale, void* stream) {
  ADVHIP_REQUIRE(x && y && rows > 0 && n > 0, "softmax_rows: bad arguments");
  const long long blocks = (rows + 3) / 4;
  ADVHIP_REQUIRE(blocks < (1ll << 31), "softmax_rows: too many rows");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, (long long)rows, n, scale);
  return check_launch("softmax_rows");
}
