// Generic im2col-free implicit-GEMM 3-D convolution for gfx950, fp32 MFMA, fused BN/residual/ReLU.
//
//   y[b, co, to, ho, wo] = act( scale[co] * sum_k A[m, k] * Wp[k, co] + shift[co] (+ res) )
//   m = (b, to, ho, wo) flattened,  k = (ci, dt, dh, dw) flattened (torch weight order),
//   A[m, k] = x[b, ci, to*st-pt+dt, ho*sh-ph+dh, wo*sw-pw+dw]   (0 outside the input)
//
// The A operand is never materialised in HBM: each block gathers its [BK x BM] slice straight
// from the NCDHW input into LDS (m is the contiguous axis of both the input rows and the LDS
// tile, so the gathers are coalesced along m), double-buffered against the MFMA loop.
//
// MFMA: v_mfma_f32_16x16x4_f32 (exact fp32, 32-cycle issue per SIMD).  A-operand lane map is
// A[row = lane&15][k = lane>>4]; with the LDS tile laid out [k][m] one ds_read_b128 per lane
// fetches the operands of FOUR row-interleaved 16-row fragments (row r of fragment j = m-offset
// 4r+j), and likewise for B, so a 64x64 wave tile issues 2 LDS reads per 16 MFMAs.  With
// 128-float rows those b128 reads are bank-conflict free (MI355X_MICROARCH.md, LDS table).
// The row interleave makes each lane own 16 consecutive m of one output channel, so the NCDHW
// epilogue stores are 16-byte vectors.
//
// Replaces nn.Conv3d + nn.BatchNorm3d(eval) + residual add + nn.ReLU of
// /root/reference/src/i3d.py:98-121, 262-272, 303-305.
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace advhip {

struct ConvArgs {
  const float* x;
  const float* w;     // [Kpad][Cout]
  const int4* ktab;   // [Kpad] {element offset, dt, dh, dw}
  const float* scale;
  const float* shift;
  const float* res;   // nullable, same shape as y
  float* y;
  int B, Cin, T, H, W, Cout;
  int st, sh, sw, pt, ph, pw;
  int To, Ho, Wo;
  int M, Kpad;
  int THWo, HWo, HW, THW;
  int tiles_m, tiles_n;
  int relu, vw;
};

template <int VW>
__device__ __forceinline__ void vec_load(const float* p, float (&v)[VW]) {
  if constexpr (VW == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else if constexpr (VW == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  } else {
    v[0] = *p;
  }
}
template <int VW>
__device__ __forceinline__ void vec_store(float* p, const float (&v)[VW]) {
  if constexpr (VW == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  } else if constexpr (VW == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  } else {
    *p = v[0];
  }
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void conv3d_igemm_f32_kernel(const ConvArgs a) {
  constexpr int BK = 16;
  constexpr int FM = BM / 32, FN = BN / 32;  // 16x16 fragments per wave along M / N
  constexpr int KR = 256 / BM;               // k-rows covered by one pass of the 256 threads
  constexpr int RA = BK / KR;                // A elements gathered per thread per k-tile
  constexpr int RB = BK * BN / 4 / 256;      // float4 of B per thread per k-tile
  static_assert(FM == 4 || FM == 2, "wave M tile must be 64 or 32");
  static_assert(FN == 4 || FN == 2, "wave N tile must be 64 or 32");

  __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = xcd_remap(blockIdx.x, a.tiles_m * a.tiles_n);
  const int tile_n = L % a.tiles_n, tile_m = L / a.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- gather coordinates of this thread's A column (one m, RA different k per tile) ----------
  const int ml = tid % BM;
  const int kr = __builtin_amdgcn_readfirstlane(tid / BM);  // wave-uniform (BM >= 64)
  const int m = m0 + ml;
  const bool mv = m < a.M;
  int b = 0, ot = 0, oh = 0, ow = 0;
  if (mv) {
    b = m / a.THWo;
    const int p = m - b * a.THWo;
    ot = p / a.HWo;
    const int q = p - ot * a.HWo;
    oh = q / a.Wo;
    ow = q - oh * a.Wo;
  }
  const int it0 = ot * a.st - a.pt, ih0 = oh * a.sh - a.ph, iw0 = ow * a.sw - a.pw;
  const unsigned Tlim = mv ? (unsigned)a.T : 0u;  // m out of range -> every tap invalid
  const int mbase = b * a.Cin * a.THW + it0 * a.HW + ih0 * a.W + iw0;

  float ra[RA];
  float rb[RB][4];

  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const int4 e = a.ktab[k0 + kr + KR * j];
      const bool v = (unsigned)(it0 + e.y) < Tlim && (unsigned)(ih0 + e.z) < (unsigned)a.H &&
                     (unsigned)(iw0 + e.w) < (unsigned)a.W;
      ra[j] = v ? a.x[mbase + e.x] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int idx = tid + 256 * j;
      const int row = idx / (BN / 4), c4 = idx % (BN / 4);
      const float4 t = *reinterpret_cast<const float4*>(a.w + (size_t)(k0 + row) * a.Cout + n0 + c4 * 4);
      rb[j][0] = t.x; rb[j][1] = t.y; rb[j][2] = t.z; rb[j][3] = t.w;
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int j = 0; j < RA; ++j) As[buf][kr + KR * j][ml] = ra[j];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int idx = tid + 256 * j;
      const int row = idx / (BN / 4), c4 = idx % (BN / 4);
      *reinterpret_cast<float4*>(&Bs[buf][row][c4 * 4]) = make_float4(rb[j][0], rb[j][1], rb[j][2], rb[j][3]);
    }
  };

  // ---- MFMA main loop ---------------------------------------------------------------------
  const int wm = wave >> 1, wn = wave & 1;  // 2 x 2 waves
  const int li = lane & 15, lg = lane >> 4;
  const int a_col = wm * (BM / 2) + FM * li;
  const int b_col = wn * (BN / 2) + FN * li;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = a.Kpad / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tiles((kt + 1) * BK);
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
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: scale/shift (+res) (+relu), NCDHW store ------------------------------------
  // accumulator element acc[jm][jn][r] of lane (li, lg):
  //   m = m0 + wm*BM/2 + FM*(4*lg + r) + jm,   n = n0 + wn*BN/2 + FN*li + jn
  const int m_lane = m0 + wm * (BM / 2) + FM * 4 * lg;  // first of FM*4 consecutive m
  const int n_lane = n0 + b_col;
  auto emit = [&](auto vw_tag) {
    constexpr int VW = decltype(vw_tag)::value;  // divides FM and THWo
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int v0 = 0; v0 < FM; v0 += VW) {
        const int mm = m_lane + FM * r + v0;
        if (mm >= a.M) continue;
        const int bb = mm / a.THWo;
        const int pp = mm - bb * a.THWo;
#pragma unroll
        for (int jn = 0; jn < FN; ++jn) {
          const int n = n_lane + jn;
          const float sc = a.scale[n], sf = a.shift[n];
          const size_t o = (size_t)(bb * a.Cout + n) * a.THWo + pp;
          float vals[VW];
#pragma unroll
          for (int e = 0; e < VW; ++e) vals[e] = acc[v0 + e][jn][r] * sc + sf;
          if (a.res) {
            float rv[VW];
            vec_load<VW>(a.res + o, rv);
#pragma unroll
            for (int e = 0; e < VW; ++e) vals[e] += rv[e];
          }
          if (a.relu) {
#pragma unroll
            for (int e = 0; e < VW; ++e) vals[e] = fmaxf(vals[e], 0.f);
          }
          vec_store<VW>(a.y + o, vals);
        }
      }
    }
  };
  if constexpr (FM == 4) {
    if (a.vw == 4) emit(std::integral_constant<int, 4>{});
    else if (a.vw == 2) emit(std::integral_constant<int, 2>{});
    else emit(std::integral_constant<int, 1>{});
  } else {
    if (a.vw >= 2) emit(std::integral_constant<int, 2>{});
    else emit(std::integral_constant<int, 1>{});
  }
}

// ---- weight packing + gather table -------------------------------------------------------------
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int K, int Kpad) {
  const long long total = (long long)Kpad * Cout;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i / Cout), co = (int)(i % Cout);
    wp[i] = (k < K) ? w[(size_t)co * K + k] : 0.f;
  }
}

__global__ void build_ktab_kernel(int4* __restrict__ ktab, int kt, int kh, int kw, int K, int Kpad, int HW, int W,
                                  int THW) {
  const int taps = kt * kh * kw;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < Kpad; k += gridDim.x * blockDim.x) {
    int4 e;
    if (k < K) {
      const int ci = k / taps, tap = k % taps;
      const int dt = tap / (kh * kw), r = tap % (kh * kw);
      const int dh = r / kw, dw = r % kw;
      e = make_int4(ci * THW + dt * HW + dh * W + dw, dt, dh, dw);
    } else {
      e = make_int4(0, 1 << 28, 0, 0);  // fails the temporal range check: contributes exact zeros
    }
    ktab[k] = e;
  }
}

__global__ void bn_fold_kernel(const float* g, const float* b, const float* mean, const float* var, float eps, int C,
                               float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float s = g[c] / sqrtf(var[c] + eps);
    scale[c] = s;
    shift[c] = b[c] - mean[c] * s;
  }
}

static int out_dim(int n, int k, int s, int p) { return (n + 2 * p - k) / s + 1; }

static int validate(const advhip_conv3d_desc* d) {
  ADVHIP_REQUIRE(d != nullptr, "conv3d: null descriptor");
  ADVHIP_REQUIRE(d->B > 0 && d->Cin > 0 && d->T > 0 && d->H > 0 && d->W > 0 && d->Cout > 0, "conv3d: non-positive extent");
  ADVHIP_REQUIRE(d->kt > 0 && d->kh > 0 && d->kw > 0 && d->st > 0 && d->sh > 0 && d->sw > 0, "conv3d: bad kernel/stride");
  ADVHIP_REQUIRE(d->pt >= 0 && d->ph >= 0 && d->pw >= 0, "conv3d: negative padding");
  ADVHIP_REQUIRE(d->T + 2 * d->pt >= d->kt && d->H + 2 * d->ph >= d->kh && d->W + 2 * d->pw >= d->kw,
                 "conv3d: kernel larger than padded input");
  ADVHIP_REQUIRE(d->Cout % 64 == 0, "conv3d: Cout=%d must be a multiple of 64", d->Cout);
  return ADVHIP_OK;
}

}  // namespace advhip

using namespace advhip;

extern "C" int advhip_conv3d_out_dims(const advhip_conv3d_desc* d, int32_t* To, int32_t* Ho, int32_t* Wo) {
  if (int rc = validate(d)) return rc;
  if (To) *To = out_dim(d->T, d->kt, d->st, d->pt);
  if (Ho) *Ho = out_dim(d->H, d->kh, d->sh, d->ph);
  if (Wo) *Wo = out_dim(d->W, d->kw, d->sw, d->pw);
  return ADVHIP_OK;
}

extern "C" int advhip_conv3d_packed_rows(const advhip_conv3d_desc* d) {
  if (int rc = validate(d)) return rc;
  const int K = d->Cin * d->kt * d->kh * d->kw;
  return (K + 15) / 16 * 16;
}

extern "C" int advhip_conv3d_pack_weight_f32(const advhip_conv3d_desc* d, const float* w, float* w_packed,
                                             void* stream) {
  if (int rc = validate(d)) return rc;
  ADVHIP_REQUIRE(w && w_packed, "pack_weight: null pointer");
  const int K = d->Cin * d->kt * d->kh * d->kw;
  const int Kpad = (K + 15) / 16 * 16;
  const long long total = (long long)Kpad * d->Cout;
  const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, w_packed, d->Cout, K, Kpad);
  return check_launch("pack_weight");
}

extern "C" int advhip_conv3d_build_ktab(const advhip_conv3d_desc* d, int32_t* ktab, void* stream) {
  if (int rc = validate(d)) return rc;
  ADVHIP_REQUIRE(ktab, "build_ktab: null pointer");
  const int K = d->Cin * d->kt * d->kh * d->kw;
  const int Kpad = (K + 15) / 16 * 16;
  hipLaunchKernelGGL(build_ktab_kernel, dim3((Kpad + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<int4*>(ktab), d->kt, d->kh, d->kw, K, Kpad, d->H * d->W, d->W,
                     d->T * d->H * d->W);
  return check_launch("build_ktab");
}

extern "C" int advhip_bn_fold_f32(const float* gamma, const float* beta, const float* mean, const float* var,
                                  float eps, int32_t C, float* scale, float* shift, void* stream) {
  ADVHIP_REQUIRE(gamma && beta && mean && var && scale && shift && C > 0, "bn_fold: bad arguments");
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, var,
                     eps, C, scale, shift);
  return check_launch("bn_fold");
}

namespace advhip {
int launch_stem(const advhip_conv3d_desc* d, const float* x, const float* w_packed, const float* scale,
                const float* shift, float* y, hipStream_t stream);  // conv_stem.hip
}

extern "C" int advhip_conv3d_bn_act_f32(const advhip_conv3d_desc* d, const float* x, const float* w_packed,
                                        const int32_t* ktab, const float* scale, const float* shift,
                                        const float* residual, float* y, void* stream) {
  if (int rc = validate(d)) return rc;
  ADVHIP_REQUIRE(x && w_packed && ktab && scale && shift && y, "conv3d: null pointer");
  ConvArgs a;
  a.x = x; a.w = w_packed; a.ktab = reinterpret_cast<const int4*>(ktab);
  a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
  a.B = d->B; a.Cin = d->Cin; a.T = d->T; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
  a.st = d->st; a.sh = d->sh; a.sw = d->sw; a.pt = d->pt; a.ph = d->ph; a.pw = d->pw;
  a.To = out_dim(d->T, d->kt, d->st, d->pt);
  a.Ho = out_dim(d->H, d->kh, d->sh, d->ph);
  a.Wo = out_dim(d->W, d->kw, d->sw, d->pw);
  const long long in_elems = (long long)d->B * d->Cin * d->T * d->H * d->W;
  const long long M = (long long)d->B * a.To * a.Ho * a.Wo;
  if (in_elems >= (1ll << 31) || M * d->Cout >= (1ll << 32) || M >= (1ll << 31)) {
    set_error("conv3d: tensor too large for 32-bit indexing (in=%lld, out=%lld elements)", in_elems, M * d->Cout);
    return ADVHIP_ERANGE;
  }
  a.M = (int)M;
  const int K = d->Cin * d->kt * d->kh * d->kw;
  a.Kpad = (K + 15) / 16 * 16;
  a.HWo = a.Ho * a.Wo; a.THWo = a.To * a.HWo;
  a.HW = d->H * d->W; a.THW = d->T * a.HW;
  a.relu = d->relu;
  a.vw = (a.THWo % 4 == 0) ? 4 : (a.THWo % 2 == 0 ? 2 : 1);

  int algo = d->algo;
  if (algo == ADVHIP_ALGO_STEM) {
    ADVHIP_REQUIRE(residual == nullptr, "conv3d: stem kernel takes no residual");
    return launch_stem(d, x, w_packed, scale, shift, y, (hipStream_t)stream);
  }
  if (algo == ADVHIP_ALGO_AUTO) {
    // Fill the 256 CUs: prefer the biggest tile that still yields >= 4 workgroups per CU.
    const long long t128 = (M + 127) / 128, t64 = (M + 63) / 64;
    const bool n128 = d->Cout % 128 == 0;
    if (n128 && t128 * (d->Cout / 128) >= 1024) algo = ADVHIP_ALGO_IGEMM_128x128;
    else if (t128 * (d->Cout / 64) >= 1024) algo = ADVHIP_ALGO_IGEMM_128x64;
    else if (n128 && t64 * (d->Cout / 128) >= 768) algo = ADVHIP_ALGO_IGEMM_64x128;
    else algo = ADVHIP_ALGO_IGEMM_64x64;
  }
  auto launch = [&](auto kern, int BM, int BN) -> int {
    ADVHIP_REQUIRE(d->Cout % BN == 0, "conv3d: Cout=%d not a multiple of the %d-wide N tile", d->Cout, BN);
    a.tiles_m = (int)((M + BM - 1) / BM);
    a.tiles_n = d->Cout / BN;
    hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("conv3d_igemm");
  };
  switch (algo) {
    case ADVHIP_ALGO_IGEMM_128x128: return launch(conv3d_igemm_f32_kernel<128, 128>, 128, 128);
    case ADVHIP_ALGO_IGEMM_128x64: return launch(conv3d_igemm_f32_kernel<128, 64>, 128, 64);
    case ADVHIP_ALGO_IGEMM_64x64: return launch(conv3d_igemm_f32_kernel<64, 64>, 64, 64);
    case ADVHIP_ALGO_IGEMM_64x128: return launch(conv3d_igemm_f32_kernel<64, 128>, 64, 128);
    default: break;
  }
  set_error("conv3d: unknown algo %d", algo);
  return ADVHIP_EINVAL;
}
