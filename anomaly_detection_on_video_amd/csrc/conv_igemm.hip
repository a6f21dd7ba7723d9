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
#include <utility>

#include "common.h"

namespace advhip {

// Division by a launch-invariant divisor as multiply-high + shifts (exact for every 32-bit n).  A
// runtime integer division costs ~40 VALU instructions on gfx950; a new workgroup's index arithmetic
// competes with the MFMA streams of the older workgroups on its CU, so it has to be short.
struct FastDiv {
  unsigned mul, sh1, sh2, d;
  __host__ static FastDiv make(unsigned d) {
    FastDiv f;
    f.d = d;
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    f.mul = (unsigned)(((1ull << l) - d) * (1ull << 32) / d + 1);
    f.sh1 = l > 1 ? 1 : l;
    f.sh2 = l > 0 ? l - 1 : 0;
    if (l == 0) { f.mul = 0; f.sh1 = 0; f.sh2 = 0; }  // d == 1
    return f;
  }
  __device__ __forceinline__ unsigned div(unsigned n) const {
    const unsigned t = __umulhi(mul, n);
    return (t + ((n - t) >> sh1)) >> sh2;
  }
};

// bits [lo, hi) set; 0 <= lo, hi <= 31
__device__ __forceinline__ unsigned bit_range(int lo, int hi) {
  return hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
}
// which taps d in [0, k) of a window starting at coordinate c0 fall inside [0, n): one bit per tap
__device__ __forceinline__ unsigned tap_bits(int c0, int k, int n) {
  const int lo = c0 < 0 ? -c0 : 0;
  const int hi = (n - c0) < k ? (n - c0) : k;
  return bit_range(lo < 31 ? lo : 31, hi < 0 ? 0 : hi);
}

// border class of a window starting at coordinate c0 (= o*s - p >= -p) of a dimension of extent n: lo taps lie before the clip,
// hi past its end (both <= p for every output of a floor-mode conv), class = lo * (p + 1) + hi; taps [lo, k - hi) are inside
__host__ __device__ __forceinline__ int u8_border_class(int c0, int k, int n, int p) {
  const int lo = c0 < 0 ? -c0 : 0;
  int hi = c0 + k - n;
  hi = hi < 0 ? 0 : (hi > p ? p : hi);
  return (lo > p ? p : lo) * (p + 1) + hi;
}

// byte 0 of an LDS dword (as it arrives in a float register) -> its value as fp32: one v_cvt_f32_ubyte0.  Every vector
// instruction beside the MFMAs costs this loop its issue time; the two-instruction exact alternative ((2^23 + b) - 2^23:
// v_or_b32_sdwa + v_add_f32) measured 2.60 ms against 2.50 ms for the stem at B = 32
__device__ __forceinline__ float u8_to_f32(float raw) { return (float)(__builtin_bit_cast(unsigned, raw) & 255u); }

struct ConvArgs {
  const float* x;
  const float* w;     // [Kpad][Cout]
  const int4* ktab;   // [Kpad] {element offset, dt, dh, dw}
  const float* scale;
  const float* shift;
  const float* res;   // nullable, same shape as y
  float* y;           // output, or the split-K partial slabs [splits][B*Cout*THWo]
  int B, Cin, T, H, W, Cout;
  int x_bstride;      // elements between consecutive samples of x (Cin*T*H*W when dense; larger for a channel slice of a wider tensor)
  int y_bstride;      // same for y (the fused-epilogue output only: split-K slabs and the residual are dense)
  int st, sh, sw, pt, ph, pw;
  int To, Ho, Wo;
  int M, Kpad;
  int THWo, HWo, HW, THW;
  int MP;             // positions per sample in the M index space (dTHWo's divisor): THWo, or THWo rounded up to 4 when a 1x1x1
                      // conv on rows that are not a multiple of 4 positions long pads every sample's rows (virtually) so that the
                      // 16-byte A pieces never straddle two samples; positions >= THWo of a sample compute garbage nobody stores
  int tiles_m, tiles_n;
  int relu, vw;       // relu: activation code 0 none, 1 ReLU, 2 GELU (erf)
  int a16;            // 1x1x1 stride-1 conv on 16-byte aligned rows: the A rows go to LDS as 16-byte LDS-DMA pieces (2-deep ring kernels)
  // AMODE 1 (stride-2-along-w stem on column-parity planes, see split_w_kernel): x = xs (B, Cin, T, H, 2, WP)
  const int2* ktab_s2w;  // [Kpad] {byte offset of the tap's 4-column piece relative to the group's window origin, (dt, dh) tap bits}
  int s2w_rowp;          // floats per input row of xs (both parities): 2 * WP
  float* avg_out;     // EPI_AVG: (B, Cout) means over the sample's positions (the conv's own output is never written)
  float* y2;          // nullable: the pre-activation value (after scale/shift/residual), y's addressing -- saved for backward
  const float* dact;  // nullable: z of a GELU, y's addressing (dense): the result is multiplied by gelu'(z) (fused GELU backward)
  // nullable LayerNorm fold (1x1x1 convs over a (C, positions) activation): conv(W.diag(g), x_raw) -> W.LN(x) - W.b:
  //   v = acc * ln_rs[m] - ln_u[n] * ln_mu[m] * ln_rs[m]   before scale / shift (ln_u = row sums of W.diag(g))
  const float* ln_u;
  const float* ln_mu;
  const float* ln_rs;
  int mix_big, mix_small, mix_mbase;  // conv1x1_mixed_tail_kernel: 128-row items, 64-row items, first row of the 64-row tiles
  int splits;         // 1 = fused epilogue; >1 = K cut into that many slices
  long long slab;     // elements per split-K slab (two-launch form: raw partial sums in y, reduced by splitk_reduce_kernel)
  // in-kernel reduction (LDS-DMA kernels): every (tile, slice) workgroup publishes its partial tile, draws a ticket on the
  // tile's arrival counter, and the workgroup that arrives last sums the slices in slice order and runs the fused epilogue
  // brick-ordered m (pooling epilogues): an m-tile is a 2 x BH x BW brick of output positions (t, h, w) of one sample
  int nbh, nbw;       // bricks per sample along h and w (nbt = pooled T)
  int Tp;             // pooled temporal extent (EPI_TPOOL output)
  FastDiv dNb, dNbhw, dNbw;
  float* part;        // [tiles][splits][BM*BN] partial tiles, fragment-major (nullptr = two-launch form)
  unsigned* cnt;      // [tiles] arrival counters, zeroed by a memset node ahead of the launch
  unsigned part_bytes;
  // fast kernel only
  int kt_, kh_, kw_;  // kernel extents (tap decode)
  int pad_off;        // pt*HW + ph*W + pw: makes every per-lane window origin offset non-negative
  unsigned x_bytes;   // buffer range of x (plus pad_off*4)
  unsigned w_bytes;   // buffer range of the packed weights (LDS-DMA kernel)
  FastDiv dTHWo, dHWo, dWo, dTilesN, dSplits;
  // uint8 frame input (stem + maxpool1 kernel, U8 = true): x = resized uint8 frames (F, FH, FW, Cin); sample b of the launch is
  // crop-clip u8_first + b = (clip, crop) of torchvision's TenCrop order (4 corners, centre, then the same five mirrored);
  // T / H / W above are the clip-local extents (frames per clip, crop size) the tap masks are taken against
  int u8_first, u8_FH, u8_FW;
  int u8_ctop, u8_cleft;    // top / left of the centre crop (round-half-to-even, as torchvision)
  float in_std;             // conv of (pixel - mean) / in_std: the mean through pad_corr, 1 / in_std through the BN scale
  const int2* ktab_u8;      // [2][Kpad] {byte offset, tap bits}: as stored, and mirrored along w
  const float* pad_corr;    // [(pt+1)^2][(ph+1)^2][(pw+1)^2][Cout]: -mean * (sum of the weights of the taps inside the clip), indexed per
                            //   dimension by border class = (taps before the clip) * (p + 1) + (taps past its end), see u8_border_class
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

template <int BM, int BN, int BK>
struct IgemmCfg {
  // the four waves of a workgroup as WAVES_M x WAVES_N: 2 x 2, or 4 x 1 for the 256-row tile (a 64 x 64 wave tile -- the 64
  // accumulators, b128 A and B fragment reads and the epilogue staging of the 128 x 128 tile's waves -- over a 64-wide N)
  static constexpr int WAVES_M = BM == 256 ? 4 : 2, WAVES_N = 4 / WAVES_M;
  static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;  // wave tile
  static constexpr int FM = WM / 16, FN = WN / 16;            // 16x16 fragments per wave along M / N
  __device__ static __forceinline__ int wave_m(int wave) { return WAVES_N == 1 ? wave : wave >> 1; }
  __device__ static __forceinline__ int wave_n(int wave) { return WAVES_N == 1 ? 0 : wave & 1; }
  static constexpr int KR = 256 / BM;               // k-rows covered by one pass of the 256 threads
  static constexpr int RA = BK / KR;                // A elements gathered per thread per k-tile
  static constexpr int RB = BK * BN / 4 / 256;      // float4 of B per thread per k-tile
  static constexpr int AB_FLOATS = 2 * BK * (BM + BN);
  static constexpr int ST_STRIDE = WM + 4;          // epilogue staging row (floats), 16-B multiple
  static constexpr int ST_FLOATS = 4 * 16 * ST_STRIDE;
  static constexpr int SMEM_FLOATS = AB_FLOATS > ST_FLOATS ? AB_FLOATS : ST_FLOATS;
};

__device__ __forceinline__ float act_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
// d/dx of 0.5 x (1 + erf(x / sqrt 2)) = 0.5 (1 + erf(x / sqrt 2)) + x exp(-x^2 / 2) / sqrt(2 pi)
__device__ __forceinline__ float act_gelu_grad(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * expf(-0.5f * x * x) * 0.39894228040143267794f;
}
// activation codes (advhip_conv3d_desc::relu): 0 none, 1 ReLU, 2 GELU, 3 GELU with GELU'(pre-activation) as the second output (instead of
// the pre-activation itself), 4 none + the result multiplied by the `dact` tensor AS IS (a GELU' saved by a code-3 forward: the fused GELU
// backward without an erf / exp in the backward epilogue)
constexpr int ACT_GELU_D = 3, ACT_MUL = 4;
__device__ __forceinline__ float act_apply(int code, float v) { return code == 1 ? fmaxf(v, 0.f) : ((code == 2 || code == ACT_GELU_D) ? act_gelu(v) : v); }
// what the second output holds for pre-activation value v
__device__ __forceinline__ float act_second(int code, float v) { return code == ACT_GELU_D ? act_gelu_grad(v) : v; }
// the `dact` factor for a stored operand value
__device__ __forceinline__ float act_dact(int code, float op) { return code == ACT_MUL ? op : act_gelu_grad(op); }

// ---- the pipelined epilogue: y = act(acc * scale + shift (+ res)) (* GELU'(z)), act in {none, ReLU, GELU (+ the pre-activation as a
// second output)} -- every conv of the I3D plan and every MGFN GEMM without a LayerNorm fold.  Same arithmetic, LDS transposition
// and store pattern as igemm_epilogue below; what differs is the order of the memory operations: below, every staged row waits for
// its scale / shift loads, then for its residual (or z) load, then stores -- FN * 16 / RPI rounds of two dependent latencies per
// tile (8 x 2 for the 128 x 64 tile, 32 x 2 on the unaligned path), which a launch of one or two rounds of workgroups (every launch
// of I3D's layers 2-4, the MGFN GEMMs) cannot hide behind other workgroups' MFMAs because the workgroups of a round reach their
// epilogues together.  Here the wave's scale / shift rows are two loads (one channel per lane, handed out by ds_bpermute) and the
// residual (or z) pieces of the tile run through a window of PLAIN_EPI_WINDOW registers: that many loads in flight, piece k + W
// issued when piece k is used.  The operand kind is a template parameter, which keeps the body free of branches around loads, so
// the compiler's s_waitcnt counts stay exact (a runtime `if (a.res)` around a load makes every later wait a vmcnt(0)).
// Window size: 8 registers measured = 12 (3 826 / 3 828 clips/s); 16 costs the 6-wave kernels 12-44 bytes of scratch and 2.6 %.
#ifndef PLAIN_EPI_WINDOW
#define PLAIN_EPI_WINDOW 8
#endif
#ifndef SPLITK_SUM_BATCH
#define SPLITK_SUM_BATCH 4
#endif
// operand read per piece: none / residual / z of GELU'(z) / a multiplier; act: ReLU flag / GELU (second output: pre-activation) / GELU
// (second output: GELU'(pre-activation))
enum EpiMode { EM_PLAIN = 0, EM_RES = 1, EM_DACT = 2, EM_GELU = 3, EM_MUL = 4, EM_GELU2 = 5 };

template <int BM, int BN, int BK, int MODE>
__device__ __forceinline__ void igemm_epilogue_plain(const ConvArgs& a, f32x4 (&acc)[IgemmCfg<BM, BN, BK>::FM][IgemmCfg<BM, BN, BK>::FN], float* smem, int m0, int n0, int wave, int lane) {
  using Cfg = IgemmCfg<BM, BN, BK>;
  constexpr int FM = Cfg::FM, FN = Cfg::FN;
  constexpr bool OPERAND = MODE == EM_RES || MODE == EM_DACT || MODE == EM_MUL;
  constexpr bool SECOND = MODE == EM_GELU || MODE == EM_GELU2;
  const int wm = Cfg::wave_m(wave), wn = Cfg::wave_n(wave);
  const int li = lane & 15, lg = lane >> 4;
  float* st = smem + wave * (16 * Cfg::ST_STRIDE);
  const bool relu = a.relu == 1;
  const float* __restrict__ opnd = MODE == EM_RES ? a.res : a.dact;
  // channel n0 + wn*WN + c of this wave's tile: lane c holds its scale / shift (WN = 32: the upper lanes repeat the lower ones)
  const int cn = n0 + wn * Cfg::WN + (lane & (Cfg::WN - 1));
  const float scv = a.scale[cn], sfv = a.shift[cn];
  auto stage = [&](int jn) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if constexpr (FM == 4) {
        *reinterpret_cast<float4*>(&st[li * Cfg::ST_STRIDE + 16 * lg + 4 * r]) = make_float4(acc[0][jn][r], acc[1][jn][r], acc[2][jn][r], acc[3][jn][r]);
      } else {
        *reinterpret_cast<float2*>(&st[li * Cfg::ST_STRIDE + 8 * lg + 2 * r]) = make_float2(acc[0][jn][r], acc[1][jn][r]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto staged = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // one value through scale / shift, operand and activation (the expressions of igemm_epilogue, in its order)
  auto finish = [&](float v, float sc, float sf, float op, float& pre) __attribute__((always_inline)) {
    v = v * sc + sf;
    if constexpr (MODE == EM_RES) v += op;
    pre = v;
    if constexpr (MODE == EM_GELU) {
      v = act_gelu(v);
    } else if constexpr (MODE == EM_GELU2) {  // y = GELU(v), second output = GELU'(v): one erf for both
      const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752440f));
      pre = cdf + v * expf(-0.5f * v * v) * 0.39894228040143267794f;
      v = v * cdf;
    } else if constexpr (MODE != EM_MUL) {
      v = relu ? fmaxf(v, 0.f) : v;
    }
    if constexpr (MODE == EM_DACT) v *= act_gelu_grad(op);
    if constexpr (MODE == EM_MUL) v *= op;
    return v;
  };
  if (a.vw == 4) {
    constexpr int LPR = Cfg::WM / 4, RPI = 64 / LPR, ITS = 16 / RPI;
    const int rrow = lane / LPR, rcol = (lane % LPR) * 4;
    const int mm = m0 + wm * Cfg::WM + rcol;
    const bool mok = mm < a.M;  // M % 4 == 0 here: the group of 4 is all-in or all-out
    int bb = 0, pp = 0;
    if (mok) { bb = (int)a.dTHWo.div((unsigned)mm); pp = mm - bb * a.MP; }
    // this lane's piece (jn, i) = channel nbase + FN * (rrow + RPI * i) + jn, positions pp .. pp + 3 of sample bb: a per-lane pointer
    // plus a uniform multiple of THWo (lanes past M point at sample 0 and drop what they read)
    const int nrow = n0 + wn * Cfg::WN + FN * rrow;
    const float* rp0 = OPERAND ? opnd + ((size_t)bb * a.Cout + nrow) * a.THWo + pp : nullptr;
    const size_t yoff = (size_t)bb * a.y_bstride + (size_t)nrow * a.THWo + pp;
    float* yp0 = a.y + yoff;
    const unsigned thwo = (unsigned)a.THWo;
    // operand pieces k = jn * ITS + i through a window of W float4 registers
    constexpr int NP = FN * ITS, W = PLAIN_EPI_WINDOW / 4;
    float rv[OPERAND ? W : 1][4];
    auto issue = [&](int k) __attribute__((always_inline)) { vec_load<4>(rp0 + (size_t)((unsigned)(FN * RPI * (k % ITS) + k / ITS) * thwo), rv[k % W]); };
    if constexpr (OPERAND) {
#pragma unroll
      for (int k = 0; k < W; ++k) issue(k);
    }
#pragma unroll
    for (int jn = 0; jn < FN; ++jn) {
      stage(jn);
#pragma unroll
      for (int i = 0; i < ITS; ++i) {
        const int row = rrow + RPI * i, k = jn * ITS + i;
        float v[4], pre[4];
        vec_load<4>(&st[row * Cfg::ST_STRIDE + rcol], v);
        const float sc = __shfl(scv, FN * row + jn), sf = __shfl(sfv, FN * row + jn);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = finish(v[e], sc, sf, OPERAND ? rv[k % W][e] : 0.f, pre[e]);
        if constexpr (OPERAND) {
          if (k + W < NP) issue(k + W);
        }
        const size_t po = (size_t)((unsigned)(FN * RPI * i + jn) * thwo);
        if constexpr (SECOND) {
          if (a.y2 && mok) vec_store<4>(a.y2 + yoff + po, pre);
        }
        if (mok) vec_store<4>(yp0 + po, v);
      }
      staged();
    }
    return;
  }
  {
    constexpr int RPI = 64 / Cfg::WM, ITS = 16 / RPI;  // channel rows per read instruction (1 or 2)
    const int rrow = lane / Cfg::WM, rcol = lane % Cfg::WM;
    const int mm = m0 + wm * Cfg::WM + rcol;
    bool mok = mm < a.M;
    int bb = 0, pp = 0;
    if (mok) {
      bb = (int)a.dTHWo.div((unsigned)mm);
      pp = mm - bb * a.MP;
      mok = pp < a.THWo;  // (virtually padded rows: see ConvArgs::MP)
    }
    if (!mok) bb = pp = 0;  // (these lanes read sample 0 and drop it)
    const int nrow = n0 + wn * Cfg::WN + FN * rrow;
    const float* rp0 = OPERAND ? opnd + ((size_t)bb * a.Cout + nrow) * a.THWo + pp : nullptr;
    const size_t yoff = (size_t)bb * a.y_bstride + (size_t)nrow * a.THWo + pp;
    float* yp0 = a.y + yoff;
    const unsigned thwo = (unsigned)a.THWo;
    constexpr int NP = FN * ITS, W = PLAIN_EPI_WINDOW;  // (one position per lane: dword pieces)
    float rv[OPERAND ? W : 1];
    auto issue = [&](int k) __attribute__((always_inline)) { rv[k % W] = rp0[(size_t)((unsigned)(FN * RPI * (k % ITS) + k / ITS) * thwo)]; };
    if constexpr (OPERAND) {
#pragma unroll
      for (int k = 0; k < W; ++k) issue(k);
    }
#pragma unroll
    for (int jn = 0; jn < FN; ++jn) {
      stage(jn);
#pragma unroll
      for (int i = 0; i < ITS; ++i) {
        const int row = rrow + RPI * i, k = jn * ITS + i;
        float pre;
        float v = st[row * Cfg::ST_STRIDE + rcol];
        v = finish(v, __shfl(scv, FN * row + jn), __shfl(sfv, FN * row + jn), OPERAND ? rv[k % W] : 0.f, pre);
        if constexpr (OPERAND) {
          if (k + W < NP) issue(k + W);
        }
        const size_t po = (size_t)((unsigned)(FN * RPI * i + jn) * thwo);
        if constexpr (SECOND) {
          if (a.y2 && mok) a.y2[yoff + po] = pre;
        }
        if (mok) yp0[po] = v;
      }
      staged();
    }
  }
}

// ---- epilogue shared by both implicit-GEMM kernels -------------------------------------------
// accumulator element acc[jm][jn][r] of lane (li, lg):
//   m = m0 + wm*WM + FM*(4*lg + r) + jm,   n = n0 + wn*WN + FN*li + jn
template <int BM, int BN, int BK>
__device__ __forceinline__ void igemm_epilogue(const ConvArgs& a, f32x4 (&acc)[IgemmCfg<BM, BN, BK>::FM][IgemmCfg<BM, BN, BK>::FN], float* smem, int split,
                                               int m0, int n0, int wave, int lane, bool fused) {
  using Cfg = IgemmCfg<BM, BN, BK>;
  constexpr int FM = Cfg::FM, FN = Cfg::FN;
  const int wm = Cfg::wave_m(wave), wn = Cfg::wave_n(wave);
  const int li = lane & 15, lg = lane >> 4;
  const int b_col = wn * Cfg::WN + FN * li;
  if (fused && !a.ln_u) {  // (uniform: kernel arguments)
    // GELU and GELU' forms: the 128 x 128 tile only (the MGFN FFN GEMMs; 4 waves per SIMD there).  In the 6- and 8-wave kernels
    // their erf temporaries spill, and scratch in a kernel costs the whole plan more than these forms gain on small layers.
    if constexpr (BM * BN >= 128 * 128) {
      if (a.relu == 2 && !a.res && !a.dact) { igemm_epilogue_plain<BM, BN, BK, EM_GELU>(a, acc, smem, m0, n0, wave, lane); return; }
      if (a.relu == ACT_GELU_D && !a.res && !a.dact) { igemm_epilogue_plain<BM, BN, BK, EM_GELU2>(a, acc, smem, m0, n0, wave, lane); return; }
      if (a.relu <= 1 && !a.y2 && a.dact && !a.res) { igemm_epilogue_plain<BM, BN, BK, EM_DACT>(a, acc, smem, m0, n0, wave, lane); return; }
    }
    // (the multiplier form has no transcendental in it: every tile takes the pipelined epilogue)
    if (a.relu == ACT_MUL && a.dact && !a.res && !a.y2) { igemm_epilogue_plain<BM, BN, BK, EM_MUL>(a, acc, smem, m0, n0, wave, lane); return; }
    if (a.relu <= 1 && !a.y2 && !a.dact) {
      if (a.res) igemm_epilogue_plain<BM, BN, BK, EM_RES>(a, acc, smem, m0, n0, wave, lane);
      else igemm_epilogue_plain<BM, BN, BK, EM_PLAIN>(a, acc, smem, m0, n0, wave, lane);
      return;
    }
  }
  float* __restrict__ yout = fused ? a.y : a.y + (size_t)split * a.slab;
  if (a.vw == 4) {
    // Coalesced path (THWo % 4 == 0): each wave transposes its tile through LDS, one fragment
    // column (16 channels x WM positions) at a time, so that global stores / residual loads are
    // whole 16-byte pieces of contiguous NCDHW rows (WM*4 bytes per channel per wave).
    float* st = smem + wave * (16 * Cfg::ST_STRIDE);
    constexpr int LPR = Cfg::WM / 4;      // lanes per staged row (16 or 8)
    constexpr int RPI = 64 / LPR;         // rows per read instruction (4 or 8)
    const int rrow = lane / LPR, rcol = (lane % LPR) * 4;
    const int mm = m0 + wm * Cfg::WM + rcol;  // this lane's 4 consecutive m in the read phase
    const bool mok = mm < a.M;                // M % 4 == 0 here, so the group is all-in or all-out
    int bb = 0, pp = 0;
    if (mok) { bb = (int)a.dTHWo.div((unsigned)mm); pp = mm - bb * a.MP; }
#pragma unroll
    for (int jn = 0; jn < FN; ++jn) {
      // write phase: 4*FM consecutive m per lane for channel row li
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (FM == 4) {
          *reinterpret_cast<float4*>(&st[li * Cfg::ST_STRIDE + 16 * lg + 4 * r]) =
              make_float4(acc[0][jn][r], acc[1][jn][r], acc[2][jn][r], acc[3][jn][r]);
        } else {
          *reinterpret_cast<float2*>(&st[li * Cfg::ST_STRIDE + 8 * lg + 2 * r]) = make_float2(acc[0][jn][r], acc[1][jn][r]);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int i = 0; i < 16 / RPI; ++i) {
        const int row = rrow + RPI * i;  // channel row within the fragment column
        float v[4];
        vec_load<4>(&st[row * Cfg::ST_STRIDE + rcol], v);
        if (mok) {
          const int n = n0 + wn * Cfg::WN + FN * row + jn;
          const size_t o = (size_t)(bb * a.Cout + n) * a.THWo + pp;                                  // dense: residual, slabs
          const size_t oy = fused ? (size_t)bb * a.y_bstride + (size_t)n * a.THWo + pp : o;  // output proper
          if (fused) {
            const float sc = a.scale[n], sf = a.shift[n];
            if (a.ln_u) {  // positions mm .. mm+3 of the flattened (b, t, h, w) index
              float rs[4], mu[4];
              vec_load<4>(a.ln_rs + mm, rs);
              vec_load<4>(a.ln_mu + mm, mu);
              const float u = a.ln_u[n];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = v[e] * rs[e] - u * (mu[e] * rs[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * sc + sf;
            if (a.res) {
              float rv[4];
              vec_load<4>(a.res + o, rv);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += rv[e];
            }
            if (a.y2) {
              float sv[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) sv[e] = act_second(a.relu, v[e]);
              vec_store<4>(a.y2 + oy, sv);
            }
            if (a.relu) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = act_apply(a.relu, v[e]);
            }
            if (a.dact) {
              float zv[4];
              vec_load<4>(a.dact + o, zv);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= act_dact(a.relu, zv[e]);
            }
          }
          vec_store<4>(yout + oy, v);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    return;
  }
  // Unaligned path (THWo not a multiple of 4 -- e.g. 2 x 7 x 7 = 98 positions per sample in layer 4 -- or pointers / strides that
  // are not 16-byte aligned): the same LDS transposition, but the read phase gives every lane ONE position, so a
  // wave-instruction stores (and reads the residual of) 64 consecutive m of a channel as whole dwords: coalesced 256-byte
  // runs instead of 8-byte pieces scattered over 16 channels (which cost 3x the bytes at HBM: profiles/r02_*_traffic_by_layer.md).
  {
    float* st = smem + wave * (16 * Cfg::ST_STRIDE);
    constexpr int RPI = 64 / Cfg::WM;      // channel rows per read instruction (1 or 2)
    const int rrow = lane / Cfg::WM, rcol = lane % Cfg::WM;
    const int mm = m0 + wm * Cfg::WM + rcol;
    bool mok = mm < a.M;
    int bb = 0, pp = 0;
    if (mok) {
      bb = (int)a.dTHWo.div((unsigned)mm);
      pp = mm - bb * a.MP;
      mok = pp < a.THWo;  // (virtually padded rows: see ConvArgs::MP)
    }
#pragma unroll
    for (int jn = 0; jn < FN; ++jn) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (FM == 4) {
          *reinterpret_cast<float4*>(&st[li * Cfg::ST_STRIDE + 16 * lg + 4 * r]) =
              make_float4(acc[0][jn][r], acc[1][jn][r], acc[2][jn][r], acc[3][jn][r]);
        } else {
          *reinterpret_cast<float2*>(&st[li * Cfg::ST_STRIDE + 8 * lg + 2 * r]) = make_float2(acc[0][jn][r], acc[1][jn][r]);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int i = 0; i < 16 / RPI; ++i) {
        const int row = rrow + RPI * i;
        float v = st[row * Cfg::ST_STRIDE + rcol];
        if (mok) {
          const int n = n0 + wn * Cfg::WN + FN * row + jn;
          const size_t o = (size_t)(bb * a.Cout + n) * a.THWo + pp;
          const size_t oy = fused ? (size_t)bb * a.y_bstride + (size_t)n * a.THWo + pp : o;
          if (fused) {
            if (a.ln_u) v = v * a.ln_rs[mm] - a.ln_u[n] * (a.ln_mu[mm] * a.ln_rs[mm]);
            v = v * a.scale[n] + a.shift[n];
            if (a.res) v += a.res[o];
            if (a.y2) a.y2[oy] = act_second(a.relu, v);
            if (a.relu) v = act_apply(a.relu, v);
            if (a.dact) v *= act_dact(a.relu, a.dact[o]);
          }
          yout[oy] = v;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
}

template <int BM, int BN, int BK>
__global__ __launch_bounds__(256) void conv3d_igemm_f32_kernel(const ConvArgs a) {
  using Cfg = IgemmCfg<BM, BN, BK>;
  constexpr int FM = Cfg::FM, FN = Cfg::FN, KR = Cfg::KR, RA = Cfg::RA, RB = Cfg::RB;
  static_assert(FM == 4 || FM == 2, "wave M tile must be 64 or 32");
  static_assert(FN == 4 || FN == 2, "wave N tile must be 64 or 32");
  static_assert(RA >= 1 && RB >= 1, "tile too small for 256 threads");

  __shared__ __attribute__((aligned(16))) float smem[Cfg::SMEM_FLOATS];
  float(*As)[BK][BM] = reinterpret_cast<float(*)[BK][BM]>(smem);
  float(*Bs)[BK][BN] = reinterpret_cast<float(*)[BK][BN]>(smem + 2 * BK * BM);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = a.tiles_m * a.tiles_n;
  int L, split;
  if (a.splits > 1) {
    // blocks b and b+8 share an XCD: keep one K-slice of the weights per XCD's L2
    split = blockIdx.x % a.splits;
    L = blockIdx.x / a.splits;
  } else {
    split = 0;
    L = xcd_remap(blockIdx.x, ntiles);
  }
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
  const int mbase = b * a.x_bstride + it0 * a.HW + ih0 * a.W + iw0;

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
  const int a_col = wm * Cfg::WM + FM * li;
  const int b_col = wn * Cfg::WN + FN * li;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // k-tiles of this split: [kt0, kt1)
  const int nk_all = a.Kpad / BK;
  const int kt0 = (nk_all * split) / a.splits;
  const int kt1 = (nk_all * (split + 1)) / a.splits;
  if (kt0 < kt1) {
    load_tiles(kt0 * BK);
    store_tiles(0);
  }
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    if (kt + 1 < kt1) load_tiles((kt + 1) * BK);
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
    if (kt + 1 < kt1) store_tiles(cur ^ 1);
    __syncthreads();
  }

  igemm_epilogue<BM, BN, BK>(a, acc, smem, split, m0, n0, wave, lane, a.splits == 1);
}

// ================================================================================================
// Fast-path implicit GEMM: same tiling / MFMA loop / epilogue as conv3d_igemm_f32_kernel, with the
// per-element gather arithmetic hoisted out of the vector pipe.  The generic kernel spends ~14 VALU
// instructions per gathered element (three range checks + 64-bit address), i.e. ~4.3 VALU per MFMA
// on 64-wide N tiles (rocprofv3 PMC, profiles/r01_pmc_notes.md) -- more than an fp32 MFMA can
// shadow.  Here every A element is one raw buffer load whose
//   * per-lane voffset  = this thread's window origin (constant for the whole K loop),
//   * scalar  soffset   = tap offset of row k from the gather table (SMEM/SALU),
//   * validity          = one masked compare against a per-thread coordinate mask built once per
//                         block (CHECK): a tap is in range iff its dt, dh and dw each are,
//                         realised as an out-of-range voffset -> the buffer unit returns 0;
//                         convs whose every tap is in range for every m (1x1x1, no padding) skip
//                         even that (CHECK = false): 0 VALU per element.
// Requires: kernel extents <= 10 per axis, x < 3.75 GiB (32-bit buffer offsets).
template <int BM, int BN, int BK, bool CHECK>
__global__ __launch_bounds__(256) void conv3d_igemm_fast_kernel(const ConvArgs a) {
  using Cfg = IgemmCfg<BM, BN, BK>;
  constexpr int FM = Cfg::FM, FN = Cfg::FN, KR = Cfg::KR, RA = Cfg::RA, RB = Cfg::RB;
  constexpr unsigned OOB = 0xFFFFFF00u;

  __shared__ __attribute__((aligned(16))) float smem[Cfg::SMEM_FLOATS];
  float(*As)[BK][BM] = reinterpret_cast<float(*)[BK][BM]>(smem);
  float(*Bs)[BK][BN] = reinterpret_cast<float(*)[BK][BN]>(smem + 2 * BK * BM);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = a.tiles_m * a.tiles_n;
  int L, split;
  if (a.splits > 1) {
    L = (int)a.dSplits.div(blockIdx.x);
    split = (int)blockIdx.x - L * a.splits;
  } else {
    split = 0;
    L = xcd_remap(blockIdx.x, ntiles);
  }
  const int tile_m = (int)a.dTilesN.div((unsigned)L), tile_n = L - tile_m * a.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- this thread's A column: window origin (voffset) and tap-validity mask ------------------
  const int ml = tid % BM;
  const int kr = __builtin_amdgcn_readfirstlane(tid / BM);
  const int m = m0 + ml;
  unsigned vbase = OOB;
  unsigned vmask = 0;
  if (m < a.M) {
    const int b = (int)a.dTHWo.div((unsigned)m);
    const int p = m - b * a.THWo;
    const int ot = (int)a.dHWo.div((unsigned)p);
    const int q = p - ot * a.HWo;
    const int oh = (int)a.dWo.div((unsigned)q);
    const int ow = q - oh * a.Wo;
    const int it0 = ot * a.st - a.pt, ih0 = oh * a.sh - a.ph, iw0 = ow * a.sw - a.pw;
    vbase = (unsigned)(b * a.x_bstride + it0 * a.HW + ih0 * a.W + iw0 + a.pad_off) * 4u;
    if constexpr (CHECK) {
      // a tap (dt,dh,dw) is inside the input iff each coordinate is: keep one bit per coordinate
      // value (bits 0-9: dt, 10-19: dh, 20-29: dw); the table holds the three bits a row needs
      vmask = tap_bits(it0, a.kt_, a.T) | (tap_bits(ih0, a.kh_, a.H) << 10) | (tap_bits(iw0, a.kw_, a.W) << 20);
    }
  }
  const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) - a.pad_off, 0, a.x_bytes, 0x00020000);

  const int2* __restrict__ ktab2 = reinterpret_cast<const int2*>(a.ktab + a.Kpad);
  float ra[RA];
  float rb[RB][4];
  // One k-step's share of the next tile's loads (issued between the MFMA groups of the current
  // tile so their scalar loads / mask tests / VMEM issue sit in the MFMAs' shadow instead of in
  // front of them).
  constexpr int KS = BK / 4;                       // k-steps per tile
  // the table entries of a tile are fetched (scalar loads) during k-step 0 and consumed by the
  // gathers of k-steps 1..KS-1, so no MFMA group waits on SMEM latency
  constexpr int A_PER_KS = (RA + KS - 2) / (KS - 1);
  constexpr int B_PER_KS = (RB + KS - 1) / KS;
  int2 ent[RA];
  auto load_entries = [&](int k0) {
#pragma unroll
    for (int j = 0; j < RA; ++j) ent[j] = ktab2[k0 + kr + KR * j];  // wave-uniform -> scalar loads
  };
  auto load_a = [&](int j) {
    const int2 e = ent[j];  // {byte offset, coordinate bits}
    unsigned voff = vbase;
    if constexpr (CHECK) voff = ((vmask & (unsigned)e.y) == (unsigned)e.y) ? vbase : OOB;
    ra[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, voff, e.x, 0));
  };
  auto load_b = [&](int k0, int j) {
    const int idx = tid + 256 * j;
    const int row = idx / (BN / 4), c4 = idx % (BN / 4);
    const float4 t = *reinterpret_cast<const float4*>(a.w + (size_t)(k0 + row) * a.Cout + n0 + c4 * 4);
    rb[j][0] = t.x; rb[j][1] = t.y; rb[j][2] = t.z; rb[j][3] = t.w;
  };
  auto load_tiles = [&](int k0) {
    load_entries(k0);
#pragma unroll
    for (int j = 0; j < RA; ++j) load_a(j);
#pragma unroll
    for (int j = 0; j < RB; ++j) load_b(k0, j);
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

  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  const int a_col = wm * Cfg::WM + FM * li;
  const int b_col = wn * Cfg::WN + FN * li;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk_all = a.Kpad / BK;
  const int kt0 = (nk_all * split) / a.splits;
  const int kt1 = (nk_all * (split + 1)) / a.splits;
  if (kt0 < kt1) {
    load_tiles(kt0 * BK);
    store_tiles(0);
  }
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    // straight-line body (one scheduling region): the last iteration re-loads its own tile into the
    // idle buffer instead of branching around the prefetch
    const int knext = (kt + 1 < kt1 ? kt + 1 : kt) * BK;
    float av[2][FM], bv[2][FN];
    auto read_frags = [&](int ks, float (&ao)[FM], float (&bo)[FN]) {
      if constexpr (FM == 4) {
        const float4 t = *reinterpret_cast<const float4*>(&As[cur][4 * ks + lg][a_col]);
        ao[0] = t.x; ao[1] = t.y; ao[2] = t.z; ao[3] = t.w;
      } else {
        const float2 t = *reinterpret_cast<const float2*>(&As[cur][4 * ks + lg][a_col]);
        ao[0] = t.x; ao[1] = t.y;
      }
      if constexpr (FN == 4) {
        const float4 t = *reinterpret_cast<const float4*>(&Bs[cur][4 * ks + lg][b_col]);
        bo[0] = t.x; bo[1] = t.y; bo[2] = t.z; bo[3] = t.w;
      } else {
        const float2 t = *reinterpret_cast<const float2*>(&Bs[cur][4 * ks + lg][b_col]);
        bo[0] = t.x; bo[1] = t.y;
      }
    };
    read_frags(0, av[0], bv[0]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      // operands of the next k-step are requested before this step's MFMAs (LDS latency hidden)
      if (ks + 1 < KS) read_frags(ks + 1, av[(ks + 1) & 1], bv[(ks + 1) & 1]);
      if (ks == 0) load_entries(knext);
#pragma unroll
      for (int j = 0; j < A_PER_KS; ++j)
        if (ks >= 1 && (ks - 1) * A_PER_KS + j < RA) load_a((ks - 1) * A_PER_KS + j);
#pragma unroll
      for (int j = 0; j < B_PER_KS; ++j)
        if (ks * B_PER_KS + j < RB) load_b(knext, ks * B_PER_KS + j);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks & 1][i], bv[ks & 1][j], acc[i][j], 0, 0, 0);
      // keep each k-step's share of the prefetch next to its MFMAs (hipcc otherwise sinks all the
      // loads to the end of the tile, right in front of the ds_writes that wait for them)
      __builtin_amdgcn_sched_barrier(0);
    }
    store_tiles(cur ^ 1);
    __syncthreads();
  }
  igemm_epilogue<BM, BN, BK>(a, acc, smem, split, m0, n0, wave, lane, a.splits == 1);
}

// Epilogue forms of the LDS-DMA kernel.  The pooling forms fuse the nn.MaxPool3d that follows the conv in
// I3Res50.forward_single (src/i3d.py:303-309) into the conv launch; they order m in bricks (see brick_* below).
enum : int {
  EPI_STD = 0,      // y = act(conv * scale + shift (+ res)), NCDHW
  EPI_TPOOL = 2,    // + MaxPool3d k(2,1,1) s(2,1,1): brick 2(t) x BM/2 flattened (h,w) of a 1x1x1 conv; exact, no halo
  EPI_POOL233 = 3,  // + ReLU + MaxPool3d k(2,3,3) s(2,2,2) p0: brick 2(t) x 4(h) x BM/8(w); per-brick maxima of every pooling
                    //   window the brick touches go to a partial tensor, stem_pool_merge_kernel maxes the 1/2/4 partials
  EPI_TSPAN2 = 4,   // plain output, m-tiles that span T: brick 2(t) x BM/2 flattened (h,w) of a (kt,1,1) conv -- the kt temporal taps
  EPI_TSPAN4 = 5,   //   of a tile read the SAME activation rows (shifted by one plane), so they hit L1/L2 instead of being fetched by
                    //   three m-tiles that run ~47 tiles apart; 4(t) x BM/4 for T = 4
  EPI_AVG = 6,      // + AdaptiveAvgPool3d((1,1,1)) (src/i3d.py:314): a 1x1x1 conv on <= 128 positions per sample, every sample's rows
                    //   padded to ONE 128-row m-tile in the m index space (ConvArgs::MP = 128): the tile's column means are the result
};
constexpr bool epi_rows(int epi) { return epi == EPI_STD || epi == EPI_AVG; }  // m runs (sample, position) row-major (no bricks)
constexpr int brick_t(int epi) { return epi == EPI_TSPAN4 ? 4 : 2; }

// ---- pooling epilogues on brick-ordered tiles (128 x 64 tile, 2 x 2 waves) ----------------------------------------------
// Wave (wm, wn) holds t plane wm of the brick (64 positions: BH rows of BW outputs) for 32 channels; accumulator element
// acc[i][jn][r] of lane (li, lg) sits at position 16*lg + 4*r + i of the plane, channel n0 + 2*(16*wn + li) + jn.  One
// fragment column jn at a time, all four waves lay their values out in LDS as [32 channels][2 planes x 64 positions]
// (scale/shift applied; the stem form also applies ReLU and zeroes positions outside the tensor -- every real value is
// >= 0 after ReLU, so 0 is neutral for the maxima), then the 256 threads pool from LDS with coalesced global accesses.
constexpr int POOL_SLOTS = 27;  // per brick and channel: 3 row slots x 9 column slots (see brick_epilogue, EPI_POOL233)

// uint8 stem: the border-class correction of a lane's brick row, fetched before the main loop (so that no table read sits
// in the epilogue of a tile): c[jn] when the 16 outputs of the row share one w class, else read per output from `row`
struct U8Corr {
  float c[2];
  const float* row;  // a.pad_corr + (class_t, class_h, 0, n_w(jn = 0))
  int uniform;
};

template <int BM, int BN, int BK, int EPI, bool U8 = false>
__device__ __forceinline__ void brick_epilogue(const ConvArgs& a, f32x4 (&acc)[BM / 32][BN / 32], float* smem, int tile_m,
                                               int tile_n, int n0, int bk_b, int bk_t, int bk_h, int bk_w, int wave, int lane,
                                               int tid, const U8Corr& u8c = U8Corr{}) {
  constexpr int FM = BM / 32, FN = BN / 32, RS = BM + 4, CH = BN / FN;  // CH channels per pass
  static_assert(FM == 4 && FN == 2, "128 x 64 tile");
  constexpr int BT = brick_t(EPI), BH = EPI == EPI_POOL233 ? 4 : 1, BW = BM / (BT * BH);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  const int c_w = wn * 16 + li;  // this lane's channel row of the pass
  if constexpr (EPI == EPI_TSPAN2 || EPI == EPI_TSPAN4) {
    // un-pooled output of a brick-ordered tile: thread = one position (dt, p) of the brick, 16 channels of the pass
#pragma unroll
    for (int jn = 0; jn < FN; ++jn) {
      const int n_w = n0 + 2 * c_w + jn;
      const float sc = a.scale[n_w], sf = a.shift[n_w];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float4*>(&smem[c_w * RS + wm * 64 + 16 * lg + 4 * r]) =
            make_float4(acc[0][jn][r] * sc + sf, acc[1][jn][r] * sc + sf, acc[2][jn][r] * sc + sf, acc[3][jn][r] * sc + sf);
      __syncthreads();
      // 256 threads = 2 channel halves x 128 positions: consecutive lanes run along the BW positions of one t plane
      const int pos = tid % BM, ch0 = tid / BM;           // ch0 in {0, 1}
      const int dt = pos / BW, pw = pos % BW;
      const int ot = bk_t * BT + dt, ow = bk_w * BW + pw;
      if (ot < a.To && ow < a.Wo) {
#pragma unroll
        for (int i = 0; i < CH / 2; ++i) {
          const int c = ch0 + 2 * i;
          const int n = n0 + 2 * c + jn;
          float v = smem[c * RS + pos];
          const size_t o = ((size_t)bk_b * a.Cout + n) * a.THWo + (size_t)ot * a.HWo + ow;
          if (a.res) v += a.res[o];
          if (a.relu) v = act_apply(a.relu, v);
          a.y[(size_t)bk_b * a.y_bstride + (size_t)n * a.THWo + (size_t)ot * a.HWo + ow] = v;
        }
      }
      __syncthreads();
    }
  } else if constexpr (EPI == EPI_TPOOL) {
#pragma unroll
    for (int jn = 0; jn < FN; ++jn) {
      const int n_w = n0 + 2 * c_w + jn;
      const float sc = a.scale[n_w], sf = a.shift[n_w];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float4*>(&smem[c_w * RS + wm * 64 + 16 * lg + 4 * r]) =
            make_float4(acc[0][jn][r] * sc + sf, acc[1][jn][r] * sc + sf, acc[2][jn][r] * sc + sf, acc[3][jn][r] * sc + sf);
      __syncthreads();
      // y[b, n, tp, p] = max_t relu(conv[b, n, 2 tp + t, p] (+ res)): thread = one position p of the brick, 8 channels
      const int pw = tid % BW, c0 = tid / BW;  // BW = 64 positions, 4 channel groups
      const int ow = bk_w * BW + pw;
      if (ow < a.Wo) {
        // (the 16 residual values of the pass in flight together: one load latency per pass instead of one per channel)
        float r0[CH / 4], r1[CH / 4];
#pragma unroll
        for (int i = 0; i < CH / 4; ++i) r0[i] = r1[i] = 0.f;
        if (a.res) {
          const float* rp = a.res + ((size_t)bk_b * a.Cout + n0 + 2 * c0 + jn) * a.THWo + (size_t)(bk_t * 2) * a.HWo + ow;
#pragma unroll
          for (int i = 0; i < CH / 4; ++i) {
            r0[i] = rp[(size_t)(8 * i) * a.THWo];
            r1[i] = rp[(size_t)(8 * i) * a.THWo + a.HWo];
          }
        }
#pragma unroll
        for (int i = 0; i < CH / 4; ++i) {
          const int c = c0 + 4 * i;
          const int n = n0 + 2 * c + jn;
          float v0 = smem[c * RS + pw] + r0[i], v1 = smem[c * RS + 64 + pw] + r1[i];
          if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
          a.y[(size_t)bk_b * a.y_bstride + ((size_t)n * a.Tp + bk_t) * a.HWo + ow] = fmaxf(v0, v1);
        }
      }
      __syncthreads();
    }
  } else {
    // Every pooling window (2 t) x (rows 2hp..2hp+2) x (cols 2wp..2wp+2) that meets the brick gets the maximum over the
    // part of it inside the brick.  Brick columns 16j..16j+15: column slot 0 = window wp=8j-1 (column 0 only), slots
    // 1..7 = wp=8j..8j+6 (complete), slot 8 = wp=8j+7 (columns 14-15).  Brick rows 4k..4k+3: row slot 0 = hp=2k-1 (row 0),
    // slot 1 = hp=2k (rows 0-2, complete), slot 2 = hp=2k+1 (rows 2-3).
    //   1. a lane holds one whole brick row (16 outputs: w = 4r + i) of one channel and one t plane in registers: BN,
    //      ReLU and the nine column-slot maxima happen there (15 max instructions);
    //   2. the 9 values go to LDS as L[channel][t][h][9 (row pitch 12)], channel pitch 100 floats (conflict-free b128 writes);
    //   3. thread (channel, column slot) maxes over t and the rows of each row slot (8 LDS reads) and stores the three
    //      row-slot values; the partial tensor is [brick][n-tile][jn][row slot][channel][column slot]: 288 consecutive
    //      floats per row slot and pass, i.e. coalesced stores.
    constexpr int LR = 12, LC = 100;
    const int ot = bk_t * 2 + wm, oh = bk_h * BH + lg;
    // bricks entirely inside the tensor (all of them at 16 x 224 x 224) skip the per-element range checks
    const bool inside = bk_t * 2 + 1 < a.To && bk_h * BH + BH - 1 < a.Ho && bk_w * BW + BW - 1 < a.Wo;
    const int wlim = a.Wo - bk_w * BW;  // columns of the brick inside the tensor
    const bool row_ok = ot < a.To && oh < a.Ho;
#pragma unroll
    for (int jn = 0; jn < FN; ++jn) {
      const int n_w = n0 + 2 * c_w + jn;
      float sc = a.scale[n_w];
      const float sf = a.shift[n_w];
      float v[16];
      if constexpr (U8) {
        // the operand was the pixel byte: sum w (pixel - mean) = acc - mean * (sum of this channel's weights over the taps
        // inside the clip), tabulated per border class (outputs with the same set of taps inside); 1 / std belongs to the scale
        sc = sc / a.in_std;
        const float* cr = u8c.row + jn;
        const int ow0 = bk_w * BW;
        if (u8c.uniform) {
          const float c0 = u8c.c[jn];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) v[4 * r + i] = fmaxf((acc[i][jn][r] + c0) * sc + sf, 0.f);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int wc = u8_border_class((ow0 + 4 * r + i) * a.sw - a.pw, a.kw_, a.W, a.pw);  // (outputs past Wo: any class, zeroed below)
              v[4 * r + i] = fmaxf((acc[i][jn][r] + cr[(size_t)wc * a.Cout]) * sc + sf, 0.f);
            }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < 4; ++i) v[4 * r + i] = fmaxf(acc[i][jn][r] * sc + sf, 0.f);
      }
      if (!inside) {  // positions outside the tensor count as 0: neutral for maxima of post-ReLU values
#pragma unroll
        for (int w = 0; w < 16; ++w) v[w] = (row_ok && w < wlim) ? v[w] : 0.f;
      }
      float pr[8], cs[9];
#pragma unroll
      for (int j = 0; j < 8; ++j) pr[j] = fmaxf(v[2 * j], v[2 * j + 1]);
      cs[0] = v[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) cs[k] = fmaxf(pr[k - 1], v[2 * k]);
      cs[8] = pr[7];
      float* lrow = smem + c_w * LC + (wm * BH + lg) * LR;
      *reinterpret_cast<float4*>(lrow) = make_float4(cs[0], cs[1], cs[2], cs[3]);
      *reinterpret_cast<float4*>(lrow + 4) = make_float4(cs[4], cs[5], cs[6], cs[7]);
      lrow[8] = cs[8];
      __syncthreads();
      float* __restrict__ P = a.y + ((size_t)(tile_m * a.tiles_n + tile_n) * FN + jn) * (CH * POOL_SLOTS);
#pragma unroll
      for (int q0 = 0; q0 < CH * 9; q0 += 256) {
        const int q = q0 + tid;
        if (q < CH * 9) {
          const int c = q / 9, k = q - c * 9;
          const float* l0 = smem + c * LC + k;
          float h[4];
#pragma unroll
          for (int hh = 0; hh < 4; ++hh) h[hh] = fmaxf(l0[hh * LR], l0[(BH + hh) * LR]);  // max over the two t planes
          P[q] = h[0];
          P[CH * 9 + q] = fmaxf(fmaxf(h[0], h[1]), h[2]);
          P[2 * CH * 9 + q] = fmaxf(h[2], h[3]);
        }
      }
      __syncthreads();
    }
  }
}

// Pooled stem output from the per-brick partial maxima: y[b, n, tp, hp, wp] = max over the 1, 2 or 4 bricks the window meets.
// One workgroup per (sample, n-tile, channel parity jn, tp, hp): it copies the [32 channels][9 column slots] blocks of the
// row slot(s) that make up pooled row hp from every brick of that brick row into LDS (1152 contiguous bytes each: coalesced),
// then writes the 32 channels' rows of Wp outputs (coalesced) -- every partial is read exactly once.
constexpr int MERGE_MAX_NBW = 16;
__global__ __launch_bounds__(256) void stem_pool_merge_kernel(const float* __restrict__ P, float* __restrict__ y, int Cout, int Tp,
                                                              int Hp, int Wp, int nbh, int nbw, int tiles_n, FastDiv dWp,
                                                              long long rows, long long ybs) {
  extern __shared__ __attribute__((aligned(16))) float L[];  // [2][nbw][288]
  const long long row = (long long)blockIdx.y * gridDim.x + blockIdx.x;  // ((((b * tiles_n + tile_n) * 2 + jn) * Tp + tp) * Hp + hp
  if (row >= rows) return;
  const int hp = (int)(row % Hp);
  long long r = row / Hp;
  const int tp = (int)(r % Tp);
  r /= Tp;
  const int jn = (int)(r & 1);
  r >>= 1;
  const int tile_n = (int)(r % tiles_n), b = (int)(r / tiles_n);
  const int tid = (int)threadIdx.x;
  const bool h2 = hp & 1;
  const int nu = h2 ? 2 : 1;
  const long long brick_row0 = (((long long)b * Tp + tp) * nbh + (hp >> 1)) * nbw;
  // 72 float4 per (row-slot, brick) block of 288 contiguous floats
  for (int e = tid; e < nu * nbw * 72; e += 256) {
    const int g = e / 72, j = e - g * 72;
    const int u = g >= nbw ? 1 : 0, wb = g - u * nbw;
    const int rs = h2 ? (u == 0 ? 2 : 0) : 1;
    const long long brick = brick_row0 + (long long)u * nbw + wb;
    *reinterpret_cast<float4*>(L + g * 288 + j * 4) =
        *reinterpret_cast<const float4*>(P + ((brick * tiles_n + tile_n) * 2 + jn) * (32 * POOL_SLOTS) + rs * 288 + j * 4);
  }
  __syncthreads();
  for (int o = tid; o < 32 * Wp; o += 256) {
    const int c = (int)dWp.div((unsigned)o), wp = o - c * Wp;
    const int wb = wp >> 3;
    const bool w2 = (wp & 7) == 7;
    const int ws = w2 ? 8 : 1 + (wp & 7);
    const float* l = L + wb * 288 + c * 9;
    float m = l[ws];
    if (w2) m = fmaxf(m, l[288]);
    if (h2) {
      m = fmaxf(m, l[nbw * 288 + ws]);
      if (w2) m = fmaxf(m, l[nbw * 288 + 288]);
    }
    const int n = tile_n * 64 + 2 * c + jn;
    y[(long long)b * ybs + (((long long)n * Tp + tp) * Hp + hp) * Wp + wp] = m;
  }
}

// ================================================================================================
// LDS-DMA variant of the fast kernel: operand tiles go HBM/L2 -> LDS directly
// (`buffer_load_dword ... lds` for the gathered A rows, `buffer_load_dwordx4 ... lds` for the packed
// weight rows) into a 3-deep ring, so there are no staging VGPRs, no ds_write, and two k-tiles stay
// in flight across the single barrier per k-tile (counted vmcnt).
template <int BM, int BN, int BK>
struct DmaCfg {
  static constexpr int KR = 256 / BM;
  static constexpr int LA = BK / KR;                 // A LDS-DMA instructions per wave per k-tile
  static constexpr int LB = BK * BN * 4 / 1024 / 4;  // B (16-byte) LDS-DMA instructions per wave per k-tile
  static constexpr int STAGE = BK * (BM + BN);       // floats per ring stage
};

using lds_ptr_t = __attribute__((address_space(3))) void*;
using i32x8 = __attribute__((ext_vector_type(8))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;

// N consecutive {byte offset, coordinate bits} table entries through the scalar cache.  Inline asm
// (load + wait in one statement) because hipcc will not use SMEM for a table it cannot prove
// unclobbered next to LDS-DMA "stores" and falls back to vector loads + waterfall loops.
template <int N>
__device__ __forceinline__ void sload_entries(const int2* base, int byte_off, int (&out)[2 * N]) {
  static_assert(N == 4 || N == 8 || N == 16, "entries per wave per k-tile");
  if constexpr (N == 4) {
    i32x8 v;
    asm volatile("s_load_dwordx8 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(base), "s"(byte_off) : "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = v[i];
  } else if constexpr (N == 8) {
    i32x16 v;
    asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(base), "s"(byte_off) : "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = v[i];
  } else {
    i32x16 v, w;
    asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(v), "=&s"(w) : "s"(base), "s"(byte_off), "s"(byte_off + 64) : "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) { out[i] = v[i]; out[16 + i] = w[i]; }
  }
}

// LDS fragment reads as inline asm: hipcc tracks pending LDS-DMA writes per LDS object only for a
// handful of DMA instructions and otherwise puts `s_waitcnt vmcnt(0)` in front of every ds_read that
// follows an LDS-DMA, which would drain the ring every k-tile.  Reads issued from asm are invisible
// to that pass; their completion is waited for by hand (lds_wait names the destinations so that
// neither the MFMAs nor a register copy can be scheduled above the wait).
template <int NF>
struct Frag;  // NF fp32 operands of one k-step (4 -> ds_read_b128, 2 -> ds_read_b64)
template <>
struct Frag<4> { f32x4 v; };
template <>
struct Frag<2> { __attribute__((ext_vector_type(2))) float v; };

template <int NF, int OFF>
__device__ __forceinline__ void lds_read(Frag<NF>& f, unsigned addr) {
  if constexpr (NF == 4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.v) : "v"(addr), "n"(OFF));
  else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f.v) : "v"(addr), "n"(OFF));
}
template <int CNT, int NA, int NB>
__device__ __forceinline__ void lds_wait(Frag<NA>& a, Frag<NB>& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a.v), "+v"(b.v) : "n"(CNT));
}

// Waves per SIMD the register allocator must leave room for = the residency LDS allows.  The
// matrix pipe is only kept busy while other workgroups' MFMA loops cover a workgroup's prologue and
// epilogue (in-kernel stamps: ~12 of 66 us per tile at K = 768), so one more resident workgroup per CU
// is worth more than spare registers: without the attribute hipcc allocates 51 VGPR + 36 AGPR for the
// 64x64x16 kernel (5 waves/SIMD) although LDS admits 6.
#ifndef ADVHIP_T256_WAVES
#define ADVHIP_T256_WAVES 2
#endif
template <int BM, int BN, int BK, int NS>
constexpr int dma_waves_per_simd() {
  constexpr int ring = NS * BK * (BM + BN), st = IgemmCfg<BM, BN, BK>::ST_FLOATS;
  constexpr int lds_bytes = (ring > st ? ring : st) * 4;
  constexpr int by_lds = 163840 / lds_bytes;  // 256-thread workgroups = one wave per SIMD each
  if (BM == 256) return by_lds < ADVHIP_T256_WAVES ? by_lds : ADVHIP_T256_WAVES;  // (study knob; the 40-KB ring admits 4)
  if (BM * BN >= 128 * 128) return by_lds < 2 ? by_lds : 2;  // 64 accumulators: 3 waves would spill
  return by_lds > 8 ? 8 : (by_lds < 1 ? 1 : by_lds);
}

// AMODE 1: the A operand of a conv with stride 2 and an odd kernel along w, gathered as 16-byte pieces from COLUMN-PARITY
// PLANES of the input (xs[b, c, t, h, par, 2 + j] = x[b, c, t, h, 2 j + par], zero columns either side: split_w_kernel).
// Tap dw of output column ow reads input column 2 ow + dw - pw = plane (dw - pw) & 1, column ow + floor((dw - pw) / 2): for a
// fixed tap four consecutive output columns read four CONSECUTIVE floats, so one lane fetches the 16 bytes of four positions
// and one wave-instruction fills two whole k-rows of the [k][128 m] tile -- 2 A instructions per wave and k-tile instead of
// 8.  The stem is bound by the issue slots its LDS-DMA instructions share with the MFMAs (82 % MFMA-busy); same K order,
// same operands, same accumulation: bit-identical to the 4-byte gather.
// ---- EPI_AVG: the mean over a sample's positions instead of the output tensor ----------------------------------------------------
// Tile = one sample (rows >= THWo are padding and count as zero) x 64 channels.  Per fragment column the waves transpose through
// LDS as igemm_epilogue's unaligned path does (lane = position: coalesced residual reads), apply scale / shift / residual /
// activation, and add the 64 positions of their chunk with the butterfly below; chunk 0 + chunk 1, divided by THWo, is the mean.
// advhip_global_avgpool_f32 adds in the SAME order (pool.hip), so the fused launch equals conv + pool bit for bit.
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int BM, int BN, int BK, bool HAS_RES>
__device__ __forceinline__ void avg_epilogue_body(const ConvArgs& a, f32x4 (&acc)[BM / 32][BN / 32], float* smem, int b, int n0, int wave, int lane, int tid) {
  using Cfg = IgemmCfg<BM, BN, BK>;
  static_assert(BM == 128 && BN == 64, "one sample per 128-row tile, 2 x 2 waves");
  constexpr int FN = Cfg::FN;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  float* st = smem + wave * (16 * Cfg::ST_STRIDE);
  float* part = smem + Cfg::ST_FLOATS;  // [2 chunks of 64 positions][64 channels]
  const int pp = wm * 64 + lane;        // this lane's position in the read phase
  const bool mok = pp < a.THWo;
  const bool relu = a.relu != 0;
  // (as igemm_epilogue_plain: the wave's 32 scale / shift values one per lane, the 32 residual values of a lane through a window of
  // PLAIN_EPI_WINDOW registers -- the 32 rows below used to wait for a scale / shift load and then a residual load each)
  const int cn = n0 + wn * Cfg::WN + (lane & (Cfg::WN - 1));
  const float scv = a.scale[cn], sfv = a.shift[cn];
  const float* rp0 = HAS_RES ? a.res + ((size_t)b * a.Cout + n0 + wn * Cfg::WN) * a.THWo + (mok ? pp : 0) : nullptr;
  const unsigned thwo = (unsigned)a.THWo;
  constexpr int NP = FN * 16, W = PLAIN_EPI_WINDOW;
  float rv[HAS_RES ? W : 1];
  auto issue = [&](int k) __attribute__((always_inline)) { rv[k % W] = rp0[(size_t)((unsigned)(FN * (k % 16) + k / 16) * thwo)]; };
  if constexpr (HAS_RES) {
#pragma unroll
    for (int k = 0; k < W; ++k) issue(k);
  }
#pragma unroll
  for (int jn = 0; jn < FN; ++jn) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<float4*>(&st[li * Cfg::ST_STRIDE + 16 * lg + 4 * r]) = make_float4(acc[0][jn][r], acc[1][jn][r], acc[2][jn][r], acc[3][jn][r]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int row = 0; row < 16; ++row) {
      const int c = wn * Cfg::WN + FN * row + jn, k = jn * 16 + row;
      float v = st[row * Cfg::ST_STRIDE + lane] * __shfl(scv, FN * row + jn) + __shfl(sfv, FN * row + jn);
      if constexpr (HAS_RES) {
        v += rv[k % W];
        if (k + W < NP) issue(k + W);
      }
      v = relu ? fmaxf(v, 0.f) : v;
      v = wave_sum64(mok ? v : 0.f);
      if (lane == 0) part[wm * 64 + c] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __syncthreads();
  if (tid < 64) a.avg_out[(size_t)b * a.Cout + n0 + tid] = (part[tid] + part[64 + tid]) / (float)a.THWo;
}

template <int BM, int BN, int BK>
__device__ __forceinline__ void avg_epilogue(const ConvArgs& a, f32x4 (&acc)[BM / 32][BN / 32], float* smem, int b, int n0, int wave, int lane, int tid) {
  if (a.res) avg_epilogue_body<BM, BN, BK, true>(a, acc, smem, b, n0, wave, lane, tid);
  else avg_epilogue_body<BM, BN, BK, false>(a, acc, smem, b, n0, wave, lane, tid);
}

// LDS floats of one workgroup of the LDS-DMA kernel: the ring, reused as the epilogue's staging / brick area
template <int BM, int BN, int BK, int NS, int EPI>
constexpr int dma_smem_floats() {
  using Cfg = IgemmCfg<BM, BN, BK>;
  constexpr int RING = NS * DmaCfg<BM, BN, BK>::STAGE;
  constexpr int BRICK_FLOATS = EPI == EPI_STD ? 0 : (EPI == EPI_AVG ? Cfg::ST_FLOATS + 2 * BN : (BN / Cfg::FN) * (BM + 4));  // one fragment column of every wave: [BN/FN channels][BM + 4]
  constexpr int SMEM0 = RING > Cfg::ST_FLOATS ? RING : Cfg::ST_FLOATS;
  return SMEM0 > BRICK_FLOATS ? SMEM0 : BRICK_FLOATS;
}

// The body of the kernel: work item `item` of `nitems` (tile, K-slice) items whose m-tiles start at row `m_base` of the M index
// space.  conv3d_igemm_dma_kernel runs it on (blockIdx.x, the whole grid, 0); conv1x1_mixed_tail_kernel on two tile heights.
template <int BM, int BN, int BK, bool CHECK, int NS = 3, int EPI = EPI_STD, bool U8 = false, int AMODE = 0>
__device__ __forceinline__ void conv3d_igemm_dma_tile(const ConvArgs& a, float* smem, const int item, const int nitems, const int m_base) {
  using Cfg = IgemmCfg<BM, BN, BK>;
  using D = DmaCfg<BM, BN, BK>;
  constexpr int FM = Cfg::FM, FN = Cfg::FN, KR = D::KR, LA = D::LA, LB = D::LB, KS = BK / 4;
  static_assert(NS == 2 || NS == 3 || NS == 4, "ring depth");
  static_assert(LB >= 1 && LA >= 1 && (NS - 2) * (LA + LB) <= 63, "tile / vmcnt budget");
  constexpr unsigned OOB = 0xFFFFFF00u;
  static_assert(EPI == EPI_STD || (BM == 128 && BN == 64), "pooling epilogues: 128 x 64 tile (wave row = one t plane of the brick)");
  static_assert(EPI != EPI_AVG || (!CHECK && NS == 2 && !U8 && (AMODE == 0 || AMODE == 2)), "the mean epilogue: 1x1x1 convs on the 2-deep ring");
  static_assert(!U8 || (EPI == EPI_POOL233 && CHECK), "uint8 frame input: the stem + maxpool1 form (an m-tile lies in one crop)");
  // AMODE 2: the 16-byte-piece form of a 1x1x1 stride-1 conv (ConvArgs::a16) known at COMPILE time: the per-thread position
  // decode, the window origin and the gather table of the other forms drop out of the prologue -- which a new workgroup executes
  // beside the older workgroups' MFMA streams (profiles/r01_pmc_notes.md: 4-6 us of a short-K tile's life)
  constexpr bool A16ONLY = AMODE == 2;
  static_assert(!A16ONLY || (!CHECK && NS == 2 && !U8 && epi_rows(EPI)), "compile-time a16: unchecked 1x1x1 convs on the 2-deep ring");
  constexpr int BRICK_T = brick_t(EPI), BRICK_H = EPI == EPI_POOL233 ? 4 : 1, BRICK_W = BM / (BRICK_T * BRICK_H);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // one (tile, K-slice) work item per workgroup (persistent grids measured slower)
  int L, split;
  // each XCD works on a contiguous range of (tile, K-slice) items: the K-slices of a tile and the n-tiles of an
  // m-tile (which re-read the same activation rows) meet in one L2
  const int it = xcd_remap(item, nitems);
  if (a.splits > 1) {
    L = (int)a.dSplits.div((unsigned)it);
    split = it - L * a.splits;
  } else {
    split = 0;
    L = it;
  }
  const int tile_m = (int)a.dTilesN.div((unsigned)L), tile_n = L - tile_m * a.tiles_n;
  const int m0 = m_base + tile_m * BM, n0 = tile_n * BN;

  const int ml = tid % BM;
  const int kr = __builtin_amdgcn_readfirstlane(tid / BM);
  const int m = m0 + ml;
  unsigned vbase = OOB;
  unsigned vmask = 0;
  // output position of this thread's A column: (sample, ot, oh, ow)
  int pb = 0, pot = 0, poh = 0, pow_ = 0;
  bool pvalid;
  int bk_b = 0, bk_t = 0, bk_h = 0, bk_w = 0;  // brick forms: sample and brick coordinates of this m-tile
  if constexpr (A16ONLY) {
    pvalid = false;  // (nothing below needs this thread's own position)
  } else if constexpr (epi_rows(EPI)) {
    pvalid = m < a.M;
    if (pvalid) {
      pb = (int)a.dTHWo.div((unsigned)m);
      const int p = m - pb * a.MP;
      pvalid = p < a.THWo;
      pot = (int)a.dHWo.div((unsigned)p);
      const int q = p - pot * a.HWo;
      poh = (int)a.dWo.div((unsigned)q);
      pow_ = q - poh * a.Wo;
    }
  } else {
    // m-tile -> (sample, brick t, brick h, brick w); row ml of the tile -> position (dt, dh, dw) inside the brick, w fastest:
    // a wave's 64 rows are one t plane of the brick, its LDS-DMA pieces BRICK_H row segments of BRICK_W outputs
    bk_b = (int)a.dNb.div((unsigned)tile_m);
    const int r1 = tile_m - bk_b * (int)a.dNb.d;
    bk_t = (int)a.dNbhw.div((unsigned)r1);
    const int r2 = r1 - bk_t * (int)a.dNbhw.d;
    bk_h = (int)a.dNbw.div((unsigned)r2);
    bk_w = r2 - bk_h * a.nbw;
    pb = bk_b;
    pot = bk_t * BRICK_T + ml / (BRICK_H * BRICK_W);
    poh = bk_h * BRICK_H + (ml / BRICK_W) % BRICK_H;
    pow_ = bk_w * BRICK_W + ml % BRICK_W;
    pvalid = pot < a.To && poh < a.Ho && pow_ < a.Wo;
  }
  int u8_flip = 0;
  if (pvalid) {
    const int it0 = pot * a.st - a.pt, ih0 = poh * a.sh - a.ph, iw0 = pow_ * a.sw - a.pw;
    if constexpr (U8) {
      // byte offset of the window origin in the frames tensor (F, FH, FW, C): the crop's corner, then the clip-local
      // coordinates; a mirrored crop walks its source columns backwards (its table holds (kw-1-dw) * C, see build_ktab_u8)
      const int bg = a.u8_first + bk_b, clip = bg / 10, crop = bg - clip * 10, j = crop >= 5 ? crop - 5 : crop;
      u8_flip = crop >= 5;
      const int top = j == 4 ? a.u8_ctop : ((j >> 1) ? a.u8_FH - a.H : 0), left = j == 4 ? a.u8_cleft : ((j & 1) ? a.u8_FW - a.W : 0);
      const int FWC = a.u8_FW * a.Cin, FHWC = a.u8_FH * FWC;
      // (mirrored crops are five_crop(hflip(frame)): column c of such a crop is source column FW - 1 - (left + c))
      const int col = u8_flip ? a.u8_FW - 1 - left - iw0 - (a.kw_ - 1) : left + iw0;
      vbase = (unsigned)((clip * a.T + it0) * FHWC + (top + ih0) * FWC + col * a.Cin + a.pad_off);
    } else {
      vbase = (unsigned)(pb * a.x_bstride + it0 * a.HW + ih0 * a.W + iw0 + a.pad_off) * 4u;
    }
    if constexpr (CHECK) {
      vmask = tap_bits(it0, a.kt_, a.T) | (tap_bits(ih0, a.kh_, a.H) << 10) | (tap_bits(iw0, a.kw_, a.W) << 20);
    }
  }
  U8Corr u8c{};
  if constexpr (U8) {
    // this lane's epilogue row: t plane (wave >> 1), brick row (lane >> 4), channels n0 + 2 * ((wave & 1) * 16 + (lane & 15)) + jn
    const int ot = bk_t * 2 + (wave >> 1), oh = bk_h * BRICK_H + (lane >> 4);
    const int tc = u8_border_class(ot * a.st - a.pt, a.kt_, a.T, a.pt), hc = u8_border_class(oh * a.sh - a.ph, a.kh_, a.H, a.ph);
    const int ow0 = bk_w * BRICK_W, owl = ow0 + BRICK_W - 1 < a.Wo ? ow0 + BRICK_W - 1 : a.Wo - 1;
    const int wc0 = u8_border_class(ow0 * a.sw - a.pw, a.kw_, a.W, a.pw), wc1 = u8_border_class(owl * a.sw - a.pw, a.kw_, a.W, a.pw);
    const int nch = (a.ph + 1) * (a.ph + 1), ncw = (a.pw + 1) * (a.pw + 1);
    u8c.row = a.pad_corr + (size_t)((tc * nch + hc) * ncw) * a.Cout + (n0 + 2 * ((wave & 1) * 16 + (lane & 15)));
    u8c.uniform = wc0 == wc1;  // (classes are runs along w: equal at both ends = one class for the whole brick row)
    u8c.c[0] = u8c.row[(size_t)wc0 * a.Cout];
    u8c.c[1] = u8c.row[(size_t)wc0 * a.Cout + 1];
  }
  // AMODE 1: lane (lane & 31) owns the 4-column group (lane & 31) * 4 of the m-tile, lane >> 5 picks the k-row of the pair
  unsigned vbase_s = OOB, vmask_s = 0;
  constexpr int S_RPI = 256 / BM;   // k-rows one 16-byte wave-instruction fills (BM / 4 lanes each)
  constexpr int S_NI = 4 / S_RPI;   // such instructions per wave and k-tile (BK = 16: this wave's k-rows 4 wave .. 4 wave + 3)
  const int rsel = lane / (BM / 4); // which of the instruction's k-rows this lane fills
  const int rm[4] = {rsel == 0 ? -1 : 0, rsel == 1 ? -1 : 0, rsel == 2 ? -1 : 0, rsel == 3 ? -1 : 0};
  if constexpr (AMODE == 1) {
    static_assert(AMODE == 0 || (BK == 16 && NS == 2 && CHECK && !U8 && (BM == 256 || BM == 128 || BM == 64) &&
                                 (EPI == EPI_STD || ((EPI == EPI_POOL233 || EPI == EPI_TSPAN2 || EPI == EPI_TSPAN4) && BM == 128))),
                  "16-byte gather pieces: the fused stem (column-parity planes), the T-spanning tiles and plain tiles of (kt,1,1) convs");
    const int ml4 = (lane % (BM / 4)) * 4;
    if constexpr (EPI == EPI_STD) {
      // (kt,1,1) stride-1 conv, H*W a multiple of 4: a group of 4 consecutive m is 4 consecutive positions of one plane, and
      // tap dt reads the same 4 positions one plane on -- 16 contiguous bytes, all valid or all padding
      const int m4 = m0 + ml4;
      if (m4 < a.M) {
        const int b4 = (int)a.dTHWo.div((unsigned)m4);
        const int p4 = m4 - b4 * a.MP;
        const int ot = (int)a.dHWo.div((unsigned)p4);
        const int it0 = ot * a.st - a.pt;
        vbase_s = (unsigned)(b4 * a.x_bstride + it0 * a.HW + (p4 - ot * a.HWo) + a.pad_off) * 4u;
        vmask_s = tap_bits(it0, a.kt_, a.T) | (1u << 10) | (1u << 20);
      }
    } else {
      const int gt = bk_t * BRICK_T + ml4 / (BRICK_H * BRICK_W), gh = bk_h * BRICK_H + (ml4 / BRICK_W) % BRICK_H, gw = bk_w * BRICK_W + ml4 % BRICK_W;
      if (gt < a.To && gh < a.Ho && gw < a.Wo) {  // (Wo % 4 == 0: a group is all inside or all outside)
        const int it0 = gt * a.st - a.pt, ih0 = gh * a.sh - a.ph;
        vbase_s = (unsigned)(bk_b * a.x_bstride + (it0 * a.H + ih0) * a.s2w_rowp + gw + a.pad_off) * 4u;
        vmask_s = tap_bits(it0, a.kt_, a.T) | (tap_bits(ih0, a.kh_, a.H) << 10);
        // T-spanning tiles read the conv's own compact table, whose entries also carry the (always valid) w bit of a 1-wide
        // kernel; a group that straddles the end of a plane reads on into the next one for positions the epilogue never stores
        if constexpr (EPI == EPI_TSPAN2 || EPI == EPI_TSPAN4) vmask_s |= 1u << 20;
      }
    }
  }
  const auto rx = U8 ? __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(const_cast<float*>(a.x)) - a.pad_off, 0, a.x_bytes, 0x00020000)
                     : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) - a.pad_off, 0, a.x_bytes, 0x00020000);
  const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
  // (the brick's crop, hence the table, is the same for the whole workgroup: keep the pointer scalar for the s_loads)
  const int2* __restrict__ ktab2 = U8 ? a.ktab_u8 + (size_t)__builtin_amdgcn_readfirstlane((bk_b + a.u8_first) % 10 >= 5 ? a.Kpad : 0)
                                      : reinterpret_cast<const int2*>(a.ktab + a.Kpad);
  constexpr int A_BYTES = U8 ? 1 : 4;  // uint8 input: `buffer_load_ubyte ... lds` puts the zero-extended byte in the lane's LDS dword
  // B rows: a wave's 64 lanes x 16 B cover RPW consecutive k-rows of the [BK][BN] tile
  constexpr int LPRB = BN / 4;
  constexpr int RPW = 64 / LPRB;
  const unsigned wvoff = (unsigned)((lane / LPRB) * a.Cout + (lane % LPRB) * 4) * 4u;
  const int a_wave_col = (wave % (BM / 64)) * 64;  // which 64-float piece of an A row this wave fills
  // 1x1x1 stride-1 convs (row k of A = channel k's positions, contiguous and 16-byte aligned: a.a16): the A tile goes to LDS
  // in 16-byte pieces -- one wave-instruction = 1 KiB = 256/BM whole k-rows -- instead of one 4-byte piece per row and
  // wave: 4x fewer VMEM issues per tile (the LDS-DMA issue of the 4-byte form costs ~10 % of such a kernel,
  // profiles/r01_pmc_notes.md).  Tap offsets are linear in k here (k * THW), so no table is read.  2-deep ring only: its
  // waits are vmcnt(0) whatever the number of pieces.
  constexpr bool CAN16 = !CHECK && NS == 2 && epi_rows(EPI);
  constexpr int RPI16 = 256 / BM, LA16 = BK * BM / 1024;
  static_assert(LA16 >= 1, "tile too small for 16-byte A pieces");
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int A16_BYTES = 16;  // (the host pass only parses this body; see gemm_kk_dma_kernel)
#else
  constexpr int A16_BYTES = 4;
#endif
  const bool a16 = A16ONLY || (CAN16 && a.a16 != 0);
  unsigned vbase16 = OOB;
  if constexpr (CAN16) {
    if (a16) {
      const int m4 = m0 + (lane % (BM / 4)) * 4;
      if (m4 < a.M) {
        const int b4 = (int)a.dTHWo.div((unsigned)m4);
        vbase16 = (unsigned)(b4 * a.x_bstride + (m4 - b4 * a.MP) + (lane / (BM / 4)) * a.THW) * 4u;
      }
    }
  }

  // wave group kr fills k-rows [kr*LA, (kr+1)*LA) of a tile: its table entries are contiguous.
  // (issue_part can also issue a 1/nparts slice of a tile's loads; the product issues whole tiles.)
  int ent[2 * LA];
  auto issue_part = [&](int k0, int stage, int part, int nparts) {
    float* As = smem + stage * D::STAGE;
    float* Bs = As + BK * BM;
    bool done16 = false;
    if constexpr (AMODE == 1) {
      // piece g = k-rows [g * S_RPI, (g + 1) * S_RPI) of the tile; this wave's four k-rows have their entries in ent[0..7]
#pragma unroll
      for (int q = 0; q < S_NI; ++q) {
        const int g = wave * S_NI + q;
        int eo, eb;
        if constexpr (S_RPI == 1) {  // (BM = 256: an instruction's 64 lanes are one k-row)
          eo = ent[2 * q];
          eb = ent[2 * q + 1];
        } else if constexpr (S_RPI == 2) {
          eo = rsel ? ent[4 * q + 2] : ent[4 * q];
          eb = rsel ? ent[4 * q + 3] : ent[4 * q + 1];
        } else {
          // four-way pick by per-lane masks (set once, outside the loop): a `rsel == i ? ent[..]` chain makes hipcc index a
          // stack copy of the entries -- scratch loads inside the k loop, which also count against the ring's vmcnt waits
          eo = (ent[0] & rm[0]) | (ent[2] & rm[1]) | (ent[4] & rm[2]) | (ent[6] & rm[3]);
          eb = (ent[1] & rm[0]) | (ent[3] & rm[1]) | (ent[5] & rm[2]) | (ent[7] & rm[3]);
        }
        const unsigned voff = ((vmask_s & (unsigned)eb) == (unsigned)eb) ? vbase_s + (unsigned)eo : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(As + g * S_RPI * BM), A16_BYTES, voff, 0, 0, 0);
      }
      done16 = true;
    }
    if constexpr (CAN16) {
      if (a16) {
#pragma unroll
        for (int q = 0; q < LA16; ++q) {
          const int g = wave * LA16 + q;  // piece g = k-rows [g * RPI16, (g + 1) * RPI16) of the tile
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(As + g * RPI16 * BM), A16_BYTES, vbase16, (k0 + g * RPI16) * a.THW * 4, 0, 0);
        }
        done16 = true;
      }
    }
    if constexpr (!A16ONLY) if (!done16) {
#pragma unroll
      for (int j = 0; j < LA; ++j) {
        if (j * nparts / LA != part) continue;
        const int krow = kr * LA + j;
        unsigned voff = vbase;
        if constexpr (CHECK) voff = ((vmask & (unsigned)ent[2 * j + 1]) == (unsigned)ent[2 * j + 1]) ? vbase : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(As + krow * BM + a_wave_col), A_BYTES, voff, ent[2 * j], 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      if (j * nparts / LB != part) continue;
      const int row0 = (wave * LB + j) * RPW;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(Bs + row0 * BN), 16, wvoff, ((k0 + row0) * a.Cout + n0) * 4, 0, 0);
    }
  };
  auto load_entries = [&](int k0) {
    if constexpr (AMODE == 1) {
      int e4[8];
      sload_entries<4>(a.ktab_s2w, (k0 + wave * 4) * 8, e4);
#pragma unroll
      for (int i = 0; i < 8; ++i) ent[i] = e4[i];
    } else if constexpr (!A16ONLY) {
      if (!a16) sload_entries<LA>(ktab2, (k0 + kr * LA) * 8, ent);
    }
  };
  auto issue_tile = [&](int k0, int stage) {
    load_entries(k0);
    issue_part(k0, stage, 0, 1);
  };

  const int wm = Cfg::wave_m(wave), wn = Cfg::wave_n(wave);
  const int li = lane & 15, lg = lane >> 4;
  const int a_col = wm * Cfg::WM + FM * li;
  const int b_col = wn * Cfg::WN + FN * li;
  // per-lane LDS byte addresses of the k-step-0 fragments in ring stage 0
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
  const unsigned a_addr0 = lds0 + (unsigned)(lg * BM + a_col) * 4u;
  const unsigned b_addr0 = lds0 + (unsigned)(BK * BM + lg * BN + b_col) * 4u;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto mfma_step = [&](const Frag<FM>& fa, const Frag<FN>& fb) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.v[i], fb.v[j], acc[i][j], 0, 0, 0);
  };

  // k-step ks of a stage sits ks*4 rows further down both tiles -> immediate offsets.
  // `pre`: also issue tile pre_k0 into ring stage `pstage` (a wave-uniform branch around the loads only: one copy of
  // the MFMA body, so the accumulators keep their registers across iterations).
  auto compute = [&](bool pre, int stage, int pre_k0, int pstage) {
    const unsigned aa = a_addr0 + (unsigned)(stage * D::STAGE) * 4u;
    const unsigned ba = b_addr0 + (unsigned)(stage * D::STAGE) * 4u;
    Frag<FM> fa[2];
    Frag<FN> fb[2];
    // the whole next tile's loads go out right after the barrier (micro-benchmark tools/dma_ceiling.hip modes 9/10 and
    // an A/B on the stack: one slice per k-step between the MFMA groups is 0.3-0.9 % slower)
    if (pre) issue_part(pre_k0, pstage, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    lds_read<FM, 0>(fa[0], aa);
    lds_read<FN, 0>(fb[0], ba);
    // uint8 input: the A operands are pixel bytes (zero-extended by the LDS-DMA), one v_cvt_f32_ubyte0 each.  The operand is
    // the pixel itself; -mean * (sum of the weights of the taps inside the clip) comes from the border-class table in the
    // epilogue (taps outside read 0 and contribute nothing, as padding should).  Most of the time one wave per SIMD is in
    // its MFMA phase (the others wait for their tiles), so a cvt -> MFMA dependency is exposed: the operands of step ks+1
    // are converted beside the second half of step ks's MFMAs, and only step 0's conversion stands in front of its MFMAs.
    auto cvt_u8 = [&](Frag<FM>& f) {
      decltype(f.v) cv;  // (built in a fresh vector: hipcc 7.2 miscompiles the in-place per-element form `v[i] = f(v[i])`)
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const float e = f.v[i];
        cv[i] = u8_to_f32(e);
      }
      f.v = cv;
    };
    auto body = [&](auto ks_c) {
      constexpr int ks = decltype(ks_c)::value;
      if constexpr (U8) {
        static_assert(!U8 || FM % 2 == 0, "uint8 input: two halves of the fragment rows");
        if constexpr (ks == 0) {
          lds_wait<0>(fa[0], fb[0]);
          cvt_u8(fa[0]);
        }
        if constexpr (ks + 1 < KS) {
          lds_read<FM, (ks + 1) * 4 * BM * 4>(fa[(ks + 1) & 1], aa);
          lds_read<FN, (ks + 1) * 4 * BN * 4>(fb[(ks + 1) & 1], ba);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FM / 2; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ks & 1].v[i], fb[ks & 1].v[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ks + 1 < KS) {
          lds_wait<0>(fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
          cvt_u8(fa[(ks + 1) & 1]);
        }
#pragma unroll
        for (int i = FM / 2; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ks & 1].v[i], fb[ks & 1].v[j], acc[i][j], 0, 0, 0);
        if constexpr (ks + 1 < KS) {
#pragma unroll
          for (int i = 0; i < FM; ++i) {  // one MFMA, one conversion of the next step beside it
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x2, 1, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        return;
      }
      if constexpr (ks + 1 < KS) {
        lds_read<FM, (ks + 1) * 4 * BM * 4>(fa[(ks + 1) & 1], aa);
        lds_read<FN, (ks + 1) * 4 * BN * 4>(fb[(ks + 1) & 1], ba);
        lds_wait<2>(fa[ks & 1], fb[ks & 1]);  // the two reads of step ks+1 may stay in flight
      } else {
        lds_wait<0>(fa[ks & 1], fb[ks & 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(fa[ks & 1], fb[ks & 1]);
      __builtin_amdgcn_sched_barrier(0);
    };
    [&]<int... I>(std::integer_sequence<int, I...>) { (body(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, KS>{});
  };

  const int nk_all = a.Kpad / BK;
  const int kt0 = (nk_all * split) / a.splits;
  const int kt1 = (nk_all * (split + 1)) / a.splits;
  // prologue: NS-1 k-tiles in flight
#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (kt0 + i < kt1) issue_tile((kt0 + i) * BK, i);
  int stage = 0;
  for (int kt = kt0; kt < kt1; ++kt) {
    // tile kt has landed once only the loads of the younger tiles in flight (<= NS-2) are outstanding
    const int younger = kt1 - 1 - kt;
    if (NS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the only tile in flight is tile kt
    else if (NS == 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (LA + LB)) : "memory");
    else if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LA + LB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");  // all waves' parts of tile kt are in LDS; stage (kt-1)%NS is free
    const bool pre = kt + NS - 1 < kt1;
    if (pre) load_entries((kt + NS - 1) * BK);
    compute(pre, stage, (kt + NS - 1) * BK, stage == 0 ? NS - 1 : stage - 1);
    stage = stage == NS - 1 ? 0 : stage + 1;
  }
  __syncthreads();  // every wave is done with the ring before the epilogue reuses it as staging
  if (a.splits > 1) {  // the launcher always gives this kernel family a partial-tile workspace
    // ---- in-kernel split-K reduction (cdna_hip_programming.md section 5, "In-launch split-K reduction", write-through form)
    // publish: thread tid owns float4 (i*FN+j) of its accumulators at [(i*FN+j)*256 + tid] of the (tile, slice) block:
    // whole 16-byte sc1 (write-through) stores, 4 KiB contiguous per store instruction of the workgroup
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    const auto rp = __builtin_amdgcn_make_buffer_rsrc(a.part, 0, a.part_bytes, 0x00020000);
    constexpr unsigned TILE_BYTES = BM * BN * 4;
    const unsigned my = (unsigned)(L * a.splits + split) * TILE_BYTES + (unsigned)tid * 16u;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rp, my + (unsigned)(i * FN + j) * 4096u, 0, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave drains its write-through stores ...
    __syncthreads();                                   // ... before ONE lane signals for all of them
    unsigned* flag = reinterpret_cast<unsigned*>(smem);
    if (tid == 0) *flag = __hip_atomic_fetch_add(a.cnt + L, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool last = *flag == (unsigned)(a.splits - 1);
    __syncthreads();  // the flag word is staging space again below
    if (!last) return;
    // (the tile's counter goes back to zero for the next launch that uses it: a caller-owned counter block is zeroed once, when
    // it is allocated -- advhip_conv3d_epilogue.splitk_counters -- and no memset runs ahead of the launch)
    if (tid == 0) __hip_atomic_store(a.cnt + L, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the last arriver: every slice of this tile has been published (write-through, drained before its ticket).  sc1
    // loads bypass this CU's L1 (never refreshed by other CUs' stores); sum in slice order -> run-to-run bit-identical
    // (in batches of G float4 per thread: the sums must not cost the main loop registers -- the occupancy target above
    // is what keeps the matrix pipe busy -- so at most G loads per lane are in flight, one slice at a time.  G = 8 fits the
    // registers of every plan kernel too and halves the number of load latencies on the last arriver's path; measured on the
    // stream it changes nothing: 3 826 / 3 841 against 3 834 / 3 844 clips/s, alternating runs on one box)
    const unsigned t0 = (unsigned)(L * a.splits) * TILE_BYTES + (unsigned)tid * 16u;
    constexpr int NV = FM * FN, G = NV < SPLITK_SUM_BATCH ? NV : SPLITK_SUM_BATCH;
    auto at = [&](int v) -> f32x4& { return acc[v / FN][v % FN]; };
#pragma unroll
    for (int g0 = 0; g0 < NV; g0 += G) {
#pragma unroll
      for (int v = 0; v < G; ++v)
        at(g0 + v) = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, t0 + (unsigned)(g0 + v) * 4096u, 0, 16));
#pragma unroll 1
      for (int sl = 1; sl < a.splits; ++sl) {
        const unsigned ts = t0 + (unsigned)sl * TILE_BYTES;
        f32x4 t[G];
#pragma unroll
        for (int v = 0; v < G; ++v)
          t[v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, ts + (unsigned)(g0 + v) * 4096u, 0, 16));
#pragma unroll
        for (int v = 0; v < G; ++v) at(g0 + v) += t[v];
      }
    }
  }
  if constexpr (EPI == EPI_AVG) {
    avg_epilogue<BM, BN, BK>(a, acc, smem, tile_m, n0, wave, lane, tid);
    return;
  } else if constexpr (EPI != EPI_STD) {
    brick_epilogue<BM, BN, BK, EPI, U8>(a, acc, smem, tile_m, tile_n, n0, bk_b, bk_t, bk_h, bk_w, wave, lane, tid, u8c);
    return;
  }
  // (one epilogue call site: a second inlined copy costs ~25 VGPRs and with them a resident workgroup per CU)
  igemm_epilogue<BM, BN, BK>(a, acc, smem, 0, m0, n0, wave, lane, true);
}

template <int BM, int BN, int BK, bool CHECK, int NS = 3, int EPI = EPI_STD, bool U8 = false, int AMODE = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(dma_waves_per_simd<BM, BN, BK, NS>(), 8)))
void conv3d_igemm_dma_kernel(const ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[dma_smem_floats<BM, BN, BK, NS, EPI>()];
  conv3d_igemm_dma_tile<BM, BN, BK, CHECK, NS, EPI, U8, AMODE>(a, smem, (int)blockIdx.x, a.tiles_m * a.tiles_n * a.splits, 0);
}

// ---- one launch, two tile heights: the tail of a launch that is just over a whole number of rounds -------------------------------
// An unsplit 1x1x1 stride-1 conv on 16-byte aligned rows (the `conv3` + residual launches, src/i3d.py:85-89, 108-121) whose 128 x 64
// tiles number a little more than the resident slots (layer3.x.conv3 at B = 32: 98 x 16 = 1 568 tiles on 256 CUs x 6 = 1 536) pays a
// second round in which 32 workgroups run alone (profiles/r05_tile_tail.txt: 62.2 us against 54.6 at exactly one round).  Here the last
// m-tile rows of the launch are cut into 64-row tiles, enough of them that every workgroup of the second round is a 64 x 64 one -- half
// the MFMA chain per k-tile, so the tail is about half as long.  Items [0, mix_big) are 128 x 64 tiles, the rest 64 x 64 tiles from row
// mix_mbase on; both bodies are the ones of conv3d_igemm_dma_kernel (same K order and accumulation per output: the same bits).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(dma_waves_per_simd<128, 64, 16, 2>(), 8)))
void conv1x1_mixed_tail_kernel(const ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[dma_smem_floats<128, 64, 16, 2, EPI_STD>()];
  static_assert(dma_smem_floats<128, 64, 16, 2, EPI_STD>() >= dma_smem_floats<64, 64, 16, 2, EPI_STD>(), "the tall tile's LDS holds the short one's");
  const int bid = (int)blockIdx.x;
  if (bid < a.mix_big) conv3d_igemm_dma_tile<128, 64, 16, false, 2, EPI_STD, false, 2>(a, smem, bid, a.mix_big, 0);
  else conv3d_igemm_dma_tile<64, 64, 16, false, 2, EPI_STD, false, 2>(a, smem, bid - a.mix_big, a.mix_small, a.mix_mbase);
}

// ================================================================================================
// Persistent, wave-specialised kernel for unsplit 1x1x1 stride-1 convs on 16-byte aligned rows (ConvArgs::a16): the `conv3`
// (+ residual) launches of every Bottleneck (src/i3d.py:85-89, 108-121) and the k = 1 `conv1`s.  OPT-IN (ADVHIP_ALGO_PERSIST_BASE):
// measured slower than the tuned one-tile-per-workgroup kernels on every B = 32 shape (profiles/r04_persist_kernel_study.md has the
// five forms that were built, their timings and in-kernel cycle stamps, and why); kept as the tested end point of that study.
//   * a launch is W x (compute units) workgroups of EIGHT waves that stay; each walks a contiguous share of its XCD's part of the
//     output tiles (n fastest: the n-tiles of an m-tile re-read the same activation rows from the XCD's L2);
//   * waves 0-3 only feed the matrix pipe: barrier, fragment reads, MFMAs.  At the end of a tile they apply scale / shift and
//     drop the tile into an LDS staging area (8 ds_write_b128 per lane), then go on with the next tile;
//   * waves 4-7 are the loaders and the epilogue.  Loader wave w fetches the k-tiles q = w (mod 4) -- whole k-tiles, plain 16-byte
//     buffer loads into registers, four k-tiles ahead and ACROSS tile boundaries, so no tile after a workgroup's first waits for a
//     fill -- and writes each to the two-stage LDS ring one interval before it is multiplied.  One interval after a tile was
//     staged they issue its residual loads; two intervals later they add, apply the activation and store whole 16-byte pieces of
//     NCDHW rows;
//   * the ONE barrier per k-tile is all the synchronisation there is: both roles execute the same number of barriers; a staged
//     tile is written after the barrier of its last k-tile and read in the third interval of its successor (nk >= 4).
// Same operand layout, k order and accumulation chain as conv3d_igemm_dma_kernel (unsplit): bit-identical results.
// (registers: two 8-wave workgroups per CU for the 128-row tile = 4 waves per SIMD = 128 VGPRs; three for the 64-row one)
constexpr int PERSIST_MIN_NK = 4;  // k-tiles per output tile the staging hand-over needs (checked by the launcher AND the kernel)
constexpr int GFX950_XCDS = 8;      // workgroups are dealt round-robin to the XCDs in dispatch order (MI355X: 8 XCDs x 32 CUs)
template <int BM, int BN, int NS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 8)))
void conv1x1_persist_kernel(const ConvArgs a, const int n_xcd) {
  constexpr int BK = 16;
  using Cfg = IgemmCfg<BM, BN, BK>;
  using D = DmaCfg<BM, BN, BK>;
  constexpr int FM = Cfg::FM, FN = Cfg::FN, LB = D::LB, KS = BK / 4;
  static_assert(NS == 2, "LDS ring depth (the depth of the prefetch is the loader waves' registers)");
  static_assert(BN == 64 && (BM == 128 || BM == 64), "tile");
  constexpr unsigned OOB = 0xFFFFFF00u;
  constexpr int RING = NS * D::STAGE;
  constexpr int RPI16 = 256 / BM, LA16 = BK * BM / 1024;  // 16-byte A pieces: k-rows per wave-instruction, instructions per loader wave and k-tile
  constexpr int LPRB = BN / 4, RPW = 64 / LPRB;
  constexpr int PER_TILE = LA16 + LB;                      // LDS-DMA instructions per loader wave and k-tile
  constexpr int SLOTS = BM / 4;                            // float4 slots per staged channel row
  constexpr int EPT = BM * BN / 4 / 256;                   // float4 per epilogue thread and tile (8 or 4)
  constexpr int ROWS_PER_PASS = 256 / SLOTS;               // channel rows the 256 epilogue threads cover per pass
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int A16_BYTES = 16;  // (the host pass only parses this body; see gemm_kk_dma_kernel)
#else
  constexpr int A16_BYTES = 4;
#endif
  __shared__ __attribute__((aligned(16))) float smem[RING + BM * BN];
  // staging: [BN channel rows][BM positions], the float4 slots of row n rotated by n (bank spread): never a ring stage

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // this workgroup's tiles: XCD x (= blockIdx % 8, the dispatch order) owns a contiguous eighth of the linear tile order
  // L = tile_m * tiles_n + tile_n, workgroup j of the XCD's gridDim / 8 a contiguous part of that
  const int NT = a.tiles_m * a.tiles_n;
  // (the launcher makes gridDim.x a multiple of n_xcd)
  const int gx = (int)gridDim.x / n_xcd, xcd = (int)blockIdx.x % n_xcd, jx = (int)blockIdx.x / n_xcd;
  const int lo = (int)(((long long)NT * xcd) / n_xcd), hi = (int)(((long long)NT * (xcd + 1)) / n_xcd);
  const int t0 = lo + (int)(((long long)(hi - lo) * jx) / gx), t1 = lo + (int)(((long long)(hi - lo) * (jx + 1)) / gx);
  if (t0 >= t1) return;
  const int nk = a.Kpad / BK;
  if (nk < PERSIST_MIN_NK) return;   // (the launcher refuses such a K; never let a shorter one race the staging area)
  const int total = (t1 - t0) * nk;  // k-tiles of this workgroup, all its output tiles one after the other
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
  const unsigned stg0 = lds0 + (unsigned)RING * 4u;

  if (wave < 4) {
    // ---------------------------------------------------------------- MFMA waves
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    const unsigned a_addr0 = lds0 + (unsigned)(lg * BM + wm * Cfg::WM + FM * li) * 4u;
    const unsigned b_addr0 = lds0 + (unsigned)(BK * BM + lg * BN + wn * Cfg::WN + FN * li) * 4u;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int jn = 0; jn < FN; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int stage) {
      const unsigned aa = a_addr0 + (unsigned)(stage * D::STAGE) * 4u;
      const unsigned ba = b_addr0 + (unsigned)(stage * D::STAGE) * 4u;
      Frag<FM> fa[2];
      Frag<FN> fb[2];
      lds_read<FM, 0>(fa[0], aa);
      lds_read<FN, 0>(fb[0], ba);
      auto body = [&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        if constexpr (ks + 1 < KS) {
          lds_read<FM, (ks + 1) * 4 * BM * 4>(fa[(ks + 1) & 1], aa);
          lds_read<FN, (ks + 1) * 4 * BN * 4>(fb[(ks + 1) & 1], ba);
          lds_wait<2>(fa[ks & 1], fb[ks & 1]);
        } else {
          lds_wait<0>(fa[ks & 1], fb[ks & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int jn = 0; jn < FN; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ks & 1].v[i], fb[ks & 1].v[jn], acc[i][jn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      [&]<int... I>(std::integer_sequence<int, I...>) { (body(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, KS>{});
    };

    // scale / shift of this lane's FN channels of the current tile (these waves' only global loads: nothing else counts on their vmcnt)
    float sc[FN], sf[FN];
    auto load_scale = [&](int L) {
      const int tm = (int)a.dTilesN.div((unsigned)L), tn = L - tm * a.tiles_n;
#pragma unroll
      for (int jn = 0; jn < FN; ++jn) {
        const int n = tn * BN + wn * Cfg::WN + FN * li + jn;
        sc[jn] = a.scale[n];
        sf[jn] = a.shift[n];
      }
    };
    load_scale(t0);
    int stage = 0, kt = 0, tile = t0;
    for (int q = 0; q < total; ++q) {
      asm volatile("s_barrier" ::: "memory");  // k-tile q is in LDS (the loaders waited for it before they arrived)
      compute(stage);
      stage = stage == NS - 1 ? 0 : stage + 1;
      if (++kt == nk) {
        kt = 0;
        // hand the tile to the epilogue waves: accumulator element acc[jm][jn][r] of lane (li, lg) is output (m, n) =
        // (wm*WM + FM*(4 lg + r) + jm, wn*WN + FN*li + jn) -- a lane holds 4 FM consecutive m of channel row n: whole float4 slots.
        // (They read the previous tile's staging in the interval after this tile's FIRST barrier: long done.  asm stores: hipcc
        // would put `s_waitcnt vmcnt(0)` in front of its own LDS stores in a kernel with LDS-DMA loads pending.)
#pragma unroll
        for (int jn = 0; jn < FN; ++jn) {
          const int n = wn * Cfg::WN + FN * li + jn;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (FM == 4) {
              const int slot = (wm * Cfg::WM + 16 * lg + 4 * r) >> 2;
              const unsigned addr = stg0 + (unsigned)(n * BM + (((slot ^ n) & (SLOTS - 1)) << 2)) * 4u;
              const f32x4 t = {acc[0][jn][r] * sc[jn] + sf[jn], acc[1][jn][r] * sc[jn] + sf[jn], acc[2][jn][r] * sc[jn] + sf[jn],
                               acc[3][jn][r] * sc[jn] + sf[jn]};
              asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(t) : "memory");
            } else {
              using f32x2 = __attribute__((ext_vector_type(2))) float;
              const int m = wm * Cfg::WM + 8 * lg + 2 * r;  // FM == 2: two consecutive m per (r): half a slot
              const int slot = m >> 2;
              const unsigned addr = stg0 + (unsigned)(n * BM + (((slot ^ n) & (SLOTS - 1)) << 2) + (m & 3)) * 4u;
              const f32x2 t = {acc[0][jn][r] * sc[jn] + sf[jn], acc[1][jn][r] * sc[jn] + sf[jn]};
              asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(t) : "memory");
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staged tile is in LDS before this wave reaches the next barrier
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int jn = 0; jn < FN; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (++tile < t1) load_scale(tile);
      }
    }
    asm volatile("s_barrier" ::: "memory");  // the last tile is staged
    return;
  }

  // ------------------------------------------------------------------ loader + epilogue waves
  // Operands go global -> registers -> LDS here: plain 16-byte buffer loads, ds_write_b128 one interval before the k-tile is
  // multiplied.  LDS-DMA does not serve a loader: with four to eight waves issuing, `buffer_load ... lds` moves ~1 KB per ~480 cycles
  // and SIMD whatever the ring depth or issue priority (in-kernel stamps, tools/persist_stamps.py: 8.5 B/clk per CU, the loaders
  // became the bottleneck at ~2 000 cycles per k-tile against 1 200 for the MFMA waves); ordinary loads have no such limit and these
  // waves have the registers.  Loader wave w takes the k-tiles q = w (mod 4), a WHOLE k-tile each (12 loads per lane), so that
  // every wave runs the same code on one register set and up to four k-tiles are in flight per workgroup: the loads of k-tile
  // q + 4 go out in the interval in which k-tile q was written to LDS, and are written three intervals later.  Every
  // vector-memory operation of these waves is visible to hipcc, which counts its waits itself (no LDS-DMA left in the kernel).
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
  constexpr int NA = BK * BM / 256, NB = BK * BN / 256;  // 16-byte loads per lane for a whole k-tile of A / B (8 or 4, and 4)
  const int lw = wave - 4;
  const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
  const unsigned wvoff = (unsigned)((lane / LPRB) * a.Cout + (lane % LPRB) * 4) * 4u;
  const int a_lane4 = (lane % (BM / 4)) * 4, a_lrow = lane / (BM / 4);
  // the k-tile this wave fetches next: k-tile p_kt of tile p_tile (global index p_q = lw, lw + 4, ...)
  int p_tile = t0, p_kt = lw, p_q = lw, p_n0 = 0;
  unsigned p_vbase = OOB;
  auto set_prefetch_tile = [&](int L) __attribute__((always_inline)) {
    const int tm = (int)a.dTilesN.div((unsigned)L), tn = L - tm * a.tiles_n;
    p_n0 = tn * BN;
    const int m4 = tm * BM + a_lane4;
    p_vbase = OOB;
    if (m4 < a.M) {
      const int b4 = (int)a.dTHWo.div((unsigned)m4);
      p_vbase = (unsigned)(b4 * a.x_bstride + (m4 - b4 * a.MP) + a_lrow * a.THW) * 4u;
    }
  };
  set_prefetch_tile(t0);  // (nk >= 4 > lw: the first k-tile of every loader wave lies in the first tile)
  u32x4 ra[NA], rb[NB];
  auto load_ktile = [&]() __attribute__((always_inline)) {  // k-tile (p_tile, p_kt) -> registers, then advance by four k-tiles
    const int k0 = p_kt * BK;
#pragma unroll
    for (int g = 0; g < NA; ++g) ra[g] = __builtin_amdgcn_raw_buffer_load_b128(rx, p_vbase, (k0 + g * RPI16) * a.THW * 4, 0);
#pragma unroll
    for (int g = 0; g < NB; ++g) rb[g] = __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff, ((k0 + g * RPW) * a.Cout + p_n0) * 4, 0);
    p_q += 4;
    p_kt += 4;
    if (p_kt >= nk) {
      p_kt -= nk;
      if (++p_tile < t1) set_prefetch_tile(p_tile);
    }
  };
  auto write_ktile = [&](int stage) __attribute__((always_inline)) {  // registers -> ring stage (the image the LDS-DMA form writes: linear in the lane)
    float* As = smem + stage * D::STAGE;
    float* Bs = As + BK * BM;
#pragma unroll
    for (int g = 0; g < NA; ++g) *reinterpret_cast<u32x4*>(As + g * RPI16 * BM + lane * 4) = ra[g];
#pragma unroll
    for (int g = 0; g < NB; ++g) *reinterpret_cast<u32x4*>(Bs + g * RPW * BN + lane * 4) = rb[g];
  };

  const int e = tid - 256;                       // 0..255
  const int slot = e % SLOTS, row0 = e / SLOTS;  // this thread's float4 slot and first channel row of the staged tile
  const bool vec = a.vw == 4;
  const bool res_vec = vec && a.res != nullptr;  // (wave-uniform: kernel arguments)
  f32x4 rv[EPT];
  int em = 0, eb = 0, epp = 0, en0 = 0;  // the tile being finished: m of this thread's slot, its sample / position, the tile's first channel
  bool eok = false;
  auto pick_up = [&](int L) __attribute__((always_inline)) {  // the staged tile's coordinates; its residual loads issued
    const int tm = (int)a.dTilesN.div((unsigned)L), tn = L - tm * a.tiles_n;
    en0 = tn * BN;
    em = tm * BM + slot * 4;
    eok = em < a.M;
    eb = 0; epp = 0;
    if (eok) { eb = (int)a.dTHWo.div((unsigned)em); epp = em - eb * a.MP; }
    if (res_vec) {
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        const int n = en0 + row0 + ROWS_PER_PASS * i;
        const float* src = eok ? a.res + ((size_t)(eb * a.Cout + n) * a.THWo + epp) : a.res;  // (rows past M: the tensor's first bytes, never used)
        rv[i] = *reinterpret_cast<const f32x4*>(src);
      }
    }
  };
  // two intervals later (the staged tile stays put until the end of its successor's LAST k-tile interval; nk >= 4): staged values
  // + residual, activation, stores -- one channel row at a time, so that only the residual prefetch holds registers meanwhile
  auto finish = [&]() __attribute__((always_inline)) {
    if (!eok) return;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int nl = row0 + ROWS_PER_PASS * i, n = en0 + nl;
      const f32x4 vv = *reinterpret_cast<const f32x4*>(smem + RING + nl * BM + (((slot ^ nl) & (SLOTS - 1)) << 2));
      const size_t o = (size_t)(eb * a.Cout + n) * a.THWo + epp;                    // dense: residual
      const size_t oy = (size_t)eb * a.y_bstride + (size_t)n * a.THWo + epp;        // output proper
      float x4[4] = {vv[0], vv[1], vv[2], vv[3]};
      if (vec) {
        if (a.res) { x4[0] += rv[i][0]; x4[1] += rv[i][1]; x4[2] += rv[i][2]; x4[3] += rv[i][3]; }
        if (a.y2) {
          float sv[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) sv[c] = act_second(a.relu, x4[c]);
          vec_store<4>(a.y2 + oy, sv);
        }
        if (a.relu) {
#pragma unroll
          for (int c = 0; c < 4; ++c) x4[c] = act_apply(a.relu, x4[c]);
        }
        vec_store<4>(a.y + oy, x4);
      } else {
        // rows that are not a multiple of 4 positions long (or unaligned pointers): one position at a time; positions >= THWo of a
        // sample are the padding of the M index space (ConvArgs::MP)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (epp + c < a.THWo) {
            float val = x4[c];
            if (a.res) val += a.res[o + c];
            if (a.y2) a.y2[oy + c] = act_second(a.relu, val);
            if (a.relu) val = act_apply(a.relu, val);
            a.y[oy + c] = val;
          }
        }
      }
    }
  };

  // prologue: every loader wave has its first k-tile on the way; k-tile 0 is in ring stage 0 before the first barrier
  int w_q = lw;  // the k-tile in this wave's registers (written to LDS in interval w_q - 1)
  if (p_q < total) load_ktile();
  if (lw == 0) {
    write_ktile(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    w_q = 4;
    if (p_q < total) load_ktile();
  }
  {
    constexpr int fin = 2;  // the k-tile interval (of the NEXT tile) in which a picked-up tile is finished (nk >= 4: the launcher checks)
    int kt = 0, tile = t0;
    bool have = false;     // a finished tile of this workgroup is waiting in the staging area
    for (int q = 0; q < total; ++q) {
      asm volatile("s_barrier" ::: "memory");  // releases k-tile q to the MFMA waves; they are done with ring stage (q - 1) & 1
      if (have) {
        if (kt == 0) pick_up(tile - 1);  // staged after the previous tile's last barrier, visible since this one
        if (kt == fin) finish();
      }
      if (w_q == q + 1 && q + 1 < total) {  // this wave's turn: k-tile q + 1 into the stage the MFMA waves have just left
        write_ktile((q + 1) & 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        w_q += 4;
        if (p_q < total) load_ktile();
      }
      if (++kt == nk) { kt = 0; ++tile; have = true; }
    }
    asm volatile("s_barrier" ::: "memory");
    pick_up(t1 - 1);
    finish();
  }
}

// ================================================================================================
// The stem on resized uint8 frames (F, FH, FW, 3), whole pixels at a time.  The byte-gather form above issues one
// `buffer_load_ubyte ... lds` per (channel, tap) row and 64 positions -- as many LDS-DMA instructions as the fp32 kernel, and
// the vector issue port those share with the MFMAs is what bounds that kernel (the extra v_cvt of the byte form then costs
// its full issue time: 2.61 vs 2.43 ms at B = 32).  Here K runs tap-major, channel-minor (k' = tap * 3 + c) and ONE 4-byte
// LDS-DMA per (tap, position) fetches the pixel's three channel bytes (+ one byte of its neighbour, never used) at the
// pixel's byte address -- LDS-DMA and ds_read_b128 take byte-granular addresses on gfx950 (tools/probe/): a third of the A
// loads for the same K.  LDS tile: A [taps][128 positions] pixel dwords, B [3 * taps][64] fp32 weights in k' order.
// MFMA k-step s of a tile contracts k' = 4 s + lg (lane group lg): tap (4 s + lg) / 3, channel (s + lg) % 3 -- both have
// period 3 in s, so a lane keeps three LDS addresses (row of its tap + its channel's byte offset: the byte select is part of
// the address) and v_cvt_f32_ubyte0 of the dword it reads there is its operand.  Epilogue and border-class correction: as
// the byte form (brick_epilogue<..., EPI_POOL233, true>).
template <int BKT>  // taps per k-tile
constexpr int tap_waves_per_simd() {
  constexpr int ring = 2 * (BKT * 128 + 3 * BKT * 64) * 4, brick = 32 * 132 * 4;
  constexpr int by_lds = 163840 / (ring > brick ? ring : brick);
  return by_lds > 6 ? 6 : by_lds;
}

template <int BKT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(tap_waves_per_simd<BKT>(), 8)))
void stem_u8_tap_kernel(const ConvArgs a) {
  constexpr int BM = 128, BN = 64, FM = 4, FN = 2, BKP = 3 * BKT, STEPS = BKP / 4, LA = BKT / 2, NPB = BKP / 4;
  static_assert(BKT == 8 || BKT == 16, "taps per k-tile");
  static_assert(STEPS % 3 == 0, "the (tap, channel) pattern of the k-steps has period 3");
  constexpr unsigned OOB = 0xFFFFFF00u;
  constexpr int A_FLOATS = BKT * BM, STAGE = A_FLOATS + BKP * BN, RING = 2 * STAGE;
  constexpr int BRICK_FLOATS = (BN / FN) * (BM + 4);
  constexpr int SMEM = RING > BRICK_FLOATS ? RING : BRICK_FLOATS;
  constexpr int BRICK_H = 4, BRICK_W = 16;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile_m = xcd_remap((int)blockIdx.x, a.tiles_m);  // Cout = 64: one n-tile
  const int ml = tid % BM;
  const int kr = __builtin_amdgcn_readfirstlane(tid / BM);
  // m-tile -> (sample, brick t, brick h, brick w); row ml -> position inside the 2 x 4 x 16 brick, w fastest
  const int bk_b = (int)a.dNb.div((unsigned)tile_m);
  const int r1 = tile_m - bk_b * (int)a.dNb.d;
  const int bk_t = (int)a.dNbhw.div((unsigned)r1);
  const int r2 = r1 - bk_t * (int)a.dNbhw.d;
  const int bk_h = (int)a.dNbw.div((unsigned)r2);
  const int bk_w = r2 - bk_h * a.nbw;
  const int pot = bk_t * 2 + ml / (BRICK_H * BRICK_W), poh = bk_h * BRICK_H + (ml / BRICK_W) % BRICK_H, pow_ = bk_w * BRICK_W + ml % BRICK_W;
  const int bg = a.u8_first + bk_b, clip = bg / 10, crop = bg - clip * 10, j5 = crop >= 5 ? crop - 5 : crop;
  const int flip = __builtin_amdgcn_readfirstlane(crop >= 5);
  unsigned vbase = OOB, vmask = 0;
  if (pot < a.To && poh < a.Ho && pow_ < a.Wo) {
    const int it0 = pot * a.st - a.pt, ih0 = poh * a.sh - a.ph, iw0 = pow_ * a.sw - a.pw;
    const int top = j5 == 4 ? a.u8_ctop : ((j5 >> 1) ? a.u8_FH - a.H : 0), left = j5 == 4 ? a.u8_cleft : ((j5 & 1) ? a.u8_FW - a.W : 0);
    const int col = flip ? a.u8_FW - 1 - left - iw0 - (a.kw_ - 1) : left + iw0;  // (mirrored crops: see the byte form)
    vbase = (unsigned)((((clip * a.T + it0) * a.u8_FH + top + ih0) * a.u8_FW + col) * 3 + a.pad_off);
    vmask = tap_bits(it0, a.kt_, a.T) | (tap_bits(ih0, a.kh_, a.H) << 10) | (tap_bits(iw0, a.kw_, a.W) << 20);
  }
  U8Corr u8c{};
  {
    const int ot = bk_t * 2 + (wave >> 1), oh = bk_h * BRICK_H + (lane >> 4);
    const int tc = u8_border_class(ot * a.st - a.pt, a.kt_, a.T, a.pt), hc = u8_border_class(oh * a.sh - a.ph, a.kh_, a.H, a.ph);
    const int ow0 = bk_w * BRICK_W, owl = ow0 + BRICK_W - 1 < a.Wo ? ow0 + BRICK_W - 1 : a.Wo - 1;
    const int wc0 = u8_border_class(ow0 * a.sw - a.pw, a.kw_, a.W, a.pw), wc1 = u8_border_class(owl * a.sw - a.pw, a.kw_, a.W, a.pw);
    const int nch = (a.ph + 1) * (a.ph + 1), ncw = (a.pw + 1) * (a.pw + 1);
    u8c.row = a.pad_corr + (size_t)((tc * nch + hc) * ncw) * a.Cout + 2 * ((wave & 1) * 16 + (lane & 15));
    u8c.uniform = wc0 == wc1;
    u8c.c[0] = u8c.row[(size_t)wc0 * a.Cout];
    u8c.c[1] = u8c.row[(size_t)wc0 * a.Cout + 1];
  }
  const auto rx = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(const_cast<float*>(a.x)) - a.pad_off, 0, a.x_bytes, 0x00020000);
  const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
  const int ntaps_pad = a.Kpad / 3;
  const int2* __restrict__ tab = a.ktab_u8 + (size_t)(flip ? ntaps_pad : 0);
  const unsigned wvoff = (unsigned)((lane / 16) * BN + (lane % 16) * 4) * 4u;  // B piece: 4 k'-rows x 64 channels = 1 KiB
  const int a_wave_col = (wave & 1) * 64;
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int DMA_BYTES = 16;  // (the host pass only parses this body; see gemm_kk_dma_kernel)
#else
  constexpr int DMA_BYTES = 4;
#endif
  int ent[2 * LA];
  auto issue_tile = [&](int t0, int stage) {  // taps [t0, t0 + BKT) and their 3 * BKT weight rows
    float* As = smem + stage * STAGE;
    float* Bs = As + A_FLOATS;
#pragma unroll
    for (int j = 0; j < LA; ++j) {
      const unsigned voff = ((vmask & (unsigned)ent[2 * j + 1]) == (unsigned)ent[2 * j + 1]) ? vbase : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(As + (kr * LA + j) * BM + a_wave_col), 4, voff, ent[2 * j], 0, 0);
    }
#pragma unroll
    for (int g = 0; g < (NPB + 3) / 4; ++g) {
      const int piece = wave + 4 * g;
      if (piece < NPB)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(Bs + piece * 4 * BN), DMA_BYTES, wvoff, (3 * t0 + 4 * piece) * BN * 4, 0, 0);
    }
  };

  const int wm = wave >> 1, wn = wave & 1, li = lane & 15, lg = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
  // k-step s = 3 q + t of a tile: this lane's operand is byte (t + lg) % 3 of the pixel dwords of tap row 4 q + (4 t + lg) / 3
  unsigned a_addr[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) a_addr[t] = lds0 + (unsigned)(((4 * t + lg) / 3) * BM + wm * 64 + FM * li) * 4u + (unsigned)((t + lg) % 3);
  const unsigned b_addr0 = lds0 + (unsigned)(A_FLOATS + lg * BN + wn * 32 + FN * li) * 4u;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int jn = 0; jn < FN; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](bool pre, int stage, int pre_t0, int pstage) {
    const unsigned so = (unsigned)(stage * STAGE) * 4u;
    Frag<FM> fa[2];
    Frag<FN> fb[2];
    if (pre) issue_tile(pre_t0, pstage);
    __builtin_amdgcn_sched_barrier(0);
    lds_read<FM, 0>(fa[0], a_addr[0] + so);
    lds_read<FN, 0>(fb[0], b_addr0 + so);
    auto body = [&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
      if constexpr (s + 1 < STEPS) {
        lds_read<FM, ((s + 1) / 3) * 4 * BM * 4>(fa[(s + 1) & 1], a_addr[(s + 1) % 3] + so);
        lds_read<FN, (s + 1) * 4 * BN * 4>(fb[(s + 1) & 1], b_addr0 + so);
        lds_wait<2>(fa[s & 1], fb[s & 1]);
      } else {
        lds_wait<0>(fa[s & 1], fb[s & 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const float e = fa[s & 1].v[i];
        const float ai = u8_to_f32(e);
#pragma unroll
        for (int jn = 0; jn < FN; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, fb[s & 1].v[jn], acc[i][jn], 0, 0, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x2, 1, 0);  // operand 0's conversion
#pragma unroll
      for (int i = 1; i < FM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);  // one MFMA, the next operand's conversion beside it
        __builtin_amdgcn_sched_group_barrier(0x2, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x8, FM * FN - (FM - 1), 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    [&]<int... I>(std::integer_sequence<int, I...>) { (body(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, STEPS>{});
  };

  const int nt = ntaps_pad / BKT;
  sload_entries<LA>(tab, (kr * LA) * 8, ent);
  issue_tile(0, 0);
  for (int kt = 0; kt < nt; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    const bool pre = kt + 1 < nt;
    if (pre) sload_entries<LA>(tab, ((kt + 1) * BKT + kr * LA) * 8, ent);
    compute(pre, kt & 1, (kt + 1) * BKT, (kt + 1) & 1);
  }
  __syncthreads();
  brick_epilogue<BM, BN, 16, EPI_POOL233, true>(a, acc, smem, tile_m, 0, 0, bk_b, bk_t, bk_h, bk_w, wave, lane, tid, u8c);
}

// ================================================================================================
// C[s][m][n] = sum_{k in slice s} A[m][k] * B[n][k]: both operands k-contiguous ("NT" product), e.g. the weight gradient
// dW[o][c] = sum_n dY[o][n] X[c][n] of a GEMM-shaped layer whose activations are stored (channel, position).  Operand
// tiles are plain row copies [rows][16 k] made by 16-byte LDS-DMA (4 lanes per 64-byte row piece, 16 rows per
// wave-instruction, LDS image linear in the lane index), 2-deep ring, one barrier per k-tile.  One ds_read_b128 per
// lane along k gives the operands of FOUR MFMAs: with lane = (row li, k-chunk lg) holding k = 4 lg + e in element e,
// MFMA e contracts over the k set {e, 4+e, 8+e, 12+e} -- the same set for A and B, so the sum over e covers the k-tile.
// 64-byte rows make those reads conflict-free (the image is linear in (li, lg)).  Fragment i of a wave = rows 16 i .. 16 i + 15.
struct GemmKKArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;          // K % 16 == 0
  int lda, ldb, ldc;    // row pitches in elements
  unsigned a_bytes, b_bytes;
  int tiles_m, tiles_n, splits;
  long long slab;       // elements between the outputs of consecutive K slices
  float* rowsum;        // nullable [splits][M]: sum over the K slice of A[m][k] (the bias gradient beside dW = dY X^T)
  long long rs_slab;    // elements between the row sums of consecutive K slices (M when they are a matrix of their own)
  // in-kernel reduction of the K slices (nullable): every (tile, slice) workgroup publishes its partial tile (and row sums),
  // draws a ticket on the tile's arrival counter, and the last arriver sums the slices in slice order and writes C / rowsum
  // (slab 0).  Counters are zero on entry and are left zero (the last arriver resets its tile's counter).
  float* part;          // [tiles][splits][BM*BN] fragment-major, then [tiles_m][splits][BM] row-sum partials
  unsigned* cnt;        // [tiles]
  unsigned part_bytes;
};

// NS: ring depth; 2 (one k-tile in flight) is what every launch uses.  A 6-deep ring for grids of one or two workgroups per CU
// (the small layers' weight gradients: few output tiles, K = 10 240) measured no gain -- those launches are bound by the
// latency chain of their slice sums and launch gaps, not by the ring (profiles/r03_gemm_nt_tune.txt).
template <int BM, int BN, int NS = 2>
__device__ __forceinline__ void gemm_kk_body(const GemmKKArgs& g, const int block) {
  constexpr int BK = 16, FM = BM / 32, FN = BN / 32;
  static_assert(NS >= 2 && NS <= 8, "ring depth");
  constexpr int STAGE = (BM + BN) * BK;  // floats
  constexpr int LA = BM * 4 / 256, LB = BN * 4 / 256;  // 16-byte LDS-DMA instructions per wave per k-tile
  static_assert(LA >= 1 && LB >= 1, "tile too small");
  // the 16-byte LDS-DMA form exists on gfx950 only: hipcc's HOST pass (which merely parses this body) silently drops the
  // kernel's launch stub when it meets it in a template-dependent statement, so the host pass sees the 4-byte form
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int DMA_BYTES = 16;
#else
  constexpr int DMA_BYTES = 4;
#endif
  __shared__ __attribute__((aligned(16))) float smem[NS * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = g.tiles_m * g.tiles_n;
  const int it = xcd_remap(block, ntiles * g.splits);
  const int L = it / g.splits, split = it - L * g.splits;
  const int tile_m = L / g.tiles_n, tile_n = L - tile_m * g.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, g.a_bytes, 0x00020000);
  const auto rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, g.b_bytes, 0x00020000);
  // lane -> (row, 16-byte chunk) of the piece its wave copies: rows beyond M / N fall outside the buffer range -> zeros
  unsigned avoff[LA], bvoff[LB];
#pragma unroll
  for (int j = 0; j < LA; ++j) {
    const int row = (wave * LA + j) * 16 + (lane >> 2);
    avoff[j] = (m0 + row < g.M) ? (unsigned)(((m0 + row) * g.lda + (lane & 3) * 4) * 4) : 0xFFFFFF00u;
  }
#pragma unroll
  for (int j = 0; j < LB; ++j) {
    const int row = (wave * LB + j) * 16 + (lane >> 2);
    bvoff[j] = (n0 + row < g.N) ? (unsigned)(((n0 + row) * g.ldb + (lane & 3) * 4) * 4) : 0xFFFFFF00u;
  }
  auto issue = [&](int kt, int stage) {
    float* As = smem + stage * STAGE;
    float* Bs = As + BM * BK;
#pragma unroll
    for (int j = 0; j < LA; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(As + (wave * LA + j) * 16 * BK), DMA_BYTES, avoff[j], kt * BK * 4, 0, 0);
#pragma unroll
    for (int j = 0; j < LB; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(Bs + (wave * LB + j) * 16 * BK), DMA_BYTES, bvoff[j], kt * BK * 4, 0, 0);
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
  const unsigned a_addr0 = lds0 + (unsigned)(((wm * (BM / 2) + li) * BK + 4 * lg) * 4);
  const unsigned b_addr0 = lds0 + (unsigned)((BM * BK + (wn * (BN / 2) + li) * BK + 4 * lg) * 4);

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // row sums of A (db = sum_n dY[o][n] beside dW): the n-tile-0 workgroup of every m-tile already has each A row in its
  // fragments -- lane (li, lg) holds k = 4 lg .. 4 lg + 3 of row 16 i + li -- so the waves of its first wave column add them up
  // (FM x 4 VALU adds per k-tile beside FM x FN x 4 MFMAs) and fold the four lane groups at the end.
  const bool do_rs = g.rowsum != nullptr && tile_n == 0 && wn == 0;  // (wave-uniform)
  float rs[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) rs[i] = 0.f;

  const int nk_all = g.K / BK;
  const int kt0 = (nk_all * split) / g.splits, kt1 = (nk_all * (split + 1)) / g.splits;
#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (kt0 + i < kt1) issue(kt0 + i, i);
  int stage = 0;
  for (int kt = kt0; kt < kt1; ++kt) {
    // tile kt has landed once only the loads of the younger tiles in flight (<= NS - 2) are outstanding
    if constexpr (NS == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      const int younger = kt1 - 1 - kt < NS - 2 ? kt1 - 1 - kt : NS - 2;
      [&]<int... Y>(std::integer_sequence<int, Y...>) {
        ((younger == Y ? (void)({ asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Y * (LA + LB)) : "memory"); }) : (void)0), ...);
      }(std::make_integer_sequence<int, NS - 1>{});
    }
    asm volatile("s_barrier" ::: "memory");  // tile kt is in LDS for every wave; the stage read in the previous iteration is free
    if (kt + NS - 1 < kt1) issue(kt + NS - 1, stage == 0 ? NS - 1 : stage - 1);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned aa = a_addr0 + (unsigned)(stage * STAGE * 4), ba = b_addr0 + (unsigned)(stage * STAGE * 4);
    Frag<4> fa[FM], fb[FN];
    [&]<int... I>(std::integer_sequence<int, I...>) { (lds_read<4, I * 16 * BK * 4>(fa[I], aa), ...); }(std::make_integer_sequence<int, FM>{});
    [&]<int... J>(std::integer_sequence<int, J...>) { (lds_read<4, J * 16 * BK * 4>(fb[J], ba), ...); }(std::make_integer_sequence<int, FN>{});
    if constexpr (FM == 2 && FN == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0].v), "+v"(fa[1].v), "+v"(fb[0].v), "+v"(fb[1].v));
    else if constexpr (FM == 4 && FN == 2)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0].v), "+v"(fa[1].v), "+v"(fa[2].v), "+v"(fa[3].v), "+v"(fb[0].v), "+v"(fb[1].v));
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0].v), "+v"(fa[1].v), "+v"(fa[2].v), "+v"(fa[3].v), "+v"(fb[0].v), "+v"(fb[1].v), "+v"(fb[2].v), "+v"(fb[3].v));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].v[e], fb[j].v[e], acc[i][j], 0, 0, 0);
    if (do_rs) {
#pragma unroll
      for (int i = 0; i < FM; ++i) rs[i] += (fa[i].v[0] + fa[i].v[1]) + (fa[i].v[2] + fa[i].v[3]);
    }
    __builtin_amdgcn_sched_barrier(0);
    stage = stage == NS - 1 ? 0 : stage + 1;
  }
  if (do_rs) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      rs[i] += __shfl_xor(rs[i], 16);
      rs[i] += __shfl_xor(rs[i], 32);
    }
  }
  int out_split = split;
  if (g.part != nullptr && g.splits > 1) {
    // ---- in-kernel reduction of the K slices (the conv kernel's write-through form: cdna_hip_programming.md section 5)
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    const auto rp = __builtin_amdgcn_make_buffer_rsrc(g.part, 0, g.part_bytes, 0x00020000);
    constexpr unsigned TILE_BYTES = BM * BN * 4;
    const unsigned my = (unsigned)(L * g.splits + split) * TILE_BYTES + (unsigned)tid * 16u;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rp, my + (unsigned)(i * FN + j) * 4096u, 0, 16);
    const unsigned rs_base = (unsigned)ntiles * (unsigned)g.splits * TILE_BYTES + (unsigned)(tile_m * g.splits) * (unsigned)(BM * 4);
    if (do_rs && lg == 0) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rs[i]), rp,
                                              rs_base + (unsigned)(split * BM + wm * (BM / 2) + 16 * i + li) * 4u, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores ...
    __syncthreads();                                   // ... before one lane signals for all of them
    unsigned* flag = reinterpret_cast<unsigned*>(smem);
    if (tid == 0) *flag = __hip_atomic_fetch_add(g.cnt + L, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool last = *flag == (unsigned)(g.splits - 1);
    if (!last) return;
    if (tid == 0) __hip_atomic_store(g.cnt + L, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // every slice has arrived: ready for the next launch
    const unsigned t0 = (unsigned)(L * g.splits) * TILE_BYTES + (unsigned)tid * 16u;
    // G float4 of SU slices are loaded before any is added (the adds stay in slice order: deterministic): the last arriver's
    // sum is a chain of dependent round trips otherwise -- 40 of them for a small output cut into 40 slices
    constexpr int NV = FM * FN, G = NV < 4 ? NV : 4, SU = NV >= 16 ? 2 : 4;
    auto at = [&](int v) -> f32x4& { return acc[v / FN][v % FN]; };
#pragma unroll
    for (int g0 = 0; g0 < NV; g0 += G) {
#pragma unroll
      for (int v = 0; v < G; ++v)
        at(g0 + v) = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, t0 + (unsigned)(g0 + v) * 4096u, 0, 16));
#pragma unroll 1
      for (int sl = 1; sl < g.splits; sl += SU) {
        f32x4 t[SU][G];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          // (slices past the end: re-read the last one and add zero times it -- keeps the loop body branch-free)
          const int ss = sl + u < g.splits ? sl + u : g.splits - 1;
          const unsigned ts = t0 + (unsigned)ss * TILE_BYTES;
#pragma unroll
          for (int v = 0; v < G; ++v)
            t[u][v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, ts + (unsigned)(g0 + v) * 4096u, 0, 16));
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          if (sl + u < g.splits) {
#pragma unroll
            for (int v = 0; v < G; ++v) at(g0 + v) += t[u][v];
          }
        }
      }
    }
    if (do_rs && lg == 0) {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const unsigned o = rs_base + (unsigned)(wm * (BM / 2) + 16 * i + li) * 4u;
        float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, o, 0, 16));
#pragma unroll 1
        for (int sl = 1; sl < g.splits; ++sl)
          v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, o + (unsigned)(sl * BM) * 4u, 0, 16));
        rs[i] = v;
      }
    }
    out_split = 0;
  }
  if (do_rs && lg == 0) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = m0 + wm * (BM / 2) + 16 * i + li;
      if (m < g.M) g.rowsum[(long long)out_split * g.rs_slab + m] = rs[i];
    }
  }
  // D[row = 4 lg + r][col = li] of fragment (i, j): m = m0 + wm*BM/2 + 16 i + 4 lg + r, n = n0 + wn*BN/2 + 16 j + li
  float* __restrict__ C = g.C + (long long)out_split * g.slab;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * (BM / 2) + 16 * i + 4 * lg + r;
      if (m >= g.M) continue;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn * (BN / 2) + 16 * j + li;
        if (n < g.N) C[(long long)m * g.ldc + n] = acc[i][j][r];
      }
    }
}

template <int BM, int BN, int NS = 2>
__global__ __launch_bounds__(256) void gemm_kk_dma_kernel(const GemmKKArgs g) {
  gemm_kk_body<BM, BN, NS>(g, (int)blockIdx.x);
}

// Several small NT products (K slices to slabs) in ONE launch: the weight gradients of the 64- and 128-channel layers are a
// handful of 64 x 64 tiles each, ~12 us of latency chain per launch and a launch gap on either side -- 31 of them per MGFN
// training step.  Items travel in the kernel arguments (no table in device memory: the launch is graph-capturable as is);
// workgroup b works on the item whose range [wg_begin, next wg_begin) holds it, exactly as gemm_kk_dma_kernel<64, 64> would.
constexpr int NT_GROUP_MAX = 32;
struct NtGroupItem {
  const float* A;
  const float* B;
  float* C;
  float* rowsum;  // nullable
  int M, N, lda, ldb;
  unsigned a_bytes, b_bytes;
  int tiles_n, wg_begin;
};
struct NtGroupArgs {
  NtGroupItem it[NT_GROUP_MAX];
  int n, K, splits, pad;
  long long slab;  // elements between the outputs (and row sums) of consecutive K slices: one slab matrix for all items
};

__global__ __launch_bounds__(256) void gemm_kk_group_kernel(const NtGroupArgs ga) {
  int i = 0;
  while (i + 1 < ga.n && (int)blockIdx.x >= ga.it[i + 1].wg_begin) ++i;  // (uniform: scalar loads from the kernel arguments)
  const NtGroupItem& t = ga.it[i];
  GemmKKArgs g;
  g.A = t.A; g.B = t.B; g.C = t.C;
  g.M = t.M; g.N = t.N; g.K = ga.K;
  g.lda = t.lda; g.ldb = t.ldb; g.ldc = t.N;
  g.a_bytes = t.a_bytes; g.b_bytes = t.b_bytes;
  g.tiles_m = (t.M + 63) / 64; g.tiles_n = t.tiles_n; g.splits = ga.splits;
  g.slab = ga.slab;
  g.rowsum = t.rowsum; g.rs_slab = ga.slab;
  g.part = nullptr; g.cnt = nullptr; g.part_bytes = 0;
  gemm_kk_body<64, 64, 2>(g, (int)blockIdx.x - t.wg_begin);
}

// ================================================================================================
// Split-bf16 variant ("bf16x3"): every fp32 operand is split into two bf16 terms, x = hi + lo (+ <= 2^-17 |x|), and
// the contraction runs as three bf16 MFMAs per fragment -- hi*hi + hi*lo + lo*hi -- accumulated in fp32
// (v_mfma_f32_16x16x32_bf16: 8192 MACs in 16 cycles, 16x the fp32 MFMA rate, so 3 of them cost 3/16 of the fp32
// instruction stream).  The dropped lo*lo term and the split residue bound the relative error of a product by ~2^-16;
// results agree with the fp32 kernels to ~1e-5 (tests), inside the 1e-3 contract but not bit-comparable -- opt-in.
//   tile 128 x BN x 32, 256 threads = 2x2 waves of 64 x BN/2 (the fp32 kernels' accumulator mapping, same epilogue);
//   A: fp32 NCDHW gather as in the fast kernel (raw buffer loads, scalar tap offsets, coordinate mask), 16 consecutive
//      k per thread, split in registers (v_cvt_pk_bf16_f32), written as [row][32 k] bf16 images (hi, lo), 80-byte rows
//      (conflict-free ds_read_b128 of 16 rows x 16 B and ds_write_b128 of consecutive rows);
//   B: weights pre-split on the device once ([2][Cout][Kpad] bf16, k contiguous), staged the same way;
//   rows of an image are stored fragment-major (row position jm*16 + i holds m = FM*i + jm), so one ds_read_b128 per
//   lane is a whole MFMA operand (8 k of one row).
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2v = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
  const f32x2v v = {x0, x1};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  const f32x2v r = {x0 - __builtin_bit_cast(float, hi << 16), x1 - __builtin_bit_cast(float, hi & 0xFFFF0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}

template <int BN, bool CHECK>
__global__ __launch_bounds__(256) void conv3d_igemm_bf16x3_kernel(const ConvArgs a) {
  constexpr int BM = 128, BK = 32;
  using Cfg = IgemmCfg<BM, BN, BK>;
  constexpr int FM = Cfg::FM, FN = Cfg::FN;
  static_assert(FM == 4 && (FN == 4 || FN == 2), "wave tile 64 x 64 or 64 x 32");
  constexpr unsigned OOB = 0xFFFFFF00u;
  constexpr int PITCH = 80;                       // bytes per image row: 64 of data + 16 of padding
  constexpr int A_IMG = BM * PITCH, B_IMG = BN * PITCH;
  constexpr int TILE_BYTES = 2 * A_IMG + 2 * B_IMG;  // A hi, A lo, B hi, B lo
  constexpr int SMEM_BYTES = TILE_BYTES > Cfg::ST_FLOATS * 4 ? TILE_BYTES : Cfg::ST_FLOATS * 4;
  constexpr int NBC = BN * 4 / 256;               // 16-byte chunks of one B image per thread

  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
  unsigned char* const Ah = smem;
  unsigned char* const Al = smem + A_IMG;
  unsigned char* const Bh = smem + 2 * A_IMG;
  unsigned char* const Bl = smem + 2 * A_IMG + B_IMG;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = a.tiles_m * a.tiles_n;
  const int it = xcd_remap((int)blockIdx.x, ntiles * a.splits);
  int L, split;
  if (a.splits > 1) {
    L = (int)a.dSplits.div((unsigned)it);
    split = it - L * a.splits;
  } else {
    split = 0;
    L = it;
  }
  const int tile_m = (int)a.dTilesN.div((unsigned)L), tile_n = L - tile_m * a.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- A: this thread fills image row p (fragment-major position) with k-half khalf (16 consecutive k)
  const int p = tid & 127;
  const int khalf = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int m = m0 + (p & 64) + FM * (p & 15) + ((p & 63) >> 4);
  unsigned vbase = OOB;
  unsigned vmask = 0;
  if (m < a.M) {
    const int b = (int)a.dTHWo.div((unsigned)m);
    const int pp = m - b * a.THWo;
    const int ot = (int)a.dHWo.div((unsigned)pp);
    const int q = pp - ot * a.HWo;
    const int oh = (int)a.dWo.div((unsigned)q);
    const int ow = q - oh * a.Wo;
    const int it0 = ot * a.st - a.pt, ih0 = oh * a.sh - a.ph, iw0 = ow * a.sw - a.pw;
    vbase = (unsigned)(b * a.x_bstride + it0 * a.HW + ih0 * a.W + iw0 + a.pad_off) * 4u;
    if constexpr (CHECK) vmask = tap_bits(it0, a.kt_, a.T) | (tap_bits(ih0, a.kh_, a.H) << 10) | (tap_bits(iw0, a.kw_, a.W) << 20);
  }
  const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) - a.pad_off, 0, a.x_bytes, 0x00020000);
  const int2* __restrict__ ktab2 = reinterpret_cast<const int2*>(a.ktab + a.Kpad);
  // ---- B: chunk c of this thread = 16 bytes (8 k) of image row (tid + 256 c) / 4
  const unsigned short* __restrict__ whi = reinterpret_cast<const unsigned short*>(a.w);
  const unsigned short* __restrict__ wlo = whi + (size_t)a.Cout * a.Kpad;
  size_t b_src[NBC];
  int b_dst[NBC];
#pragma unroll
  for (int c = 0; c < NBC; ++c) {
    const int idx = tid + 256 * c;
    const int q = idx >> 2, part = idx & 3;             // image row (position), 16-byte part of its 64 bytes
    constexpr int WN = Cfg::WN;
    const int ql = q % WN;
    const int n = n0 + (q / WN) * WN + FN * (ql & 15) + (ql >> 4);
    b_src[c] = (size_t)n * a.Kpad + part * 8;            // in bf16 elements, + k0 per tile
    b_dst[c] = q * PITCH + part * 16;
  }

  float ra[16];
  uint4 rbh[NBC], rbl[NBC];
  auto load_tile = [&](int k0) {
    int ent[32];
    sload_entries<16>(ktab2, (k0 + khalf * 16) * 8, ent);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      unsigned voff = vbase;
      if constexpr (CHECK) voff = ((vmask & (unsigned)ent[2 * j + 1]) == (unsigned)ent[2 * j + 1]) ? vbase : OOB;
      ra[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, voff, ent[2 * j], 0));
    }
#pragma unroll
    for (int c = 0; c < NBC; ++c) {
      rbh[c] = *reinterpret_cast<const uint4*>(whi + b_src[c] + k0);
      rbl[c] = *reinterpret_cast<const uint4*>(wlo + b_src[c] + k0);
    }
  };
  auto store_tile = [&]() {
    unsigned h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split_pair(ra[2 * j], ra[2 * j + 1], h[j], l[j]);
    const int off = p * PITCH + khalf * 32;
    *reinterpret_cast<uint4*>(Ah + off) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4*>(Ah + off + 16) = make_uint4(h[4], h[5], h[6], h[7]);
    *reinterpret_cast<uint4*>(Al + off) = make_uint4(l[0], l[1], l[2], l[3]);
    *reinterpret_cast<uint4*>(Al + off + 16) = make_uint4(l[4], l[5], l[6], l[7]);
#pragma unroll
    for (int c = 0; c < NBC; ++c) {
      *reinterpret_cast<uint4*>(Bh + b_dst[c]) = rbh[c];
      *reinterpret_cast<uint4*>(Bl + b_dst[c]) = rbl[c];
    }
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  const int a_rd = (wm * Cfg::WM + li) * PITCH + lg * 16;  // + jm * 16 rows
  const int b_rd = (wn * Cfg::WN + li) * PITCH + lg * 16;  // + jn * 16 rows

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk_all = a.Kpad / BK;
  const int kt0 = (nk_all * split) / a.splits;
  const int kt1 = (nk_all * (split + 1)) / a.splits;
  if (kt0 < kt1) {
    load_tile(kt0 * BK);
    store_tile();
  }
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const bool more = kt + 1 < kt1;
    if (more) load_tile((kt + 1) * BK);  // in flight under this tile's MFMAs
    bf16x8 bh[FN], bl[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      bh[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bh + b_rd + j * 16 * PITCH));
      bl[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bl + b_rd + j * 16 * PITCH));
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Ah + a_rd + i * 16 * PITCH));
      const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Al + a_rd + i * 16 * PITCH));
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();  // every wave has read this tile
    if (more) store_tile();
    __syncthreads();
  }
  igemm_epilogue<BM, BN, BK>(a, acc, reinterpret_cast<float*>(smem), split, m0, n0, wave, lane, a.splits == 1);
}

// fp32 weights (Cout, K) (torch layout: K = (ci, dt, dh, dw) contiguous) -> [2][Cout][Kpad] bf16: hi image, lo image
__global__ void pack_weight_bf16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Cout, int K, int Kpad) {
  const long long total = (long long)Cout * Kpad;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int n = (int)(i / Kpad), k = (int)(i - (long long)n * Kpad);
    const float v = k < K ? w[(size_t)n * K + k] : 0.f;
    unsigned h, l;
    split_pair(v, 0.f, h, l);
    out[i] = (unsigned short)(h & 0xFFFFu);
    out[total + i] = (unsigned short)(l & 0xFFFFu);
  }
}

// y = act(sum_s slab[s] * scale[c] + shift[c] (+ res)), slabs in NCDHW like y
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ scale,
                                     const float* __restrict__ shift, const float* __restrict__ res,
                                     float* __restrict__ y, long long total, int THWo, int Cout, int splits, int relu,
                                     int vec4) {
  if (vec4) {
    const long long n4 = total / 4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
      float4 s = reinterpret_cast<const float4*>(ws)[i];
      for (int k = 1; k < splits; ++k) {
        const float4 t = reinterpret_cast<const float4*>(ws + (size_t)k * total)[i];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
      const int c = (int)((i * 4 / THWo) % Cout);
      const float sc = scale[c], sf = shift[c];
      float4 o = make_float4(s.x * sc + sf, s.y * sc + sf, s.z * sc + sf, s.w * sc + sf);
      if (res) {
        const float4 r = reinterpret_cast<const float4*>(res)[i];
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
      if (relu) { o.x = act_apply(relu, o.x); o.y = act_apply(relu, o.y); o.z = act_apply(relu, o.z); o.w = act_apply(relu, o.w); }
      reinterpret_cast<float4*>(y)[i] = o;
    }
  } else {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
      float s = ws[i];
      for (int k = 1; k < splits; ++k) s += ws[(size_t)k * total + i];
      const int c = (int)((i / THWo) % Cout);
      float o = s * scale[c] + shift[c];
      if (res) o += res[i];
      if (relu) o = act_apply(relu, o);
      y[i] = o;
    }
  }
}

// ---- weight packing + gather table -------------------------------------------------------------
// wp[k][co] = w[co][k] (zero rows for K <= k < Kpad): a 32 x 32-tile transpose through LDS, so that both the reads (k
// contiguous) and the writes (co contiguous) are coalesced -- the MGFN training step re-packs every GEMM weight each step
// (4 M elements per stage-2 FFN matrix: 21-25 us for the one-thread-per-element form, whose reads strode by K).
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int K, int Kpad) {
  __shared__ float tile[32][33];
  const int tiles_k = (Kpad + 31) / 32, tiles_c = (Cout + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int t = blockIdx.x; t < tiles_k * tiles_c; t += gridDim.x) {
    const int k0 = (t % tiles_k) * 32, c0 = (t / tiles_k) * 32;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
      const int co = c0 + r, k = k0 + tx;
      tile[r][tx] = (co < Cout && k < K) ? w[(size_t)co * K + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
      const int k = k0 + r, co = c0 + tx;
      if (k < Kpad && co < Cout) wp[(size_t)k * Cout + co] = tile[tx][r];
    }
    __syncthreads();
  }
}

// The packed operand of a Conv1d's TRANSPOSED conv (its input gradient dX = conv1d(dY; W'), W'[c][o][j] = W[o][c][k-1-j]) straight
// from the parameter w (Cout, Cin, k): wp[(o * k + j)][c] = w[o][c][k - 1 - j], zero rows above Cout * k.
__global__ void pack_weight_dx_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int k, int Kpad) {
  const long long total = (long long)Kpad * Cin;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(i / Cin), c = (int)(i % Cin);
    const int o = row / k, j = row - o * k;
    wp[i] = (o < Cout) ? w[((size_t)o * Cin + c) * k + (k - 1 - j)] : 0.f;
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
    // second table for the fast kernel: {byte offset, required coordinate bits}; bit 30 is never
    // set in a thread's mask, so padding rows (and kernels wider than 10) always read as zero
    int2* t2 = reinterpret_cast<int2*>(ktab + Kpad);
    int2 f;
    if (k < K && kt <= 10 && kh <= 10 && kw <= 10) {
      f = make_int2(e.x * 4, (int)((1u << e.y) | (1u << (10 + e.z)) | (1u << (20 + e.w))));
    } else {
      f = make_int2(0, (int)(1u << 30));
    }
    t2[k] = f;
  }
}

// ---- column-parity planes for a stride-2-along-w conv (conv3d_igemm_dma_kernel<..., AMODE = 1>) ---------------------------
// xs[row][par][S2W_PADL + j] = x[row][2 j + par] for j < W / 2, zero in the S2W_PADL columns before and the columns after;
// row = (b, c, t, h), WP = W / 2 + S2W_PAD floats per plane.  One thread per output element.
constexpr int S2W_PADL = 2, S2W_PAD = 4;
__global__ __launch_bounds__(256) void split_w_kernel(const float* __restrict__ x, float* __restrict__ xs, long long rows, int W, int WP) {
  const long long total = rows * 2 * WP;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long rp = i / WP;           // (row, par)
    const int idx = (int)(i - rp * WP);
    const long long row = rp >> 1;
    const int par = (int)(rp & 1), j = idx - S2W_PADL;
    xs[i] = (j >= 0 && 2 * j + par < W) ? x[row * W + 2 * j + par] : 0.f;
  }
}

// {byte offset, (dt, dh) tap bits} per k-row for the parity-plane layout: tap dw of a window whose 4-column group starts at
// output column ow0 reads plane (dw - pw) & 1, columns ow0 + floor((dw - pw) / 2) + S2W_PADL ... + 3
__global__ void build_ktab_s2w_kernel(int2* __restrict__ tab, int kt, int kh, int kw, int pw, int K, int Kpad, int H, int T, int WP) {
  const int taps = kt * kh * kw, rowp = 2 * WP;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < Kpad; k += gridDim.x * blockDim.x) {
    int2 f = make_int2(0, (int)(1u << 30));  // padding rows: never inside
    if (k < K) {
      const int ci = k / taps, tap = k % taps;
      const int dt = tap / (kh * kw), r = tap % (kh * kw);
      const int dh = r / kw, dw = r % kw;
      const int e = dw - pw, par = e & 1, off = (e - par) / 2;  // floor division by 2 (e - par is even)
      f = make_int2((((ci * T + dt) * H + dh) * rowp + par * WP + off + S2W_PADL) * 4, (int)((1u << dt) | (1u << (10 + dh))));
    }
    tab[k] = f;
  }
}

// ---- tables of the uint8-frame stem (conv3d_igemm_dma_kernel<..., U8 = true>) ---------------------------------------------
// {byte offset, tap bits} per k-row for frames stored (F, FH, FW, C): table 0 as stored, table 1 for the mirrored crops
// (tap dw of a mirrored crop is source column -dw; the kernel moves its window origin (kw-1) columns left to keep offsets >= 0)
__global__ void build_ktab_u8_kernel(int2* __restrict__ tab, int kt, int kh, int kw, int C, int K, int Kpad, int FWC, int FHWC) {
  const int taps = kt * kh * kw;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < Kpad; k += gridDim.x * blockDim.x) {
    int2 f = make_int2(0, (int)(1u << 30)), g = f;  // padding rows: never inside
    if (k < K) {
      const int ci = k / taps, tap = k % taps;
      const int dt = tap / (kh * kw), r = tap % (kh * kw);
      const int dh = r / kw, dw = r % kw;
      const int bits = (int)((1u << dt) | (1u << (10 + dh)) | (1u << (20 + dw)));
      f = make_int2(ci + dt * FHWC + dh * FWC + dw * C, bits);
      g = make_int2(ci + dt * FHWC + dh * FWC + (kw - 1 - dw) * C, bits);
    }
    tab[k] = f;
    tab[Kpad + k] = g;
  }
}

// tables of stem_u8_tap_kernel: {byte offset of the tap's pixel, tap bits} for [2][taps_pad] (as stored, mirrored), and the weights
// re-ordered tap-major: wt[(tap * C + c)][n] = wp[(c * taps + tap)][n], zero rows for the padding taps
__global__ void build_ktab_u8_taps_kernel(int2* __restrict__ tab, int kt, int kh, int kw, int C, int taps_pad, int FW, int FHW) {
  const int taps = kt * kh * kw;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < taps_pad; t += gridDim.x * blockDim.x) {
    int2 f = make_int2(0, (int)(1u << 30)), g = f;
    if (t < taps) {
      const int dt = t / (kh * kw), r = t % (kh * kw);
      const int dh = r / kw, dw = r % kw;
      const int bits = (int)((1u << dt) | (1u << (10 + dh)) | (1u << (20 + dw)));
      f = make_int2((dt * FHW + dh * FW + dw) * C, bits);
      g = make_int2((dt * FHW + dh * FW + (kw - 1 - dw)) * C, bits);
    }
    tab[t] = f;
    tab[taps_pad + t] = g;
  }
}

__global__ void pack_weight_taps_kernel(const float* __restrict__ wp, float* __restrict__ wt, int Cout, int C, int taps, int taps_pad) {
  const long long total = (long long)taps_pad * C * Cout;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i % Cout), k = (int)(i / Cout);
    const int t = k / C, c = k % C;
    wt[i] = t < taps ? wp[(size_t)(c * taps + t) * Cout + n] : 0.f;
  }
}

// corr[ct][ch][cw][n] = -mean * sum of w[k][n] over the taps (dt, dh, dw) inside the clip for border classes (ct, ch, cw)
__global__ void u8_pad_corr_kernel(const float* __restrict__ wp, float* __restrict__ corr, int total, int Cin, int kt, int kh, int kw,
                                   int pt, int ph, int pw, int Cout, float mean) {
  const int nch = (ph + 1) * (ph + 1), ncw = (pw + 1) * (pw + 1);
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int n = idx % Cout, c = idx / Cout;
    const int wc = c % ncw, hc = (c / ncw) % nch, tc = c / (ncw * nch);
    const int t0 = tc / (pt + 1), t1 = kt - tc % (pt + 1), h0 = hc / (ph + 1), h1 = kh - hc % (ph + 1), w0 = wc / (pw + 1), w1 = kw - wc % (pw + 1);
    double sum = 0.0;
    for (int ci = 0; ci < Cin; ++ci)
      for (int dt = t0; dt < t1; ++dt)
        for (int dh = h0; dh < h1; ++dh)
          for (int dw = w0; dw < w1; ++dw) sum += (double)wp[(size_t)(((ci * kt + dt) * kh + dh) * kw + dw) * Cout + n];
    corr[idx] = (float)(-(double)mean * sum);
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
  return (K + 31) / 32 * 32;
}

extern "C" int advhip_conv3d_pack_weight_f32(const advhip_conv3d_desc* d, const float* w, float* w_packed,
                                             void* stream) {
  if (int rc = validate(d)) return rc;
  ADVHIP_REQUIRE(w && w_packed, "pack_weight: null pointer");
  const int K = d->Cin * d->kt * d->kh * d->kw;
  const int Kpad = (K + 31) / 32 * 32;
  const long long tiles = (long long)((Kpad + 31) / 32) * ((d->Cout + 31) / 32);
  const int grid = (int)std::min<long long>(tiles, 256 * 16);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, w_packed, d->Cout, K, Kpad);
  return check_launch("pack_weight");
}

extern "C" int advhip_conv1d_pack_weight_dx_f32(const float* w, float* w_packed, int32_t Cout, int32_t Cin, int32_t k, void* stream) {
  ADVHIP_REQUIRE(w && w_packed && Cout > 0 && Cin > 0 && k > 0 && k <= 10, "pack_weight_dx: bad arguments");
  const int Kpad = (Cout * k + 31) / 32 * 32;
  const long long total = (long long)Kpad * Cin;
  const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(pack_weight_dx_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, w_packed, Cout, Cin, k, Kpad);
  return check_launch("pack_weight_dx");
}

extern "C" int advhip_conv3d_pack_weight_bf16x3(const advhip_conv3d_desc* d, const float* w, void* w_split, void* stream) {
  if (int rc = validate(d)) return rc;
  ADVHIP_REQUIRE(w && w_split, "pack_weight_bf16x3: null pointer");
  const int K = d->Cin * d->kt * d->kh * d->kw;
  const int Kpad = (K + 31) / 32 * 32;
  const long long total = (long long)Kpad * d->Cout;
  const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(pack_weight_bf16x3_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w,
                     reinterpret_cast<unsigned short*>(w_split), d->Cout, K, Kpad);
  return check_launch("pack_weight_bf16x3");
}

extern "C" int advhip_conv3d_build_ktab(const advhip_conv3d_desc* d, int32_t* ktab, void* stream) {
  if (int rc = validate(d)) return rc;
  ADVHIP_REQUIRE(ktab, "build_ktab: null pointer");
  const int K = d->Cin * d->kt * d->kh * d->kw;
  const int Kpad = (K + 31) / 32 * 32;
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
struct Choice {
  int algo;    // ADVHIP_ALGO_IGEMM_*
  int splits;  // >= 1
};

// persistent family: id = ADVHIP_ALGO_PERSIST_BASE + tile (2: 128 x 64, 3: 64 x 64) + 8 * (workgroups per CU - 1)
static bool is_persist(int algo) { return algo >= ADVHIP_ALGO_PERSIST_BASE && algo < ADVHIP_ALGO_PERSIST_BASE + 32; }
static int persist_tile(int algo) { return (algo - ADVHIP_ALGO_PERSIST_BASE) & 7; }
static int persist_wgs_per_cu(int algo) { return ((algo - ADVHIP_ALGO_PERSIST_BASE) >> 3) + 1; }

static void tile_of(int algo, int* BM, int* BN, int* BK) {
  if (algo == ADVHIP_ALGO_TSPAN_128x64 || algo == ADVHIP_ALGO_MIXED_128x64) { *BM = 128; *BN = 64; *BK = 16; return; }
  if (is_persist(algo)) { *BM = persist_tile(algo) == ADVHIP_ALGO_IGEMM_128x64 ? 128 : 64; *BN = 64; *BK = 16; return; }
  if (algo == ADVHIP_ALGO_DMA2_BASE + ADVHIP_ALGO_IGEMM_256x64) { *BM = 256; *BN = 64; *BK = 16; return; }
  if (algo >= ADVHIP_ALGO_DMA2_BASE) algo -= ADVHIP_ALGO_DMA2_BASE;
  if (algo >= ADVHIP_ALGO_BF16X3_BASE) algo -= ADVHIP_ALGO_BF16X3_BASE;
  if (algo >= ADVHIP_ALGO_DMA4_BASE) algo -= ADVHIP_ALGO_DMA4_BASE;
  if (algo >= ADVHIP_ALGO_DMA_BASE) algo -= ADVHIP_ALGO_DMA_BASE;
  if (algo >= ADVHIP_ALGO_FAST_BASE) algo -= ADVHIP_ALGO_FAST_BASE;
  const int t = (algo - 1) & 3;
  *BM = (t == 0 || t == 1) ? 128 : 64;
  *BN = (t == 0 || t == 3) ? 128 : 64;
  *BK = algo >= ADVHIP_ALGO_IGEMM_128x128x32 ? 32 : 16;
}

// Which (family, tile) ids have a kernel in this library.  Ids inside a family's range without an instantiation
// (e.g. DMA_BASE + 5) are rejected up front: a launch switch that fell through would return OK with y unwritten.
static bool instantiated(int algo) {
  if (algo == ADVHIP_ALGO_TSPAN_128x64 || algo == ADVHIP_ALGO_MIXED_128x64) return true;
  if (algo >= ADVHIP_ALGO_PERSIST_BASE) return is_persist(algo) && (persist_tile(algo) == ADVHIP_ALGO_IGEMM_128x64 || persist_tile(algo) == ADVHIP_ALGO_IGEMM_64x64) && persist_wgs_per_cu(algo) <= 3;
  auto tile_in = [](int t, unsigned mask) { return t >= 1 && t <= 8 && ((mask >> t) & 1u); };
  constexpr unsigned ALL = 0x1FEu, NO5 = ALL & ~(1u << 5);
  if (algo == ADVHIP_ALGO_DMA2_BASE + ADVHIP_ALGO_IGEMM_256x64) return true;
  if (algo >= ADVHIP_ALGO_DMA2_BASE) return tile_in(algo - ADVHIP_ALGO_DMA2_BASE, NO5);
  if (algo >= ADVHIP_ALGO_BF16X3_BASE) return tile_in(algo - ADVHIP_ALGO_BF16X3_BASE, (1u << 5) | (1u << 6));
  if (algo >= ADVHIP_ALGO_DMA4_BASE) return tile_in(algo - ADVHIP_ALGO_DMA4_BASE, (1u << 2) | (1u << 3) | (1u << 4));
  if (algo >= ADVHIP_ALGO_DMA_BASE) return tile_in(algo - ADVHIP_ALGO_DMA_BASE, NO5);
  if (algo >= ADVHIP_ALGO_FAST_BASE) return tile_in(algo - ADVHIP_ALGO_FAST_BASE, ALL);
  return tile_in(algo, ALL);
}

// the fast kernel covers kernels up to 10x10x10 on tensors addressable by 32-bit byte offsets
static bool fast_ok(const advhip_conv3d_desc* d, long long in_elems, long long w_elems) {
  const long long pad_off = (long long)d->pt * d->H * d->W + (long long)d->ph * d->W + d->pw;
  return d->kt <= 10 && d->kh <= 10 && d->kw <= 10 && (in_elems + pad_off) * 4 < 0xF0000000ll && w_elems * 4 < 0xF0000000ll;
}

// Heuristic used when the caller does not pin algo/splits (the Python engine normally pins both
// from a measured table).  Goal: >= ~3 workgroups per CU; big tiles when M*N is plentiful, split-K
// when it is not and K is long.
static Choice choose(const advhip_conv3d_desc* d, long long M, int Kpad) {
  Choice c{d->algo, d->splits};
  const int N = d->Cout;
  if (c.algo == ADVHIP_ALGO_AUTO) {
    // what the measured table (tuned/gfx950.json) picks almost everywhere: 64x64x16 tiles -- many
    // small workgroups (6 per CU) hide each other's prologue/epilogue -- on the LDS-DMA kernel
    const long long in_elems = (long long)d->B * d->Cin * d->T * d->H * d->W;
    c.algo = ADVHIP_ALGO_IGEMM_64x64;
    if (fast_ok(d, in_elems, (long long)Kpad * N)) c.algo += ADVHIP_ALGO_DMA_BASE;
  }
  if (c.splits <= 0) {
    int BM, BN, BK;
    tile_of(c.algo, &BM, &BN, &BK);
    const long long tiles = ((M + BM - 1) / BM) * (N / BN);
    const int nk = Kpad / BK;
    int s = 1;
    while (tiles * s < 1024 && s < 9 && nk / (s + 1) >= 8) ++s;
    c.splits = s;
  }
  return c;
}

struct Geometry {
  int To, Ho, Wo;
  long long M;
  int K, Kpad;
};
static Geometry geometry(const advhip_conv3d_desc* d) {
  Geometry g;
  g.To = out_dim(d->T, d->kt, d->st, d->pt);
  g.Ho = out_dim(d->H, d->kh, d->sh, d->ph);
  g.Wo = out_dim(d->W, d->kw, d->sw, d->pw);
  g.M = (long long)d->B * g.To * g.Ho * g.Wo;
  g.K = d->Cin * d->kt * d->kh * d->kw;
  g.Kpad = (g.K + 31) / 32 * 32;
  return g;
}
}  // namespace advhip

namespace advhip {
// the LDS-DMA kernel families reduce split-K partials inside the launch (no second kernel)
static bool reduces_in_kernel(int algo) {
  return (algo >= ADVHIP_ALGO_DMA_BASE && algo < ADVHIP_ALGO_BF16X3_BASE) || algo >= ADVHIP_ALGO_DMA2_BASE;
}

// compute units of the current device (the persistent kernels size their grid from it)
static int device_cus() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}
// Split-K workspace: two-launch form = [splits][M*Cout] slabs in y's layout; in-kernel form = [tiles] arrival counters
// (padded to 256 bytes) followed by [tiles][splits][BM*BN] fragment-major partial tiles.
struct SplitLayout {
  int64_t cnt_bytes, part_bytes;
  int64_t total() const { return cnt_bytes + part_bytes; }
};
static SplitLayout split_layout(const advhip_conv3d_desc* d, const Geometry& g, const Choice& c) {
  if (c.splits <= 1) return {0, 0};
  if (!reduces_in_kernel(c.algo)) return {0, (int64_t)c.splits * g.M * d->Cout * (int64_t)sizeof(float)};
  int BM, BN, BK;
  tile_of(c.algo, &BM, &BN, &BK);
  const int64_t tiles = ((g.M + BM - 1) / BM) * (d->Cout / BN);
  return {(tiles * 4 + 255) / 256 * 256, tiles * c.splits * BM * BN * (int64_t)sizeof(float)};
}
}  // namespace advhip

namespace advhip {
static void set_bricks(ConvArgs& a, int nbt, int nbh, int nbw);  // defined with the pooling launchers below

// the 2-deep-ring LDS-DMA kernel of a tile: unchecked (1x1x1, no padding), 16-byte gather pieces ((kt,1,1) stride-1 convs on
// planes of a multiple of 4 positions, 16-deep k-tiles), or the 4-byte gather
// ADVHIP_EXTRA_LDS (study builds only, default 0): unused dynamic LDS per workgroup of the 2-deep-ring kernels -- fewer workgroups of
// one launch per CU, so that launches of different lanes share a CU (profiles/r05_studies.md section 5)
#ifndef ADVHIP_EXTRA_LDS
#define ADVHIP_EXTRA_LDS 0
#endif
template <int BM_, int BN_, int BK_>
static void launch_dma2(bool nocheck, bool s16, dim3 grid, hipStream_t st, const ConvArgs& a) {
  constexpr unsigned XL = ADVHIP_EXTRA_LDS;
  if (nocheck) {
    if (a.a16) hipLaunchKernelGGL((conv3d_igemm_dma_kernel<BM_, BN_, BK_, false, 2, EPI_STD, false, 2>), grid, dim3(256), XL, st, a);
    else hipLaunchKernelGGL((conv3d_igemm_dma_kernel<BM_, BN_, BK_, false, 2>), grid, dim3(256), XL, st, a);
    return;
  }
  if constexpr (BK_ == 16) {
    if (s16) {
      hipLaunchKernelGGL((conv3d_igemm_dma_kernel<BM_, BN_, BK_, true, 2, EPI_STD, false, 1>), grid, dim3(256), XL, st, a);
      return;
    }
  }
  hipLaunchKernelGGL((conv3d_igemm_dma_kernel<BM_, BN_, BK_, true, 2>), grid, dim3(256), XL, st, a);
}

}

extern "C" int64_t advhip_conv3d_workspace_bytes(const advhip_conv3d_desc* d) {
  if (validate(d)) return -1;
  const Geometry g = geometry(d);
  const Choice c = choose(d, g.M, g.Kpad);
  if (!instantiated(c.algo)) {
    set_error("conv3d: algo %d is not instantiated in this library", c.algo);
    return -1;
  }
  return split_layout(d, g, c).total();
}

extern "C" int advhip_conv3d_bn_act_f32(const advhip_conv3d_desc* d, const float* x, const float* w_packed,
                                        const int32_t* ktab, const float* scale, const float* shift,
                                        const float* residual, float* y, void* workspace, int64_t workspace_bytes,
                                        void* stream) {
  return advhip_conv3d_bn_act_strided_f32(d, x, 0, w_packed, ktab, scale, shift, residual, y, 0, workspace, workspace_bytes, stream);
}

extern "C" int advhip_conv3d_bn_act_strided_f32(const advhip_conv3d_desc* d, const float* x, int64_t x_batch_stride,
                                                const float* w_packed, const int32_t* ktab, const float* scale,
                                                const float* shift, const float* residual, float* y,
                                                int64_t y_batch_stride, void* workspace, int64_t workspace_bytes,
                                                void* stream) {
  return advhip_conv3d_bn_act_ex_f32(d, x, x_batch_stride, w_packed, ktab, scale, shift, residual, y, y_batch_stride, nullptr,
                                     workspace, workspace_bytes, stream);
}

extern "C" int advhip_conv3d_bn_act_ex_f32(const advhip_conv3d_desc* d, const float* x, int64_t x_batch_stride,
                                           const float* w_packed, const int32_t* ktab, const float* scale, const float* shift,
                                           const float* residual, float* y, int64_t y_batch_stride,
                                           const advhip_conv3d_epilogue* ep, void* workspace, int64_t workspace_bytes,
                                           void* stream) {
  if (int rc = validate(d)) return rc;
  float* y_preact = ep ? ep->y_preact : nullptr;
  const float* dact_z = ep ? ep->dact_z : nullptr;
  float* avg_out = ep ? ep->avgpool_out : nullptr;
  const bool ln = ep && ep->ln_u;
  ADVHIP_REQUIRE(!ep || (!ep->ln_u == !ep->ln_mu && !ep->ln_u == !ep->ln_rs), "conv3d: the LayerNorm fold needs ln_u, ln_mu and ln_rs together");
  ADVHIP_REQUIRE(d->relu >= 0 && d->relu <= ACT_MUL, "conv3d: unknown activation code %d (0 none, 1 ReLU, 2 GELU, 3 GELU + GELU' output, 4 multiplier)", d->relu);
  ADVHIP_REQUIRE(d->relu != ACT_GELU_D || (ep && ep->y_preact), "conv3d: activation code 3 writes GELU'(pre-activation) to the epilogue's second output: y_preact is null");
  ADVHIP_REQUIRE(d->relu != ACT_MUL || (ep && ep->dact_z && residual == nullptr), "conv3d: activation code 4 multiplies by the epilogue's dact_z tensor (no residual): it is null");
  ADVHIP_REQUIRE(x && w_packed && ktab && scale && shift && (y || avg_out), "conv3d: null pointer");
  const Geometry g = geometry(d);
  ConvArgs a;
  a.x = x; a.w = w_packed; a.ktab = reinterpret_cast<const int4*>(ktab);
  a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
  a.y2 = y_preact; a.dact = dact_z;
  a.ln_u = ln ? ep->ln_u : nullptr; a.ln_mu = ln ? ep->ln_mu : nullptr; a.ln_rs = ln ? ep->ln_rs : nullptr;
  a.B = d->B; a.Cin = d->Cin; a.T = d->T; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
  a.st = d->st; a.sh = d->sh; a.sw = d->sw; a.pt = d->pt; a.ph = d->ph; a.pw = d->pw;
  a.To = g.To; a.Ho = g.Ho; a.Wo = g.Wo;
  const long long x_dense = (long long)d->Cin * d->T * d->H * d->W, y_dense = (long long)d->Cout * g.To * g.Ho * g.Wo;
  const long long xbs = x_batch_stride > 0 ? x_batch_stride : x_dense, ybs = y_batch_stride > 0 ? y_batch_stride : y_dense;
  ADVHIP_REQUIRE(xbs >= x_dense && ybs >= y_dense, "conv3d: batch strides (%lld, %lld) smaller than one sample (%lld, %lld)", xbs, ybs, x_dense, y_dense);
  // span of x in elements (the last sample is not padded out to the stride)
  const long long in_elems = (long long)(d->B - 1) * xbs + x_dense;
  const long long M = g.M;
  if (in_elems >= (1ll << 31) || (long long)d->B * ybs >= (1ll << 32) || M * d->Cout >= (1ll << 32) || M >= (1ll << 31)) {
    set_error("conv3d: tensor too large for 32-bit indexing (in=%lld, out=%lld elements)", in_elems, M * d->Cout);
    return ADVHIP_ERANGE;
  }
  a.M = (int)M;
  a.x_bstride = (int)xbs;
  a.y_bstride = (int)ybs;
  a.Kpad = g.Kpad;
  a.HWo = a.Ho * a.Wo; a.THWo = a.To * a.HWo;
  a.HW = d->H * d->W; a.THW = d->T * a.HW;
  a.relu = d->relu;
  a.MP = a.THWo;
  a.dTHWo = FastDiv::make((unsigned)a.THWo);
  a.dHWo = FastDiv::make((unsigned)a.HWo);
  a.dWo = FastDiv::make((unsigned)a.Wo);
  // widest epilogue vector (floats) the output rows, the batch stride and the pointers are all aligned to
  auto aligned_to = [&](int v) {
    const uintptr_t bytes = (uintptr_t)v * sizeof(float);
    return a.THWo % v == 0 && ybs % v == 0 && (uintptr_t)y % bytes == 0 && (uintptr_t)residual % bytes == 0 &&
           (uintptr_t)workspace % bytes == 0 && (uintptr_t)y_preact % bytes == 0 && (uintptr_t)dact_z % bytes == 0 &&
           (uintptr_t)a.ln_mu % bytes == 0 && (uintptr_t)a.ln_rs % bytes == 0;
  };
  a.vw = aligned_to(4) ? 4 : (aligned_to(2) ? 2 : 1);

  Choice c = choose(d, M, g.Kpad);
  if (avg_out != nullptr) {  // conv + global mean in one launch: one sample per 128-row tile of the 2-deep LDS-DMA kernel, unsplit
    ADVHIP_REQUIRE(d->kt == 1 && d->kh == 1 && d->kw == 1 && d->st == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 && d->ph == 0 && d->pw == 0 &&
                       g.K == g.Kpad && a.THWo <= 128 && d->Cout % 64 == 0 && ((uintptr_t)x & 15) == 0 && !ln && y_preact == nullptr && dact_z == nullptr && a.relu <= 1,
                   "conv3d: avgpool_out needs a 1x1x1 stride-1 conv on <= 128 positions per sample (%d), Cin %% 16 == 0, Cout %% 64 == 0, x 16-byte aligned, "
                   "no other epilogue operand, activation none or ReLU", a.THWo);
    c.algo = ADVHIP_ALGO_DMA2_BASE + ADVHIP_ALGO_IGEMM_128x64;
    c.splits = 1;
  }
  if (ybs != y_dense && !reduces_in_kernel(c.algo)) c.splits = 1;  // the separate split-K reduce pass writes a dense y
  ADVHIP_REQUIRE(!ln || (d->kt == 1 && d->kh == 1 && d->kw == 1 && d->st == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 && d->ph == 0 && d->pw == 0),
                 "conv3d: the LayerNorm fold applies to 1x1x1 stride-1 convs (per-position statistics of the input)");
  ADVHIP_REQUIRE((y_preact == nullptr && dact_z == nullptr && !ln) || c.splits == 1 || reduces_in_kernel(c.algo),
                 "conv3d: pre-activation output / GELU-backward multiplier need a fused epilogue (algo %d with %d splits reduces in a second launch)", c.algo, c.splits);
  int BM, BN, BK;
  tile_of(c.algo, &BM, &BN, &BK);
  ADVHIP_REQUIRE(instantiated(c.algo), "conv3d: algo %d is not instantiated in this library", c.algo);
  const bool fast = c.algo >= ADVHIP_ALGO_FAST_BASE;  // includes the LDS-DMA and split-bf16 ids
  if (fast) {
    ADVHIP_REQUIRE(fast_ok(d, in_elems, (long long)g.Kpad * d->Cout),
                   "conv3d: fast kernel needs kernel extents <= 10 and < 3.75 GiB operands (k=%d,%d,%d)", d->kt, d->kh, d->kw);
    a.kt_ = d->kt; a.kh_ = d->kh; a.kw_ = d->kw;
    a.pad_off = d->pt * d->H * d->W + d->ph * d->W + d->pw;
    a.x_bytes = (unsigned)((in_elems + a.pad_off) * 4);
    a.w_bytes = (unsigned)((long long)g.Kpad * d->Cout * 4);
  }
  ADVHIP_REQUIRE(d->Cout % BN == 0, "conv3d: Cout=%d not a multiple of the %d-wide N tile", d->Cout, BN);
  ADVHIP_REQUIRE(g.Kpad % BK == 0 || BK == 16, "conv3d: internal: Kpad");
  a.splits = c.splits;
  a.slab = M * d->Cout;
  const int nk = (g.Kpad + BK - 1) / BK;
  ADVHIP_REQUIRE(c.splits >= 1 && c.splits <= nk && c.splits <= 64, "conv3d: bad split count %d (k-tiles %d)", c.splits, nk);
  a.part = nullptr; a.cnt = nullptr; a.part_bytes = 0;
  a.mix_big = a.mix_small = a.mix_mbase = 0;
  const SplitLayout lay = split_layout(d, g, c);
  if (c.splits > 1) {
    ADVHIP_REQUIRE(workspace != nullptr && workspace_bytes >= lay.total(),
                   "conv3d: split-K needs a %lld-byte workspace (got %lld); query advhip_conv3d_workspace_bytes",
                   (long long)lay.total(), (long long)workspace_bytes);
    if (reduces_in_kernel(c.algo)) {
      ADVHIP_REQUIRE(lay.part_bytes < 0xF0000000ll && (uintptr_t)workspace % 16 == 0,
                     "conv3d: split-K partial tiles need a 16-byte aligned workspace below 3.75 GiB (%lld bytes)", (long long)lay.part_bytes);
      a.cnt = reinterpret_cast<unsigned*>(workspace);
      a.part = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + lay.cnt_bytes);
      a.part_bytes = (unsigned)lay.part_bytes;
    } else {
      a.y = reinterpret_cast<float*>(workspace);
    }
  }
  // every tap of every output position inside the input: no validity mask needed
  const bool nocheck = d->kt == 1 && d->kh == 1 && d->kw == 1 && d->pt == 0 && d->ph == 0 && d->pw == 0 && g.K == g.Kpad;
  // 1x1x1 stride-1 conv whose rows are not a multiple of 4 positions long (layer 4: 2 x 7 x 7 = 98), 2-deep LDS-DMA ring, unsplit,
  // plain epilogue: pad every sample's rows to a multiple of 4 IN THE M INDEX SPACE ONLY (ConvArgs::MP), so that the A tile
  // can go to LDS as 16-byte pieces (a group of 4 positions never straddles two samples; the last group of a row reads
  // on into the next channel's row -- or past the tensor, where the buffer range check returns zeros -- for positions nobody stores)
  long long Mv = M;
  bool a16pad = false;
  if (nocheck && d->st == 1 && d->sh == 1 && d->sw == 1 && a.THW % 4 != 0 && ((uintptr_t)x & 15) == 0 && c.splits == 1 &&
      ((c.algo > ADVHIP_ALGO_DMA2_BASE && c.algo <= ADVHIP_ALGO_DMA2_BASE + ADVHIP_ALGO_IGEMM_256x64) || c.algo == ADVHIP_ALGO_MIXED_128x64 || is_persist(c.algo)) && !ln && y_preact == nullptr && dact_z == nullptr) {
    a.MP = (a.THWo + 3) / 4 * 4;
    Mv = (long long)d->B * a.MP;
    if (Mv < (1ll << 31)) {
      a16pad = true;
      a.M = (int)Mv;
      a.dTHWo = FastDiv::make((unsigned)a.MP);
    } else {
      a.MP = a.THWo;
      Mv = M;
    }
  }
  if (avg_out != nullptr) {  // (the same virtual padding, to the whole tile: the last groups of a sample read on into rows nobody adds up)
    a.MP = 128;
    Mv = (long long)d->B * a.MP;
    ADVHIP_REQUIRE(Mv < (1ll << 31), "conv3d: too many samples for the fused mean");
    a16pad = true;
    a.M = (int)Mv;
    a.dTHWo = FastDiv::make((unsigned)a.MP);
    a.avg_out = avg_out;
  }
  a.tiles_m = (int)((Mv + BM - 1) / BM);
  a.tiles_n = d->Cout / BN;
  a.dTilesN = FastDiv::make((unsigned)a.tiles_n);
  a.dSplits = FastDiv::make((unsigned)c.splits);
  int tspan_bt = 0;
  if (c.algo == ADVHIP_ALGO_TSPAN_128x64) {
    // m-tiles that span T: (kt,1,1) stride-1 "same" convs only; the spatial plane is presented as one flattened row
    ADVHIP_REQUIRE(d->kh == 1 && d->kw == 1 && d->st == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0 && 2 * d->pt + 1 == d->kt,
                   "conv3d: ADVHIP_ALGO_TSPAN needs a (kt,1,1) stride-1 conv with padding kt/2 (k=%d,%d,%d)", d->kt, d->kh, d->kw);
    ADVHIP_REQUIRE(c.splits == 1 && g.To % 2 == 0 && y_preact == nullptr && dact_z == nullptr && !ln,
                   "conv3d: ADVHIP_ALGO_TSPAN runs unsplit on an even number of frames (T=%d, splits=%d) without the MGFN epilogue operands", g.To, c.splits);
    tspan_bt = g.To % 4 == 0 ? 4 : 2;
    a.H = 1; a.W = d->H * d->W; a.Ho = 1; a.Wo = a.HWo;
    a.kh_ = 1; a.kw_ = 1;
    set_bricks(a, g.To / tspan_bt, 1, (a.Wo + 128 / tspan_bt - 1) / (128 / tspan_bt));
    a.Tp = 0;
    // 16-byte gather pieces: a tile row segment is BW consecutive positions of one flattened plane, tap dt of a k-row reads
    // the same segment one plane on -- four consecutive positions are four consecutive floats (at a 4-byte aligned address:
    // LDS-DMA takes it, tools/probe/), so a lane fetches a 4-position group and a wave-instruction two whole k-rows, offsets
    // from the conv's own compact table {(ci*T + dt)*HW*4, tap bits}
    a.ktab_s2w = reinterpret_cast<const int2*>(a.ktab + g.Kpad);
    a.s2w_rowp = a.W;
  }
  dim3 grid((unsigned)(a.tiles_m * a.tiles_n * c.splits));
  hipStream_t st = (hipStream_t)stream;
  if (a.cnt != nullptr) {
    const int64_t tiles = (int64_t)a.tiles_m * a.tiles_n;
    if (ep && ep->splitk_counters && ep->splitk_counter_bytes >= tiles * 4) {
      // the caller's counter block: zero when allocated, and every launch leaves it zero (the last arriver of a tile resets its word)
      ADVHIP_REQUIRE((uintptr_t)ep->splitk_counters % 4 == 0, "conv3d: misaligned split-K counter block");
      a.cnt = reinterpret_cast<unsigned*>(ep->splitk_counters);
    } else if (hipMemsetAsync(a.cnt, 0, (size_t)lay.cnt_bytes, st) != hipSuccess) {  // (a memset node: graph-capturable, replayed first)
      return check_launch("conv3d split-K counters");
    }
  }
  // BK = 32 needs Kpad % 32 == 0: the packed weights are padded to 16 rows only, but the k-table
  // marks rows >= K invalid and the weight rows read beyond Kpad must exist -> require it.
  if (BK == 32) ADVHIP_REQUIRE(g.Kpad % 32 == 0, "conv3d: BK=32 variants need K padded to 32 (K=%d)", g.K);
  // (kt,1,1) stride-1 conv whose planes are a multiple of 4 positions long: a group of 4 consecutive m is 16 contiguous bytes for
  // every tap (all inside or all padding) -- 16-byte gather pieces, offsets from the conv's own compact table
  const bool s16 = !nocheck && d->kh == 1 && d->kw == 1 && d->st == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0 && a.HW % 4 == 0 &&
                   d->kt <= 10 && g.K % 4 == 0;
  if (s16) a.ktab_s2w = reinterpret_cast<const int2*>(a.ktab + g.Kpad);
  // ... and rows of 4 consecutive positions contiguous and 16-byte aligned in x: 16-byte LDS-DMA pieces for A
  a.a16 = (a16pad || (nocheck && d->st == 1 && d->sh == 1 && d->sw == 1 && a.THW % 4 == 0 && xbs % 4 == 0 && ((uintptr_t)x & 15) == 0)) ? 1 : 0;
#define ADVHIP_FAST_CASE(ID, BM_, BN_, BK_)                                                                         \
  case ADVHIP_ALGO_FAST_BASE + ID:                                                                                  \
    if (nocheck) hipLaunchKernelGGL((conv3d_igemm_fast_kernel<BM_, BN_, BK_, false>), grid, dim3(256), 0, st, a);   \
    else hipLaunchKernelGGL((conv3d_igemm_fast_kernel<BM_, BN_, BK_, true>), grid, dim3(256), 0, st, a);            \
    break;
#define ADVHIP_DMA_CASE(ID, BM_, BN_, BK_)                                                                        \
  case ADVHIP_ALGO_DMA_BASE + ID:                                                                                 \
    if (nocheck) hipLaunchKernelGGL((conv3d_igemm_dma_kernel<BM_, BN_, BK_, false>), grid, dim3(256), 0, st, a);  \
    else hipLaunchKernelGGL((conv3d_igemm_dma_kernel<BM_, BN_, BK_, true>), grid, dim3(256), 0, st, a);           \
    break;
#define ADVHIP_DMA4_CASE(ID, BM_, BN_, BK_)                                                                          \
  case ADVHIP_ALGO_DMA4_BASE + ID:                                                                                   \
    if (nocheck) hipLaunchKernelGGL((conv3d_igemm_dma_kernel<BM_, BN_, BK_, false, 4>), grid, dim3(256), 0, st, a);  \
    else hipLaunchKernelGGL((conv3d_igemm_dma_kernel<BM_, BN_, BK_, true, 4>), grid, dim3(256), 0, st, a);           \
    break;
#define ADVHIP_DMA2_CASE(ID, BM_, BN_, BK_)                                   \
  case ADVHIP_ALGO_DMA2_BASE + ID:                                            \
    launch_dma2<BM_, BN_, BK_>(nocheck, s16, grid, st, a);                    \
    break;
#define ADVHIP_BF16X3_CASE(ID, BN_)                                                                               \
  case ADVHIP_ALGO_BF16X3_BASE + ID:                                                                               \
    if (nocheck) hipLaunchKernelGGL((conv3d_igemm_bf16x3_kernel<BN_, false>), grid, dim3(256), 0, st, a);          \
    else hipLaunchKernelGGL((conv3d_igemm_bf16x3_kernel<BN_, true>), grid, dim3(256), 0, st, a);                   \
    break;
  if (avg_out != nullptr) {
    // (a.a16 is always set here -- the rows are padded in the M index space -- so the compile-time a16 form applies)
    hipLaunchKernelGGL((conv3d_igemm_dma_kernel<128, 64, 16, false, 2, EPI_AVG, false, 2>), grid, dim3(256), 0, st, a);
    return check_launch("conv3d + mean");
  }
  if (c.algo == ADVHIP_ALGO_MIXED_128x64) {
    ADVHIP_REQUIRE(a.a16 != 0 && c.splits == 1 && !ln && dact_z == nullptr && y_preact == nullptr,
                   "conv3d: ADVHIP_ALGO_MIXED_128x64 takes unsplit 1x1x1 stride-1 convs on 16-byte aligned rows without the MGFN epilogue operands "
                   "(k=%d,%d,%d, splits=%d)", d->kt, d->kh, d->kw, c.splits);
    // 128-row m-tiles, and how many of the last ones to cut into 64-row tiles: all the workgroups past the last whole round of
    // resident slots (6 per CU) must be short ones -- cutting `c` tall m-tile rows adds c * tiles_n workgroups, so c >= rem / tiles_n.
    // A launch that is below one round, on a round boundary or far into a round is the plain 128 x 64 launch.
    const long long mt = (Mv + 127) / 128, tn = a.tiles_n, slots = 6ll * device_cus();
    const long long rem = (mt * tn) % slots;
    long long cut = (mt * tn > slots && rem > 0 && rem * 4 <= slots) ? (rem + tn - 1) / tn : 0;
    if (cut > mt) cut = mt;
    if (cut == 0) {
      hipLaunchKernelGGL((conv3d_igemm_dma_kernel<128, 64, 16, false, 2, EPI_STD, false, 2>), grid, dim3(256), 0, st, a);
      return check_launch("conv3d mixed (plain)");
    }
    const long long big_rows = (mt - cut) * 128, small_mt = (Mv - big_rows + 63) / 64;
    a.mix_big = (int)((mt - cut) * tn);
    a.mix_small = (int)(small_mt * tn);
    a.mix_mbase = (int)big_rows;
    hipLaunchKernelGGL(conv1x1_mixed_tail_kernel, dim3((unsigned)(a.mix_big + a.mix_small)), dim3(256), 0, st, a);
    return check_launch("conv3d mixed tail");
  }
  if (is_persist(c.algo)) {
    // workgroups that stay: `wgs_per_cu` per compute unit (a multiple of 8 in all, at most one per tile), each walking its share of the tiles
    ADVHIP_REQUIRE(g.Kpad >= 16 * PERSIST_MIN_NK, "conv3d: the persistent kernels need K >= %d (%d k-tiles per output tile), got %d", 16 * PERSIST_MIN_NK,
                   PERSIST_MIN_NK, g.K);
    ADVHIP_REQUIRE(a.a16 != 0 && c.splits == 1 && !ln && dact_z == nullptr,
                   "conv3d: the persistent kernels (algo %d) take unsplit 1x1x1 stride-1 convs on 16-byte aligned rows without the LayerNorm fold / "
                   "GELU-backward operands (k=%d,%d,%d, splits=%d)", c.algo, d->kt, d->kh, d->kw, c.splits);
    const long long nt = (long long)a.tiles_m * a.tiles_n;
    long long g = (long long)persist_wgs_per_cu(c.algo) * device_cus();
    if (g > nt) g = nt;
    g = g / GFX950_XCDS * GFX950_XCDS;
    if (g < GFX950_XCDS) g = GFX950_XCDS;
    const dim3 pgrid((unsigned)g);
    switch (persist_tile(c.algo)) {
      case ADVHIP_ALGO_IGEMM_128x64: hipLaunchKernelGGL((conv1x1_persist_kernel<128, 64, 2>), pgrid, dim3(512), 0, st, a, GFX950_XCDS); break;
      default: hipLaunchKernelGGL((conv1x1_persist_kernel<64, 64, 2>), pgrid, dim3(512), 0, st, a, GFX950_XCDS); break;
    }
    return check_launch("conv3d persistent");
  }
  switch (c.algo) {
    case ADVHIP_ALGO_TSPAN_128x64:
      if (tspan_bt == 4) hipLaunchKernelGGL((conv3d_igemm_dma_kernel<128, 64, 16, true, 2, EPI_TSPAN4, false, 1>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((conv3d_igemm_dma_kernel<128, 64, 16, true, 2, EPI_TSPAN2, false, 1>), grid, dim3(256), 0, st, a);
      break;
    ADVHIP_DMA2_CASE(ADVHIP_ALGO_IGEMM_128x128, 128, 128, 16)
    ADVHIP_DMA2_CASE(ADVHIP_ALGO_IGEMM_128x64, 128, 64, 16)
    ADVHIP_DMA2_CASE(ADVHIP_ALGO_IGEMM_256x64, 256, 64, 16)
    ADVHIP_DMA2_CASE(ADVHIP_ALGO_IGEMM_64x64, 64, 64, 16)
    ADVHIP_DMA2_CASE(ADVHIP_ALGO_IGEMM_64x128, 64, 128, 16)
    ADVHIP_DMA2_CASE(ADVHIP_ALGO_IGEMM_128x64x32, 128, 64, 32)
    ADVHIP_DMA2_CASE(ADVHIP_ALGO_IGEMM_64x64x32, 64, 64, 32)
    ADVHIP_DMA2_CASE(ADVHIP_ALGO_IGEMM_64x128x32, 64, 128, 32)
    ADVHIP_BF16X3_CASE(ADVHIP_ALGO_IGEMM_128x128x32, 128)
    ADVHIP_BF16X3_CASE(ADVHIP_ALGO_IGEMM_128x64x32, 64)
    ADVHIP_DMA4_CASE(ADVHIP_ALGO_IGEMM_128x64, 128, 64, 16)
    ADVHIP_DMA4_CASE(ADVHIP_ALGO_IGEMM_64x64, 64, 64, 16)
    ADVHIP_DMA4_CASE(ADVHIP_ALGO_IGEMM_64x128, 64, 128, 16)
    ADVHIP_DMA_CASE(ADVHIP_ALGO_IGEMM_128x128, 128, 128, 16)
    ADVHIP_DMA_CASE(ADVHIP_ALGO_IGEMM_128x64, 128, 64, 16)
    ADVHIP_DMA_CASE(ADVHIP_ALGO_IGEMM_64x64, 64, 64, 16)
    ADVHIP_DMA_CASE(ADVHIP_ALGO_IGEMM_64x128, 64, 128, 16)
    ADVHIP_DMA_CASE(ADVHIP_ALGO_IGEMM_128x64x32, 128, 64, 32)
    ADVHIP_DMA_CASE(ADVHIP_ALGO_IGEMM_64x64x32, 64, 64, 32)
    ADVHIP_DMA_CASE(ADVHIP_ALGO_IGEMM_64x128x32, 64, 128, 32)
    ADVHIP_FAST_CASE(ADVHIP_ALGO_IGEMM_128x128, 128, 128, 16)
    ADVHIP_FAST_CASE(ADVHIP_ALGO_IGEMM_128x64, 128, 64, 16)
    ADVHIP_FAST_CASE(ADVHIP_ALGO_IGEMM_64x64, 64, 64, 16)
    ADVHIP_FAST_CASE(ADVHIP_ALGO_IGEMM_64x128, 64, 128, 16)
    ADVHIP_FAST_CASE(ADVHIP_ALGO_IGEMM_128x128x32, 128, 128, 32)
    ADVHIP_FAST_CASE(ADVHIP_ALGO_IGEMM_128x64x32, 128, 64, 32)
    ADVHIP_FAST_CASE(ADVHIP_ALGO_IGEMM_64x64x32, 64, 64, 32)
    ADVHIP_FAST_CASE(ADVHIP_ALGO_IGEMM_64x128x32, 64, 128, 32)
    case ADVHIP_ALGO_IGEMM_128x128: hipLaunchKernelGGL((conv3d_igemm_f32_kernel<128, 128, 16>), grid, dim3(256), 0, st, a); break;
    case ADVHIP_ALGO_IGEMM_128x64: hipLaunchKernelGGL((conv3d_igemm_f32_kernel<128, 64, 16>), grid, dim3(256), 0, st, a); break;
    case ADVHIP_ALGO_IGEMM_64x64: hipLaunchKernelGGL((conv3d_igemm_f32_kernel<64, 64, 16>), grid, dim3(256), 0, st, a); break;
    case ADVHIP_ALGO_IGEMM_64x128: hipLaunchKernelGGL((conv3d_igemm_f32_kernel<64, 128, 16>), grid, dim3(256), 0, st, a); break;
    case ADVHIP_ALGO_IGEMM_128x128x32: hipLaunchKernelGGL((conv3d_igemm_f32_kernel<128, 128, 32>), grid, dim3(256), 0, st, a); break;
    case ADVHIP_ALGO_IGEMM_128x64x32: hipLaunchKernelGGL((conv3d_igemm_f32_kernel<128, 64, 32>), grid, dim3(256), 0, st, a); break;
    case ADVHIP_ALGO_IGEMM_64x64x32: hipLaunchKernelGGL((conv3d_igemm_f32_kernel<64, 64, 32>), grid, dim3(256), 0, st, a); break;
    case ADVHIP_ALGO_IGEMM_64x128x32: hipLaunchKernelGGL((conv3d_igemm_f32_kernel<64, 128, 32>), grid, dim3(256), 0, st, a); break;
    default:
      set_error("conv3d: algo %d is not instantiated in this library", c.algo);
      return ADVHIP_EINVAL;
  }
  if (int rc = check_launch("conv3d_igemm")) return rc;
  if (c.splits > 1 && a.part == nullptr) {
    const long long total = a.slab;
    const int vec4 = a.vw == 4 ? 1 : 0;  // rows, slabs, residual and y all 16-byte aligned
    const long long work = vec4 ? total / 4 : total;
    const int rgrid = (int)std::min<long long>((work + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rgrid), dim3(256), 0, st, reinterpret_cast<const float*>(workspace), scale,
                       shift, residual, y, total, a.THWo, d->Cout, c.splits, d->relu, vec4);
    return check_launch("splitk_reduce");
  }
  return ADVHIP_OK;
}

// ---- conv + max-pool in one launch (brick-ordered LDS-DMA kernel) ----------------------------------------------------
namespace advhip {
static int pool_out(int n, int k, int s) { return n >= k ? (n - k) / s + 1 : 0; }

// shared argument set-up of the two pooling forms; `flat_hw`: present the (1x1x1, stride 1) conv as H = 1, W = H*W
static int fill_pool_args(ConvArgs& a, const advhip_conv3d_desc* d, const Geometry& g, const float* x, int64_t x_batch_stride,
                          const float* w_packed, const int32_t* ktab, const float* scale, const float* shift, bool flat_hw) {
  a.x = x; a.w = w_packed; a.ktab = reinterpret_cast<const int4*>(ktab);
  a.scale = scale; a.shift = shift; a.res = nullptr; a.y = nullptr; a.y2 = nullptr; a.dact = nullptr;
  a.ln_u = nullptr; a.ln_mu = nullptr; a.ln_rs = nullptr;
  a.B = d->B; a.Cin = d->Cin; a.T = d->T; a.Cout = d->Cout;
  a.H = flat_hw ? 1 : d->H; a.W = flat_hw ? d->H * d->W : d->W;
  a.st = d->st; a.sh = d->sh; a.sw = d->sw; a.pt = d->pt; a.ph = d->ph; a.pw = d->pw;
  a.To = g.To; a.Ho = flat_hw ? 1 : g.Ho; a.Wo = flat_hw ? g.Ho * g.Wo : g.Wo;
  const long long x_dense = (long long)d->Cin * d->T * d->H * d->W;
  const long long xbs = x_batch_stride > 0 ? x_batch_stride : x_dense;
  ADVHIP_REQUIRE(xbs >= x_dense, "conv3d+pool: x batch stride %lld smaller than one sample (%lld)", xbs, x_dense);
  const long long in_elems = (long long)(d->B - 1) * xbs + x_dense;
  ADVHIP_REQUIRE(fast_ok(d, in_elems, (long long)g.Kpad * d->Cout) && g.M * d->Cout < (1ll << 32) && in_elems < (1ll << 31),
                 "conv3d+pool: tensor too large for 32-bit offsets");
  a.M = (int)g.M;
  a.x_bstride = (int)xbs;
  a.Kpad = g.Kpad;
  a.HWo = g.Ho * g.Wo; a.THWo = g.To * a.HWo;
  a.HW = d->H * d->W; a.THW = d->T * a.HW;
  a.relu = d->relu;
  a.vw = 1;
  a.a16 = 0;
  a.MP = a.THWo;
  a.dTHWo = FastDiv::make((unsigned)a.THWo); a.dHWo = FastDiv::make((unsigned)a.HWo); a.dWo = FastDiv::make((unsigned)a.Wo);
  a.kt_ = d->kt; a.kh_ = d->kh; a.kw_ = d->kw;
  a.pad_off = d->pt * d->H * d->W + d->ph * d->W + d->pw;
  a.x_bytes = (unsigned)((in_elems + a.pad_off) * 4);
  a.w_bytes = (unsigned)((long long)g.Kpad * d->Cout * 4);
  a.splits = 1; a.slab = 0; a.part = nullptr; a.cnt = nullptr; a.part_bytes = 0;
  a.tiles_n = d->Cout / 64;
  a.dTilesN = FastDiv::make((unsigned)a.tiles_n);
  a.dSplits = FastDiv::make(1u);
  return ADVHIP_OK;
}

static void set_bricks(ConvArgs& a, int nbt, int nbh, int nbw) {
  a.nbh = nbh; a.nbw = nbw;
  a.dNb = FastDiv::make((unsigned)(nbt * nbh * nbw));
  a.dNbhw = FastDiv::make((unsigned)(nbh * nbw));
  a.dNbw = FastDiv::make((unsigned)nbw);
  a.tiles_m = a.B * nbt * nbh * nbw;
}
}  // namespace advhip

extern "C" int advhip_conv3d_pool_out_dims(const advhip_conv3d_desc* d, int32_t pkt, int32_t pkh, int32_t pkw, int32_t pst,
                                           int32_t psh, int32_t psw, int32_t* Tp, int32_t* Hp, int32_t* Wp) {
  if (int rc = validate(d)) return rc;
  ADVHIP_REQUIRE(pkt > 0 && pkh > 0 && pkw > 0 && pst > 0 && psh > 0 && psw > 0, "conv3d+pool: bad pooling window");
  const Geometry g = geometry(d);
  if (Tp) *Tp = pool_out(g.To, pkt, pst);
  if (Hp) *Hp = pool_out(g.Ho, pkh, psh);
  if (Wp) *Wp = pool_out(g.Wo, pkw, psw);
  return ADVHIP_OK;
}

extern "C" int64_t advhip_conv3d_relu_maxpool233_workspace_bytes(const advhip_conv3d_desc* d) {
  if (validate(d)) return -1;
  const Geometry g = geometry(d);
  const int Tp = pool_out(g.To, 2, 2), Hp = pool_out(g.Ho, 3, 2), Wp = pool_out(g.Wo, 3, 2);
  if (Tp <= 0 || Hp <= 0 || Wp <= 0) return 0;
  const int64_t bricks = (int64_t)d->B * Tp * ((2 * Hp + 1 + 3) / 4) * ((2 * Wp + 1 + 15) / 16);
  return bricks * d->Cout * POOL_SLOTS * (int64_t)sizeof(float);
}

extern "C" int advhip_conv3d_bn_relu_maxpool233_f32(const advhip_conv3d_desc* d, const float* x, int64_t x_batch_stride,
                                                    const float* w_packed, const int32_t* ktab, const float* scale,
                                                    const float* shift, float* y, int64_t y_batch_stride, void* workspace,
                                                    int64_t workspace_bytes, void* stream) {
  if (int rc = validate(d)) return rc;
  ADVHIP_REQUIRE(x && w_packed && ktab && scale && shift && y, "conv3d+pool233: null pointer");
  const Geometry g = geometry(d);
  const int Tp = pool_out(g.To, 2, 2), Hp = pool_out(g.Ho, 3, 2), Wp = pool_out(g.Wo, 3, 2);
  ADVHIP_REQUIRE(Tp > 0 && Hp > 0 && Wp > 0, "conv3d+pool233: conv output (%d,%d,%d) smaller than the (2,3,3) window", g.To, g.Ho, g.Wo);
  const int64_t need = advhip_conv3d_relu_maxpool233_workspace_bytes(d);
  ADVHIP_REQUIRE(workspace != nullptr && workspace_bytes >= need && need < 0xF0000000ll,
                 "conv3d+pool233: needs a %lld-byte workspace (got %lld)", (long long)need, (long long)workspace_bytes);
  const long long y_dense = (long long)d->Cout * Tp * Hp * Wp;
  const long long ybs = y_batch_stride > 0 ? y_batch_stride : y_dense;
  ADVHIP_REQUIRE(ybs >= y_dense, "conv3d+pool233: y batch stride %lld smaller than one pooled sample (%lld)", ybs, y_dense);
  ConvArgs a;
  if (int rc = fill_pool_args(a, d, g, x, x_batch_stride, w_packed, ktab, scale, shift, false)) return rc;
  a.y = reinterpret_cast<float*>(workspace);  // per-brick partial maxima
  a.y_bstride = 0; a.Tp = Tp; a.relu = 1;
  const int nbh = (2 * Hp + 1 + 3) / 4, nbw = (2 * Wp + 1 + 15) / 16;
  set_bricks(a, Tp, nbh, nbw);
  // (every check before the first launch: a rejected call must not leave half an op enqueued)
  ADVHIP_REQUIRE(nbw <= MERGE_MAX_NBW, "conv3d+pool233: pooled width %d above %d", Wp, MERGE_MAX_NBW * 8 - 1);
  const long long rows = (long long)d->B * a.tiles_n * 2 * Tp * Hp;
  ADVHIP_REQUIRE(rows < (1ll << 31), "conv3d+pool233: too many output rows");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(a.tiles_m * a.tiles_n));
  const bool nocheck = d->kt == 1 && d->kh == 1 && d->kw == 1 && d->pt == 0 && d->ph == 0 && d->pw == 0 && g.K == g.Kpad;
  if (nocheck) hipLaunchKernelGGL((conv3d_igemm_dma_kernel<128, 64, 16, false, 2, EPI_POOL233>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv3d_igemm_dma_kernel<128, 64, 16, true, 2, EPI_POOL233>), grid, dim3(256), 0, st, a);
  if (int rc = check_launch("conv3d+pool233")) return rc;
  const unsigned gx = (unsigned)std::min<long long>(rows, 1 << 20), gy = (unsigned)((rows + gx - 1) / gx);
  hipLaunchKernelGGL(stem_pool_merge_kernel, dim3(gx, gy), dim3(256), (size_t)2 * nbw * 288 * sizeof(float), st, reinterpret_cast<const float*>(workspace), y, d->Cout, Tp,
                     Hp, Wp, nbh, nbw, a.tiles_n, FastDiv::make((unsigned)Wp), rows, ybs);
  return check_launch("stem_pool_merge");
}

// ---- the same stem on column-parity planes of its input (16-byte A pieces for the stride-2 gather) --------------------------
namespace advhip {
static int s2w_check(const advhip_conv3d_desc* d, const Geometry& g) {
  ADVHIP_REQUIRE(d->sw == 2 && d->kw % 2 == 1 && d->pw == d->kw / 2 && d->pw <= 2 * S2W_PADL && d->W % 2 == 0 && d->W >= 8,
                 "conv3d s2w: needs stride 2, an odd kernel <= %d with 'same' padding along w and an even W (k=%d s=%d p=%d W=%d)",
                 4 * S2W_PADL + 1, d->kw, d->sw, d->pw, d->W);
  ADVHIP_REQUIRE(g.Wo % 4 == 0, "conv3d s2w: output width %d is not a multiple of 4", g.Wo);
  ADVHIP_REQUIRE(d->kt <= 10 && d->kh <= 10, "conv3d s2w: kernel extents above 10");
  return ADVHIP_OK;
}
}  // namespace advhip

extern "C" int32_t advhip_split_w_plane_floats(int32_t W) { return W > 0 && W % 2 == 0 ? W / 2 + advhip::S2W_PAD : -1; }

extern "C" int advhip_split_w_f32(const float* x, float* xs, int64_t rows, int32_t W, void* stream) {
  ADVHIP_REQUIRE(x && xs && rows > 0 && W > 0 && W % 2 == 0, "split_w: bad arguments");
  const int WP = W / 2 + S2W_PAD;
  const long long total = rows * 2 * WP;
  const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 64);
  hipLaunchKernelGGL(split_w_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, xs, (long long)rows, W, WP);
  return check_launch("split_w");
}

extern "C" int advhip_conv3d_s2w_build_ktab(const advhip_conv3d_desc* d, int32_t* ktab_s2w, void* stream) {
  if (int rc = validate(d)) return rc;
  ADVHIP_REQUIRE(ktab_s2w, "conv3d s2w build_ktab: null pointer");
  const Geometry g = geometry(d);
  if (int rc = s2w_check(d, g)) return rc;
  hipLaunchKernelGGL(build_ktab_s2w_kernel, dim3((g.Kpad + 255) / 256), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<int2*>(ktab_s2w), d->kt,
                     d->kh, d->kw, d->pw, g.K, g.Kpad, d->H, d->T, d->W / 2 + S2W_PAD);
  return check_launch("build_ktab_s2w");
}

extern "C" int advhip_conv3d_s2w_bn_relu_maxpool233_f32(const advhip_conv3d_desc* d, const float* xs, int64_t xs_batch_stride,
                                                        const float* w_packed, const int32_t* ktab_s2w, const float* scale,
                                                        const float* shift, float* y, int64_t y_batch_stride, void* workspace,
                                                        int64_t workspace_bytes, void* stream) {
  if (int rc = validate(d)) return rc;
  ADVHIP_REQUIRE(xs && w_packed && ktab_s2w && scale && shift && y, "conv3d s2w+pool233: null pointer");
  const Geometry g = geometry(d);
  if (int rc = s2w_check(d, g)) return rc;
  const int Tp = pool_out(g.To, 2, 2), Hp = pool_out(g.Ho, 3, 2), Wp = pool_out(g.Wo, 3, 2);
  ADVHIP_REQUIRE(Tp > 0 && Hp > 0 && Wp > 0, "conv3d s2w+pool233: conv output (%d,%d,%d) smaller than the (2,3,3) window", g.To, g.Ho, g.Wo);
  const int64_t need = advhip_conv3d_relu_maxpool233_workspace_bytes(d);
  ADVHIP_REQUIRE(workspace != nullptr && workspace_bytes >= need && need < 0xF0000000ll,
                 "conv3d s2w+pool233: needs a %lld-byte workspace (got %lld)", (long long)need, (long long)workspace_bytes);
  const long long y_dense = (long long)d->Cout * Tp * Hp * Wp;
  const long long ybs = y_batch_stride > 0 ? y_batch_stride : y_dense;
  ADVHIP_REQUIRE(ybs >= y_dense, "conv3d s2w+pool233: y batch stride %lld smaller than one pooled sample (%lld)", ybs, y_dense);
  const int WP = d->W / 2 + S2W_PAD, rowp = 2 * WP;
  const long long xs_dense = (long long)d->Cin * d->T * d->H * rowp;
  const long long xbs = xs_batch_stride > 0 ? xs_batch_stride : xs_dense;
  ADVHIP_REQUIRE(xbs >= xs_dense && ((uintptr_t)xs & 15) == 0, "conv3d s2w+pool233: xs batch stride %lld smaller than one sample (%lld) or xs not 16-byte aligned", xbs, xs_dense);
  ConvArgs a;
  if (int rc = fill_pool_args(a, d, g, xs, 0, w_packed, ktab_s2w, scale, shift, false)) return rc;
  // the gather's view of the input: rows of 2 * WP floats
  const long long in_elems = (long long)(d->B - 1) * xbs + xs_dense;
  ADVHIP_REQUIRE(in_elems < (1ll << 30), "conv3d s2w+pool233: input above 4 GiB");
  a.x_bstride = (int)xbs;
  a.s2w_rowp = rowp;
  a.ktab_s2w = reinterpret_cast<const int2*>(ktab_s2w);
  a.pad_off = (d->pt * d->H + d->ph) * rowp;
  a.x_bytes = (unsigned)((in_elems + a.pad_off) * 4);
  a.y = reinterpret_cast<float*>(workspace);  // per-brick partial maxima
  a.y_bstride = 0; a.Tp = Tp; a.relu = 1;
  const int nbh = (2 * Hp + 1 + 3) / 4, nbw = (2 * Wp + 1 + 15) / 16;
  set_bricks(a, Tp, nbh, nbw);
  ADVHIP_REQUIRE(nbw <= MERGE_MAX_NBW, "conv3d s2w+pool233: pooled width %d above %d", Wp, MERGE_MAX_NBW * 8 - 1);
  const long long rows = (long long)d->B * a.tiles_n * 2 * Tp * Hp;
  ADVHIP_REQUIRE(rows < (1ll << 31), "conv3d s2w+pool233: too many output rows");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((conv3d_igemm_dma_kernel<128, 64, 16, true, 2, EPI_POOL233, false, 1>), dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(256), 0, st, a);
  if (int rc = check_launch("conv3d s2w+pool233")) return rc;
  const unsigned gx = (unsigned)std::min<long long>(rows, 1 << 20), gy = (unsigned)((rows + gx - 1) / gx);
  hipLaunchKernelGGL(stem_pool_merge_kernel, dim3(gx, gy), dim3(256), (size_t)2 * nbw * 288 * sizeof(float), st, reinterpret_cast<const float*>(workspace), y, d->Cout, Tp,
                     Hp, Wp, nbh, nbw, a.tiles_n, FastDiv::make((unsigned)Wp), rows, ybs);
  return check_launch("stem_pool_merge");
}

// ---- stem + maxpool1 straight from resized uint8 frames (TenCrop, float conversion and normalisation in the load stage) ----
namespace advhip {
static int u8_check_frames(const advhip_conv3d_desc* d, int64_t F, int FH, int FW) {
  ADVHIP_REQUIRE(F > 0 && F % d->T == 0, "conv3d u8: %lld frames are not whole clips of %d", (long long)F, d->T);
  ADVHIP_REQUIRE(FH >= d->H && FW >= d->W, "conv3d u8: frames (%d x %d) smaller than the %d x %d crop", FH, FW, d->H, d->W);
  ADVHIP_REQUIRE(d->kt <= 10 && d->kh <= 10 && d->kw <= 10, "conv3d u8: kernel extents above 10");
  ADVHIP_REQUIRE(F * FH * FW * d->Cin < (1ll << 31) - (1 << 24), "conv3d u8: frames tensor above 2 GiB");
  return ADVHIP_OK;
}
}  // namespace advhip

extern "C" int advhip_conv3d_u8_table_sizes(const advhip_conv3d_desc* d, int64_t* ktab_ints, int64_t* corr_floats) {
  if (int rc = validate(d)) return rc;
  const Geometry g = geometry(d);
  if (ktab_ints) *ktab_ints = 4ll * g.Kpad;
  if (corr_floats) *corr_floats = (int64_t)(d->pt + 1) * (d->pt + 1) * (d->ph + 1) * (d->ph + 1) * (d->pw + 1) * (d->pw + 1) * d->Cout;
  return ADVHIP_OK;
}

extern "C" int advhip_conv3d_u8_build_tables(const advhip_conv3d_desc* d, int32_t FH, int32_t FW, const float* w_packed, float mean,
                                             int32_t* ktab_u8, float* corr, void* stream) {
  if (int rc = validate(d)) return rc;
  ADVHIP_REQUIRE(w_packed && ktab_u8 && corr, "conv3d u8 tables: null pointer");
  if (int rc = u8_check_frames(d, d->T, FH, FW)) return rc;
  const Geometry g = geometry(d);
  int64_t total = 0;
  advhip_conv3d_u8_table_sizes(d, nullptr, &total);
  ADVHIP_REQUIRE(total < (1ll << 28), "conv3d u8 tables: padding (%d,%d,%d) too large", d->pt, d->ph, d->pw);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(build_ktab_u8_kernel, dim3((g.Kpad + 255) / 256), dim3(256), 0, st, reinterpret_cast<int2*>(ktab_u8), d->kt, d->kh,
                     d->kw, d->Cin, g.K, g.Kpad, FW * d->Cin, FH * FW * d->Cin);
  hipLaunchKernelGGL(u8_pad_corr_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, st, w_packed, corr,
                     (int)total, d->Cin, d->kt, d->kh, d->kw, d->pt, d->ph, d->pw, d->Cout, mean);
  return check_launch("conv3d_u8_build_tables");
}

extern "C" int advhip_conv3d_u8_tencrop_bn_relu_maxpool233_f32(const advhip_conv3d_desc* d, const uint8_t* frames, int64_t F, int32_t FH,
                                                               int32_t FW, int64_t first_crop_clip, const float* w_packed,
                                                               const int32_t* ktab_u8, const float* corr,
                                                               const float* scale, const float* shift, float stdv, float* y,
                                                               int64_t y_batch_stride, void* workspace, int64_t workspace_bytes,
                                                               void* stream) {
  if (int rc = validate(d)) return rc;
  ADVHIP_REQUIRE(frames && w_packed && ktab_u8 && corr && scale && shift && y, "conv3d u8+pool233: null pointer");
  ADVHIP_REQUIRE(d->pt < d->kt && d->ph < d->kh && d->pw < d->kw, "conv3d u8+pool233: padding not smaller than the kernel");
  ADVHIP_REQUIRE(stdv != 0.f, "conv3d u8+pool233: std must be non-zero");
  if (int rc = u8_check_frames(d, F, FH, FW)) return rc;
  ADVHIP_REQUIRE(first_crop_clip >= 0 && first_crop_clip + d->B <= F / d->T * 10,
                 "conv3d u8+pool233: crop-clips [%lld, %lld) outside the %lld clips x 10 crops of the frames", (long long)first_crop_clip,
                 (long long)first_crop_clip + d->B, (long long)(F / d->T));
  const Geometry g = geometry(d);
  const int Tp = pool_out(g.To, 2, 2), Hp = pool_out(g.Ho, 3, 2), Wp = pool_out(g.Wo, 3, 2);
  ADVHIP_REQUIRE(Tp > 0 && Hp > 0 && Wp > 0, "conv3d u8+pool233: conv output (%d,%d,%d) smaller than the (2,3,3) window", g.To, g.Ho, g.Wo);
  const int64_t need = advhip_conv3d_relu_maxpool233_workspace_bytes(d);
  ADVHIP_REQUIRE(workspace != nullptr && workspace_bytes >= need && need < 0xF0000000ll,
                 "conv3d u8+pool233: needs a %lld-byte workspace (got %lld)", (long long)need, (long long)workspace_bytes);
  const long long y_dense = (long long)d->Cout * Tp * Hp * Wp;
  const long long ybs = y_batch_stride > 0 ? y_batch_stride : y_dense;
  ADVHIP_REQUIRE(ybs >= y_dense, "conv3d u8+pool233: y batch stride %lld smaller than one pooled sample (%lld)", ybs, y_dense);
  ConvArgs a;
  if (int rc = fill_pool_args(a, d, g, reinterpret_cast<const float*>(frames), 0, w_packed, nullptr, scale, shift, false)) return rc;
  a.u8_first = (int)first_crop_clip; a.u8_FH = FH; a.u8_FW = FW;
  // torchvision center_crop: int(round((H - crop) / 2.0)) with Python's round-half-to-even
  auto half_even = [](int v) { return (v % 2 == 0) ? v / 2 : ((v / 2) % 2 == 0 ? v / 2 : v / 2 + 1); };
  a.u8_ctop = half_even(FH - d->H); a.u8_cleft = half_even(FW - d->W);
  a.in_std = stdv;  // (the mean went into `corr` when the tables were built)
  a.ktab_u8 = reinterpret_cast<const int2*>(ktab_u8); a.pad_corr = corr;
  // byte offsets: the window origin of a border output lies up to (pt, ph, pw + kw - 1) before the crop's corner
  a.pad_off = (d->pt * FH * FW + d->ph * FW + d->pw + d->kw) * d->Cin;
  a.x_bytes = (unsigned)(F * FH * FW * d->Cin + a.pad_off);
  a.y = reinterpret_cast<float*>(workspace);
  a.y_bstride = 0; a.Tp = Tp; a.relu = 1;
  const int nbh = (2 * Hp + 1 + 3) / 4, nbw = (2 * Wp + 1 + 15) / 16;
  set_bricks(a, Tp, nbh, nbw);
  // (every check before the first launch: a rejected call must not leave half an op enqueued)
  ADVHIP_REQUIRE(nbw <= MERGE_MAX_NBW, "conv3d u8+pool233: pooled width %d above %d", Wp, MERGE_MAX_NBW * 8 - 1);
  const long long rows = (long long)d->B * a.tiles_n * 2 * Tp * Hp;
  ADVHIP_REQUIRE(rows < (1ll << 31), "conv3d u8+pool233: too many output rows");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((conv3d_igemm_dma_kernel<128, 64, 16, true, 2, EPI_POOL233, true>), dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(256), 0, st, a);
  if (int rc = check_launch("conv3d u8+pool233")) return rc;
  const unsigned gx = (unsigned)std::min<long long>(rows, 1 << 20), gy = (unsigned)((rows + gx - 1) / gx);
  hipLaunchKernelGGL(stem_pool_merge_kernel, dim3(gx, gy), dim3(256), (size_t)2 * nbw * 288 * sizeof(float), st, reinterpret_cast<const float*>(workspace), y, d->Cout, Tp,
                     Hp, Wp, nbh, nbw, a.tiles_n, FastDiv::make((unsigned)Wp), rows, ybs);
  return check_launch("stem_pool_merge");
}

// ---- the same from whole pixels (stem_u8_tap_kernel): tables and launch ----------------------------------------------------
namespace advhip {
constexpr int U8_TAPS_PER_TILE = 8;  // (16: 2.62 ms against 2.50 at B = 32 -- 3 % more padded K and 4 instead of 6 workgroups per CU)
static int u8_taps_pad(const advhip_conv3d_desc* d) {
  const int taps = d->kt * d->kh * d->kw;
  return (taps + U8_TAPS_PER_TILE - 1) / U8_TAPS_PER_TILE * U8_TAPS_PER_TILE;
}
}  // namespace advhip

extern "C" int advhip_conv3d_u8_taps_table_sizes(const advhip_conv3d_desc* d, int64_t* ktab_ints, int64_t* corr_floats, int64_t* w_taps_floats) {
  if (int rc = validate(d)) return rc;
  if (int rc = advhip_conv3d_u8_table_sizes(d, nullptr, corr_floats)) return rc;
  const int tp = u8_taps_pad(d);
  if (ktab_ints) *ktab_ints = 4ll * tp;
  if (w_taps_floats) *w_taps_floats = (int64_t)tp * d->Cin * d->Cout;
  return ADVHIP_OK;
}

extern "C" int advhip_conv3d_u8_taps_build_tables(const advhip_conv3d_desc* d, int32_t FH, int32_t FW, const float* w_packed, float mean,
                                                  int32_t* ktab_taps, float* corr, float* w_taps, void* stream) {
  if (int rc = validate(d)) return rc;
  ADVHIP_REQUIRE(w_packed && ktab_taps && corr && w_taps, "conv3d u8 taps tables: null pointer");
  ADVHIP_REQUIRE(d->Cin == 3 && d->Cout == 64, "conv3d u8 taps: 3-channel pixels and 64 output channels (Cin=%d, Cout=%d)", d->Cin, d->Cout);
  if (int rc = u8_check_frames(d, d->T, FH, FW)) return rc;
  int64_t total = 0;
  advhip_conv3d_u8_table_sizes(d, nullptr, &total);
  ADVHIP_REQUIRE(total < (1ll << 28), "conv3d u8 taps tables: padding (%d,%d,%d) too large", d->pt, d->ph, d->pw);
  const int taps = d->kt * d->kh * d->kw, tp = u8_taps_pad(d);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(build_ktab_u8_taps_kernel, dim3((tp + 255) / 256), dim3(256), 0, st, reinterpret_cast<int2*>(ktab_taps), d->kt, d->kh, d->kw,
                     d->Cin, tp, FW, FH * FW);
  hipLaunchKernelGGL(pack_weight_taps_kernel, dim3((unsigned)(((long long)tp * d->Cin * d->Cout + 255) / 256)), dim3(256), 0, st, w_packed, w_taps,
                     d->Cout, d->Cin, taps, tp);
  hipLaunchKernelGGL(u8_pad_corr_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, st, w_packed, corr,
                     (int)total, d->Cin, d->kt, d->kh, d->kw, d->pt, d->ph, d->pw, d->Cout, mean);
  return check_launch("conv3d_u8_taps_build_tables");
}

extern "C" int advhip_conv3d_u8_taps_tencrop_bn_relu_maxpool233_f32(const advhip_conv3d_desc* d, const uint8_t* frames, int64_t F, int32_t FH,
                                                                    int32_t FW, int64_t readable_bytes, int64_t first_crop_clip,
                                                                    const float* w_taps, const int32_t* ktab_taps, const float* corr,
                                                                    const float* scale, const float* shift, float stdv, float* y,
                                                                    int64_t y_batch_stride, void* workspace, int64_t workspace_bytes,
                                                                    void* stream) {
  if (int rc = validate(d)) return rc;
  ADVHIP_REQUIRE(frames && w_taps && ktab_taps && corr && scale && shift && y, "conv3d u8 taps+pool233: null pointer");
  ADVHIP_REQUIRE(d->Cin == 3 && d->Cout == 64, "conv3d u8 taps+pool233: 3-channel pixels and 64 output channels (Cin=%d, Cout=%d)", d->Cin, d->Cout);
  ADVHIP_REQUIRE(d->pt < d->kt && d->ph < d->kh && d->pw < d->kw, "conv3d u8 taps+pool233: padding not smaller than the kernel");
  ADVHIP_REQUIRE(stdv != 0.f, "conv3d u8 taps+pool233: std must be non-zero");
  if (int rc = u8_check_frames(d, F, FH, FW)) return rc;
  const int64_t fbytes = F * FH * FW * 3;
  ADVHIP_REQUIRE(readable_bytes >= fbytes + 1, "conv3d u8 taps+pool233: the frames allocation must extend one byte past the last pixel "
                 "(pixels are fetched as 4-byte pieces): %lld readable, %lld needed", (long long)readable_bytes, (long long)fbytes + 1);
  ADVHIP_REQUIRE(first_crop_clip >= 0 && first_crop_clip + d->B <= F / d->T * 10,
                 "conv3d u8 taps+pool233: crop-clips [%lld, %lld) outside the %lld clips x 10 crops of the frames", (long long)first_crop_clip,
                 (long long)first_crop_clip + d->B, (long long)(F / d->T));
  const Geometry g = geometry(d);
  const int Tp = pool_out(g.To, 2, 2), Hp = pool_out(g.Ho, 3, 2), Wp = pool_out(g.Wo, 3, 2);
  ADVHIP_REQUIRE(Tp > 0 && Hp > 0 && Wp > 0, "conv3d u8 taps+pool233: conv output (%d,%d,%d) smaller than the (2,3,3) window", g.To, g.Ho, g.Wo);
  const int64_t need = advhip_conv3d_relu_maxpool233_workspace_bytes(d);
  ADVHIP_REQUIRE(workspace != nullptr && workspace_bytes >= need && need < 0xF0000000ll,
                 "conv3d u8 taps+pool233: needs a %lld-byte workspace (got %lld)", (long long)need, (long long)workspace_bytes);
  const long long y_dense = (long long)d->Cout * Tp * Hp * Wp;
  const long long ybs = y_batch_stride > 0 ? y_batch_stride : y_dense;
  ADVHIP_REQUIRE(ybs >= y_dense, "conv3d u8 taps+pool233: y batch stride %lld smaller than one pooled sample (%lld)", ybs, y_dense);
  ConvArgs a;
  if (int rc = fill_pool_args(a, d, g, reinterpret_cast<const float*>(frames), 0, w_taps, nullptr, scale, shift, false)) return rc;
  a.u8_first = (int)first_crop_clip; a.u8_FH = FH; a.u8_FW = FW;
  auto half_even = [](int v) { return (v % 2 == 0) ? v / 2 : ((v / 2) % 2 == 0 ? v / 2 : v / 2 + 1); };
  a.u8_ctop = half_even(FH - d->H); a.u8_cleft = half_even(FW - d->W);
  a.in_std = stdv;
  a.ktab_u8 = reinterpret_cast<const int2*>(ktab_taps); a.pad_corr = corr;
  const int tp = u8_taps_pad(d);
  a.Kpad = 3 * tp;
  a.w_bytes = (unsigned)((long long)a.Kpad * d->Cout * 4);
  a.pad_off = (d->pt * FH * FW + d->ph * FW + d->pw + d->kw) * 3;
  a.x_bytes = (unsigned)(std::min<int64_t>(readable_bytes, fbytes + 4) + a.pad_off);
  a.y = reinterpret_cast<float*>(workspace);
  a.y_bstride = 0; a.Tp = Tp; a.relu = 1;
  const int nbh = (2 * Hp + 1 + 3) / 4, nbw = (2 * Wp + 1 + 15) / 16;
  set_bricks(a, Tp, nbh, nbw);
  // (every check before the first launch: a rejected call must not leave half an op enqueued)
  ADVHIP_REQUIRE(nbw <= MERGE_MAX_NBW, "conv3d u8 taps+pool233: pooled width %d above %d", Wp, MERGE_MAX_NBW * 8 - 1);
  const long long rows = (long long)d->B * a.tiles_n * 2 * Tp * Hp;
  ADVHIP_REQUIRE(rows < (1ll << 31), "conv3d u8 taps+pool233: too many output rows");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((stem_u8_tap_kernel<U8_TAPS_PER_TILE>), dim3((unsigned)a.tiles_m), dim3(256), 0, st, a);
  if (int rc = check_launch("conv3d u8 taps+pool233")) return rc;
  const unsigned gx = (unsigned)std::min<long long>(rows, 1 << 20), gy = (unsigned)((rows + gx - 1) / gx);
  hipLaunchKernelGGL(stem_pool_merge_kernel, dim3(gx, gy), dim3(256), (size_t)2 * nbw * 288 * sizeof(float), st, reinterpret_cast<const float*>(workspace), y, d->Cout, Tp,
                     Hp, Wp, nbh, nbw, a.tiles_n, FastDiv::make((unsigned)Wp), rows, ybs);
  return check_launch("stem_pool_merge");
}

extern "C" int advhip_conv3d_bn_act_maxpool211_f32(const advhip_conv3d_desc* d, const float* x, int64_t x_batch_stride,
                                                   const float* w_packed, const int32_t* ktab, const float* scale,
                                                   const float* shift, const float* residual, float* y,
                                                   int64_t y_batch_stride, void* stream) {
  if (int rc = validate(d)) return rc;
  ADVHIP_REQUIRE(x && w_packed && ktab && scale && shift && y, "conv3d+pool211: null pointer");
  ADVHIP_REQUIRE(d->kt == 1 && d->kh == 1 && d->kw == 1 && d->st == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 && d->ph == 0 && d->pw == 0,
                 "conv3d+pool211: 1x1x1 stride-1 convs only (k=%d,%d,%d)", d->kt, d->kh, d->kw);
  const Geometry g = geometry(d);
  ADVHIP_REQUIRE(g.K == g.Kpad, "conv3d+pool211: Cin=%d must be a multiple of 32", d->Cin);
  const int Tp = pool_out(g.To, 2, 2);
  ADVHIP_REQUIRE(Tp > 0, "conv3d+pool211: T=%d smaller than the temporal window", g.To);
  const long long y_dense = (long long)d->Cout * Tp * g.Ho * g.Wo;
  const long long ybs = y_batch_stride > 0 ? y_batch_stride : y_dense;
  ADVHIP_REQUIRE(ybs >= y_dense && (long long)d->B * ybs < (1ll << 32), "conv3d+pool211: bad y batch stride %lld (one pooled sample: %lld)", ybs, y_dense);
  ConvArgs a;
  if (int rc = fill_pool_args(a, d, g, x, x_batch_stride, w_packed, ktab, scale, shift, true)) return rc;
  a.res = residual; a.y = y; a.y_bstride = (int)ybs; a.Tp = Tp;
  set_bricks(a, Tp, 1, (a.Wo + 63) / 64);
  const dim3 grid((unsigned)(a.tiles_m * a.tiles_n));
  hipLaunchKernelGGL((conv3d_igemm_dma_kernel<128, 64, 16, false, 2, EPI_TPOOL>), grid, dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("conv3d+pool211");
}


// ---- "NT" product of two k-contiguous operands (weight gradients of the GEMM-shaped MGFN layers) --------------------------
namespace advhip {
constexpr long long GEMM_NT_CNT_BYTES = 65536;  // arrival counters: the first 64 KiB of the workspace, whatever the shape

static void gemm_nt_tile(int32_t M, int32_t N, int32_t splits, int32_t& tile, int& bm, int& bn) {
  if (tile == 0) tile = ((long long)((M + 127) / 128) * ((N + 63) / 64) * splits >= 1536) ? 2 : 1;
  bm = tile == 1 ? 64 : 128;
  bn = tile == 3 ? 128 : 64;
}

static int gemm_nt_launch(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, int64_t lda, int64_t ldb,
                          int64_t ldc, int32_t splits, int64_t slab_stride, int32_t tile, float* rowsum_a, void* workspace,
                          int64_t workspace_bytes, void* stream, int64_t rowsum_slab_stride = 0) {
  ADVHIP_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "gemm_nt: bad arguments");
  // (row pitches need not be multiples of 4 floats: a 16-byte LDS-DMA piece takes any 4-byte aligned address on gfx950, tools/probe/ --
  // the 2 049-float rows of the scorer's (positions, channels + magnitude) input are read as they are stored)
  ADVHIP_REQUIRE(K % 16 == 0 && ((uintptr_t)A & 3) == 0 && ((uintptr_t)B & 3) == 0,
                 "gemm_nt: K=%d must be a multiple of 16 and the operands 4-byte aligned", K);
  ADVHIP_REQUIRE(lda >= K && ldb >= K && ldc >= N, "gemm_nt: row pitch smaller than a row");
  const long long a_bytes = ((long long)(M - 1) * lda + K) * 4, b_bytes = ((long long)(N - 1) * ldb + K) * 4;
  ADVHIP_REQUIRE(a_bytes < 0xF0000000ll && b_bytes < 0xF0000000ll, "gemm_nt: operand above 3.75 GiB");
  ADVHIP_REQUIRE(splits >= 1 && splits <= K / 16 && (splits == 1 || workspace != nullptr || slab_stride >= (long long)(M - 1) * ldc + N),
                 "gemm_nt: bad split count %d", splits);
  ADVHIP_REQUIRE(tile >= 0 && tile <= 3, "gemm_nt: tile id %d (0 auto, 1 64x64, 2 128x64, 3 128x128)", tile);
  GemmKKArgs g;
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K;
  g.lda = (int)lda; g.ldb = (int)ldb; g.ldc = (int)ldc;
  g.a_bytes = (unsigned)a_bytes; g.b_bytes = (unsigned)b_bytes;
  g.splits = splits; g.slab = slab_stride; g.rowsum = rowsum_a;
  g.rs_slab = rowsum_slab_stride > 0 ? rowsum_slab_stride : M;
  ADVHIP_REQUIRE(g.rs_slab >= M, "gemm_nt: row-sum slab stride smaller than M");
  int bm, bn;
  gemm_nt_tile(M, N, splits, tile, bm, bn);
  g.tiles_m = (M + bm - 1) / bm;
  g.tiles_n = (N + bn - 1) / bn;
  const long long blocks = (long long)g.tiles_m * g.tiles_n * splits;
  ADVHIP_REQUIRE(blocks < (1ll << 31), "gemm_nt: too many tiles");
  g.part = nullptr; g.cnt = nullptr; g.part_bytes = 0;
  if (workspace != nullptr && splits > 1) {
    const long long ntiles = (long long)g.tiles_m * g.tiles_n;
    const long long cnt_bytes = GEMM_NT_CNT_BYTES;  // a FIXED head: launches of different shapes share one zero-filled workspace
    ADVHIP_REQUIRE(ntiles * 4 <= cnt_bytes, "gemm_nt: %lld output tiles above the %lld the counter block holds", ntiles, cnt_bytes / 4);
    const long long part_bytes = ntiles * splits * bm * bn * 4 + (long long)g.tiles_m * splits * bm * 4;
    ADVHIP_REQUIRE(part_bytes < 0xF0000000ll, "gemm_nt: partial tiles above 3.75 GiB");
    ADVHIP_REQUIRE(workspace_bytes >= cnt_bytes + part_bytes && ((uintptr_t)workspace & 255) == 0,
                   "gemm_nt: needs a 256-byte aligned workspace of %lld bytes (got %lld)", cnt_bytes + part_bytes, (long long)workspace_bytes);
    g.cnt = reinterpret_cast<unsigned*>(workspace);
    g.part = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + cnt_bytes);
    g.part_bytes = (unsigned)part_bytes;
  }
  const dim3 grid((unsigned)blocks);
  if (tile == 3) hipLaunchKernelGGL((gemm_kk_dma_kernel<128, 128>), grid, dim3(256), 0, (hipStream_t)stream, g);
  else if (tile == 2) hipLaunchKernelGGL((gemm_kk_dma_kernel<128, 64>), grid, dim3(256), 0, (hipStream_t)stream, g);
  else hipLaunchKernelGGL((gemm_kk_dma_kernel<64, 64>), grid, dim3(256), 0, (hipStream_t)stream, g);
  return check_launch("gemm_nt");
}
}  // namespace advhip

extern "C" int advhip_gemm_nt_f32(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, int64_t lda,
                                  int64_t ldb, int64_t ldc, int32_t splits, int64_t slab_stride, void* stream) {
  return advhip::gemm_nt_launch(A, B, C, M, N, K, lda, ldb, ldc, splits, slab_stride, 0, nullptr, nullptr, 0, stream);
}

extern "C" int advhip_gemm_nt_rowsum_f32(const float* A, const float* B, float* C, float* rowsum_a, int32_t M, int32_t N, int32_t K,
                                         int64_t lda, int64_t ldb, int64_t ldc, int32_t splits, int64_t slab_stride, int32_t tile,
                                         void* stream) {
  return advhip::gemm_nt_launch(A, B, C, M, N, K, lda, ldb, ldc, splits, slab_stride, tile, rowsum_a, nullptr, 0, stream);
}

// dst[i] = sum over s (in order) of src[s * stride + i]: the K slices of a small NT product and of its row sums, laid out as
// one [splits][n] matrix, in one pass
namespace advhip {
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n, int splits,
                                                        long long stride) {
  // eight slices are loaded before any is added (the adds stay in slice order): a chain of `splits` dependent loads otherwise
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float v = src[i];
    int sl = 1;
    for (; sl + 8 <= splits; sl += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = src[(long long)(sl + u) * stride + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; sl < splits; ++sl) v += src[(long long)sl * stride + i];
    dst[i] = v;
  }
}
}  // namespace advhip

extern "C" int advhip_gemm_nt_slabs_f32(const float* A, const float* B, float* slabs, int32_t M, int32_t N, int32_t K, int64_t lda,
                                        int64_t ldb, int32_t splits, int32_t tile, int32_t with_rowsum, void* stream) {
  const long long stride = (long long)M * N + (with_rowsum ? M : 0);
  return advhip::gemm_nt_launch(A, B, slabs, M, N, K, lda, ldb, N, splits, stride, tile, with_rowsum ? slabs + (long long)M * N : nullptr,
                                nullptr, 0, stream, stride);
}

extern "C" int advhip_gemm_nt_group_slabs_f32(const advhip_nt_item* items, int32_t n_items, int32_t K, int32_t splits, int64_t slab_stride,
                                              void* stream) {
  using namespace advhip;
  ADVHIP_REQUIRE(items && n_items > 0 && K > 0 && K % 16 == 0 && splits >= 1 && splits <= K / 16, "gemm_nt_group: bad arguments (K=%d, splits=%d)", K, splits);
  for (int base = 0; base < n_items; base += NT_GROUP_MAX) {
    NtGroupArgs ga;
    ga.n = std::min(NT_GROUP_MAX, n_items - base);
    ga.K = K; ga.splits = splits; ga.pad = 0; ga.slab = slab_stride;
    long long wg = 0;
    for (int i = 0; i < ga.n; ++i) {
      const advhip_nt_item& s = items[base + i];
      ADVHIP_REQUIRE(s.A && s.B && s.C && s.M > 0 && s.N > 0, "gemm_nt_group: item %d: bad arguments", base + i);
      ADVHIP_REQUIRE(s.lda % 4 == 0 && s.ldb % 4 == 0 && s.lda >= K && s.ldb >= K && ((uintptr_t)s.A & 15) == 0 && ((uintptr_t)s.B & 15) == 0,
                     "gemm_nt_group: item %d: operands must be 16-byte aligned with row pitches that are multiples of 4 and >= K", base + i);
      const long long a_bytes = ((long long)(s.M - 1) * s.lda + K) * 4, b_bytes = ((long long)(s.N - 1) * s.ldb + K) * 4;
      ADVHIP_REQUIRE(a_bytes < 0xF0000000ll && b_bytes < 0xF0000000ll && s.lda < (1ll << 31) && s.ldb < (1ll << 31), "gemm_nt_group: item %d: operand above 3.75 GiB", base + i);
      ADVHIP_REQUIRE(slab_stride >= (long long)s.M * s.N, "gemm_nt_group: slab stride smaller than item %d's output", base + i);
      NtGroupItem& t = ga.it[i];
      t.A = s.A; t.B = s.B; t.C = s.C; t.rowsum = s.rowsum;
      t.M = s.M; t.N = s.N; t.lda = (int)s.lda; t.ldb = (int)s.ldb;
      t.a_bytes = (unsigned)a_bytes; t.b_bytes = (unsigned)b_bytes;
      t.tiles_n = (s.N + 63) / 64;
      t.wg_begin = (int)wg;
      wg += (long long)((s.M + 63) / 64) * t.tiles_n * splits;
      ADVHIP_REQUIRE(wg < (1ll << 31), "gemm_nt_group: too many tiles");
    }
    for (int i = ga.n; i < NT_GROUP_MAX; ++i) ga.it[i] = NtGroupItem{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0u, 0u, 0, 0x7FFFFFFF};
    hipLaunchKernelGGL(gemm_kk_group_kernel, dim3((unsigned)wg), dim3(256), 0, (hipStream_t)stream, ga);
    if (int rc = check_launch("gemm_nt_group")) return rc;
  }
  return ADVHIP_OK;
}

extern "C" int advhip_sum_slabs_f32(const float* slabs, float* out, int64_t n, int32_t splits, int64_t stride, void* stream) {
  ADVHIP_REQUIRE(slabs && out && n > 0 && splits >= 1 && stride >= n, "sum_slabs: bad arguments");
  const int grid = (int)std::min<long long>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(advhip::sum_slabs_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, slabs, out, (long long)n, splits, (long long)stride);
  return check_launch("sum_slabs");
}

extern "C" int64_t advhip_gemm_nt_workspace_bytes(int32_t M, int32_t N, int32_t splits, int32_t tile) {
  if (M <= 0 || N <= 0 || splits <= 1 || tile < 0 || tile > 3) return 0;
  int bm, bn;
  advhip::gemm_nt_tile(M, N, splits, tile, bm, bn);
  const long long tm = (M + bm - 1) / bm, tn = (N + bn - 1) / bn;
  return advhip::GEMM_NT_CNT_BYTES + tm * tn * splits * bm * bn * 4 + tm * splits * bm * 4;
}

extern "C" int advhip_gemm_nt_reduced_f32(const float* A, const float* B, float* C, float* rowsum_a, int32_t M, int32_t N, int32_t K,
                                          int64_t lda, int64_t ldb, int64_t ldc, int32_t splits, int32_t tile, void* workspace,
                                          int64_t workspace_bytes, void* stream) {
  ADVHIP_REQUIRE(splits == 1 || workspace != nullptr, "gemm_nt_reduced: K slices need a workspace");
  return advhip::gemm_nt_launch(A, B, C, M, N, K, lda, ldb, ldc, splits, 0, tile, rowsum_a, workspace, workspace_bytes, stream);
}
