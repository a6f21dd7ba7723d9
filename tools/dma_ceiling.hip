// Micro-benchmark: the conv kernel's loop shape (64x64x16 tile, 4 waves, 3-deep LDS ring, one barrier per k-tile,
// 16 fp32 MFMAs per wave per k-tile) fed by LDS-DMA in different instruction shapes, to price the memory side:
//   mode 0: no loads                                    (the ceiling of the loop itself)
//   mode 1: A as 4 x buffer_load_dword...lds per wave (64 consecutive floats each) + B as 1 x dwordx4   = the product kernel
//   mode 2: A as 1 x dwordx4 (1 KiB contiguous per wave) + B as 1 x dwordx4
//   mode 3: A as 1 x dwordx4 in a channels-last gather shape (8 rows x 128 B, row pitch `pitch`) + B as 1 x dwordx4
//   mode 4: only the A loads of mode 1;  mode 5: only the A load of mode 3;  mode 6: only B
//   mode 7: mode 1 with the address stream of layer1.1.conv1 in NCDHW (k-row = channel plane 48 KB apart, 3 temporal taps,
//           189 m-tiles per sample) -- every k-row of a tile in a different page;  mode 8: the same rows packed densely
//   mode 9: mode 7 with the exact byte offsets (48400-byte planes, 12100-byte taps: rows only 4-byte aligned)
//   mode 10: mode 9 with the offsets from a table through s_load_dwordx8 + wait per k-tile, and the five loads
//            issued one per MFMA group instead of all after the barrier (the product kernel's issue order)
// A bytes walk through a `footprint`-byte buffer (small = L2 resident, large = HBM stream); B is a 1 MiB buffer
// every workgroup shares.  All addresses are 16-byte aligned and stay inside the buffers (modulo arithmetic).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -o tools/_build/dma_ceiling tools/dma_ceiling.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <utility>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using lds_ptr_t = __attribute__((address_space(3))) void*;

constexpr int STAGE = 16 * 128;  // floats: A 16x64 + B 16x64

template <int MODE, int SHAPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8)))
void loop_kernel(const float* __restrict__ abuf, unsigned a_mask, const float* __restrict__ bbuf, float* out, int ktiles, int pitch, const int2* __restrict__ tab, unsigned long long* stamps) {
  __shared__ __attribute__((aligned(16))) float smem[3 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 3 * STAGE; i += 256) smem[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  constexpr bool HAS_A = MODE != 0 && MODE != 6;
  constexpr bool HAS_B = MODE == 1 || MODE == 2 || MODE == 3 || MODE >= 6;
  constexpr bool A_DWORD = MODE == 1 || MODE == 4 || MODE >= 7;
  constexpr int NA = HAS_A ? (A_DWORD ? 4 : 1) : 0, NB = HAS_B ? 1 : 0;
  const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(abuf), 0, a_mask + 1u, 0x00020000);
  const auto rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bbuf), 0, 1u << 20, 0x00020000);
  // this workgroup's A stream starts at its own 64-row panel
  const unsigned a_base = (unsigned)blockIdx.x * 64u * (unsigned)pitch;
  unsigned a_lane;
  if constexpr (A_DWORD) a_lane = (unsigned)lane * 4u;                               // 64 consecutive floats of one k-row
  else if constexpr (MODE == 2) a_lane = (unsigned)lane * 16u;                        // 1 KiB contiguous
  else a_lane = (unsigned)(wave * 8 + (lane >> 3)) * (unsigned)pitch + (lane & 7) * 16u;  // 8 rows x 128 B
  const unsigned b_lane = (unsigned)(lane >> 4) * 1024u + (lane & 15) * 16u;

  int ent[8];
  auto issue = [&](int kt, int stage, int part = -1) {
    float* As = smem + stage * STAGE;
    float* Bs = As + 16 * 64;
    if constexpr (HAS_A) {
      if constexpr (A_DWORD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (part >= 0 && part != j) continue;
          // k-row (wave*4+j) of tile kt: rows are `pitch` bytes apart (a channel plane apart, in the product kernel)
          const unsigned r = (unsigned)(kt * 16 + wave * 4 + j);
          unsigned so;
          if constexpr (MODE == 7) so = ((blockIdx.x / 189u) * (256u * 48384u) + (blockIdx.x % 189u) * 256u + (r / 3u) * 48384u + (r % 3u) * 12032u) & a_mask & ~255u;
          else if constexpr (MODE == 8) so = ((blockIdx.x * 768u + r) * 256u) & a_mask & ~255u;
          else if constexpr (MODE >= 9) so = ((blockIdx.x / 189u) % 32u) * (256u * 48400u) + (blockIdx.x % 189u) * 256u + (MODE == 10 ? (unsigned)ent[2 * j] : (r / 3u) * 48400u + (r % 3u) * 12100u);
          else so = (a_base + r * 4096u * 3u) & a_mask & ~255u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(As + (wave * 4 + j) * 64), 4, a_lane, so, 0, 0);
        }
      } else if constexpr (MODE == 2) {
        const unsigned so = (a_base + (unsigned)(kt * 4 + wave) * 4096u * 3u) & a_mask & ~1023u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(As + wave * 256), 16, a_lane, so, 0, 0);
      } else {
        // 32 rows (of the 64-row panel; the other 32 belong to the second half k-tile) x 128 B at channel offset kt*128
        const unsigned so = (a_base + (unsigned)(kt >> 1) * 128u % (unsigned)pitch + (kt & 1) * 32u * (unsigned)pitch) & a_mask & ~127u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(As + wave * 256), 16, a_lane & (a_mask >> 1), so & (a_mask >> 1), 0, 0);
      }
    }
    if (HAS_B && (part < 0 || part == 0)) {
      const unsigned so = ((unsigned)(kt * 16 + wave * 4) * 1024u) & ((1u << 20) - 1u) & ~4095u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(Bs + wave * 256), 16, b_lane, so, 0, 0);
    }
  };

  auto load_ent = [&](int kt) {
    if constexpr (MODE == 10) {
      __attribute__((ext_vector_type(8))) int v;
      asm volatile("s_load_dwordx8 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(tab), "s"((kt * 16 + wave * 4) * 8) : "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) ent[i] = v[i];
    }
  };
  const int wm = wave >> 1, wn = wave & 1, li = lane & 15, lg = lane >> 4;
  const unsigned base = (unsigned)(size_t)(lds_ptr_t)smem;
  // SHAPE 0: v_mfma_f32_16x16x4_f32, 2x2 fragments per wave;  SHAPE 1: v_mfma_f32_32x32x2_f32, one 32x32 fragment per wave
  const unsigned a_addr = SHAPE == 0 ? base + (unsigned)(lg * 64 + wm * 32 + 2 * li) * 4u : base + (unsigned)((lane >> 5) * 64 + wm * 32 + (lane & 31)) * 4u;
  const unsigned b_addr = SHAPE == 0 ? base + (unsigned)(16 * 64 + lg * 64 + wn * 32 + 2 * li) * 4u
                                     : base + (unsigned)(16 * 64 + (lane >> 5) * 64 + wn * 32 + (lane & 31)) * 4u;
  f32x4 acc[2][2] = {};
  f32x16 acc32 = {};
  f32x2 fa[2], fb[2];
  float ga[2], gb[2];
  load_ent(0);
  issue(0, 0);
  load_ent(1);
  issue(1, 1);
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  int stage = 0;
  for (int kt = 0; kt < ktiles; ++kt) {
    if constexpr (NA + NB > 0) {
      if (kt + 1 < ktiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");
    const bool pre = kt + 2 < ktiles;
    const int pstage = stage == 0 ? 2 : stage - 1;
    if constexpr (MODE == 10) { if (pre) load_ent(kt + 2); }
    else if (pre) issue(kt + 2, pstage);
    const unsigned aa = a_addr + stage * (STAGE * 4), ba = b_addr + stage * (STAGE * 4);
    if constexpr (SHAPE == 0) {
      asm volatile("ds_read_b64 %0, %1" : "=v"(fa[0]) : "v"(aa));
      asm volatile("ds_read_b64 %0, %1" : "=v"(fb[0]) : "v"(ba));
      auto body = [&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        if constexpr (ks < 3) {
          asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fa[(ks + 1) & 1]) : "v"(aa), "n"((ks + 1) * 4 * 64 * 4));
          asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fb[(ks + 1) & 1]) : "v"(ba), "n"((ks + 1) * 4 * 64 * 4));
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[ks & 1]), "+v"(fb[ks & 1]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[ks & 1]), "+v"(fb[ks & 1]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE == 10) { if (pre) issue(kt + 2, pstage, ks); }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ks & 1][i], fb[ks & 1][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      body(std::integral_constant<int, 0>{});
      body(std::integral_constant<int, 1>{});
      body(std::integral_constant<int, 2>{});
      body(std::integral_constant<int, 3>{});
    } else {
      asm volatile("ds_read_b32 %0, %1" : "=v"(ga[0]) : "v"(aa));
      asm volatile("ds_read_b32 %0, %1" : "=v"(gb[0]) : "v"(ba));
      auto body = [&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;  // 8 k-steps of 2
        if constexpr (ks < 7) {
          asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(ga[(ks + 1) & 1]) : "v"(aa), "n"((ks + 1) * 2 * 64 * 4));
          asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(gb[(ks + 1) & 1]) : "v"(ba), "n"((ks + 1) * 2 * 64 * 4));
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ga[ks & 1]), "+v"(gb[ks & 1]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ga[ks & 1]), "+v"(gb[ks & 1]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE == 10 && (ks & 1) == 0) { if (pre) issue(kt + 2, pstage, ks / 2); }
        acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[ks & 1], gb[ks & 1], acc32, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      [&]<int... I>(std::integer_sequence<int, I...>) { (body(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, 8>{});
    }
    stage = stage == 2 ? 0 : stage + 1;
  }
  if (stamps != nullptr && tid == 0) {
    stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  for (int i = 0; i < 16; ++i) s += acc32[i];
  if (s == 12345.678f) out[tid] = s;  // keep the accumulators live
}

static const int2* g_tab;
static unsigned long long* g_stamps;
__global__ void fill_random(float* p, size_t n, unsigned seed, int relu) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    float v = (float)(int)h * (1.0f / 2147483648.0f);
    p[i] = relu ? fmaxf(v, 0.f) : v;
  }
}

template <int M, int SH>
static void run(const float* a, unsigned a_mask, const float* b, float* out, int blocks, int ktiles, int pitch, int reps) {
  static const char* names[] = {"no loads", "A 4 x dword + B 1 x dwordx4 per wave (product mix)", "A 1 x dwordx4 contiguous + B 1 x dwordx4",
                                "A 1 x dwordx4 channels-last rows + B 1 x dwordx4", "A 4 x dword only", "A 1 x dwordx4 channels-last rows only",
                                "B 1 x dwordx4 only", "product mix, NCDHW address stream (48 KB k-row pitch)", "product mix, rows packed densely",
                                "product mix, exact NCDHW offsets (4-byte aligned)", "... + offset table via s_load, loads between MFMAs"};
  if (M >= 9 && ktiles > 48) ktiles = 48;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  loop_kernel<M, SH><<<blocks, 256>>>(a, a_mask, b, out, ktiles, pitch, g_tab, g_stamps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) loop_kernel<M, SH><<<blocks, 256>>>(a, a_mask, b, out, ktiles, pitch, g_tab, g_stamps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  double flop = (double)blocks * 4 * ktiles * 16.0 * 2048.0;
  double gbs = (double)blocks * ktiles * 8192.0 / ms / 1e6;
  // in-kernel clock: cycles / (100 MHz ticks) over the k-loop, summed over all workgroups of the last launch
  static unsigned long long h[2 * 12288];
  hipMemcpy(h, g_stamps, sizeof(h), hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int i = 0; i < blocks && i < 12288; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
  printf("mode %2d %s %-52s %8.3f ms %7.1f TFLOP/s %5.1f %%  (%5.0f GB/s tile loads)  clock %.2f GHz, %.0f cycles per k-tile\n", M, SH ? "32x32x2" : "16x16x4",
         names[M], ms, flop / ms / 1e9, 100.0 * flop / ms / 1e9 / 157.3, gbs, cyc / rt * 0.1, cyc / blocks / ktiles);
  fflush(stdout);
}

template <int SH>
static void dispatch(int mode, const float* a, unsigned mask, const float* b, float* out, int blocks, int ktiles, int pitch, int reps) {
  switch (mode) {
#define CASE(M) case M: run<M, SH>(a, mask, b, out, blocks, ktiles, pitch, reps); break;
    CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10)
#undef CASE
    default: break;
  }
}

// usage: dma_ceiling <data 0|1|2> <reps> <shape 0|1> mode [mode ...]     (data: 0 constant, 1 random, 2 random with A post-ReLU)
int main(int argc, char** argv) {
  const int random_data = argc > 1 ? atoi(argv[1]) : 0;
  const int reps = argc > 2 ? atoi(argv[2]) : 40;
  const int shape = argc > 3 ? atoi(argv[3]) : 0;
  const int ktiles = 48, footprint_mb = 512, pitch = 1024;
  const int blocks = 256 * 6 * 8;
  const size_t abytes = (size_t)footprint_mb << 20;
  float *a, *b, *out;
  hipMalloc(&a, abytes);
  hipMalloc(&b, 1 << 20);
  hipMalloc(&out, 4096);
  if (random_data) {
    fill_random<<<4096, 256>>>(a, abytes / 4, 1u, random_data == 2);
    fill_random<<<256, 256>>>(b, (1 << 20) / 4, 2u, 0);
    hipDeviceSynchronize();
  } else {
    hipMemset(a, 0x3c, abytes);
    hipMemset(b, 0x3c, 1 << 20);
  }
  hipMalloc(&g_stamps, 2 * 12288 * 8);
  const unsigned mask = (unsigned)(abytes - 1);
  {
    static int2 h[768 + 64];
    for (int r = 0; r < 768 + 64; ++r) h[r] = make_int2((r / 3 % 256) * 48400 + (r % 3) * 12100, 0);
    int2* d;
    hipMalloc(&d, sizeof(h));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    g_tab = d;
  }
  printf("ktiles %d, A footprint %d MiB, %d workgroups, %d launches per line, data %s\n", ktiles, footprint_mb, blocks, reps,
         random_data == 0 ? "constant" : random_data == 1 ? "random" : "random, A post-ReLU");
  for (int i = 4; i < argc; ++i) {
    if (shape) dispatch<1>(atoi(argv[i]), a, mask, b, out, blocks, ktiles, pitch, reps);
    else dispatch<0>(atoi(argv[i]), a, mask, b, out, blocks, ktiles, pitch, reps);
  }
  hipFree(a);
  hipFree(b);
  hipFree(out);
  return 0;
}
