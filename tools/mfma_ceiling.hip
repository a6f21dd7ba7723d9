// Micro-benchmark: what fraction of the fp32 MFMA peak does the conv kernel's inner-loop *shape* reach on
// gfx950 when nothing else is in the way?  Variants strip the loop down to
//   0: bare v_mfma_f32_16x16x4_f32 stream (4 accumulators per wave, as the 64x64 tile's 2x2 fragments)
//   1: + the fragment reads (one ds_read_b64 per operand per k-step, double buffered, counted lgkmcnt)
//   2: + one s_barrier per k-tile (4 k-steps)
//   3: as 2, but short workgroups (48 k-tiles each, many workgroups) = dispatch / drain included
//   4: as 3 + a coalesced 64x64 fp32 tile store per workgroup
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_build/mfma_ceiling tools/mfma_ceiling.hip
// Run on the GPU box: tools/_build/mfma_ceiling [waves_per_simd]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

template <int VARIANT>
__global__ __launch_bounds__(256) void loop_kernel(float* out, int ktiles, int lds_pad_floats) {
  extern __shared__ float smem[];  // dynamic size sets the residency
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 3 * 16 * 128; i += 256) smem[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1, li = lane & 15, lg = lane >> 4;
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  const unsigned a_addr = base + (unsigned)(lg * 64 + wm * 32 + 2 * li) * 4u;
  const unsigned b_addr = base + (unsigned)(16 * 64 + lg * 64 + wn * 32 + 2 * li) * 4u;
  f32x4 acc[2][2] = {};
  f32x2 fa[2] = {{1.f, 2.f}, {1.f, 2.f}}, fb[2] = {{3.f, 4.f}, {3.f, 4.f}};
  int stage = 0;
  for (int kt = 0; kt < ktiles; ++kt) {
    if constexpr (VARIANT >= 2) asm volatile("s_barrier" ::: "memory");
    const unsigned aa = a_addr + stage * (16 * 128 * 4), ba = b_addr + stage * (16 * 128 * 4);
    if constexpr (VARIANT >= 1) {
      asm volatile("ds_read_b64 %0, %1" : "=v"(fa[0]) : "v"(aa));
      asm volatile("ds_read_b64 %0, %1" : "=v"(fb[0]) : "v"(ba));
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if constexpr (VARIANT >= 1) {
        if (ks < 3) {
          asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fa[(ks + 1) & 1]) : "v"(aa), "n"(0));
          asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fb[(ks + 1) & 1]) : "v"(ba), "n"(0));
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[ks & 1]), "+v"(fb[ks & 1]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[ks & 1]), "+v"(fb[ks & 1]));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ks & 1][i], fb[ks & 1][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    stage = stage == 2 ? 0 : stage + 1;
  }
  if constexpr (VARIANT == 4) {
    float* o = out + (size_t)blockIdx.x * 4096;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[((wn * 2 + j) * 16 + lg * 4 + r) * 64 + (wm * 2 + i) * 16 + li] = acc[i][j][r];
  } else {
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) out[tid] = s;  // keep the accumulators live
  }
}

template <int V>
static void run(const char* what, int blocks, int ktiles, int waves_per_simd, float* out) {
  // dynamic LDS so that exactly `waves_per_simd` 256-thread workgroups fit per CU (160 KB LDS)
  size_t lds = (160 * 1024 / waves_per_simd) & ~1023u;
  if (lds > 64 * 1024) lds = 64 * 1024;
  if (lds < 3 * 16 * 128 * 4) lds = 3 * 16 * 128 * 4;
  hipFuncSetAttribute((const void*)loop_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  loop_kernel<V><<<blocks, 256, lds>>>(out, ktiles, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) loop_kernel<V><<<blocks, 256, lds>>>(out, ktiles, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  double flop = (double)blocks * 4 /*waves*/ * ktiles * 16.0 /*mfma*/ * 2048.0;
  printf("variant %d %-58s blocks %6d ktiles %5d lds %6zu: %8.3f ms  %7.1f TFLOP/s  %5.1f %% of 157.3\n", V, what, blocks, ktiles, lds, ms,
         flop / ms / 1e9, 100.0 * flop / ms / 1e9 / 157.3);
  fflush(stdout);
}

int main(int argc, char** argv) {
  int wps = argc > 1 ? atoi(argv[1]) : 6;
  float* out;
  const int slots = 256 * wps;
  hipMalloc(&out, (size_t)slots * 40 * 4096 * 4);
  run<0>("bare MFMA stream, one long workgroup per slot", slots, 48 * 40, wps, out);
  run<1>("+ fragment ds_reads", slots, 48 * 40, wps, out);
  run<2>("+ s_barrier per k-tile", slots, 48 * 40, wps, out);
  run<3>("short workgroups (48 k-tiles), 40 per slot", slots * 40, 48, wps, out);
  run<4>("short workgroups + 64x64 tile store", slots * 40, 48, wps, out);
  run<3>("short workgroups (8 k-tiles), 240 per slot", slots * 240, 8, wps, out);
  hipFree(out);
  return 0;
}
