// Effective shader clock under load: a one-wave kernel that sleeps for a while and reports how many shader-clock ticks (s_memtime) and
// constant-rate ticks (s_memrealtime, 100 MHz) went by.  Launched on a side stream while the extract stream runs
// (tools/clock_under_load.py): is the 2.4 GHz the roofline is priced against the clock the MFMA-heavy stream actually gets?
#include <hip/hip_runtime.h>
__global__ void clock_probe_kernel(unsigned long long* out, int sleeps) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < sleeps; ++i) __builtin_amdgcn_s_sleep(127);
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = r1 - r0;
  }
}
extern "C" int clock_probe_launch(unsigned long long* out, int blocks, int sleeps, void* stream) {
  hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, out, sleeps);
  return (int)hipGetLastError();
}
