// Probe (one wave): does `ds_read_b128` accept byte addresses that are not 16-byte (or even 4-byte) aligned on gfx950?
//   hipcc --offload-arch=gfx950 -O2 lds_unaligned_probe.hip -o lds_unaligned_probe && ./lds_unaligned_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__global__ void probe(unsigned byte_off, unsigned lane_stride, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned char s[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) s[i] = (unsigned char)((i * 7 + 3) & 0xFF);
  __syncthreads();
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)s + byte_off + threadIdx.x * lane_stride;
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}

int main() {
  unsigned* o;
  (void)hipMalloc(&o, 1024);
  const unsigned cases[][2] = {{0, 16}, {4, 16}, {8, 16}, {1, 16}, {2, 16}, {3, 64}, {1, 64}};
  for (auto& c : cases) {
    printf("ds_read_b128 off=%u lane_stride=%u ... ", c[0], c[1]); fflush(stdout);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, c[0], c[1], o);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("ERROR %s\n", hipGetErrorString(e)); return 1; }
    unsigned char got[1024];
    (void)hipMemcpy(got, o, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64 && !bad; ++l)
      for (int b = 0; b < 16; ++b) {
        const unsigned i = c[0] + l * c[1] + b;
        if (got[l * 16 + b] != (unsigned char)((i * 7 + 3) & 0xFF)) { bad = 1; printf("MISMATCH lane %d byte %d ", l, b); break; }
      }
    printf("%s\n", bad ? "-> differs" : "OK"); fflush(stdout);
  }
  return 0;
}
