// Probe (one wave per launch): which address alignments does LDS-DMA (`buffer_load_{dword,dwordx3,dwordx4} ... lds`) accept on gfx950?
// Each case copies from a byte buffer at a chosen byte offset into LDS and writes the LDS image out; the host compares.
//   hipcc --offload-arch=gfx950 -O2 dma_align_probe.hip -o dma_align_probe && ./dma_align_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using lds_ptr_t = __attribute__((address_space(3))) void*;

template <int BYTES>
__global__ void probe(const unsigned char* src, unsigned nbytes, unsigned byte_off, unsigned lane_stride, unsigned* out) {
  __shared__ unsigned s[64 * 4];
  for (int i = threadIdx.x; i < 256; i += 64) s[i] = 0xDEADBEEFu;
  __syncthreads();
  const auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, nbytes, 0x00020000);
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned vo = byte_off + threadIdx.x * lane_stride;
  if constexpr (BYTES == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)s, 4, vo, 0, 0, 0);
  else if constexpr (BYTES == 12) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)s, 12, vo, 0, 0, 0);
  else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)s, 16, vo, 0, 0, 0);
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = s[i];
}

int main() {
  const unsigned N = 1 << 16;
  std::vector<unsigned char> h(N);
  for (unsigned i = 0; i < N; ++i) h[i] = (unsigned char)((i * 7 + 3) & 0xFF);
  unsigned char* d; unsigned* o;
  (void)hipMalloc(&d, N); (void)hipMalloc(&o, 1024);
  (void)hipMemcpy(d, h.data(), N, hipMemcpyHostToDevice);
  struct Case { int bytes; unsigned off, stride; };
  const Case cases[] = {{4, 0, 4}, {4, 1, 6}, {4, 2, 6}, {4, 3, 3}, {16, 0, 16}, {16, 8, 16}, {16, 4, 16}, {16, 4, 8}, {12, 0, 12}, {12, 4, 12}, {12, 4, 8}, {16, 2, 6}};
  for (const Case& c : cases) {
    printf("case bytes=%d off=%u lane_stride=%u ... ", c.bytes, c.off, c.stride); fflush(stdout);
    (void)hipMemset(o, 0, 1024);
    if (c.bytes == 4) hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64), 0, 0, d, N, c.off, c.stride, o);
    else if (c.bytes == 12) hipLaunchKernelGGL(probe<12>, dim3(1), dim3(64), 0, 0, d, N, c.off, c.stride, o);
    else hipLaunchKernelGGL(probe<16>, dim3(1), dim3(64), 0, 0, d, N, c.off, c.stride, o);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("ERROR %s\n", hipGetErrorString(e)); return 1; }
    unsigned got[256];
    (void)hipMemcpy(got, o, 1024, hipMemcpyDeviceToHost);
    // expected LDS image: lane l's BYTES bytes land at LDS byte l*BYTES (dword: l*4; x3: l*12; x4: l*16)
    int bad = 0;
    for (int l = 0; l < 64 && !bad; ++l)
      for (int b = 0; b < c.bytes; ++b) {
        const unsigned char want = h[c.off + l * c.stride + b];
        const unsigned char have = reinterpret_cast<unsigned char*>(got)[l * c.bytes + b];
        if (want != have) { bad = 1; printf("MISMATCH lane %d byte %d: want %02x have %02x (lds dword %08x) ", l, b, want, have, got[(l * c.bytes + b) / 4]); break; }
      }
    printf("%s\n", bad ? "-> differs" : "OK"); fflush(stdout);
  }
  return 0;
}
