// Shared helpers for the gfx950 kernels behind include/advhip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/advhip.h"

namespace advhip {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return ADVHIP_ELAUNCH;
  }
  return ADVHIP_OK;
}

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// MI355X: 8 XCDs, blocks are dealt round-robin over them.  Map the hardware block id to a
// logical tile id so that each XCD works on a contiguous range of logical tiles (neighbouring
// tiles share operand panels -> same L2).  Bijective for any grid size.  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  constexpr int NX = 8;
  const int q = nwg >> 3, r = nwg & 7;  // unsigned-style shifts: no signed-division fix-ups
  const int xcd = bid & 7, k = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}

using f32x4 = __attribute__((ext_vector_type(4))) float;

}  // namespace advhip

#define ADVHIP_REQUIRE(cond, ...)        \
  do {                                   \
    if (!(cond)) {                       \
      advhip::set_error(__VA_ARGS__);    \
      return ADVHIP_EINVAL;              \
    }                                    \
  } while (0)
