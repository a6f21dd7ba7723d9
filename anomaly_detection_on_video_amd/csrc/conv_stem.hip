// Dedicated stem convolution (Cin=3, k(5,7,7), s(2,2,2), p(2,3,3)) -- placeholder until the LDS
// halo-tile kernel lands; the generic implicit-GEMM path handles the stem meanwhile.
#include "common.h"

namespace advhip {
int launch_stem(const advhip_conv3d_desc*, const float*, const float*, const float*, const float*, float*,
                hipStream_t) {
  set_error("conv3d: ADVHIP_ALGO_STEM not built yet");
  return ADVHIP_EINVAL;
}
}  // namespace advhip
