// LDS-DMA 3x3 forward convs with the input's BatchNorm(+PReLU) applied to the LDS image (and the normalised activation written
// back for the weight-gradient GEMM): 14x14 / 28x28 (8 waves) and the 56x56 layers -- own translation unit (gemm_dev.h)
#include "conv_glds_impl.h"
int launch_conv_glds_x(GemmNT p, hipStream_t st) {
  if (p.W == 14) return launch_glds<14, 14, 32, 4, false, 128, false, true>(p, st);
  if (p.W == 28) return launch_glds<28, 7, 40, 4, false, 128, false, true>(p, st);
  if (p.W == 56 && p.C == 64 && p.N == 64) return launch_glds<56, 4, 48, 2, false, 64, true, true>(p, st);
  if (p.W == 56 && p.C == 64 && p.N == 128) return launch_glds<56, 4, 48, 2, false, 128, true, true>(p, st);
  FEDFR_REQUIRE(false, "conv3x3_glds_x: unsupported shape");
  return FEDFR_OK;
}
