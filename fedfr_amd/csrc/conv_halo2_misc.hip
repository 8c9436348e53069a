// tuning variants of the 3x3 halo conv (64-wide N tiles, 8 waves per tile; options halo_bn64 / halo_waves) and the
// fused epilogue on 64-wide tiles.  Kept out of the hot instantiations' translation units (gemm_dev.h).
#include "conv_halo2_impl.h"
int launch_conv_halo2_misc(GemmNT p, int bn64, int waves8, hipStream_t st) {
  const bool fused = p.bpart != nullptr;
  if (p.W == 14) {
    if (bn64) return fused ? launch_halo2<64, 14, 2, true>(p, st) : launch_halo2<64, 14, 2, false>(p, st);
    if (waves8) return fused ? launch_halo2<128, 14, 4, true>(p, st) : launch_halo2<128, 14, 4, false>(p, st);
  } else {
    if (bn64) return fused ? launch_halo2<64, 28, 2, true>(p, st) : launch_halo2<64, 28, 2, false>(p, st);
    if (waves8) return fused ? launch_halo2<128, 28, 4, true>(p, st) : launch_halo2<128, 28, 4, false>(p, st);
  }
  FEDFR_REQUIRE(false, "conv3x3_halo2_misc: no variant selected");
  return FEDFR_OK;
}
