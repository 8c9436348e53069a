// LDS-DMA 3x3 dgrad, 28x28 maps, 8 waves, TWO image tiles per workgroup, with the BatchNorm-backward reduction of the layer in front in
// its epilogue (the BatchNorm input tile is fetched behind the K loop: this instantiation sits at the register limit) -- ONE partial row per
// workgroup = 256 rows at B = 128, few enough for the channel-sliced apply pass to reduce itself.  Alone in its translation unit (gemm_dev.h)
#include "conv_glds_impl.h"
int launch_conv_glds8_fused_w28s(GemmNT p, hipStream_t st) { return launch_glds<28, 7, 40, 4, true, 128, false, 2>(p, st); }
