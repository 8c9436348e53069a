// LDS-DMA 3x3 conv, 28x28 maps, 8 waves, TWO image tiles per workgroup: the forward convs that leave BatchNorm statistics (one partial row
// per workgroup = 256 rows at B = 128 instead of 1024, few enough for the channel-sliced BatchNorm pass to reduce itself) -- alone in its
// translation unit (gemm_dev.h)
#include "conv_glds_impl.h"
int launch_conv_glds8_w28_stats(GemmNT p, hipStream_t st) { return launch_glds<28, 7, 40, 4, false, 128, false, 2>(p, st); }
