// LDS-DMA 3x3 conv, 28x28 maps (7-row tiles, 270-row LDS image -> 36 pieces) -- alone in its translation unit (gemm_dev.h)
#include "conv_glds_impl.h"
int launch_conv_glds_w28(GemmNT p, hipStream_t st) { return launch_glds<28, 7, 36, 2, false>(p, st); }
