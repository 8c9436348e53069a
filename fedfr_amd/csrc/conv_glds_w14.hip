// LDS-DMA 3x3 conv, 14x14 maps (whole-image tiles, 256-row LDS image = 32 pieces) -- alone in its translation unit (gemm_dev.h)
#include "conv_glds_impl.h"
int launch_conv_glds_w14(GemmNT p, hipStream_t st) { return launch_glds<14, 14, 32, 2, false>(p, st); }
