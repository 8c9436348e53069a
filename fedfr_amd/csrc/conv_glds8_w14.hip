// LDS-DMA 3x3 conv, 14x14 maps, 8 waves (two per SIMD) -- alone in its translation unit (gemm_dev.h)
#include "conv_glds_impl.h"
int launch_conv_glds8_w14(GemmNT p, hipStream_t st) { return launch_glds<14, 14, 32, 4, false>(p, st); }
