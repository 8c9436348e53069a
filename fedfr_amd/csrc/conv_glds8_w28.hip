// LDS-DMA 3x3 conv, 28x28 maps, 8 waves (two per SIMD; 40 pieces so that every wave owns 5) -- alone in its translation unit
#include "conv_glds_impl.h"
int launch_conv_glds8_w28(GemmNT p, hipStream_t st) { return launch_glds<28, 7, 40, 4, false>(p, st); }
