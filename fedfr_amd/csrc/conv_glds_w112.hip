// LDS-DMA 3x3 conv, 112x112 maps, 64 -> 64 channels (2-row tiles = 224 pixels, one channel chunk, 64-wide output tile, 4 waves)
// -- alone in its translation unit (gemm_dev.h)
#include "conv_glds_impl.h"
int launch_conv_glds_w112(GemmNT p, hipStream_t st) { return launch_glds<112, 2, 60, 2, false, 64, true>(p, st); }
