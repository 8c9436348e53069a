// LDS-DMA 3x3 conv, 56x56 maps, 64 -> 64 channels (4-row tiles = 224 pixels, one channel chunk, 64-channel output tile, 4 waves;
// 80 KB of LDS -> two workgroups per CU) -- alone in its translation unit (gemm_dev.h)
#include "conv_glds_impl.h"
int launch_conv_glds_w56(GemmNT p, hipStream_t st) { return launch_glds<56, 4, 48, 2, false, 64, true>(p, st); }
