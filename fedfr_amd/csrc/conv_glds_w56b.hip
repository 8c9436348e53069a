// LDS-DMA 3x3 conv, 56x56 maps, the two mixed-width layers of the stage-2 head: 64 -> 128 (one input chunk, 128-wide output tile,
// 4 waves) and its dgrad 128 -> 64 (two chunks, 64-wide output tile, 4 waves) -- own translation unit (gemm_dev.h)
#include "conv_glds_impl.h"
int launch_conv_glds_w56_c64_n128(GemmNT p, hipStream_t st) { return launch_glds<56, 4, 48, 2, false, 128, true>(p, st); }
int launch_conv_glds_w56_c128_n64(GemmNT p, hipStream_t st) { return launch_glds<56, 4, 48, 2, false, 64, false>(p, st); }
