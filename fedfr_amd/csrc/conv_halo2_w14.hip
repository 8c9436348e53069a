// plain (fwd / dgrad) 3x3 halo conv, 14x14 maps, 128x128 tiles, 4 waves -- alone in its translation unit (gemm_dev.h)
#include "conv_halo2_impl.h"
int launch_conv_halo2_w14(GemmNT p, hipStream_t st) { return launch_halo2<128, 14, 2, false>(p, st); }
