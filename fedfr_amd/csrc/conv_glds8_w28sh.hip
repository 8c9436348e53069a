// LDS-DMA 3x3 conv, 28x28 maps, 8 waves, two image tiles per workgroup, with the train-mode BatchNorm (+PReLU) behind it applied by the launch
// itself (GemmNT::hout, conv_glds_impl.h HF) -- alone in its translation unit (gemm_dev.h)
#include "conv_glds_impl.h"
int launch_conv_glds8_w28_handoff(GemmNT p, hipStream_t st) { return launch_glds<28, 7, 40, 4, false, 128, false, 2, true>(p, st); }
