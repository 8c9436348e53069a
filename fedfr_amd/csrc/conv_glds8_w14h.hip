// LDS-DMA 3x3 conv, 14x14 maps, 8 waves, with the train-mode BatchNorm (+PReLU) behind it applied by the launch itself: the workgroups hand
// their statistics rows to each other (GemmNT::hout, conv_glds_impl.h HF) -- alone in its translation unit (gemm_dev.h)
#include "conv_glds_impl.h"
int launch_conv_glds8_w14_handoff(GemmNT p, hipStream_t st) { return launch_glds<14, 14, 32, 4, false, 128, false, 1, true>(p, st); }
