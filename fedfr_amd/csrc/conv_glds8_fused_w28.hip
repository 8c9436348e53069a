// LDS-DMA 3x3 dgrad with the BN-backward reduction fused into its epilogue, 28x28 maps, 8 waves -- own translation unit (gemm_dev.h)
#include "conv_glds_impl.h"
int launch_conv_glds8_fused_w28(GemmNT p, hipStream_t st) { return launch_glds<28, 7, 40, 4, true>(p, st); }
