// LDS-DMA 3x3 dgrad with the BN-backward reduction fused into its epilogue, 14x14 maps, 8 waves -- own translation unit (gemm_dev.h)
#include "conv_glds_impl.h"
int launch_conv_glds8_fused_w14(GemmNT p, hipStream_t st) { return launch_glds<14, 14, 32, 4, true>(p, st); }
