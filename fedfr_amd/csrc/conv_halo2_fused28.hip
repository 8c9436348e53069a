// 3x3 halo dgrad with the fused BN-backward reduction epilogue, 28x28 maps -- alone in its translation unit (gemm_dev.h)
#include "conv_halo2_impl.h"
int launch_conv_halo2_fused_w28(GemmNT p, hipStream_t st) { return launch_halo2<128, 28, 2, true>(p, st); }
