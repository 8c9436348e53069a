// LDS-DMA 3x3 dgrad convs whose epilogue applies a bare PReLU's backward to the OUTPUT (sphnet: GemmNT::bmom == 2) — the one fused form that keeps
// the per-element reduction pass (conv_glds_impl.h: PAPPLY), apart from the iresnet instantiations so that those do not carry its registers
#include "conv_glds_impl.h"
int launch_conv_glds8_fused_w14_papply(GemmNT p, hipStream_t st) { return launch_glds<14, 14, 32, 4, true, 128, false, 1, true>(p, st); }
int launch_conv_glds8_fused_w28_papply(GemmNT p, hipStream_t st) { return launch_glds<28, 7, 40, 4, true, 128, false, 1, true>(p, st); }
int launch_conv_glds8_fused_w28s_papply(GemmNT p, hipStream_t st) { return launch_glds<28, 7, 40, 4, true, 128, false, 2, true>(p, st); }
