// device helpers + launch-side profiling scope shared by the GEMM translation units.
// Each kernel family lives in its own .hip file ON PURPOSE: hipcc lets co-compiled template instantiations perturb each
// other's register allocation (accumulators rotating through misaligned AGPR ranges cost the conv kernel 15-50 %).
#pragma once
#include <vector>
#include "gemm.h"

#if FEDFR_FP16
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#else
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#endif

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
// Tile loads are BRANCH-FREE raw buffer loads: an out-of-range lane gets voffset = num_records and the hardware
// bounds check returns zeros.  (Predicated `if (ok) v = *p` loads made hipcc emit s_waitcnt vmcnt(0) after every
// load, serialising the 8 loads of a K-step: 2x slower.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}

// sum over the 16 lanes of a DPP row (lanes sharing lane >> 4), result in every lane: 4 VALU adds with DPP operands
// (quad xor 1, quad xor 2, half-row mirror, row mirror) instead of 4 ds_bpermute round trips per value.
__device__ __forceinline__ float row16_sum(float v) {
#define FEDFR_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), (ctrl), 0xf, 0xf, true))
  FEDFR_DPP_ADD(0xB1);    // quad_perm [1,0,3,2]
  FEDFR_DPP_ADD(0x4E);    // quad_perm [2,3,0,1]
  FEDFR_DPP_ADD(0x141);   // row_half_mirror
  FEDFR_DPP_ADD(0x140);   // row_mirror
#undef FEDFR_DPP_ADD
  return v;
}



// ---- optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg); state lives in gemm.hip
struct ProfScope {
  void* slot_ = nullptr;
  hipStream_t st_;
  ProfScope(int slot, double flops, hipStream_t st, double bytes = 0.0);   // bytes: ALGORITHMIC HBM bytes of the launch (operands read once, output written once)
  ~ProfScope();
};
double gemm_nt_alg_bytes(const GemmNT& p, int splits);      // algorithmic HBM bytes of a launch (gemm.hip)
double gemm_tn_alg_bytes(const GemmTN& p, int splits);
extern int g_tn_glds;
int launch_conv_glds8_w14(GemmNT p, hipStream_t st);       // conv_glds8_w14.hip  same, 8 waves per tile
int launch_conv_glds8_w28(GemmNT p, hipStream_t st);       // conv_glds8_w28.hip
int launch_conv_glds8_w28_stats(GemmNT p, hipStream_t st); // conv_glds8_w28s.hip  two image tiles per workgroup, one BatchNorm partial row each
extern int g_conv28_tpw2;
int launch_conv_glds_w56(GemmNT p, hipStream_t st);        // conv_glds_w56.hip  56x56, C = N = 64
int launch_conv_glds_w112(GemmNT p, hipStream_t st);       // conv_glds_w112.hip 112x112, C = N = 64
int launch_conv_glds_w56_c64_n128(GemmNT p, hipStream_t st);   // conv_glds_w56b.hip
int launch_conv_glds_w56_c128_n64(GemmNT p, hipStream_t st);
// conv_c64p.hip: persistent 3x3 conv for the 64 -> 64 channel layers (112x112 / 56x56), filter bank in registers
extern int g_conv_c64p;
bool conv_c64p_applies(const GemmNT& p);
int conv_c64p_grid(int M);
int launch_conv_c64p(GemmNT p, hipStream_t st);
int launch_conv_glds8_fused_w14(GemmNT p, hipStream_t st); // conv_glds8_fused_w14.hip  + BN-backward reduction epilogue
int launch_conv_glds8_fused_w28(GemmNT p, hipStream_t st); // conv_glds8_fused_w28.hip
int launch_conv_glds8_fused_w28s(GemmNT p, hipStream_t st); // conv_glds8_fused_w28s.hip
int launch_conv_glds8_fused_w14_papply(GemmNT p, hipStream_t st);    // conv_glds8_fused_papply.hip: the three above with sphnet's PReLU-apply epilogue (bmom == 2)
int launch_conv_glds8_fused_w28_papply(GemmNT p, hipStream_t st);
int launch_conv_glds8_fused_w28s_papply(GemmNT p, hipStream_t st);
// gemm_nt_glds.hip: the register-staged NT kernel's shapes with both operands fetched by LDS-DMA into a ring of stages
extern int g_nt_glds;
bool gemm_nt_glds_applies(const GemmNT& p, int BM, int splits);
int launch_nt_glds(const GemmNT& p, int BM, int splits, int slot, hipStream_t st);
int launch_tn_glds(GemmTN p, int splits, hipStream_t st);   // gemm_tn_glds.hip  wgrad GEMM, LDS-DMA operand ring
bool gemm_tn_glds_applies(int NI, int NJ, int C, int mode);
bool gemm_tn_glds_shape_ok(int NI, int NJ, int C, int mode);
int gemm_tn_glds_pick_splits(int Kp, int NI, int NJ);
// wgrad9.hip: 3x3 stride-1 weight gradient, all nine taps per workgroup, operands staged once
extern int g_wgrad9;
bool wgrad9_applies(const GemmTN& p);
bool wgrad9_applies_shape(int Kp, int NI, int NJ, int C, int W, int stride);
bool wgrad9_shape_ok(int Kp, int NI, int NJ, int C, int W, int stride);
int wgrad9_pick_splits(int Kp, int NI, int NJ, int W);
int launch_wgrad9(const GemmTN& p, int splits, hipStream_t st);
// wgrad9p.hip: the two same-shape 3x3 / stride-1 weight gradients of a residual block in one launch, 64 x 64 x 9 taps per workgroup
extern int g_wgrad9p;
bool wgrad9p_applies(const GemmTN& a, const GemmTN& b);
int wgrad9p_pick_splits(int Kp, int NI, int NJ, int W);
bool wgrad9p_job_ok(const GemmTN& a, int splits, const W9PJob& job);
int launch_wgrad9_pair(const GemmTN& a, const GemmTN& b, int splits, hipStream_t st, const W9PJob* job = nullptr);
