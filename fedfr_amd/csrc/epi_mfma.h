// Epilogue statistics on the matrix cores: helpers shared by the LDS-DMA conv kernels (conv_glds_impl.h, conv_c64p.hip).
//
// A conv epilogue's BatchNorm sums are column sums over the pixels of the 16-bit output tile Y (and of products with the BatchNorm input tile X)
// staged in LDS.  With one transposed 16 x 16 x 32 fragment per 32-pixel step (2 ds_read_b64_tr_b16; the same registers serve as A and B operand):
//   sum_m Y[m][n] = (ones x Y)[.][n]      sum_m Y[m][n]^2 = diag(Y^T Y)[n]      sum_m Y[m][n] X[m][n] = diag(Y^T X)[n]
// Products of two 16-bit values are exact in fp32, the accumulation is fp32.  A PReLU in front of the BatchNorm (dz = dy * (z <= 0 ? alpha : 1),
// z = x * sc + sh) is a THRESHOLD on x:  z <= 0  <=>  sgn(sc) x <= T = -sh / |sc|, evaluated on the fragment registers (below).
#pragma once
#include "gemm_dev.h"

struct PreluThr {
  unsigned t162, nsgn2;      // fp16 storage: (t16, t16) with t16 = the largest fp16 <= T, and (-sgn(sc), -sgn(sc)), packed
  float Tf, nsf;             // bf16 storage: T and -sgn(sc) in fp32
};
__device__ __forceinline__ PreluThr prelu_threshold(float sc, float sh) {
  PreluThr th;
  const float T = -sh / fabsf(sc);
  unsigned hb = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)T);      // round to nearest ...
  const float hf = (float)__builtin_bit_cast(_Float16, (unsigned short)hb);
  if (hf > T) hb = hb == 0u ? 0x8001u : ((hb & 0x8000u) ? hb + 1u : hb - 1u);   // ... then down to the largest fp16 <= T
  if (sc == 0.f) hb = sh <= 0.f ? 0x7c00u : 0xfc00u;                            // z = sh everywhere: always / never in the PReLU's negative branch
  th.t162 = hb | (hb << 16);
  th.nsgn2 = sc < 0.f ? 0x3c003c00u : 0xbc00bc00u;
  th.Tf = sc == 0.f ? (sh <= 0.f ? __builtin_inff() : -__builtin_inff()) : T;
  th.nsf = sc < 0.f ? 1.f : -1.f;
  return th;
}
// DYPOS = (z > 0 ? dy : 0) on a fragment (8 values of one channel): e = t - sgn(sc) x has the exact sign (the fp16 difference of two fp16
// values; an fp32 FMA of a bf16 value), sign bit set <=> z > 0.  fp16: one packed FMA + one packed shift + one AND per pair; bf16: 7 operations.
__device__ __forceinline__ s16x8_t prelu_pos(const s16x8_t& d8, const s16x8_t& x8, const PreluThr& th) {
#if FEDFR_FP16
  typedef __attribute__((ext_vector_type(8))) _Float16 h8_t;
  const _Float16 ns1 = __builtin_bit_cast(_Float16, (unsigned short)(th.nsgn2 & 0xffffu)), tt1 = __builtin_bit_cast(_Float16, (unsigned short)(th.t162 & 0xffffu));
  const h8_t ns8 = {ns1, ns1, ns1, ns1, ns1, ns1, ns1, ns1}, tt8 = {tt1, tt1, tt1, tt1, tt1, tt1, tt1, tt1};
  const h8_t e8 = __builtin_elementwise_fma(__builtin_bit_cast(h8_t, x8), ns8, tt8);
  const s16x8_t m8 = __builtin_bit_cast(s16x8_t, e8) >> 15;
  return d8 & m8;
#else
  typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
  const u4_t du = __builtin_bit_cast(u4_t, d8), xu = __builtin_bit_cast(u4_t, x8);
  u4_t pw;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float el = __builtin_fmaf(__uint_as_float(xu[j] << 16), th.nsf, th.Tf), eh = __builtin_fmaf(__uint_as_float(xu[j] & 0xffff0000u), th.nsf, th.Tf);
    const unsigned ml = (unsigned)((int)__float_as_uint(el) >> 31), mh = (unsigned)((int)__float_as_uint(eh) >> 31);
    pw[j] = du[j] & ((ml & 0xffffu) | (mh & 0xffff0000u));
  }
  return __builtin_bit_cast(s16x8_t, pw);
#endif
}
__device__ __forceinline__ bf16x8_t mfma_ones8() {
  s16x8_t one8;
#pragma unroll
  for (int j = 0; j < 8; ++j) one8[j] = FEDFR_FP16 ? (short)0x3c00 : (short)0x3f80;
  return __builtin_bit_cast(bf16x8_t, one8);
}
// the diagonal element of a 16 x 16 accumulator block that this lane holds (lanes with (lane & 15) >> 2 == lane >> 4: register lane & 3)
__device__ __forceinline__ float mfma_diag(const f32x4_t& g, int l15) {
  const int qd = l15 & 3;
  return qd == 0 ? g[0] : qd == 1 ? g[1] : qd == 2 ? g[2] : g[3];
}
