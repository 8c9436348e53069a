// LDS layout helpers of the TN (reduction-major operands) GEMM kernels: chunk swizzle + ds_read_b64_tr_b16 fragment reads.
#pragma once
#include "gemm_dev.h"

template <int RB>   // tile row bytes (128 or 256): chunk swizzle that makes tr-reads and b128 writes conflict-free
__device__ __forceinline__ int tn_swz(int row) {
  if (RB == 256) return ((row & 3) << 1) | (((row >> 3) & 1) << 3);
  return (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2);
}

// per-lane byte offset (inside one LDS stage) of the first tr-read of the fragment for column block `colblk`;
// the second read is +4*RB, the ks=1 half +32*RB (the swizzle only depends on row bits 0-1 and 3, which those
// offsets do not touch), so every in-loop address is base + register + immediate.
template <int RB>
__device__ __forceinline__ int tn_frag_off(int colblk, int lane) {
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int col = colblk + 4 * pp;
  const int r0 = 8 * g + q;
  return r0 * RB + (((col >> 3) ^ tn_swz<RB>(r0)) << 4) + ((col >> 2) & 1) * 8;
}
template <int RB>
__device__ __forceinline__ bf16x8_t tn_frag_tr(const unsigned char* stage, int off, int ks) {
  typedef __attribute__((address_space(3))) s16x4_t* lds_p;
  const unsigned char* a0 = stage + off + ks * 32 * RB;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)a0);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0 + 4 * RB));
  s16x8_t v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8_t, v);
}
template <int RB>   // validation fallback: scalar LDS gathers (no transpose-read instruction)
__device__ __forceinline__ bf16x8_t tn_frag_scalar(const unsigned char* tile, int ks, int colblk, int lane) {
  const int g = lane >> 4, li = lane & 15;
  const int col = colblk + li;
  const int c = col >> 3, e = col & 7;
  s16x8_t v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int rr = ks * 32 + 8 * g + j;
    v[j] = *reinterpret_cast<const short*>(tile + rr * RB + ((c ^ tn_swz<RB>(rr)) << 4) + e * 2);
  }
  return __builtin_bit_cast(bf16x8_t, v);
}

