// Output stage shared by the NT kernels (gemm.hip: register-staged operands; gemm_nt_glds.hip: LDS-DMA operand ring).
#pragma once
#include <type_traits>
#include "gemm_dev.h"

// acc[ni][mi][reg]: n = wn*(BN/WN)+ni*16+lg*4+reg ; m = wm*(BM/WM)+mi*16+l15.  `smem` is reused as the staging tile: every wave must be
// past its last operand read (and no LDS-DMA may be in flight) when this is called.  fp32 split-K slabs (p.Cf), or the bf16 tile through
// LDS with the per-column (sum, sum of squares) partial row of the BatchNorm that follows (p.stats) and the parity-class scatter of a
// stride-2 dgrad (p.par_on).
template <int BM, int BN, int WM, int WN, int NTHREADS>
__device__ __forceinline__ void nt_epilogue(const GemmNT& p, f32x4_t (&acc)[BN / WN / 16][BM / WM / 16], unsigned char* smem, int bm, int m0,
                                            int n0, int split, int wm, int wn, int tid, int lane) {
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  const int l15 = lane & 15, lg = lane >> 4;
  if (p.Cf) {
    float* slab = p.Cf + (size_t)split * p.M * p.N;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const int m = m0 + wm * (BM / WM) + mi * 16 + l15;
        const int n = n0 + wn * (BN / WN) + ni * 16 + lg * 4;
        if (m < p.M && n < p.N) *reinterpret_cast<float4*>(slab + (size_t)m * p.N + n) =
            make_float4(acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]);
      }
    return;
  }
  constexpr int CST = BN * 2 + 16;   // staged C row stride in bytes
  unsigned char* sC = smem;
  float ssum[TN][4], ssq[TN][4];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni)
#pragma unroll
    for (int q = 0; q < 4; ++q) ssum[ni][q] = ssq[ni][q] = 0.f;
  // (two copies of the staging loop behind one uniform branch: dgrad / fc launches carry no statistics, and the conversions back to fp32, adds and
  // FMAs of the sums are VALU time the matrix cores sit through)
  auto stage_tile = [&](auto ST_) {
    constexpr bool ST = decltype(ST_)::value;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const int ml = wm * (BM / WM) + mi * 16 + l15;
        const int nl = wn * (BN / WN) + ni * 16 + lg * 4;
        bf16_t h[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          h[q] = f2bf(acc[ni][mi][q]);
          if constexpr (ST) {
            const float v = bf2f(h[q]);
            ssum[ni][q] += v;
            ssq[ni][q] += v * v;
          }
        }
        uint2 pk;
        pk.x = (unsigned)h[0] | ((unsigned)h[1] << 16);
        pk.y = (unsigned)h[2] | ((unsigned)h[3] << 16);
        *reinterpret_cast<uint2*>(sC + ml * CST + nl * 2) = pk;
      }
  };
  if (p.stats) stage_tile(std::true_type{}); else stage_tile(std::false_type{});
  if (p.stats) {
    float* prow = p.stats + (size_t)(bm * WM + wm) * 2 * p.N;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float a = row16_sum(ssum[ni][q]), b = row16_sum(ssq[ni][q]);
        const int n = n0 + wn * (BN / WN) + ni * 16 + lg * 4 + q;
        if (l15 == 0 && n < p.N) {
          prow[n] = a;
          prow[p.N + n] = b;
        }
      }
  }
  __syncthreads();
  constexpr int CPR = BN / 8;   // 16-B chunks per staged row
  for (int idx = tid; idx < BM * CPR; idx += NTHREADS) {
    const int row = idx / CPR, c = idx - row * CPR;
    const int m = m0 + row, n = n0 + c * 8;
    if (m < p.M && n < p.N) {
      size_t mo = (size_t)m;
      if (p.par_on) {
        const int hw = p.Ho * p.Wo, img = m / hw, rem = m - img * hw, h2 = rem / p.Wo, w2 = rem - h2 * p.Wo;
        mo = ((size_t)img * p.outH + 2 * h2 + p.par_h) * p.outW + 2 * w2 + p.par_w;
      }
      *reinterpret_cast<uint4*>(p.Cb + mo * p.ldc + n) = *reinterpret_cast<const uint4*>(sC + row * CST + c * 16);
    }
  }
}
