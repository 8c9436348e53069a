// halo kernel v1 (masked, swizzled) for 3x3/s1 convs on maps other than 14x14 / 28x28 -- own translation unit (gemm_dev.h)
#include <algorithm>
#include "gemm_dev.h"

// =====================================================================================================
// 3x3 / stride-1 / pad-1 convolution (fwd, and dgrad with the flipped shadow) with an LDS-resident HALO tile.
// The generic kernel above re-fetches the 128-pixel activation tile for each of the 9 taps; here the
// 128 + 2(W+1) consecutive NHWC pixels a tile can touch are staged ONCE per 64-channel chunk and every tap reads
// its shifted window from LDS (row = pixel + r*W + s), masked per lane for image borders.  Activation traffic
// through L2->CU and VGPR->LDS drops ~7x (14x14) .. 3x (112x112); the weight tile [BN][64] per (tap, chunk) stays
// register-staged and double buffered.  K order: chunk outer, tap inner (only the fp32 summation order changes).
// =====================================================================================================
template <int BN, int AH>   // AH = halo 16-B chunks per thread = ceil((128 + 2W + 2) * 8 / 256)
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(GemmNT p, int a_bytes_lds) {
  constexpr int BM = 128, WM = 2, WN = 2;
  constexpr int BI = BN / 32;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int B_BYTES = BN * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + a_bytes_lds;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bn = lid % p.nbn, bm = lid / p.nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  const int W = p.W, NR = BM + 2 * W + 2;
  const int l15 = lane & 15, lg = lane >> 4;
  const int ch = tid & 7, rbase = tid >> 3;
  const int npix = p.M;                       // stride 1: input pixels == output pixels
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);

  // per-lane 9-bit tap validity for each of the TM fragment rows this lane feeds
  unsigned vmask[TM];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
    const int m = m0 + wm * (BM / WM) + mi * 16 + l15;
    unsigned msk = 0;
    if (m < p.M) {
      const int hw = p.H * W;
      const int rem = m % hw;
      const int h = rem / W, w = rem - h * W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
        if ((unsigned)hh < (unsigned)p.H && (unsigned)ww < (unsigned)W) msk |= 1u << t;
      }
    }
    vmask[mi] = msk;
  }

  uint4 rh[AH], rb[BI];
  auto load_halo = [&](int cc) {
#pragma unroll
    for (int i = 0; i < AH; ++i) {
      const int e = tid + 256 * i;
      const int rl = e >> 3, c = e & 7;
      const int pix = m0 - (W + 1) + rl;
      const bool ok = rl < NR && (unsigned)pix < (unsigned)npix;
      const unsigned off = ((unsigned)pix * (unsigned)p.C + (unsigned)(cc * 64 + c * 8)) * 2u;
      rh[i] = buf_load16(rsA, ok ? off : p.a_bytes);
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int i = 0; i < AH; ++i) {
      const int e = tid + 256 * i;
      const int rl = e >> 3, c = e & 7;
      if (rl < NR) *reinterpret_cast<uint4*>(sA + rl * 128 + ((c ^ (rl & 7)) << 4)) = rh[i];
    }
  };
  auto load_b = [&](int tap, int cc) {
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int n = n0 + rbase + 32 * i;
      const unsigned off = ((unsigned)n * (unsigned)p.K + (unsigned)(tap * p.C + cc * 64 + ch * 8)) * 2u;
      rb[i] = buf_load16(rsB, n < p.N ? off : p.b_bytes);
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int row = rbase + 32 * i;
      *reinterpret_cast<uint4*>(sB + buf * B_BYTES + row * 128 + ((ch ^ (row & 7)) << 4)) = rb[i];
    }
  };

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int cpt = p.C >> 6;
  load_halo(0);
  load_b(0, 0);
  store_halo();
  store_b(0);
  __syncthreads();
  int buf = 0;
  const bf16x8_t zfrag = __builtin_bit_cast(bf16x8_t, make_uint4(0, 0, 0, 0));
  for (int cc = 0; cc < cpt; ++cc) {
    for (int tap = 0; tap < 9; ++tap) {
      const bool last_tap = tap == 8;
      const bool more = !(last_tap && cc + 1 == cpt);
      if (more) load_b(last_tap ? 0 : tap + 1, last_tap ? cc + 1 : cc);
      if (last_tap && cc + 1 < cpt) load_halo(cc + 1);
      const int r = tap / 3, sft = r * W + (tap - 3 * r);
      const unsigned char* cB = sB + buf * B_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int c = ks * 4 + lg;
        bf16x8_t fb[TN], fa[TM];
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
          const int row = wn * (BN / WN) + ni * 16 + l15;
          fb[ni] = *reinterpret_cast<const bf16x8_t*>(cB + row * 128 + ((c ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
          const int row = wm * (BM / WM) + mi * 16 + l15 + sft;
          const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(sA + row * 128 + ((c ^ (row & 7)) << 4));
          fa[mi] = ((vmask[mi] >> tap) & 1u) ? v : zfrag;
        }
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
          for (int mi = 0; mi < TM; ++mi) acc[ni][mi] = MFMA16(fb[ni], fa[mi], acc[ni][mi]);
      }
      if (more) store_b(buf ^ 1);
      __syncthreads();                       // next weight tile visible; everyone is done with this tap's reads
      if (last_tap && cc + 1 < cpt) {
        store_halo();                        // safe: all waves passed the barrier above => no reader of the old halo
        __syncthreads();
      }
      buf ^= 1;
    }
  }

  // ---- epilogue (identical to gemm_nt_kernel's bf16 path) ----
  constexpr int CST = BN * 2 + 16;
  unsigned char* sC = smem;
  float ssum[TN][4], ssq[TN][4];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni)
#pragma unroll
    for (int q = 0; q < 4; ++q) ssum[ni][q] = ssq[ni][q] = 0.f;
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
        const float v = bf2f(h[q]);
        ssum[ni][q] += v;
        ssq[ni][q] += v * v;
      }
      uint2 pk;
      pk.x = (unsigned)h[0] | ((unsigned)h[1] << 16);
      pk.y = (unsigned)h[2] | ((unsigned)h[3] << 16);
      *reinterpret_cast<uint2*>(sC + ml * CST + nl * 2) = pk;
    }
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
  constexpr int CPR = BN / 8;
  for (int idx = tid; idx < BM * CPR; idx += 256) {
    const int row = idx / CPR, c = idx - row * CPR;
    const int m = m0 + row, n = n0 + c * 8;
    if (m < p.M && n < p.N)
      *reinterpret_cast<uint4*>(p.Cb + (size_t)m * p.ldc + n) = *reinterpret_cast<const uint4*>(sC + row * CST + c * 16);
  }
}


template <int BN, int AH>
static int launch_halo(GemmNT p, hipStream_t st) {
  const int nbm = ceil_div(p.M, 128);
  p.nbn = ceil_div(p.N, BN);
  const int NR = 128 + 2 * p.W + 2;
  const int a_lds = (int)align_up((size_t)NR * 128, 256);
  constexpr size_t kEpi = (size_t)128 * (BN * 2 + 16);
  size_t lds = (size_t)a_lds + 2 * (size_t)BN * 128;
  if (lds < kEpi) lds = kEpi;
  static PerDeviceOnce attr_once;     // hipFuncSetAttribute is per device (a Server process may drive several)
  attr_once.run([&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<BN, AH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        160 * 1024);
  });
  ProfScope prof(11, 2.0 * p.M * p.N * (double)p.K, st);
  hipLaunchKernelGGL((conv3x3_halo_kernel<BN, AH>), dim3(nbm * p.nbn), dim3(256), lds, st, p, a_lds);
  FEDFR_LAUNCH_CHECK("conv3x3_halo");
  return FEDFR_OK;
}


int launch_conv_halo1(GemmNT p, hipStream_t st) {
  const int ah = ceil_div((128 + 2 * p.W + 2) * 8, 256);
  if (p.N <= 64) {
    if (ah <= 6) return launch_halo<64, 6>(p, st);
    if (ah <= 8) return launch_halo<64, 8>(p, st);
    return launch_halo<64, 12>(p, st);
  }
  if (ah <= 6) return launch_halo<128, 6>(p, st);
  if (ah <= 8) return launch_halo<128, 8>(p, st);
  return launch_halo<128, 12>(p, st);
}
