// conv3x3_halo2_kernel + launcher template; included by one translation unit per hot instantiation (see gemm_dev.h).
#pragma once
#include <algorithm>
#include "gemm_dev.h"

// In-kernel phase stamps (s_memtime shader clock + 100 MHz wall clock) for tools/stamp_halo2.hip; compiled out of the product.
#ifdef FEDFR_HALO2_STAMPS
#define HALO2_STAMP(i)                                                                      \
  do {                                                                                      \
    if (threadIdx.x == 0 && p.dbg) {                                                        \
      p.dbg[(size_t)blockIdx.x * 16 + 2 * (i)] = __builtin_readcyclecounter();             \
      p.dbg[(size_t)blockIdx.x * 16 + 2 * (i) + 1] = wall_clock64();                        \
    }                                                                                       \
  } while (0)
#else
#define HALO2_STAMP(i) do { } while (0)
#endif

// =====================================================================================================
// halo kernel v2 (W = 14 / 28, i.e. 84 of iresnet100's 103 convs): the LDS image is laid out in ZERO-PADDED image
// coordinates — every image row gets a zero pixel left and right, every image a zero row above and below — so a
// filter tap is a pure constant shift ((r*(W+2) + s) rows) with NO per-lane border masks, and with a linear
// 160-byte row stride (conflict-free for ds_read_b128 without XOR) + W as a template constant the 9 tap offsets
// are instruction immediates.  v1 spent 136 VALU instructions per 32 MFMAs on masks and swizzled addresses.
// =====================================================================================================
template <int BN, int W_, int WN, bool FUSED>   // FUSED: BN-backward reduction epilogue.  WN = 2: 4 waves (64x64 wave tiles at BN=128); WN = 4: 8 waves (64x32)
__global__ __launch_bounds__(128 * WN) void conv3x3_halo2_kernel(GemmNT p, int nr_rows) {
  constexpr int BM = 128, WM = 2, PW = W_ + 2, RS = 160, NT = 64 * WM * WN;
  constexpr int BI = BN * 8 / NT, BROWS = NT / 8;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int B_BYTES = BN * 128;
  constexpr int NSRC = BM + 2 * W_ + 2;                 // source pixels a tile can touch
  constexpr int AH = (NSRC * 8 + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  const int a_bytes_lds = (nr_rows * RS + 255) & ~255;
  unsigned char* sB = smem + a_bytes_lds;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  HALO2_STAMP(0);
  const int wm = wave / WN, wn = wave % WN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bn = lid % p.nbn, bm = lid / p.nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  const int l15 = lane & 15, lg = lane >> 4;
  const int ch = tid & 7, rbase = tid >> 3;
  const int npix = p.M, HW = p.H * W_, PIMG = (p.H + 2) * PW;
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);
  auto qof = [&](int pix) {                             // padded coordinate of a flattened (img, h, w) pixel
    const int img = pix / HW, rem = pix - img * HW;
    const int h = rem / W_, w = rem - h * W_;
    return img * PIMG + (h + 1) * PW + w + 1;
  };
  const int qb = qof(m0) - (PW + 1);

  // zero the whole image once: padding positions are never written afterwards
  for (int i = tid * 16; i < a_bytes_lds; i += NT * 16) *reinterpret_cast<uint4*>(sA + i) = make_uint4(0, 0, 0, 0);

  // staging plan (same rows every chunk): source byte offset (without the channel-chunk term) and LDS destination
  unsigned src_off[AH];
  int dst_off[AH];
#pragma unroll
  for (int i = 0; i < AH; ++i) {
    const int e = tid + NT * i;
    const int rl = e >> 3, c = e & 7;
    const int pix = m0 - (W_ + 1) + rl;
    bool ok = rl < NSRC && (unsigned)pix < (unsigned)npix;
    int row = 0;
    if (ok) {
      row = qof(pix) - qb;
      ok = (unsigned)row < (unsigned)nr_rows;
    }
    src_off[i] = ok ? ((unsigned)pix * (unsigned)p.C + (unsigned)(c * 8)) * 2u : 0xffffffffu;
    dst_off[i] = ok ? row * RS + c * 16 : -1;
  }
  // A-fragment base addresses (tap (0,0)); rows >= M are clamped (their results are masked in the epilogue)
  int a_addr[TM];
  bool m_ok[TM];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
    const int m = m0 + wm * (BM / WM) + mi * 16 + l15;
    m_ok[mi] = m < p.M;
    a_addr[mi] = (qof(m_ok[mi] ? m : p.M - 1) - (PW + 1) - qb) * RS + lg * 16;
  }

  uint4 rh[AH], rb[BI];
  auto load_halo = [&](int cc) {
#pragma unroll
    for (int i = 0; i < AH; ++i)
      rh[i] = buf_load16(rsA, src_off[i] == 0xffffffffu ? p.a_bytes : src_off[i] + (unsigned)(cc * 128));
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int i = 0; i < AH; ++i)
      if (dst_off[i] >= 0) *reinterpret_cast<uint4*>(sA + dst_off[i]) = rh[i];
  };
  const unsigned b_row0 = ((unsigned)(n0 + rbase) * (unsigned)p.K + (unsigned)(ch * 8)) * 2u;
  auto load_b = [&](int tap, int cc) {
    const unsigned koff = (unsigned)(tap * p.C + cc * 64) * 2u;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int n = n0 + rbase + BROWS * i;
      rb[i] = buf_load16(rsB, n < p.N ? b_row0 + (unsigned)(BROWS * i) * (unsigned)p.K * 2u + koff : p.b_bytes);
    }
  };
  const int b_st = rbase * 128 + ((ch ^ (rbase & 7)) << 4);
  auto store_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < BI; ++i) *reinterpret_cast<uint4*>(sB + buf * B_BYTES + b_st + i * BROWS * 128) = rb[i];
  };
  int b_addr[TN];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
    const int row = wn * (BN / WN) + ni * 16 + l15;
    b_addr[ni] = row * 128 + ((lg ^ (row & 7)) << 4);      // ks = 1 flips chunk bit 2: XOR 64
  }

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int cpt = p.C >> 6;
  load_halo(0);
  load_b(0, 0);
  __syncthreads();                   // zero fill complete before real pixels land
  store_halo();
  store_b(0);
  __syncthreads();
  HALO2_STAMP(1);
  int buf = 0;
  for (int cc = 0; cc < cpt; ++cc) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      constexpr int kDummy = 0;
      (void)kDummy;
      const bool last_tap = tap == 8;
      const bool more = !(last_tap && cc + 1 == cpt);
      if (more) load_b(last_tap ? 0 : tap + 1, last_tap ? cc + 1 : cc);
      if (last_tap && cc + 1 < cpt) load_halo(cc + 1);
      const int toff = ((tap / 3) * PW + (tap % 3)) * RS;          // compile-time per unrolled tap
      const unsigned char* cB = sB + buf * B_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t fb[TN], fa[TM];
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) fb[ni] = *reinterpret_cast<const bf16x8_t*>(cB + (b_addr[ni] ^ (ks * 64)));
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) fa[mi] = *reinterpret_cast<const bf16x8_t*>(sA + a_addr[mi] + toff + ks * 64);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
          for (int mi = 0; mi < TM; ++mi) acc[ni][mi] = MFMA16(fb[ni], fa[mi], acc[ni][mi]);
      }
      if (more) store_b(buf ^ 1);
      __syncthreads();
      if (last_tap && cc + 1 < cpt) {
        store_halo();
        __syncthreads();
      }
      buf ^= 1;
    }
  }

  HALO2_STAMP(2);
  // ---- epilogue: as gemm_nt_kernel's bf16 path; rows >= M contribute nothing to the statistics ----
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
        const float v = m_ok[mi] ? bf2f(h[q]) : 0.f;
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
  if constexpr (!FUSED) {
    for (int idx = tid; idx < BM * CPR; idx += NT) {
      const int row = idx / CPR, c = idx - row * CPR;
      const int m = m0 + row, n = n0 + c * 8;
      if (m < p.M && n < p.N)
        *reinterpret_cast<uint4*>(p.Cb + (size_t)m * p.ldc + n) = *reinterpret_cast<const uint4*>(sC + row * CST + c * 16);
    }
    HALO2_STAMP(3);
  } else {
  // ---- fused BN-backward reduction: this thread owns chunk column c (8 channels) of rows rg, rg + NT/CPR, ... ----
  constexpr int RG = NT / CPR;
  const int c = tid % CPR, rg = tid / CPR;
  const int n = n0 + c * 8;
  const bool n_ok = n < p.N;
  float mean[8], rstd[8], ga[8], be[8], al[8], s1[8], s2[8], s3[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int nn = n_ok ? n + q : 0;
    mean[q] = p.bmean[nn]; rstd[q] = p.brstd[nn];
    ga[q] = p.bgamma ? p.bgamma[nn] : 1.f; be[q] = p.bbeta ? p.bbeta[nn] : 0.f; al[q] = p.balpha ? p.balpha[nn] : 1.f;
    s1[q] = s2[q] = s3[q] = 0.f;
  }
  const bool has_alpha = p.balpha != nullptr;
  for (int row = rg; row < BM; row += RG) {
    const int m = m0 + row;
    if (m < p.M && n_ok) {
      const uint4 dv = *reinterpret_cast<const uint4*>(sC + row * CST + c * 16);
      *reinterpret_cast<uint4*>(p.Cb + (size_t)m * p.ldc + n) = dv;
      float dy[8], xv[8];
      unpack8(dv, dy);
      unpack8(*reinterpret_cast<const uint4*>(p.bx + (size_t)m * p.N + n), xv);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float xh = (xv[q] - mean[q]) * rstd[q];
        float dz = dy[q];
        if (has_alpha) {
          const float z = ga[q] * xh + be[q];
          if (z <= 0.f) {
            s3[q] += dy[q] * z;
            dz = dy[q] * al[q];
          }
        }
        s1[q] += dz;
        s2[q] += dz * xh;
      }
    }
  }
  __syncthreads();                                   // everyone is done reading the staged C tile
  float* red = reinterpret_cast<float*>(smem);       // [RG][3][BN]
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    red[(rg * 3 + 0) * BN + c * 8 + q] = s1[q];
    red[(rg * 3 + 1) * BN + c * 8 + q] = s2[q];
    red[(rg * 3 + 2) * BN + c * 8 + q] = s3[q];
  }
  __syncthreads();
  for (int i = tid; i < 3 * BN; i += NT) {
    const int which = i / BN, col = i - which * BN;
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < RG; ++r) t += red[(r * 3 + which) * BN + col];
    if (n0 + col < p.N) p.bpart[((size_t)bm * 3 + which) * p.N + n0 + col] = t;
  }
  }
}

template <int BN, int W_, int WN, bool FUSED>
static int launch_halo2(GemmNT p, hipStream_t st) {
  const int nbm = ceil_div(p.M, 128);
  p.nbn = ceil_div(p.N, BN);
  // exact LDS image height: max over tiles of the padded-coordinate span, plus the halo on both sides
  const int PW = W_ + 2, HW = p.H * W_, PIMG = (p.H + 2) * PW;
  auto qof = [&](int pix) {
    const int img = pix / HW, rem = pix - img * HW;
    return img * PIMG + (rem / W_ + 1) * PW + rem % W_ + 1;
  };
  int span = 0;
  for (int t = 0; t < nbm; ++t) {
    const int a = t * 128, b = (a + 127 < p.M - 1) ? a + 127 : p.M - 1;
    span = std::max(span, qof(b) - qof(a));
  }
  const int nr = span + 2 * (PW + 1) + 1;
  const size_t a_lds = ((size_t)nr * 160 + 255) & ~(size_t)255;
  constexpr size_t kEpi = (size_t)128 * (BN * 2 + 16), kRed = (size_t)(128 * WN / (BN / 8)) * 3 * BN * sizeof(float);
  size_t lds = a_lds + 2 * (size_t)BN * 128;
  if (lds < kEpi) lds = kEpi;
  if (lds < kRed) lds = kRed;
  FEDFR_REQUIRE(lds <= 160 * 1024, "conv3x3_halo2: LDS image too large (%zu bytes)", lds);
  FEDFR_REQUIRE(FUSED == (p.bpart != nullptr), "conv3x3_halo2: fused/plain variant mismatch");
  if (p.bpart) {
    FEDFR_REQUIRE(p.bx && p.bmean && p.brstd && p.ldc == p.N, "conv3x3_halo2: fused BN-bwd reduction needs bx/mean/rstd and ldc == N");
    if (p.bwd_fused) *p.bwd_fused = nbm;
  }
  static PerDeviceOnce attr_once;     // hipFuncSetAttribute is per device (a Server process may drive several)
  attr_once.run([&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo2_kernel<BN, W_, WN, FUSED>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        160 * 1024);
  });
  ProfScope prof(BN == 64 ? 10 : (W_ == 14 ? 8 : 9), 2.0 * p.M * p.N * (double)p.K, st);
  hipLaunchKernelGGL((conv3x3_halo2_kernel<BN, W_, WN, FUSED>), dim3(nbm * p.nbn), dim3(128 * WN), lds, st, p, nr);
  FEDFR_LAUNCH_CHECK("conv3x3_halo2");
  return FEDFR_OK;
}

