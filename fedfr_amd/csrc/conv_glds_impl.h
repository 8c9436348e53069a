// conv3x3_glds_kernel + launcher template; one translation unit per instantiation (see gemm_dev.h).
#pragma once
#include <type_traits>
#include <algorithm>
#include "gemm_dev.h"
#include "epi_mfma.h"

// In-kernel phase stamps for tools/stamp (compiled out of the product)
#ifdef FEDFR_HALO2_STAMPS
#define GLDS_STAMP(i)                                                                       \
  do {                                                                                      \
    if (threadIdx.x == 0 && p.dbg) {                                                        \
      p.dbg[(size_t)blockIdx.x * 16 + 2 * (i)] = __builtin_readcyclecounter();             \
      p.dbg[(size_t)blockIdx.x * 16 + 2 * (i) + 1] = wall_clock64();                        \
    }                                                                                       \
  } while (0)
#else
#define GLDS_STAMP(i) do { } while (0)
#endif
// timing ablations for tools/stamp (results are WRONG with any bit set): 1 no in-loop DMA, 2 no barrier, 4 no fragment reads
#ifndef GLDS_RASTER
#define GLDS_RASTER 1      // padded-raster M index with fragment reuse across vertical taps (14x14 / 28x28 instantiations)
#endif
#ifndef GLDS_ASPREAD
#define GLDS_ASPREAD 1     // padded-raster kernels: the next channel chunk's image is requested one LDS-DMA piece per tap over the first taps (0: all pieces at tap 0, rounds 1-5)
#endif
#ifndef GLDS_ABLATE
#define GLDS_ABLATE 0      // timing ablations (WRONG results): 1 no in-loop LDS-DMA, 2 no per-tap barrier, 4 no fragment reads, 8 no prologue DMA / wait for the second tile of a workgroup, 16 no in-loop vmcnt waits (DMA still issued; RST path)
#endif
#ifndef GLDS_FUSED_ABLATE
#define GLDS_FUSED_ABLATE 0      // timing ablations of the fused BN-backward epilogue (WRONG results): 1 no x loads, 2 no per-element pass, 4 no reduction
#endif
#ifndef GLDS_MFMA_STATS
#define GLDS_MFMA_STATS 1  // forward (plain) launches of the 196-pixel tiles: BatchNorm partial sums of the staged output tile on the matrix cores (below)
#endif
#ifdef GLDS_SETPRIO
#define GLDS_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define GLDS_PRIO(x) do { } while (0)
#endif

// =====================================================================================================
// 3x3 / stride-1 / pad-1 convolution (fwd, and dgrad with the flipped shadow) for 14x14 / 28x28 maps, Cout % 128 == 0,
// C % 128 == 0, as an implicit GEMM whose operands reach LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`).
//
//   * Why: in-kernel stamps (tools/stamp) of the register-staged halo2 kernel showed its K loop at 32 % of the MFMA rate —
//     hipcc sinks the next tap's global loads to just before their ds_write, so every tap paid an L2 round trip.  LDS-DMA has
//     no register destination to sink towards: weight tiles 2-3 taps ahead and the next channel chunk's image are in flight
//     while a tap computes, retired by COUNTED s_waitcnt vmcnt(N) + one raw s_barrier per tap (never vmcnt(0) in the loop).
//   * Tile = 196 output pixels x 128 output channels: a whole 14x14 image, or 7 rows of a 28x28 image.  At the benchmark
//     batch (128) every 14x14 layer is exactly 256 tiles = one per CU (the 128-pixel tiling gave 392 = 1.53 rounds), a tile
//     never straddles images, and prologue / epilogue are paid once per 56 (not 32) MFMAs per tap and wave.
//   * A operand: the tile's pixels plus halo as a ZERO-PADDED image in LDS, one 128-B row (64 channels) per padded pixel, so a
//     filter tap is a constant row shift.  Out-of-image rows are fetched with an out-of-range buffer offset, which the hardware
//     turns into zeros in LDS (tools/probe/glds_probe.hip): no zero-fill pass, no masks.  Two image buffers (chunk cc / cc+1).
//   * B operand: one 128 (out channels) x 64 (k) slice of the KRSC weights per tap, ring of 4 buffers.
//   * LDS rows are 128 B with the 16-B chunk index XOR-ed by (row & 7): conflict-free ds_read_b128 for both operands.  An
//     LDS-DMA wave-instruction writes 1 KiB linearly (lane l -> base + 16 l), so the swizzle is applied to the per-lane SOURCE
//     address: lane l of a piece covering rows 8p..8p+7 fetches row 8p + (l >> 3), logical chunk (l & 7) ^ (l >> 3).
//   * 4 waves (2 x 2), wave tile 112 (7 fragments; 196 = 12.25 fragments, the tail is masked) x 64, v_mfma_f32_16x16x32_bf16.
//     The barrier sits in the MIDDLE of a tap, the first fragments of the next tap are read behind it, and fragment reads /
//     DMA issue are threaded between the MFMAs with sched_group_barrier.
//   * Epilogue: bf16 tile through LDS, per-column sum / sum-of-squares partials for the following BatchNorm in the layout of
//     the 128-row kernels (gemm_nt_stat_rows rows; the rows this tiling does not need are written as zeros).
// =====================================================================================================
template <int N_>
__device__ __forceinline__ void glds_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

__device__ __forceinline__ void glds_store_out(bf16_t* dst, const uint4& v) { st16_out(dst, v); }
template <int W_, int R_, int NPA, int WN, bool FUSED, int BN_ = 128, bool ONECHUNK = false, int TPW = 1, bool PAPPLY = false>   // PAPPLY (FUSED only): also carries sphnet's PReLU-apply epilogue (p.bmom == 2: the OUTPUT becomes dz — a per-element pass no matrix-core formulation covers); TPW: image tiles a workgroup computes one after the other (2: half as many BatchNorm partial rows — one per workgroup, summed over its tiles and both wave rows — so that the channel-sliced BatchNorm pass can reduce them itself, bn_sliced.hip); BN_: output-channel tile; ONECHUNK: C == 64 (one channel chunk, single image buffer); FUSED: BN-backward reduction in the (dgrad) epilogue; tile = R_ image rows; NPA = LDS-DMA pieces (1 KiB = 8 image rows) per image buffer;
                                             // WN = 2: 4 waves, one per SIMD (112 x 64 wave tiles); WN = 4: 8 waves, two per SIMD (112 x 32)
__global__ __launch_bounds__(128 * WN) void conv3x3_glds_kernel(GemmNT p, int stat_rows) {
  constexpr int PT = R_ * W_, BN = BN_, WM = 2, PW = W_ + 2, NW = WM * WN, NT = 64 * NW;
  constexpr int PWL = (PW + 7) & ~7;                      // LDS image pitch in rows: a multiple of 8, so that the swizzle key (row & 7) does not depend on
                                                         // the vertical tap offset -> A addresses need 3 (horizontal) variants, dy is an instruction immediate
  constexpr int TM = 7, TN = BN / WN / 16, MW = TM * 16;   // 112 fragment rows per wave, 224 per tile (196 valid)
  constexpr int AP = NPA / NW, BP = (BN / 8) / NW;       // LDS-DMA pieces per wave: image buffer / weight tile
  constexpr int NABUF = ONECHUNK ? 1 : 2;
  constexpr int NRD = TM + TN, MPR = (TM * TN) / NRD;    // fragment reads per k-half; MFMAs threaded per read
  constexpr int TPI = W_ / R_;                          // tiles per image
  constexpr int A_BYTES = NPA * 1024, B_BYTES = BN * 128, NB = 4;
  // RST: the GEMM's M index is the position q = y * PWL + x in the padded raster of the LDS image (the PWL - W_ filler columns of a
  // row are computed and thrown away: 224 positions either way), so the fragment of 16-position block rb for vertical tap dy IS the
  // fragment of block rb + dy * PWL / 16 for dy = 0: taps run dx-major and every A fragment is read from LDS once per horizontal tap
  // (9-11 row-block reads per k-step and dx instead of 21): 90 instead of 162 fragment reads per channel chunk and wave.
  constexpr bool RST = GLDS_RASTER && (W_ == 14 || W_ == 28) && R_ * PWL == 2 * MW;
  constexpr int VG = PWL / 16, NRB = TM + 2 * VG;
  static_assert(PT <= 2 * MW && PT > MW && W_ % R_ == 0 && (R_ + 2) * PWL <= NPA * 8 && NPA % NW == 0 && (BN / 8) % NW == 0 && MPR >= 1 &&
                    !(FUSED && ONECHUNK), "tile geometry");
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* sA = smem;                       // [2][A_BYTES]
  unsigned char* sB = smem + NABUF * A_BYTES;     // [NB][B_BYTES]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  GLDS_STAMP(0);
  const int wm = wave / WN, wn = wave % WN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bn = lid % p.nbn, btw = lid / p.nbn;          // workgroup btw computes image tiles btw * TPW .. + TPW - 1
  int bt = btw * TPW;
  int img = bt / TPI, y0 = (bt - img * TPI) * R_;
  int m0 = bt * PT;
  const int n0 = bn * BN;
  // XLATE (FUSED with several tiles per workgroup, round 3): the two-tiles instantiation sits at the register limit, so the BatchNorm input
  // tile cannot ride through the K loop in registers.  It is requested right BEHIND the loop (after the drain of the tail DMAs: the loads
  // fly while the accumulators are staged) and the tiles' column sums meet in LDS: ONE partial row per workgroup.
  constexpr bool XLATE = FUSED && TPW > 1;
  const int l15 = lane & 15, lg = lane >> 4;
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);

  // ---- LDS-DMA source plan.  Lane l of every piece: row (l >> 3) of the piece, logical 16-B chunk (l & 7) ^ (l >> 3).
  // LDS image row r <-> padded pixel (y0 + r / PWL, r % PWL) of image img; padded row 0 / H+1, column 0 / W+1 and the pitch
  // filler columns are zero.
  const int prow = lane >> 3, pch = (lane & 7) ^ prow;
  constexpr unsigned OOB = 0xfffffff0u;                 // beyond any buffer: the load writes zeros to LDS
  unsigned a_src[AP];                                   // byte offset of this lane's pixel/chunk at channel chunk 0
  auto plan_tile = [&]() {
#pragma unroll
    for (int j = 0; j < AP; ++j) {
      const int r = (j * NW + wave) * 8 + prow;
      const int ry = r / PWL, rx = r - ry * PWL;
      const int yy = y0 + ry;
      const bool ok = ry < R_ + 2 && yy >= 1 && yy <= W_ && rx >= 1 && rx <= W_;
      const unsigned pix = (unsigned)(img * (W_ * W_) + (yy - 1) * W_ + (rx - 1));
      a_src[j] = ok ? (pix * (unsigned)p.C + (unsigned)(pch * 8)) * 2u : OOB;
    }
  };
  plan_tile();
  const unsigned b_src = ((unsigned)(n0 + wave * BP * 8 + prow) * (unsigned)p.K + (unsigned)(pch * 8)) * 2u;   // + j * 8 rows
  const unsigned b_rstep = 8u * (unsigned)p.K * 2u;

  auto issue_a = [&](int cc, int abuf, bool live) {
    const unsigned coff = (unsigned)cc * 128u;
#pragma unroll
    for (int j = 0; j < AP; ++j) {
      const unsigned vo = (live && a_src[j] != OOB) ? a_src[j] + coff : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(sA + abuf * A_BYTES + (j * NW + wave) * 1024), 16, (int)vo, 0, 0, 0);
    }
  };
  auto issue_a1 = [&](int j, int cc, int abuf, bool live) {      // piece j of this wave only (GLDS_ASPREAD: one piece per tap)
    const unsigned coff = (unsigned)cc * 128u;
    const unsigned vo = (live && a_src[j] != OOB) ? a_src[j] + coff : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(sA + abuf * A_BYTES + (j * NW + wave) * 1024), 16, (int)vo, 0, 0, 0);
  };
  auto issue_b = [&](int tap, int cc, int bbuf, bool live) {
    const unsigned koff = (unsigned)(tap * p.C + cc * 64) * 2u;
#pragma unroll
    for (int j = 0; j < BP; ++j) {
      const unsigned vo = live ? b_src + (unsigned)j * b_rstep + koff : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(sB + bbuf * B_BYTES + (wave * BP + j) * 1024), 16, (int)vo, 0, 0, 0);
    }
  };

  // ---- fragment addresses
  // A-fragment byte offsets inside an image buffer for the three horizontal taps (vertical taps and the buffer are immediates)
  int a_adr[3][TM];
  int a_base[3];                                        // RST: row-block 0 of this wave for the three horizontal taps; block rb adds rb * 2048
  bool m_ok[TM];
  int m_pix[TM];                                        // pixel inside the tile (row of the staged output tile)
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
    if constexpr (RST) {
      const int q = wm * MW + mi * 16 + l15;
      const int y = q / PWL, x = q - y * PWL;
      m_ok[mi] = x < W_;
      m_pix[mi] = m_ok[mi] ? y * W_ + x : 0;
    } else {
      const int t = wm * MW + mi * 16 + l15;              // pixel inside the tile
      m_ok[mi] = t < PT;
      m_pix[mi] = t;
      const int tt = m_ok[mi] ? t : 0;
      const int y = tt / W_, x = tt - y * W_;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int r = y * PWL + x + dx;
        a_adr[dx][mi] = r * 128 + ((lg ^ (r & 7)) << 4);  // k-step 1 flips chunk bit 2: XOR 64
      }
    }
  }
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int r = wm * MW + l15 + dx;
    a_base[dx] = r * 128 + ((lg ^ (r & 7)) << 4);
  }
  int b_addr[TN];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
    const int row = wn * (BN / WN) + ni * 16 + l15;
    b_addr[ni] = row * 128 + ((lg ^ (row & 7)) << 4);      // ks = 1 flips chunk bit 2: XOR 64
  }

  f32x4_t acc[TN][TM];
  // MST (round 5): in-kernel stamps priced the forward epilogue at 4.2 us of a 29 us launch — 1.9 us of it VALU work on the accumulators (round to
  // 16 bit, back to fp32, mask, sum, square-sum: ~7 operations x 56 values per lane at 4 clocks x 2 waves per SIMD) + 1.1 us of DPP row sums and
  // 4-byte stores, all with the matrix cores idle.  The statistics are column sums of the STAGED tile Y (196 pixels x 128 channels, 16-bit, in LDS):
  //   sum_m Y[m][n]   = (ones x Y)[.][n]          sum_m Y[m][n]^2 = diag(Y^T Y)[n]
  // — wave w takes channels 16 w .. 16 w + 15: per 32-pixel step ONE transposed fragment (2 ds_read_b64_tr_b16) serves as A and B operand of the
  // Gram MFMA and as B of the ones MFMA; 7 steps = 14 reads + 14 MFMAs per wave for what took ~300 VALU operations per lane.  Products of two
  // 16-bit values are exact in fp32 and the accumulation is fp32 as before (another summation order).  One partial row per tile (was two).
  constexpr bool MST = GLDS_MFMA_STATS && !FUSED && PT == 196 && BN == 16 * NW;
  float ssum[TN][4], ssq[TN][4];                        // BatchNorm partials of this wave's columns (one tile)
  float* sStat = reinterpret_cast<float*>(smem + NABUF * A_BYTES + NB * B_BYTES);     // TPW > 1: [2 wave rows][2 statistics][BN], the workgroup's running partial row

  const int cpt = p.C >> 6;
  auto read_frags = [&](bf16x8_t (&fa)[TM], bf16x8_t (&fb)[TN], int aoff, int dx, const unsigned char* cB, int ks) {
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) fb[ni] = *reinterpret_cast<const bf16x8_t*>(cB + (b_addr[ni] ^ (ks * 64)));
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) fa[mi] = *reinterpret_cast<const bf16x8_t*>(sA + (a_adr[dx][mi] ^ (ks * 64)) + aoff);
  };
  auto mfma_all = [&](const bf16x8_t (&fa)[TM], const bf16x8_t (&fb)[TN]) {
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) acc[ni][mi] = MFMA16(fb[ni], fa[mi], acc[ni][mi]);
  };
#pragma unroll 1
  for (int ti = 0; ti < TPW; ++ti) {
  if (TPW > 1 && ti > 0) {
    bt = btw * TPW + ti;
    img = bt / TPI; y0 = (bt - img * TPI) * R_; m0 = bt * PT;
    plan_tile();
  }
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // FUSED: this thread's share of the BatchNorm input tile x[PT][BN] (same rows / columns as the output tile: chunk column c of rows rg,
  // rg + RGF, ...) is fetched into registers right behind the prologue's DMA (below) and rides through the K loop (28 VGPRs at 8 waves; the kernel
  // has one workgroup per CU, so 256 are there).  The first version fetched it by LDS-DMA after the loop and paid the round trip plus a barrier in
  // the open: the epilogue cost 13 us on a 28 us kernel, more than the separate reduction pass it replaces.
  constexpr int CPRF = BN / 8, RGF = NT / CPRF, XN = FUSED ? (PT + RGF - 1) / RGF : 1;
  uint4 xr[XN];
  // prologue: image of chunk 0, weight tiles of taps 0..2; wait for the image and tap 0, fetch tap 0's first fragments
  const bool skip_pro = (GLDS_ABLATE & 8) && TPW > 1 && ti > 0;      // timing ablation: what the second tile's prologue costs (it computes on stale LDS contents)
  if (!skip_pro) {
  issue_a(0, 0, true);
  issue_b(0, 0, 0, true);
  issue_b(RST ? 3 : 1, 0, 1, true);                      // RST walks the taps dx-major: (dy, dx) = (0,0), (1,0), (2,0), (0,1), ...
  issue_b(RST ? 6 : 2, 0, 2, true);
  }
  // (round 5) Where the x loads go: in front of the prologue's DMA they delayed the first barrier by 1.1 us per launch, right behind it still by
  // 0.8 us (in-kernel stamps: the eight waves' requests share one queue, 57 KB of x in front of the last wave's weights).  They are issued at the
  // END of the first tap (x_loads below), when only the loop's 16 KB per tap are in flight, and the counted waits of taps 1 and 2 leave them out.
  constexpr int XPRE = (FUSED && !XLATE) ? XN : 0;
  static_assert(XPRE == 0 || RST, "the fused instantiations run the padded-raster loop");
  auto x_loads = [&]() {
    if constexpr (XPRE > 0) {
      __builtin_amdgcn_sched_barrier(0);
      const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.bx, (unsigned)((size_t)p.M * p.N * 2));
      const int c_ = tid % CPRF, rg_ = tid / CPRF;
#pragma unroll
      for (int i = 0; i < XN; ++i) {
        const int row = rg_ + i * RGF;
        xr[i] = (GLDS_FUSED_ABLATE & 1) ? make_uint4(0, 0, 0, 0) : buf_load16(rsX, row < PT ? ((unsigned)(m0 + row) * (unsigned)p.N + (unsigned)(n0 + c_ * 8)) * 2u : OOB);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  bf16x8_t f0a[TM], f0b[TN], f1a[TM], f1b[TN];
  if (!skip_pro) glds_wait_vmcnt<2 * BP>();
  __builtin_amdgcn_s_barrier();
  GLDS_STAMP(1);
  if constexpr (RST) {
    bf16x8_t fa[2][NRB], fb[2][TN];
    auto rd_a = [&](int ks, int rb, int dx, int boff) {
      fa[ks][rb] = *reinterpret_cast<const bf16x8_t*>(sA + ((a_base[dx] ^ (ks * 64)) + rb * 2048 + boff));
    };
    auto rd_b = [&](int ks, const unsigned char* cB) {
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) fb[ks][ni] = *reinterpret_cast<const bf16x8_t*>(cB + (b_addr[ni] ^ (ks * 64)));
    };
#pragma unroll
    for (int rb = 0; rb < TM; ++rb) rd_a(0, rb, 0, 0);
    rd_b(0, sB);
    int bbuf = 0;                                          // ring slot of the current tap
    for (int cc2 = 0; cc2 < cpt; cc2 += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int cc = cc2 + h;
        if (ONECHUNK && h == 1) break;
        const bool more_c = cc + 1 < cpt;
        // one tap, iteration IT of the dx-major order (dx = IT / 3, dy = IT % 3; weight slice = tap dy * 3 + dx)
        auto tap_body = [&](auto IT_) {
          constexpr int it = decltype(IT_)::value;
          constexpr int dx = it / 3, dy = it % 3;
          constexpr int itn = it == 8 ? 0 : it + 1, dxn = itn / 3, dyn = itn % 3;
          constexpr int lo1 = dy == 0 ? 0 : TM + (dy - 1) * VG, hi1 = dy == 0 ? TM : TM + dy * VG;       // new row blocks of this tap
          constexpr int lo2 = dyn == 0 ? 0 : TM + (dyn - 1) * VG, hi2 = dyn == 0 ? TM : TM + dyn * VG;   // ... of the next tap
          constexpr int NR1 = hi1 - lo1 + TN, NR2 = hi2 - lo2 + TN, NM = TM * TN;
          const int boff = (ONECHUNK ? 0 : h) * A_BYTES;
          const int boffn = (ONECHUNK ? 0 : (it == 8 ? (h ^ 1) : h)) * A_BYTES;
          const unsigned char* cB = sB + bbuf * B_BYTES;
          // ---- first half: k-step 0 MFMAs; this tap's NEW k-step 1 fragments are read between them
          if (!(GLDS_ABLATE & 4)) {
#pragma unroll
            for (int rb = lo1; rb < hi1; ++rb) rd_a(1, rb, dx, boff);
            rd_b(1, cB);
          }
          GLDS_PRIO(1);
#pragma unroll
          for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) acc[ni][mi] = MFMA16(fb[0][ni], fa[0][mi + dy * VG], acc[ni][mi]);
          GLDS_PRIO(0);
#pragma unroll
          for (int i = 0; i < NR1; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, (NM / NR1) > 0 ? (NM / NR1) : 1, 0);   // MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                               // 1 DS read
          }
          __builtin_amdgcn_sched_barrier(0);
          if (!(GLDS_ABLATE & 17)) {
#if GLDS_ASPREAD
            // The next chunk's image goes out ONE piece per tap (taps 0 .. AP - 1, behind the tap's weight pieces) instead of AP pieces at tap 0.  This
            // tap waits for the weight slice of tap it + 1, issued at tap it - 2; younger than it: the image piece of tap it - 2, then tap it - 1's
            // weight slice and image piece (and, in the first chunk of a fused launch, the x tile requested at the END of tap 0).
            constexpr int NA = ONECHUNK ? 0 : ((it >= 2 && it - 2 < AP) ? 1 : 0) + ((it >= 1 && it - 1 < AP) ? 1 : 0);
            if (XPRE > 0 && (it == 1 || it == 2) && h == 0 && cc2 == 0) glds_wait_vmcnt<BP + NA + XPRE>();
            else glds_wait_vmcnt<BP + NA>();
#else
            if (!ONECHUNK && (it == 1 || it == 2)) {
              if (XPRE > 0 && h == 0 && cc2 == 0) glds_wait_vmcnt<BP + AP + XPRE>();      // (the x tile, requested at the end of tap 0, stays in flight)
              else glds_wait_vmcnt<BP + AP>();
            } else glds_wait_vmcnt<BP>();
#endif
          }
          if (!(GLDS_ABLATE & 2)) __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          // ---- second half: k-step 1 MFMAs; the next tap's new k-step 0 fragments and this tap's DMA issue between them
          const int nb = (bbuf + 1) & (NB - 1);
          if (!(GLDS_ABLATE & 4)) {
#pragma unroll
            for (int rb = lo2; rb < hi2; ++rb) rd_a(0, rb, dxn, boffn);
            rd_b(0, sB + nb * B_BYTES);
          }
          if (!(GLDS_ABLATE & 1)) {
            constexpr int i3 = it + 3, j3 = i3 >= 9 ? i3 - 9 : i3;
            const int cc3 = i3 >= 9 ? cc + 1 : cc;
            issue_b((j3 % 3) * 3 + j3 / 3, cc3, (bbuf + 3) & (NB - 1), cc3 < cpt);
#if GLDS_ASPREAD
            if constexpr (!ONECHUNK && it < AP) issue_a1(it, cc + 1, h ^ 1, more_c);
#else
            if (!ONECHUNK && it == 0) issue_a(cc + 1, h ^ 1, more_c);
#endif
          }
          GLDS_PRIO(1);
#pragma unroll
          for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) acc[ni][mi] = MFMA16(fb[1][ni], fa[1][mi + dy * VG], acc[ni][mi]);
          GLDS_PRIO(0);
          constexpr int NE = NR2 > BP ? NR2 : BP, MP2 = (NM / NE) > 0 ? (NM / NE) : 1;
#pragma unroll
          for (int i = 0; i < NE; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, MP2, 0);
            if (i < NR2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (i < BP + ((GLDS_ASPREAD && !ONECHUNK && it < AP) ? 1 : 0)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM (LDS-DMA piece); placing them evenly over the half's MFMAs instead measured equal (profiles/r06_ab_conv_dma_spread_v1.txt)
          }
          __builtin_amdgcn_sched_barrier(0);
          if (XPRE > 0 && it == 0 && h == 0 && cc2 == 0) x_loads();
          bbuf = nb;
        };
        tap_body(std::integral_constant<int, 0>{}); tap_body(std::integral_constant<int, 1>{}); tap_body(std::integral_constant<int, 2>{});
        tap_body(std::integral_constant<int, 3>{}); tap_body(std::integral_constant<int, 4>{}); tap_body(std::integral_constant<int, 5>{});
        tap_body(std::integral_constant<int, 6>{}); tap_body(std::integral_constant<int, 7>{}); tap_body(std::integral_constant<int, 8>{});
      }
    }
  } else {
  read_frags(f0a, f0b, 0, 0, sB, 0);

  // Per tap (K = 64 = two MFMA k-steps): [28 MFMA k-step 0 | ds_read k-step 1] [vmcnt + barrier: tap+1's weights landed, everybody
  // is done with tap-1's slot] [28 MFMA k-step 1 | ds_read tap+1 k-step 0 | DMA: weights of tap+3, at tap 0 the next chunk's image].
  int bbuf = 0;                                          // ring slot of the current tap
  for (int cc2 = 0; cc2 < cpt; cc2 += 2) {               // channel chunks in pairs: the image buffer index is a compile-time constant
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int cc = cc2 + h;
      if (ONECHUNK && h == 1) break;
      const bool more_c = cc + 1 < cpt;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int aoff = (ONECHUNK ? 0 : h) * A_BYTES + (tap / 3) * PWL * 128;       // buffer + vertical tap: immediates
        const int tapn = tap == 8 ? 0 : tap + 1;
        const int aoffn = (ONECHUNK ? 0 : (tap == 8 ? (h ^ 1) : h)) * A_BYTES + (tapn / 3) * PWL * 128;
        const unsigned char* cB = sB + bbuf * B_BYTES;
        // ---- first half: k-step 0 MFMAs, with the fragment reads of k-step 1 threaded between them
        if (!(GLDS_ABLATE & 4)) read_frags(f1a, f1b, aoff, tap % 3, cB, 1);
        GLDS_PRIO(1);
        mfma_all(f0a, f0b);
        GLDS_PRIO(0);
#pragma unroll
        for (int i = 0; i < NRD; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);   // MFMAs
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 DS read
        }
        __builtin_amdgcn_sched_barrier(0);
        // Loads younger than the tile we need (tap+1, issued two taps ago) stay in flight: the tile of tap+2 and the next
        // chunk's image when it was issued after it (at tap 0 of this chunk: seen from taps 1 and 2).
        if (!(GLDS_ABLATE & 1)) { if (!ONECHUNK && (tap == 1 || tap == 2)) glds_wait_vmcnt<BP + AP>(); else glds_wait_vmcnt<BP>(); }
        if (!(GLDS_ABLATE & 2)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- second half: k-step 1 MFMAs, threaded with the reads of tap+1 / k-step 0 and this tap's DMA issue
        const int nb = (bbuf + 1) & (NB - 1);
        if (!(GLDS_ABLATE & 4)) read_frags(f0a, f0b, aoffn, tapn % 3, sB + nb * B_BYTES, 0);
        if (!(GLDS_ABLATE & 1)) {
          const int t3 = tap + 3;
          const int tap3 = t3 >= 9 ? t3 - 9 : t3, cc3 = t3 >= 9 ? cc + 1 : cc;
          issue_b(tap3, cc3, (bbuf + 3) & (NB - 1), cc3 < cpt);
          if (!ONECHUNK && tap == 0) issue_a(cc + 1, h ^ 1, more_c);
        }
        GLDS_PRIO(1);
        mfma_all(f1a, f1b);
        GLDS_PRIO(0);
#pragma unroll
        for (int i = 0; i < BP; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);   // MFMAs
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 DS read
          __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // 1 VMEM (LDS-DMA piece)
        }
#pragma unroll
        for (int i = 0; i < NRD - 2 * BP; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        bbuf = nb;
      }
    }
  }
  }
  // FUSED: the per-channel coefficients of the epilogue's reduction are requested HERE, in front of the drain below, as 16-byte buffer loads:
  // left to hipcc they became 62 dword loads each issued right before its first use (19 exposed waits = 10 of the epilogue's 11 us).
  // fc[a][h]: array a (mean, rstd, gamma, beta, alpha), channels n0 + c * 8 + 4 h .. + 3; fm / fr: mean / rstd of column n0 + tid.
  constexpr bool MSTF = GLDS_MFMA_STATS && FUSED && PT == 196 && BN == 16 * NW;      // the fused epilogues' column sums on the matrix cores (below)
  constexpr bool OLDF = FUSED && (PAPPLY || !MSTF);                                   // the per-element reduction pass is compiled in
  float4 fc[OLDF ? 5 : 1][2];
  float fm = 0.f, fr = 0.f;
  auto load_coefs = [&](int z) {                            // z: zero (opaque in the several-tiles variant, so that nothing here is hoisted in front of the K loop)
    const unsigned cb = (unsigned)p.N * 4u;
    const unsigned co = (unsigned)(n0 + (tid % (BN / 8)) * 8 + z) * 4u;
    const float* arr[5] = {p.bmean, p.brstd, p.bgamma, p.bbeta, p.balpha};
    const float dflt[5] = {0.f, 1.f, 1.f, 0.f, 1.f};
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      if constexpr (XLATE) {
        // branch-free (a branch between these loads and the barrier below made hipcc wait for the x tile in front of it): without PReLU the
        // five vectors are read from `mean` and never used
        const __amdgpu_buffer_rsrc_t rs = make_rsrc((p.balpha && arr[a]) ? arr[a] : p.bmean, cb);
        const uint4 lo = buf_load16(rs, co), hi = buf_load16(rs, co + 16u);
        fc[a][0] = make_float4(__uint_as_float(lo.x), __uint_as_float(lo.y), __uint_as_float(lo.z), __uint_as_float(lo.w));
        fc[a][1] = make_float4(__uint_as_float(hi.x), __uint_as_float(hi.y), __uint_as_float(hi.z), __uint_as_float(hi.w));
      } else if (arr[a] && (a < 2 || p.balpha)) {            // gamma / beta / alpha matter to the PReLU variant only (wave-uniform)
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(arr[a], cb);
        const uint4 lo = buf_load16(rs, co), hi = buf_load16(rs, co + 16u);
        fc[a][0] = make_float4(__uint_as_float(lo.x), __uint_as_float(lo.y), __uint_as_float(lo.z), __uint_as_float(lo.w));
        fc[a][1] = make_float4(__uint_as_float(hi.x), __uint_as_float(hi.y), __uint_as_float(hi.z), __uint_as_float(hi.w));
      } else {
        fc[a][0] = fc[a][1] = make_float4(dflt[a], dflt[a], dflt[a], dflt[a]);
      }
    }
    if (!XLATE && tid < BN) { fm = p.bmean[n0 + tid + z]; fr = p.brstd[n0 + tid + z]; }     // (XLATE: read once, behind the tile loop)
  };
  // MSTF (round 5, after MST above): the fused epilogues' column sums on the matrix cores as well.  The BatchNorm input tile X joins the staged
  // tile in LDS (same layout, behind it) and per 32-pixel step one transposed fragment of each gives
  //   sum dy = (ones x DY)[.][n]      sum dy * x = diag(DY^T X)[n]      (forward raw moments: + diag(DY^T DY))
  // With a PReLU in front (p.balpha: dz = dy * (z <= 0 ? alpha : 1), z = x * sc + sh) the mask is not bilinear — but it is a THRESHOLD on x:
  // z <= 0  <=>  sgn(sc) x <= T = -sh / |sc|  <=>  t16 - sgn(sc) x >= 0 with t16 = the largest fp16 <= T, and the fp16 difference of two fp16 values
  // has the exact sign.  One packed FMA + one packed shift + one AND per PAIR of elements build DYPOS = (z > 0 ? dy : 0) on the fragment, and
  //   sum dz = S(dypos) + alpha (S(dy) - S(dypos)),  sum dz * x likewise,  sum_{z <= 0} dy * z = sc (S(dy x) - S(dypos x)) + sh (S(dy) - S(dypos))
  // (the bf16 build has no packed arithmetic on its storage type: it compares the two halves of a fragment register as fp32 against T itself — 7
  // operations per pair instead of 3; sphnet's PReLU-apply form, whose OUTPUT is dz, keeps the per-element pass: PAPPLY instantiations).
  const bool mstf = MSTF && !(PAPPLY && p.bmom == 2);
  if constexpr (OLDF && !XLATE) { if (!mstf) load_coefs(0); }
  // drain the (zero-writing) tail DMAs before the staging buffer is reused
  glds_wait_vmcnt<0>();
  if constexpr (XLATE) {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.bx, (unsigned)((size_t)p.M * p.N * 2));
    const int c_ = (tid + z) % CPRF, rg_ = (tid + z) / CPRF;  // (behind the opaque zero: as tile-loop invariants these would live through the K loop)
#pragma unroll
    for (int i = 0; i < XN; ++i) {
      const int row = rg_ + i * RGF;
      xr[i] = buf_load16(rsX, row < PT ? ((unsigned)(m0 + row) * (unsigned)p.N + (unsigned)(n0 + c_ * 8)) * 2u : OOB);
    }
    if constexpr (OLDF) { if (!mstf) load_coefs(z); }
  }
  // MSTF: this lane's channel (n0 + 16 wave + (lane & 15)) coefficients, requested now, used behind the MFMA chain
  float cf[5] = {0.f, 1.f, 1.f, 0.f, 1.f};
  if constexpr (MSTF) {
    if (mstf) {
      int z = 0;
      if constexpr (TPW > 1) asm volatile("v_mov_b32 %0, 0" : "=v"(z));
      const int ch = n0 + (((tid + z) >> 6) << 4) + ((tid + z) & 15);
      if (!p.bmom || p.balpha) { cf[0] = p.bmean[ch]; cf[1] = p.brstd[ch]; }
      if (p.balpha) { cf[2] = p.bgamma ? p.bgamma[ch] : 1.f; cf[3] = p.bbeta ? p.bbeta[ch] : 0.f; cf[4] = p.balpha[ch]; }
    }
  }
  // every wave is past its last fragment read (LDS only: __syncthreads() would also wait for the x / coefficient loads just issued — vmcnt(7) in the
  // several-tiles variant, whose x tile was requested a few instructions ago)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  constexpr int XOFF = NABUF * A_BYTES;                 // FUSED: the weight ring becomes the reduction scratch of the epilogue

  GLDS_STAMP(2);
  // ---- epilogue: bf16 tile through LDS; fragment rows >= 196 contribute nothing ----
  constexpr int CST = BN * 2 + 16;
  unsigned char* sC = smem;
#pragma unroll
  for (int ni = 0; ni < TN; ++ni)
#pragma unroll
    for (int q = 0; q < 4; ++q) ssum[ni][q] = ssq[ni][q] = 0.f;
  float esc_[TN][4], esh_[TN][4], eal_[TN][4];
  const bool has_esc = !FUSED && p.esc != nullptr;       // (the launcher refuses the output epilogue on a fused variant; compiled out of it: its 24
                                                         // conditional loads made every staging write of the fused two-tiles variant wait for the x tile)
  if (has_esc) {
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * (BN / WN) + ni * 16 + lg * 4 + q;
        esc_[ni][q] = p.esc[n]; esh_[ni][q] = p.esh[n]; eal_[ni][q] = p.ealpha ? p.ealpha[n] : 1.f;
      }
  }
  // (two copies of the staging loop behind ONE uniform branch: with the test inside, every one of the 56 values per lane paid a scalar branch
  // around the eval-mode affine — 62 s_cbranch in a phase that is pure overhead for the matrix cores)
  auto stage_tile = [&](auto ESC_) {
  constexpr bool ESC = decltype(ESC_)::value;
#pragma unroll
  for (int ni = 0; ni < TN; ++ni)
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
      int ml = m_pix[mi];
      bool mok = m_ok[mi];
      if constexpr (TPW > 1 && RST) {
        // recomputed per tile behind an opaque zero: as loop invariants of the tile loop the 14 values would be hoisted in front of the
        // main loop and live through it (256 VGPRs + spills instead of 194)
        int z;
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
        const int q_ = wm * MW + mi * 16 + l15 + z;
        const int y_ = q_ / PWL, x_ = q_ - y_ * PWL;
        mok = x_ < W_;
        ml = mok ? y_ * W_ + x_ : 0;
      }
      // (padded-raster kernels: the filler columns' lanes write to a dump row behind the tile instead of sitting out a predicated store — 14
      // s_and_saveexec / taken-branch pairs per lane in a phase that is pure overhead for the matrix cores; nothing reads row PT: the copy-out stops
      // in front of it, the transposed statistics reads select zeros beyond the tile, the x tile of the fused variants starts 1 KB-aligned behind it)
      constexpr bool DUMP = RST && (PT + 1) * (BN * 2 + 16) <= ((PT * (BN * 2 + 16) + 1023) & ~1023);
      if constexpr (DUMP) ml = mok ? ml : PT;
      const int nl = wn * (BN / WN) + ni * 16 + lg * 4;
      bf16_t h[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float a = acc[ni][mi][q];
        if constexpr (ESC) {                                 // eval-mode BatchNorm (+PReLU) of the output, on the fp32 accumulator
          a = a * esc_[ni][q] + esh_[ni][q];
          if (p.ealpha) a = a > 0.f ? a : eal_[ni][q] * a;
        }
        h[q] = f2bf(a);
        if constexpr (!FUSED && !MST) {                    // (the fused instantiations leave their statistics in the reduction pass below, never these)
          const float v = mok ? bf2f(h[q]) : 0.f;
          ssum[ni][q] += v;
          ssq[ni][q] += v * v;
        }
      }
      uint2 pk;
      pk.x = (unsigned)h[0] | ((unsigned)h[1] << 16);
      pk.y = (unsigned)h[2] | ((unsigned)h[3] << 16);
      if constexpr (XLATE) {
        // inline asm on purpose: the several-tiles fused variant has just requested its x tile, and hipcc — which cannot see that the asm waits of
        // the K loop retired every LDS-DMA — puts a counted vmcnt wait in front of each LDS store it emits itself: 5 of the 7 x loads, in the open
        typedef __attribute__((ext_vector_type(2))) unsigned st8_u2_t;
        const st8_u2_t pv = {pk.x, pk.y};
        const unsigned sadr = (unsigned)reinterpret_cast<size_t>((lds_ptr_t)(sC + ml * CST + nl * 2));
        if (DUMP || mok) asm volatile("ds_write_b64 %0, %1" ::"v"(sadr), "v"(pv) : "memory");
      } else {
        if (DUMP || mok) *reinterpret_cast<uint2*>(sC + ml * CST + nl * 2) = pk;
      }
    }
  };
  if (has_esc) stage_tile(std::true_type{}); else stage_tile(std::false_type{});
  GLDS_STAMP(4);                                        // (staged: conversions, statistics sums and LDS writes of this wave are issued)
  if (!MST && TPW > 1 && p.stats) {                     // this tile's partials join the workgroup's row in LDS (a region no DMA touches)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float a = row16_sum(ssum[ni][q]), b = row16_sum(ssq[ni][q]);
        const int col = wn * (BN / WN) + ni * 16 + lg * 4 + q;
        if (l15 == 0) {                                   // one writer per (wave row, column): no race, fixed summation order
          float* d = sStat + wm * 2 * BN + col;
          d[0] = ti == 0 ? a : d[0] + a;
          d[BN] = ti == 0 ? b : d[BN] + b;
        }
      }
  }
  if (!MST && TPW == 1 && p.stats) {
    const int ntile = gridDim.x / p.nbn;
    float* prow_ = p.stats + (size_t)(bt * WM + wm) * 2 * p.N;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float a = row16_sum(ssum[ni][q]), b = row16_sum(ssq[ni][q]);
        const int n = n0 + wn * (BN / WN) + ni * 16 + lg * 4 + q;
        if (l15 == 0) {
          prow_[n] = a;
          prow_[p.N + n] = b;
        }
      }
    // the BatchNorm finalize sums gemm_nt_stat_rows(M, N) partial rows (128-pixel tiling): zero the ones this tiling leaves
    for (int row = ntile * WM + bt * WM + wm; row < stat_rows; row += ntile * WM)
      if (lane < 2 * (BN / WN / 4)) {                       // (sum, sumsq) x this wave's BN/WN columns, one float4 per lane
        constexpr int LPR = BN / WN / 4;
        float* z = p.stats + (size_t)row * 2 * p.N + (lane / LPR) * p.N + n0 + wn * (BN / WN);
        *reinterpret_cast<float4*>(z + (lane % LPR) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
  }
  constexpr int SXOFF = (PT * CST + 1023) & ~1023;
  static_assert(!MSTF || 2 * SXOFF <= NABUF * A_BYTES + NB * B_BYTES, "staged tile + BatchNorm input tile must fit below the statistics row");
  if constexpr (MSTF) {
    if (mstf) {
      int tz = tid;
      if constexpr (XLATE) {
        int z;
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
        tz += z;
      }
      const int c = tz % CPRF, rg = tz / CPRF;
#pragma unroll
      for (int i = 0; i < XN; ++i) {
        const int row = rg + i * RGF;
        if ((i + 1) * RGF <= PT || row < PT) *reinterpret_cast<uint4*>(smem + SXOFF + row * CST + c * 16) = xr[i];
      }
    }
  }
  GLDS_STAMP(5);
  if constexpr (XLATE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the asm staging writes above)
  __syncthreads();
  GLDS_STAMP(6);
  constexpr int CPR = BN / 8;
  if constexpr (MST) {
    // (indices behind an opaque zero when a workgroup loops over tiles: as tile-loop invariants they would be hoisted in front of the K loop and
    // live through it — the two-tiles instantiation spilled with them)
    int tidz = tid;
    if constexpr (TPW > 1) {
      int z;
      asm volatile("v_mov_b32 %0, 0" : "=v"(z));
      tidz += z;
    }
    const int l15z = tidz & 15, lgz = (tidz >> 4) & 3, wavez = tidz >> 6;
    if (!(TPW == 1 && (p.eadd || p.Cb2))) {
      // plain tile: all LDS reads first, then the stores (the store helper is an asm statement the compiler will not move a load across)
      constexpr int NIT = (PT * CPR + NT - 1) / NT;
      uint4 v[NIT];
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int idx = tidz + k * NT;
        const int row = idx / CPR, c = idx - row * CPR;
        if ((k + 1) * NT <= PT * CPR || idx < PT * CPR) v[k] = *reinterpret_cast<const uint4*>(sC + row * CST + c * 16);
      }
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int idx = tidz + k * NT;
        const int row = idx / CPR, c = idx - row * CPR;
        if ((k + 1) * NT <= PT * CPR || idx < PT * CPR) glds_store_out(p.Cb + (size_t)(m0 + row) * p.ldc + n0 + c * 8, v[k]);
      }
    }
    if (p.stats) {
      typedef __attribute__((address_space(3))) s16x4_t* lds_tr_p;
      constexpr int KS = (PT + 31) / 32;
      // lane (g = lane >> 4, li = lane & 15) reads 8 B of row 8 g + (li >> 2) (+ 4 for the second read), columns 4 (li & 3) .. + 3 of this wave's 16;
      // the transposing read hands lane li column li's values of rows 8 g .. 8 g + 3 — a 16 x 16 x 32 operand fragment of Y^T / Y
      const unsigned char* tb = sC + (8 * lgz + (l15z >> 2)) * CST + (wavez * 16 + 4 * (l15z & 3)) * 2;
      f32x4_t gq = {0.f, 0.f, 0.f, 0.f}, gs = {0.f, 0.f, 0.f, 0.f};
      s16x8_t one8;
#pragma unroll
      for (int j = 0; j < 8; ++j) one8[j] = FEDFR_FP16 ? (short)0x3c00 : (short)0x3f80;
      const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, one8);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(tb + ks * 32 * CST));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(tb + ks * 32 * CST + 4 * CST));
        if ((ks + 1) * 32 > PT) {                            // rows beyond the tile hold stale LDS contents: select zeros (PT % 4 == 0: a group of 4 rows is in or out)
          const bool lo_ok = ks * 32 + 8 * lgz < PT, hi_ok = ks * 32 + 8 * lgz + 4 < PT;
#pragma unroll
          for (int j = 0; j < 4; ++j) { lo[j] = lo_ok ? lo[j] : (short)0; hi[j] = hi_ok ? hi[j] : (short)0; }
        }
        s16x8_t f8;
        f8[0] = lo[0]; f8[1] = lo[1]; f8[2] = lo[2]; f8[3] = lo[3];
        f8[4] = hi[0]; f8[5] = hi[1]; f8[6] = hi[2]; f8[7] = hi[3];
        const bf16x8_t fr = __builtin_bit_cast(bf16x8_t, f8);
        gq = MFMA16(fr, fr, gq);                               // Y^T Y: lane holds rows 4 g .. 4 g + 3 of column li
        gs = MFMA16(ones, fr, gs);                             // every row = the column sums
      }
      // the diagonal of the 16 x 16 Gram block lives in the 16 lanes with (li >> 2) == g, register li & 3
      const int qd = l15z & 3;
      const float sq = qd == 0 ? gq[0] : qd == 1 ? gq[1] : qd == 2 ? gq[2] : gq[3];
      if ((l15z >> 2) == lgz) {
        const int col = wavez * 16 + l15z;
        if constexpr (TPW == 1) {
          float* prow_ = p.stats + (size_t)(bt * WM) * 2 * p.N + n0 + col;
          prow_[0] = gs[0];
          prow_[p.N] = sq;
          prow_[2 * (size_t)p.N] = 0.f;                        // the tile's second row slot (the register path left one row per wave row)
          prow_[3 * (size_t)p.N] = 0.f;
        } else {
          float* d = sStat + col;                              // the workgroup's running row: own column, no race, fixed order
          d[0] = ti == 0 ? gs[0] : d[0] + gs[0];
          d[BN] = ti == 0 ? sq : d[BN] + sq;
        }
      }
      if constexpr (TPW == 1) {
        // the BatchNorm finalize sums gemm_nt_stat_rows(M, N) partial rows (128-pixel tiling): zero the ones this tiling leaves
        const int ntile = gridDim.x / p.nbn;
        for (int row = ntile * WM + bt * WM + wm; row < stat_rows; row += ntile * WM)
          if (lane < 2 * (BN / WN / 4)) {
            constexpr int LPR = BN / WN / 4;
            float* z = p.stats + (size_t)row * 2 * p.N + (lane / LPR) * p.N + n0 + wn * (BN / WN);
            *reinterpret_cast<float4*>(z + (lane % LPR) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
          }
      }
    }
  }
  if constexpr (!FUSED) {
    if (!MST || (TPW == 1 && (p.eadd || p.Cb2)))
    for (int idx = tid; idx < PT * CPR; idx += NT) {
      const int row = idx / CPR, c = idx - row * CPR;
      uint4 v = *reinterpret_cast<const uint4*>(sC + row * CST + c * 16);
      const size_t go = (size_t)(m0 + row) * p.ldc + n0 + c * 8;
      if (TPW == 1 && (p.eadd || p.Cb2)) {                 // (compiled out of the two-tiles-per-workgroup instantiation, which sits at the register limit: with
                                                           // this block in it spilled — 256 VGPRs + 128 B of scratch per lane, 34 -> 51 us per 28x28 conv; round 3)
        float f[8];
        unpack8(v, f);
        if (p.eadd) {                                        // + identity path (one more bf16 rounding than the separate bn_apply pass)
          float g[8];
          unpack8(*reinterpret_cast<const uint4*>(p.eadd + go), g);
#pragma unroll
          for (int q = 0; q < 8; ++q) f[q] += g[q];
          v = pack8(f);
          unpack8(v, f);
        }
        if (p.Cb2) {                                         // the next block's bn1 of the (bf16) output; or sphnet's activation (gemm.h)
          float y2[8];
          if (p.e2alpha || !p.esc2) {
            float a2[8];
            if (p.e2add) unpack8(*reinterpret_cast<const uint4*>(p.e2add + go), a2);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int n = n0 + c * 8 + q;
              float t = f[q] * (p.esc2 ? p.esc2[n] : 1.f) + (p.esh2 ? p.esh2[n] : 0.f);
              if (p.e2alpha) t = t > 0.f ? t : p.e2alpha[n] * t;
              y2[q] = p.e2add ? t + a2[q] : t;
            }
          } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) y2[q] = f[q] * p.esc2[n0 + c * 8 + q] + p.esh2[n0 + c * 8 + q];
          }
          *reinterpret_cast<uint4*>(p.Cb2 + go) = pack8(y2);
        }
      }
      glds_store_out(p.Cb + go, v);
    }
  } else if (mstf) {
    if constexpr (MSTF) {
    int tidz = tid;
    if constexpr (TPW > 1) {
      int z;
      asm volatile("v_mov_b32 %0, 0" : "=v"(z));
      tidz += z;
    }
    const int l15z = tidz & 15, lgz = (tidz >> 4) & 3, wavez = tidz >> 6;
    const bool diag = (l15z >> 2) == lgz;
    const int col = wavez * 16 + l15z;
    // (the coefficient arithmetic comes BEFORE the copy-out: vmcnt counts loads and stores on gfx9 and they retire out of order with each other,
    // so a wait for the coefficient loads placed behind the tile's stores is a vmcnt(0) that sits out the write-through stores — ~1 us, asm-checked)
    const bool prelu = p.balpha != nullptr;
    // PReLU threshold of this lane's channel (epi_mfma.h): z = x * sc + sh, sc = gamma * rstd, sh = beta - mean * sc
    const float sc_ = cf[2] * cf[1], sh_ = cf[3] - cf[0] * sc_;
    PreluThr th = {0u, 0u, 0.f, 0.f};
    if (prelu) th = prelu_threshold(sc_, sh_);
    asm volatile("" ::"v"(th.t162), "v"(th.nsgn2), "v"(sc_), "v"(sh_), "v"(th.Tf), "v"(th.nsf), "v"(cf[0]), "v"(cf[1]), "v"(cf[4]));     // (materialised here)
    __builtin_amdgcn_sched_barrier(0);
    {
      constexpr int NIT = (PT * CPR + NT - 1) / NT;
      uint4 v[NIT];
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int idx = tidz + k * NT;
        const int row = idx / CPR, c = idx - row * CPR;
        if ((k + 1) * NT <= PT * CPR || idx < PT * CPR) v[k] = *reinterpret_cast<const uint4*>(sC + row * CST + c * 16);
      }
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int idx = tidz + k * NT;
        const int row = idx / CPR, c = idx - row * CPR;
        if ((k + 1) * NT <= PT * CPR || idx < PT * CPR) glds_store_out(p.Cb + (size_t)(m0 + row) * p.ldc + n0 + c * 8, v[k]);
      }
    }
    typedef __attribute__((address_space(3))) s16x4_t* lds_tr_p;
    constexpr int KS = (PT + 31) / 32;
    const unsigned char* tb = sC + (8 * lgz + (l15z >> 2)) * CST + (wavez * 16 + 4 * (l15z & 3)) * 2;
    f32x4_t g1 = {0.f, 0.f, 0.f, 0.f}, g2 = {0.f, 0.f, 0.f, 0.f}, g3 = {0.f, 0.f, 0.f, 0.f}, g4 = {0.f, 0.f, 0.f, 0.f};
    s16x8_t one8;
#pragma unroll
    for (int j = 0; j < 8; ++j) one8[j] = FEDFR_FP16 ? (short)0x3c00 : (short)0x3f80;
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, one8);
    const bool mom = p.bmom != 0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(tb + ks * 32 * CST));
      s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(tb + ks * 32 * CST + 4 * CST));
      s16x4_t xl = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(tb + SXOFF + ks * 32 * CST));
      s16x4_t xh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(tb + SXOFF + ks * 32 * CST + 4 * CST));
      if ((ks + 1) * 32 > PT) {
        const bool lo_ok = ks * 32 + 8 * lgz < PT, hi_ok = ks * 32 + 8 * lgz + 4 < PT;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          lo[j] = lo_ok ? lo[j] : (short)0; hi[j] = hi_ok ? hi[j] : (short)0;
          xl[j] = lo_ok ? xl[j] : (short)0; xh[j] = hi_ok ? xh[j] : (short)0;
        }
      }
      s16x8_t d8, x8;
      d8[0] = lo[0]; d8[1] = lo[1]; d8[2] = lo[2]; d8[3] = lo[3]; d8[4] = hi[0]; d8[5] = hi[1]; d8[6] = hi[2]; d8[7] = hi[3];
      x8[0] = xl[0]; x8[1] = xl[1]; x8[2] = xl[2]; x8[3] = xl[3]; x8[4] = xh[0]; x8[5] = xh[1]; x8[6] = xh[2]; x8[7] = xh[3];
      const bf16x8_t dfr = __builtin_bit_cast(bf16x8_t, d8), xfr = __builtin_bit_cast(bf16x8_t, x8);
      g1 = MFMA16(ones, dfr, g1);                            // every row: sum over the pixels of dy (forward: y)
      g2 = MFMA16(dfr, xfr, g2);                             // diagonal: sum dy * x
      if (mom) g3 = MFMA16(dfr, dfr, g3);                    // forward raw moments: sum y * y
      if (prelu) {
        const s16x8_t pu = prelu_pos(d8, x8, th);
        const bf16x8_t pfr = __builtin_bit_cast(bf16x8_t, pu);
        g3 = MFMA16(ones, pfr, g3);                          // sum dypos
        g4 = MFMA16(pfr, xfr, g4);                           // diagonal: sum dypos * x
      }
    }
    const int qd = l15z & 3;
    float t0 = g1[0];
    float t1 = qd == 0 ? g2[0] : qd == 1 ? g2[1] : qd == 2 ? g2[2] : g2[3];
    float t2 = qd == 0 ? g3[0] : qd == 1 ? g3[1] : qd == 2 ? g3[2] : g3[3];
    if (prelu) {
      const float sp = g3[0], spx = qd == 0 ? g4[0] : qd == 1 ? g4[1] : qd == 2 ? g4[2] : g4[3];
      const float sn = t0 - sp, snx = t1 - spx;              // sums over the elements with z <= 0
      t0 = sp + cf[4] * sn;
      t1 = spx + cf[4] * snx;
      t2 = sc_ * snx + sh_ * sn;
    }
    const float fm_ = cf[0], fr_ = cf[1];
    if (diag) {
      if constexpr (XLATE) {
        float* d = sStat + col;
        d[0] = ti == 0 ? t0 : d[0] + t0;
        d[BN] = ti == 0 ? t1 : d[BN] + t1;
        d[2 * BN] = ti == 0 ? t2 : d[2 * BN] + t2;
      } else {
        float* o = p.bpart + (size_t)bt * 3 * p.N + n0 + col;
        o[0] = t0;
        o[p.N] = p.bmom ? t1 : fr_ * (t1 - fm_ * t0);
        o[2 * (size_t)p.N] = t2;
      }
    }
    }
  } else if constexpr (OLDF) {
    // ---- fused BN-backward reduction (ew_bn_bwd_reduce on this tile), per-element form (PAPPLY instantiations: sphnet's PReLU-apply epilogue):
    // thread owns chunk column c (8 channels) of rows rg, rg + RG, ...;
    // dy from the staged tile, x from the registers loaded before the K loop.  The workgroup owns the CU while it does this, so the pass is
    // kept to the fewest VALU operations: it accumulates sum dz and sum dz * x on the RAW x (one FMA and one add per element without
    // PReLU; with PReLU its input z = x * scale + shift costs one more FMA, a compare and three selects), and the tile's
    // sum dz * xhat = rstd (sum dz * x - mean sum dz) is formed once per column in the reduction below.
    constexpr int RG = RGF;
    int tz = tid;
    if constexpr (XLATE) {
      int z;
      asm volatile("v_mov_b32 %0, 0" : "=v"(z));
      tz += z;
    }
    const int c = tz % CPR, rg = tz / CPR;
    const int n = n0 + c * 8;
    float s1[8], s2[8], s3[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) s1[q] = s2[q] = s3[q] = 0.f;
    if (p.balpha) {
      float sc[8], sh[8], al[8];
      const bool papply = p.bmom == 2;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        auto at = [&](int a) { const float4 v = fc[a][q >> 2]; return (q & 3) == 0 ? v.x : (q & 3) == 1 ? v.y : (q & 3) == 2 ? v.z : v.w; };
        // bmom == 2 (sphnet): a bare PReLU(+bias) sits in front — z = x + bias (carried in bbeta), no statistics
        const float g = papply ? 1.f : at(2) * at(1);
        sc[q] = g;
        sh[q] = papply ? (p.bbeta ? at(3) : 0.f) : at(3) - at(0) * g;      // (no bias: the several-tiles variant's branch-free loads read `mean` in its place)
        al[q] = at(4);
      }
#pragma unroll
      for (int i = 0; i < XN; ++i) {
        const int row = rg + i * RG;
        if (row < PT) {
          const uint4 dv = *reinterpret_cast<const uint4*>(sC + row * CST + c * 16);
          if (!papply) glds_store_out(p.Cb + (size_t)(m0 + row) * p.ldc + n, dv);
          if (GLDS_FUSED_ABLATE & 2) continue;
          float dy[8], xv[8], dzv[8];
          unpack8(dv, dy);
          unpack8(xr[i], xv);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float z = xv[q] * sc[q] + sh[q];
            const bool neg = z <= 0.f;
            s3[q] += neg ? dy[q] * z : 0.f;
            const float dz = neg ? dy[q] * al[q] : dy[q];
            dzv[q] = dz;
            s1[q] += dz;
            s2[q] += dz * xv[q];
          }
          if (papply) glds_store_out(p.Cb + (size_t)(m0 + row) * p.ldc + n, pack8(dzv));   // ... and the OUTPUT is the masked gradient
        }
      }
    } else if (p.bmom) {                                   // forward: raw moments (sum y, sum y x, sum y y)
#pragma unroll
      for (int i = 0; i < XN; ++i) {
        const int row = rg + i * RG;
        if (row < PT) {
          const uint4 dv = *reinterpret_cast<const uint4*>(sC + row * CST + c * 16);
          glds_store_out(p.Cb + (size_t)(m0 + row) * p.ldc + n, dv);
          float dy[8], xv[8];
          unpack8(dv, dy);
          unpack8(xr[i], xv);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            s1[q] += dy[q];
            s2[q] += dy[q] * xv[q];
            s3[q] += dy[q] * dy[q];
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < XN; ++i) {
        const int row = rg + i * RG;
        if (row < PT) {
          const uint4 dv = *reinterpret_cast<const uint4*>(sC + row * CST + c * 16);
          glds_store_out(p.Cb + (size_t)(m0 + row) * p.ldc + n, dv);
          if (GLDS_FUSED_ABLATE & 2) continue;
          float dy[8], xv[8];
          unpack8(dv, dy);
          unpack8(xr[i], xv);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            s1[q] += dy[q];
            s2[q] += dy[q] * xv[q];
          }
        }
      }
    }
    if (GLDS_FUSED_ABLATE & 4) return;
    // the weight ring (drained before the staging above) is the reduction scratch
    float* red = reinterpret_cast<float*>(smem + XOFF);      // [RG][3][BN]
    *reinterpret_cast<float4*>(red + (rg * 3 + 0) * BN + c * 8) = make_float4(s1[0], s1[1], s1[2], s1[3]);
    *reinterpret_cast<float4*>(red + (rg * 3 + 0) * BN + c * 8 + 4) = make_float4(s1[4], s1[5], s1[6], s1[7]);
    *reinterpret_cast<float4*>(red + (rg * 3 + 1) * BN + c * 8) = make_float4(s2[0], s2[1], s2[2], s2[3]);
    *reinterpret_cast<float4*>(red + (rg * 3 + 1) * BN + c * 8 + 4) = make_float4(s2[4], s2[5], s2[6], s2[7]);
    *reinterpret_cast<float4*>(red + (rg * 3 + 2) * BN + c * 8) = make_float4(s3[0], s3[1], s3[2], s3[3]);
    *reinterpret_cast<float4*>(red + (rg * 3 + 2) * BN + c * 8 + 4) = make_float4(s3[4], s3[5], s3[6], s3[7]);
    __syncthreads();
    if (tid < BN) {                                          // one thread per column: the three sums over the row groups
      float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll 8
      for (int r = 0; r < RG; ++r) {
        t0 += red[(r * 3 + 0) * BN + tid];
        t1 += red[(r * 3 + 1) * BN + tid];
        t2 += red[(r * 3 + 2) * BN + tid];
      }
      if constexpr (XLATE) {                               // the workgroup's running row (own column: no race, fixed summation order)
        float* d = sStat + tid;
        d[0] = ti == 0 ? t0 : d[0] + t0;
        d[BN] = ti == 0 ? t1 : d[BN] + t1;
        d[2 * BN] = ti == 0 ? t2 : d[2 * BN] + t2;
      } else {
        float* o = p.bpart + (size_t)bt * 3 * p.N + n0 + tid;
        o[0] = t0;
        o[p.N] = p.bmom == 2 ? t2 : (p.bmom ? t1 : fr * (t1 - fm * t0));     // (bmom == 2: rows (sum dz, sum dy z over z <= 0), what a PReLU's parameter sums need)
        o[2 * (size_t)p.N] = t2;
      }
    }
  }
  if (TPW > 1) __syncthreads();                         // the staged output tile has been read: the next tile's image may land on it
  }   // tiles of this workgroup
  GLDS_STAMP(7);
  if constexpr (XLATE) {
    if (tid < BN) {
      const float t0 = sStat[tid], t1 = sStat[BN + tid], t2 = sStat[2 * BN + tid];
      float* o = p.bpart + (size_t)btw * 3 * p.N + n0 + tid;
      o[0] = t0;
      o[p.N] = p.bmom == 2 ? t2 : (p.bmom ? t1 : p.brstd[n0 + tid] * (t1 - p.bmean[n0 + tid] * t0));
      o[2 * (size_t)p.N] = t2;
    }
  }
  if constexpr (TPW > 1) {
    // one partial row per workgroup: wave row 0 + wave row 1 of the LDS partials (every tile's epilogue ended with a barrier)
    if (p.stats) {
      const int nrows = gridDim.x / p.nbn;
      for (int i = tid; i < 2 * BN; i += NT) {
        const int stat = i / BN, col = i - stat * BN;
        p.stats[(size_t)btw * 2 * p.N + (size_t)stat * p.N + n0 + col] = MST ? sStat[i] : sStat[i] + sStat[2 * BN + i];
      }
      // rows the finalize kernel's row count (128-pixel tiling) has beyond ours: zeros
      for (int row = nrows + btw; row < stat_rows; row += nrows)
        for (int i = tid; i < 2 * BN / 4; i += NT) {
          const int stat = i / (BN / 4), c4 = i - stat * (BN / 4);
          *reinterpret_cast<float4*>(p.stats + (size_t)row * 2 * p.N + (size_t)stat * p.N + n0 + c4 * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
  }
  GLDS_STAMP(3);
}

template <int W_, int R_, int NPA, int WN, bool FUSED, int BN_ = 128, bool ONECHUNK = false, int TPW = 1, bool PAPPLY = false>
static int launch_glds(GemmNT p, hipStream_t st) {
  constexpr int PT = R_ * W_;
  FEDFR_REQUIRE(p.N % BN_ == 0 && (ONECHUNK ? p.C == 64 : p.C % 128 == 0) && p.H == W_ && p.W == W_ && p.M % (W_ * W_) == 0 && p.K == 9 * p.C && p.ldc % 8 == 0,
                "conv3x3_glds: unsupported shape (N=%d C=%d H=%d W=%d M=%d)", p.N, p.C, p.H, p.W, p.M);
  FEDFR_REQUIRE(FUSED == (p.bpart != nullptr), "conv3x3_glds: fused / plain variant mismatch");
  if (FUSED) {
    FEDFR_REQUIRE(p.bx && p.bmean && p.brstd && p.ldc == p.N, "conv3x3_glds: fused BN-bwd reduction needs bx / mean / rstd and ldc == N");
    static_assert(!FUSED || (size_t)(128 * WN / 16) * 3 * 128 * 4 <= 4 * (size_t)128 * 128, "reduction scratch must fit the weight ring");
    FEDFR_REQUIRE(!p.stats, "conv3x3_glds: a fused-epilogue launch leaves reduction rows (bpart), not forward statistics");
    FEDFR_REQUIRE(p.bmom != 2 || p.balpha, "conv3x3_glds: the PReLU-apply epilogue needs the slopes");
    FEDFR_REQUIRE(p.bmom != 2 || PAPPLY, "conv3x3_glds: the PReLU-apply epilogue lives in the PAPPLY instantiations");
    if (p.bwd_fused) *p.bwd_fused = p.M / PT / TPW;
  }
  p.nbn = p.N / BN_;
  FEDFR_REQUIRE((p.M / PT) % TPW == 0, "conv3x3_glds: %d tiles do not split into groups of %d", p.M / PT, TPW);
  const int ntile = p.M / PT / TPW;                     // workgroups per output-channel tile
  constexpr size_t lds = (ONECHUNK ? 1 : 2) * (size_t)NPA * 1024 + 4 * (size_t)BN_ * 128 + (TPW > 1 ? 4 * (size_t)BN_ * 4 : 0);
  static_assert(lds >= (size_t)PT * (BN_ * 2 + 16) && lds <= 160 * 1024, "LDS budget");
  static PerDeviceOnce attr_once;     // hipFuncSetAttribute is per device (a Server process may drive several)
  attr_once.run([&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_glds_kernel<W_, R_, NPA, WN, FUSED, BN_, ONECHUNK, TPW, PAPPLY>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  ProfScope prof(W_ == 14 ? 12 : (W_ == 28 ? 13 : 15),   /* slot 15: 56x56 and 112x112 */ 2.0 * p.M * p.N * (double)p.K, st, gemm_nt_alg_bytes(p, 1));
  hipLaunchKernelGGL((conv3x3_glds_kernel<W_, R_, NPA, WN, FUSED, BN_, ONECHUNK, TPW, PAPPLY>), dim3(ntile * p.nbn), dim3(128 * WN), lds, st, p, gemm_nt_stat_rows(p.M, p.N));
  FEDFR_LAUNCH_CHECK("conv3x3_glds");
  return FEDFR_OK;
}
