// NT implicit GEMM (gemm.h: GemmNT) whose two operands reach LDS by LDS-DMA: the general-shape sibling of the 3x3 / stride-1 kernels in
// conv_glds_impl.h, for what is left on the register-staged gemm_nt_kernel: the 7x7 stage's 3x3 convs (fwd + dgrad), the stride-2 3x3
// convs (fwd, and dgrad by output-parity class), the 1x1 / stride-2 downsample convs, plain [M][K] x [N][K]^T products.
//
//   * Same tile and LDS layout as gemm_nt_kernel (BM x 64 and BN x 64 bf16 slices, one 128-B row per pixel / output channel, 16-B chunk
//     index XOR-ed by row & 7), so fragment reads and the epilogue (nt_epilogue.h) are shared.  What changes is how a K-step's slices get
//     there: `buffer_load_dwordx4 ... lds` straight from global into a ring of NS stages, NS - 1 K-steps in flight, retired with a
//     COUNTED s_waitcnt vmcnt + one raw s_barrier per K-step.  The register-staged kernel's loads have a register destination that hipcc
//     sinks to just before the ds_write, so each K-step paid an L2 round trip (in-kernel stamps, round 1); LDS-DMA has none.
//   * A wave-instruction writes 1 KiB of LDS linearly (lane l -> base + 16 l) = 8 rows x 8 chunks, so the swizzle is applied to the
//     per-lane SOURCE: lane l of a piece fetches row (l >> 3) of the piece, logical chunk (l & 7) ^ (l >> 3).  A gathered row that falls
//     outside the image (padding, inserted zeros of a dgrad, M / N / K tails) is fetched from an out-of-range buffer offset, which the
//     hardware turns into zeros in LDS: no masks, no zero-fill pass.
//   * Beyond the last K-step the pipeline keeps issuing (out-of-range) pieces, so the vmcnt immediates are the same in every iteration.
#include "nt_epilogue.h"

#ifndef NT_GLDS_NS
#define NT_GLDS_NS 4     // ring stages of the 8-wave 128 x 128 tile (5 = all 160 KB of LDS)
#endif
namespace {
template <int N_>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

template <int BM, int BN, int WM, int WN, int NS, bool CONV>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_glds_kernel(GemmNT p_) {
  GemmNT p = p_;
  if (p.par_on == 2) {                                  // the four output-parity classes of a stride-2 dgrad in one launch (see gemm_nt_kernel)
    const int cls = 3 - (int)blockIdx.z;
    p.par_h = cls >> 1; p.par_w = cls & 1;
    p.ksteps_total = (1 + p.par_h) * (1 + p.par_w) * p.cpt;
    p.ksteps_per_split = p.ksteps_total;
  }
  constexpr int NW = WM * WN, NT = 64 * NW;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int AP = (BM / 8) / NW, BP = (BN / 8) / NW, PER = AP + BP;     // LDS-DMA pieces (8 rows) per wave and K-step
  constexpr int NRD = TM + TN, NM = TM * TN;
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0 && NS >= 3 && NS <= 5 && AP >= 1 && BP >= 1, "tile geometry");
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bn = lid % p.nbn, bm = lid / p.nbn;
  const int split = blockIdx.y;
  const int m0 = bm * BM, n0 = bn * BN;
  const int kt0 = split * p.ksteps_per_split;
  const int kt1 = min(kt0 + p.ksteps_per_split, p.ksteps_total);

  // ---- LDS-DMA source plan: piece j * NW + wave of a slice = its rows 8 (j * NW + wave) .. + 7
  const int prow = lane >> 3, pch = (lane & 7) ^ prow;
  constexpr unsigned OOB = 0xfffffff0u;
  // A row of this lane in piece j: gather state (conv) or row offset (plain).  Everything below is branch-free on purpose: a branch in
  // the K loop ends the scheduling region, and the MFMA / ds_read / LDS-DMA interleave (sched_group_barrier) only works inside one.
  int a_hb[AP], a_wb[AP];
  unsigned a_base[AP];                                  // byte offset of (image, chunk column) resp. of the row
  bool a_ok[AP];
#pragma unroll
  for (int j = 0; j < AP; ++j) {
    const int m = m0 + (j * NW + wave) * 8 + prow;
    a_ok[j] = m < p.M;
    if (CONV) {
      const int mm = a_ok[j] ? m : 0;
      const int hw = p.Ho * p.Wo;
      const int img = mm / hw, rem = mm - img * hw;
      int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      if (p.par_on) {                                   // class row (img, h2, w2) -> output pixel (2 h2 + par_h, 2 w2 + par_w)
        ho = 2 * ho + p.par_h;
        wo = 2 * wo + p.par_w;
      }
      a_hb[j] = ho * p.stride - p.pad;
      a_wb[j] = wo * p.stride - p.pad;
      a_base[j] = ((unsigned)(img * p.H * p.W) * (unsigned)p.C + (unsigned)(pch * 8)) * 2u;
    } else {
      a_hb[j] = a_wb[j] = 0;
      a_base[j] = ((unsigned)(a_ok[j] ? m : 0) * (unsigned)p.lda + (unsigned)(pch * 8)) * 2u;
    }
  }
  unsigned b_base[BP];
  bool b_ok[BP];
#pragma unroll
  for (int j = 0; j < BP; ++j) {
    const int n = n0 + (j * NW + wave) * 8 + prow;
    b_ok[j] = n < p.N;
    b_base[j] = ((unsigned)(b_ok[j] ? n : 0) * (unsigned)p.K + (unsigned)(pch * 8)) * 2u;
  }
  // filter tap / channel chunk of the NEXT K-step to be issued (wave-uniform)
  int kti = kt0;
  int tap = kt0 / p.cpt, cc = kt0 - tap * p.cpt;
  int r = tap / p.S, s = tap - r * p.S;
  const int s0 = p.par_on ? 1 - p.par_w : 0, tstep = p.par_on ? 2 : 1;   // parity class: taps r = 1 - par_h (+2), s = 1 - par_w (+2) only
  if (p.par_on) {
    const int nS = 1 + p.par_w;
    r = (1 - p.par_h) + 2 * (tap / nS);
    s = s0 + 2 * (tap % nS);
  }
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);
  const int upm = p.up - 1, ups = p.up >> 1;     // up in {1,2}: parity mask / shift
  const unsigned rowB = (unsigned)(p.W * p.C * 2), pixB = (unsigned)(p.C * 2);     // < 2^24 (checked by the launcher): 24-bit multiplies
  // issue the slices of K-step kti into ring slot `slot`, then step (r, s, cc) / kti
  auto issue = [&](int slot) {
    unsigned char* sA = smem + slot * STAGE;
    unsigned char* sB = sA + A_BYTES;
    const bool live = kti < kt1;
    const unsigned kA = (unsigned)(CONV ? cc * 128 : kti * 128);
    const bool klive = live & (CONV | (kti * 64 + pch * 8 < p.K));
#pragma unroll
    for (int j = 0; j < AP; ++j) {
      bool ok = klive & a_ok[j];
      unsigned off = a_base[j] + kA;
      if (CONV) {
        int hp = a_hb[j] + r, wp = a_wb[j] + s;
        ok = ok & (((hp | wp) & upm) == 0);
        hp >>= ups;
        wp >>= ups;
        ok = ok & ((unsigned)hp < (unsigned)p.H) & ((unsigned)wp < (unsigned)p.W);
        off += __umul24((unsigned)hp, rowB) + __umul24((unsigned)wp, pixB);
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(sA + (j * NW + wave) * 1024), 16, (int)(ok ? off : OOB), 0, 0, 0);
    }
    const unsigned kB = (unsigned)(CONV ? ((r * p.S + s) * p.cpt + cc) * 128 : kti * 128);   // == kti * 128 unless taps are skipped
#pragma unroll
    for (int j = 0; j < BP; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(sB + (j * NW + wave) * 1024), 16, (int)((klive & b_ok[j]) ? b_base[j] + kB : OOB), 0, 0, 0);
    ++kti;
    cc += 1;
    const bool wc = cc == p.cpt;
    cc = wc ? 0 : cc;
    s += wc ? tstep : 0;
    const bool ws = s >= p.S;
    s = ws ? s0 : s;
    r += ws ? tstep : 0;
  };

  // ---- fragment addresses inside a stage (k-half 1 flips chunk bit 2: XOR 64)
  const int l15 = lane & 15, lg = lane >> 4;
  int a_addr[TM], b_addr[TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
    const int row = wm * (BM / WM) + mi * 16 + l15;
    a_addr[mi] = row * 128 + ((lg ^ (row & 7)) << 4);
  }
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
    const int row = wn * (BN / WN) + ni * 16 + l15;
    b_addr[ni] = A_BYTES + row * 128 + ((lg ^ (row & 7)) << 4);
  }
  f32x4_t acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t f0a[TM], f0b[TN], f1a[TM], f1b[TN];
  auto read_frags = [&](bf16x8_t (&fa)[TM], bf16x8_t (&fb)[TN], const unsigned char* st, int ks) {
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) fb[ni] = *reinterpret_cast<const bf16x8_t*>(st + (b_addr[ni] ^ (ks * 64)));
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) fa[mi] = *reinterpret_cast<const bf16x8_t*>(st + (a_addr[mi] ^ (ks * 64)));
  };
  auto mfma_all = [&](const bf16x8_t (&fa)[TM], const bf16x8_t (&fb)[TN]) {
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) acc[ni][mi] = MFMA16(fb[ni], fa[mi], acc[ni][mi]);
  };

  // ---- prologue: K-steps kt0 .. kt0 + NS - 2 in flight; wait for the first
#pragma unroll
  for (int i = 0; i < NS - 1; ++i) issue(i);
  wait_vmcnt<(NS - 2) * PER>();
  __builtin_amdgcn_s_barrier();
  read_frags(f0a, f0b, smem, 0);
  int slot = 0;                                          // ring slot of the current K-step
  // Per K-step (64 = two MFMA k-halves): [MFMAs of half 0 | reads of half 1] [vmcnt + barrier: the next K-step's slices have landed and
  // everybody is done with the previous K-step's slot] [MFMAs of half 1 | reads of the next K-step's half 0 | DMA of K-step + NS - 1
  // into the slot just freed].
  for (int kt = kt0; kt < kt1; ++kt) {
    const unsigned char* cur = smem + slot * STAGE;
    const int nslot = slot + 1 == NS ? 0 : slot + 1, fslot = slot == 0 ? NS - 1 : slot - 1;
    read_frags(f1a, f1b, cur, 1);
    mfma_all(f0a, f0b);
#pragma unroll
    for (int i = 0; i < NRD; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, (NM / NRD) > 0 ? (NM / NRD) : 1, 0);   // MFMAs
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                               // 1 DS read
    }
    __builtin_amdgcn_sched_barrier(0);
    wait_vmcnt<(NS - 3) * PER>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_frags(f0a, f0b, smem + nslot * STAGE, 0);
    issue(fslot);
    mfma_all(f1a, f1b);
    constexpr int NE = NRD > PER ? NRD : PER, MP2 = (NM / NE) > 0 ? (NM / NE) : 1;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, MP2, 0);
      if (i < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (i < PER) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                  // 1 VMEM (LDS-DMA piece)
    }
    __builtin_amdgcn_sched_barrier(0);
    slot = nslot;
  }
  wait_vmcnt<0>();                                       // the (out-of-range) pieces issued past kt1 write zeros into the ring
  __syncthreads();
  nt_epilogue<BM, BN, WM, WN, NT>(p, acc, smem, bm, m0, n0, split, wm, wn, tid, lane);
}

template <int BM, int BN, int WM, int WN, int NS, bool CONV>
int launch_impl2(GemmNT p, int splits, int slot, hipStream_t st) {
  const int nbm = ceil_div(p.M, BM);
  p.nbn = ceil_div(p.N, BN);
  p.ksteps_per_split = ceil_div(p.ksteps_total, splits);
  FEDFR_REQUIRE(ceil_div(p.ksteps_total, p.ksteps_per_split) == splits, "gemm_nt_glds: splits=%d leaves an empty split (ksteps=%d)", splits,
                p.ksteps_total);
  constexpr size_t kRing = (size_t)NS * (BM + BN) * 128, kEpi = (size_t)BM * (BN * 2 + 16);
  constexpr size_t lds = kRing > kEpi ? kRing : kEpi;
  static PerDeviceOnce attr_once;     // hipFuncSetAttribute is per device (a Server process may drive several)
  attr_once.run([&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_glds_kernel<BM, BN, WM, WN, NS, CONV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  const dim3 grid(nbm * p.nbn, splits, p.par_on == 2 ? 4 : 1);
  ProfScope prof(slot, 2.0 * p.M * p.N * (p.par_on == 2 ? 64.0 * 9 * p.cpt : p.par_on ? 64.0 * p.ksteps_total : (double)p.K), st, gemm_nt_alg_bytes(p, splits));
  hipLaunchKernelGGL((gemm_nt_glds_kernel<BM, BN, WM, WN, NS, CONV>), grid, dim3(64 * WM * WN), lds, st, p);
  FEDFR_LAUNCH_CHECK("gemm_nt_glds");
  return FEDFR_OK;
}
template <int BM, int BN, int WM, int WN, int NS>
int launch_impl(const GemmNT& p, int splits, int slot, hipStream_t st) {
  if (p.mode == 1) {
    FEDFR_REQUIRE((long long)p.W * p.C * 2 < (1 << 24) && p.H < (1 << 15), "gemm_nt_glds: image row too large for the 24-bit address multiplies");
    return launch_impl2<BM, BN, WM, WN, NS, true>(p, splits, slot, st);
  }
  return launch_impl2<BM, BN, WM, WN, NS, false>(p, splits, slot, st);
}
}  // namespace

int g_nt_glds = 4;   // option "nt_glds": 0 register-staged gemm_nt_kernel; 1 / 2 this kernel with 4 / 8 waves (two per SIMD) per 128-row tile and
                     // the 64-row shapes on 64-row tiles; 3 / 4 the same with the 64-row shapes on 128-row tiles where the layouts agree

// p: as prepared by gemm_nt_launch_one (cpt, ksteps_total, a_bytes / b_bytes set); BM / WM = the register-staged kernel's choice for the
// shape (the BatchNorm partial-row layout, gemm_nt_stat_rows, depends on them).  Measured (profiles/r02_ab_nt_glds_v1.txt): the ring pays
// from about 16 K-steps per workgroup (7x7 3x3 convs 71 -> 44.5 us, stride-2 3x3 forward 55-65 -> 46-52 us); short K loops (1x1 convs,
// the parity classes of a stride-2 dgrad) are better off with two register-staged workgroups per CU than with one 128 KB ring.
bool gemm_nt_glds_applies(const GemmNT& p, int BM, int splits) {
  const bool pays = (g_nt_glds & 8) || (!p.par_on && ceil_div(p.ksteps_total, splits) >= 16);     // + 8: every shape the kernel can serve (tests)
  return (g_nt_glds & 7) > 0 && p.N > 64 && (BM == 128 || BM == 64) && pays && !p.bpart && !p.esc && !p.eadd && !p.Cb2;
}

int launch_nt_glds(const GemmNT& p, int BM, int splits, int slot, hipStream_t st) {
  FEDFR_REQUIRE(gemm_nt_glds_applies(p, BM, splits), "gemm_nt_glds: unsupported problem");
  // a 64-row shape on 128-row tiles (option value 3 / 4): WM = 2 wave rows of 64 pixels leave the BatchNorm partial rows where the
  // 64-row tiling puts them, as long as both tilings agree on the row count
  const int v = g_nt_glds & 7;
  const bool up = v >= 3 && BM == 64 && (!p.stats || ceil_div(p.M, 64) == 2 * ceil_div(p.M, 128));
  if (BM == 128 || up) return (v & 1) == 0 ? launch_impl<128, 128, 2, 4, NT_GLDS_NS>(p, splits, slot, st) : launch_impl<128, 128, 2, 2, 4>(p, splits, slot, st);
  return launch_impl<64, 128, 1, 4, 3>(p, splits, slot, st);       // 72 KB: two workgroups per CU
}
