// Weight-gradient GEMM with LDS-DMA operand staging -- alone in its translation unit (gemm_dev.h).
//
//   C[i][j] = sum_p P[p][i] * Q(p, j)      P = dy [pixels][Cout], Q(p, (tap, ci)) = x[pixel p shifted by the tap][ci]
//
// Same math and LDS image as gemm_tn_kernel<128,128> (reduction-major operands, 256-B tile rows, chunk swizzle tn_swz<256>,
// fragments by ds_read_b64_tr_b16), different pipeline: in-kernel stamps on the register-staged kernels showed hipcc sinking
// the next K-step's global loads to just before their ds_write, so every K-step paid an L2 round trip.  Here both operand
// tiles of K-step kt+3 travel by `buffer_load_dwordx4 ... lds` into a 4-stage ring while kt computes; they are retired by a
// COUNTED s_waitcnt vmcnt(8) and ONE raw s_barrier per K-step, placed between the two MFMA k-halves so that the next stage's
// first fragments are read behind it; sched_group_barrier threads the tr-reads and the DMA issue between the MFMAs.
// An LDS-DMA wave-instruction writes 1 KiB linearly (lane l -> base + 16 l) = 4 tile rows of 256 B, so the chunk swizzle is
// applied to the per-lane SOURCE address; filter-tap shifts and image borders of Q are per-lane source offsets too, with
// out-of-image rows fetched from an out-of-range buffer offset (the hardware writes zeros, tools/probe/glds_probe.hip).
#include "gemm_tn_dev.h"

namespace {
template <int N_>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

constexpr int TILE_B = 64 * 256;            // one operand tile: 64 pixel rows x 128 channels bf16
constexpr int STAGE_B = 2 * TILE_B;

// Ring: NS stages, the tiles of K-step it + D are issued during the second half of K-step it.  NS = 4, D = 3 (128 KiB, one block per
// CU) writes the slot K-step it - 1 released; NS = 2, D = 2 (64 KiB, TWO blocks per CU) writes the slot of K-step it itself, whose
// fragments are all in registers once the mid-step barrier has been passed.
template <int WI, int WJ, int NS>   // wave grid: 2 x 2 (one wave per SIMD, 64 x 64 wave tiles) or 2 x 4 (two per SIMD, 64 x 32)
__device__ __forceinline__ void tn_glds_body(const GemmTN& p, const int bid, const int nblk) {
  constexpr int D = NS == 2 ? 2 : NS - 1;
  constexpr int TI = 128, TJ = 128, RB = 256, NW = WI * WJ;
  constexpr int FI = TI / WI / 16, FJ = TJ / WJ / 16, NF = FI + FJ, NM = FI * FJ;   // fragments / MFMAs per k-half
  constexpr int RSTEP = 4 * NW, NROW = 64 / RSTEP;          // tile rows between a lane's rows; row steps per stage (4 or 2)
  constexpr int DPS = 2 * NROW;                             // LDS-DMA instructions per stage and wave
  static_assert(NM >= NF + NROW, "not enough MFMA slots for reads + DMA");
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave / WJ, wj = wave % WJ;
  // 1-D grid over (split, tile), split-major, XCD-remapped: an XCD owns whole K-splits (see gemm_tn_kernel)
  const int lid = xcd_remap(bid, nblk);
  const int split = lid / p.ntiles, tile = lid - split * p.ntiles;
  const int bj = tile % p.nbj, bi = tile / p.nbj;
  const int i0 = bi * TI, j0 = bj * TJ;
  const int kt0 = split * p.ksteps_per_split;
  const int kt1 = min(kt0 + p.ksteps_per_split, p.ksteps_total);
  const int nsteps = kt1 - kt0;

  // ---- DMA source plan.  A piece = 4 tile rows; this wave owns pieces j*NW + wave, i.e. tile rows RSTEP j + 4 wave + r.
  // Lane l: row r = l >> 4 of the piece, physical chunk l & 15; tn_swz<256>(row) = ((row & 3) << 1) | (bit 3 of row) << 3,
  // and bit 3 of (RSTEP j + 4 wave + r) is bit 1 of wave (RSTEP is 16 or 32).
  const int r = lane >> 4;
  const int lc = (lane & 15) ^ ((r << 1) | (((wave >> 1) & 1) << 3));       // logical 16-B chunk (8 channels)
  constexpr unsigned OOB = 0xfffffff0u;
  const __amdgpu_buffer_rsrc_t rsP = make_rsrc(p.P, p.p_bytes), rsQ = make_rsrc(p.Q, p.q_bytes);
  const int row0 = 4 * wave + r;                                               // first tile row of this lane
  unsigned p_off = ((unsigned)(kt0 * 64 + row0) * (unsigned)p.ldp + (unsigned)(i0 + lc * 8)) * 2u;
  const unsigned p_j = (unsigned)RSTEP * (unsigned)p.ldp * 2u;
  // filter tap of this tile (TJ = 128 <= C: a tile never straddles taps).  Q gather state of the lane's current row: output
  // pixel (ho, wo) and the byte offset of its tap-shifted input pixel, all advanced RSTEP pixels at a time with adds and selects
  // only (the register-staged kernel's per-step fastdiv / 32-bit multiplies were ~1000 VALU cycles per K-step here, where
  // one wave per SIMD has nothing to hide them behind).
  const int tap = j0 / p.C, cj0 = j0 - tap * p.C;
  const int tr_ = tap / p.S, ts_ = tap - tr_ * p.S;
  const int sshift = p.stride == 2 ? 1 : 0;
  const int hp0 = tr_ - p.pad, wp0 = ts_ - p.pad;                             // input row / col = (ho << sshift) + hp0, ...
  const unsigned PIX_B = (unsigned)(p.stride * p.C * 2), ROW_B = (unsigned)(p.stride * p.W * p.C * 2), IMG_B = (unsigned)(p.H * p.W * p.C * 2);
  // RSTEP pixels = dIMG images + d16H rows + d16W columns (d16H < Ho, d16W < Wo: one conditional wrap per level suffices)
  const int d16W = RSTEP % p.Wo, d16H = (RSTEP / p.Wo) % p.Ho, dIMG = (RSTEP / p.Wo) / p.Ho;
  const unsigned STEP16 = (unsigned)d16W * PIX_B + (unsigned)d16H * ROW_B + (unsigned)dIMG * IMG_B;
  const unsigned WRAPW = ROW_B - (unsigned)p.Wo * PIX_B, WRAPH = IMG_B - (unsigned)p.Ho * ROW_B;
  int q_ho, q_wo;
  unsigned q_off;
  {
    const unsigned m = (unsigned)(kt0 * 64 + row0);
    const unsigned img = fdiv(m, p.dHoWo), rem = m - img * p.dHoWo.d;
    const unsigned ho = fdiv(rem, p.dWo);
    q_ho = (int)ho; q_wo = (int)(rem - ho * p.dWo.d);
    q_off = img * IMG_B + ho * ROW_B + (unsigned)q_wo * PIX_B + (unsigned)((hp0 * p.W + wp0) * p.C * 2) + (unsigned)((cj0 + lc * 8) * 2);
  }
  int m_row = kt0 * 64 + row0;                                                 // pixel index of the lane's current row

  // one row step: DMA of this lane's chunk of tile row RSTEP j + 4 wave + r of both operands, then advance the state by RSTEP
  // pixels; after the last call of a stage the state sits 64 pixels further = the same row of the next K-step.
  auto issue_row = [&](int slot, int j, bool live) {
    unsigned char* sP = smem + slot * STAGE_B;
    const bool rok = live && m_row < p.Kp;
    const unsigned vp = rok ? p_off : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lds_ptr_t)(sP + (j * NW + wave) * 1024), 16, (int)vp, 0, 0, 0);
    const int hp = (q_ho << sshift) + hp0, wp = (q_wo << sshift) + wp0;
    const bool ok = rok && (unsigned)hp < (unsigned)p.H && (unsigned)wp < (unsigned)p.W;
    const unsigned vq = ok ? q_off : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_ptr_t)(sP + TILE_B + (j * NW + wave) * 1024), 16, (int)vq, 0, 0, 0);
    q_wo += d16W;
    const bool cw = q_wo >= p.Wo;
    q_wo -= cw ? p.Wo : 0;
    q_ho += d16H + (cw ? 1 : 0);
    const bool chh = q_ho >= p.Ho;
    q_ho -= chh ? p.Ho : 0;
    q_off += STEP16 + (cw ? WRAPW : 0u) + (chh ? WRAPH : 0u);
    p_off += p_j;
    m_row += RSTEP;
  };
  auto issue_stage = [&](int slot, bool live) {
#pragma unroll
    for (int j = 0; j < NROW; ++j) issue_row(slot, j, live);
  };

  int fop[FI], foq[FJ];
#pragma unroll
  for (int ti = 0; ti < FI; ++ti) fop[ti] = tn_frag_off<RB>(wi * (TI / WI) + ti * 16, lane);
#pragma unroll
  for (int tj = 0; tj < FJ; ++tj) foq[tj] = tn_frag_off<RB>(wj * (TJ / WJ) + tj * 16, lane) + TILE_B;

  f32x4_t acc[FJ][FI];
#pragma unroll
  for (int a = 0; a < FJ; ++a)
#pragma unroll
    for (int b = 0; b < FI; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // Fragment tr-reads are inline asm: hipcc treats the ds_read_tr builtin as aliasing the in-flight LDS-DMA and drains it with
  // vmcnt(0) before every read.  The asm reads are invisible to its waitcnt pass, so the two s_waitcnt lgkmcnt(0) below are
  // ours; each is followed by sched_barrier(0) so that no MFMA consuming the fragments is hoisted above it (guide rule 18).
  typedef __attribute__((address_space(3))) unsigned char* lds_uc_t;
  const unsigned lds0 = (unsigned)(size_t)(lds_uc_t)smem;
#define TR_READ(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
  // 64-bit read results are kept as 2 x u32 and only re-typed (pure register tuples, no VALU) where they are consumed, i.e.
  // behind the lgkmcnt wait: element-wise 16-bit handling made hipcc emit v_bfi on the destination right after the asm,
  // racing the LDS return.
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
  u32x2_t f0[2 * NF], f1[2 * NF];                          // (lo, hi) pairs: fragments 0..FJ-1 = Q (per tj), FJ..NF-1 = P (per ti)
  unsigned fo[NF];
#pragma unroll
  for (int t = 0; t < FJ; ++t) fo[t] = lds0 + (unsigned)foq[t];
#pragma unroll
  for (int t = 0; t < FI; ++t) fo[FJ + t] = lds0 + (unsigned)fop[t];
  auto frag = [](const u32x2_t& lo, const u32x2_t& hi) {
    const u32x4_t v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8_t, v);
  };
  // one k-half: NM MFMAs on `cur`; the 2 NF tr-reads of the other fragment set go two per MFMA in front of the first NF,
  // the stage's NROW DMA row steps behind the last NROW
#define HALF(cur, oth, stage_off, KS, DMA_SLOT, DMA_LIVE)                                                   \
  _Pragma("unroll") for (int tj = 0; tj < FJ; ++tj) {                                                       \
    _Pragma("unroll") for (int ti = 0; ti < FI; ++ti) {                                                     \
      const int m = tj * FI + ti;                                                                           \
      if (m < NF) {                                                                     \
        const unsigned a = fo[m] + (stage_off);                                                             \
        TR_READ(oth[2 * m], a, (KS) * 32 * RB);                                                             \
        TR_READ(oth[2 * m + 1], a, (KS) * 32 * RB + 4 * RB);                                                \
      }                                                                                                     \
      acc[tj][ti] = MFMA16(frag(cur[2 * tj], cur[2 * tj + 1]), frag(cur[2 * (FJ + ti)], cur[2 * (FJ + ti) + 1]), acc[tj][ti]); \
      if ((DMA_SLOT) >= 0 && m >= NM - NROW) issue_row((DMA_SLOT), m - (NM - NROW), (DMA_LIVE));            \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
    }                                                                                                       \
  }

  // prologue: D stages in flight, wait for the first, fetch its k-half-0 fragments
#pragma unroll
  for (int s = 0; s < D; ++s) issue_stage(s, nsteps > s);
  wait_vmcnt<(D - 1) * DPS>();
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int m = 0; m < NF; ++m) {
    TR_READ(f0[2 * m], fo[m], 0);
    TR_READ(f0[2 * m + 1], fo[m], 4 * RB);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  for (int it = 0; it < nsteps; ++it) {
    const unsigned cur_off = (unsigned)((it & (NS - 1)) * STAGE_B), nxt_off = (unsigned)(((it + 1) & (NS - 1)) * STAGE_B);
    static_assert((NS & (NS - 1)) == 0, "ring size must be a power of two");
    // ---- first half: k-half 0 MFMAs; reads of this stage's k-half 1
    HALF(f0, f1, cur_off, 1, -1, false)
    wait_vmcnt<(D - 2) * DPS>();                                   // stage it+1 landed (it+2 .. it+D-1 may be in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // f1 arrived
    __builtin_amdgcn_s_barrier();                                  // ... for every wave; everyone is done reading stage it-1
    __builtin_amdgcn_sched_barrier(0);
    // ---- second half: k-half 1 MFMAs; reads of stage it+1's k-half 0 in front of the first eight, the DMA of stage it+D one row
    // step behind every other one of the last eight
    HALF(f1, f0, nxt_off, 0, (it + D) & (NS - 1), it + D < nsteps)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
#undef HALF
#undef TR_READ
  wait_vmcnt<0>();                                          // tail (zero-writing) DMAs retire before the block's LDS is released

  // D[row = j][col = i]: j = j0 + wj*64 + tj*16 + (lane>>4)*4 + reg ; i = i0 + wi*64 + ti*16 + (lane&15)
  float* slab = p.out + (size_t)split * p.NI * p.NJ;
#pragma unroll
  for (int tj = 0; tj < FJ; ++tj)
#pragma unroll
    for (int ti = 0; ti < FI; ++ti) {
      const int i = i0 + wi * (TI / WI) + ti * 16 + (lane & 15);
      const int j = j0 + wj * (TJ / WJ) + tj * 16 + (lane >> 4) * 4;
      *reinterpret_cast<float4*>(slab + (size_t)i * p.NJ + j) = make_float4(acc[tj][ti][0], acc[tj][ti][1], acc[tj][ti][2], acc[tj][ti][3]);
    }
}

template <int WI, int WJ>
__global__ __launch_bounds__(64 * WI * WJ) void gemm_tn_glds_kernel(GemmTN p) {
  tn_glds_body<WI, WJ, 4>(p, (int)blockIdx.x, (int)gridDim.x);
}
}  // namespace

bool gemm_tn_glds_shape_ok(int NI, int NJ, int C, int mode) { return mode == 1 && NI % 128 == 0 && NJ % 128 == 0 && C % 128 == 0; }
bool gemm_tn_glds_applies(int NI, int NJ, int C, int mode) { return g_tn_glds > 0 && gemm_tn_glds_shape_ok(NI, NJ, C, mode); }

// one block per CU (128 KiB of LDS): choose the K-split count that minimises rounds x (K-steps per split + fixed overhead)
int gemm_tn_glds_pick_splits(int Kp, int NI, int NJ) {
  const int tiles = (NI / 128) * (NJ / 128), ksteps = ceil_div(Kp, 64);
  int best = 1;
  long long best_cost = -1;
  for (int s = 1; s <= 64 && s <= ksteps / 4 + 1; ++s) {
    const int per = ceil_div(ksteps, s);
    if (ceil_div(ksteps, per) != s) continue;                  // would leave an empty trailing split
    const long long cost = (long long)ceil_div((long long)tiles * s, 256) * (per + 8);
    if (best_cost < 0 || cost < best_cost) { best = s; best_cost = cost; }
  }
  return best;
}

int launch_tn_glds(GemmTN p, int splits, hipStream_t st) {
  FEDFR_REQUIRE(gemm_tn_glds_applies(p.NI, p.NJ, p.C, p.mode) && p.use_tr, "gemm_tn_glds: unsupported problem");
  p.nbj = p.NJ / 128;
  p.ntiles = (p.NI / 128) * p.nbj;
  p.ksteps_total = ceil_div(p.Kp, 64);
  p.ksteps_per_split = ceil_div(p.ksteps_total, splits);
  FEDFR_REQUIRE(ceil_div(p.ksteps_total, p.ksteps_per_split) == splits, "gemm_tn_glds: splits=%d leaves an empty split", splits);
  constexpr size_t lds = (size_t)4 * STAGE_B;
  static PerDeviceOnce attr_once;     // hipFuncSetAttribute is per device (a Server process may drive several)
  attr_once.run([&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_glds_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_glds_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  ProfScope prof(14, 2.0 * p.NI * p.NJ * (double)p.Kp, st, gemm_tn_alg_bytes(p, splits));
  if (g_tn_glds >= 2) hipLaunchKernelGGL((gemm_tn_glds_kernel<2, 4>), dim3(p.ntiles * splits), dim3(512), lds, st, p);
  else hipLaunchKernelGGL((gemm_tn_glds_kernel<2, 2>), dim3(p.ntiles * splits), dim3(256), lds, st, p);
  FEDFR_LAUNCH_CHECK("gemm_tn_glds");
  return FEDFR_OK;
}
