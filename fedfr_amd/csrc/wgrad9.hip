// 3x3 / stride-1 weight-gradient kernel that loads every operand byte ONCE for all nine filter taps -- alone in its translation
// unit (gemm_dev.h).
//
//   dW[co][ty][tx][ci] = sum over (img, y, x) of  dy[img][y][x][co] * xin[img][y + ty - 1][x + tx - 1][ci]
//
// gemm_tn_glds_kernel computes this as nine independent GEMM column tiles, each re-staging the (tap-shifted) input tile: 64 FLOP per
// staged byte, and the ablations (tools notes in DESIGN.md) show it bound by LDS-DMA issue + fragment reads, not by MFMA (an MFMA-only
// build runs in 22 us, the full kernel in 41).  Here a workgroup owns a 32 (co) x 64 (ci) block of ALL nine taps and walks over whole
// images (14x14) or 7-row bands (28x28):
//   * the reduction index is the position q = y * PW + x in a raster of pitch PW = 16 / 32 (the pad columns x >= W hold zeros on the dy
//     side, so whatever they meet on the input side contributes nothing); a band is K = 7 x 32 = 14 x 16 = 224 positions = 7 MFMA steps;
//   * the input band is staged once, with its halo, in the same raster: the operand of tap (ty, tx) is the SAME LDS tile read at the
//     constant row shift ty * PW + tx -- 47-51 KB staged per 14.5 MFLOP = 150 FLOP per staged byte;
//   * K runs in 16-row groups: an MFMA step takes group 2k as its low and group 2k+1 as its high k-half (the same permutation on both
//     operands), so the fragment of vertical tap ty at step k is groups (2k + ty PW/16, +1) -- a group is read from LDS once per
//     horizontal tap and serves up to three steps from registers: 8 ds_read_b64_tr_b16 per 9 MFMAs instead of 24 per 16;
//   * two stages (band s computing, band s+1 landing by LDS-DMA), ONE s_barrier per band.
// Accumulators: 9 taps x one 16x16 block per wave (8 waves: 4 ci-blocks x 2 co-blocks) = 36 VGPRs; the kernel is held to 128 VGPRs so
// that the BatchNorm-backward kernels of the main stream still fit beside it on every SIMD (a 64 x 64 block with 72 accumulator
// registers took 211: no other wave fitted, and its 16 K-splits wrote 37.7 MB of fp32 slabs per launch -- 12 us of stores; 32 x 64
// halves both the slab count and the bytes).
#include "gemm_tn_dev.h"
#ifndef W9_READS_FIRST
#define W9_READS_FIRST 0
#endif
#ifndef W9_ABLATE
#define W9_ABLATE 0     // timing experiments only: 1 no in-loop DMA, 2 no in-loop fragment reads, 4 no MFMA, 8 no band barrier
#endif

int g_wgrad9 = 1;   // option "wgrad9": this kernel for 3x3 / stride-1 layers on 14 / 28 / 56 / 112-wide maps, Cin % 64 == 0, Cout % 32 == 0

namespace {
template <int N_>
__device__ __forceinline__ void w9_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}
__device__ __forceinline__ int w9_swz(int row) { return (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2); }   // == tn_swz<128>

struct W9 {
  const bf16_t* dy;     // [B][W][W][cout]
  const bf16_t* x;      // [B][W][W][cin]
  float* out;           // [splits][cout][9 cin]
  int cout, cin, nci;   // nci = cin / 64
  int ntiles;           // (cout / 32) * nci
  int nstages, per_split;
  int W, lg;            // image width = 14 << lg; a stage = one 14 x 14 sub-image, (1 << lg)^2 of them per image
  unsigned dy_bytes, x_bytes;
};

template <int W_, int R_, int LOG_PW>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(3, 3))) void wgrad9_kernel(W9 p) {
  constexpr int PW = 1 << LOG_PW, KP = R_ * PW, KS = KP / 32;
  constexpr int VG = PW / 16, WIN = 2 + 2 * VG, NG = 2 * KS + 2 * VG;     // groups per vertical tap step / live window / groups per band
  constexpr int BANDS = W_ / R_;
  constexpr int P_PIECES = KP / 16;                                       // dy tile: 64-B rows (32 co), 16 rows per 1-KiB piece
  constexpr int Q_ROWS = (NG * 16 + 2 + 7) / 8 * 8, Q_PIECES = Q_ROWS / 8;
  constexpr int NP = P_PIECES + Q_PIECES, NPW = (NP + 7) / 8;
  constexpr int P_B = KP * 64, Q_B = Q_ROWS * 128, STAGE_B = P_B + Q_B;
  static_assert(KP % 32 == 0 && W_ % R_ == 0 && W_ <= PW && (R_ + 2) * PW + 2 <= Q_ROWS, "band geometry");
  static_assert(P_B % 1024 == 0 && (NG - 1) * 2048 < 65536, "ds_read immediate offsets");
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef __attribute__((address_space(3))) unsigned char* lds_uc_t;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = wave >> 1, ch = wave & 1;                 // ci block of 16, co block of 16
  const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int split = lid / p.ntiles, tile = lid - split * p.ntiles;
  const int tco = tile / p.nci, tci = tile - tco * p.nci;
  const int co0 = tco * 32, ci0 = tci * 64;
  const int s0 = split * p.per_split;
  const int nst = min(p.per_split, p.nstages - s0);

  // ---- LDS-DMA.  A piece = 1 KiB = one wave-instruction: 8 input-tile rows of 128 B (lane l -> row l >> 3, physical 16-B chunk
  // l & 7, which holds logical chunk (l & 7) ^ swz(row)) or 16 dy-tile rows of 64 B.  Pieces 0..P_PIECES-1 = dy tile, the rest = input
  // tile; wave w takes pieces w, w + 8, ...
  // (a wave past the end repeats the last piece: same bytes to the same place), so every wave issues NPW per stage.
  const __amdgpu_buffer_rsrc_t rsP = make_rsrc(p.dy, p.dy_bytes), rsQ = make_rsrc(p.x, p.x_bytes);
  constexpr unsigned OOB = 0xfffffff0u;
  const int prow = lane >> 3;
  auto issue_stage = [&](int s, int buf) {                  // s < 0: zero fill
    // stage -> (image, sub-image row, sub-image column); W_ x R_ is the sub-image (whole image when lg = 0, BANDS row bands else)
    const int per_img = BANDS << (2 * p.lg);
    const int img = s / per_img, rem = s - img * per_img;
    const int sub = rem / BANDS, band = rem - sub * BANDS;
    const int y0 = (sub >> p.lg) * W_ + band * R_, x0 = (sub & ((1 << p.lg) - 1)) * W_;
    const int Wi = p.W;
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const int piece = min(j * 8 + wave, NP - 1);          // wave-uniform
      if (piece < P_PIECES) {                               // dy: lane l -> row l >> 2 of the piece, 16-B chunk l & 3 (64-B rows need no swizzle)
        const int r = piece * 16 + (lane >> 2);
        const int yl = r >> LOG_PW, xx = r & (PW - 1);
        const bool ok = s >= 0 && xx < W_;
        const unsigned off = ((unsigned)((img * Wi + y0 + yl) * Wi + x0 + xx) * (unsigned)p.cout + (unsigned)(co0 + (lane & 3) * 8)) * 2u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lds_ptr_t)(smem + buf * STAGE_B + piece * 1024), 16, (int)(ok ? off : OOB), 0, 0, 0);
      } else {                                              // input with halo: the neighbouring sub-images' pixels, zeros outside the image
        const int qp = piece - P_PIECES;
        const int r = qp * 8 + prow;
        const int ql = r >> LOG_PW, xx = r & (PW - 1);
        const int lc = (lane & 7) ^ w9_swz(r);
        const int y = y0 + ql - 1, xc = x0 + xx - 1;
        const bool ok = s >= 0 && ql < R_ + 2 && xx < W_ + 2 && (unsigned)y < (unsigned)Wi && (unsigned)xc < (unsigned)Wi;
        const unsigned off = ((unsigned)((img * Wi + y) * Wi + xc) * (unsigned)p.cin + (unsigned)(ci0 + lc * 8)) * 2u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_ptr_t)(smem + buf * STAGE_B + P_B + qp * 1024), 16, (int)(ok ? off : OOB), 0, 0, 0);
      }
    }
  };

  // ---- fragment addresses.  One tr-read = 16 consecutive tile rows u = 4 g + q (lane group g, q = (lane & 15) >> 2) of a 16-channel
  // column block; lane (lane & 3) addresses 8 B at channel 4 (lane & 3).  Input-side reads start tx rows lower (three variants: the
  // swizzle key moves with the row), every other shift is a multiple of 16 rows = an immediate.
  const int g = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
  const unsigned lds0 = (unsigned)(size_t)(lds_uc_t)smem;
  auto frag_off = [&](int u, int colblk) {
    const int col = colblk + 4 * pp;
    return (unsigned)(u * 128 + (((col >> 3) ^ w9_swz(u)) << 4) + ((col >> 2) & 1) * 8);
  };
  unsigned offQ[3];
  const unsigned offP = lds0 + (unsigned)((4 * g + q4) * 64 + (ch * 16 + 4 * pp) * 2);     // dy tile: 64-B rows, 1-KiB groups
#pragma unroll
  for (int tx = 0; tx < 3; ++tx) offQ[tx] = lds0 + P_B + frag_off(tx + 4 * g + q4, cb * 16);

  f32x4_t acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // tr-reads are inline asm with our own lgkmcnt waits: hipcc drains the in-flight LDS-DMA with vmcnt(0) before the builtin form
  // (gemm_tn_glds.hip).  Results stay whole 64-bit tuples until they are consumed behind the wait.
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
#define W9_READ(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
  u32x2_t Pf[2 * KS], Qg[3][NG];
  auto frag = [](const u32x2_t& lo, const u32x2_t& hi) {
    const u32x4_t v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8_t, v);
  };

  // prologue: bands s0 and s0 + 1 in flight, first one landed, its first K-step's fragments fetched
  issue_stage(s0, 0);
  issue_stage(nst > 1 ? s0 + 1 : -1, 1);
  w9_wait_vmcnt<NPW>();
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 2; ++j) W9_READ(Pf[j], offP, j * 1024);
#pragma unroll
  for (int tx = 0; tx < 3; ++tx)
#pragma unroll
    for (int j = 0; j < WIN; ++j) W9_READ(Qg[tx][j], offQ[tx], j * 2048);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  for (int it = 0; it < nst; ++it) {
    const unsigned cur = (unsigned)((it & 1) * STAGE_B), nxt = (unsigned)(((it + 1) & 1) * STAGE_B);
    unsigned aQ[3], nQ[3];
    const unsigned aP = offP + cur, nP = offP + nxt;
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) { aQ[tx] = offQ[tx] + cur; nQ[tx] = offQ[tx] + nxt; }
#pragma unroll
    for (int kb = 0; kb < KS; ++kb) {
      constexpr int NRL = 2 + 3 * WIN;                       // reads issued in the last step of a band (next band's first fragments)
      const bool last = kb == KS - 1;
      if (last) {
        // band it+1 has landed (its DMA was issued a whole band ago); behind the barrier every wave has also finished reading band
        // it's buffer (its last fragments arrived at the end of the previous step), so band it+2 may overwrite it
        w9_wait_vmcnt<0>();
        if (!(W9_ABLATE & 8)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (!(W9_ABLATE & 1)) issue_stage(it + 2 < nst ? s0 + it + 2 : -1, it & 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      // read r of this step (compile-time after unrolling).  steady state: dy groups 2kb+2, 2kb+3, then the two new input groups of
      // every horizontal tap; last step: the next band's dy groups 0, 1 and input groups 0 .. WIN-1
      auto issue_read = [&](int r) {
        if (W9_ABLATE & 2) return;
        if (!last) {
          if (r < 2) W9_READ(Pf[2 * kb + 2 + r], aP, (2 * kb + 2 + r) * 1024);
          else { const int tx = (r - 2) >> 1, j = 2 * kb + WIN + ((r - 2) & 1); W9_READ(Qg[tx][j], aQ[tx], j * 2048); }
        } else {
          if (r < 2) W9_READ(Pf[r], nP, r * 1024);
          else { const int tx = (r - 2) / WIN, j = (r - 2) % WIN; W9_READ(Qg[tx][j], nQ[tx], j * 2048); }
        }
      };
      // W9_READS_FIRST: all of the step's reads in front of its nine MFMAs (144 cycles of cover for the last of them); otherwise one
      // read behind each MFMA, which leaves the wait at the end of the step exposed to the LDS latency of the last reads
      const int nr = last ? NRL : 8;
      const int r0 = W9_READS_FIRST ? nr : (last ? NRL - 9 : 0);
#pragma unroll
      for (int r = 0; r < NRL; ++r)
        if (r < r0) issue_read(r);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          const int m = ty * 3 + tx;
          if (!(W9_ABLATE & 4)) acc[m] = MFMA16(frag(Qg[tx][2 * kb + ty * VG], Qg[tx][2 * kb + ty * VG + 1]), frag(Pf[2 * kb], Pf[2 * kb + 1]), acc[m]);
          if (r0 + m < nr) issue_read(r0 + m);
          __builtin_amdgcn_sched_barrier(0);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef W9_READ
  w9_wait_vmcnt<0>();                                        // zero-fill DMAs of the tail retire before the LDS is released

  // D[m = ci][n = co]: lane holds ci = cb 16 + (lane >> 4) 4 + {0..3} for co = ch 16 + (lane & 15) -> one float4 per tap
  float* slab = p.out + (size_t)split * p.cout * 9 * p.cin;
  const int NJ = 9 * p.cin;
  const int co = co0 + ch * 16 + (lane & 15);
  const int ci = ci0 + cb * 16 + (lane >> 4) * 4;
#pragma unroll
  for (int t = 0; t < 9; ++t)
    if (!(W9_ABLATE & 16) || acc[t][0] == 12345.f)
      *reinterpret_cast<float4*>(slab + (size_t)co * NJ + t * p.cin + ci) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
}

template <int W_, int R_, int LOG_PW>
constexpr size_t w9_lds() {
  constexpr int PW = 1 << LOG_PW, KP = R_ * PW, KS = KP / 32, VG = PW / 16, NG = 2 * KS + 2 * VG;
  return 2 * (size_t)(KP * 64 + (NG * 16 + 2 + 7) / 8 * 8 * 128);
}
}  // namespace

extern int g_tn_use_tr;
// image widths served: 14 << lg, as 14 x 14 sub-images with real halos (a 28 x 28 map as four 7-row bands of pitch 32, which the
// kernel template also expresses, measured the same: 46.1 vs 46.4 us per 128 -> 128 layer)
static int w9_lg(int W) {
  for (int lg = 0; lg < 4; ++lg)
    if (W == (14 << lg)) return lg;
  return -1;
}
// (the shape alone: what gemm_tn_max_splits sizes workspaces with, whatever the switches say at that moment)
bool wgrad9_shape_ok(int Kp, int NI, int NJ, int C, int W, int stride) {
  return stride == 1 && w9_lg(W) >= 0 && C > 0 && C % 64 == 0 && NI % 32 == 0 && NJ == 9 * C && Kp % (W * W) == 0;
}
bool wgrad9_applies_shape(int Kp, int NI, int NJ, int C, int W, int stride) {
  return g_wgrad9 && g_tn_use_tr && wgrad9_shape_ok(Kp, NI, NJ, C, W, stride);
}
bool wgrad9_applies(const GemmTN& p) {
  return p.mode == 1 && p.use_tr && p.S == 3 && p.pad == 1 && p.H == p.W && p.Ho == p.H && p.Wo == p.W && p.ldp == p.NI &&
         wgrad9_applies_shape(p.Kp, p.NI, p.NJ, p.C, p.W, p.stride);
}
static int w9_stages(int Kp, int W) {
  const int images = Kp / (W * W);
  return images << (2 * w9_lg(W));
}
// one workgroup per CU: as many K-splits (whole stages) as it takes to put ~256 workgroups on the chip, at least two stages each
static const int g_wgrad9_wgs = 256;   // workgroups a launch aims for: one per CU (512 = half as long each, twice the slabs: lost in rounds 3-5, no switch)
int wgrad9_pick_splits(int Kp, int NI, int NJ, int W) {
  const int stages = w9_stages(Kp, W);
  const int tiles = (NI / 32) * (NJ / 9 / 64);
  int splits = g_wgrad9_wgs / tiles;
  if (splits < 1) splits = 1;
  if (splits > stages / 2) splits = stages / 2 > 0 ? stages / 2 : 1;
  const int per = ceil_div(stages, splits);
  return ceil_div(stages, per);
}

int launch_wgrad9(const GemmTN& g, int splits, hipStream_t st) {
  FEDFR_REQUIRE(wgrad9_applies(g), "wgrad9: unsupported problem");
  W9 p{};
  p.dy = g.P; p.x = g.Q; p.out = g.out; p.cout = g.NI; p.cin = g.C; p.nci = g.C / 64;
  p.ntiles = (g.NI / 32) * p.nci;
  p.W = g.W; p.lg = w9_lg(g.W);
  p.nstages = w9_stages(g.Kp, g.W);
  p.per_split = ceil_div(p.nstages, splits);
  FEDFR_REQUIRE(splits >= 1 && ceil_div(p.nstages, p.per_split) == splits, "wgrad9: splits=%d leaves an empty split", splits);
  p.dy_bytes = g.p_bytes; p.x_bytes = g.q_bytes;
  const dim3 grid(p.ntiles * splits);
  ProfScope prof(16, 2.0 * g.NI * g.NJ * (double)g.Kp, st, gemm_tn_alg_bytes(g, 1));      // (the split-K slabs are overhead, not algorithmic)
  constexpr size_t lds14 = w9_lds<14, 14, 4>();
  static PerDeviceOnce attr_once;     // hipFuncSetAttribute is per device (a Server process may drive several)
  attr_once.run([&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad9_kernel<14, 14, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds14);
  });
  hipLaunchKernelGGL((wgrad9_kernel<14, 14, 4>), grid, dim3(512), lds14, st, p);
  FEDFR_LAUNCH_CHECK("wgrad9");
  return FEDFR_OK;
}
