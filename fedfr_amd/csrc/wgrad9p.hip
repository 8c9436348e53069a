// The two same-shape 3x3 / stride-1 weight gradients of a residual block in ONE launch, 64 (co) x 64 (ci) x 9 taps per workgroup --
// alone in its translation unit (gemm_dev.h); built with hipcc's default MFMA form (accumulators in AGPRs), see the Makefile.
//
// wgrad9_kernel (wgrad9.hip) is LDS-read bound: its workgroup owns 32 x 64 x 9 taps, a wave a 16 x 16 block of every tap, i.e. 8
// ds_read_b64_tr_b16 per 9 MFMAs -- 229 KB of fragment reads + 47 KB of LDS-DMA writes per 14x14 sub-image against 2016 MFMA cycles per
// SIMD (128 B/clk: 2160).  A 64 x 64 block halves the reads per MFMA, but a single layer then has half as many output tiles and needs
// twice the K-splits to fill 256 CUs: 16 slabs of fp32 per launch = 37.7 MB of stores (measured in round 1: faster loop, slower kernel).
// A residual block has TWO such layers (conv1, conv2: same shape in every block but a stage's first): both in one grid keeps
// 8 K-splits per layer with 64 x 64 tiles --
//   * workgroup = (layer, K-split, 64 x 64 tile), 4 waves = one per SIMD, wave (cih, coh) owns the 32 (ci) x 32 (co) quadrant of all nine
//     taps: 36 accumulators (144 AGPRs); per 32-position K-step 36 MFMAs and 16 fragment reads (4 of dy, 12 of the input) -- 0.44 reads
//     per MFMA instead of 0.89, LDS traffic 57 % of the MFMA time;
//   * everything else as in wgrad9.hip: reduction index = position in a raster of pitch 16 over a 14 x 14 sub-image (pad columns zero on
//     the dy side), the input sub-image staged once with its halo, tap (ty, tx) = the same LDS tile at row shift 16 ty + tx, 16-row
//     groups shared by the three vertical taps from registers, two stages (one computing, the next landing by LDS-DMA), one barrier per
//     sub-image, inline-asm tr-reads with manual lgkmcnt waits;
//   * one wave per SIMD, so nothing hides a stall: reads run one K-step ahead, issued behind the first MFMAs of a step and waited for
//     at its end; the 16 LDS-DMA instructions of a stage are threaded between the MFMAs of the step that follows the barrier; their
//     source offsets are scalar arithmetic plus ONE per-lane term (see the plan below); MFMAs are hand-written with the accumulator as a
//     tied in/out AGPR operand, and every MFMA operand is an aligned pair of 16-row groups (see the K loop).
// 132 VGPRs + 144 AGPRs at one wave per SIMD (wgrad9: 2 x 144 of 512).  On by default since round 3 (g_wgrad9p below).
#include "gemm_tn_dev.h"
#ifndef W9P_PRIO
#define W9P_PRIO 0       // wave priority of the pair's waves (the BatchNorm-backward passes that share its CUs run at BNS_PRIO, bn_sliced.hip)
#endif
#ifndef W9P_SPREAD
#define W9P_SPREAD 3     // a stage's LDS-DMA pieces are issued over SIX K-steps, three per step (round 6; 1 / 2: over four steps, back to back / eight MFMAs apart);
                         // 0: all sixteen in the last K-step of the sub-image before (rounds 3-5).  Same-box: 15.14 -> 15.04 ms per step, 69.5 -> 66.6 us per pair
                         // (profiles/r06_ab_w9p_dma_spread_v1.txt)
#endif
#ifndef W9P_ABLATE
#define W9P_ABLATE 0     // timing experiments only: 1 no in-loop DMA, 2 no in-loop fragment reads, 4 no MFMA, 16 no slab stores, 32 no slab loads by the reduction job
#endif

// option "wgrad9p".  History (rounds 2-3; it was off at first): measured on one box, the pair runs in 66-69 us against 89 us for the two single-layer launches with
// their slab reductions (s3 256x256@14, B = 128), the single-stream step drops 18.88 -> 18.11 ms, and two clients sharing the GPU gain
// 2.7 % (8 275 -> 8 502 img/s) -- but ONE client's dual-stream step gets 0.1-0.3 ms SLOWER (17.6 -> 17.85): conv, dgrad and weight-gradient
// workgroups never share a CU (registers), so the streams interleave at workgroup granularity, and the main stream now waits for
// 65-90 us workgroups instead of 40-60 us ones (rocprof: conv3x3_glds 38.3 -> 30.4 us per launch, but the BatchNorm passes stretch
// and the overlap the second stream bought is gone).  Server.train switches it on when several clients train concurrently.
int g_wgrad9p = 1;   // option "wgrad9p".  On since round 3 TOGETHER with fuse_bnbwd = 2 (net.hip): alone either loses in the dual-stream step, the pair gains 0.2 ms (profiles/r03_ab_options_final_v1.txt)

namespace {
template <int N_>
__device__ __forceinline__ void w9p_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}
// LDS rows are 128 B (8 chunks of 16 B); a lane quad of a transposed read takes 32 B (a chunk PAIR) of one row, a 32-lane group (the unit the
// LDS serves a ds_read_b64_tr_b16 in: 2 x 32 lanes, 64 banks of 4 B) eight consecutive rows.  Rows of equal parity share their 32-bank half, so the
// four of them need four different chunk pairs: the pair index is XOR-ed with row bits 1-2.  (Rounds 2-5 used row bits 1 and 3 — tn_swz<128>, laid
// out for 16-lane service groups: rows u and u + 4 then met in the same banks, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50 exactly,
// profiles/r06_pmc_sq_step_v1.txt.)
#ifndef W9P_SWZ_OLD
#define W9P_SWZ_OLD 0
#endif
__device__ __forceinline__ int w9p_swz(int row) { return W9P_SWZ_OLD ? ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2)) : (((row >> 1) & 3) << 1); }

// The split-K slabs of the PREVIOUS paired launch on this stream, summed by this launch's workgroups beside their own work (round 4).  A
// stand-alone reduction is bandwidth-bound at 5-7 us per layer plus its launch boundary: 13 of the 66 us a pair cost.  Here a workgroup owns
// `per` consecutive float4 outputs (its thread t the outputs o * 256 + t) and works through them in UNITS of four slab loads: one unit is
// requested per 14 x 14 sub-image (in its second K-step, four single loads between MFMAs) and summed one sub-image later, again between
// MFMAs (the loop's own vmcnt(0) at the end of every sub-image has retired the loads by then): 24 registers, ~25 instructions per sub-image,
// all in the shadow of the MFMA stream.  (The first version ran 230 mostly scalar instructions per sub-image in the open, behind the
// barrier: +6 us per launch — one wave per SIMD pays every instruction it issues outside an MFMA's shadow.)
// Summation order = the stand-alone kernels' (ew.hip), so the two paths agree bit for bit (template parameter JM):
//   1 narrow (reduce_slabs_kernel):      a = s_0; a += s_1; ... ascending slabs                       unit j of an output = slabs 4j .. 4j+3
//   2 wide   (reduce_slabs_wide_kernel, 32 slabs): a_q = (s_q + s_(q+8)) + (s_(q+16) + s_(q+24)), result = ((a_0 + a_1) + a_2) + ...   unit q = a_q
// The narrow order starts an output's accumulator at -0.0f (x + -0.0 == x for every x, signed zeros included: "a = first term" without a select); the
// wide order starts at +0.0f, as reduce_slabs_wide_kernel's lanes do — so both agree with their stand-alone kernels down to the sign of a zero.
struct W9PJobDev {
  const float* slab[2];
  float* dst[2];
  unsigned n4, per;     // float4 per layer / per workgroup (n4 % per == 0: a workgroup's share lies in ONE layer)
  int nsplit;
  int upo, units;       // units per output; units per thread = outputs per thread * upo
  unsigned sb, sr;      // byte step between the slab groups of consecutive units / between the four loads of a unit
};

struct W9P {
  W9PJobDev job;
  const bf16_t* dy[2];  // [B][W][W][cout]
  const bf16_t* x[2];   // [B][W][W][cin]
  float* out[2];        // [splits][cout][9 cin]
  int cout, cin, nci;   // nci = cin / 64
  int ntiles;           // (cout / 64) * nci, per layer
  int nstages, per_split, splits;
  int W, lg;            // image width = 14 << lg; a stage = one 14 x 14 sub-image, (1 << lg)^2 of them per image
  unsigned dy_bytes, x_bytes;
};

constexpr int W_ = 14, PW = 16, KP = 14 * PW, KS = KP / 32;          // 224 positions = 7 K-steps per sub-image
constexpr int NG = 2 * KS + 2, WIN = 4;                             // 16-row groups per sub-image (with the two halo rows) / live per step
constexpr int P_ROWS = KP, Q_ROWS = (NG * 16 + 2 + 7) / 8 * 8;      // 224 / 264 rows of 128 B
constexpr int P_PIECES = P_ROWS / 8, Q_PIECES = Q_ROWS / 8, NP = P_PIECES + Q_PIECES, NPW = (NP + 3) / 4;
constexpr int ZG = 2048;                                            // one 16-row group of zeros in front of and behind the dy tile (see the K loop)
constexpr int P_B = ZG + P_ROWS * 128 + ZG, Q_B = Q_ROWS * 128, STAGE_B = P_B + Q_B;
static_assert(P_B % 1024 == 0 && STAGE_B % 1024 == 0 && (NG - 1) * 2048 < 65536 && 2 * STAGE_B <= 160 * 1024, "stage geometry");

template <int JM>   // 0: no slab-reduction job; 1 / 2: narrow / wide order (W9PJobDev)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wgrad9p_kernel(W9P p) {
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef __attribute__((address_space(3))) unsigned char* lds_uc_t;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

#if W9P_PRIO
  __builtin_amdgcn_s_setprio(W9P_PRIO);
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cih = wave >> 1, coh = wave & 1;               // 32-channel halves of the 64 x 64 block
  const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int ps = lid / p.ntiles, tile = lid - ps * p.ntiles;     // (layer, split) groups are contiguous: one XCD fetches their images once
  const int prob = ps / p.splits, split = ps - prob * p.splits;
  const int tco = tile / p.nci, tci = tile - tco * p.nci;
  const int co0 = tco * 64, ci0 = tci * 64;
  const int s0 = split * p.per_split;
  const int nst = min(p.per_split, p.nstages - s0);

  // ---- LDS-DMA plan.  A piece = 1 KiB = one wave-instruction = 8 tile rows of 128 B: lane l -> row l >> 3, physical 16-B chunk l & 7,
  // which holds logical chunk (l & 7) ^ swz(row).  Round j of wave w is piece 4 j + w: rounds 0..6 are the 28 pieces of the dy tile,
  // rounds 7..15 the input tile with its halo (33 pieces + filler).  Two pieces make a raster row of 16 positions, so
  //   * the piece's raster row 2 j + (w >> 1) and its half (w & 1) are scalars: the source offset of a piece is scalar arithmetic
  //     (stage base + j * row step + a per-wave constant) plus ONE per-lane term (column within the half row, swizzled chunk), the same
  //     for every round of a wave;
  //   * what must read as zero is marked IN the offset (operands are < 1 GiB, checked by the launcher): + 2^31 on the lanes of a column
  //     outside the image (dy pad columns 14, 15: always; input column 0 / 15: when the sub-image touches the left / right border),
  //     + 2^30 on a whole piece whose row is outside (top / bottom halo at the border, filler rows, or no sub-image left to fetch) --
  //     no branches, no per-piece lane masks: a piece costs a scalar add, a vector add and the load.
  const __amdgpu_buffer_rsrc_t rsP = make_rsrc(p.dy[prob], p.dy_bytes), rsQ = make_rsrc(p.x[prob], p.x_bytes);
  constexpr int MARK_LANE = (int)0x80000000u, MARK_ROW = 0x40000000;
  const int Wi = p.W;
  const int prow = lane >> 3, odd = wave & 1, whalf = wave >> 1;
  const int lchunk = ((lane & 7) ^ (W9P_SWZ_OLD ? ((((prow >> 1) & 1) << 1) | (odd << 2)) : (((prow >> 1) & 3) << 1))) * 16;     // swz(row): row bits 1-2 = prow bits 1-2 (a piece is 8 aligned rows)
  const int vlaneP = (prow * p.cout * 2 + lchunk) | ((odd && prow >= 6) ? MARK_LANE : 0);
  const int laneQ = prow * p.cin * 2 + lchunk;
  const bool q_edge_lane = odd ? prow == 7 : prow == 0;    // this wave's pieces hold raster column 15 (odd half) or 0 (even half) in that lane row
  const int rowstepP = 2 * Wi * p.cout * 2, rowstepQ = 2 * Wi * p.cin * 2;            // two raster rows (one round)
  const int waveP = (whalf * Wi + odd * 8) * p.cout * 2, waveQ = (whalf * Wi + odd * 8) * p.cin * 2;
  int st_baseP = 0, st_baseQ = 0, st_top = 0, st_bot = 0, vlaneQ = laneQ;    // of the stage being issued
  auto stage_setup = [&](int s, bool live) __attribute__((always_inline)) {
    const int per_img = 1 << (2 * p.lg);
    const int img = s >> (2 * p.lg), sub = s & (per_img - 1);
    const int y0 = (sub >> p.lg) * W_, x0 = (sub & ((1 << p.lg) - 1)) * W_;
    const int pix = (img * Wi + y0) * Wi + x0;
    st_baseP = live ? (pix * p.cout + co0) * 2 + waveP : MARK_ROW;
    st_baseQ = live ? ((pix - Wi - 1) * p.cin + ci0) * 2 + waveQ : MARK_ROW;        // raster position (0, 0) of the input tile = pixel (y0 - 1, x0 - 1)
    st_top = (live && y0 != 0) ? 0 : MARK_ROW;              // marker of the top / bottom halo row
    st_bot = (live && y0 + W_ != Wi) ? 0 : MARK_ROW;
    const bool edge = odd ? x0 + W_ == Wi : x0 == 0;        // scalar: this wave's border column lies outside the image
    vlaneQ = (edge && q_edge_lane) ? (laneQ | MARK_LANE) : laneQ;
  };
  auto issue_piece = [&](int j, int buf) __attribute__((always_inline)) {                  // j: compile-time
    if (j < P_PIECES / 4) {
      const int vo = st_baseP + j * rowstepP + vlaneP;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lds_ptr_t)(smem + buf * STAGE_B + ZG + (j * 4 + wave) * 1024), 16, vo, 0, 0, 0);
    } else {
      const int jq = j - P_PIECES / 4;                      // raster row 2 jq + whalf of the input tile: 0 = top halo, 15 = bottom halo, >= 16 filler
      int soff = st_baseQ + jq * rowstepQ;
      if (jq == 0) soff |= whalf ? 0 : st_top;              // (scalar selects)
      if (jq == 7) soff |= whalf ? st_bot : 0;
      if (jq >= 8) soff = MARK_ROW;
      const int vo = soff + vlaneQ;
      const int qpiece = min(jq * 4 + wave, Q_PIECES - 1);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_ptr_t)(smem + buf * STAGE_B + P_B + qpiece * 1024), 16, vo, 0, 0, 0);
    }
  };

  // ---- fragment addresses.  One tr-read = 16 consecutive tile rows u = 4 g + q (lane group g, q = (lane & 15) >> 2) of a 16-channel
  // column block; lane (lane & 3) addresses 8 B at channel 4 (lane & 3).  Input-side reads start tx rows lower (the swizzle key moves
  // with the row: one address per (tx, block)), every other shift is a multiple of 16 rows = an immediate.
  const int g = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
  const unsigned lds0 = (unsigned)(size_t)(lds_uc_t)smem;
  auto frag_off = [&](int u, int colblk) __attribute__((always_inline)) {
    const int col = colblk + 4 * pp;
    return (unsigned)(u * 128 + (((col >> 3) ^ w9p_swz(u)) << 4) + ((col >> 2) & 1) * 8);
  };
  unsigned aP[2], nP[2], aQ[3][2], nQ[3][2];               // fragment addresses in the buffer being computed on / the other one
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    aP[b] = lds0 + frag_off(4 * g + q4, coh * 32 + b * 16);           // group -1 (the zero group in front of the dy tile): dy group j is at + (j + 1) * 2048
    nP[b] = aP[b] + STAGE_B;
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      aQ[tx][b] = lds0 + P_B + frag_off(tx + 4 * g + q4, cih * 32 + b * 16);
      nQ[tx][b] = aQ[tx][b] + STAGE_B;
    }
  }
  {                                                        // the zero groups of both buffers (no DMA ever touches them)
    const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int buf = 0; buf < 2; ++buf) {
      if (tid < 128) *reinterpret_cast<uint4*>(smem + buf * STAGE_B + tid * 16) = z;
      else *reinterpret_cast<uint4*>(smem + buf * STAGE_B + P_B - ZG + (tid - 128) * 16) = z;
    }
  }

  // ---- the previous launch's slab reduction (W9PJobDev).  Everything but the lane's float4 offset is wave-uniform; the state below is all there
  // is: jvoff = byte offset of this lane's float4 of the output being REQUESTED inside the layer (OOB: the lane has no output left — its loads
  // then read zeros and it stores nothing), jcv = the same for the unit being SUMMED, jdone = that unit completes its output.
  constexpr unsigned JOOB = 0xffffffffu;
  const unsigned jbase = (unsigned)blockIdx.x * p.job.per;
  const int jlayer = JM && jbase >= p.job.n4 ? 1 : 0;
  const unsigned jfirst = jbase - (jlayer ? p.job.n4 : 0u);
  const __amdgpu_buffer_rsrc_t rsJ = make_rsrc(p.job.slab[jlayer], (unsigned)p.job.nsplit * p.job.n4 * 16u);
  unsigned char* const jdst = reinterpret_cast<unsigned char*>(p.job.dst[jlayer]);
  constexpr float jz = JM == 2 ? 0.f : -0.f;
  const f32x4_t jneg0 = {jz, jz, jz, jz};
  f32x4_t jr[4] = {jneg0, jneg0, jneg0, jneg0}, jacc = jneg0, jt0 = jneg0, jt1 = jneg0;
  unsigned jvoff = JM && (unsigned)tid < p.job.per ? (jfirst + (unsigned)tid) * 16u : JOOB, jcv = JOOB;
  int jj = 0, jo = 0;              // unit within the output / output being requested
  unsigned jsb = 0;                // slab-group byte offset of that unit
  bool jdone = false;
  auto job_load = [&](int r) __attribute__((always_inline)) {     // load r of the unit being requested (soffset: a wave-uniform slab offset; out-of-range voffset reads as zeros)
    if constexpr (JM != 0) {
      const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsJ, (W9P_ABLATE & 32) ? (int)JOOB : (int)jvoff, (int)(jsb + (unsigned)r * p.job.sr), 0);
      jr[r] = __builtin_bit_cast(f32x4_t, v);
    }
  };
  auto job_advance = [&]() __attribute__((always_inline)) {       // behind the four loads of a unit: remember what the unit is for, step to the next one
    if constexpr (JM != 0) {
      jcv = jvoff;
      jdone = jj == p.job.upo - 1;
      ++jj; jsb += p.job.sb;
      if (jdone) {
        jj = 0; jsb = 0; ++jo;
        jvoff = (unsigned)(jo * 256 + tid) < p.job.per ? jvoff + 4096u : JOOB;      // (an exhausted lane stays exhausted: per is not reached again)
      }
    }
  };
  auto job_add = [&](int r) __attribute__((always_inline)) {      // step r of summing the unit in jr
    if constexpr (JM == 1) {
      jacc += jr[r];
    } else if constexpr (JM == 2) {
      if (r == 0) jt0 = jr[0] + jr[1];
      else if (r == 1) jt1 = jr[2] + jr[3];
      else if (r == 2) jt0 = jt0 + jt1;
      else jacc += jt0;
    }
  };
  auto job_finish = [&]() __attribute__((always_inline)) {        // the unit is summed: an output that is complete goes out
    if constexpr (JM != 0) {
      if (jdone) {
        if (jcv != JOOB) *reinterpret_cast<float4*>(jdst + jcv) = make_float4(jacc[0], jacc[1], jacc[2], jacc[3]);
        jacc = jneg0;
      }
    }
  };

  f32x4_t acc[9][2][2];                                    // [tap][ci block][co block]
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[t][a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- K loop.  K runs in 16-row groups; an MFMA takes an ALIGNED pair of groups (2k, 2k+1) as the low / high half of its 32-deep
  // operand on the input side (pairs Q_k), always: a pair that straddles two aligned ones would have to be assembled from halves of
  // two register tuples (4 v_mov per operand; the single-layer kernel pays that for every middle-row tap).  With groups g and the
  // vertical tap ty shifting the input by ty groups:
  //   ty = 0:  sum_g dy_g x_g      = sum_k  E_k . Q_k        E_k = dy groups (2k, 2k+1)
  //   ty = 2:  sum_g dy_g x_(g+2)  = sum_k  E_k . Q_(k+1)
  //   ty = 1:  sum_g dy_g x_(g+1)  = sum_k  O_k . Q_k        O_k = dy groups (2k-1, 2k), k = 0 .. KS, with dy_(-1) = dy_(2 KS) = 0:
  // the odd pairs O_k are read from LDS as tuples of their own (2 extra reads per co block and step), the two zero groups around the dy
  // tile give the ends, and the middle row costs one extra K-step per sub-image (12 MFMAs of 264).
  // tr-reads are inline asm with our own lgkmcnt waits: hipcc drains the in-flight LDS-DMA with vmcnt(0) before the builtin form
  // (gemm_tn_glds.hip).  Results stay whole 64-bit tuples until they are consumed behind the wait.
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
#define W9P_READ(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
  u32x2_t Ef[2][2 * KS], Of[2][KS + 1][2], Qg[3][2][NG];
  auto frag = [](const u32x2_t& lo, const u32x2_t& hi) __attribute__((always_inline)) {
    const u32x4_t v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8_t, v);
  };
  // dy group j (-1 .. 2 KS) sits at byte (j + 1) * 2048 from aP; read r of the fragments a K-step needs first:
  //   E_k halves (4: co block, half), O_k halves (4), Q pairs k and k + 1 (24: tx, ci block, 4 groups)
  auto read_E = [&](const unsigned* base, int k, int r) __attribute__((always_inline)) { const int b = r >> 1, h = r & 1; W9P_READ(Ef[b][2 * k + h], base[b], (2 * k + h + 1) * 2048); };
  auto read_O = [&](const unsigned* base, int k, int r) __attribute__((always_inline)) { const int b = r >> 1, h = r & 1; W9P_READ(Of[b][k][h], base[b], (2 * k + h) * 2048); };

  // prologue: sub-images s0 and s0 + 1 in flight, the first one landed, its first K-step's fragments fetched
  stage_setup(s0, true);
#pragma unroll
  for (int j = 0; j < NPW; ++j) issue_piece(j, 0);
  if (nst > 1) {
    stage_setup(s0 + 1, true);
#pragma unroll
    for (int j = 0; j < NPW; ++j) issue_piece(j, 1);
    w9p_wait_vmcnt<NPW>();
  } else {
    w9p_wait_vmcnt<0>();
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the zero groups are written
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < 4; ++r) { read_E(aP, 0, r); read_O(aP, 0, r); }
#pragma unroll
  for (int tx = 0; tx < 3; ++tx)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int j = 0; j < 2; ++j) W9P_READ(Qg[tx][b][j], aQ[tx][b], j * 2048);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  for (int it = 0; it < nst; ++it) {
    const bool dma = it + 2 < nst;                          // sub-image it + 2 goes where sub-image it is being read from
#pragma unroll
    for (int kb = 0; kb < KS; ++kb) {
      const bool last = kb == KS - 1;
      if (last) {
        // sub-image it+1 has landed (its DMA was issued a whole sub-image ago); behind the barrier every wave has also finished reading
        // sub-image it's buffer (its last fragments arrived at the end of the previous step), so sub-image it+2 may overwrite it
        w9p_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        stage_setup(s0 + it + 2, dma);                      // nothing left to fetch: every piece reads as zeros (no memory traffic)
      }
      // read r of this step (compile-time after unrolling): the fragments of the NEXT step.  steady state: E_(kb+1), O_(kb+1) (8),
      // Q pair kb + 2 (12), in the step before the last also O_KS (4); last step: the next sub-image's E_0, O_0 and Q pair 0 (20) -- its
      // Q pair 1 is fetched at the top of step 0 (12 reads in front of the regular ones, waited for before the bottom-row taps): 24
      // fewer registers live across the loop edge, where the allocator otherwise parks fragments in AGPRs and shuffles them back
      const int nr = last ? 20 : (kb == KS - 2 ? 24 : (kb == 0 ? 32 : 20));
      auto issue_read = [&](int r0) __attribute__((always_inline)) {
        if (W9P_ABLATE & 2) return;
        int r = r0;
        if (kb == 0) {
          if (r < 12) { const int tx = r >> 2, b = (r >> 1) & 1, j = 2 + (r & 1); W9P_READ(Qg[tx][b][j], aQ[tx][b], j * 2048); return; }
          r -= 12;
        }
        if (!last) {
          if (r < 4) read_E(aP, kb + 1, r);
          else if (r < 8) read_O(aP, kb + 1, r - 4);
          else if (r < 20) { const int q = r - 8, tx = q >> 2, b = (q >> 1) & 1, j = 2 * kb + 4 + (q & 1); W9P_READ(Qg[tx][b][j], aQ[tx][b], j * 2048); }
          else read_O(aP, KS, r - 20);
        } else {
          if (r < 4) read_E(nP, 0, r);
          else if (r < 8) read_O(nP, 0, r - 4);
          else { const int q = r - 8, tx = q >> 2, b = (q >> 1) & 1, j = q & 1; W9P_READ(Qg[tx][b][j], nQ[tx][b], j * 2048); }
        }
      };
      int slot = 0;
      // MFMA m of this step: ty-major (0: E_kb . Q_kb, 1: O_kb . Q_kb, 2: E_kb . Q_(kb+1)); the last step appends the middle row's extra
      // K-step O_KS . Q_KS
      const int nty = last ? 4 : 3;
#pragma unroll
      for (int tyi = 0; tyi < 4; ++tyi) {
        if (tyi >= nty) break;
        const int ty = tyi == 3 ? 1 : tyi;
        const int qk = tyi == 3 ? KS : (tyi == 2 ? kb + 1 : kb);       // input pair
        if (kb == 0 && tyi == 2 && !(W9P_ABLATE & 2)) {    // Q pair 1 (reads 0..11 of this step; 12 more have been issued since) has arrived
          asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              if (!(W9P_ABLATE & 4)) {
                const bf16x8_t fq = frag(Qg[tx][a][2 * qk], Qg[tx][a][2 * qk + 1]);
                const bf16x8_t fp = ty == 1 ? frag(Of[b][tyi == 3 ? KS : kb][0], Of[b][tyi == 3 ? KS : kb][1]) : frag(Ef[b][2 * kb], Ef[b][2 * kb + 1]);
                // inline asm with the accumulator as ONE tied in/out AGPR operand: with the builtin the allocator let 16-18 of the 36
                // loop-carried accumulators end an iteration in other registers than they started in and rotated them back through VGPRs at
                // the loop head (~200 v_accvgpr moves per sub-image).  An accumulator is next touched 35 MFMAs later: no hazard to cover.
                #if FEDFR_FP16
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[ty * 3 + tx][a][b]) : "v"(fq), "v"(fp));
#else
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[ty * 3 + tx][a][b]) : "v"(fq), "v"(fp));
#endif
              }
              if (slot < nr) issue_read(slot);
#if W9P_SPREAD
              // Round 6: the 16 LDS-DMA pieces of a stage go out a few per K-step over several steps — the last step of sub-image it (behind its barrier: the
              // buffer is free) and the first steps of sub-image it + 1 — instead of all in the last step, where 16 VMEM issues between 48 MFMAs beside 20
              // fragment reads stall the wave (a piece costs ~60 cycles among bare MFMAs, 100-185 inside a phase that already carries eight:
              // MI355X_MICROARCH.md).  The stage is first read a whole sub-image later; the vmcnt(0) at the top of the NEXT last step still covers every
              // piece.  stage_setup's state (set in the last step) persists until the next last step.
              {
                constexpr int PPS = NPW / 4;
                static_assert(NPW % 4 == 0, "pieces per wave and stage split over four K-steps");
                // (slot windows that do not depend on kb, so that few call sites survive in the rolled K-step body: the full-unroll budget.  Steps 0-2 use
                // slots 32 .. 35, behind step 0's 32 fragment reads and step 1's job.  Sub-image 0 re-issues stage 1, already in flight from the prologue: the
                // same bytes to the same place.)
#if W9P_SPREAD == 1
                if (last && slot >= 2 && slot - 2 < PPS && !(W9P_ABLATE & 1)) issue_piece(slot - 2, it & 1);
                if (!last && kb < 3 && slot >= 32 && slot - 32 < PPS && !(W9P_ABLATE & 1)) issue_piece(PPS * (kb + 1) + slot - 32, (it & 1) ^ 1);
#elif W9P_SPREAD == 2      // the same four steps, the pieces eight MFMAs apart (slots 4, 12, 20, 28)
                if (last && (slot & 7) == 4 && slot < 32 && !(W9P_ABLATE & 1)) issue_piece(slot >> 3, it & 1);
                if (!last && kb < 3 && (slot & 7) == 4 && slot < 32 && !(W9P_ABLATE & 1)) issue_piece(PPS * (kb + 1) + (slot >> 3), (it & 1) ^ 1);
#else                      // six steps (the last one and steps 0-4 of the next sub-image): 3 + 3 + 3 + 3 + 2 + 2 pieces at slots 33 .. 35
                if (last && slot >= 33 && slot < 36 && !(W9P_ABLATE & 1)) issue_piece(slot - 33, it & 1);
                if (!last && kb < 5 && slot >= 33 && slot < 36 && 3 * (kb + 1) + slot - 33 < NPW && !(W9P_ABLATE & 1)) issue_piece(3 * (kb + 1) + slot - 33, (it & 1) ^ 1);
#endif
              }
#else
              if (last && slot >= 2 && slot - 2 < NPW && !(W9P_ABLATE & 1)) issue_piece(slot - 2, it & 1);
#endif
              if (JM != 0 && kb == 1) {                     // the slab-reduction job: slots the second K-step leaves free (20 reads, no DMA)
                if (slot >= 20 && slot < 24) job_add(slot - 20);          // sum the unit requested one sub-image ago ...
                if (slot == 24) job_finish();
                if (slot >= 26 && slot < 30) job_load(slot - 26);         // ... and request the next one
                if (slot == 30) job_advance();
              }
              __builtin_amdgcn_sched_barrier(0);
              ++slot;
            }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {                           // the other buffer becomes the current one
      const unsigned t = aP[b]; aP[b] = nP[b]; nP[b] = t;
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) { const unsigned u = aQ[tx][b]; aQ[tx][b] = nQ[tx][b]; nQ[tx][b] = u; }
    }
  }
#undef W9P_READ
  w9p_wait_vmcnt<0>();
  if constexpr (JM != 0) {                                  // the last sub-image's unit; then the units a short split had no sub-images for
    for (int u = nst - 1; u < p.job.units; ++u) {
      if (u >= nst) {
#pragma unroll
        for (int r = 0; r < 4; ++r) job_load(r);
        job_advance();
        w9p_wait_vmcnt<0>();
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) job_add(r);
      job_finish();
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results are read by VALU moves next (hand-written MFMAs: no automatic hazard nops)

  // D[m = ci][n = co]: lane holds ci = block 16 + (lane >> 4) 4 + {0..3} for co = block 16 + (lane & 15) -> one float4 per tap and block pair
  float* slab = p.out[prob] + (size_t)split * p.cout * 9 * p.cin;
  const int NJ = 9 * p.cin;
  // (tap outer, ci block inner: the two 64-B halves of a 128-B line leave back to back — 0.7 us per launch over the ci-block-outer order;
  // non-temporal stores measured neutral, profiles/r04_ab_wgrad9p_store_order_v1.txt)
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int co = co0 + coh * 32 + b * 16 + (lane & 15);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int a = 0; a < 2; ++a)
        if (!(W9P_ABLATE & 16) || acc[t][a][b][0] == 12345.f)
          *reinterpret_cast<f32x4_t*>(slab + (size_t)co * NJ + t * p.cin + ci0 + cih * 32 + a * 16 + (lane >> 4) * 4) = acc[t][a][b];
  }
}
}  // namespace

static int w9p_lg(int W) {
  for (int lg = 0; lg < 4; ++lg)
    if (W == (14 << lg)) return lg;
  return -1;
}
bool wgrad9p_applies(const GemmTN& a, const GemmTN& b) {
  return g_wgrad9p && wgrad9_applies(a) && wgrad9_applies(b) && a.NI == b.NI && a.NJ == b.NJ && a.Kp == b.Kp && a.C == b.C && a.W == b.W &&
         a.NI % 64 == 0 && a.C % 64 == 0;
}
// 2 layers x tiles x splits ~ 256 workgroups (one per CU), at least two sub-images per split
int wgrad9p_pick_splits(int Kp, int NI, int NJ, int W) {
  const int stages = (Kp / (W * W)) << (2 * w9p_lg(W));
  const int tiles = (NI / 64) * (NJ / 9 / 64);
  int splits = 128 / tiles;
  if (splits < 1) splits = 1;
  if (splits > stages / 2) splits = stages / 2 > 0 ? stages / 2 : 1;
  const int per = ceil_div(stages, splits);
  return ceil_div(stages, per);
}

// the job's geometry for a carrying launch of `grid` workgroups with `per_split` sub-images each; false = it does not fit (stand-alone reduction)
static bool w9p_job_geometry(const W9PJob& job, int grid, int per_split, W9PJobDev* d, int* mode = nullptr) {
  if (!job.n || (job.n & 3) || job.nsplit < 1 || !job.slab[0] || !job.slab[1] || !job.dst[0] || !job.dst[1] || grid < 1) return false;
  const size_t n4 = job.n / 4, total4 = 2 * n4;
  if (total4 % (size_t)grid) return false;
  const size_t per = total4 / grid;
  if (n4 % per || (unsigned long long)job.nsplit * n4 * 16ull >= (1ull << 32)) return false;
  const bool wide = job.nsplit >= 16 && n4 <= 65536;       // == reduce_slabs_launch's choice (ew.hip): same summation order either way
  if (wide ? job.nsplit != 32 : (job.nsplit % 4) != 0) return false;
  const int nout = (int)((per + 255) / 256);
  const int upo = wide ? 8 : job.nsplit / 4;
  if ((long long)nout * upo > per_split) return false;     // one unit per sub-image: everything rides under the MFMAs
  if (d) {
    const unsigned s1 = (unsigned)n4 * 16u;                // one slab of one layer, in bytes
    d->slab[0] = job.slab[0]; d->slab[1] = job.slab[1]; d->dst[0] = job.dst[0]; d->dst[1] = job.dst[1];
    d->n4 = (unsigned)n4; d->per = (unsigned)per; d->nsplit = job.nsplit; d->upo = upo; d->units = nout * upo;
    d->sb = wide ? s1 : 4 * s1; d->sr = wide ? 8 * s1 : s1;
    *mode = wide ? 2 : 1;
  }
  return true;
}
int g_wgrad9p_bg = 1;   // option "wgrad9p_bg": a paired launch sums the previous pair's split-K slabs itself (0: stand-alone reduce_slabs launches)
bool wgrad9p_job_ok(const GemmTN& a, int splits, const W9PJob& job) {
  if (!g_wgrad9p_bg || splits < 1) return false;
  const int lg = w9p_lg(a.W);
  if (lg < 0 || a.NI % 64 || a.C % 64) return false;
  const int nstages = (a.Kp / (a.W * a.W)) << (2 * lg);
  const int grid = 2 * (a.NI / 64) * (a.C / 64) * splits;
  return w9p_job_geometry(job, grid, nstages / splits, nullptr);        // (the shortest split)
}

int launch_wgrad9_pair(const GemmTN& a, const GemmTN& b, int splits, hipStream_t st, const W9PJob* job) {
  FEDFR_REQUIRE(wgrad9p_applies(a, b), "wgrad9_pair: unsupported problem pair");
  W9P p{};
  p.dy[0] = a.P; p.dy[1] = b.P; p.x[0] = a.Q; p.x[1] = b.Q; p.out[0] = a.out; p.out[1] = b.out;
  p.cout = a.NI; p.cin = a.C; p.nci = a.C / 64;
  p.ntiles = (a.NI / 64) * p.nci;
  p.W = a.W; p.lg = w9p_lg(a.W);
  p.nstages = (a.Kp / (a.W * a.W)) << (2 * p.lg);
  p.per_split = ceil_div(p.nstages, splits);
  p.splits = splits;
  FEDFR_REQUIRE(splits >= 1 && ceil_div(p.nstages, p.per_split) == splits, "wgrad9_pair: splits=%d leaves an empty split", splits);
  p.dy_bytes = a.p_bytes; p.x_bytes = a.q_bytes;
  FEDFR_REQUIRE(a.p_bytes == b.p_bytes && a.q_bytes == b.q_bytes, "wgrad9_pair: operand sizes differ");
  FEDFR_REQUIRE(a.p_bytes < (1u << 30) - (1u << 24) && a.q_bytes < (1u << 30) - (1u << 24), "wgrad9_pair: operands must be smaller than 1 GiB");
  const dim3 grid(2 * p.ntiles * splits);
  int jm = 0;
  if (job && job->n)
    FEDFR_REQUIRE(w9p_job_geometry(*job, (int)grid.x, p.nstages / splits, &p.job, &jm), "wgrad9_pair: the slab-reduction job does not fit this launch (check gemm_tn_w9pair_job_ok)");
  ProfScope prof(16, 2.0 * 2.0 * a.NI * a.NJ * (double)a.Kp, st, gemm_tn_alg_bytes(a, 1) + gemm_tn_alg_bytes(b, 1));      // (the split-K slabs are overhead, not algorithmic)
  constexpr size_t lds = 2 * (size_t)STAGE_B;
  static PerDeviceOnce attr_once;     // hipFuncSetAttribute is per device (a Server process may drive several)
  attr_once.run([&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad9p_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad9p_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad9p_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  if (jm == 1) hipLaunchKernelGGL(wgrad9p_kernel<1>, grid, dim3(256), lds, st, p);
  else if (jm == 2) hipLaunchKernelGGL(wgrad9p_kernel<2>, grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL(wgrad9p_kernel<0>, grid, dim3(256), lds, st, p);
  FEDFR_LAUNCH_CHECK("wgrad9_pair");
  return FEDFR_OK;
}
