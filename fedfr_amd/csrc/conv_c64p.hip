// 3x3 / stride-1 / pad-1 convolution for the 64 -> 64 channel layers (112x112 and 56x56 maps; forward, and dgrad with the flipped
// shadow) as a PERSISTENT kernel with the whole filter bank in registers -- alone in its translation unit (gemm_dev.h).
//
// Why: on these layers K = 9 * 64 = 576 is tiny.  The tile-per-workgroup LDS-DMA kernel (conv_glds_impl.h, <112,2,...,64,true>) spends
// more time in its prologue (first image / weight round trip) and epilogue than in its 18 MFMA steps, once per 224-pixel tile and 28 times
// per CU on the 112x112 layer: 264 us for 118 GFLOP (0.18 of the MFMA peak, 3.5x the layer's HBM bound).  Here
//   * a workgroup (4 waves, one per SIMD) keeps its slice of ALL nine taps' weights in registers: wave (wm, wn) owns output channels
//     [32 wn, 32 wn + 32) = 2 fragments x 9 taps x 2 k-steps = 36 fragments = 144 VGPRs, loaded once per launch -- no weight traffic
//     through LDS at all, every LDS read feeds the pixel operand;
//   * it walks over a contiguous range of 224-pixel tiles (2 rows of a 112-wide map / 4 rows of a 56-wide one: 14 fragments of 16
//     pixels, no masked rows) with the NEXT tile's zero-padded image landing by LDS-DMA (`buffer_load ... lds`, out-of-image rows
//     fetched with an out-of-range offset = zeros) while the current one computes: two image buffers, one vmcnt(0) + barrier per tile;
//   * per tile 18 k-steps x (7 ds_read_b128 + 14 MFMA 16x16x32) per wave; the bf16 tile leaves through LDS in 16-byte rows;
//   * BatchNorm partial sums accumulate in registers over all tiles of the workgroup: 2 partial rows per workgroup, the rest of the
//     gemm_nt_stat_rows(M, N) rows the finalize kernel sums are written as zeros;
//   * dgrad launches (BWD = 1 / 2, round 3): the BatchNorm-backward reduction of the layer in front (ew_bn_bwd_reduce on (dx, bn_x): sum dz,
//     sum dz * xhat, sum dy * min(z, 0); 2 = with the PReLU mask) rides in the copy-out: the thread that stores a 16-byte piece of the
//     output tile has fetched the same piece of bn_x while the tile's MFMAs ran, sums stay in registers over all tiles of the workgroup,
//     ONE partial row [3][64] per workgroup at the end.  Replaces a pass over two 51 / 205 MB tensors per layer.
// LDS rows are 128 B (one padded pixel x 64 channels) with the 16-B chunk index XOR-ed by (row & 7), applied on the DMA source address
// (an LDS-DMA instruction writes 1 KiB linearly), exactly as in conv_glds_impl.h.
#include <algorithm>
#include "gemm_dev.h"
#include "epi_mfma.h"

int g_conv_c64p = 1;   // option "conv_c64p"
#ifndef C64P_MFMA_STATS
#define C64P_MFMA_STATS 1   // BatchNorm sums of the epilogue on the matrix cores (epi_mfma.h) instead of ~4-9 VALU operations per output element
#endif
#ifndef C64P_ABLATE
#define C64P_ABLATE 0   // timing experiments only (results are WRONG with any bit set): 1 no per-tile image DMA, 2 no output stores, 4 no MFMA loop, 8 no statistics arithmetic in the staging, 16 no BatchNorm-backward sums in the copy-out
#endif

namespace {
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int W_, int R_, bool STATS, int BWD>
__global__ __launch_bounds__(256) void conv3x3_c64p_kernel(GemmNT p, int ntiles, int per_wg, int stat_rows) {
  static_assert(!(STATS && BWD), "forward statistics and the backward reduction never meet in one launch");
  constexpr int PT = R_ * W_, PW = W_ + 2, PWL = (PW + 7) & ~7;
  constexpr int NPA = ((R_ + 2) * PWL + 7) / 8;            // LDS-DMA pieces (1 KiB = 8 image rows) per image buffer
  constexpr int AP = (NPA + 3) / 4;                        // pieces per wave (the last round may be partial)
  constexpr int A_BYTES = NPA * 1024;
  constexpr int TM = 7, TN = 2, CST = 64 * 2 + 16;
  static_assert(PT == 224 && W_ % R_ == 0 && 2 * A_BYTES + PT * CST <= 160 * 1024, "tile geometry");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* sA = smem;                                // [2][A_BYTES]
  unsigned char* sC = smem + 2 * A_BYTES;                  // [PT][CST]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lg = lane >> 4;
  const int t_beg = (int)blockIdx.x * per_wg, t_end = min(t_beg + per_wg, ntiles);
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);
  constexpr int TPI = W_ / R_;                             // tiles per image
  constexpr unsigned OOB = 0xfffffff0u;                    // beyond any buffer: the load writes zeros to LDS

  // ---- the BatchNorm finalize sums stat_rows partial rows: rows [2 * gridDim.x, stat_rows) are not produced by this tiling -> zeros
  if constexpr (STATS) {
    const int first = 2 * (int)gridDim.x, n4 = (stat_rows - first) * 32;          // float4s (a row = 2 x 64 floats)
    for (int i = (int)blockIdx.x * 256 + tid; i < n4; i += (int)gridDim.x * 256)
      reinterpret_cast<float4*>(p.stats + (size_t)first * 128)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (t_beg >= t_end) {                                    // (cannot happen with the launcher's grid; keeps every wave's exit unconditional)
    return;
  }

  // ---- weights -> registers: fragment (tap, ks, ni): lane holds B[n = 32 wn + 16 ni + l15][k = 64 tap + 32 ks + 8 lg .. + 7]
  bf16x8_t wf[9][2][TN];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const unsigned off = ((unsigned)(wn * 32 + ni * 16 + l15) * (unsigned)p.K + (unsigned)(tap * 64 + ks * 32 + lg * 8)) * 2u;
        wf[tap][ks][ni] = __builtin_bit_cast(bf16x8_t, buf_load16(rsB, off));
      }

  // ---- LDS-DMA source plan of a tile.  Lane l of every piece: row (l >> 3) of the piece, logical 16-B chunk (l & 7) ^ (l >> 3).
  // LDS image row r <-> padded pixel (y0 - 1 + r / PWL, r % PWL - 1) of image img
  // PWL / 8 pieces per padded image row, so a piece never straddles rows: its row and column block are wave-uniform (scalar registers),
  // the lane contributes (prow, pch) only, and per tile a piece costs a scalar add + ~4 vector instructions (a naive per-lane
  // r -> (row, column) -> pixel -> offset chain cost ~1 us of VALU per tile, as much as half the MFMA time)
  const int prow = lane >> 3, pch = (lane & 7) ^ prow;
  constexpr int PPR = PWL / 8;
  static_assert(PWL % 8 == 0 && W_ % 8 == 0, "pieces must not straddle padded rows");
  const int lane_off = prow * 128 + pch * 16;
  int p_soff[AP], p_ry[AP], p_edge[AP];                    // wave-uniform: byte offset of the piece's first LDS row, its padded row, 1 = first / 2 = last piece of a row
#pragma unroll
  for (int j = 0; j < AP; ++j) {
    const int piece = j * 4 + wave, ry = piece / PPR, cb = piece - ry * PPR;
    p_ry[j] = ry;
    p_soff[j] = (ry * W_ + cb * 8) * 128;
    p_edge[j] = cb == 0 ? 1 : (cb == PPR - 1 ? 2 : 0);
  }
  const bool first_ok = prow != 0, last_ok = prow == 0;     // column 0 is padding; the last piece of a row holds column W_ (prow 0) and filler
  auto issue_a = [&](int tile, int abuf) {
    const int img = tile / TPI, y0 = (tile - img * TPI) * R_;
    const int base = (img * (W_ * W_) + (y0 - 1) * W_ - 1) * 128;      // byte offset of padded (row 0, column 0) of this tile; valid rows land >= 0
#pragma unroll
    for (int j = 0; j < AP; ++j) {
      const int piece = j * 4 + wave;
      if (piece < NPA) {                                   // wave-uniform
        const int yy = y0 + p_ry[j];
        const bool row_ok = p_ry[j] < R_ + 2 && yy >= 1 && yy <= W_;          // scalar
        const bool col_ok = p_edge[j] == 0 ? true : (p_edge[j] == 1 ? first_ok : last_ok);
        const unsigned vo = (row_ok && col_ok) ? (unsigned)(base + p_soff[j] + lane_off) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(sA + abuf * A_BYTES + piece * 1024), 16, (int)vo, 0, 0, 0);
      }
    }
  };

  // W_ a multiple of 16 (112): a fragment never straddles image rows, row y = t / W_ is wave-uniform per (wm, mi) and the swizzle key
  // (r & 7) = (l15 + dx) & 7 does not depend on mi (PWL and 16 are multiples of 8): ONE address register per horizontal tap, the
  // fragment index is an immediate.  Otherwise (56): one register per (dx, mi).
  constexpr bool ROWFRAG = W_ == 112;                      // tile row == wm, fragment mi = columns [16 mi, 16 mi + 16)
  constexpr int NADR = ROWFRAG ? 1 : TM;
  int a_adr[3][NADR];
#pragma unroll
  for (int mi = 0; mi < NADR; ++mi) {
    const int t = wm * 112 + mi * 16 + l15;
    const int y = t / W_, x = t - y * W_;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int r = y * PWL + x + dx;
      a_adr[dx][mi] = r * 128 + ((lg ^ (r & 7)) << 4);    // k-step 1 flips chunk bit 2: XOR 64
    }
  }
  // LDS byte address of this lane's staging slot for fragment (mi = 0, ni = 0): pixel 112 wm + l15, channels 32 wn + 4 lg .. + 3
  const unsigned sc_base = (unsigned)reinterpret_cast<size_t>((lds_ptr_t)sC);
  const unsigned sc_adr = sc_base + (unsigned)((wm * 112 + l15) * CST + (wn * 32 + lg * 4) * 2);
  const unsigned co_adr = sc_base + (unsigned)((tid >> 3) * CST + (tid & 7) * 16);     // copy-out: row tid / 8 (+ 32 i), 16-B chunk tid % 8

  float ssum[TN][4], ssq[TN][4];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni)
#pragma unroll
    for (int q = 0; q < 4; ++q) ssum[ni][q] = ssq[ni][q] = 0.f;

  // MST (round 5): with ONE wave per SIMD nothing covers the epilogue's VALU work — a timing build without the statistics arithmetic runs the
  // 112x112 layer in 114 instead of 134 us forward, 105 instead of 148 us in the dgrad with the BatchNorm-backward sums.  They are column sums of
  // the staged tile (and of its products with the BatchNorm input tile): on the matrix cores (epi_mfma.h) wave w takes channels 16 w .. 16 w + 15,
  // per 32-pixel step one transposed fragment of the staged tile (BWD: and one of the x tile, which the copy-out threads park in the image buffer
  // this tile has just finished with), accumulators persistent over the tiles of the workgroup.  LDS accesses are inline asm (the next tile's
  // LDS-DMA is in flight; hipcc would drain it in front of every LDS access it can see), software-pipelined two steps deep.
  // Measured in the step (profiles/r05_ab_c64p_mfma_stats_v1.txt): forward 39.7 -> 37.6 us (56x56) / 131 -> 126 us (112x112), PReLU dgrad 56.0 -> 46.9 us,
  // plain dgrad 143 -> 141 us at 112x112 but 42.3 -> 43.2 us at 56x56 (its 4 VALU operations per element cost what the x tile's trip through
  // LDS + one more barrier cost): that one keeps the register sums.
  constexpr bool MST = C64P_MFMA_STATS != 0 && (STATS || BWD == 2 || (BWD == 1 && W_ == 112));
  f32x4_t g1 = {0.f, 0.f, 0.f, 0.f}, g2 = {0.f, 0.f, 0.f, 0.f}, g3 = {0.f, 0.f, 0.f, 0.f}, g4 = {0.f, 0.f, 0.f, 0.f};
  const unsigned tr_rel = (unsigned)((8 * lg + (l15 >> 2)) * CST + (wave * 16 + 4 * (l15 & 3)) * 2);     // this lane's transposed-read offset inside a [224][CST] tile
  const unsigned sa_base = (unsigned)reinterpret_cast<size_t>((lds_ptr_t)sA);
  const bf16x8_t ones = mfma_ones8();
  PreluThr th = {0u, 0u, 0.f, 0.f};
  float c_sc = 1.f, c_sh = 0.f, c_al = 1.f;                // BWD == 2: this lane's channel 16 wave + l15
  if constexpr (MST && BWD == 2) {
    const int n = wave * 16 + l15;
    c_sc = p.bgamma[n] * p.brstd[n];
    c_sh = p.bbeta[n] - p.bmean[n] * c_sc;
    c_al = p.balpha[n];
    th = prelu_threshold(c_sc, c_sh);
  }

  // !MST, BWD: this thread stores chunk (tid & 7) = channels 8 (tid & 7) .. + 7 of rows tid / 8 + 32 i of EVERY tile: 8 running sums per quantity
  float b1[8], b2[8], b3[8], bsc[8], bsh[8], bal[8];
  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(BWD ? (const void*)p.bx : (const void*)p.A, BWD ? (unsigned)((size_t)p.M * 64 * 2) : p.a_bytes);
  if constexpr (BWD != 0 && !MST) {
#pragma unroll
    for (int q = 0; q < 8; ++q) b1[q] = b2[q] = b3[q] = 0.f;
    if constexpr (BWD == 2) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int n = (tid & 7) * 8 + q;
        const float g = p.bgamma[n] * p.brstd[n];
        bsc[q] = g;
        bsh[q] = p.bbeta[n] - p.bmean[n] * g;
        bal[q] = p.balpha[n];
      }
    }
  }

  issue_a(t_beg, 0);
  for (int tile = t_beg; tile < t_end; ++tile) {
    const int cur = (tile - t_beg) & 1;
    // this tile's image has landed (own pieces: counted vmcnt — the 7 output stores of the previous tile were issued AFTER those DMA
    // pieces and may stay in flight; everybody's pieces: barrier); the previous tile's staging reads are done
    if (tile == t_beg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((C64P_ABLATE & 2) ? 1 : PT * 8 / 256) : "memory");
    __builtin_amdgcn_s_barrier();
    if (tile + 1 < t_end && !(C64P_ABLATE & 1)) issue_a(tile + 1, cur ^ 1);
    const unsigned char* cA = sA + cur * A_BYTES;
    uint4 xr[7];                                             // BWD: bn_x at the pieces this thread stores below; lands while the MFMAs run
    if constexpr (BWD != 0) {
      const unsigned xo = ((unsigned)tile * PT + (unsigned)(tid >> 3)) * 128u + (unsigned)(tid & 7) * 16u;
#pragma unroll
      for (int i = 0; i < 7; ++i) xr[i] = buf_load16(rsX, xo + (unsigned)i * 4096u);
    }

    f32x4_t acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // 18 k-steps (tap, ks); the pixel fragments of step s + 1 are read while the MFMAs of step s run (two register sets)
    bf16x8_t fa[2][TM];
    auto rd = [&](int set, int step) {
      const int tap = step >> 1, ks = step & 1, dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
        fa[set][mi] = *reinterpret_cast<const bf16x8_t*>(cA + (a_adr[dx][ROWFRAG ? 0 : mi] ^ (ks * 64)) + (ROWFRAG ? mi * 2048 : 0) + dy * (PWL * 128));
    };
    rd(0, 0);
#pragma unroll
    for (int step = 0; step < ((C64P_ABLATE & 4) ? 1 : 18); ++step) {
      if (step + 1 < 18) rd((step + 1) & 1, step + 1);
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) acc[ni][mi] = MFMA16(wf[step >> 1][step & 1][ni], fa[step & 1][mi], acc[ni][mi]);
      if (step + 1 < 18) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMAs
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
        }
      }
      __builtin_amdgcn_sched_barrier(0);     // hipcc otherwise sinks every read to just before its first use (one exposed LDS latency per 2 MFMAs)
    }

    // ---- epilogue: bf16 tile through LDS.  D: channel n = 32 wn + 16 ni + 4 lg + q, pixel = m_pix[mi]
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        bf16_t h[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          h[q] = f2bf(acc[ni][mi][q]);
          if constexpr (STATS && !MST && !(C64P_ABLATE & 8)) {
            const float v = bf2f(h[q]);
            ssum[ni][q] += v;
            ssq[ni][q] += v * v;
          }
        }
        uint2 pk;
        pk.x = (unsigned)h[0] | ((unsigned)h[1] << 16);
        pk.y = (unsigned)h[2] | ((unsigned)h[3] << 16);
        // inline asm on purpose: with the next tile's LDS-DMA in flight hipcc puts s_waitcnt vmcnt(0) in front of any LDS store it can
        // see (it cannot prove that sC and the DMA target are disjoint), which serialised the image transfer with the epilogue
        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(sc_adr), "v"(pk), "n"(mi * 16 * CST + ni * 32) : "memory");
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // own staging writes done ...
    __builtin_amdgcn_s_barrier();                           // ... everybody's (raw barrier: __syncthreads() would drain the DMA as well)
    if constexpr (MST && BWD != 0) {
      // the BatchNorm input pieces this thread fetched while the MFMAs ran -> image buffer `cur` (idle until the tile after next is requested; every wave is past
      // its last fragment read of it: the barrier above), same [row][CST] layout as the staged tile
      const unsigned xw = sa_base + (unsigned)(cur * A_BYTES) + (unsigned)((tid >> 3) * CST + (tid & 7) * 16);
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const u32x4_t xv = {xr[i].x, xr[i].y, xr[i].z, xr[i].w};
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(xw), "v"(xv), "n"(i * 32 * CST) : "memory");
      }
    }
    const size_t m0 = (size_t)tile * PT;
    {
      // copy-out, 7 rows per thread: the staging reads are inline asm for the same reason as the writes above (hipcc would wait for
      // the in-flight image DMA before the first ds_read it can see); all seven are issued, then one wait
      u32x4_t v0, v1, v2, v3, v4, v5, v6;
      asm volatile(
          "ds_read_b128 %0, %7 offset:%8\n\tds_read_b128 %1, %7 offset:%9\n\tds_read_b128 %2, %7 offset:%10\n\tds_read_b128 %3, %7 offset:%11\n\t"
          "ds_read_b128 %4, %7 offset:%12\n\tds_read_b128 %5, %7 offset:%13\n\tds_read_b128 %6, %7 offset:%14\n\ts_waitcnt lgkmcnt(0)"
          : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6)
          : "v"(co_adr), "n"(0 * 32 * CST), "n"(1 * 32 * CST), "n"(2 * 32 * CST), "n"(3 * 32 * CST), "n"(4 * 32 * CST), "n"(5 * 32 * CST),
            "n"(6 * 32 * CST)
          : "memory");
      bf16_t* out = p.Cb + (m0 + (tid >> 3)) * p.ldc + (tid & 7) * 8;
      const size_t rs = (size_t)32 * p.ldc;
      const bool full = !(C64P_ABLATE & 2);
      *reinterpret_cast<u32x4_t*>(out) = v0;
      if (full) {
        *reinterpret_cast<u32x4_t*>(out + rs) = v1;
        *reinterpret_cast<u32x4_t*>(out + 2 * rs) = v2;
        *reinterpret_cast<u32x4_t*>(out + 3 * rs) = v3;
        *reinterpret_cast<u32x4_t*>(out + 4 * rs) = v4;
        *reinterpret_cast<u32x4_t*>(out + 5 * rs) = v5;
        *reinterpret_cast<u32x4_t*>(out + 6 * rs) = v6;
      }
      if constexpr (BWD != 0 && !MST && !(C64P_ABLATE & 16)) {
        // exactly the sums of ew_bn_bwd_reduce on the stored (bf16) dx, on the RAW x: sum dz * xhat = rstd (sum dz x - mean sum dz) is formed
        // once per column at the end (as in the LDS-DMA kernel's fused epilogue, conv_glds_impl.h)
        const u32x4_t vv[7] = {v0, v1, v2, v3, v4, v5, v6};
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          float dy[8], xv[8];
          unpack8(__builtin_bit_cast(uint4, vv[i]), dy);
          unpack8(xr[i], xv);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            if constexpr (BWD == 2) {
              const float z = xv[q] * bsc[q] + bsh[q];
              const bool neg = z <= 0.f;
              b3[q] += neg ? dy[q] * z : 0.f;
              const float dz = neg ? dy[q] * bal[q] : dy[q];
              b1[q] += dz;
              b2[q] += dz * xv[q];
            } else {
              b1[q] += dy[q];
              b2[q] += dy[q] * xv[q];
            }
          }
        }
      }
    }
    if constexpr (MST && BWD != 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own x pieces are in LDS ...
      __builtin_amdgcn_s_barrier();                         // ... everybody's
    }
    if constexpr (MST && (STATS || BWD != 0)) {
      // ---- the tile's column sums on the matrix cores: 7 steps of 32 pixels; step j's fragments are requested two steps ahead, the wait that
      // retires them is tied to their registers ("+v") so that no MFMA is scheduled in front of it
      constexpr int NRD = BWD != 0 ? 4 : 2;                  // transposed reads per step (staged tile; BWD: + x tile)
      const unsigned trc = sc_base + tr_rel, trx = sa_base + (unsigned)(cur * A_BYTES) + tr_rel;
      s16x4_t lo[7], hi[7], xl[7], xh[7];
      // (macros, not lambdas: clang refuses asm operands that name captured variables)
#define C64P_ISSUE(j)                                                                                                                          \
  do {                                                                                                                                         \
    if constexpr (BWD != 0)                                                                                                                    \
      asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%6\n\tds_read_b64_tr_b16 %1, %4 offset:%7\n\tds_read_b64_tr_b16 %2, %5 offset:%6\n\t"      \
                   "ds_read_b64_tr_b16 %3, %5 offset:%7"                                                                                       \
                   : "=&v"(lo[j]), "=&v"(hi[j]), "=&v"(xl[j]), "=&v"(xh[j])                                                                    \
                   : "v"(trc), "v"(trx), "n"((j) * 32 * CST), "n"((j) * 32 * CST + 4 * CST)                                                    \
                   : "memory");                                                                                                                \
    else                                                                                                                                       \
      asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                                                \
                   : "=&v"(lo[j]), "=&v"(hi[j])                                                                                                \
                   : "v"(trc), "n"((j) * 32 * CST), "n"((j) * 32 * CST + 4 * CST)                                                              \
                   : "memory");                                                                                                                \
  } while (0)
  // step j's fragments have landed when at most `left` younger reads are outstanding
#define C64P_RETIRE(j, left)                                                                                                                   \
  do {                                                                                                                                         \
    if constexpr (BWD != 0) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(lo[j]), "+v"(hi[j]), "+v"(xl[j]), "+v"(xh[j]) : "n"(left) : "memory"); \
    else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(lo[j]), "+v"(hi[j]) : "n"(left) : "memory");                                             \
  } while (0)
      auto step = [&](auto J_) {
        constexpr int j = decltype(J_)::value;
        s16x8_t d8;
        d8[0] = lo[j][0]; d8[1] = lo[j][1]; d8[2] = lo[j][2]; d8[3] = lo[j][3]; d8[4] = hi[j][0]; d8[5] = hi[j][1]; d8[6] = hi[j][2]; d8[7] = hi[j][3];
        const bf16x8_t dfr = __builtin_bit_cast(bf16x8_t, d8);
        g1 = MFMA16(ones, dfr, g1);                          // every row: the column sums of the tile
        if constexpr (STATS) {
          g2 = MFMA16(dfr, dfr, g2);                         // diagonal: sums of squares
        } else {
          s16x8_t x8;
          x8[0] = xl[j][0]; x8[1] = xl[j][1]; x8[2] = xl[j][2]; x8[3] = xl[j][3]; x8[4] = xh[j][0]; x8[5] = xh[j][1]; x8[6] = xh[j][2]; x8[7] = xh[j][3];
          const bf16x8_t xfr = __builtin_bit_cast(bf16x8_t, x8);
          g2 = MFMA16(dfr, xfr, g2);                         // diagonal: sum dy * x
          if constexpr (BWD == 2) {
            const bf16x8_t pfr = __builtin_bit_cast(bf16x8_t, prelu_pos(d8, x8, th));
            g3 = MFMA16(ones, pfr, g3);                      // sum of dy over z > 0
            g4 = MFMA16(pfr, xfr, g4);                       // diagonal: sum of dy * x over z > 0
          }
        }
      };
      using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
      using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
      using I6 = std::integral_constant<int, 6>;
      C64P_ISSUE(0); C64P_ISSUE(1); C64P_ISSUE(2);
      C64P_RETIRE(0, 2 * NRD); step(I0{}); C64P_ISSUE(3);
      C64P_RETIRE(1, 2 * NRD); step(I1{}); C64P_ISSUE(4);
      C64P_RETIRE(2, 2 * NRD); step(I2{}); C64P_ISSUE(5);
      C64P_RETIRE(3, 2 * NRD); step(I3{}); C64P_ISSUE(6);
      C64P_RETIRE(4, 2 * NRD); step(I4{});
      C64P_RETIRE(5, NRD); step(I5{});
      C64P_RETIRE(6, 0); step(I6{});
#undef C64P_ISSUE
#undef C64P_RETIRE
    }
  }
  if constexpr (MST && BWD != 0) {
    // one partial row [3][64] per workgroup, straight from the 16 diagonal lanes of every wave
    if ((l15 >> 2) == lg) {
      const int n = wave * 16 + l15;
      float t0 = g1[0], t1 = mfma_diag(g2, l15), t2 = 0.f;
      if constexpr (BWD == 2) {
        const float sp = g3[0], spx = mfma_diag(g4, l15);
        const float sn = t0 - sp, snx = t1 - spx;            // sums over the elements with z <= 0
        t0 = sp + c_al * sn;
        t1 = spx + c_al * snx;
        t2 = c_sc * snx + c_sh * sn;
      }
      float* o = p.bpart + (size_t)blockIdx.x * 3 * 64 + n;
      o[0] = t0;
      o[64] = p.brstd[n] * (t1 - p.bmean[n] * t0);
      o[128] = t2;
    }
  }
  if constexpr (MST && STATS) {
    // rows 2 b (the sums) and 2 b + 1 (zeros: the register path left one row per wave row)
    if ((l15 >> 2) == lg) {
      const int n = wave * 16 + l15;
      float* prow_ = p.stats + (size_t)((int)blockIdx.x * 2) * 128;
      prow_[n] = g1[0];
      prow_[64 + n] = mfma_diag(g2, l15);
      prow_[128 + n] = 0.f;
      prow_[192 + n] = 0.f;
    }
  }
  if constexpr (BWD != 0 && !MST) {
    // one partial row per workgroup: the 32 row groups meet in LDS (the image buffers are idle: the last tile's MFMA loop is behind the
    // staging barrier of its epilogue, no DMA is in flight)
    __syncthreads();
    float* red = reinterpret_cast<float*>(sA);               // [32][3][64]
    const int rg = tid >> 3, c8 = (tid & 7) * 8;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      red[(rg * 3 + 0) * 64 + c8 + q] = b1[q];
      red[(rg * 3 + 1) * 64 + c8 + q] = b2[q];
      red[(rg * 3 + 2) * 64 + c8 + q] = BWD == 2 ? b3[q] : 0.f;
    }
    __syncthreads();
    if (tid < 64) {
      float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) {
        t0 += red[(r * 3 + 0) * 64 + tid];
        t1 += red[(r * 3 + 1) * 64 + tid];
        t2 += red[(r * 3 + 2) * 64 + tid];
      }
      float* o = p.bpart + (size_t)blockIdx.x * 3 * 64 + tid;
      o[0] = t0;
      o[64] = p.brstd[tid] * (t1 - p.bmean[tid] * t0);
      o[128] = t2;
    }
  }
  if constexpr (STATS && !MST) {
    float* prow_ = p.stats + (size_t)((int)blockIdx.x * 2 + wm) * 128;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float a = row16_sum(ssum[ni][q]), b = row16_sum(ssq[ni][q]);
        const int n = wn * 32 + ni * 16 + lg * 4 + q;
        if (l15 == 0) {
          prow_[n] = a;
          prow_[64 + n] = b;
        }
      }
  }
}

template <int W_, int R_, bool STATS, int BWD>
int launch_c64p(GemmNT p, hipStream_t st) {
  constexpr int PT = R_ * W_, PWL = (W_ + 2 + 7) & ~7, NPA = ((R_ + 2) * PWL + 7) / 8;
  constexpr size_t lds = 2 * (size_t)NPA * 1024 + (size_t)PT * (64 * 2 + 16);
  const int ntiles = p.M / PT;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int per_wg = ceil_div(ntiles, std::min(ntiles, cus));
  const int grid = ceil_div(ntiles, per_wg);
  const int stat_rows = gemm_nt_stat_rows(p.M, p.N);
  FEDFR_REQUIRE(!p.stats || 2 * grid <= stat_rows, "conv3x3_c64p: %d partial rows do not fit gemm_nt_stat_rows = %d", 2 * grid, stat_rows);
  static PerDeviceOnce attr_once;     // hipFuncSetAttribute is per device (a Server process may drive several)
  attr_once.run([&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c64p_kernel<W_, R_, STATS, BWD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  if (BWD) {
    FEDFR_REQUIRE(p.bx && p.bmean && p.brstd && (BWD == 1 || (p.bgamma && p.bbeta && p.balpha)), "conv3x3_c64p: the fused BatchNorm-backward reduction needs bx / mean / rstd (and gamma / beta / alpha with PReLU)");
    if (p.bwd_fused) *p.bwd_fused = grid;                  // one partial row [3][64] per workgroup
  }
  ProfScope prof(15, 2.0 * p.M * p.N * (double)p.K, st, gemm_nt_alg_bytes(p, 1));           // slot 15: the 64-channel 3x3 layers (56x56, 112x112)
  hipLaunchKernelGGL((conv3x3_c64p_kernel<W_, R_, STATS, BWD>), dim3(grid), dim3(256), lds, st, p, ntiles, per_wg, stat_rows);
  FEDFR_LAUNCH_CHECK("conv3x3_c64p");
  return FEDFR_OK;
}
}  // namespace

int conv_c64p_grid(int M) {                 // workgroups of a launch over M output pixels (224-pixel tiles dealt out evenly, at most one workgroup per CU)
  const int ntiles = M / 224;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int per_wg = ceil_div(ntiles, std::min(ntiles, cus));
  return ceil_div(ntiles, per_wg);
}
bool conv_c64p_applies(const GemmNT& p) {
  return g_conv_c64p && p.mode == 1 && p.S == 3 && p.C == 64 && p.N == 64 && p.K == 576 && p.stride == 1 && p.pad == 1 && p.up == 1 && p.H == p.W &&
         (p.W == 112 || p.W == 56) && p.Ho == p.H && p.Wo == p.W && p.M % (p.W * p.W) == 0 && p.Cb && p.ldc == 64 && !p.Cf && !(p.bpart && p.stats) &&
         !p.esc && !p.eadd && !p.Cb2 && !p.par_on && !p.bmom;
}
int launch_conv_c64p(GemmNT p, hipStream_t st) {
  FEDFR_REQUIRE(conv_c64p_applies(p), "conv3x3_c64p: unsupported shape");
  if (p.bpart) {
    if (p.balpha) return p.W == 112 ? launch_c64p<112, 2, false, 2>(p, st) : launch_c64p<56, 4, false, 2>(p, st);
    return p.W == 112 ? launch_c64p<112, 2, false, 1>(p, st) : launch_c64p<56, 4, false, 1>(p, st);
  }
  if (p.stats) return p.W == 112 ? launch_c64p<112, 2, true, 0>(p, st) : launch_c64p<56, 4, true, 0>(p, st);
  return p.W == 112 ? launch_c64p<112, 2, false, 0>(p, st) : launch_c64p<56, 4, false, 0>(p, st);
}
