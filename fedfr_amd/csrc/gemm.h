// Descriptors + launchers for the two MFMA workhorse kernels (bf16 in, fp32 accumulate).
//
//  gemm_nt : C[m][n] = sum_k A(m,k) * B[n][k]      A = plain row-major or conv gather (NHWC)
//            used for conv fwd, conv dgrad (flipped/transposed weight shadow), fc fwd.
//  gemm_tn : C[i][j] = sum_p P[p][i] * Q(p,j)      both operands reduction-major in memory
//            (LDS transpose reads, ds_read_b64_tr_b16); used for conv wgrad, fc dgrad/wgrad.
#pragma once
#include "common.h"

struct GemmNT {
  const bf16_t* A;
  const bf16_t* B;        // [N][K], K contiguous
  int M, N, K;
  int mode;               // 0 plain A[M][lda]; 1 conv gather
  int H, W, C;            // gather source tensor [img][H][W][C]
  int Ho, Wo;             // row m -> (img, ho, wo)
  int S;                  // filter width (tap = r*S + s)
  int stride, pad, up;    // pos = ho*stride + r - pad; up==2: pos must be even, then pos/2
  int lda;
  int cpt;                // 64-wide k-chunks per tap (C/64)
  bf16_t* Cb;             // bf16 out [M][ldc] (or null)
  int ldc;
  float* Cf;              // fp32 split-K slabs [splits][M][N] (or null)
  float* stats;           // [gridM*WM][2][N] per-column partial (sum, sumsq) of the bf16-rounded output
  int ksteps_total, ksteps_per_split;
  int nbn;                // number of N tiles (for the 1-D XCD-swizzled grid)
  unsigned a_bytes, b_bytes;   // buffer-descriptor ranges (filled by the launcher)
  // optional fused BatchNorm-backward reduction on the OUTPUT tile (dgrad launches): with dy = this GEMM's bf16
  // output and x = the BN input [M][N], writes per-M-tile partials [ceil(M/128)][3][N] of (sum dz, sum dz*xhat,
  // sum dy*min(z,0)) exactly like ew_bn_bwd_reduce.  Only the halo2 kernel implements it: *bwd_fused (host) is set
  // to the number of partial rows when it did, left untouched otherwise.
  const bf16_t* bx;
  const float *bmean, *brstd, *bgamma, *bbeta, *balpha;
  float* bpart;
  int* bwd_fused;
  // bmom != 0 (forward launches, round 3): the same epilogue leaves RAW moments of the output y against a same-shape tensor bx instead —
  // rows [3][N] of (sum y, sum y * bx, sum y * y) — from which the pass that follows derives the statistics of y AND of
  // bn(y) + bx (the next block's bn1) without a pass over that sum (net.hip, conv2 of a residual block).  bmean / brstd: any readable [N].
  int bmom;               // 2 (dgrad launches, sphnet): a bare PReLU(+bias) precedes the conv — the OUTPUT becomes dz = dy * prelu'(bx + bias) (bias in
                          // bbeta or null, slopes in balpha), rows [3][N] = (sum dz, sum dy z over z <= 0, same): the PReLU's backward pass disappears
  // stride-2 dgrad by output-parity class (gemm.hip: nt_launch_parity): this launch computes the output pixels (2 h2 + par_h, 2 w2 + par_w)
  // of a [img][outH][outW] map with only the filter taps that reach a real (non-inserted-zero) input: 1, 2, 2 or 4 of the 9
  int par_on, par_h, par_w, outH, outW;
  // optional OUTPUT epilogue of the LDS-DMA conv kernels (eval-mode forward, where the BatchNorm that follows a conv is a known
  // per-channel affine): y = acc * esc[n] + esh[n] on the fp32 accumulators, PReLU with ealpha[n] if given, + eadd[m][n] (bf16, the
  // identity path) if given; Cb2 (optional) receives y2 = y * esc2[n] + esh2[n] (the next block's bn1 of this block's output)
  const float *esc, *esh, *ealpha;
  const bf16_t* eadd;
  const float *esc2, *esh2;
  bf16_t* Cb2;
  // ... and, for nets whose activation follows the conv with no normalisation in between (sphnet: t = prelu(conv + bias) (+ identity)):
  // Cb2 = prelu_{e2alpha}(y * esc2 + esh2) + e2add with esc2 / esh2 / e2add optional (1 / 0 / none) — the same fp32 expressions on the
  // same bf16-rounded conv output as the separate ew_bn_apply pass, so both forms give identical bits; Cb keeps the raw conv output
  const float* e2alpha;
  const bf16_t* e2add;
  unsigned long long* dbg;   // diagnostics builds only (tools/stamp_halo2.hip): per-block in-kernel clock stamps
};

struct GemmTN {
  const bf16_t* P;        // [Kp][ldp]  (dy side)
  const bf16_t* Q;        // plain [Kp][ldq] or conv gather source NHWC
  int Kp, NI, NJ;
  int mode;               // 0 plain; 1 conv gather (row p -> (img,ho,wo) of the dy tensor)
  int H, W, C, Ho, Wo, S, stride, pad;
  FastDiv dHoWo, dWo, dHo;
  int ldp, ldq;
  float* out;             // [splits][NI][NJ]
  int ksteps_total, ksteps_per_split;
  int nbj, ntiles;        // number of j tiles / of (i,j) tiles
  int use_tr;             // 1: ds_read_b64_tr_b16 fragments; 0: scalar LDS gathers (validation fallback)
  unsigned p_bytes, q_bytes;   // buffer-descriptor ranges (filled by the launcher)
};

// rows of partial stats the NT kernel writes for a given M (needed to size / finalize)
int gemm_nt_stat_rows(int M, int N);
bool gemm_nt_fused28_two_tiles(int M);   // a fused (BatchNorm-backward reduction) 28x28 dgrad of M output pixels leaves M / 392 partial rows
int gemm_nt_stat_rows_live(int M, int N, int C, int W, int ksize, int stride);   // leading rows that are not zero filler
// shapes whose conv runs on an LDS-DMA kernel that implements the output epilogue (esc / eadd / Cb2)
bool gemm_nt_conv_epilogue_ok(int W, int C, int N, int M, int ksize, int stride);
// number of splits / workspace helpers
int gemm_nt_pick_splits(int M, int N, int K);
int gemm_nt_launch(GemmNT p, int splits, hipStream_t st);
int gemm_tn_launch(GemmTN p, int splits, hipStream_t st);
// the two same-shape 3x3 / stride-1 weight gradients of a residual block on the paired nine-tap kernel (wgrad9p.hip)
bool gemm_tn_w9pair_ok(const GemmTN& a, const GemmTN& b);
int gemm_tn_w9pair_splits(const GemmTN& a);
// The split-K slabs of an EARLIER paired launch, summed by a later paired launch's workgroups beside their own work (round 4: the
// separate reduce_slabs launches cost 13 of the 66 us of a pair).  dst[l] = sum over s of slab[l][s * n + i], in the summation order of the
// stand-alone kernels (ew_reduce_slabs: ascending slabs, or the 8-lane order of its wide form), so both paths give the same bits.
struct W9PJob {
  const float* slab[2];   // [nsplit][n] each
  float* dst[2];
  size_t n;               // floats per layer (0 = no job)
  int nsplit;
};
// can a paired launch of this shape carry `job`? (its loads must fit beside the launch's own sub-images: see wgrad9p.hip)
bool gemm_tn_w9pair_job_ok(const GemmTN& a, int splits, const W9PJob& job);
int gemm_tn_launch_w9pair(GemmTN a, GemmTN b, int splits, hipStream_t st, const W9PJob* job = nullptr);
// Wo > 0: conv weight gradient on a Wo x Wo output map (lets the nine-tap kernel, wgrad9.hip, be chosen)
int gemm_tn_pick_splits(int Kp, int NI, int NJ, int C_or_0, int Wo = 0, int stride = 1);
int gemm_tn_max_splits(int Kp, int NI, int NJ, int C_or_0, int Wo, int stride);
void gemm_tn_tiles(int NI, int NJ, int C_or_0, int* TI, int* TJ);

// optional HIP-event timing of every NT/TN launch (slots: see gemm.hip)
void gemm_profile_enable(int on);
int gemm_profile_read(int slot, double* total_ms, long long* launches, double* flops);
int gemm_profile_read_bytes(int slot, double* bytes);      // algorithmic HBM bytes of the same launches
