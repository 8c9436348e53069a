// HBM-bound elementwise / column-reduction kernels: train-mode BatchNorm (stats finalize, apply,
// backward), PReLU, residual add, layout conversions, split-K slab reduction, stem conv.
// All activations are NHWC bf16 viewed as [M = N*H*W][C]; per-channel parameters are fp32.
#pragma once
#include "common.h"

// 1 / sqrt(v) in fp64 (v > 0): the hardware estimate + two Newton steps.  `1.0 / sqrt(v)` is a square-root and a divide sequence — ~70 dependent
// fp64 operations — on the critical path of EVERY BatchNorm pass (each workgroup derives its coefficients itself, bn_sliced.hip), and so were the
// divisions by the pixel count: the passes multiply by its reciprocal, formed while their loads are in flight.  Every path (sliced passes, finalize
// kernels) uses the same expressions, so fused and unfused passes still agree bit for bit.
__device__ __forceinline__ double bn_rsqrt(double v) {
  double y = __builtin_amdgcn_rsq(v);
  y = y * (1.5 - 0.5 * v * y * y);
  y = y * (1.5 - 0.5 * v * y * y);
  return y;
}

// ---- forward BN -------------------------------------------------------------------------------------
// partials: [P][2][C] (sum, sumsq).  tmp must hold 64*2*C floats when P > 1024 (two-stage reduce).
int ew_bn_finalize(const float* partials, int P, int C, double count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps,
                   float* scale, float* shift, float* save_mean, float* save_rstd, float* tmp, hipStream_t st);
// eval mode: scale/shift from running stats
int ew_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, float* scale, float* shift, hipStream_t st);

struct BnApply {
  const bf16_t* x1; const float* sc1; const float* sh1; const float* alpha;   // y = prelu?(x1*sc1+sh1)
  const bf16_t* x2; const float* sc2; const float* sh2;                       // + (x2*sc2+sh2)  (sc2 null => + x2)
  bf16_t* y;
  int M, C;
  int nchw_hw;        // >0: write y as [img][C][hw] (flatten order of the reference fc), hw = nchw_hw
  float* stats;       // [grid][2][C] partial (sum, sumsq) of y, or null
};
int ew_bn_apply_grid(int M, int C);          // number of partial rows the kernel writes
int ew_bn_apply(const BnApply& p, hipStream_t st);

// ---- backward BN (+PReLU) --------------------------------------------------------------------------
struct BnBwd {
  const bf16_t* dy;       // grad wrt (prelu?)(bn(x))
  const bf16_t* x;        // BN input (conv output)
  const float* mean; const float* rstd; const float* gamma; const float* beta;
  const float* alpha;     // non-null => PReLU after the BN
  const float *sc, *sh;   // optional: the forward pass's (scale, shift) of this BN — the PReLU mask is then sign(x sc + sh), the forward's expression
  int M, C;
  float* partials;        // [grid][3][C]
  // apply stage
  const float* coef;      // [3][C] from ew_bn_bwd_finalize: dx = a dz + A x + B
  const bf16_t* add;      // optional same-shape addend (identity-path gradient)
  const bf16_t* add_up;   // optional compact [img][H/2][W/2][C] addend placed at even (h, w)
  int H, W;               // needed for add_up
  bf16_t* dx;
  // optional: while dx is produced, also reduce it as the dy of the NEXT BatchNorm backward (the one that consumes dx; no PReLU):
  // partial sums (sum dx, sum dx * xhat_next, 0) per workgroup -> npart [ew_bn_bwd_apply_grid][3][C], saving that BN's reduce pass
  const bf16_t* nx;       // next BN's input tensor, same [M][C] shape
  const float *nmean, *nrstd;
  float* npart;
  // ... with nalpha: the next BatchNorm is followed by a PReLU (the stem's; round 3): (nsc, nsh) = its forward (scale, shift), and the rows
  // carry all three sums of ew_bn_bwd_reduce (sum dz, sum dz xhat, sum dx z over z <= 0).  Variant without own PReLU / same-shape addend only.
  const float *nsc, *nsh, *nalpha;
};
int ew_bn_bwd_apply_grid(int M, int C);
int ew_bn_bwd_grid(int M, int C);
int ew_bn_bwd_reduce(const BnBwd& p, hipStream_t st);
int ew_bn_bwd_finalize(const float* partials, int P, int C, double count, const float* gamma, const float* mean, const float* rstd,
                       float* dgamma, float* dbeta, float* dalpha, float* coef, hipStream_t st);
int ew_bn_bwd_apply(const BnBwd& p, hipStream_t st);

// ---- channel-sliced BatchNorm passes that reduce their partial rows themselves: no finalize launch (bn_sliced.hip) -------------
// partial rows everywhere: [row][statistic][C] fp32, the layout the kernels above and the conv epilogues write
struct BnApplyS {
  const float* part; int P;                       // statistics of x1: P rows [2][C] (sum, sumsq)
  double count; const float *gamma, *beta; float *rm, *rv; float momentum, eps;
  float *scale, *shift, *mean, *rstd;             // [C] each, written (what ew_bn_finalize leaves)
  const bf16_t* x1; const float* alpha;           // y = prelu?(bn(x1)) ...
  const bf16_t* x2;                               // ... + x2 (identity path), or null
  bf16_t* y;
  int M, C;
  float* stats;                                   // statistics of y: ew_bn_sliced_rows(M, C) rows [2][C], or null; must not alias `part`
  int G, ppg;                                     // set by the launcher
};
// bn_apply2 (round 3): out = bn3(x1) + x2 AND y2 = bnN(out) in one pass, both BatchNorms in training mode, from the RAW moments the conv that
// produced x1 left (rows [3][C] of sum x1, sum x1 * x2, sum x1 * x1; GemmNT::bmom) and the known statistics (xmean, xrstd) of x2:
//   mean(out) = sc3 mean(x1) + sh3 + mean(x2),  var(out) = sc3^2 var(x1) + var(x2) + 2 sc3 cov(x1, x2)
// so the next block's bn1 needs no pass over `out` (the rounding of `out` to bf16 is the only thing these moments do not see: ~1e-6 of the variance).
struct BnApply2S {
  const float* part; int P;                       // P rows [3][C]
  double count; float momentum, eps;
  const float *gamma, *beta; float *rm, *rv; float *scale, *shift, *mean, *rstd;              // the BatchNorm of x1 (bn3)
  const float *xmean, *xrstd;                     // saved statistics of x2 (this block's bn1)
  const float *ngamma, *nbeta; float *nrm, *nrv; float *nscale, *nshift, *nmean, *nrstd;      // the BatchNorm of out (the next block's bn1)
  const bf16_t *x1, *x2;
  bf16_t *y, *y2;
  int M, C;
  int G, ppg;                                     // set by the launcher
};
bool ew_bn_apply2_sliced_ok(int M, int C, int P);
int ew_bn_apply2_sliced(BnApply2S p, hipStream_t st);
struct BnBwdS {
  const bf16_t *dy, *x;
  const float *mean, *rstd, *gamma, *alpha;       // alpha non-null => PReLU after the BN
  const float *sc, *sh;                           // the forward's (scale, shift): required with alpha (PReLU mask = sign(x sc + sh))
  int M, C;
  float* partials;                                // reduce pass: ew_bn_sliced_rows(M, C) rows [3][C] written
  const float* part_in; int P; double count;      // apply pass: the P rows [3][C] to reduce
  float *dgamma, *dbeta, *dalpha;                 // apply pass: parameter gradients written (null: skipped)
  const bf16_t* add;                              // optional same-shape addend (identity-path gradient)
  bf16_t* dx;
  const bf16_t* nx; const float *nmean, *nrstd;   // optional: dx reduced as the dy of the BatchNorm that consumes it ...
  float* npart;                                   // ... into ew_bn_sliced_rows(M, C) rows [3][C]; must not alias `part_in`
  int G, ppg;
};
int ew_bn_sliced_rows(int M, int C, bool backward = false);
bool ew_bn_sliced_ok(int M, int C, int P_in, bool backward);
int ew_bn_apply_sliced(BnApplyS p, hipStream_t st);
int ew_bn_bwd_reduce_sliced(BnBwdS p, hipStream_t st);
int ew_bn_bwd_apply_sliced(BnBwdS p, hipStream_t st);

// ---- BatchNorm1d on fp32 [B][C] (the `features` layer) ------------------------------------------------
int ew_bn1d_fwd(const float* x, float* y, int B, int C, const float* gamma, const float* beta,
                float* running_mean, float* running_var, float momentum, float eps, int training,
                float* save_mean, float* save_rstd, hipStream_t st);
// dx = bn1d backward; also dbeta (bias grad), and colsum(dx) -> fc bias grad
int ew_bn1d_bwd(const float* dy, const float* x, float* dx, int B, int C, const float* gamma,
                const float* save_mean, const float* save_rstd, float* dbeta, float* dx_colsum,
                bf16_t* dx_bf16, bf16_t* dx_bf16_t, int ldt, hipStream_t st, int frozen = 0);   // frozen: eval-mode BN inside a training net (dx = g rstd dy)

// ---- split-K slab reductions ---------------------------------------------------------------------------
int ew_reduce_slabs(float* dst, const float* slabs, int nsplit, size_t n, const float* bias, int bias_n,
                    hipStream_t st);
int ew_reduce_slabs2(float* dst0, const float* slabs0, float* dst1, const float* slabs1, int nsplit, size_t n, hipStream_t st);
int ew_reduce_slabs_bf16(bf16_t* dst, const float* slabs, int nsplit, size_t n, hipStream_t st);

// ---- conversions -----------------------------------------------------------------------------------------------
int ew_cast_f32_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t st);
// KRSC fp32 [Cout][R][S][Cin] -> dgrad shadow bf16 [Cin][R][S][Cout] with (r,s) flipped
int ew_weight_dgrad_shadow(const float* w, bf16_t* dst, int Cout, int R, int S, int Cin, hipStream_t st);
// fp32 [B][C][HW] -> bf16 [B][HW][C]
int ew_nchw_f32_to_nhwc_bf16(const float* src, bf16_t* dst, int B, int C, int HW, hipStream_t st);
// bf16 [R][Ccols] -> bf16 [Ccols][R]
int ew_transpose_bf16(const bf16_t* src, bf16_t* dst, int R, int Ccols, hipStream_t st);

// ---- stem: conv3x3(3->64, s1, p1) on fp32 NCHW input ----------------------------------------------------------
// w: KRSC fp32 [64][3][3][3]; y: NHWC bf16 [B][H][W][64]; stats partials [stem_stat_rows][2][64]
int ew_stem_stat_rows(int B, int H, int W);
int ew_stem_fwd(const float* x, const float* w, bf16_t* y, float* stats, int B, int H, int W, hipStream_t st);
// dw (KRSC fp32 [64][3][3][3]) = sum over pixels; tmp holds stem_wgrad_blocks*64*32 floats
int ew_stem_wgrad_blocks(int B, int H, int W);
// x0 != nullptr: dy is the gradient wrt the stem's activation and the kernel applies the BatchNorm + PReLU backward (coef [3][64] of
// bn_bwd_finalize, the forward's scale / shift, the slopes) to the tile on its way in, bit-identical to bn_bwd_apply + the plain form
int ew_stem_wgrad(const float* x, const bf16_t* dy, float* dw, float* tmp, int B, int H, int W, hipStream_t st, const bf16_t* x0 = nullptr,
                  const float* coef = nullptr, const float* sc = nullptr, const float* sh = nullptr, const float* alpha = nullptr);
int ew_preprocess_u8(const unsigned char* src, const unsigned char* flip, float* dst, int B, int H, int W, hipStream_t st);
int ew_bias_prelu_bwd(const bf16_t* dy, const bf16_t* x, const float* bias, const float* alpha, int M, int C, float* partials,
                      float* coef, float* dbias, float* dalpha, const bf16_t* add, bf16_t* dx, hipStream_t st);
// PReLU(+bias) backward as ONE pass: g = dy (+ add, then written to gsum), dz = g * prelu'(x + bias) -> dz; rows [ew_prelu_bwd_rows][2][C] =
// (sum dz | sum g (x + bias) over x + bias <= 0) for ew_prelu_bwd_finalize (-> dbias, dalpha; either may be null)
int ew_prelu_bwd_rows(int M, int C);
int ew_prelu_bwd_pass(const bf16_t* dy, const bf16_t* add, const bf16_t* x, const float* bias, const float* alpha, int M, int C, bf16_t* gsum,
                      bf16_t* dz, float* rows, hipStream_t st);
int ew_prelu_bwd_finalize(const float* rows, int P, int C, float* dbias, float* dalpha, hipStream_t st);
// ... of up to kMaxPreluFin PReLUs in ONE launch (sphnet: 62 per step, 8.7 us each on the weight-gradient stream that bounds its backward pass)
constexpr int kMaxPreluFin = 64;
struct PreluFinEntry { unsigned long long rows_off; long long dbias_off, dalpha_off; int P, C, blk0, nv_row; };   // byte offset into `base`; float offsets into `grads` (dbias < 0: none); nv_row: statistics per row (2: the pass's rows, 3: a fused dgrad epilogue's, of which the first two are read)
struct PreluFinTable { int n, blocks; PreluFinEntry e[kMaxPreluFin]; };
int ew_prelu_bwd_finalize_multi(const unsigned char* base, float* grads, const PreluFinTable& t, hipStream_t st);
int ew_pad_input_nhwc(const float* src, bf16_t* dst, int B, int C, int HW, int Cpad, hipStream_t st);
// all dgrad shadows of a network in one launch (one 64x64 transpose tile per workgroup, table passed by value)
constexpr int kMaxShadowEntries = 112;     // 112 x 32 B: the by-value table stays under the 4 KiB kernel-argument limit
struct ShadowEntry { unsigned long long src, dst; unsigned short cout64, cin64, rs, pad; unsigned first_blk; };   // offsets in elements
struct ShadowTable { int n; int pad; ShadowEntry e[kMaxShadowEntries]; };
int ew_weight_dgrad_shadow_multi(const float* params, bf16_t* shadow, ShadowTable& t, hipStream_t st);
// eval-mode (scale, shift) of many BatchNorms in one launch: offsets (fp32 elements) into the flat parameter / buffer / saved-statistics
// tensors; g_off < 0: weight = 1 (by-value table, one launch per table-full)
constexpr int kMaxBnEvalEntries = 160;      // 160 x 24 B stays under the 4 KiB kernel-argument limit
struct BnEvalEntry { int C, g_off, b_off, rm_off, rv_off, save_off; };
struct BnEvalTable { int n; float eps; BnEvalEntry e[kMaxBnEvalEntries]; };
int ew_bn_eval_coeffs_multi(const float* params, const float* bufs, float* save, const BnEvalTable& t, hipStream_t st);

// nn.Dropout(p, inplace=True) on the flattened bn2 output (iresnet.py:96,169): counter-based mask (seed, step, index), kept values scaled by
// 1 / (1 - p); mask: one byte per element, kept for the backward pass (dx: fp32, same element order)
int ew_dropout_fwd(bf16_t* t, unsigned char* mask, size_t n, float p, unsigned long long seed, unsigned long long step, hipStream_t st);
int ew_dropout_bwd(float* dx, const unsigned char* mask, size_t n, float p, hipStream_t st);
