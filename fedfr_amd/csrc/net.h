// Host-side plan for an iresnet (reference: backbones/iresnet.py:60-172).  The plan owns NO device memory:
// it only computes offsets into caller-provided flat buffers and sequences kernel launches on the
// caller's stream.  Buffers (all allocated by the host framework, e.g. torch tensors):
//   params  fp32  [param_count]      trainable region first (state_dict order), then frozen features.weight
//                                    conv weights are KRSC ([Cout][kh][kw][Cin]) == channels_last OIHW views
//   grads   fp32  [trainable_count]  same layout as the trainable region
//   bufs    fp32  [buffer_count]     BN running_mean / running_var
//   shadow  bf16  [shadow_count]     [0, trainable_count): bf16 mirror of params; then dgrad-layout conv copies
//   act     bytes [act_bytes]        saved activations (NHWC bf16) + per-BN saved (scale, shift, mean, rstd)
//   ws      bytes [ws_bytes]         gradient ping-pong buffers, split-K slabs, reduction partials
#pragma once
#include <string>
#include <vector>
#include "common.h"

struct NetTensor {          // one state_dict entry
  std::string name;
  int kind;                 // 0 conv(KRSC) 1 bn_w 2 bn_b 3 prelu 4 fc_w 5 fc_b 6 running_mean 7 running_var 8 num_batches_tracked
  int region;               // 0 params, 1 bufs, 2 nbt (index = offset)
  long long offset;         // element offset inside the region
  int shape[4];             // reference (OIHW / 1-D / 2-D) shape, unused dims = 0
  int ndim;
};

struct ConvD {
  int Cin, Cout, R, stride, Hin, Hout;
  long long w_off;          // params / grads / fwd shadow offset
  long long wd_off;         // dgrad shadow offset (absolute in shadow buffer)
};
struct BnD {
  int C;
  long long g_off, b_off;   // params
  long long rm_off, rv_off; // bufs
  long long save_off;       // float offset in the act "bnsave" region: scale, shift, mean, rstd (4*C floats)
};
struct BlockD {
  int Cin, Cout, Hin, Hout, stride;
  bool has_ds;
  BnD bn1, bn2, bn3, bnds;
  ConvD conv1, conv2, ds;
  long long alpha_off;
  long long x_off, a1_off, c1_off, a2_off, c2_off, d_off, out_off;   // bf16 element offsets in act
};

// ---- sphnet (reference backbones/sphnet.py:4-73): conv3x3 (+bias on the stride-2 stage heads) -> PReLU units, residual blocks of two ----
struct SphUnit {
  ConvD conv;                 // as the kernels see it (the first conv's 3 input channels are zero-padded to 64)
  long long w_param_off;      // the parameter's own offset ([Cout][3][3][Cin_real] KRSC)
  int cin_real;
  long long b_off, a_off;     // conv bias (-1: none), PReLU slope
  long long in_off, c_off, t_off;   // bf16 arena offsets: input, raw conv output, activated output
  size_t dz_off, rows_off;    // workspace byte offsets: gradient wrt the raw conv output (the weight-gradient operand); PReLU partial rows
};
struct SphBlockD { SphUnit u1, u2; };
struct SphStageD { SphUnit head; std::vector<SphBlockD> blocks; int C, H; };

// dual-stream backward: generations of (dc2, dc1, dd) the weight-gradient stream may lag behind the main stream.  With 2 the main
// stream waited 1.4 ms per step for weight gradients of two blocks ago (mostly in the 7x7 / early 14x14 stages, whose weight GEMMs
// are long and whose main-stream kernels are short); 4 generations cost 1.2 GB more workspace.
#ifndef FEDFR_WGRAD_DEPTH
#define FEDFR_WGRAD_DEPTH 4
#endif
constexpr int kWgradDepth = FEDFR_WGRAD_DEPTH;
constexpr int kSlabRegions = 4;         // split-K slab regions of `slab_floats` each in a plan's workspace: a paired weight-gradient launch writes two, and the
                                        // NEXT pair writes the other two while it sums these (wgrad9p.hip, W9PJob); every plan layout allocates all of them
constexpr int kSlicedRowsMax = 256;     // partial rows a channel-sliced BatchNorm pass writes at most (ew_bn_sliced_rows)
struct FedfrNet {
  int layers[4];
  int B, Bp, HW, F;                     // batch, batch padded to 8, input side, feature dim
  std::vector<NetTensor> tensors;
  std::vector<BlockD> blocks;
  ConvD stem;
  BnD stem_bn, bn2, feat_bn;
  long long stem_alpha_off;
  long long fc_w_off, fc_b_off;
  long long c0_off, a0_off, t_off;      // stem conv out, stem act, flattened bn2 output [B][C*hw] (bf16, NCHW order)
  long long yfc_off, feat_save_off;     // float offsets (act float region): fc output [B][F]; features mean/rstd
  long long param_count, trainable_count, buffer_count, nbt_count, shadow_count;
  long long act_bf16_count, act_float_off_bytes, act_bytes;
  // workspace layout (byte offsets)
  size_t ws_bytes;
  size_t ws_g[2], ws_t[6], ws_t2[3 * (kWgradDepth - 1)], ws_part, ws_part2, ws_slab, ws_small, ws_fc, ws_stem = 0;   // ws_t2: further copies of t0/t2/t4 (dual-stream backward)
  mutable std::vector<hipEvent_t> events;                                  // fork/join events of the dual-stream backward (host objects)
  size_t g_elems, part_floats, slab_floats;
  int final_hw, final_C, fc_in;
  // nn.Dropout between bn2 and fc (iresnet.py:96,169): p = 0 off; the mask of the last training forward lives in the arena
  float dropout_p = 0.f;
  unsigned long long dropout_seed = 100;
  mutable unsigned long long dropout_step = 0;      // counts training forwards (the mask is a function of (seed, step, index))
  long long mask_off_bytes = -1;                    // byte offset of the mask [B * fc_in] inside `act`
  // nn.BatchNorm modules put into eval() inside a training net (IResNet.freeze_BN(test_mode=True), iresnet.py:140-147): set by a forward
  // pass with training = 2 (running statistics normalise, nothing is updated, activations are kept), read by the backward pass
  mutable bool bn_frozen = false;
  int sph_type = 0;                     // 20 / 64: a sphnet plan (net_create_sphere); the iresnet fields above are unused then
  std::vector<SphStageD> sph;
  long long sph_xin_off = 0, sph_flat_off = 0;          // arena: padded NHWC input [B][112][112][64]; (== last activation) NCHW-flat [B][512*49]
  long long sph_w0pad_off = 0;                          // shadow: the first conv's weights padded to 64 input channels [64][3][3][64]
  size_t sph_ws_g[4] = {0, 0, 0, 0}, sph_ws_w0g = 0, sph_ws_dfe = 0;   // workspace: gradient ping-pong + two dgrad outputs; padded stem weight gradient; bf16 dfeats + transpose
  bool block_only = false;              // plan of a lone IBasicBlock (net_create_block): no stem, no bn2/fc/features tail
  long long dx_off = -1;                // block_only: bf16 arena offset of the gradient wrt the block input [B*Hin*Hin][Cin]
};

FedfrNet* net_create(const int layers[4], int batch, int in_hw, int num_features);
// lone IBasicBlock(cin, cout, stride) on a hin x hin map; with it net_forward takes x = fp32 NCHW [B][cin][hin][hin] (feats unused, may be
// null) and net_backward takes dfeats = fp32 NCHW [B][cout][hout][hout]; y / dx are read from the arena (fedfr_net_act_info)
FedfrNet* net_create_block(int cin, int cout, int stride, int hin, int batch);
// sphnet (type 20 / 64, 112 x 112 input, 512 features): same buffers and entry points as an iresnet plan (no BatchNorm: bufs / nbt are empty);
// the optimiser update is not folded into its backward pass (NetSgd::done_from = trainable_count)
FedfrNet* net_create_sphere(int type, int batch);
int net_prepare_weights(const FedfrNet* n, const float* params, bf16_t* shadow, int fwd_shadow_too, hipStream_t st);
int net_forward(const FedfrNet* n, const float* x, const float* params, float* bufs, const bf16_t* shadow,
                unsigned char* act, unsigned char* ws, float* feats, int training, hipStream_t st);
// aux == nullptr: everything on `st`.  Otherwise all weight-gradient GEMMs (off the dgrad -> BN critical path) run on
// `aux`, forked/joined with events; on return both streams' work is ordered before anything enqueued later on `st`.
// SGD folded into the backward pass: parameter ranges whose gradients are final (a whole stage, or the bn2 / fc / features tail) are
// updated on the weight-gradient stream while the main stream is still walking the earlier stages.  `done_from` (out): parameters
// [done_from, trainable_count) have been updated when the call's work completes; the caller updates [0, done_from).
struct NetSgd {
  float* params; bf16_t* shadow; float* mom;      // writable views of the (same) parameter / bf16 mirror buffers, momentum buffer
  float lr, mu, wd; int first;
  long long done_from;
  float gscale = 1.f;                             // 1 / loss scale of the incoming gradient (fp16-storage build); the update kernels undo it
  unsigned* overflow = nullptr;                   // device word the update kernels set when they skipped a non-finite gradient element
};
int net_backward(const FedfrNet* n, const float* x, const float* dfeats, const float* params, const bf16_t* shadow,
                 unsigned char* act, unsigned char* ws, float* grads, hipStream_t st, hipStream_t aux, NetSgd* sgd = nullptr);

// fp32 validation path (net_f32.hip): the same plan with fp32 activations and exact-fp32 arithmetic (im2col + fp32-MFMA GEMM, two-pass
// BatchNorm in fp64) — slow by design, whole-network plans, dropout 0.  arena: net_f32_arena_floats(n) floats (activations kept for the
// backward pass), ws: net_f32_ws_floats(n) floats.  Semantics of net_forward / net_backward (training 0 / 1).
size_t net_f32_arena_floats(const FedfrNet* n);
size_t net_f32_ws_floats(const FedfrNet* n);
int net_f32_forward(const FedfrNet* n, const float* x, const float* params, float* bufs, float* arena, float* ws, float* feats, int training,
                    hipStream_t st);
int net_f32_backward(const FedfrNet* n, const float* dfeats, const float* params, float* arena, float* ws, float* grads, hipStream_t st);
