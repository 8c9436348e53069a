// iresnet plan + forward/backward sequencing (host code only; see net.h).
// Reference behaviour: backbones/iresnet.py:46-57 (IBasicBlock.forward), :158-172 (IResNet.forward);
// backward is the hand-derived adjoint of that graph (checked against autograd of the CPU oracle).
#include "net.h"
#include "optim.h"
#include <algorithm>
#include <cstdio>
#include "ew.h"
#include "gemm.h"
#include "head.h"

static const int kPlanes[4] = {64, 128, 256, 512};
static const float kBnEps = 1e-5f, kBnMomentum = 0.1f;
int g_tn_use_tr = 1;   // option "tn_use_tr": 0 selects the scalar-LDS fallback fragments (validation)

namespace {
struct Builder {
  FedfrNet* n;
  long long poff = 0, boff = 0, nbt = 0, saveoff = 0;
  std::vector<NetTensor> frozen;
  long long add_param(const std::string& name, int kind, std::initializer_list<int> shape) {
    NetTensor t;
    t.name = name; t.kind = kind; t.region = 0; t.offset = poff; t.ndim = (int)shape.size();
    long long numel = 1; int i = 0;
    for (int s : shape) { t.shape[i++] = s; numel *= s; }
    for (; i < 4; ++i) t.shape[i] = 0;
    n->tensors.push_back(t);
    const long long o = poff;
    poff += (numel + 3) / 4 * 4;
    return o;
  }
  BnD add_bn(const std::string& p, int C, bool frozen_weight = false) {
    BnD b;
    b.C = C;
    if (frozen_weight) {
      // features.weight: requires_grad False (iresnet.py:99-100) -> lives after the trainable region
      NetTensor t; t.name = p + ".weight"; t.kind = 1; t.region = 0; t.offset = -1; t.ndim = 1;
      t.shape[0] = C; t.shape[1] = t.shape[2] = t.shape[3] = 0;
      n->tensors.push_back(t);
      b.g_off = -1;
    } else {
      b.g_off = add_param(p + ".weight", 1, {C});
    }
    b.b_off = add_param(p + ".bias", 2, {C});
    NetTensor t; t.ndim = 1; t.shape[0] = C; t.shape[1] = t.shape[2] = t.shape[3] = 0; t.region = 1;
    t.name = p + ".running_mean"; t.kind = 6; t.offset = boff; b.rm_off = boff; boff += C; n->tensors.push_back(t);
    t.name = p + ".running_var"; t.kind = 7; t.offset = boff; b.rv_off = boff; boff += C; n->tensors.push_back(t);
    t.name = p + ".num_batches_tracked"; t.kind = 8; t.region = 2; t.offset = nbt++; t.ndim = 0; t.shape[0] = 0; n->tensors.push_back(t);
    b.save_off = saveoff;
    saveoff += 4 * (long long)C;
    return b;
  }
  ConvD add_conv(const std::string& name, int Cin, int Cout, int R, int stride, int Hin) {
    ConvD c;
    c.Cin = Cin; c.Cout = Cout; c.R = R; c.stride = stride; c.Hin = Hin; c.Hout = Hin / stride;
    c.w_off = add_param(name + ".weight", 0, {Cout, Cin, R, R});
    c.wd_off = -1;
    return c;
  }
  // p: "layerL.i." inside a network, "" for a lone block (reference IBasicBlock attribute order, iresnet.py:37-43)
  BlockD add_block(const std::string& p, int cin, int cout, int stride, int Hin, bool ds) {
    BlockD k;
    k.Cin = cin; k.Cout = cout; k.stride = stride; k.Hin = Hin; k.Hout = Hin / stride; k.has_ds = ds;
    k.bn1 = add_bn(p + "bn1", k.Cin);
    k.conv1 = add_conv(p + "conv1", k.Cin, k.Cout, 3, 1, k.Hin);
    k.bn2 = add_bn(p + "bn2", k.Cout);
    k.alpha_off = add_param(p + "prelu.weight", 3, {k.Cout});
    k.conv2 = add_conv(p + "conv2", k.Cout, k.Cout, 3, k.stride, k.Hin);
    k.bn3 = add_bn(p + "bn3", k.Cout);
    if (k.has_ds) {
      k.ds = add_conv(p + "downsample.0", k.Cin, k.Cout, 1, k.stride, k.Hin);
      k.bnds = add_bn(p + "downsample.1", k.Cout);
    }
    return k;
  }
};
}  // namespace

// shadow offsets, activation arena and workspace layout of a plan whose tensor list and block list are complete
static void plan_layout(FedfrNet* n, Builder& b, int in_hw) {
  const int batch = n->B, num_features = n->F;
  n->trainable_count = b.poff;
  for (auto& t : n->tensors)
    if (t.region == 0 && t.offset < 0) { t.offset = b.poff; n->feat_bn.g_off = b.poff; b.poff += (t.shape[0] + 3) / 4 * 4; }
  n->param_count = b.poff;
  n->buffer_count = b.boff;
  n->nbt_count = b.nbt;
  // ---- shadow: bf16 mirror of the trainable region, then dgrad-layout copies of every block conv ----
  long long soff = (n->trainable_count + 7) / 8 * 8;
  for (auto& k : n->blocks) {
    k.conv1.wd_off = soff; soff += (long long)k.conv1.Cin * 9 * k.conv1.Cout;
    k.conv2.wd_off = soff; soff += (long long)k.conv2.Cin * 9 * k.conv2.Cout;
    if (k.has_ds) { k.ds.wd_off = soff; soff += (long long)k.ds.Cin * k.ds.Cout; }
  }
  n->shadow_count = soff;
  // ---- activation arena (bf16 elements) ----
  const long long Bq = batch;
  long long aoff = 0;
  auto take = [&](long long elems) { const long long o = aoff; aoff += (elems + 7) / 8 * 8; return o; };
  const long long M0 = Bq * in_hw * in_hw;
  long long prev, gmax;
  if (n->block_only) {                      // a lone IBasicBlock: its input tensor and the gradient wrt it live in the arena too
    const BlockD& k0 = n->blocks.front();
    prev = take(M0 * k0.Cin);
    n->dx_off = take(M0 * k0.Cin);
    n->c0_off = n->a0_off = -1;
    gmax = M0 * k0.Cin;
  } else {
    n->c0_off = take(M0 * 64);
    n->a0_off = take(M0 * 64);
    prev = n->a0_off;
    gmax = M0 * 64;
  }
  for (auto& k : n->blocks) {
    const long long Mi = Bq * k.Hin * k.Hin, Mo = Bq * k.Hout * k.Hout;
    k.x_off = prev;
    k.a1_off = take(Mi * k.Cin);
    k.c1_off = take(Mi * k.Cout);
    k.a2_off = take(Mi * k.Cout);
    k.c2_off = take(Mo * k.Cout);
    k.d_off = k.has_ds ? take(Mo * k.Cout) : -1;
    k.out_off = take(Mo * k.Cout);
    prev = k.out_off;
    gmax = std::max(gmax, std::max(Mi * k.Cin, Mi * k.Cout));
  }
  n->t_off = take(Bq * n->fc_in);
  n->act_bf16_count = aoff;
  n->act_float_off_bytes = (long long)align_up((size_t)aoff * 2, 256);
  long long foff = b.saveoff;                 // bnsave floats first
  n->yfc_off = foff; foff += Bq * num_features;
  n->feat_save_off = foff; foff += 2 * (long long)num_features;
  n->mask_off_bytes = n->act_float_off_bytes + (long long)align_up((size_t)foff * 4, 256);
  n->act_bytes = n->mask_off_bytes + (long long)align_up((size_t)Bq * n->fc_in, 256);
  // ---- workspace ----
  n->g_elems = (size_t)gmax;
  long long part = (long long)ew_stem_stat_rows(batch, in_hw, in_hw) * 128;
  long long slab = (long long)ew_stem_wgrad_blocks(batch, in_hw, in_hw) * 2048;
  auto upd_part = [&](long long M, int C) {
    part = std::max(part, (long long)ew_bn_apply_grid((int)M, C) * 2 * C);
    part = std::max(part, (long long)ew_bn_bwd_grid((int)M, C) * 3 * C);
    part = std::max(part, (long long)ew_bn_bwd_apply_grid((int)M, C) * 3 * C);
  };
  auto upd_conv = [&](const ConvD& c) {
    const long long Mo = Bq * c.Hout * c.Hout;
    part = std::max(part, (long long)gemm_nt_stat_rows((int)Mo, c.Cout) * 2 * c.Cout);
    const int NJ = c.R * c.R * c.Cin;
    slab = std::max(slab, (long long)gemm_tn_max_splits((int)Mo, c.Cout, NJ, c.Cin, c.Hout, c.stride) * c.Cout * NJ);
  };
  upd_part(M0, 64);
  int H = in_hw;
  for (auto& k : n->blocks) {
    const long long Mi = Bq * k.Hin * k.Hin, Mo = Bq * k.Hout * k.Hout;
    upd_part(Mi, k.Cin); upd_part(Mi, k.Cout); upd_part(Mo, k.Cout);
    upd_conv(k.conv1); upd_conv(k.conv2);
    if (k.has_ds) upd_conv(k.ds);
    H = k.Hout;
  }
  upd_part(Bq * H * H, 512);
  if (n->fc_in > 0)
    slab = std::max(slab, (long long)gemm_nt_pick_splits(batch, num_features, n->fc_in) * Bq * num_features);
  n->part_floats = (size_t)part;
  n->slab_floats = (size_t)slab;
  size_t w = 0;
  auto wtake = [&](size_t bytes) { const size_t o = w; w += align_up(bytes, 256); return o; };
  for (int i = 0; i < 2; ++i) n->ws_g[i] = wtake(n->g_elems * 2);
  for (int i = 0; i < 6; ++i) n->ws_t[i] = wtake(n->g_elems * 2);
  for (int i = 0; i < 3 * (kWgradDepth - 1); ++i) n->ws_t2[i] = wtake(n->g_elems * 2);
  n->ws_part = wtake(n->part_floats * 4);
  n->ws_part2 = wtake((size_t)kSlicedRowsMax * 3 * 1024 * 4);   // rows written by the channel-sliced BatchNorm passes (bn_sliced.hip)
  n->ws_slab = wtake(n->slab_floats * 4 * kSlabRegions);      // four regions: paired weight-gradient GEMMs write their two slab sets side by side, and the NEXT pair
                                                   // writes the other two while it sums this pair's (wgrad9p.hip, W9PJob)
  // small: coef[3*512] | finalize tmp [64*2*512] | dyfc f32 [B*F] | dyb bf16 [B*F] | dybt bf16 [F*Bp]
  n->ws_small = wtake((size_t)(3 * 512 + 64 * 2 * 512) * 4 + (size_t)Bq * num_features * 4 + (size_t)Bq * num_features * 2 +
                      (size_t)num_features * n->Bp * 2 + 1024);
  n->ws_fc = wtake((size_t)n->Bp * n->fc_in * 4);
  n->ws_stem = wtake((size_t)ew_stem_wgrad_blocks(batch, in_hw, in_hw) * 2048 * 4);     // the stem weight gradient's own partials: it runs on the main stream
                                                                                        // while the weight-gradient stream is still in the slab regions
  n->ws_bytes = w;
}

FedfrNet* net_create(const int layers[4], int batch, int in_hw, int num_features) {
  if (batch <= 0 || in_hw <= 0 || (in_hw % 16) != 0 || num_features <= 0 || (num_features % 64) != 0) {
    fedfr_set_error("net_create: need batch>0, in_hw%%16==0, num_features%%64==0 (got %d, %d, %d)", batch, in_hw, num_features);
    return nullptr;
  }
  for (int i = 0; i < 4; ++i)
    if (layers[i] <= 0) { fedfr_set_error("net_create: layers[%d]=%d", i, layers[i]); return nullptr; }
  FedfrNet* n = new FedfrNet();
  for (int i = 0; i < 4; ++i) n->layers[i] = layers[i];
  n->B = batch; n->Bp = (batch + 7) / 8 * 8; n->HW = in_hw; n->F = num_features;
  Builder b; b.n = n;
  // ---- tensors in reference state_dict order (iresnet.py:76-98, :37-43) ----
  n->stem = b.add_conv("conv1", 3, 64, 3, 1, in_hw);
  n->stem_bn = b.add_bn("bn1", 64);
  n->stem_alpha_off = b.add_param("prelu.weight", 3, {64});
  int inpl = 64, H = in_hw;
  for (int s = 0; s < 4; ++s) {
    for (int i = 0; i < layers[s]; ++i) {
      char pfx[64];
      snprintf(pfx, sizeof pfx, "layer%d.%d.", s + 1, i);
      BlockD k = b.add_block(pfx, (i == 0) ? inpl : kPlanes[s], kPlanes[s], (i == 0) ? 2 : 1, H, i == 0);
      n->blocks.push_back(k);
      H = k.Hout;
    }
    inpl = kPlanes[s];
  }
  n->bn2 = b.add_bn("bn2", 512);
  n->final_hw = H; n->final_C = 512; n->fc_in = 512 * H * H;
  n->fc_w_off = b.add_param("fc.weight", 4, {num_features, n->fc_in});
  n->fc_b_off = b.add_param("fc.bias", 5, {num_features});
  n->feat_bn = b.add_bn("features", num_features, /*frozen_weight=*/true);
  plan_layout(n, b, in_hw);
  return n;
}

// A lone IBasicBlock (reference backbones/iresnet.py:28-57) as a plan of its own: same block code as inside a network (net_forward /
// net_backward run their block loop over this one block and skip stem and tail), tensors in the reference block's state_dict order.
// Used by the block-level parity tests (tests/golden/block.npz) — x / dy / y / dx cross the ABI as fp32 NCHW like the reference's.
FedfrNet* net_create_block(int cin, int cout, int stride, int hin, int batch) {
  const bool ds = stride != 1;
  if (batch <= 0 || hin <= 0 || (hin % stride) != 0 || cin <= 0 || cout <= 0 || (cin % 64) || (cout % 64) || (stride != 1 && stride != 2) ||
      (stride == 1 && cin != cout)) {
    fedfr_set_error("block_create: need channels%%64==0, stride 1 (cin==cout) or 2 (with downsample), even map (got %d->%d s%d @%d)", cin, cout,
                    stride, hin);
    return nullptr;
  }
  FedfrNet* n = new FedfrNet();
  for (int i = 0; i < 4; ++i) n->layers[i] = 0;
  n->block_only = true;
  n->B = batch; n->Bp = (batch + 7) / 8 * 8; n->HW = hin; n->F = 64;
  Builder b; b.n = n;
  n->blocks.push_back(b.add_block("", cin, cout, stride, hin, ds));
  n->final_hw = n->blocks[0].Hout; n->final_C = cout; n->fc_in = 0;
  n->fc_w_off = n->fc_b_off = -1;
  plan_layout(n, b, hin);
  return n;
}

// ---------------------------------------------------------------------------------------------------------
struct Ctx {
  const FedfrNet* n;
  const float* params; float* bufs; const bf16_t* shadow; bf16_t* actb; float* actf; unsigned char* ws; float* grads;
  hipStream_t st;
  float* part() const { return reinterpret_cast<float*>(ws + n->ws_part); }
  // second partial-row buffer: a channel-sliced pass reduces the rows it is handed while (other workgroups of the same launch) write the
  // rows of its own output, so those go to whichever buffer it does not read
  float* part2() const { return reinterpret_cast<float*>(ws + n->ws_part2); }
  float* part_other(const float* in) const { return in == part2() ? part() : part2(); }
  float* slab(int which = 0) const { return reinterpret_cast<float*>(ws + n->ws_slab) + (size_t)which * n->slab_floats; }
  float* coef() const { return reinterpret_cast<float*>(ws + n->ws_small); }
  float* ftmp() const { return coef() + 3 * 512; }
  float* dyfc() const { return ftmp() + 64 * 2 * 512; }
  bf16_t* dyb() const { return reinterpret_cast<bf16_t*>(dyfc() + (size_t)n->B * n->F); }
  bf16_t* dybt() const { return dyb() + (size_t)n->B * n->F; }
  bf16_t* g(int i) const { return reinterpret_cast<bf16_t*>(ws + n->ws_g[i]); }
  bf16_t* t(int i) const { return reinterpret_cast<bf16_t*>(ws + n->ws_t[i]); }
  // tensors read by the weight-gradient GEMMs (dc2 = slot 0, dc1 = slot 1, dd = slot 2), double buffered per block parity
  bf16_t* tw(int slot, int par) const { return reinterpret_cast<bf16_t*>(ws + (par ? n->ws_t2[(par - 1) * 3 + slot] : n->ws_t[slot * 2])); }
  float* save(const BnD& b, int which) const { return actf + b.save_off + (long long)which * b.C; }   // 0 scale 1 shift 2 mean 3 rstd
  const float* gamma(const BnD& b) const { return params + b.g_off; }
  const float* beta(const BnD& b) const { return params + b.b_off; }
};

static int conv_fwd(const Ctx& c, const ConvD& cv, const bf16_t* in, bf16_t* out, bool stats) {
  GemmNT p{};
  p.A = in; p.B = c.shadow + cv.w_off;
  p.M = c.n->B * cv.Hout * cv.Hout; p.N = cv.Cout; p.K = cv.R * cv.R * cv.Cin;
  p.mode = 1; p.H = cv.Hin; p.W = cv.Hin; p.C = cv.Cin; p.Ho = cv.Hout; p.Wo = cv.Hout; p.S = cv.R;
  p.stride = cv.stride; p.pad = (cv.R == 3) ? 1 : 0; p.up = 1;
  p.Cb = out; p.ldc = cv.Cout; p.Cf = nullptr; p.stats = stats ? c.part() : nullptr;
  return gemm_nt_launch(p, 1, c.st);
}
// eval mode: the BatchNorm (+PReLU, + identity, + the next block's bn1 as a second output) that follows a conv applied in the conv
// kernel's epilogue (GemmNT::esc ...), for the layers whose kernel implements it
int g_eval_fuse = 1;   // option "eval_fuse"
static bool conv_epilogue_ok(const Ctx& c, const ConvD& cv) {
  return g_eval_fuse && gemm_nt_conv_epilogue_ok(cv.Hin, cv.Cin, cv.Cout, c.n->B * cv.Hout * cv.Hout, cv.R, cv.stride);
}
static int conv_fwd_ep(const Ctx& c, const ConvD& cv, const bf16_t* in, bf16_t* out, const BnD& bn, const float* alpha, const bf16_t* add,
                       bf16_t* out2, const BnD* bn2) {
  GemmNT p{};
  p.A = in; p.B = c.shadow + cv.w_off;
  p.M = c.n->B * cv.Hout * cv.Hout; p.N = cv.Cout; p.K = cv.R * cv.R * cv.Cin;
  p.mode = 1; p.H = cv.Hin; p.W = cv.Hin; p.C = cv.Cin; p.Ho = cv.Hout; p.Wo = cv.Hout; p.S = cv.R;
  p.stride = cv.stride; p.pad = 1; p.up = 1;
  p.Cb = out; p.ldc = cv.Cout; p.Cf = nullptr; p.stats = nullptr;
  p.esc = c.save(bn, 0); p.esh = c.save(bn, 1); p.ealpha = alpha; p.eadd = add;
  if (out2) { p.Cb2 = out2; p.esc2 = c.save(*bn2, 0); p.esh2 = c.save(*bn2, 1); }
  return gemm_nt_launch(p, 1, c.st);
}
int g_fwd_xmom = 1;       // option "fwd_xmom": bn3 + identity + the next block's bn1 as one pass from conv2's raw moments (14x14 / 28x28 blocks)
int g_fuse_bnbwd28 = 1;   // option "fuse_bnbwd28": ... and in the two-tiles 28x28 dgrad (with fuse_bnbwd != 0)
int g_c64p_bnbwd = 1;   // option "c64p_bnbwd": BN-backward reduction in the epilogue of the persistent 64-channel dgrad kernel (with fuse_bnbwd != 0)
extern int g_conv_c64p;
int g_fuse_bnbwd = 2;   // (round 3: 2 is the default, together with wgrad9p = 1 — see the end of this comment)  option "fuse_bnbwd": BN-backward reduction in the 3x3 dgrad epilogue (1: every layer a fused kernel serves; 2: the 14x14
                        // layers only, whose 128 partial rows the channel-sliced apply pass reduces itself).  LDS-DMA kernel: the x tile rides
                        // through the K loop in registers, the coefficients are requested in front of the drain: +3.7 us on a 28.6 us dgrad
                        // (round 1: x by LDS-DMA after the loop and 62 lazily issued coefficient loads, +11 us).  Off by default all the same:
                        // alone it removes 58 reduce launches (-0.35 ms of kernel time) and LENGTHENS the dual-stream step (16.96 -> 17.25 ms),
                        // because the reduce passes were windows in which the weight-gradient workgroups share the CUs, and the convolution
                        // that now follows sooner cannot start on a CU a weight-gradient workgroup still holds (54.9 us per fused dgrad in the
                        // dual-stream trace, 32.3 alone).  With the paired weight-gradient kernel (wgrad9p = 1) beside it the step is back at
                        // 16.94: both sides of the overlap have to shrink together (profiles/r02_ab_fuse_bnbwd_v2.txt).
                        // Round 3 (after the fork / join events lost their system-scope fence): the pair fuse_bnbwd = 2 + wgrad9p = 1 measures
                        // 17.00 / 17.01 ms against 17.21 / 17.15 for the old defaults, each alone still loses (17.40 / 17.34): both are on.
// dx (at the conv's INPUT resolution) = conv_transpose(dy).  If `bn` is given, the kernel may also produce the
// BN-backward partial sums of (dx, bn_x) in its epilogue; *fused_rows > 0 then (else run ew_bn_bwd_reduce).
static int conv_dgrad(const Ctx& c, const ConvD& cv, const bf16_t* dy, bf16_t* dx, const BnD* bn = nullptr,
                      const bf16_t* bn_x = nullptr, const float* alpha = nullptr, int* fused_rows = nullptr) {
  GemmNT p{};
  // option value 2: only the 14x14 layers (one partial row per image: few enough for the channel-sliced apply pass to reduce itself)
  // ... and (round 3) the 64 -> 64 layers of the 56x56 / 112x112 maps on the persistent kernel: one row per workgroup (conv_c64p.hip)
  const bool c64 = g_c64p_bnbwd && g_conv_c64p && cv.R == 3 && cv.stride == 1 && cv.Cin == 64 && cv.Cout == 64 && (cv.Hin == 56 || cv.Hin == 112);
  // ... and the 28x28 layers on the two-tiles LDS-DMA kernel: one row per workgroup = 256 at B = 128, which the sliced apply pass takes
  const bool w28 = g_fuse_bnbwd28 && cv.R == 3 && cv.stride == 1 && cv.Hin == 28 && cv.Cin % 128 == 0 && cv.Cout % 128 == 0 &&
                   gemm_nt_fused28_two_tiles(c.n->B * 28 * 28) && ew_bn_sliced_ok(c.n->B * 28 * 28, cv.Cin, c.n->B * 28 * 28 / 392, true);
  if (bn && fused_rows && g_fuse_bnbwd && (g_fuse_bnbwd == 1 || c64 || w28 || (cv.R == 3 && cv.stride == 1 && cv.Hin == 14 && cv.Cin % 128 == 0 && cv.Cout % 128 == 0))) {
    p.bx = bn_x; p.bmean = c.save(*bn, 2); p.brstd = c.save(*bn, 3); p.bgamma = c.gamma(*bn); p.bbeta = c.beta(*bn);
    p.balpha = alpha; p.bpart = c.part(); p.bwd_fused = fused_rows;
  }
  p.B = c.shadow + cv.wd_off; p.A = dy;
  p.N = cv.Cin; p.K = cv.R * cv.R * cv.Cout;
  p.Cb = dx; p.ldc = cv.Cin; p.Cf = nullptr; p.stats = nullptr;
  if (cv.R == 1) {          // 1x1 stride-s: plain GEMM at the OUTPUT resolution (compact result; caller up-samples)
    p.mode = 0; p.M = c.n->B * cv.Hout * cv.Hout; p.lda = cv.Cout;
  } else {
    p.mode = 1; p.M = c.n->B * cv.Hin * cv.Hin;
    p.H = cv.Hout; p.W = cv.Hout; p.C = cv.Cout; p.Ho = cv.Hin; p.Wo = cv.Hin; p.S = 3;
    p.stride = 1; p.pad = 1; p.up = cv.stride;
  }
  return gemm_nt_launch(p, 1, c.st);
}
static GemmTN wgrad_problem(const Ctx& c, const ConvD& cv, const bf16_t* in, const bf16_t* dy) {
  GemmTN p{};
  p.P = dy; p.Q = in;
  p.Kp = c.n->B * cv.Hout * cv.Hout; p.NI = cv.Cout; p.NJ = cv.R * cv.R * cv.Cin;
  p.mode = 1; p.H = cv.Hin; p.W = cv.Hin; p.C = cv.Cin; p.Ho = cv.Hout; p.Wo = cv.Hout; p.S = cv.R;
  p.stride = cv.stride; p.pad = (cv.R == 3) ? 1 : 0;
  p.ldp = cv.Cout; p.ldq = 0; p.use_tr = g_tn_use_tr;
  return p;
}
static int conv_wgrad(const Ctx& c, const ConvD& cv, const bf16_t* in, const bf16_t* dy, hipStream_t st) {
  GemmTN p = wgrad_problem(c, cv, in, dy);
  const int splits = gemm_tn_pick_splits(p.Kp, p.NI, p.NJ, p.C, p.Wo, p.stride);
  float* dst = c.grads + cv.w_off;
  if (splits == 1) {
    p.out = dst;
    return gemm_tn_launch(p, 1, st);
  }
  // the slab workspace was sized at plan creation (gemm_tn_max_splits over every kernel choice)
  FEDFR_REQUIRE((size_t)splits * p.NI * p.NJ <= c.n->slab_floats, "conv_wgrad: %d split-K slabs of %d x %d exceed the plan's slab workspace "
                "(%zu floats)", splits, p.NI, p.NJ, c.n->slab_floats);
  p.out = c.slab();
  FEDFR_TRY(gemm_tn_launch(p, splits, st));
  return ew_reduce_slabs(dst, c.slab(), splits, (size_t)p.NI * p.NJ, nullptr, 0, st);
}
// Slab sets of a paired nine-tap launch whose reduction has not been issued yet: the NEXT paired launch on the stream carries it out beside its own
// work (wgrad9p.hip, W9PJob), or wgrad_flush() issues the stand-alone launches — before anything else uses the slab workspace or reads the
// gradients (another weight-gradient kernel, the fused SGD of a finished stage, the end of the pass).
struct WgradPending {
  W9PJob job{};          // job.n == 0: nothing pending
  int set = 1;           // slab regions {2 set, 2 set + 1} hold the pending pair (the next pair writes the other set)
};
static int wgrad_flush(const Ctx& c, WgradPending* pd, hipStream_t st) {
  if (!pd || !pd->job.n) return FEDFR_OK;
  const W9PJob j = pd->job;
  pd->job.n = 0;
  // two reduction launches on purpose: ONE launch for both layers (ew_reduce_slabs2) measured 16.00 vs 15.85 ms/step same-box in round 3 —
  // the main stream waits for whatever the weight-gradient stream has resident, and two short kernels release the CUs sooner than one long
  FEDFR_TRY(ew_reduce_slabs(j.dst[0], j.slab[0], j.nsplit, j.n, nullptr, 0, st));
  return ew_reduce_slabs(j.dst[1], j.slab[1], j.nsplit, j.n, nullptr, 0, st);
}
int g_wgrad_pair_reduce = 1;   // option "wgrad_pair_reduce": one slab-reduction launch for the two 3x3 weight gradients of a block
// the two 3x3 weight gradients of a residual block; same shape (every block but a stage's first): one paired launch
#ifdef FEDFR_DEBUG
int g_dbg_skip = 0;   // option "dbg_skip" (-DFEDFR_DEBUG builds only; timing experiments, WRONG results): 1 = no weight-gradient launches of the residual blocks' 3x3 convs
#endif
static int conv_wgrad2(const Ctx& c, const ConvD& cva, const bf16_t* ina, const bf16_t* dya, const ConvD& cvb, const bf16_t* inb,
                       const bf16_t* dyb, hipStream_t st, WgradPending* pd = nullptr) {
#ifdef FEDFR_DEBUG
  if (g_dbg_skip & 1) return FEDFR_OK;
#endif
  GemmTN a = wgrad_problem(c, cva, ina, dya), b = wgrad_problem(c, cvb, inb, dyb);
  if (gemm_tn_w9pair_ok(a, b)) {                        // both on the 64 x 64 nine-tap kernel, one launch
    const int sp = gemm_tn_w9pair_splits(a);
    FEDFR_REQUIRE((size_t)sp * a.NI * a.NJ <= c.n->slab_floats, "conv_wgrad2: %d split-K slabs of %d x %d exceed the plan's slab workspace (%zu floats)",
                  sp, a.NI, a.NJ, c.n->slab_floats);
    const int set = pd ? pd->set ^ 1 : 0;
    static_assert(kSlabRegions >= 4, "two slab sets of two regions each");
    a.out = c.slab(2 * set); b.out = c.slab(2 * set + 1);
    // the previous pair's slabs: summed by this launch where its shape allows, else by their own launches first
    const bool carry = pd && pd->job.n && gemm_tn_w9pair_job_ok(a, sp, pd->job);
    if (pd && pd->job.n && !carry) FEDFR_TRY(wgrad_flush(c, pd, st));
    FEDFR_TRY(gemm_tn_launch_w9pair(a, b, sp, st, carry ? &pd->job : nullptr));
    W9PJob mine{};
    mine.slab[0] = a.out; mine.slab[1] = b.out; mine.dst[0] = c.grads + cva.w_off; mine.dst[1] = c.grads + cvb.w_off;
    mine.n = (size_t)a.NI * a.NJ; mine.nsplit = sp;
    if (pd) {
      pd->job = mine; pd->set = set;
      return FEDFR_OK;
    }
    WgradPending now; now.job = mine;
    return wgrad_flush(c, &now, st);
  }
  FEDFR_TRY(wgrad_flush(c, pd, st));                    // every other kernel below uses slab regions 0 / 1 itself
  const int splits = gemm_tn_pick_splits(a.Kp, a.NI, a.NJ, a.C, a.Wo, a.stride);
  // same shape, same split count: each launch writes its own slab set and ONE launch reduces both (45 launches fewer per step on the
  // weight-gradient stream, and the second GEMM does not wait behind the first one's reduction)
  if (g_wgrad_pair_reduce && splits >= 2 && a.NI == b.NI && a.NJ == b.NJ && a.Kp == b.Kp && a.C == b.C && a.Wo == b.Wo && a.stride == b.stride &&
      (size_t)splits * a.NI * a.NJ <= c.n->slab_floats) {
    a.out = c.slab(0); b.out = c.slab(1);
    FEDFR_TRY(gemm_tn_launch(a, splits, st));
    FEDFR_TRY(gemm_tn_launch(b, splits, st));
    return ew_reduce_slabs2(c.grads + cva.w_off, c.slab(0), c.grads + cvb.w_off, c.slab(1), splits, (size_t)a.NI * a.NJ, st);
  }
  FEDFR_TRY(conv_wgrad(c, cva, ina, dya, st));
  return conv_wgrad(c, cvb, inb, dyb, st);
}
struct Rows { const float* ptr; int P; };     // where a BatchNorm's partial statistics rows are
static int bn_coeffs(const Ctx& c, const BnD& b, Rows r, double count, bool training) {
  if (training)
    return ew_bn_finalize(r.ptr, r.P, b.C, count, c.gamma(b), c.beta(b), c.bufs + b.rm_off, c.bufs + b.rv_off, kBnMomentum,
                          kBnEps, c.save(b, 0), c.save(b, 1), c.save(b, 2), c.save(b, 3), c.ftmp(), c.st);
  return FEDFR_OK;      // eval mode: every BatchNorm's (scale, shift) was computed up front by eval_coeffs_all (one launch)
}
// eval-mode coefficients of all 2-D BatchNorms of the network in one launch (they depend on parameters and running statistics only;
// one small kernel per BatchNorm cost 154 launches = 0.6 ms of a 5.7 ms forward at batch 128)
static int eval_coeffs_all(const Ctx& c) {
  BnEvalTable t{};
  t.eps = kBnEps;
  auto flush = [&]() -> int {
    if (t.n == 0) return FEDFR_OK;
    const int rc = ew_bn_eval_coeffs_multi(c.params, c.bufs, c.actf, t, c.st);
    t.n = 0;
    return rc;
  };
  auto add = [&](const BnD& b) -> int {
    if (t.n == kMaxBnEvalEntries) FEDFR_TRY(flush());
    BnEvalEntry& e = t.e[t.n++];
    e.C = b.C; e.g_off = (int)b.g_off; e.b_off = (int)b.b_off; e.rm_off = (int)b.rm_off; e.rv_off = (int)b.rv_off; e.save_off = (int)b.save_off;
    return FEDFR_OK;
  };
  if (!c.n->block_only) FEDFR_TRY(add(c.n->stem_bn));
  for (const auto& k : c.n->blocks) {
    FEDFR_TRY(add(k.bn1)); FEDFR_TRY(add(k.bn2)); FEDFR_TRY(add(k.bn3));
    if (k.has_ds) FEDFR_TRY(add(k.bnds));
  }
  if (!c.n->block_only) FEDFR_TRY(add(c.n->bn2));
  return flush();
}
static int apply(const Ctx& c, const bf16_t* x1, const BnD& b1, const float* alpha, const bf16_t* x2, const BnD* b2, bf16_t* y,
                 int M, bool stats, int nchw_hw = 0) {
  BnApply a{};
  a.x1 = x1; a.sc1 = c.save(b1, 0); a.sh1 = c.save(b1, 1); a.alpha = alpha;
  a.x2 = x2; a.sc2 = b2 ? c.save(*b2, 0) : nullptr; a.sh2 = b2 ? c.save(*b2, 1) : nullptr;
  a.y = y; a.M = M; a.C = b1.C; a.nchw_hw = nchw_hw; a.stats = stats ? c.part() : nullptr;
  return ew_bn_apply(a, c.st);
}

// train-mode BatchNorm whose statistics are the partial rows `r`, applied: y = prelu?(bn(x1)) (+ x2).  One channel-sliced launch that reduces
// the rows itself where the shape allows (bn_sliced.hip), else finalize + the row-slab pass.  *out (optional) = where the statistics of y are.
static int bn_apply_train(const Ctx& c, const BnD& b, Rows r, const bf16_t* x1, const float* alpha, const bf16_t* x2, bf16_t* y, int M,
                          Rows* out) {
  if (ew_bn_sliced_ok(M, b.C, r.P, false)) {
    BnApplyS a{};
    a.part = r.ptr; a.P = r.P; a.count = (double)M; a.gamma = c.gamma(b); a.beta = c.beta(b);
    a.rm = c.bufs + b.rm_off; a.rv = c.bufs + b.rv_off; a.momentum = kBnMomentum; a.eps = kBnEps;
    a.scale = c.save(b, 0); a.shift = c.save(b, 1); a.mean = c.save(b, 2); a.rstd = c.save(b, 3);
    a.x1 = x1; a.alpha = alpha; a.x2 = x2; a.y = y; a.M = M; a.C = b.C;
    a.stats = out ? c.part_other(r.ptr) : nullptr;
    if (out) *out = Rows{a.stats, ew_bn_sliced_rows(M, b.C)};
    return ew_bn_apply_sliced(a, c.st);
  }
  FEDFR_TRY(bn_coeffs(c, b, r, (double)M, true));
  FEDFR_TRY(apply(c, x1, b, alpha, x2, nullptr, y, M, out != nullptr));
  if (out) *out = Rows{c.part(), ew_bn_apply_grid(M, b.C)};
  return FEDFR_OK;
}

static int sph_prepare_weights(const FedfrNet* n, const float* params, bf16_t* shadow, int fwd_shadow_too, hipStream_t st);
static int sph_forward(const FedfrNet* n, const float* x, const float* params, const bf16_t* shadow, unsigned char* act, unsigned char* ws,
                       float* feats, hipStream_t st);
static int sph_backward(const FedfrNet* n, const float* dfeats, const float* params, const bf16_t* shadow, unsigned char* act, unsigned char* ws,
                        float* grads, hipStream_t st, hipStream_t aux, NetSgd* sgd);
int net_prepare_weights(const FedfrNet* n, const float* params, bf16_t* shadow, int fwd_shadow_too, hipStream_t st) {
  FEDFR_REQUIRE(n && params && shadow, "prepare_weights: null");
  if (n->sph_type) return sph_prepare_weights(n, params, shadow, fwd_shadow_too, st);
  if (fwd_shadow_too) FEDFR_TRY(ew_cast_f32_bf16(params, shadow, (size_t)n->trainable_count, st));
  // every conv's dgrad-layout copy ([Cin][taps flipped][Cout] bf16) in ONE launch (was one small kernel per conv: 105 per step)
  ShadowTable t{};
  auto add = [&](const ConvD& cv) {
    ShadowEntry& e = t.e[t.n++];
    e.src = (unsigned long long)cv.w_off; e.dst = (unsigned long long)cv.wd_off;
    e.cout64 = (unsigned short)(cv.Cout / 64); e.cin64 = (unsigned short)(cv.Cin / 64); e.rs = (unsigned short)(cv.R * cv.R);
  };
  for (const auto& k : n->blocks) {
    FEDFR_REQUIRE(k.conv1.Cout % 64 == 0 && k.conv1.Cin % 64 == 0 && k.conv2.Cout % 64 == 0 && k.conv2.Cin % 64 == 0,
                  "prepare_weights: channels not a multiple of 64");
    if (t.n + 3 > kMaxShadowEntries) {                  // deeper nets than iresnet100: one launch per table-full
      FEDFR_TRY(ew_weight_dgrad_shadow_multi(params, shadow, t, st));
      t.n = 0;
    }
    add(k.conv1); add(k.conv2);
    if (k.has_ds) add(k.ds);
  }
  if (t.n) FEDFR_TRY(ew_weight_dgrad_shadow_multi(params, shadow, t, st));
  return FEDFR_OK;
}

int net_forward(const FedfrNet* n, const float* x, const float* params, float* bufs, const bf16_t* shadow,
                unsigned char* act, unsigned char* ws, float* feats, int training, hipStream_t st) {
  if (n && n->sph_type) {                             // sphnet: no BatchNorm, so no buffers and no train / eval difference
    FEDFR_REQUIRE(x && params && shadow && act && ws && feats, "net_forward (sphnet): null buffer");
    return sph_forward(n, x, params, shadow, act, ws, feats, st);
  }
  FEDFR_REQUIRE(n && x && params && bufs && shadow && act && ws && (feats || n->block_only), "net_forward: null buffer");
  Ctx c{n, params, bufs, shadow, reinterpret_cast<bf16_t*>(act), reinterpret_cast<float*>(act + n->act_float_off_bytes), ws, nullptr, st};
  FEDFR_REQUIRE(training >= 0 && training <= 2, "net_forward: training must be 0 (eval), 1 (train) or 2 (train, BatchNorms frozen in eval mode)");
  // training = 2: every BatchNorm normalises with its running statistics and updates nothing (freeze_BN(test_mode=True)) while the net
  // trains — dropout stays on and every activation the backward pass reads is kept: the per-layer eval path, not the fused epilogues
  const bool tr = training == 1, frozen = training == 2;
  n->bn_frozen = frozen;
  const int B = n->B, HW = n->HW;
  const int M0 = B * HW * HW;
  bf16_t* A = c.actb;
  if (!tr) FEDFR_TRY(eval_coeffs_all(c));
  Rows prev{c.part(), 0};                  // statistics of the tensor the next BatchNorm normalises
  if (n->block_only) {
    // lone block: x (fp32 NCHW) -> NHWC bf16 block input; an identity "apply" pass leaves the column statistics bn1 needs, exactly
    // where the previous block's output pass leaves them inside a network
    const BlockD& k0 = n->blocks.front();
    FEDFR_TRY(ew_nchw_f32_to_nhwc_bf16(x, c.g(0), B, k0.Cin, HW * HW, st));
    BnApply a{};
    a.x1 = c.g(0); a.y = A + k0.x_off; a.M = M0; a.C = k0.Cin; a.stats = tr ? c.part() : nullptr;
    FEDFR_TRY(ew_bn_apply(a, st));
    prev.P = ew_bn_apply_grid(M0, k0.Cin);
  } else {
    // stem: conv -> BN -> PReLU   (iresnet.py:160-162)
    FEDFR_TRY(ew_stem_fwd(x, params + n->stem.w_off, A + n->c0_off, tr ? c.part() : nullptr, B, HW, HW, st));
    FEDFR_TRY(bn_coeffs(c, n->stem_bn, Rows{c.part(), ew_stem_stat_rows(B, HW, HW)}, (double)M0, tr));
    FEDFR_TRY(apply(c, A + n->c0_off, n->stem_bn, params + n->stem_alpha_off, nullptr, nullptr, A + n->a0_off, M0, tr));
    prev.P = ew_bn_apply_grid(M0, 64);
  }
  bool a1_ready = false;                    // eval: the previous block's conv2 epilogue already wrote this block's bn1(x)
  for (size_t bi = 0; bi < n->blocks.size(); ++bi) {
    const BlockD& k = n->blocks[bi];
    const int Mi = B * k.Hin * k.Hin, Mo = B * k.Hout * k.Hout;
    if (!tr && !frozen && g_eval_fuse) {
      // ---- eval mode: BatchNorms are known affines -> they ride in the conv epilogues where the kernel has one
      if (!a1_ready) FEDFR_TRY(apply(c, A + k.x_off, k.bn1, nullptr, nullptr, nullptr, A + k.a1_off, Mi, false));
      a1_ready = false;
      if (conv_epilogue_ok(c, k.conv1)) {
        FEDFR_TRY(conv_fwd_ep(c, k.conv1, A + k.a1_off, A + k.a2_off, k.bn2, params + k.alpha_off, nullptr, nullptr, nullptr));
      } else {
        FEDFR_TRY(conv_fwd(c, k.conv1, A + k.a1_off, A + k.c1_off, false));
        FEDFR_TRY(apply(c, A + k.c1_off, k.bn2, params + k.alpha_off, nullptr, nullptr, A + k.a2_off, Mi, false));
      }
      if (k.has_ds) FEDFR_TRY(conv_fwd(c, k.ds, A + k.x_off, A + k.d_off, false));
      if (conv_epilogue_ok(c, k.conv2)) {
        const bf16_t* idn = A + k.x_off;
        if (k.has_ds) {                                                 // identity = bnds(ds(x)), materialised in the (unused) c2 slot
          FEDFR_TRY(apply(c, A + k.d_off, k.bnds, nullptr, nullptr, nullptr, A + k.c2_off, Mo, false));
          idn = A + k.c2_off;
        }
        const BlockD* nx = bi + 1 < n->blocks.size() ? &n->blocks[bi + 1] : nullptr;
        FEDFR_TRY(conv_fwd_ep(c, k.conv2, A + k.a2_off, A + k.out_off, k.bn3, nullptr, idn, nx ? A + nx->a1_off : nullptr, nx ? &nx->bn1 : nullptr));
        a1_ready = nx != nullptr;
      } else {
        FEDFR_TRY(conv_fwd(c, k.conv2, A + k.a2_off, A + k.c2_off, false));
        if (k.has_ds) FEDFR_TRY(apply(c, A + k.c2_off, k.bn3, nullptr, A + k.d_off, &k.bnds, A + k.out_off, Mo, false));
        else FEDFR_TRY(apply(c, A + k.c2_off, k.bn3, nullptr, A + k.x_off, nullptr, A + k.out_off, Mo, false));
      }
      continue;
    }
    if (!tr) {                                // eval mode without the fused epilogues: known affines, plain passes
      FEDFR_TRY(apply(c, A + k.x_off, k.bn1, nullptr, nullptr, nullptr, A + k.a1_off, Mi, false));
      FEDFR_TRY(conv_fwd(c, k.conv1, A + k.a1_off, A + k.c1_off, false));
      FEDFR_TRY(apply(c, A + k.c1_off, k.bn2, params + k.alpha_off, nullptr, nullptr, A + k.a2_off, Mi, false));
      FEDFR_TRY(conv_fwd(c, k.conv2, A + k.a2_off, A + k.c2_off, false));
      if (k.has_ds) {
        FEDFR_TRY(conv_fwd(c, k.ds, A + k.x_off, A + k.d_off, false));
        FEDFR_TRY(apply(c, A + k.c2_off, k.bn3, nullptr, A + k.d_off, &k.bnds, A + k.out_off, Mo, false));
      } else {
        FEDFR_TRY(apply(c, A + k.c2_off, k.bn3, nullptr, A + k.x_off, nullptr, A + k.out_off, Mo, false));
      }
      continue;
    }
    // a1 = bn1(x) -> conv1 -> a2 = prelu(bn2(c1)) -> conv2(stride) -> bn3(c2) + identity.  Statistics: the pass that produced x left
    // them in `prev`; a conv's epilogue leaves its output's in c.part()
    // round 3 (option fwd_xmom): where conv2 runs on an LDS-DMA kernel with the moment epilogue, out = bn3(c2) + x AND the next block's
    // bn1(out) leave ONE pass (bn_apply2, ew.h): conv2 also sums c2 * x, and the statistics of `out` follow from those moments and the
    // statistics of x this block's bn1 saved — the next block's bn1 pass disappears
    const BlockD* nxb = bi + 1 < n->blocks.size() ? &n->blocks[bi + 1] : nullptr;
    const int xm_rows = k.Hout == 14 ? Mo / 196 : Mo / 392;
    const bool xmom = g_fwd_xmom && nxb && !k.has_ds && k.conv2.R == 3 && k.conv2.stride == 1 && (k.Hout == 14 || k.Hout == 28) &&
                      k.conv2.Cin % 128 == 0 && k.Cout % 128 == 0 && k.conv2.Cin == k.Cout && nxb->bn1.C == k.Cout &&
                      gemm_nt_conv_epilogue_ok(k.Hout, k.conv2.Cin, k.Cout, Mo, 3, 1) &&          // (mirrors the dispatch: small problems take the generic kernel)
                      (k.Hout == 14 || gemm_nt_fused28_two_tiles(Mo)) && ew_bn_apply2_sliced_ok(Mo, k.Cout, xm_rows);
    if (a1_ready) {                             // the previous block's output pass wrote a1 = bn1(x) and bn1's statistics
      FEDFR_TRY(conv_fwd(c, k.conv1, A + k.a1_off, A + k.c1_off, true));
      a1_ready = false;
    } else {
      FEDFR_TRY(bn_apply_train(c, k.bn1, prev, A + k.x_off, nullptr, nullptr, A + k.a1_off, Mi, nullptr));
      FEDFR_TRY(conv_fwd(c, k.conv1, A + k.a1_off, A + k.c1_off, true));
    }
    const Rows r1{c.part(), gemm_nt_stat_rows_live(Mi, k.Cout, k.conv1.Cin, k.conv1.Hin, k.conv1.R, k.conv1.stride)};
    if (xmom) {
      FEDFR_TRY(bn_apply_train(c, k.bn2, r1, A + k.c1_off, params + k.alpha_off, nullptr, A + k.a2_off, Mi, nullptr));
      GemmNT p{};
      const ConvD& cv = k.conv2;
      p.A = A + k.a2_off; p.B = c.shadow + cv.w_off;
      p.M = Mo; p.N = cv.Cout; p.K = 9 * cv.Cin;
      p.mode = 1; p.H = cv.Hin; p.W = cv.Hin; p.C = cv.Cin; p.Ho = cv.Hout; p.Wo = cv.Hout; p.S = 3;
      p.stride = 1; p.pad = 1; p.up = 1;
      p.Cb = A + k.c2_off; p.ldc = cv.Cout;
      int rows = 0;
      p.bx = A + k.x_off; p.bmean = c.save(k.bn3, 2); p.brstd = c.save(k.bn3, 3); p.bpart = c.part(); p.bmom = 1; p.bwd_fused = &rows;
      FEDFR_TRY(gemm_nt_launch(p, 1, c.st));
      FEDFR_REQUIRE(rows == xm_rows, "net_forward: the moment epilogue left %d rows, %d expected", rows, xm_rows);
      BnApply2S a{};
      a.part = c.part(); a.P = rows; a.count = (double)Mo; a.momentum = kBnMomentum; a.eps = kBnEps;
      a.gamma = c.gamma(k.bn3); a.beta = c.beta(k.bn3); a.rm = c.bufs + k.bn3.rm_off; a.rv = c.bufs + k.bn3.rv_off;
      a.scale = c.save(k.bn3, 0); a.shift = c.save(k.bn3, 1); a.mean = c.save(k.bn3, 2); a.rstd = c.save(k.bn3, 3);
      a.xmean = c.save(k.bn1, 2); a.xrstd = c.save(k.bn1, 3);
      const BnD& nb = nxb->bn1;
      a.ngamma = c.gamma(nb); a.nbeta = c.beta(nb); a.nrm = c.bufs + nb.rm_off; a.nrv = c.bufs + nb.rv_off;
      a.nscale = c.save(nb, 0); a.nshift = c.save(nb, 1); a.nmean = c.save(nb, 2); a.nrstd = c.save(nb, 3);
      a.x1 = A + k.c2_off; a.x2 = A + k.x_off; a.y = A + k.out_off; a.y2 = A + nxb->a1_off; a.M = Mo; a.C = k.Cout;
      FEDFR_TRY(ew_bn_apply2_sliced(a, c.st));
      a1_ready = true;
      prev = Rows{c.part(), 0};
      continue;
    } else {
      FEDFR_TRY(bn_apply_train(c, k.bn2, r1, A + k.c1_off, params + k.alpha_off, nullptr, A + k.a2_off, Mi, nullptr));
      FEDFR_TRY(conv_fwd(c, k.conv2, A + k.a2_off, A + k.c2_off, true));
    }
    const Rows r2{c.part(), gemm_nt_stat_rows_live(Mo, k.Cout, k.conv2.Cin, k.conv2.Hin, k.conv2.R, k.conv2.stride)};
    if (k.has_ds) {
      FEDFR_TRY(bn_coeffs(c, k.bn3, r2, (double)Mo, true));
      FEDFR_TRY(conv_fwd(c, k.ds, A + k.x_off, A + k.d_off, true));
      FEDFR_TRY(bn_coeffs(c, k.bnds, Rows{c.part(), gemm_nt_stat_rows(Mo, k.Cout)}, (double)Mo, true));
      FEDFR_TRY(apply(c, A + k.c2_off, k.bn3, nullptr, A + k.d_off, &k.bnds, A + k.out_off, Mo, true));
      prev = Rows{c.part(), ew_bn_apply_grid(Mo, k.Cout)};
    } else {
      FEDFR_TRY(bn_apply_train(c, k.bn3, r2, A + k.c2_off, nullptr, A + k.x_off, A + k.out_off, Mo, &prev));
    }
  }
  if (n->block_only) return FEDFR_OK;
  // bn2 -> flatten (NCHW order) -> fc -> features   (iresnet.py:167-171)
  const BlockD& last = n->blocks.back();
  const int hw = n->final_hw * n->final_hw, Mf = B * hw;
  FEDFR_TRY(bn_coeffs(c, n->bn2, prev, (double)Mf, tr));
  FEDFR_TRY(apply(c, A + last.out_off, n->bn2, nullptr, nullptr, nullptr, A + n->t_off, Mf, false, hw));
  if ((tr || frozen) && n->dropout_p > 0.f)            // nn.Dropout(p, inplace=True) on the flattened bn2 output (iresnet.py:169); identity in eval mode
    FEDFR_TRY(ew_dropout_fwd(A + n->t_off, act + n->mask_off_bytes, (size_t)B * n->fc_in, n->dropout_p, n->dropout_seed, n->dropout_step++, st));
  {
    GemmNT p{};
    p.A = A + n->t_off; p.B = shadow + n->fc_w_off; p.M = B; p.N = n->F; p.K = n->fc_in; p.mode = 0; p.lda = n->fc_in;
    p.Cb = nullptr; p.Cf = c.slab(); p.stats = nullptr;
    const int splits = gemm_nt_pick_splits(B, n->F, n->fc_in);
    FEDFR_REQUIRE((size_t)splits * B * n->F <= n->slab_floats, "net_forward: fc split-K slabs exceed the plan's slab workspace");
    FEDFR_TRY(gemm_nt_launch(p, splits, st));
    FEDFR_TRY(ew_reduce_slabs(c.actf + n->yfc_off, c.slab(), splits, (size_t)B * n->F, params + n->fc_b_off, n->F, st));
  }
  FEDFR_TRY(ew_bn1d_fwd(c.actf + n->yfc_off, feats, B, n->F, params + n->feat_bn.g_off, params + n->feat_bn.b_off,
                        bufs + n->feat_bn.rm_off, bufs + n->feat_bn.rv_off, kBnMomentum, kBnEps, tr ? 1 : 0,
                        c.actf + n->feat_save_off, c.actf + n->feat_save_off + n->F, st));
  return FEDFR_OK;
}

// debug capture (tests): when set, net_backward copies the gradient entering every block (bf16 NHWC [B*Hout*Hout][Cout], last block first)
// and finally the gradient wrt the first block's input, back to back into this caller-owned device buffer
bf16_t* g_dbg_grads = nullptr;
size_t g_dbg_grads_elems = 0;
static int dbg_capture(const bf16_t* src, size_t elems, size_t* off, hipStream_t st) {
  if (!g_dbg_grads) return FEDFR_OK;
  FEDFR_REQUIRE(*off + elems <= g_dbg_grads_elems, "net_backward: debug gradient buffer too small (%zu + %zu > %zu elements)", *off, elems,
                g_dbg_grads_elems);
  if (hipMemcpyAsync(g_dbg_grads + *off, src, elems * 2, hipMemcpyDeviceToDevice, st) != hipSuccess) {
    fedfr_set_error("net_backward: debug capture copy failed");
    return FEDFR_ERR_HIP;
  }
  *off += elems;
  return FEDFR_OK;
}
static const int g_wgrad_depth = kWgradDepth;   // generations of weight-gradient operands in flight (fewer: the main stream waits for the weight-gradient stream)
int g_fuse_bnred_next = 1;   // option "fuse_bnred_next": a BN-backward apply pass also reduces its output for the BN that consumes it
// BatchNorm (+PReLU) backward: dx = a dz + A x + B (+ addend).  `have`: partial rows that already exist (left by the pass that produced
// dy); else a reduce pass runs first.  With next_bn, dx is also reduced as the dy of that BatchNorm's backward -> *next_rows.
// Channel-sliced passes without a finalize launch where the shape allows (bn_sliced.hip), else reduce / finalize / apply of ew.hip.
static int bn_bwd(const Ctx& c, const BnD& b, const float* alpha, const bf16_t* dy, const bf16_t* x, int M, const bf16_t* add,
                  const bf16_t* add_up, int H, bf16_t* dx, long long alpha_off, Rows have = Rows{nullptr, 0}, const BnD* next_bn = nullptr,
                  const bf16_t* next_x = nullptr, Rows* next_rows = nullptr, const float* next_alpha = nullptr, bool coef_only = false) {
  // (coef_only: reduce if needed + finalize, no apply pass — the consumer applies c.coef() itself: the stem's weight gradient; row-slab form only)
  // (next_alpha: the consuming BatchNorm has a PReLU behind it — the stem's; served by the row-slab apply pass without own PReLU / addend)
#ifdef FEDFR_DBG_SKIP14
  // timing experiment only (WRONG results; tools/build_ablate.sh net.hip FEDFR_DBG_SKIP14 1): the bn1 / bn3 backward passes of the 14x14 stage are free —
  // the upper bound of anything that takes them out of the dispatcher's hands
  if (FEDFR_DBG_SKIP14 && !alpha && !c.n->block_only && M == c.n->B * 196 && b.C == 256) return FEDFR_OK;
#endif
  const bool nxt = next_bn && next_x && next_rows && g_fuse_bnred_next && next_bn->C == b.C &&
                   (!next_alpha || (!alpha && !add && !ew_bn_sliced_ok(M, b.C, have.P > 0 ? have.P : ew_bn_sliced_rows(M, b.C, true), true)));
  // frozen BatchNorm (eval mode inside a training net): mean / rstd were constants, so dx = gamma rstd dz — the same passes with an infinite
  // count (the two mean terms vanish); dgamma / dbeta / dalpha are the same sums
  const double count = c.n->bn_frozen ? HUGE_VAL : (double)M;
  if (!coef_only && !add_up && ew_bn_sliced_ok(M, b.C, have.P > 0 ? have.P : ew_bn_sliced_rows(M, b.C, true), true)) {
    BnBwdS p{};
    p.dy = dy; p.x = x; p.mean = c.save(b, 2); p.rstd = c.save(b, 3); p.gamma = c.gamma(b); p.alpha = alpha;
    p.sc = c.save(b, 0); p.sh = c.save(b, 1); p.M = M; p.C = b.C; p.count = count;
    p.dgamma = c.grads + b.g_off; p.dbeta = c.grads + b.b_off; p.dalpha = alpha ? c.grads + alpha_off : nullptr;
    p.add = add; p.dx = dx;
    if (have.P > 0) {
      p.part_in = have.ptr; p.P = have.P;
    } else {
      p.partials = c.part();
      FEDFR_TRY(ew_bn_bwd_reduce_sliced(p, c.st));
      p.part_in = c.part(); p.P = ew_bn_sliced_rows(M, b.C, true);
    }
    if (nxt) {
      p.nx = next_x; p.nmean = c.save(*next_bn, 2); p.nrstd = c.save(*next_bn, 3); p.npart = c.part_other(p.part_in);
      *next_rows = Rows{p.npart, ew_bn_sliced_rows(M, b.C, true)};
    }
    return ew_bn_bwd_apply_sliced(p, c.st);
  }
  BnBwd p{};
  if (nxt) {
    // dx is the dy of next_bn's backward: its (sum, sum * xhat) partials ride along in this apply pass (one tensor read instead of
    // a separate two-tensor reduce kernel); they land in the shared partial buffer, which this BN's finalize has finished reading
    p.nx = next_x; p.nmean = c.save(*next_bn, 2); p.nrstd = c.save(*next_bn, 3); p.npart = c.part();
    if (next_alpha) { p.nsc = c.save(*next_bn, 0); p.nsh = c.save(*next_bn, 1); p.nalpha = next_alpha; }
    *next_rows = Rows{c.part(), ew_bn_bwd_apply_grid(M, b.C)};
  }
  p.dy = dy; p.x = x; p.mean = c.save(b, 2); p.rstd = c.save(b, 3); p.gamma = c.gamma(b); p.beta = c.beta(b); p.alpha = alpha;
  p.sc = c.save(b, 0); p.sh = c.save(b, 1);
  p.M = M; p.C = b.C; p.partials = c.part(); p.coef = c.coef(); p.add = add; p.add_up = add_up; p.H = H; p.W = H; p.dx = dx;
  if (have.P <= 0) FEDFR_TRY(ew_bn_bwd_reduce(p, c.st));      // else: the producing pass already wrote the partials
  FEDFR_TRY(ew_bn_bwd_finalize(have.P > 0 ? have.ptr : c.part(), have.P > 0 ? have.P : ew_bn_bwd_grid(M, b.C), b.C, count, c.gamma(b), c.save(b, 2),
                               c.save(b, 3), c.grads + b.g_off, c.grads + b.b_off, alpha ? c.grads + alpha_off : nullptr, c.coef(), c.st));
  if (coef_only) return FEDFR_OK;
  return ew_bn_bwd_apply(p, c.st);
}

int g_stem_fuse_wgrad = 1;   // option "stem_fuse_wgrad": the stem's BatchNorm + PReLU backward is applied by its weight-gradient kernel on load (no d(conv output) tensor)
int g_stem_bnred = 1;     // option "stem_bnred": the stem's BatchNorm-backward reduction rides in the first block's bn1 apply pass
// (fc's weight gradient runs on the weight-gradient stream, and fork / join events are created with hipEventDisableSystemFence — a system-scope
// release per event costs the main stream ~5 us: options "fc_wgrad_aux" / "event_nofence" of rounds 3-5, removed in round 6 after losing every sweep)
// fork/join helpers for the dual-stream backward (events are created once per plan)
namespace {
struct Fork {
  const FedfrNet* n;
  hipStream_t main, aux;
  size_t next = 0;
  bool ok = true;
  hipEvent_t ev() {
    if (next == n->events.size()) {
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) { ok = false; return nullptr; }
      n->events.push_back(e);
    }
    return n->events[next++];
  }
  // everything enqueued so far on `from` happens before anything enqueued later on `to`
  void order(hipStream_t from, hipStream_t to) {
    if (!aux) return;
    hipEvent_t e = ev();
    if (!e || hipEventRecord(e, from) != hipSuccess || hipStreamWaitEvent(to, e, 0) != hipSuccess) ok = false;
  }
  hipEvent_t mark(hipStream_t s) {            // record now, wait later
    if (!aux) return nullptr;
    hipEvent_t e = ev();
    if (!e || hipEventRecord(e, s) != hipSuccess) ok = false;
    return e;
  }
  void wait(hipStream_t s, hipEvent_t e) {
    if (aux && e && hipStreamWaitEvent(s, e, 0) != hipSuccess) ok = false;
  }
};
}  // namespace

int net_backward(const FedfrNet* n, const float* x, const float* dfeats, const float* params, const bf16_t* shadow,
                 unsigned char* act, unsigned char* ws, float* grads, hipStream_t st, hipStream_t aux, NetSgd* sgd) {
  if (n && n->sph_type) {
    FEDFR_REQUIRE(dfeats && params && shadow && act && ws && grads, "net_backward (sphnet): null buffer");
    return sph_backward(n, dfeats, params, shadow, act, ws, grads, st, aux, sgd);
  }
  FEDFR_REQUIRE(n && (x || n->block_only) && dfeats && params && shadow && act && ws && grads, "net_backward: null buffer");
  Ctx c{n, params, nullptr, shadow, reinterpret_cast<bf16_t*>(act), reinterpret_cast<float*>(act + n->act_float_off_bytes), ws, grads, st};
  const int B = n->B, F = n->F, HW = n->HW;
  bf16_t* A = c.actb;
  Fork fk{n, st, aux};
  const hipStream_t wst = aux ? aux : st;          // stream of the weight-gradient GEMMs
  if (n->block_only) fk.order(st, wst);             // aux starts after everything already queued on main (forward pass); the full net forks below
  Rows pend{nullptr, 0};                           // partial rows of the next bn3 already reduced by the apply pass that produced its dy
  if (n->block_only) {
    const BlockD& k0 = n->blocks.front();
    FEDFR_TRY(ew_nchw_f32_to_nhwc_bf16(dfeats, c.g(1), B, k0.Cout, k0.Hout * k0.Hout, st));      // dy of the block, fp32 NCHW like the reference's
  } else {
  // ---- features (BN1d) backward; fc.bias grad = colsum(d y_fc) ----
  if (n->Bp != B && hipMemsetAsync(c.dybt(), 0, (size_t)F * n->Bp * 2, st) != hipSuccess) {      // zero padding of the batch axis of dY^T (Bp = B rounded up to 8)
    fedfr_set_error("net_backward: hipMemsetAsync failed");
    return FEDFR_ERR_HIP;
  }
  FEDFR_TRY(ew_bn1d_bwd(dfeats, c.actf + n->yfc_off, c.dyfc(), B, F, params + n->feat_bn.g_off, c.actf + n->feat_save_off,
                        c.actf + n->feat_save_off + F, grads + n->feat_bn.b_off, grads + n->fc_b_off, c.dyb(), c.dybt(), n->Bp, st,
                        n->bn_frozen ? 1 : 0));
  float* dxfc = reinterpret_cast<float*>(ws + n->ws_fc);
  // the first fork: the weight-gradient stream starts behind the forward pass and the features' backward; fc's weight gradient is its first
  // kernel (round 3: it ran on the main stream, 20 us in front of everything else)
  fk.order(st, wst);
  {  // fc.weight grad [F][fc_in] = dY^T X   (both operands batch-major -> TN kernel)
    GemmTN p{};
    p.P = c.dyb(); p.Q = A + n->t_off; p.Kp = B; p.NI = F; p.NJ = n->fc_in; p.mode = 0; p.ldp = F; p.ldq = n->fc_in;
    p.out = grads + n->fc_w_off; p.use_tr = g_tn_use_tr;
    FEDFR_TRY(gemm_tn_launch(p, 1, wst));
  }
  {  // dX [Bp][fc_in] = dY W   (reduction over F: P = dY^T [F][Bp], Q = W [F][fc_in])
    GemmTN p{};
    p.P = c.dybt(); p.Q = shadow + n->fc_w_off; p.Kp = F; p.NI = n->Bp; p.NJ = n->fc_in; p.mode = 0; p.ldp = n->Bp; p.ldq = n->fc_in;
    p.out = dxfc; p.use_tr = g_tn_use_tr;
    FEDFR_TRY(gemm_tn_launch(p, 1, st));
  }
  const BlockD& last = n->blocks.back();
  const int hw = n->final_hw * n->final_hw, Mf = B * hw;
  if (n->dropout_p > 0.f) FEDFR_TRY(ew_dropout_bwd(dxfc, act + n->mask_off_bytes, (size_t)B * n->fc_in, n->dropout_p, st));
  FEDFR_TRY(ew_nchw_f32_to_nhwc_bf16(dxfc, c.g(0), B, n->final_C, hw, st));
  FEDFR_TRY(bn_bwd(c, n->bn2, nullptr, c.g(0), A + last.out_off, Mf, nullptr, nullptr, 0, c.g(1), 0, Rows{nullptr, 0}, &last.bn3, A + last.c2_off, &pend));
  }
  int cur = 1;
  size_t dbg_off = 0;
  // fused SGD (NetSgd): [sgd_lo, sgd_hi) = parameter range whose gradients are complete on the main stream and whose weight gradients are
  // all queued on the weight-gradient stream; it is updated there behind the next fork (which orders it after everything queued on main).
  // Nothing reads those parameters again in this pass: dgrad uses the bf16 shadows, BatchNorm backward its own layer's gamma / slope.
  long long sgd_lo = -1, sgd_hi = -1;
  if (sgd && !n->block_only) {
    FEDFR_REQUIRE(sgd->params == params && sgd->shadow == shadow && sgd->mom, "net_backward: fused SGD wants the pass's own parameter / shadow buffers");
    sgd->done_from = n->trainable_count;
    sgd_lo = n->bn2.g_off; sgd_hi = n->trainable_count;     // bn2, fc, features: their gradients were written at the top of this pass
  }
  auto sgd_flush = [&]() -> int {
    if (sgd_lo < 0 || sgd_lo >= sgd_hi) return FEDFR_OK;
    FEDFR_REQUIRE((sgd_lo & 3) == 0, "net_backward: fused SGD range not 16-byte aligned");
    FEDFR_TRY(optim_sgd(sgd->params + sgd_lo, grads + sgd_lo, sgd->mom + sgd_lo, sgd->shadow + sgd_lo, (size_t)(sgd_hi - sgd_lo), sgd->lr, sgd->mu,
                        sgd->wd, sgd->first, wst, sgd->gscale, sgd->overflow));
    sgd->done_from = sgd_lo;
    sgd_hi = sgd_lo; sgd_lo = -1;
    return FEDFR_OK;
  };
  hipEvent_t wdone[kWgradDepth] = {};              // "all weight GEMMs of the block of this generation have finished"
  WgradPending pend_w;                             // a paired weight-gradient launch whose slabs the next one sums (wgrad9p.hip)
  for (int bi = (int)n->blocks.size() - 1; bi >= 0; --bi) {
    const BlockD& k = n->blocks[bi];
    const int Mi = B * k.Hin * k.Hin, Mo = B * k.Hout * k.Hout;
    const int par = bi % g_wgrad_depth;
    const bf16_t* g = c.g(cur);
    bf16_t* gin = c.g(cur ^ 1);
    FEDFR_TRY(dbg_capture(g, (size_t)Mo * k.Cout, &dbg_off, st));
    bf16_t *dc2 = c.tw(0, par), *da2 = c.t(1), *dc1 = c.tw(1, par), *da1 = c.t(3), *dd = c.tw(2, par), *dxd = c.t(5);
    fk.wait(st, wdone[par]);                         // the weight GEMMs kWgradDepth blocks ago were the last readers of dc2/dc1/dd[par]
    // out = bn3(c2) + identity
    FEDFR_TRY(bn_bwd(c, k.bn3, nullptr, g, A + k.c2_off, Mo, nullptr, nullptr, 0, dc2, 0, pend));
    pend = Rows{nullptr, 0};
    int f2 = 0, f1 = 0;
    FEDFR_TRY(conv_dgrad(c, k.conv2, dc2, da2, &k.bn2, A + k.c1_off, params + k.alpha_off, &f2));
    // a2 = prelu(bn2(c1))
    FEDFR_TRY(bn_bwd(c, k.bn2, params + k.alpha_off, da2, A + k.c1_off, Mi, nullptr, nullptr, 0, dc1, k.alpha_off, Rows{c.part(), f2}));
    // identity path first (its BN reduction uses the shared partial buffer), then conv1's dgrad whose epilogue may
    // leave bn1's partial sums there for the bn_bwd that follows immediately
    if (k.has_ds) {
      FEDFR_TRY(bn_bwd(c, k.bnds, nullptr, g, A + k.d_off, Mo, nullptr, nullptr, 0, dd, 0));
    }
    // ONE fork per block: every event record costs the main stream a ~8 us bubble (kernel trace), so the block's two or three
    // weight-gradient GEMMs are released together once their last operand (dc2, dc1, dd) exists
    fk.order(st, wst);
    if (sgd_lo >= 0 && sgd_lo < sgd_hi) FEDFR_TRY(wgrad_flush(c, &pend_w, wst));      // the fused SGD reads the finished stage's gradients
    FEDFR_TRY(sgd_flush());
    FEDFR_TRY(conv_wgrad2(c, k.conv2, A + k.a2_off, dc2, k.conv1, A + k.a1_off, dc1, wst, &pend_w));
    if (k.has_ds) {
      FEDFR_TRY(wgrad_flush(c, &pend_w, wst));
      FEDFR_TRY(conv_wgrad(c, k.ds, A + k.x_off, dd, wst));
      FEDFR_TRY(conv_dgrad(c, k.ds, dd, dxd));
    }
    wdone[par] = fk.mark(wst);
    FEDFR_TRY(conv_dgrad(c, k.conv1, dc1, da1, &k.bn1, A + k.x_off, nullptr, &f1));
    // a1 = bn1(x)
    const BlockD* prev = bi > 0 ? &n->blocks[bi - 1] : nullptr;      // its bn3 consumes gin next
    if (k.has_ds && !prev && !n->block_only && g_stem_bnred) {
      // the first block: its input gradient is the dy of the STEM's BatchNorm (+PReLU) backward, reduced here (one pass over two 205 MB tensors fewer)
      FEDFR_TRY(bn_bwd(c, k.bn1, nullptr, da1, A + k.x_off, Mi, nullptr, dxd, k.Hin, gin, 0, Rows{c.part(), f1}, &n->stem_bn, A + n->c0_off, &pend,
                       params + n->stem_alpha_off));
    } else if (k.has_ds) {
      FEDFR_TRY(bn_bwd(c, k.bn1, nullptr, da1, A + k.x_off, Mi, nullptr, dxd, k.Hin, gin, 0, Rows{c.part(), f1}, prev ? &prev->bn3 : nullptr,
                       prev ? A + prev->c2_off : nullptr, &pend));
    } else {
      FEDFR_TRY(bn_bwd(c, k.bn1, nullptr, da1, A + k.x_off, Mi, g, nullptr, 0, gin, 0, Rows{c.part(), f1}, prev ? &prev->bn3 : nullptr,
                       prev ? A + prev->c2_off : nullptr, &pend));
    }
    cur ^= 1;
    if (sgd && !n->block_only && k.has_ds && bi > 0) sgd_lo = k.bn1.g_off;      // a stage is complete: its range goes out behind the next fork
  }
  FEDFR_TRY(wgrad_flush(c, &pend_w, wst));
  // join: callers see all grads.  It sits at the very END: neither the stem's BatchNorm backward (two passes over 205 MB tensors; its dz goes to
  // t(1) — da2 of the blocks, main stream only — instead of t(0), which block 0's weight gradient may still be reading) nor the stem's
  // weight gradient (own partials, ws_stem) waits for the last weight gradients of stage 1
  if (n->block_only) fk.order(wst, st);
  const int M0 = B * HW * HW;
  FEDFR_TRY(dbg_capture(c.g(cur), (size_t)M0 * n->blocks.front().Cin, &dbg_off, st));
  if (n->block_only) {                               // the gradient wrt the block input stays readable in the arena
    if (hipMemcpyAsync(A + n->dx_off, c.g(cur), (size_t)M0 * n->blocks.front().Cin * 2, hipMemcpyDeviceToDevice, st) != hipSuccess) {
      fedfr_set_error("net_backward: hipMemcpyAsync failed");
      return FEDFR_ERR_HIP;
    }
  } else {
  // ---- stem: a0 = prelu(bn1(conv1(x))) ----
  bf16_t* dz0 = c.t(1);
  const bool fuse = g_stem_fuse_wgrad != 0;
  FEDFR_TRY(bn_bwd(c, n->stem_bn, params + n->stem_alpha_off, c.g(cur), A + n->c0_off, M0, nullptr, nullptr, 0, dz0, n->stem_alpha_off, pend, nullptr, nullptr,
                   nullptr, nullptr, fuse));
  if (fuse)     // the only reader of d(conv output) is the weight gradient: it applies the backward to its operand tile (two 205 MB streams fewer at batch 128)
    FEDFR_TRY(ew_stem_wgrad(x, c.g(cur), grads + n->stem.w_off, reinterpret_cast<float*>(ws + n->ws_stem), B, HW, HW, st, A + n->c0_off, c.coef(),
                            c.save(n->stem_bn, 0), c.save(n->stem_bn, 1), params + n->stem_alpha_off));
  else
  FEDFR_TRY(ew_stem_wgrad(x, dz0, grads + n->stem.w_off, reinterpret_cast<float*>(ws + n->ws_stem), B, HW, HW, st));
  fk.order(wst, st);
  }
  if (!fk.ok) {
    fedfr_set_error("net_backward: HIP event record/wait failed");
    return FEDFR_ERR_HIP;
  }
  return FEDFR_OK;
}

#include "net_sph.inc"
