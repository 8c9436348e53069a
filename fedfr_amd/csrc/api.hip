// C-ABI exports (include/fedfr_hip.h): thin argument-checking wrappers over the internal launchers.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "../../include/fedfr_hip.h"
#include "common.h"
#include "ew.h"
#include "gemm.h"
#include "head.h"
#include "net.h"
#include "optim.h"

static thread_local char g_err[512] = "";
extern int g_tn_use_tr;
extern int g_fuse_bnbwd;
extern int g_tn_glds;
extern int g_nt_glds;
extern int g_wgrad_pair_reduce;
extern int g_wgrad9;
extern int g_conv_c64p;
extern int g_bn_sliced;
extern int g_wgrad9p, g_wgrad9p_bg;
extern int g_conv28_tpw2;
extern int g_eval_fuse;
extern int g_dgrad_parity;
extern int g_fuse_bnred_next;
extern int g_sph_fin_multi;
extern int g_sph_fuse_prelu_bwd;
extern int g_sph_pair_wgrad;
#ifdef FEDFR_DEBUG
extern int g_dbg_skip;
#endif
extern int g_c64p_bnbwd;
extern int g_stem_bnred, g_stem_fuse_wgrad;
extern int g_fwd_xmom;
extern int g_fuse_bnbwd28;

void fedfr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
int fedfr_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fedfr_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return FEDFR_ERR_HIP;
  }
  return FEDFR_OK;
}
#define ST(s) reinterpret_cast<hipStream_t>(s)
#define BF(p) reinterpret_cast<const bf16_t*>(p)
#define BFM(p) reinterpret_cast<bf16_t*>(p)

extern "C" {

int fedfr_version(void) { return 100; }
int fedfr_storage_dtype(void) { return FEDFR_FP16 ? 1 : 0; }
const char* fedfr_last_error_string(void) { return g_err; }
// ---- process-global switches (tuning / validation).  ONE table: name, variable, clamp (the default is the variable's initialiser, read before the first change), so that fedfr_set_option, fedfr_get_option and
// fedfr_option_info (what bench.py lists as `options_non_default`) cannot disagree.  kind: 0 = boolean (any non-zero value -> 1),
// 1 = integer clamped to [lo, hi], 2 = integer taken as given.  Switches that produce WRONG results (timing experiments) exist only in a
// -DFEDFR_DEBUG build of the library.
namespace {
struct OptRow { const char* name; int* var; int kind; int lo, hi; };
const OptRow kOptions[] = {
    {"tn_use_tr", &g_tn_use_tr, 0, 0, 1},                  // 0: scalar-LDS fallback fragments of the register-staged TN kernel (validation)
    {"fuse_bnred_next", &g_fuse_bnred_next, 0, 0, 1},
    {"dgrad_parity", &g_dgrad_parity, 1, 0, 2},
    {"wgrad_pair_reduce", &g_wgrad_pair_reduce, 0, 0, 1},
    {"nt_glds", &g_nt_glds, 1, 0, 15},                    // 0 register-staged NT kernel, 1..4 LDS-DMA operand ring where it pays, + 8 everywhere it can (gemm_nt_glds.hip)
    {"tn_glds", &g_tn_glds, 2, 0, 0},                      // 0 register-staged kernel, 1 LDS-DMA with 4 waves, 2 LDS-DMA with 8 waves
    {"eval_fuse", &g_eval_fuse, 0, 0, 1},                  // eval-mode forward: BatchNorm (+PReLU, +identity, +next bn1) in the conv epilogues
    {"wgrad9", &g_wgrad9, 0, 0, 1},                        // nine-tap weight-gradient kernel for 3x3 / stride-1 layers
    {"fuse_bnbwd", &g_fuse_bnbwd, 1, 0, 2},
    {"conv_c64p", &g_conv_c64p, 0, 0, 1},                  // persistent register-resident-weights kernel for the 64 -> 64 channel 3x3 layers (112x112 / 56x56)
    {"bn_sliced", &g_bn_sliced, 0, 0, 1},                  // channel-sliced BatchNorm passes without finalize launches (bn_sliced.hip)
    {"conv28_tpw2", &g_conv28_tpw2, 1, 0, 2},              // 28x28 convs with two image tiles per workgroup (1: forward, 2: dgrad too)
    {"wgrad9p_bg", &g_wgrad9p_bg, 0, 0, 1},                // a paired launch sums the PREVIOUS pair's split-K slabs beside its own work (0: stand-alone reduce_slabs launches)
    {"wgrad9p", &g_wgrad9p, 0, 0, 1},                      // paired 64 x 64 nine-tap weight-gradient kernel for the two 3x3 / stride-1 layers of a residual block
    {"fuse_bnbwd28", &g_fuse_bnbwd28, 0, 0, 1},
    {"fwd_xmom", &g_fwd_xmom, 0, 0, 1},
    {"stem_bnred", &g_stem_bnred, 0, 0, 1},
    {"stem_fuse_wgrad", &g_stem_fuse_wgrad, 0, 0, 1},      // the stem's BatchNorm + PReLU backward applied inside its weight-gradient kernel (no apply pass, no d(conv output) tensor)
    {"sph_fuse_prelu_bwd", &g_sph_fuse_prelu_bwd, 0, 0, 1},
    {"sph_fin_multi", &g_sph_fin_multi, 0, 0, 1},
    {"sph_pair_wgrad", &g_sph_pair_wgrad, 0, 0, 1},
    {"c64p_bnbwd", &g_c64p_bnbwd, 0, 0, 1},
#ifdef FEDFR_DEBUG
    {"dbg_skip", &g_dbg_skip, 2, 0, 0},                    // WRONG results: 1 = no weight-gradient launches of the residual blocks' 3x3 convs (timing bound)
#endif
};
constexpr int kNumOptions = (int)(sizeof(kOptions) / sizeof(kOptions[0]));
int g_opt_defaults[kNumOptions];
bool g_opt_defaults_set = false;
void opt_capture_defaults() {
  if (g_opt_defaults_set) return;
  for (int i = 0; i < kNumOptions; ++i) g_opt_defaults[i] = *kOptions[i].var;
  g_opt_defaults_set = true;
}
const OptRow* opt_find(const char* name, int* idx = nullptr) {
  if (!name) return nullptr;
  for (int i = 0; i < kNumOptions; ++i)
    if (!strcmp(name, kOptions[i].name)) {
      if (idx) *idx = i;
      return &kOptions[i];
    }
  return nullptr;
}
}  // namespace

int fedfr_set_option(const char* name, int value) {
  opt_capture_defaults();
  const OptRow* o = opt_find(name);
  if (!o) {
    fedfr_set_error("set_option: unknown option '%s'", name ? name : "(null)");
    return FEDFR_ERR_ARG;
  }
  if (o->kind == 0) value = value ? 1 : 0;
  else if (o->kind == 1) value = value < o->lo ? o->lo : value > o->hi ? o->hi : value;
  *o->var = value;
  return FEDFR_OK;
}

// current value of a switch (so that a caller that changes one for a while can put the previous value back)
int fedfr_get_option(const char* name, int* value) {
  FEDFR_REQUIRE(name && value, "get_option: null argument");
  const OptRow* o = opt_find(name);
  if (!o) {
    fedfr_set_error("get_option: unknown option '%s'", name);
    return FEDFR_ERR_ARG;
  }
  *value = *o->var;
  return FEDFR_OK;
}

// enumeration of the switches: number of rows / row i = (name, current value, value the library starts with).  bench.py prints every row whose
// value differs from its default next to the number it measures.
int fedfr_option_count(void) { return kNumOptions; }
int fedfr_option_info(int index, const char** name, int* value, int* default_value) {
  FEDFR_REQUIRE(index >= 0 && index < kNumOptions && name && value && default_value, "option_info: bad index %d", index);
  opt_capture_defaults();
  *name = kOptions[index].name;
  *value = *kOptions[index].var;
  *default_value = g_opt_defaults[index];
  return FEDFR_OK;
}

int fedfr_profile_enable(int on) {
  gemm_profile_enable(on);
  return FEDFR_OK;
}
int fedfr_profile_read_bytes(int slot, double* bytes) {
  FEDFR_REQUIRE(bytes && gemm_profile_read_bytes(slot, bytes) == 0, "profile_read_bytes: bad slot %d", slot);
  return FEDFR_OK;
}
int fedfr_profile_read(int slot, double* total_ms, long long* launches, double* flops) {
  FEDFR_REQUIRE(total_ms && launches && flops, "profile_read: null");
  const int rc = gemm_profile_read(slot, total_ms, launches, flops);
  FEDFR_REQUIRE(rc == 0, "profile_read: slot %d failed (%d) — synchronise the stream first", slot, rc);
  return FEDFR_OK;
}

// ---- net -------------------------------------------------------------------------------------------
fedfr_net_t* fedfr_net_create(const int* layers4, int batch, int in_hw, int num_features) {
  if (!layers4) {
    fedfr_set_error("net_create: layers is null");
    return nullptr;
  }
  return net_create(layers4, batch, in_hw, num_features);
}
fedfr_net_t* fedfr_block_create(int cin, int cout, int stride, int hin, int batch) {
  return net_create_block(cin, cout, stride, hin, batch);
}
fedfr_net_t* fedfr_net_create_sphere(int type, int batch) { return net_create_sphere(type, batch); }
extern bf16_t* g_dbg_grads;
extern size_t g_dbg_grads_elems;
int fedfr_net_debug_capture(uint16_t* buf, size_t elems) {
  g_dbg_grads = BFM(buf);
  g_dbg_grads_elems = buf ? elems : 0;
  return FEDFR_OK;
}
int fedfr_net_set_dropout(fedfr_net_t* n, float p, unsigned long long seed, long long* mask_offset_bytes) {
  FEDFR_REQUIRE(n && !n->block_only && p >= 0.f && p < 1.f, "net_set_dropout: need a network plan and 0 <= p < 1");
  FEDFR_REQUIRE(((size_t)n->B * n->fc_in) % 8 == 0, "net_set_dropout: batch * fc_in must be a multiple of 8");
  n->dropout_p = p;
  n->dropout_seed = seed;
  n->dropout_step = 0;
  if (mask_offset_bytes) *mask_offset_bytes = n->mask_off_bytes;
  return FEDFR_OK;
}
int fedfr_net_set_dropout_step(fedfr_net_t* n, unsigned long long step) {
  FEDFR_REQUIRE(n && !n->block_only, "net_set_dropout_step: need a network plan");
  n->dropout_step = step;
  return FEDFR_OK;
}
void fedfr_net_destroy(fedfr_net_t* net) {
  if (!net) return;
  for (auto e : net->events) (void)hipEventDestroy(e);
  delete net;
}
int fedfr_net_query(const fedfr_net_t* n, int what, long long* out) {
  FEDFR_REQUIRE(n && out, "net_query: null");
  switch (what) {
    case FEDFR_Q_PARAM_COUNT: *out = n->param_count; break;
    case FEDFR_Q_TRAINABLE_COUNT: *out = n->trainable_count; break;
    case FEDFR_Q_BUFFER_COUNT: *out = n->buffer_count; break;
    case FEDFR_Q_NBT_COUNT: *out = n->nbt_count; break;
    case FEDFR_Q_SHADOW_COUNT: *out = n->shadow_count; break;
    case FEDFR_Q_ACT_BYTES: *out = n->act_bytes; break;
    case FEDFR_Q_WS_BYTES: *out = (long long)n->ws_bytes; break;
    case FEDFR_Q_NUM_TENSORS: *out = (long long)n->tensors.size(); break;
    case FEDFR_Q_FC_IN: *out = n->fc_in; break;
    default: fedfr_set_error("net_query: unknown key %d", what); return FEDFR_ERR_ARG;
  }
  return FEDFR_OK;
}
int fedfr_net_tensor_info(const fedfr_net_t* n, int i, char* name, int name_cap, int* kind, int* region, long long* offset,
                          int* ndim, int* shape4) {
  FEDFR_REQUIRE(n && i >= 0 && i < (int)n->tensors.size() && name && name_cap > 0 && kind && region && offset && ndim && shape4,
                "net_tensor_info: bad args");
  const NetTensor& t = n->tensors[i];
  snprintf(name, name_cap, "%s", t.name.c_str());
  *kind = t.kind; *region = t.region; *offset = t.offset; *ndim = t.ndim;
  for (int k = 0; k < 4; ++k) shape4[k] = t.shape[k];
  return FEDFR_OK;
}
int fedfr_net_act_info(const fedfr_net_t* n, int block, int which, long long* offset, int* rows, int* channels) {
  FEDFR_REQUIRE(n && offset && rows && channels, "net_act_info: null");
  const int B = n->B;
  if (block < 0) {          // which: 0 stem conv out (c0), 1 stem activation (a0), 2 flattened bn2 output t [B][fc_in]
    FEDFR_REQUIRE(which >= 0 && which <= 2 && !n->block_only, "net_act_info: bad stem selector");
    if (which == 2) { *offset = n->t_off; *rows = B; *channels = n->fc_in; return FEDFR_OK; }
    *offset = which == 0 ? n->c0_off : n->a0_off; *rows = B * n->HW * n->HW; *channels = 64;
    return FEDFR_OK;
  }
  FEDFR_REQUIRE(block < (int)n->blocks.size() && which >= 0 && which <= 7, "net_act_info: bad selector");
  const BlockD& k = n->blocks[block];
  const int Mi = B * k.Hin * k.Hin, Mo = B * k.Hout * k.Hout;
  if (which == 7) {          // gradient wrt the input of a lone block (fedfr_block_create plans only)
    FEDFR_REQUIRE(n->block_only, "net_act_info: selector 7 (dx) exists only for fedfr_block_create plans");
    *offset = n->dx_off; *rows = Mi; *channels = k.Cin;
    return FEDFR_OK;
  }
  switch (which) {           // 0 x, 1 a1, 2 c1, 3 a2, 4 c2, 5 d, 6 out
    case 0: *offset = k.x_off; *rows = Mi; *channels = k.Cin; break;
    case 1: *offset = k.a1_off; *rows = Mi; *channels = k.Cin; break;
    case 2: *offset = k.c1_off; *rows = Mi; *channels = k.Cout; break;
    case 3: *offset = k.a2_off; *rows = Mi; *channels = k.Cout; break;
    case 4: *offset = k.c2_off; *rows = Mo; *channels = k.Cout; break;
    case 5: *offset = k.d_off; *rows = Mo; *channels = k.Cout; break;
    default: *offset = k.out_off; *rows = Mo; *channels = k.Cout; break;
  }
  return FEDFR_OK;
}
int fedfr_net_prepare_weights(const fedfr_net_t* n, const float* params, uint16_t* shadow, int fwd_shadow_too, void* stream) {
  return net_prepare_weights(n, params, BFM(shadow), fwd_shadow_too, ST(stream));
}
int fedfr_net_forward(const fedfr_net_t* n, const float* x, const float* params, float* bufs, const uint16_t* shadow, void* act,
                      void* ws, float* feats, int training, void* stream) {
  return net_forward(n, x, params, bufs, BF(shadow), (unsigned char*)act, (unsigned char*)ws, feats, training, ST(stream));
}
int fedfr_net_backward(const fedfr_net_t* n, const float* x, const float* dfeats, const float* params, const uint16_t* shadow,
                       void* act, void* ws, float* grads, void* stream) {
  return net_backward(n, x, dfeats, params, BF(shadow), (unsigned char*)act, (unsigned char*)ws, grads, ST(stream), nullptr);
}
int fedfr_net_backward2(const fedfr_net_t* n, const float* x, const float* dfeats, const float* params, const uint16_t* shadow,
                        void* act, void* ws, float* grads, void* stream, void* aux_stream) {
  FEDFR_REQUIRE(aux_stream == nullptr || aux_stream != stream, "net_backward2: aux_stream must differ from stream (pass NULL for single-stream)");
  return net_backward(n, x, dfeats, params, BF(shadow), (unsigned char*)act, (unsigned char*)ws, grads, ST(stream), ST(aux_stream));
}

int fedfr_net_backward2_sgd(const fedfr_net_t* n, const float* x, const float* dfeats, float* params, uint16_t* shadow, void* act, void* ws,
                            float* grads, float* momentum, float lr, float mu, float wd, int first, long long* done_from, void* stream,
                            void* aux_stream) {
  FEDFR_REQUIRE(aux_stream == nullptr || aux_stream != stream, "net_backward2_sgd: aux_stream must differ from stream (pass NULL for single-stream)");
  FEDFR_REQUIRE(momentum && done_from, "net_backward2_sgd: null momentum buffer / done_from");
  NetSgd sg{params, BFM(shadow), momentum, lr, mu, wd, first, 0};
  const int rc = net_backward(n, x, dfeats, params, BF(shadow), (unsigned char*)act, (unsigned char*)ws, grads, ST(stream), ST(aux_stream), &sg);
  *done_from = sg.done_from;
  return rc;
}
int fedfr_net_backward2_sgd_scaled(const fedfr_net_t* n, const float* x, const float* dfeats, float* params, uint16_t* shadow, void* act,
                                   void* ws, float* grads, float* momentum, float lr, float mu, float wd, int first, float grad_scale,
                                   unsigned* overflow, long long* done_from, void* stream, void* aux_stream) {
  FEDFR_REQUIRE(aux_stream == nullptr || aux_stream != stream, "net_backward2_sgd_scaled: aux_stream must differ from stream (pass NULL for single-stream)");
  FEDFR_REQUIRE(momentum && done_from, "net_backward2_sgd_scaled: null momentum buffer / done_from");
  FEDFR_REQUIRE(grad_scale > 0.f, "net_backward2_sgd_scaled: grad_scale must be positive");
  NetSgd sg{params, BFM(shadow), momentum, lr, mu, wd, first, 0, grad_scale, overflow};
  const int rc = net_backward(n, x, dfeats, params, BF(shadow), (unsigned char*)act, (unsigned char*)ws, grads, ST(stream), ST(aux_stream), &sg);
  *done_from = sg.done_from;
  return rc;
}

// ---- fp32 validation path (net_f32.hip) --------------------------------------------------------------------------------------
size_t fedfr_net_f32_arena_floats(const fedfr_net_t* n) { return n ? net_f32_arena_floats(n) : 0; }
size_t fedfr_net_f32_ws_floats(const fedfr_net_t* n) { return n ? net_f32_ws_floats(n) : 0; }
int fedfr_net_f32_forward(const fedfr_net_t* n, const float* x, const float* params, float* bufs, float* arena, float* ws, float* feats,
                          int training, void* stream) {
  return net_f32_forward(n, x, params, bufs, arena, ws, feats, training, ST(stream));
}
int fedfr_net_f32_backward(const fedfr_net_t* n, const float* dfeats, const float* params, float* arena, float* ws, float* grads, void* stream) {
  return net_f32_backward(n, dfeats, params, arena, ws, grads, ST(stream));
}

// ---- single convolutions ---------------------------------------------------------------------------------
static int conv_args_ok(int batch, int hin, int cin, int cout, int ksize, int stride) {
  FEDFR_REQUIRE(batch > 0 && hin > 0 && (cin % 64) == 0 && (cout % 64) == 0 && (ksize == 1 || ksize == 3) &&
                    (stride == 1 || stride == 2) && hin % stride == 0,
                "conv2d: unsupported geometry batch=%d hin=%d cin=%d cout=%d k=%d s=%d", batch, hin, cin, cout, ksize, stride);
  return FEDFR_OK;
}
int fedfr_conv2d_stat_rows(int batch, int hout, int cout) { return gemm_nt_stat_rows(batch * hout * hout, cout); }
int fedfr_conv2d_fwd(const uint16_t* x, const uint16_t* w, uint16_t* y, float* stats, int batch, int hin, int cin, int cout,
                     int ksize, int stride, void* stream) {
  FEDFR_TRY(conv_args_ok(batch, hin, cin, cout, ksize, stride));
  FEDFR_REQUIRE(x && w && y, "conv2d_fwd: null tensor");
  const int hout = hin / stride;
  GemmNT p{};
  p.A = BF(x); p.B = BF(w); p.M = batch * hout * hout; p.N = cout; p.K = ksize * ksize * cin;
  p.mode = 1; p.H = hin; p.W = hin; p.C = cin; p.Ho = hout; p.Wo = hout; p.S = ksize; p.stride = stride;
  p.pad = ksize == 3 ? 1 : 0; p.up = 1; p.Cb = BFM(y); p.ldc = cout; p.stats = stats;
  return gemm_nt_launch(p, 1, ST(stream));
}
int fedfr_conv2d_fwd_moments(const uint16_t* x, const uint16_t* w, uint16_t* y, int batch, int hin, int cin, int cout, const uint16_t* other,
                             float* partials, int* rows, void* stream) {
  FEDFR_TRY(conv_args_ok(batch, hin, cin, cout, 3, 1));
  FEDFR_REQUIRE(x && w && y && other && partials && rows, "conv2d_fwd_moments: null argument");
  GemmNT p{};
  p.A = BF(x); p.B = BF(w); p.M = batch * hin * hin; p.N = cout; p.K = 9 * cin;
  p.mode = 1; p.H = hin; p.W = hin; p.C = cin; p.Ho = hin; p.Wo = hin; p.S = 3; p.stride = 1; p.pad = 1; p.up = 1;
  p.Cb = BFM(y); p.ldc = cout;
  p.bx = BF(other); p.bmean = partials; p.brstd = partials;   // (any readable fp32 array of >= cout elements: read, never used in this mode)
  p.bpart = partials; p.bmom = 1;
  *rows = 0;
  p.bwd_fused = rows;
  return gemm_nt_launch(p, 1, ST(stream));
}
int fedfr_bn_apply2_sliced_ok(int M, int C, int rows) { return ew_bn_apply2_sliced_ok(M, C, rows) ? 1 : 0; }
int fedfr_bn_apply2_sliced(const float* partials, int P, double count, float momentum, float eps, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float* scale, float* shift, float* save_mean, float* save_rstd,
                           const float* x2_mean, const float* x2_rstd, const float* next_gamma, const float* next_beta,
                           float* next_running_mean, float* next_running_var, float* next_scale, float* next_shift, float* next_save_mean,
                           float* next_save_rstd, const uint16_t* x1, const uint16_t* x2, uint16_t* y, uint16_t* y2, int M, int C,
                           void* stream) {
  BnApply2S a{};
  a.part = partials; a.P = P; a.count = count; a.momentum = momentum; a.eps = eps;
  a.gamma = gamma; a.beta = beta; a.rm = running_mean; a.rv = running_var; a.scale = scale; a.shift = shift; a.mean = save_mean; a.rstd = save_rstd;
  a.xmean = x2_mean; a.xrstd = x2_rstd;
  a.ngamma = next_gamma; a.nbeta = next_beta; a.nrm = next_running_mean; a.nrv = next_running_var; a.nscale = next_scale; a.nshift = next_shift;
  a.nmean = next_save_mean; a.nrstd = next_save_rstd;
  a.x1 = BF(x1); a.x2 = BF(x2); a.y = BFM(y); a.y2 = BFM(y2); a.M = M; a.C = C;
  return ew_bn_apply2_sliced(a, ST(stream));
}
int fedfr_conv2d_dgrad(const uint16_t* dy, const uint16_t* wd, uint16_t* dx, int batch, int hin, int cin, int cout, int ksize,
                       int stride, void* stream) {
  FEDFR_TRY(conv_args_ok(batch, hin, cin, cout, ksize, stride));
  FEDFR_REQUIRE(dy && wd && dx, "conv2d_dgrad: null tensor");
  const int hout = hin / stride;
  GemmNT p{};
  p.A = BF(dy); p.B = BF(wd); p.N = cin; p.K = ksize * ksize * cout; p.Cb = BFM(dx); p.ldc = cin;
  if (ksize == 1) {   // compact result at the OUTPUT resolution: dx[m_out][cin]
    p.mode = 0; p.M = batch * hout * hout; p.lda = cout;
  } else {
    p.mode = 1; p.M = batch * hin * hin; p.H = hout; p.W = hout; p.C = cout; p.Ho = hin; p.Wo = hin; p.S = 3;
    p.stride = 1; p.pad = 1; p.up = stride;
  }
  return gemm_nt_launch(p, 1, ST(stream));
}
int fedfr_conv2d_dgrad_bnbwd(const uint16_t* dy, const uint16_t* wd, uint16_t* dx, int batch, int hin, int cin, int cout, int ksize,
                             int stride, const uint16_t* bn_x, const float* mean, const float* rstd, const float* gamma,
                             const float* beta, const float* alpha, float* partials, int* fused_rows, void* stream) {
  FEDFR_TRY(conv_args_ok(batch, hin, cin, cout, ksize, stride));
  FEDFR_REQUIRE(dy && wd && dx && bn_x && mean && rstd && partials && fused_rows && ksize == 3, "conv2d_dgrad_bnbwd: bad args");
  const int hout = hin / stride;
  GemmNT p{};
  p.A = BF(dy); p.B = BF(wd); p.N = cin; p.K = 9 * cout; p.Cb = BFM(dx); p.ldc = cin;
  p.mode = 1; p.M = batch * hin * hin; p.H = hout; p.W = hout; p.C = cout; p.Ho = hin; p.Wo = hin; p.S = 3;
  p.stride = 1; p.pad = 1; p.up = stride;
  p.bx = BF(bn_x); p.bmean = mean; p.brstd = rstd; p.bgamma = gamma; p.bbeta = beta; p.balpha = alpha; p.bpart = partials;
  *fused_rows = 0;
  p.bwd_fused = fused_rows;
  return gemm_nt_launch(p, 1, ST(stream));
}
size_t fedfr_conv2d_wgrad_ws_bytes(int batch, int hin, int cin, int cout, int ksize, int stride) {
  const int hout = hin / (stride > 0 ? stride : 1);
  const int NJ = ksize * ksize * cin;
  return (size_t)gemm_tn_max_splits(batch * hout * hout, cout, NJ, cin, hout, stride) * cout * NJ * sizeof(float);
}
int fedfr_conv2d_wgrad(const uint16_t* x, const uint16_t* dy, float* dw, void* ws, size_t ws_bytes, int batch, int hin, int cin,
                       int cout, int ksize, int stride, void* stream) {
  FEDFR_TRY(conv_args_ok(batch, hin, cin, cout, ksize, stride));
  FEDFR_REQUIRE(x && dy && dw, "conv2d_wgrad: null tensor");
  const int hout = hin / stride;
  GemmTN p{};
  p.P = BF(dy); p.Q = BF(x); p.Kp = batch * hout * hout; p.NI = cout; p.NJ = ksize * ksize * cin;
  p.mode = 1; p.H = hin; p.W = hin; p.C = cin; p.Ho = hout; p.Wo = hout; p.S = ksize; p.stride = stride;
  p.pad = ksize == 3 ? 1 : 0; p.ldp = cout; p.use_tr = g_tn_use_tr;
  const int splits = gemm_tn_pick_splits(p.Kp, p.NI, p.NJ, p.C, p.Wo, p.stride);
  if (splits == 1) {
    p.out = dw;
    return gemm_tn_launch(p, 1, ST(stream));
  }
  if (!ws || ws_bytes < (size_t)splits * p.NI * p.NJ * sizeof(float)) {
    fedfr_set_error("conv2d_wgrad: workspace too small (%zu bytes)", ws_bytes);
    return FEDFR_ERR_WORKSPACE;
  }
  p.out = (float*)ws;
  FEDFR_TRY(gemm_tn_launch(p, splits, ST(stream)));
  return ew_reduce_slabs(dw, (const float*)ws, splits, (size_t)p.NI * p.NJ, nullptr, 0, ST(stream));
}
int fedfr_conv2d_wgrad_pair(const uint16_t* xa, const uint16_t* dya, float* dwa, const uint16_t* xb, const uint16_t* dyb, float* dwb,
                            void* ws, size_t ws_bytes, int batch, int hin, int cin, int cout, int ksize, int stride, void* stream) {
  FEDFR_TRY(conv_args_ok(batch, hin, cin, cout, ksize, stride));
  FEDFR_REQUIRE(xa && dya && dwa && xb && dyb && dwb, "conv2d_wgrad_pair: null tensor");
  const int hout = hin / stride;
  GemmTN p{};
  p.Kp = batch * hout * hout; p.NI = cout; p.NJ = ksize * ksize * cin;
  p.mode = 1; p.H = hin; p.W = hin; p.C = cin; p.Ho = hout; p.Wo = hout; p.S = ksize; p.stride = stride;
  p.pad = ksize == 3 ? 1 : 0; p.ldp = cout; p.use_tr = g_tn_use_tr;
  const int splits = gemm_tn_pick_splits(p.Kp, p.NI, p.NJ, p.C, p.Wo, p.stride);
  const size_t one = (size_t)splits * p.NI * p.NJ * sizeof(float);
  GemmTN a = p, b = p;
  a.P = BF(dya); a.Q = BF(xa); b.P = BF(dyb); b.Q = BF(xb);
  if (gemm_tn_w9pair_ok(a, b)) {                            // 3x3 / stride-1 layers the paired nine-tap kernel takes (wgrad9p.hip)
    const int sp = gemm_tn_w9pair_splits(a);
    const size_t each = (size_t)sp * p.NI * p.NJ * sizeof(float);
    if (!ws || ws_bytes < 2 * each) {
      fedfr_set_error("conv2d_wgrad_pair: workspace too small (%zu bytes, need %zu)", ws_bytes, 2 * each);
      return FEDFR_ERR_WORKSPACE;
    }
    a.out = (float*)ws; b.out = (float*)((char*)ws + each);
    FEDFR_TRY(gemm_tn_launch_w9pair(a, b, sp, ST(stream)));
    FEDFR_TRY(ew_reduce_slabs(dwa, a.out, sp, (size_t)p.NI * p.NJ, nullptr, 0, ST(stream)));
    return ew_reduce_slabs(dwb, b.out, sp, (size_t)p.NI * p.NJ, nullptr, 0, ST(stream));
  }
  // shapes the paired kernel does not take: two ordinary launches
  FEDFR_TRY(fedfr_conv2d_wgrad(xa, dya, dwa, ws, ws_bytes, batch, hin, cin, cout, ksize, stride, stream));
  return fedfr_conv2d_wgrad(xb, dyb, dwb, ws, ws_bytes, batch, hin, cin, cout, ksize, stride, stream);
}
int fedfr_weight_shadows(const float* w, uint16_t* wb, uint16_t* wdb, int cout, int ksize, int cin, void* stream) {
  FEDFR_REQUIRE(w, "weight_shadows: null");
  if (wb) FEDFR_TRY(ew_cast_f32_bf16(w, BFM(wb), (size_t)cout * ksize * ksize * cin, ST(stream)));
  if (wdb) FEDFR_TRY(ew_weight_dgrad_shadow(w, BFM(wdb), cout, ksize, ksize, cin, ST(stream)));
  return FEDFR_OK;
}
int fedfr_gemm_nt(const uint16_t* A, const uint16_t* B, float* C, void* ws, size_t ws_bytes, int M, int N, int K, void* stream) {
  FEDFR_REQUIRE(A && B && C, "gemm_nt: null");
  GemmNT p{};
  p.A = BF(A); p.B = BF(B); p.M = M; p.N = N; p.K = K; p.mode = 0; p.lda = K;
  const int splits = gemm_nt_pick_splits(M, N, K);
  if (splits == 1) {
    p.Cf = C;
    return gemm_nt_launch(p, 1, ST(stream));
  }
  if (!ws || ws_bytes < (size_t)splits * M * N * sizeof(float)) {
    fedfr_set_error("gemm_nt: workspace too small: need %zu bytes", (size_t)splits * M * N * sizeof(float));
    return FEDFR_ERR_WORKSPACE;
  }
  p.Cf = (float*)ws;
  FEDFR_TRY(gemm_nt_launch(p, splits, ST(stream)));
  return ew_reduce_slabs(C, (const float*)ws, splits, (size_t)M * N, nullptr, 0, ST(stream));
}
int fedfr_gemm_tn(const uint16_t* P, const uint16_t* Q, float* C, int Kp, int NI, int NJ, void* stream) {
  GemmTN p{};
  p.P = BF(P); p.Q = BF(Q); p.Kp = Kp; p.NI = NI; p.NJ = NJ; p.mode = 0; p.ldp = NI; p.ldq = NJ; p.out = C; p.use_tr = g_tn_use_tr;
  return gemm_tn_launch(p, 1, ST(stream));
}
int fedfr_stem_stat_rows(int batch, int hw) { return ew_stem_stat_rows(batch, hw, hw); }
int fedfr_stem_fwd(const float* x, const float* w, uint16_t* y, float* stats, int batch, int hw, void* stream) {
  return ew_stem_fwd(x, w, BFM(y), stats, batch, hw, hw, ST(stream));
}
size_t fedfr_stem_wgrad_ws_bytes(int batch, int hw) { return (size_t)ew_stem_wgrad_blocks(batch, hw, hw) * 2048 * sizeof(float); }
int fedfr_stem_wgrad(const float* x, const uint16_t* dy, float* dw, void* ws, int batch, int hw, void* stream) {
  return ew_stem_wgrad(x, BF(dy), dw, (float*)ws, batch, hw, hw, ST(stream));
}

// ---- streams -------------------------------------------------------------------------------------------------
// A HIP stream of the LOWEST priority the device offers, for the weight-gradient stream of fedfr_net_backward2: workgroups of the
// critical path (forward, dgrad -> BatchNorm-backward chain, on the caller's ordinary stream) are then dispatched first whenever
// both streams have work.  (torch.cuda.Stream clamps priorities to [-1, 0]; HIP offers +1.)  Destroy with fedfr_stream_destroy.
int fedfr_stream_create_low_priority(void** stream) {
  FEDFR_REQUIRE(stream, "stream_create_low_priority: null");
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
  hipStream_t s = nullptr;
  const hipError_t e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least);
  if (e != hipSuccess) {
    fedfr_set_error("hipStreamCreateWithPriority: %s", hipGetErrorString(e));
    return FEDFR_ERR_HIP;
  }
  *stream = (void*)s;
  return FEDFR_OK;
}
int fedfr_stream_destroy(void* stream) {
  if (stream && hipStreamDestroy((hipStream_t)stream) != hipSuccess) {
    fedfr_set_error("hipStreamDestroy failed");
    return FEDFR_ERR_HIP;
  }
  return FEDFR_OK;
}

// ---- BN ----------------------------------------------------------------------------------------------------
int fedfr_bn_finalize(const float* partials, int P, int C, double count, const float* gamma, const float* beta, float* rm,
                      float* rv, float momentum, float eps, float* scale, float* shift, float* save_mean, float* save_rstd,
                      float* tmp, void* stream) {
  return ew_bn_finalize(partials, P, C, count, gamma, beta, rm, rv, momentum, eps, scale, shift, save_mean, save_rstd, tmp, ST(stream));
}
int fedfr_bn_apply_stat_rows(int M, int C) { return ew_bn_apply_grid(M, C); }
int fedfr_bn_apply(const uint16_t* x1, const float* sc1, const float* sh1, const float* alpha, const uint16_t* x2, const float* sc2,
                   const float* sh2, uint16_t* y, int M, int C, int nchw_hw, float* stats, void* stream) {
  BnApply a{};
  a.x1 = BF(x1); a.sc1 = sc1; a.sh1 = sh1; a.alpha = alpha; a.x2 = BF(x2); a.sc2 = sc2; a.sh2 = sh2; a.y = BFM(y);
  a.M = M; a.C = C; a.nchw_hw = nchw_hw; a.stats = stats;
  return ew_bn_apply(a, ST(stream));
}
int fedfr_bn_bwd_rows(int M, int C) { return ew_bn_bwd_grid(M, C); }
int fedfr_bn_bwd(const uint16_t* dy, const uint16_t* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                 const float* alpha, int M, int C, float* partials, float* coef, float* dgamma, float* dbeta, float* dalpha,
                 const uint16_t* add, const uint16_t* add_up, int H, uint16_t* dx, void* stream) {
  BnBwd p{};
  p.dy = BF(dy); p.x = BF(x); p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.beta = beta; p.alpha = alpha; p.M = M; p.C = C;
  p.partials = partials; p.coef = coef; p.add = BF(add); p.add_up = BF(add_up); p.H = H; p.W = H; p.dx = BFM(dx);
  FEDFR_TRY(ew_bn_bwd_reduce(p, ST(stream)));
  FEDFR_TRY(ew_bn_bwd_finalize(partials, ew_bn_bwd_grid(M, C), C, (double)M, gamma, mean, rstd, dgamma, dbeta, dalpha, coef, ST(stream)));
  return ew_bn_bwd_apply(p, ST(stream));
}

int fedfr_bn_sliced_rows(int M, int C, int backward) { return ew_bn_sliced_rows(M, C, backward != 0); }
int fedfr_bn_sliced_ok(int M, int C, int rows_in, int backward) { return ew_bn_sliced_ok(M, C, rows_in, backward != 0) ? 1 : 0; }
int fedfr_bn_apply_sliced(const float* partials, int P, double count, const float* gamma, const float* beta, float* rm, float* rv,
                          float momentum, float eps, float* scale, float* shift, float* save_mean, float* save_rstd, const uint16_t* x1,
                          const float* alpha, const uint16_t* x2, uint16_t* y, int M, int C, float* stats, void* stream) {
  FEDFR_REQUIRE(ew_bn_sliced_ok(M, C, P, false), "bn_apply_sliced: shape M=%d C=%d with %d partial rows is not served (fedfr_bn_sliced_ok)", M, C, P);
  BnApplyS a{};
  a.part = partials; a.P = P; a.count = count; a.gamma = gamma; a.beta = beta; a.rm = rm; a.rv = rv; a.momentum = momentum; a.eps = eps;
  a.scale = scale; a.shift = shift; a.mean = save_mean; a.rstd = save_rstd; a.x1 = BF(x1); a.alpha = alpha; a.x2 = BF(x2); a.y = BFM(y);
  a.M = M; a.C = C; a.stats = stats;
  return ew_bn_apply_sliced(a, ST(stream));
}
int fedfr_bn_bwd_sliced(const uint16_t* dy, const uint16_t* x, const float* mean, const float* rstd, const float* gamma, const float* alpha,
                        const float* sc, const float* sh, int M, int C, float* partials, int rows_in, float* dgamma, float* dbeta,
                        float* dalpha, const uint16_t* add, uint16_t* dx, const uint16_t* nx, const float* nmean, const float* nrstd,
                        float* npart, void* stream) {
  FEDFR_REQUIRE(partials && ew_bn_sliced_ok(M, C, rows_in > 0 ? rows_in : ew_bn_sliced_rows(M, C, true), true),
                "bn_bwd_sliced: shape M=%d C=%d (%d rows) is not served (fedfr_bn_sliced_ok)", M, C, rows_in);
  BnBwdS p{};
  p.dy = BF(dy); p.x = BF(x); p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.alpha = alpha; p.sc = sc; p.sh = sh; p.M = M; p.C = C;
  p.count = (double)M; p.partials = partials;
  if (rows_in <= 0) FEDFR_TRY(ew_bn_bwd_reduce_sliced(p, ST(stream)));
  p.part_in = partials; p.P = rows_in > 0 ? rows_in : ew_bn_sliced_rows(M, C, true);
  p.dgamma = dgamma; p.dbeta = dbeta; p.dalpha = dalpha; p.add = BF(add); p.dx = BFM(dx);
  p.nx = BF(nx); p.nmean = nmean; p.nrstd = nrstd; p.npart = npart;
  return ew_bn_bwd_apply_sliced(p, ST(stream));
}

// ---- head ------------------------------------------------------------------------------------------------------
int fedfr_normalize_rows(const float* x, float* xn, float* inv, int R, int D, float eps, void* stream) {
  return head_normalize_rows(x, xn, inv, R, D, eps, ST(stream));
}
int fedfr_normalize_rows_bwd(const float* xn, const float* inv, const float* dxn, float* dx, int R, int D, float beta, void* stream) {
  return head_normalize_rows_bwd(xn, inv, dxn, dx, R, D, beta, ST(stream));
}
int fedfr_sgemm_splitk(const float* A, const float* B, float* C, int M, int N, int K, long long sam, long long sak, long long sbk, long long sbn,
                       int ldc, float alpha, int splits, long long slab_stride, void* stream) {
  return head_sgemm_splitk(A, B, C, M, N, K, sam, sak, sbk, sbn, ldc, alpha, splits, slab_stride, ST(stream));
}
int fedfr_softmax_ce_fused(float* z, const long long* label, int R, int C, int ldz, float s, float m, int arcface, float inv_batch, float* prob_t,
                           int nslab, long long slab_stride, void* stream) {
  return head_softmax_ce_fused(z, label, R, C, ldz, s, m, arcface, inv_batch, prob_t, nslab, slab_stride, ST(stream));
}
int fedfr_normalize_rows_bwd_slabs(const float* xn, const float* inv_norm, const float* dxn, int nslab, long long slab_stride, float* dx, int R,
                                   int D, float beta, void* stream) {
  return head_normalize_rows_bwd_slabs(xn, inv_norm, dxn, nslab, slab_stride, dx, R, D, beta, ST(stream));
}
int fedfr_sgemm(const float* A, const float* B, float* C, int M, int N, int K, long long sam, long long sak, long long sbk,
                long long sbn, int ldc, float alpha, float beta, const float* bias, void* stream) {
  return head_sgemm(A, B, C, M, N, K, sam, sak, sbk, sbn, ldc, alpha, beta, bias, ST(stream));
}
int fedfr_margin_rowmax(float* z, const long long* label, int R, int C, int ldz, float s, float m, int arc, float* row_max,
                        float* dmul, void* stream) {
  return head_margin_rowmax(z, label, R, C, ldz, s, m, arc, row_max, dmul, ST(stream));
}
int fedfr_exp_rowsum(float* z, int R, int C, int ldz, const float* row_max, float* row_sum, void* stream) {
  return head_exp_rowsum(z, R, C, ldz, row_max, row_sum, ST(stream));
}
int fedfr_softmax_grad(float* z, const long long* label, int R, int C, int ldz, const float* row_sum, const float* dmul, float s,
                       float inv_batch, float* prob_t, void* stream) {
  return head_softmax_grad(z, label, R, C, ldz, row_sum, dmul, s, inv_batch, prob_t, ST(stream));
}
int fedfr_margin_bwd(const float* dlogits, const long long* label, const float* dmul, float s, int R, int C, float* dcos, void* stream) {
  return head_margin_bwd(dlogits, label, dmul, s, R, C, dcos, ST(stream));
}
int fedfr_exp_rowsum_target(float* z, const long long* label, int R, int C, int ldz, const float* row_max, float* sums2, void* stream) {
  return head_exp_rowsum_target(z, label, R, C, ldz, row_max, sums2, ST(stream));
}
int fedfr_nll_mean_ratio(const float* num, const float* den, int R, float floor_, float* loss, void* stream) {
  return head_nll_mean_ratio(num, den, R, floor_, loss, ST(stream));
}
int fedfr_nll_mean(const float* prob_t, int R, float floor_, float* loss, void* stream) {
  return head_nll_mean(prob_t, R, floor_, loss, ST(stream));
}
int fedfr_bce_logits(const float* cosv, const long long* label, const float* bias, int B, int C, float m, float r, float t,
                     float* z, unsigned char* gt, float* dzdcos, void* stream) {
  return head_bce_logits(cosv, label, bias, B, C, m, r, t, z, gt, dzdcos, ST(stream));
}
int fedfr_bce_loss(const float* z, const unsigned char* gt, const float* dzdcos, int B, int C, float r, float lam, float loss_scale,
                   float* dz, float* dcos, float* row_loss, void* stream) {
  return head_bce_loss(z, gt, dzdcos, B, C, r, lam, loss_scale, dz, dcos, row_loss, ST(stream));
}
int fedfr_sgemm_colflag(const float* A, const float* B, int M, int N, int K, long long sam, long long sak, long long sbk,
                        long long sbn, float alpha, float thr, unsigned char* flags, void* stream) {
  return head_sgemm_colflag(A, B, M, N, K, sam, sak, sbk, sbn, alpha, thr, flags, ST(stream));
}
int fedfr_class_accumulate(const float* feats, const long long* label, int B, int D, int C, float* sums, float* counts, void* stream) {
  return head_class_accumulate(feats, label, B, D, C, sums, counts, ST(stream));
}
int fedfr_roc_histogram(const float* feats, const long long* label, int N, int D, int T, unsigned long long* hist, void* stream) {
  return head_roc_histogram(feats, label, N, D, T, hist, ST(stream));
}
int fedfr_bias_prelu_bwd(const uint16_t* dy, const uint16_t* x, const float* bias, const float* alpha, int M, int C, float* partials,
                         float* coef, float* dbias, float* dalpha, const uint16_t* add, uint16_t* dx, void* stream) {
  return ew_bias_prelu_bwd(BF(dy), BF(x), bias, alpha, M, C, partials, coef, dbias, dalpha, BF(add), BFM(dx), ST(stream));
}
int fedfr_pad_input_nhwc(const float* src_nchw, uint16_t* dst_nhwc, int B, int C, int HW, int Cpad, void* stream) {
  return ew_pad_input_nhwc(src_nchw, BFM(dst_nhwc), B, C, HW, Cpad, ST(stream));
}
int fedfr_preprocess_u8(const unsigned char* src_hwc, const unsigned char* flip, float* dst_nchw, int B, int H, int W, void* stream) {
  return ew_preprocess_u8(src_hwc, flip, dst_nchw, B, H, W, ST(stream));
}
int fedfr_contrastive(const float* feats, const float* global_feats, const float* last_feats, int B, int D, float temperature,
                      float* row_loss, float* dfeats, void* stream) {
  return head_contrastive(feats, global_feats, last_feats, B, D, temperature, row_loss, dfeats, ST(stream));
}
int fedfr_colsum_f32(const float* x, int R, int C, float* out, void* stream) { return head_colsum_f32(x, R, C, out, ST(stream)); }
int fedfr_sum_scale(const float* x, int n, float scale, float* out, void* stream) { return head_sum_scale(x, n, scale, out, ST(stream)); }

// ---- optimiser / aggregation / PartialFC ------------------------------------------------------------------------
int fedfr_sgd_step(float* params, const float* grads, float* buf, uint16_t* shadow, size_t n, float lr, float momentum,
                   float weight_decay, int first_step, void* stream) {
  return optim_sgd(params, const_cast<float*>(grads), buf, BFM(shadow), n, lr, momentum, weight_decay, first_step, ST(stream));
}
int fedfr_sgd_step_scaled(float* params, float* grads, float* buf, uint16_t* shadow, size_t n, float lr, float momentum, float weight_decay,
                          int first_step, float grad_scale, unsigned* overflow, void* stream) {
  FEDFR_REQUIRE(grad_scale > 0.f, "sgd_step_scaled: grad_scale must be positive");
  return optim_sgd(params, grads, buf, BFM(shadow), n, lr, momentum, weight_decay, first_step, ST(stream), grad_scale, overflow);
}
int fedfr_fedavg_axpy(float* dst, const float* src, float w, size_t n, int accumulate, void* stream) {
  return optim_fedavg_axpy(dst, src, w, n, accumulate, ST(stream));
}
int fedfr_fedavg_multi(float* dst, const float* const* srcs, const float* ws, int k, size_t n, int accumulate, void* stream) {
  return optim_fedavg_multi(dst, srcs, ws, k, n, accumulate, ST(stream));
}
int fedfr_fedavg_i64(float* acc, const long long* src, float w, int n, int accumulate, long long* out_trunc, void* stream) {
  return optim_fedavg_i64(acc, src, w, n, accumulate, out_trunc, ST(stream));
}
int fedfr_pfc_rand(float* perm, int n, unsigned long long seed, unsigned long long step, void* stream) {
  return optim_pfc_rand(perm, n, seed, step, ST(stream));
}
int fedfr_pfc_localize(long long* label, int n, long long class_start, int num_local, float* perm, void* stream) {
  return optim_pfc_localize(label, n, class_start, num_local, perm, ST(stream));
}
int fedfr_pfc_topk(const float* perm, int n, int k, long long* index, int* npos_out, void* stream) {
  return optim_pfc_topk(perm, n, k, index, npos_out, ST(stream));
}
int fedfr_pfc_positive(const float* perm, int n, long long* index, int* count, void* stream) {
  return optim_pfc_positive(perm, n, index, count, ST(stream));
}
int fedfr_pfc_remap(long long* label, int n, const long long* index, int k, void* stream) {
  return optim_pfc_remap(label, n, index, k, ST(stream));
}
int fedfr_rows_gather(float* dst, const float* src, const long long* index, int k, int D, int table_rows, void* stream) {
  return optim_rows(dst, src, index, k, D, 0, table_rows, ST(stream));
}
int fedfr_rows_scatter(float* dst, const float* src, const long long* index, int k, int D, int table_rows, void* stream) {
  return optim_rows(dst, src, index, k, D, 1, table_rows, ST(stream));
}

}  // extern "C"
