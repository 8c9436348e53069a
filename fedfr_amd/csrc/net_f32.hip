// fp32 VALIDATION path of the backbone: the same network plan (net.h: blocks, parameter / buffer offsets, reference state_dict order) walked
// with fp32 activations and exact-fp32 arithmetic — what north_star's "1e-3 fp32" clause is checked with, and the yardstick that separates
// the bf16 STORAGE noise of the product path from kernel error (reference: backbones/iresnet.py:46-57 block, :158-172 forward, autograd of both).
//
// Not a performance path: convolutions are im2col + the head's GEMM kernel with fp64 accumulation (head.hip, v_mfma_f64_16x16x4_f64 on the
// exactly widened fp32 operands; with a sequential fp32 accumulation over up to 10^5 positions the path was 3x further from the fp64
// evaluation of a step than the fp32 reference is),
// BatchNorm is a two-pass column reduction in fp64 + an elementwise pass, everything is one launch per operation.  ~40 TFLOP/s: an
// iresnet100 step at batch 8 takes ~0.1 s.  NHWC fp32 [M = B*H*W][C] everywhere; weights are the fp32 master copies in KRSC order.
#include <vector>
#include "net.h"
#include "ew.h"
#include "head.h"

namespace {
constexpr float kEps = 1e-5f, kMomentum = 0.1f;

// ---- layout conversions -------------------------------------------------------------------------------------------------
__global__ void f32_nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int HW) {
  const size_t n = (size_t)B * C * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t r = i / C;
    const int hw = (int)(r % HW), b = (int)(r / HW);
    dst[i] = src[((size_t)b * C + c) * HW + hw];
  }
}
__global__ void f32_nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int HW) {
  const size_t n = (size_t)B * C * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int hw = (int)(i % HW);
    const size_t r = i / HW;
    const int c = (int)(r % C), b = (int)(r / C);
    dst[i] = src[((size_t)b * HW + hw) * C + c];
  }
}
// ---- im2col / col2im (R x R taps, zero padding) ------------------------------------------------------------------------------
// cols[m][tap][c] = x[img][ho*s - pad + r][wo*s - pad + q][c]
__global__ void f32_im2col_kernel(const float* __restrict__ x, float* __restrict__ cols, int B, int H, int W, int C, int Ho, int Wo, int R,
                                  int stride, int pad) {
  const size_t n = (size_t)B * Ho * Wo * R * R * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t t = i / C;
    const int tap = (int)(t % (R * R));
    t /= R * R;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho), b = (int)(t / Ho);
    const int h = ho * stride - pad + tap / R, w = wo * stride - pad + tap % R;
    cols[i] = (h >= 0 && h < H && w >= 0 && w < W) ? x[(((size_t)b * H + h) * W + w) * C + c] : 0.f;
  }
}
// transposed-conv gather for the data gradient: cols[m_in][tap][co] = dy[img][(h + pad - r) / s][(w + pad - q) / s][co] where that is an
// output position (else 0): dx = cols . Wd is then ONE long-K GEMM with a single rounding at its end (dy . W followed by a col2im that
// adds nine separately rounded fp32 terms was 3x noisier than the reference's fp32 backward, measured against the fp64 evaluation)
__global__ void f32_im2col_t_kernel(const float* __restrict__ dy, float* __restrict__ cols, int B, int H, int W, int Co, int Ho, int Wo, int R,
                                    int stride, int pad) {
  const size_t n = (size_t)B * H * W * R * R * Co;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % Co);
    size_t t = i / Co;
    const int tap = (int)(t % (R * R));
    t /= R * R;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H), b = (int)(t / H);
    const int hn = h + pad - tap / R, wn = w + pad - tap % R;
    float v = 0.f;
    if (hn >= 0 && wn >= 0 && hn % stride == 0 && wn % stride == 0) {
      const int ho = hn / stride, wo = wn / stride;
      if (ho < Ho && wo < Wo) v = dy[(((size_t)b * Ho + ho) * Wo + wo) * Co + co];
    }
    cols[i] = v;
  }
}
// Wd[tap][co][ci] = W[co][tap][ci]
__global__ void f32_wperm_kernel(const float* __restrict__ w, float* __restrict__ wd, int Co, int T, int Ci) {
  const size_t n = (size_t)Co * T * Ci;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Ci);
    const size_t t = i / Ci;
    const int co = (int)(t % Co), tap = (int)(t / Co);
    wd[i] = w[((size_t)co * T + tap) * Ci + ci];
  }
}
// ---- column statistics: part[slice][v][C] (fp64), block = 32 channels x 8 row groups, grid (C / 32, slices) ---------------------------------------
// v = 0: sum a[m][c], v = 1: sum b[m][c], v = 2: sum d[m][c] of the per-element triple an op produces
struct ColOp {
  int kind;                 // 0: (x, x^2, 0)   1: BatchNorm(+PReLU) backward sums (dz, dz * xhat, dy * z over z <= 0)
  const float *x, *dy, *mean, *rstd, *scale, *beta, *alpha;      // scale = gamma * rstd as the forward pass stored it
};
__device__ __forceinline__ void col_triple(const ColOp& o, size_t idx, int c, double& a, double& b, double& d) {
  const float x = o.x[idx];
  if (o.kind == 0) { a = x; b = (double)x * x; d = 0.0; return; }
  const float xh = (x - o.mean[c]) * o.rstd[c];
  const float dy = o.dy[idx];
  float dz = dy;
  d = 0.0;
  if (o.alpha) {
    // the PReLU input exactly as the forward pass computed it (f32_bn_apply_kernel: same expression, same rounding): an element within
    // an ulp of the kink must land on the same side in both passes — gamma * xhat + beta differs in the last bit and flipped a handful
    // of derivatives (parameter gradients of single layers 1e-3 off the fp64 evaluation, everything else 1e-6)
    const float z = (x - o.mean[c]) * o.scale[c] + (o.beta ? o.beta[c] : 0.f);
    if (z <= 0.f) { d = (double)dy * z; dz = dy * o.alpha[c]; }
  }
  a = dz;
  b = (double)dz * xh;
}
__global__ __launch_bounds__(256) void f32_colsum_kernel(ColOp o, int M, int C, double* __restrict__ part) {
  __shared__ double red[3][8][33];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int S = gridDim.y, sl = blockIdx.y;
  double a = 0.0, b = 0.0, d = 0.0;
  if (c < C)
    for (int m = sl * 8 + rg; m < M; m += S * 8) {
      double ta, tb, td;
      col_triple(o, (size_t)m * C + c, c, ta, tb, td);
      a += ta; b += tb; d += td;
    }
  red[0][rg][cl] = a; red[1][rg][cl] = b; red[2][rg][cl] = d;
  __syncthreads();
  if (rg < 3 && c < C) {
    double t = 0.0;
    for (int i = 0; i < 8; ++i) t += red[rg][i][cl];
    part[((size_t)sl * 3 + rg) * C + c] = t;
  }
}
// BatchNorm forward finalize: mean / rstd / scale / shift (+ running statistics in training)
__global__ void f32_bn_finalize_kernel(const double* __restrict__ part, int S, int C, double count, const float* gamma, const float* beta, float* rm,
                                       float* rv, int training, float* save /* [4][C]: scale, shift, mean, rstd */) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double mean, var;
  if (training) {
    double s = 0.0, q = 0.0;
    for (int i = 0; i < S; ++i) { s += part[((size_t)i * 3 + 0) * C + c]; q += part[((size_t)i * 3 + 1) * C + c]; }
    mean = s / count;
    var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rm[c] = (float)((1.0 - kMomentum) * (double)rm[c] + kMomentum * mean);
    rv[c] = (float)((1.0 - kMomentum) * (double)rv[c] + kMomentum * unb);
  } else {
    mean = rm[c];
    var = rv[c];
  }
  const double rstd = 1.0 / sqrt(var + (double)kEps);
  const double g = gamma ? gamma[c] : 1.0, bt = beta ? beta[c] : 0.0;
  save[c] = (float)(g * rstd);
  save[C + c] = (float)(bt - mean * g * rstd);
  save[2 * C + c] = (float)mean;
  save[3 * C + c] = (float)rstd;
}
// y = prelu?(x * scale + shift) (+ add)
__global__ void f32_bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ save, const float* beta, const float* alpha,
                                    const float* __restrict__ add, float* __restrict__ y, size_t n, int C) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    float v = (x[i] - save[2 * C + c]) * save[c] + (beta ? beta[c] : 0.f);     // centred form: x * scale + shift cancels when |mean| >> std
    if (alpha) v = v > 0.f ? v : alpha[c] * v;
    if (add) v += add[i];
    y[i] = v;
  }
}
// BatchNorm(+PReLU) backward: parameter gradients + coefficients from the column sums, then dx = a dz + A xhat_part ...
__global__ void f32_bn_bwd_finalize_kernel(const double* __restrict__ part, int S, int C, double count, const float* gamma, const float* save,
                                           float* dgamma, float* dbeta, float* dalpha, float* coef /* [2][C]: mean(dz), mean(dz xhat) */) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double t1 = 0.0, t2 = 0.0, t3 = 0.0;
  for (int i = 0; i < S; ++i) {
    t1 += part[((size_t)i * 3 + 0) * C + c];
    t2 += part[((size_t)i * 3 + 1) * C + c];
    t3 += part[((size_t)i * 3 + 2) * C + c];
  }
  if (dgamma) dgamma[c] = (float)t2;
  if (dbeta) dbeta[c] = (float)t1;
  if (dalpha) dalpha[c] = (float)t3;
  coef[c] = (float)(t1 / count);
  coef[C + c] = (float)(t2 / count);
}
__global__ void f32_bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ save, const float* gamma,
                                        const float* beta, const float* alpha, const float* __restrict__ coef, const float* __restrict__ add,
                                        float* __restrict__ dx, size_t n, int C) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const float mean = save[2 * C + c], rstd = save[3 * C + c];
    const float xh = (x[i] - mean) * rstd;
    const float g = gamma ? gamma[c] : 1.f;
    float dz = dy[i];
    if (alpha) {
      const float z = (x[i] - mean) * save[c] + (beta ? beta[c] : 0.f);      // the forward's expression (see col_triple)
      if (z <= 0.f) dz *= alpha[c];
    }
    float v = g * rstd * (dz - coef[c] - xh * coef[C + c]);
    if (add) v += add[i];
    dx[i] = v;
  }
}
__global__ void f32_add_kernel(const float* a, const float* b, float* y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = a[i] + b[i];
}

inline dim3 ew_grid(size_t n) { return dim3((unsigned)std::min<size_t>((n + 255) / 256, 65535)); }

// ---- plan layout (floats) -------------------------------------------------------------------------------------------------------
struct BlkOff { size_t x, a1, c1, a2, c2, d, out, s1, s2, s3, sd; };
struct Layout {
  size_t x0, c0, a0, s_stem, s_bn2, t, tn, yfc, s_feat, total;
  std::vector<BlkOff> blk;
  size_t max_act, max_cols, max_w;
};
Layout make_layout(const FedfrNet* n) {
  Layout L{};
  size_t o = 0;
  auto take = [&](size_t k) { const size_t r = o; o += (k + 63) / 64 * 64; return r; };
  const size_t B = n->B;
  const size_t M0 = B * n->HW * n->HW;
  L.max_act = 0; L.max_cols = 0; L.max_w = 0;
  auto upd = [&](size_t M, size_t C) { L.max_act = std::max(L.max_act, M * C); };
  auto updc = [&](const ConvD& c) {
    L.max_cols = std::max(L.max_cols, B * c.Hout * c.Hout * (size_t)c.R * c.R * c.Cin);        // forward / weight-gradient operand
    L.max_cols = std::max(L.max_cols, B * c.Hin * c.Hin * (size_t)c.R * c.R * c.Cout);         // data-gradient operand
    L.max_w = std::max(L.max_w, (size_t)c.Cout * c.R * c.R * c.Cin);
  };
  L.x0 = take(M0 * 3); L.c0 = take(M0 * 64); L.a0 = take(M0 * 64); L.s_stem = take(4 * 64);
  upd(M0, 64); updc(n->stem);
  size_t prev = L.a0;
  for (const auto& k : n->blocks) {
    const size_t Mi = B * k.Hin * k.Hin, Mo = B * k.Hout * k.Hout;
    BlkOff b{};
    b.x = prev;
    b.a1 = take(Mi * k.Cin); b.c1 = take(Mi * k.Cout); b.a2 = take(Mi * k.Cout); b.c2 = take(Mo * k.Cout);
    b.d = k.has_ds ? take(Mo * k.Cout) : 0;
    b.out = take(Mo * k.Cout);
    b.s1 = take(4 * k.Cin); b.s2 = take(4 * k.Cout); b.s3 = take(4 * k.Cout); b.sd = k.has_ds ? take(4 * k.Cout) : 0;
    prev = b.out;
    L.blk.push_back(b);
    upd(Mi, k.Cin); upd(Mi, k.Cout); upd(Mo, k.Cout);
    updc(k.conv1); updc(k.conv2);
    if (k.has_ds) updc(k.ds);
  }
  const size_t hw = (size_t)n->final_hw * n->final_hw;
  L.s_bn2 = take(4 * 512); L.tn = take(B * hw * 512); L.t = take(B * n->fc_in); L.yfc = take(B * n->F); L.s_feat = take(2 * n->F);
  upd(B * hw, 512);
  L.max_act = std::max(L.max_act, B * (size_t)n->fc_in);
  L.total = o;
  return L;
}
struct Ws { float *cols, *g0, *g1, *t0, *t1, *t2, *t3, *coef, *wd; double* part; };
constexpr int kSlices = 64;
size_t ws_floats(const Layout& L) { return L.max_cols + 6 * L.max_act + L.max_w + 2 * 2048 + 64 * 10 + (size_t)kSlices * 3 * 2048 * 2; }
Ws carve(float* w, const Layout& L) {
  Ws s;
  size_t o = 0;
  auto take = [&](size_t k) { float* r = w + o; o += (k + 63) / 64 * 64; return r; };
  s.cols = take(L.max_cols);
  s.g0 = take(L.max_act); s.g1 = take(L.max_act); s.t0 = take(L.max_act); s.t1 = take(L.max_act); s.t2 = take(L.max_act); s.t3 = take(L.max_act);
  s.coef = take(2 * 2048);
  s.wd = take(L.max_w);
  s.part = reinterpret_cast<double*>(take((size_t)kSlices * 3 * 2048 * 2));
  return s;
}

struct Cx {
  const FedfrNet* n; const float* params; float* bufs; float* A; Ws w; float* grads; hipStream_t st;
};
int launch_ok(const char* what) { return fedfr_check_launch(what); }

int bn_forward(const Cx& c, const BnD& b, const float* x, size_t M, float* save, const float* alpha, const float* add, float* y, bool training) {
  FEDFR_REQUIRE(b.C <= 2048, "fp32 path: BatchNorm wider than 2048 channels");
  if (training) {
    ColOp o{}; o.kind = 0; o.x = x;
    hipLaunchKernelGGL(f32_colsum_kernel, dim3(ceil_div(b.C, 32), kSlices), dim3(256), 0, c.st, o, (int)M, b.C, c.w.part);
    FEDFR_TRY(launch_ok("f32_colsum"));
  }
  hipLaunchKernelGGL(f32_bn_finalize_kernel, dim3(ceil_div(b.C, 256)), dim3(256), 0, c.st, c.w.part, kSlices, b.C, (double)M,
                     b.g_off >= 0 ? c.params + b.g_off : nullptr, c.params + b.b_off, c.bufs + b.rm_off, c.bufs + b.rv_off, training ? 1 : 0, save);
  FEDFR_TRY(launch_ok("f32_bn_finalize"));
  const size_t n = M * b.C;
  hipLaunchKernelGGL(f32_bn_apply_kernel, ew_grid(n), dim3(256), 0, c.st, x, save, c.params + b.b_off, alpha, add, y, n, b.C);
  return launch_ok("f32_bn_apply");
}
// conv forward: y[Mo][Cout] = im2col(x) . W^T
int conv_forward(const Cx& c, const ConvD& cv, const float* x, float* y, int Cin_override = 0) {
  const int B = c.n->B, Cin = Cin_override ? Cin_override : cv.Cin;
  const int pad = cv.R == 3 ? 1 : 0, K = cv.R * cv.R * Cin;
  const size_t Mo = (size_t)B * cv.Hout * cv.Hout;
  hipLaunchKernelGGL(f32_im2col_kernel, ew_grid(Mo * K), dim3(256), 0, c.st, x, c.w.cols, B, cv.Hin, cv.Hin, Cin, cv.Hout, cv.Hout, cv.R, cv.stride, pad);
  FEDFR_TRY(launch_ok("f32_im2col"));
  return head_sgemm_f64acc(c.w.cols, c.params + cv.w_off, y, (int)Mo, cv.Cout, K, K, 1, 1, K, cv.Cout, 1.f, 0.f, nullptr, c.st);
}
// conv backward: dW = dy^T . im2col(x) (assigned), and (dx != null) dx = col2im(dy . W)
int conv_backward(const Cx& c, const ConvD& cv, const float* x, const float* dy, float* dx, int Cin_override = 0) {
  const int B = c.n->B, Cin = Cin_override ? Cin_override : cv.Cin;
  const int pad = cv.R == 3 ? 1 : 0, K = cv.R * cv.R * Cin;
  const size_t Mo = (size_t)B * cv.Hout * cv.Hout;
  hipLaunchKernelGGL(f32_im2col_kernel, ew_grid(Mo * K), dim3(256), 0, c.st, x, c.w.cols, B, cv.Hin, cv.Hin, Cin, cv.Hout, cv.Hout, cv.R, cv.stride, pad);
  FEDFR_TRY(launch_ok("f32_im2col"));
  // dW[co][kk] = sum_m dy[m][co] cols[m][kk]
  FEDFR_TRY(head_sgemm_f64acc(dy, c.w.cols, c.grads + cv.w_off, cv.Cout, K, (int)Mo, 1, cv.Cout, K, 1, K, 1.f, 0.f, nullptr, c.st));
  if (!dx) return FEDFR_OK;
  // dx[m_in][ci] = sum over (tap, co) of gathered dy . Wd
  const int T = cv.R * cv.R, Kd = T * cv.Cout;
  const size_t Mi = (size_t)B * cv.Hin * cv.Hin;
  hipLaunchKernelGGL(f32_wperm_kernel, ew_grid((size_t)Kd * Cin), dim3(256), 0, c.st, c.params + cv.w_off, c.w.wd, cv.Cout, T, Cin);
  FEDFR_TRY(launch_ok("f32_wperm"));
  hipLaunchKernelGGL(f32_im2col_t_kernel, ew_grid(Mi * Kd), dim3(256), 0, c.st, dy, c.w.cols, B, cv.Hin, cv.Hin, cv.Cout, cv.Hout, cv.Hout, cv.R, cv.stride, pad);
  FEDFR_TRY(launch_ok("f32_im2col_t"));
  return head_sgemm_f64acc(c.w.cols, c.w.wd, dx, (int)Mi, Cin, Kd, Kd, 1, Cin, 1, Cin, 1.f, 0.f, nullptr, c.st);
}
int bn_backward(const Cx& c, const BnD& b, const float* alpha, long long alpha_off, const float* dy, const float* x, size_t M, const float* save,
                const float* add, float* dx) {
  ColOp o{}; o.kind = 1; o.x = x; o.dy = dy; o.mean = save + 2 * b.C; o.rstd = save + 3 * b.C;
  const float* gamma = b.g_off >= 0 ? c.params + b.g_off : nullptr;
  o.scale = save; o.beta = c.params + b.b_off; o.alpha = alpha;
  hipLaunchKernelGGL(f32_colsum_kernel, dim3(ceil_div(b.C, 32), kSlices), dim3(256), 0, c.st, o, (int)M, b.C, c.w.part);
  FEDFR_TRY(launch_ok("f32_colsum"));
  hipLaunchKernelGGL(f32_bn_bwd_finalize_kernel, dim3(ceil_div(b.C, 256)), dim3(256), 0, c.st, c.w.part, kSlices, b.C, (double)M, gamma, save,
                     b.g_off >= 0 ? c.grads + b.g_off : nullptr, c.grads + b.b_off, alpha ? c.grads + alpha_off : nullptr, c.w.coef);
  FEDFR_TRY(launch_ok("f32_bn_bwd_finalize"));
  const size_t n = M * b.C;
  hipLaunchKernelGGL(f32_bn_bwd_apply_kernel, ew_grid(n), dim3(256), 0, c.st, dy, x, save, gamma, o.beta, alpha, c.w.coef, add, dx, n, b.C);
  return launch_ok("f32_bn_bwd_apply");
}
}  // namespace

size_t net_f32_arena_floats(const FedfrNet* n) { return make_layout(n).total; }
size_t net_f32_ws_floats(const FedfrNet* n) { return ws_floats(make_layout(n)); }

int net_f32_forward(const FedfrNet* n, const float* x, const float* params, float* bufs, float* arena, float* ws, float* feats, int training,
                    hipStream_t st) {
  FEDFR_REQUIRE(n && x && params && bufs && arena && ws && feats, "net_f32_forward: null buffer");
  FEDFR_REQUIRE(!n->block_only, "net_f32_forward: whole-network plans only");
  FEDFR_REQUIRE(training == 0 || training == 1, "net_f32_forward: training must be 0 or 1");
  FEDFR_REQUIRE(n->dropout_p == 0.f || !training, "net_f32_forward: dropout is not part of the fp32 validation path");
  const Layout L = make_layout(n);
  Cx c{n, params, bufs, arena, carve(ws, L), nullptr, st};
  float* A = arena;
  const bool tr = training != 0;
  const int B = n->B, HW = n->HW;
  const size_t M0 = (size_t)B * HW * HW;
  hipLaunchKernelGGL(f32_nchw_to_nhwc_kernel, ew_grid(M0 * 3), dim3(256), 0, st, x, A + L.x0, B, 3, HW * HW);
  FEDFR_TRY(launch_ok("f32_nchw_to_nhwc"));
  FEDFR_TRY(conv_forward(c, n->stem, A + L.x0, A + L.c0));                            // iresnet.py:160-162
  FEDFR_TRY(bn_forward(c, n->stem_bn, A + L.c0, M0, A + L.s_stem, params + n->stem_alpha_off, nullptr, A + L.a0, tr));
  for (size_t bi = 0; bi < n->blocks.size(); ++bi) {                                  // iresnet.py:46-57
    const BlockD& k = n->blocks[bi];
    const BlkOff& o = L.blk[bi];
    const size_t Mi = (size_t)B * k.Hin * k.Hin, Mo = (size_t)B * k.Hout * k.Hout;
    FEDFR_TRY(bn_forward(c, k.bn1, A + o.x, Mi, A + o.s1, nullptr, nullptr, A + o.a1, tr));
    FEDFR_TRY(conv_forward(c, k.conv1, A + o.a1, A + o.c1));
    FEDFR_TRY(bn_forward(c, k.bn2, A + o.c1, Mi, A + o.s2, params + k.alpha_off, nullptr, A + o.a2, tr));
    FEDFR_TRY(conv_forward(c, k.conv2, A + o.a2, A + o.c2));
    const float* idn = A + o.x;
    if (k.has_ds) {
      FEDFR_TRY(conv_forward(c, k.ds, A + o.x, c.w.t0));
      FEDFR_TRY(bn_forward(c, k.bnds, c.w.t0, Mo, A + o.sd, nullptr, nullptr, A + o.d, tr));      // (the raw 1x1 output is recomputed by the backward pass)
      idn = A + o.d;
    }
    FEDFR_TRY(bn_forward(c, k.bn3, A + o.c2, Mo, A + o.s3, nullptr, idn, A + o.out, tr));
  }
  const int hw = n->final_hw * n->final_hw;
  const size_t Mf = (size_t)B * hw;
  FEDFR_TRY(bn_forward(c, n->bn2, A + L.blk.back().out, Mf, A + L.s_bn2, nullptr, nullptr, A + L.tn, tr));     // iresnet.py:167
  hipLaunchKernelGGL(f32_nhwc_to_nchw_kernel, ew_grid(Mf * 512), dim3(256), 0, st, A + L.tn, A + L.t, B, 512, hw);    // torch.flatten of NCHW
  FEDFR_TRY(launch_ok("f32_nhwc_to_nchw"));
  FEDFR_TRY(head_sgemm_f64acc(A + L.t, params + n->fc_w_off, A + L.yfc, B, n->F, n->fc_in, n->fc_in, 1, 1, n->fc_in, n->F, 1.f, 0.f, params + n->fc_b_off, st));
  return ew_bn1d_fwd(A + L.yfc, feats, B, n->F, nullptr, params + n->feat_bn.b_off, bufs + n->feat_bn.rm_off, bufs + n->feat_bn.rv_off, kMomentum, kEps,
                     tr ? 1 : 0, A + L.s_feat, A + L.s_feat + n->F, st);
}

int net_f32_backward(const FedfrNet* n, const float* dfeats, const float* params, float* arena, float* ws, float* grads, hipStream_t st) {
  FEDFR_REQUIRE(n && dfeats && params && arena && ws && grads, "net_f32_backward: null buffer");
  FEDFR_REQUIRE(!n->block_only, "net_f32_backward: whole-network plans only");
  const Layout L = make_layout(n);
  Cx c{n, params, nullptr, arena, carve(ws, L), grads, st};
  float* A = arena;
  const int B = n->B, F = n->F;
  float *g = c.w.g0, *gin = c.w.g1;
  {
    // features (BatchNorm1d, weight frozen at 1) -> fc -> flatten -> bn2
    float* dyfc = c.w.t0;
    FEDFR_TRY(ew_bn1d_bwd(dfeats, A + L.yfc, dyfc, B, F, nullptr, A + L.s_feat, A + L.s_feat + F, grads + n->feat_bn.b_off, grads + n->fc_b_off, nullptr,
                          nullptr, 0, st));
    // fc.weight grad [F][fc_in] = dY^T X;  dX [B][fc_in] = dY W
    FEDFR_TRY(head_sgemm_f64acc(dyfc, A + L.t, grads + n->fc_w_off, F, n->fc_in, B, 1, F, n->fc_in, 1, n->fc_in, 1.f, 0.f, nullptr, st));
    FEDFR_TRY(head_sgemm_f64acc(dyfc, params + n->fc_w_off, c.w.t1, B, n->fc_in, F, F, 1, n->fc_in, 1, n->fc_in, 1.f, 0.f, nullptr, st));
    const int hw = n->final_hw * n->final_hw;
    const size_t Mf = (size_t)B * hw;
    hipLaunchKernelGGL(f32_nchw_to_nhwc_kernel, ew_grid(Mf * 512), dim3(256), 0, st, c.w.t1, c.w.t2, B, 512, hw);
    FEDFR_TRY(launch_ok("f32_nchw_to_nhwc"));
    FEDFR_TRY(bn_backward(c, n->bn2, nullptr, 0, c.w.t2, A + L.blk.back().out, Mf, A + L.s_bn2, nullptr, g));
  }
  for (int bi = (int)n->blocks.size() - 1; bi >= 0; --bi) {
    const BlockD& k = n->blocks[bi];
    const BlkOff& o = L.blk[bi];
    const size_t Mi = (size_t)B * k.Hin * k.Hin, Mo = (size_t)B * k.Hout * k.Hout;
    float *dc2 = c.w.t0, *da2 = c.w.t1, *dc1 = c.w.t2, *da1 = c.w.t3;
    FEDFR_TRY(bn_backward(c, k.bn3, nullptr, 0, g, A + o.c2, Mo, A + o.s3, nullptr, dc2));
    FEDFR_TRY(conv_backward(c, k.conv2, A + o.a2, dc2, da2));
    FEDFR_TRY(bn_backward(c, k.bn2, params + k.alpha_off, k.alpha_off, da2, A + o.c1, Mi, A + o.s2, nullptr, dc1));
    FEDFR_TRY(conv_backward(c, k.conv1, A + o.a1, dc1, da1));
    if (k.has_ds) {
      // identity path: d = bnds(conv1x1(x)); the raw 1x1 output is recomputed (one GEMM) rather than kept
      float *raw = dc2, *dd = da2, *dxd = dc1;
      FEDFR_TRY(conv_forward(c, k.ds, A + o.x, raw));
      FEDFR_TRY(bn_backward(c, k.bnds, nullptr, 0, g, raw, Mo, A + o.sd, nullptr, dd));
      FEDFR_TRY(conv_backward(c, k.ds, A + o.x, dd, dxd));
      FEDFR_TRY(bn_backward(c, k.bn1, nullptr, 0, da1, A + o.x, Mi, A + o.s1, dxd, gin));
    } else {
      FEDFR_TRY(bn_backward(c, k.bn1, nullptr, 0, da1, A + o.x, Mi, A + o.s1, g, gin));
    }
    std::swap(g, gin);
  }
  // stem: a0 = prelu(bn1(conv1(x)))
  const size_t M0 = (size_t)B * n->HW * n->HW;
  FEDFR_TRY(bn_backward(c, n->stem_bn, params + n->stem_alpha_off, n->stem_alpha_off, g, A + L.c0, M0, A + L.s_stem, nullptr, c.w.t0));
  return conv_backward(c, n->stem, A + L.x0, c.w.t0, nullptr);
}
