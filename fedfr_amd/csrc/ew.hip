// HBM-bound kernels (see ew.h).  Mapping used throughout for [M][C] bf16 tensors: one thread owns 8
// consecutive channels (16-byte vector load/store), C/8 threads cover a row, 256/(C/8) rows per pass,
// blocks stride over row slabs; per-channel partials are reduced through LDS and written as one row per
// block (deterministic, no atomics), then finalised in fp64 by a tiny second kernel.
#include "ew.h"
#include "gemm_tn_dev.h"
#include "gemm_dev.h"   // ProfScope: HIP-event timing of a launch on its stream (bench.py roofline leg); slots 20.. carry algorithmic bytes

#define EW_THREADS 256
// streaming 16-B load of data this pass is the last reader of for a long while (EW_NT=1: non-temporal, keeps L2 / MALL for the data
// the next kernel needs)
#ifndef EW_NT
#define EW_NT 1     // same-box A/B: 21.05 -> 20.91 ms/step
#endif
__device__ __forceinline__ uint4 ew_ld16(const bf16_t* p) {
#if EW_NT
  typedef __attribute__((ext_vector_type(4))) unsigned u4v;
  const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
#else
  return *reinterpret_cast<const uint4*>(p);
#endif
}
// EW_XCD=1: block -> row-slab mapping through xcd_remap, i.e. XCD x streams rows [x M/8, (x+1) M/8) = the images whose conv tiles
// ran on XCD x (the conv kernels use the same remap), so a tensor is read from the L2 its producer left it in and the next conv finds
// its input image in its own L2.  Partial rows stay indexed by the logical slab id: results are bit-identical.
#ifndef EW_XCD
#define EW_XCD 1
#endif
__device__ __forceinline__ int ew_block_id() {
#if EW_XCD
  return xcd_remap((int)blockIdx.x, (int)gridDim.x);
#else
  return (int)blockIdx.x;
#endif
}
#ifndef EW_UNROLL_ALPHA
#define EW_UNROLL_ALPHA 2   // bn_bwd_reduce<PReLU>
#endif
#ifndef EW_UNROLL_HEAVY
#define EW_UNROLL_HEAVY 2   // bn_bwd_apply with PReLU or with the next BN's reduction
#endif
#ifndef EW_UNROLL
#define EW_UNROLL 4       // rows whose loads are issued together in the streaming BN kernels
#endif

static inline int rows_per_pass(int C) { return EW_THREADS / (C >> 3); }
static int slab_rows(int M, int C, int max_blocks) {
  const int rpp = rows_per_pass(C);
  long long rows = (long long)rpp * 8;
  const long long need = (M + max_blocks - 1) / max_blocks;
  if (rows < need) rows = need;
  rows = (rows + rpp - 1) / rpp * rpp;
  return (int)rows;
}
static int check_mc(int M, int C, const char* who) {
  if (M <= 0 || C <= 0 || (C & 7) || (C >> 3) > EW_THREADS) {
    fedfr_set_error("%s: unsupported shape M=%d C=%d (need C%%8==0, C<=2048)", who, M, C);
    return FEDFR_ERR_ARG;
  }
  return FEDFR_OK;
}

// =====================================================================================================
// forward BN finalize
// =====================================================================================================
// stage A: [P][W] -> [S][W] partial sums (W = ncols), block = 32 cols x 32 row-groups
__global__ __launch_bounds__(1024) void colsum_stage_kernel(const float* __restrict__ in, int P, int W,
                                                           float* __restrict__ out, int S) {
  __shared__ double red[32][33];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + cl;
  const int slice = blockIdx.y;
  double s = 0.0;
  if (col < W)
    for (int row = slice + S * rg; row < P; row += S * 32) s += (double)in[(size_t)row * W + col];
  red[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && col < W) {
    double t = 0.0;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) t += red[i][cl];
    out[(size_t)slice * W + col] = (float)t;
  }
}

// block = 32 channels x 32 row-groups.  Loads are 4-way unrolled (independent accumulators) and the cross-row-group
// reduction is two-level (8 partial sums x 4) so the serial dependent chain stays short: these kernels are pure latency.
__device__ __forceinline__ double rg_reduce(double v, double (*red)[33], int rg, int cl) {
  red[rg][cl] = v;
  __syncthreads();
  if (rg < 4) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += red[rg + 4 * i][cl];
    red[rg][cl] = t;
  }
  __syncthreads();
  return (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ part, int P, int C, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* running_mean, float* running_var, float momentum, float eps,
                                                          float* scale, float* shift, float* save_mean, float* save_rstd) {
  __shared__ double rs[32][33], rq[32][33];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = min(blockIdx.x * 32 + cl, C - 1);
  double s0 = 0.0, s1 = 0.0, q0 = 0.0, q1 = 0.0;
  int row = rg;
  for (; row + 32 < P; row += 64) {
    const float a0 = part[(size_t)row * 2 * C + c], b0 = part[(size_t)row * 2 * C + C + c];
    const float a1 = part[(size_t)(row + 32) * 2 * C + c], b1 = part[(size_t)(row + 32) * 2 * C + C + c];
    s0 += (double)a0; q0 += (double)b0; s1 += (double)a1; q1 += (double)b1;
  }
  if (row < P) {
    s0 += (double)part[(size_t)row * 2 * C + c];
    q0 += (double)part[(size_t)row * 2 * C + C + c];
  }
  const double ts = rg_reduce(s0 + s1, rs, rg, cl);
  const double tq = rg_reduce(q0 + q1, rq, rg, cl);
  if (rg == 0 && blockIdx.x * 32 + cl < C) {
    const double ic = 1.0 / count;
    const double mean = ts * ic;
    double var = tq * ic - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = bn_rsqrt(var + (double)eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = (float)((double)g * rstd);
    scale[c] = sc;
    shift[c] = (float)((double)b - mean * (double)g * rstd);
    save_mean[c] = (float)mean;
    save_rstd[c] = (float)rstd;
    if (running_mean) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
      running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
    }
  }
}

// ---- finalize, 8 channels per block ----------------------------------------------------------------
// The 32-channel kernels above pull P x 2C (or 3C) floats through C/32 = 2..16 CUs: 150-300 KB per CU at one CU's ~0.1 TB/s, i.e. they
// are bound by per-CU bandwidth, not by latency.  Here a block owns 8 channels (C/8 = 8..64 blocks), thread (row-group, half) loads one
// float4 per statistic and row with every load of a trip issued before the first add, sums in fp64, and the 128 row-groups meet in LDS.
#ifndef EW_FIN8
#define EW_FIN8 1
#endif
#define FIN_RG 128
template <int NS>
__device__ __forceinline__ void fin8_accumulate(const float* __restrict__ part, int P, int C, int c0, int ns_live, double (*tot)[8]) {
  __shared__ double red[NS * 8][FIN_RG + 1];
  const int half = threadIdx.x & 1, rg = threadIdx.x >> 1;
  const int W = NS * C;
  double acc[NS][4];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[s][j] = 0.0;
  const float* base = part + c0 + half * 4;
  for (int r = rg; r < P; r += 2 * FIN_RG) {
    const int r1 = r + FIN_RG;
    const bool ok1 = r1 < P;
    float4 v0[NS], v1[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const bool live = s < ns_live;
      v0[s] = live ? *reinterpret_cast<const float4*>(base + (size_t)r * W + s * C) : make_float4(0.f, 0.f, 0.f, 0.f);
      v1[s] = (live && ok1) ? *reinterpret_cast<const float4*>(base + (size_t)r1 * W + s * C) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      acc[s][0] += (double)v0[s].x + (double)v1[s].x; acc[s][1] += (double)v0[s].y + (double)v1[s].y;
      acc[s][2] += (double)v0[s].z + (double)v1[s].z; acc[s][3] += (double)v0[s].w + (double)v1[s].w;
    }
  }
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[s * 8 + half * 4 + j][rg] = acc[s][j];
  __syncthreads();
  // value v = (stat, channel) summed by 8 consecutive lanes: 16 entries each, then 3 xor steps inside the lane group
  const int v = threadIdx.x >> 3, k = threadIdx.x & 7;
  double t = 0.0;
  if (v < NS * 8) {
#pragma unroll
    for (int i = 0; i < FIN_RG / 8; ++i) t += red[v][i * 8 + k];
  }
  t += __shfl_xor(t, 1, 64);
  t += __shfl_xor(t, 2, 64);
  t += __shfl_xor(t, 4, 64);
  if (v < NS * 8 && k == 0) tot[v >> 3][v & 7] = t;
  __syncthreads();
}

__global__ __launch_bounds__(256) void bn_finalize8_kernel(const float* __restrict__ part, int P, int C, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* running_mean, float* running_var, float momentum, float eps,
                                                          float* scale, float* shift, float* save_mean, float* save_rstd) {
  __shared__ double tot[2][8];
  const int c0 = blockIdx.x * 8;
  fin8_accumulate<2>(part, P, C, c0, 2, tot);
  if (threadIdx.x < 8) {
    const int c = c0 + threadIdx.x;
    const double ic = 1.0 / count;
    const double mean = tot[0][threadIdx.x] * ic;
    double var = tot[1][threadIdx.x] * ic - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = bn_rsqrt(var + (double)eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    scale[c] = (float)((double)g * rstd);
    shift[c] = (float)((double)b - mean * (double)g * rstd);
    save_mean[c] = (float)mean;
    save_rstd[c] = (float)rstd;
    if (running_mean) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
      running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
    }
  }
}

int ew_bn_finalize(const float* partials, int P, int C, double count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                   float* save_mean, float* save_rstd, float* tmp, hipStream_t st) {
  FEDFR_REQUIRE(partials && P > 0 && C > 0 && scale && shift && save_mean && save_rstd, "bn_finalize: bad args");
  ProfScope prof(23, (double)P * 2 * C * 4, st);
  const float* src = partials;
  if (P > 1024) {
    FEDFR_REQUIRE(tmp != nullptr, "bn_finalize: P=%d needs a tmp buffer", P);
    const int S = 64, W = 2 * C;
    hipLaunchKernelGGL(colsum_stage_kernel, dim3(ceil_div(W, 32), S), dim3(1024), 0, st, partials, P, W, tmp, S);
    FEDFR_LAUNCH_CHECK("colsum_stage");
    src = tmp;
    P = S;
  }
  if (EW_FIN8 && (C & 7) == 0)
    hipLaunchKernelGGL(bn_finalize8_kernel, dim3(C / 8), dim3(256), 0, st, src, P, C, count, gamma, beta,
                       running_mean, running_var, momentum, eps, scale, shift, save_mean, save_rstd);
  else
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 32)), dim3(1024), 0, st, src, P, C, count, gamma, beta,
                       running_mean, running_var, momentum, eps, scale, shift, save_mean, save_rstd);
  FEDFR_LAUNCH_CHECK("bn_finalize");
  return FEDFR_OK;
}

__global__ void bn_eval_coeffs_kernel(int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float rstd = 1.f / sqrtf(rv[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale[c] = g * rstd;
  shift[c] = b - rm[c] * g * rstd;
}
__global__ __launch_bounds__(256) void bn_eval_coeffs_multi_kernel(const float* __restrict__ params, const float* __restrict__ bufs,
                                                                  float* __restrict__ save, BnEvalTable t) {
  const BnEvalEntry e = t.e[blockIdx.x];
  for (int c = blockIdx.y * 256 + threadIdx.x; c < e.C; c += gridDim.y * 256) {
    const float rstd = 1.f / sqrtf(bufs[e.rv_off + c] + t.eps);
    const float g = e.g_off >= 0 ? params[e.g_off + c] : 1.f, b = params[e.b_off + c];
    save[e.save_off + c] = g * rstd;                                   // scale
    save[e.save_off + e.C + c] = b - bufs[e.rm_off + c] * g * rstd;    // shift
    save[e.save_off + 2 * e.C + c] = bufs[e.rm_off + c];               // mean / rstd as the backward pass of a training net with frozen
    save[e.save_off + 3 * e.C + c] = rstd;                             // BatchNorms reads them (freeze_BN, iresnet.py:140-147)
  }
}
int ew_bn_eval_coeffs_multi(const float* params, const float* bufs, float* save, const BnEvalTable& t, hipStream_t st) {
  FEDFR_REQUIRE(params && bufs && save && t.n > 0 && t.n <= kMaxBnEvalEntries, "bn_eval_coeffs_multi: bad args");
  hipLaunchKernelGGL(bn_eval_coeffs_multi_kernel, dim3(t.n, 2), dim3(256), 0, st, params, bufs, save, t);
  FEDFR_LAUNCH_CHECK("bn_eval_coeffs_multi");
  return FEDFR_OK;
}
int ew_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                      float* scale, float* shift, hipStream_t st) {
  FEDFR_REQUIRE(C > 0 && rm && rv && scale && shift, "bn_eval_coeffs: bad args");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, C, gamma, beta, rm, rv, eps, scale, shift);
  FEDFR_LAUNCH_CHECK("bn_eval_coeffs");
  return FEDFR_OK;
}

// =====================================================================================================
// forward BN apply (+PReLU) (+second normalised/identity input) (+stats of the result)
// =====================================================================================================
__device__ __forceinline__ void load8f(const float* p, int c0, float* v, float dflt) {
  if (p) {
    const float4 a = *reinterpret_cast<const float4*>(p + c0), b = *reinterpret_cast<const float4*>(p + c0 + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = dflt;
  }
}

// X2 / STATS as template parameters: as runtime tests inside the row body they were (uniform) branches between the loads of a batch and
// in front of every store, and each branch ends the region the loads of EW_UNROLL rows are scheduled in
template <bool X2, bool STATS>
__global__ __launch_bounds__(EW_THREADS) void bn_apply_kernel(BnApply p, int slab) {
  extern __shared__ float red[];   // [rpp][2C] when stats
  const int tpr = p.C >> 3, rpp = EW_THREADS / tpr;
  const int cl = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  const bool active = rl < rpp;
  const int c0 = cl * 8;
  float sc1[8], sh1[8], al[8], sc2[8], sh2[8];
  load8f(p.sc1, c0, sc1, 1.f);
  load8f(p.sh1, c0, sh1, 0.f);
  load8f(p.alpha, c0, al, 1.f);
  load8f(p.sc2, c0, sc2, 1.f);
  load8f(p.sh2, c0, sh2, 0.f);
  const bool has_alpha = p.alpha != nullptr;
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
  const int bid = ew_block_id();
  const int mbeg = bid * slab;
  const int mend = min(p.M, mbeg + slab);
  auto one = [&](int m, const uint4& v1, const uint4& v2) {
    const size_t off = (size_t)m * p.C + c0;
    float f[8];
    unpack8(v1, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = f[j] * sc1[j] + sh1[j];
      if (has_alpha) v = v > 0.f ? v : al[j] * v;
      f[j] = v;
    }
    if (X2) {
      float g[8];
      unpack8(v2, g);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += g[j] * sc2[j] + sh2[j];
    }
    const uint4 o = pack8(f);
    if (STATS) {
      float r[8];
      unpack8(o, r);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s[j] += r[j];
        q[j] += r[j] * r[j];
      }
    }
    if (p.nchw_hw > 0) {
      const int img = m / p.nchw_hw, hw = m - img * p.nchw_hw;
      const bf16_t* ob = reinterpret_cast<const bf16_t*>(&o);
#pragma unroll
      for (int j = 0; j < 8; ++j) p.y[((size_t)img * p.C + c0 + j) * p.nchw_hw + hw] = ob[j];
    } else {
      *reinterpret_cast<uint4*>(p.y + off) = o;
    }
  };
  if (active) {
    int m = mbeg + rl;
    for (; m + (EW_UNROLL - 1) * rpp < mend; m += EW_UNROLL * rpp) {    // loads of EW_UNROLL rows first (see bn_bwd_reduce)
      uint4 v1[EW_UNROLL], v2[EW_UNROLL];
#pragma unroll
      for (int u = 0; u < EW_UNROLL; ++u) {
        const size_t off = (size_t)(m + u * rpp) * p.C + c0;
        v1[u] = ew_ld16(p.x1 + off);
        v2[u] = X2 ? *reinterpret_cast<const uint4*>(p.x2 + off) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < EW_UNROLL; ++u) one(m + u * rpp, v1[u], v2[u]);
    }
    for (; m < mend; m += rpp) {
      const size_t off = (size_t)m * p.C + c0;
      one(m, *reinterpret_cast<const uint4*>(p.x1 + off), X2 ? *reinterpret_cast<const uint4*>(p.x2 + off) : make_uint4(0, 0, 0, 0));
    }
  }
  if (STATS) {
    const int W = 2 * p.C;
    if (active) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[rl * W + c0 + j] = s[j];
        red[rl * W + p.C + c0 + j] = q[j];
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < W; i += EW_THREADS) {
      float t = 0.f;
      for (int r = 0; r < rpp; ++r) t += red[r * W + i];
      p.stats[(size_t)bid * W + i] = t;
    }
  }
}

int ew_bn_apply_grid(int M, int C) { return ceil_div(M, slab_rows(M, C, 1024)); }

int ew_bn_apply(const BnApply& p, hipStream_t st) {
  FEDFR_TRY(check_mc(p.M, p.C, "bn_apply"));
  FEDFR_REQUIRE(p.x1 && p.y, "bn_apply: null tensor");
  const int slab = slab_rows(p.M, p.C, 1024);
  const int grid = ceil_div(p.M, slab);
  const size_t lds = p.stats ? (size_t)rows_per_pass(p.C) * 2 * p.C * sizeof(float) : 0;
  ProfScope prof(20, (double)p.M * p.C * 2 * (p.x2 ? 3 : 2), st);
  if (p.x2) {
    if (p.stats) hipLaunchKernelGGL((bn_apply_kernel<true, true>), dim3(grid), dim3(EW_THREADS), lds, st, p, slab);
    else hipLaunchKernelGGL((bn_apply_kernel<true, false>), dim3(grid), dim3(EW_THREADS), lds, st, p, slab);
  } else {
    if (p.stats) hipLaunchKernelGGL((bn_apply_kernel<false, true>), dim3(grid), dim3(EW_THREADS), lds, st, p, slab);
    else hipLaunchKernelGGL((bn_apply_kernel<false, false>), dim3(grid), dim3(EW_THREADS), lds, st, p, slab);
  }
  FEDFR_LAUNCH_CHECK("bn_apply");
  return FEDFR_OK;
}

// =====================================================================================================
// backward BN (+PReLU)
// =====================================================================================================
// These kernels run on the main stream while the weight-gradient GEMMs occupy every CU from the aux stream (2 waves per SIMD x 104
// VGPRs, 128 KB LDS): what is left per CU is ~300 VGPRs per lane and 32 KB of LDS, so they are written to be small — one template
// instance per (PReLU, next-BN reduction) case so that unused per-channel vectors cost no registers, the normalisation folded into
// two per-channel coefficients, and the cross-row reduction done in registers (wave shuffles) before 4 rows per statistic meet in LDS.

// sums v[NV][8] (8 channels x NV statistics per thread) over all rows of the workgroup and writes dst[s * C + c].
// shfl (host-decided: C/8 is a power of two <= 64): rows of a wave meet by xor-shuffles, LDS holds [4 waves][NV C]; otherwise [rpp][NV C].
template <int NV>
__device__ __forceinline__ void ew_block_colsum(float (*v)[8], int C, int tpr, int rpp, int cl, int rl, bool active, bool shfl,
                                                float* red, float* dst) {
  const int W = NV * C, c0 = cl * 8;
  int rows = rpp, row = rl;
  if (shfl) {
    for (int o = 32; o >= tpr; o >>= 1)
#pragma unroll
      for (int s = 0; s < NV; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[s][j] += __shfl_xor(v[s][j], o, 64);
    rows = EW_THREADS / 64;
    row = threadIdx.x >> 6;
    active = (threadIdx.x & 63) < tpr;
  }
  if (active) {
#pragma unroll
    for (int s = 0; s < NV; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) red[row * W + s * C + c0 + j] = v[s][j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < W; i += EW_THREADS) {
    float t = 0.f;
    for (int r = 0; r < rows; ++r) t += red[r * W + i];
    dst[i] = t;
  }
}
static inline bool ew_shfl_ok(int C) { const int tpr = C >> 3; return tpr <= 64 && (tpr & (tpr - 1)) == 0; }
static inline size_t ew_colsum_lds(int C, int nv) {
  return (size_t)(ew_shfl_ok(C) ? EW_THREADS / 64 : rows_per_pass(C)) * nv * C * sizeof(float);
}

// PReLU pre-activation z = G x + H: the forward's own (scale, shift) when the caller has them (same expression as bn_apply, so the
// mask is the forward's), else derived from gamma / rstd / mean / beta
__device__ __forceinline__ void ew_load_gh(const BnBwd& p, int c0, const float* mean, const float* rstd, float* G, float* H) {
  if (p.sc) {
    load8f(p.sc, c0, G, 1.f);
    load8f(p.sh, c0, H, 0.f);
  } else {
    float ga[8], be[8];
    load8f(p.gamma, c0, ga, 1.f);
    load8f(p.beta, c0, be, 0.f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      G[j] = ga[j] * rstd[j];
      H[j] = be[j] - mean[j] * G[j];
    }
  }
}

// NTM: cache policy of the two streams — 0 plain loads, 1 x non-temporal (the forward pass wrote it a whole pass ago: it comes from HBM and
// nobody reads it again soon), 2 both (tools/probe/hbm_stream_probe.hip: a cold r2w1 pass of a 51 MB map takes 48 us with plain loads and 34
// with non-temporal ones)
template <bool ALPHA, int NTM>
__global__ __launch_bounds__(EW_THREADS, ALPHA ? 4 : 5) void bn_bwd_reduce_kernel(BnBwd p, int slab, int shfl) {
  extern __shared__ float red[];
  constexpr int UNR = ALPHA ? EW_UNROLL_ALPHA : EW_UNROLL;     // rows of loads in flight (the PReLU variant carries 24 more per-channel registers)
  const int tpr = p.C >> 3, rpp = EW_THREADS / tpr;
  const int cl = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  const bool active = rl < rpp;
  const int c0 = cl * 8;
  float mean[8], G[8], H[8], al[8];
  load8f(p.mean, c0, mean, 0.f);
  if (ALPHA) {
    float rstd[8];
    load8f(p.rstd, c0, rstd, 1.f);
    ew_load_gh(p, c0, mean, rstd, G, H);
    load8f(p.alpha, c0, al, 1.f);
  }
  float acc[3][8];                  // sum dz | sum dz (x - mean), scaled by rstd at the end = sum dz xhat | sum dy z over z <= 0
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = acc[2][j] = 0.f;
  const int bid = ew_block_id();
  const int mbeg = bid * slab, mend = min(p.M, mbeg + slab);
  auto accum = [&](const uint4& vd, const uint4& vx) {
    float dy[8], x[8];
    unpack8(vd, dy);
    unpack8(vx, x);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float dz = dy[j];
      if (ALPHA) {
        const float z = x[j] * G[j] + H[j];
        if (z <= 0.f) {
          acc[2][j] += dy[j] * z;
          dz = dy[j] * al[j];
        }
      }
      acc[0][j] += dz;
      acc[1][j] += dz * (x[j] - mean[j]);
    }
  };
  if (active) {
    // UNR rows per trip with all their loads issued first: in the network dy / x come cold from HBM and one 16-B load
    // pair in flight per thread left the kernel latency-bound (2x its warm-cache time)
    int m = mbeg + rl;
    for (; m + (UNR - 1) * rpp < mend; m += UNR * rpp) {
      uint4 vd[UNR], vx[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const size_t off = (size_t)(m + u * rpp) * p.C + c0;
        vd[u] = NTM >= 2 ? ew_ld16(p.dy + off) : *reinterpret_cast<const uint4*>(p.dy + off);
        vx[u] = NTM >= 1 ? ew_ld16(p.x + off) : *reinterpret_cast<const uint4*>(p.x + off);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        accum(vd[u], vx[u]);
        __builtin_amdgcn_sched_barrier(0);          // one row's temporaries at a time (the scheduler otherwise interleaves all four: +50 VGPRs)
      }
    }
    for (; m < mend; m += rpp) {
      const size_t off = (size_t)m * p.C + c0;
      accum(*reinterpret_cast<const uint4*>(p.dy + off), *reinterpret_cast<const uint4*>(p.x + off));
    }
  }
  {
    float rstd[8];
    load8f(p.rstd, c0, rstd, 1.f);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[1][j] *= rstd[j];
  }
  ew_block_colsum<3>(acc, p.C, tpr, rpp, cl, rl, active, shfl != 0, red, p.partials + (size_t)bid * 3 * p.C);
}

// workgroups a row-slab bn_bwd_reduce / bn_bwd_apply launch aims for (= partial rows it leaves).  Round 3 swept both and the loads' cache policy
// (profiles/r03_ab_ew_rowslab_options_v1.txt): nothing moved the step, the sweep's switches are gone
// Round 6: the apply pass aims for 768 workgroups (2048 in rounds 3-5) = ONE dispatch round of its three-per-CU variants on 256 CUs, and a third of
// the partial rows for the finalize launch behind it: -0.03 ... -0.04 ms per step same-box on two boxes (profiles/r06_ab_bwd_apply_blocks_v1.txt; 512 and
// 1024, and 256 / 1024 for the reduce pass: equal)
constexpr int kEwReduceBlocks = 512, kEwBwdApplyBlocks = 768;
int ew_bn_bwd_grid(int M, int C) { return ceil_div(M, slab_rows(M, C, kEwReduceBlocks)); }

int ew_bn_bwd_reduce(const BnBwd& p, hipStream_t st) {
  FEDFR_TRY(check_mc(p.M, p.C, "bn_bwd_reduce"));
  FEDFR_REQUIRE(p.dy && p.x && p.partials, "bn_bwd_reduce: null tensor");   // mean / rstd null: 0 / 1 (bias + PReLU backward)
  FEDFR_REQUIRE(!p.sc == !p.sh, "bn_bwd_reduce: scale and shift come together");
  const int slab = slab_rows(p.M, p.C, kEwReduceBlocks);
  const int grid = ceil_div(p.M, slab);
  const size_t lds = ew_colsum_lds(p.C, 3);
  const int shfl = ew_shfl_ok(p.C) ? 1 : 0;
  ProfScope prof(21, (double)p.M * p.C * 2 * 2, st);
#define BWD_RED(A, N) hipLaunchKernelGGL((bn_bwd_reduce_kernel<A, N>), dim3(grid), dim3(EW_THREADS), lds, st, p, slab, shfl)
  if (p.alpha) BWD_RED(true, 0);
  else BWD_RED(false, 0);
#undef BWD_RED
  FEDFR_LAUNCH_CHECK("bn_bwd_reduce");
  return FEDFR_OK;
}

// coef [3][C] for the apply pass, dx = a dz + A x + B  (== a (dz - mean(dz) - xhat mean(dz xhat)), xhat = (x - mean) rstd):
//   a = gamma rstd, A = -a rstd mean(dz xhat), B = a (rstd mean mean(dz xhat) - mean(dz))
__global__ __launch_bounds__(256) void bn_bwd_finalize8_kernel(const float* __restrict__ part, int P, int C, double count,
                                                              const float* gamma, const float* mean, const float* rstd, float* dgamma,
                                                              float* dbeta, float* dalpha, float* coef) {
  __shared__ double tot[3][8];
  const int c0 = blockIdx.x * 8;
  fin8_accumulate<3>(part, P, C, c0, dalpha ? 3 : 2, tot);      // the third statistic (sum dy z over z <= 0) only feeds dalpha
  if (threadIdx.x < 8) {
    const int c = c0 + threadIdx.x;
    const double t1 = tot[0][threadIdx.x], t2 = tot[1][threadIdx.x];
    if (dgamma) dgamma[c] = (float)t2;
    if (dbeta) dbeta[c] = (float)t1;
    if (dalpha) dalpha[c] = (float)tot[2][threadIdx.x];
    const double g = gamma ? (double)gamma[c] : 1.0, r = rstd ? (double)rstd[c] : 1.0, mu = mean ? (double)mean[c] : 0.0;
    const double a = (double)(float)(g * r), cb = t1 * (1.0 / count), cc = t2 * (1.0 / count);
    coef[c] = (float)a;
    coef[C + c] = (float)(-a * cc * r);
    coef[2 * C + c] = (float)(a * (cc * r * mu - cb));
  }
}

int ew_bn_bwd_finalize(const float* partials, int P, int C, double count, const float* gamma, const float* mean, const float* rstd,
                       float* dgamma, float* dbeta, float* dalpha, float* coef, hipStream_t st) {
  FEDFR_REQUIRE(partials && P > 0 && C > 0 && (C & 7) == 0 && coef, "bn_bwd_finalize: bad args");
  ProfScope prof(24, (double)P * 3 * C * 4, st);
  hipLaunchKernelGGL(bn_bwd_finalize8_kernel, dim3(C / 8), dim3(256), 0, st, partials, P, C, count, gamma, mean, rstd,
                     dgamma, dbeta, dalpha, coef);
  FEDFR_LAUNCH_CHECK("bn_bwd_finalize");
  return FEDFR_OK;
}

struct BnBwdDiv {
  FastDiv dHW, dW;
};

template <bool ALPHA, int NX, bool ADD>      // NX: 0 none, 1 the next BatchNorm's reduction rides along, 2 ... and that BatchNorm has a PReLU behind it
__global__ __launch_bounds__(EW_THREADS, ((ALPHA && NX) || NX == 2) ? 3 : (ALPHA || NX || ADD) ? 4 : 5) void bn_bwd_apply_kernel(BnBwd p, BnBwdDiv dv, int slab, int shfl) {
  const int tpr = p.C >> 3, rpp = EW_THREADS / tpr;
  const int cl = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  const bool active = rl < rpp;
  if (!active && !NX) return;
  extern __shared__ float red[];                    // next BN's reduction (NX)
  constexpr int UNR = (ALPHA || NX || ADD) ? EW_UNROLL_HEAVY : EW_UNROLL;   // rows of loads in flight: the heavier variants trade one for registers (two waves beside wgrad9)
  const int c0 = cl * 8;
  float ca[8], cA[8], cB[8], G[8], H[8], al[8], nmean[8], nG[8], nH[8], nal[8];
  float nacc[NX == 2 ? 3 : 2][8];                    // sum dz | sum dz (x_next - mean_next), scaled by rstd_next at the end | (NX == 2) sum dx z over z <= 0
  load8f(p.coef, c0, ca, 1.f);
  load8f(p.coef + p.C, c0, cA, 0.f);
  load8f(p.coef + 2 * p.C, c0, cB, 0.f);
  if (ALPHA) {
    float mean[8], rstd[8];
    load8f(p.sc ? nullptr : p.mean, c0, mean, 0.f);
    load8f(p.sc ? nullptr : p.rstd, c0, rstd, 1.f);
    ew_load_gh(p, c0, mean, rstd, G, H);
    load8f(p.alpha, c0, al, 1.f);
  }
  if (NX) {
    load8f(p.nmean, c0, nmean, 0.f);
#pragma unroll
    for (int j = 0; j < 8; ++j) nacc[0][j] = nacc[1][j] = 0.f;
    if (NX == 2) {
      load8f(p.nsc, c0, nG, 1.f);
      load8f(p.nsh, c0, nH, 0.f);
      load8f(p.nalpha, c0, nal, 1.f);
#pragma unroll
      for (int j = 0; j < 8; ++j) nacc[NX == 2 ? 2 : 0][j] = 0.f;
    }
  }
  const int bid = ew_block_id();
  const int mbeg = bid * slab, mend = min(p.M, mbeg + slab);
  // every tensor of a row (dy, x, the identity-path addend, the next BN's input) is fetched in the batch in front of the arithmetic:
  // loads issued inside the per-row code wait out a full memory latency each
  auto one = [&](int m, const uint4& vd, const uint4& vx, const uint4& va, const uint4& vn) {
    const size_t off = (size_t)m * p.C + c0;
    float dy[8], x[8], o[8];
    unpack8(vd, dy);
    unpack8(vx, x);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float dz = dy[j];
      if (ALPHA) {
        const float z = x[j] * G[j] + H[j];
        if (z <= 0.f) dz = dy[j] * al[j];
      }
      o[j] = ca[j] * dz + (cA[j] * x[j] + cB[j]);
    }
    if (ADD) {
      float a[8];
      unpack8(va, a);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += a[j];
    }
    if (p.add_up) {
      const unsigned img = fdiv((unsigned)m, dv.dHW);
      const unsigned rem = (unsigned)m - img * dv.dHW.d;
      const unsigned h = fdiv(rem, dv.dW), w = rem - h * dv.dW.d;
      if (((h | w) & 1u) == 0u) {
        const size_t uoff = (((size_t)img * (p.H >> 1) + (h >> 1)) * (p.W >> 1) + (w >> 1)) * p.C + c0;
        float a[8];
        unpack8(*reinterpret_cast<const uint4*>(p.add_up + uoff), a);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += a[j];
      }
    }
    const uint4 ov = pack8(o);
    *reinterpret_cast<uint4*>(p.dx + off) = ov;
    if (NX) {                                        // the next BN sees the bf16-rounded dx, exactly as its own reduce pass would
      float dn[8], xn[8];
      unpack8(ov, dn);
      unpack8(vn, xn);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float dzn = dn[j];
        if (NX == 2) {                                 // exactly bn_bwd_reduce's PReLU branch
          const float z = xn[j] * nG[j] + nH[j];
          if (z <= 0.f) {
            nacc[NX == 2 ? 2 : 0][j] += dn[j] * z;
            dzn = dn[j] * nal[j];
          }
        }
        nacc[0][j] += dzn;
        nacc[1][j] += dzn * (xn[j] - nmean[j]);
      }
    }
  };
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  int m = active ? mbeg + rl : mend;
  for (; m + (UNR - 1) * rpp < mend; m += UNR * rpp) {      // loads of UNR rows first (see bn_bwd_reduce)
    uint4 vd[UNR], vx[UNR], va[UNR], vn[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const size_t off = (size_t)(m + u * rpp) * p.C + c0;
      vd[u] = ew_ld16(p.dy + off);
      vx[u] = ew_ld16(p.x + off);
      va[u] = ADD ? *reinterpret_cast<const uint4*>(p.add + off) : zero4;
      vn[u] = NX ? *reinterpret_cast<const uint4*>(p.nx + off) : zero4;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      one(m + u * rpp, vd[u], vx[u], va[u], vn[u]);
      __builtin_amdgcn_sched_barrier(0);            // see bn_bwd_reduce
    }
  }
  for (; m < mend; m += rpp) {
    const size_t off = (size_t)m * p.C + c0;
    one(m, *reinterpret_cast<const uint4*>(p.dy + off), *reinterpret_cast<const uint4*>(p.x + off),
        ADD ? *reinterpret_cast<const uint4*>(p.add + off) : zero4, NX ? *reinterpret_cast<const uint4*>(p.nx + off) : zero4);
  }
  if (NX) {
    float nrstd[8];
    load8f(p.nrstd, c0, nrstd, 1.f);
#pragma unroll
    for (int j = 0; j < 8; ++j) nacc[1][j] *= nrstd[j];
    float* row = p.npart + (size_t)bid * 3 * p.C;
    if (NX == 2) {
      ew_block_colsum<3>(nacc, p.C, tpr, rpp, cl, rl, active, shfl != 0, red, row);
    } else {
      ew_block_colsum<2>(nacc, p.C, tpr, rpp, cl, rl, active, shfl != 0, red, row);
      for (int i = threadIdx.x; i < p.C; i += EW_THREADS) row[2 * p.C + i] = 0.f;
    }
  }
}

int ew_bn_bwd_apply_grid(int M, int C) { return ceil_div(M, slab_rows(M, C, kEwBwdApplyBlocks)); }

int ew_bn_bwd_apply(const BnBwd& p, hipStream_t st) {
  FEDFR_TRY(check_mc(p.M, p.C, "bn_bwd_apply"));
  FEDFR_REQUIRE(p.dy && p.x && p.coef && p.dx, "bn_bwd_apply: null tensor");
  FEDFR_REQUIRE(!p.sc == !p.sh, "bn_bwd_apply: scale and shift come together");
  BnBwdDiv dv;
  dv.dHW = make_fastdiv(1);
  dv.dW = make_fastdiv(1);
  if (p.add_up) {
    FEDFR_REQUIRE(p.H > 0 && p.W > 0 && !(p.H & 1) && !(p.W & 1) && p.M % (p.H * p.W) == 0, "bn_bwd_apply: add_up needs even H, W");
    dv.dHW = make_fastdiv((unsigned)(p.H * p.W));
    dv.dW = make_fastdiv((unsigned)p.W);
  }
  if (p.nx) FEDFR_REQUIRE(p.nmean && p.nrstd && p.npart, "bn_bwd_apply: next-BN reduction needs mean / rstd / partials");
  const bool nxa = p.nx && p.nalpha;
  if (nxa) FEDFR_REQUIRE(p.nsc && p.nsh && !p.alpha && !p.add, "bn_bwd_apply: the PReLU form of the next-BN reduction needs that BN's (scale, shift) and serves the plain variant only");
  const int slab = slab_rows(p.M, p.C, kEwBwdApplyBlocks);
  const dim3 grid(ceil_div(p.M, slab));
  const size_t lds = p.nx ? ew_colsum_lds(p.C, nxa ? 3 : 2) : 0;
  const int shfl = ew_shfl_ok(p.C) ? 1 : 0;
#define BWD_APPLY(A, N, D) hipLaunchKernelGGL((bn_bwd_apply_kernel<A, N, D>), grid, dim3(EW_THREADS), lds, st, p, dv, slab, shfl)
  const int variant = nxa ? 8 : ((p.alpha ? 4 : 0) | (p.nx ? 2 : 0) | (p.add ? 1 : 0));
  // algorithmic bytes: dy + x read, dx written, + the identity addend (compact when up-sampled), + the next BN's input when its reduction rides along
  ProfScope prof(22, (double)p.M * p.C * 2 * (3.0 + (p.add ? 1.0 : 0.0) + (p.add_up ? 0.25 : 0.0) + (p.nx ? 1.0 : 0.0)), st);
  switch (variant) {
    case 0: BWD_APPLY(false, 0, false); break;
    case 1: BWD_APPLY(false, 0, true); break;
    case 2: BWD_APPLY(false, 1, false); break;
    case 3: BWD_APPLY(false, 1, true); break;
    case 4: BWD_APPLY(true, 0, false); break;
    case 5: BWD_APPLY(true, 0, true); break;
    case 6: BWD_APPLY(true, 1, false); break;
    case 7: BWD_APPLY(true, 1, true); break;
    default: BWD_APPLY(false, 2, false); break;
  }
#undef BWD_APPLY
  FEDFR_LAUNCH_CHECK("bn_bwd_apply");
  return FEDFR_OK;
}

// =====================================================================================================
// BatchNorm1d on fp32 [B][C]
// =====================================================================================================
// BatchNorm1d kernels: 64 channels x BN1D_RG row groups per workgroup (one thread per channel walking all B rows three times was a
// 60 us latency chain for a 128 x 512 tensor); fp64 sums, the row groups meet in LDS.
constexpr int BN1D_RG = 8;
__device__ __forceinline__ double bn1d_rg_sum(double v, double (*red)[64], int rg, int cl) {
  __syncthreads();                                   // previous use of red is over
  red[rg][cl] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < BN1D_RG; ++i) t += red[i][cl];
  return t;
}
// CACHE (B <= BN1D_NR x BN1D_RG rows): a thread's rows are fetched ONCE, all loads in flight together, and the three passes run on registers
// (round 4: the kernel sits on the serial head chain between the forward and the backward pass; same operations in the same order)
constexpr int BN1D_NR = 16;
template <bool CACHE>
__global__ __launch_bounds__(64 * BN1D_RG) void bn1d_fwd_kernel(const float* x, float* y, int B, int C, const float* gamma, const float* beta,
                                float* rm, float* rv, float momentum, float eps, int training, float* save_mean,
                                float* save_rstd) {
  __shared__ double red[BN1D_RG][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = min(blockIdx.x * 64 + cl, C - 1);
  const bool own = blockIdx.x * 64 + cl < C && rg == 0;
  float xv[CACHE ? BN1D_NR : 1];
  if constexpr (CACHE) {
#pragma unroll
    for (int j = 0; j < BN1D_NR; ++j) {
      const int b = rg + j * BN1D_RG;
      xv[j] = x[(size_t)min(b, B - 1) * C + c];
    }
  }
  float mean, rstd;
  if (training) {
    double s = 0.0;
    if constexpr (CACHE) {
#pragma unroll
      for (int j = 0; j < BN1D_NR; ++j) if (rg + j * BN1D_RG < B) s += (double)xv[j];
    } else {
      for (int b = rg; b < B; b += BN1D_RG) s += (double)x[(size_t)b * C + c];
    }
    const double mu = bn1d_rg_sum(s, red, rg, cl) / B;
    double v = 0.0;
    if constexpr (CACHE) {
#pragma unroll
      for (int j = 0; j < BN1D_NR; ++j) if (rg + j * BN1D_RG < B) { const double d = (double)xv[j] - mu; v += d * d; }
    } else {
      for (int b = rg; b < B; b += BN1D_RG) {
        const double d = (double)x[(size_t)b * C + c] - mu;
        v += d * d;
      }
    }
    v = bn1d_rg_sum(v, red, rg, cl);
    const double var = v / B;
    mean = (float)mu;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (own) {
      const double unb = B > 1 ? v / (B - 1) : var;
      rm[c] = (float)((1.0 - momentum) * (double)rm[c] + momentum * mu);
      rv[c] = (float)((1.0 - momentum) * (double)rv[c] + momentum * unb);
    }
  } else {
    mean = rm[c];
    rstd = 1.f / sqrtf(rv[c] + eps);
  }
  if (save_mean && own) {
    save_mean[c] = mean;
    save_rstd[c] = rstd;
  }
  if (blockIdx.x * 64 + cl >= C) return;
  const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  if constexpr (CACHE) {
#pragma unroll
    for (int j = 0; j < BN1D_NR; ++j) {
      const int b = rg + j * BN1D_RG;
      if (b < B) y[(size_t)b * C + c] = (xv[j] - mean) * rstd * g + bt;
    }
  } else {
    for (int b = rg; b < B; b += BN1D_RG) y[(size_t)b * C + c] = (x[(size_t)b * C + c] - mean) * rstd * g + bt;
  }
}

int ew_bn1d_fwd(const float* x, float* y, int B, int C, const float* gamma, const float* beta, float* rm, float* rv,
                float momentum, float eps, int training, float* save_mean, float* save_rstd, hipStream_t st) {
  FEDFR_REQUIRE(x && y && B > 0 && C > 0 && rm && rv, "bn1d_fwd: bad args");
  if (B <= BN1D_NR * BN1D_RG)
    hipLaunchKernelGGL(bn1d_fwd_kernel<true>, dim3(ceil_div(C, 64)), dim3(64 * BN1D_RG), 0, st, x, y, B, C, gamma, beta, rm, rv, momentum, eps,
                       training, save_mean, save_rstd);
  else
    hipLaunchKernelGGL(bn1d_fwd_kernel<false>, dim3(ceil_div(C, 64)), dim3(64 * BN1D_RG), 0, st, x, y, B, C, gamma, beta, rm, rv, momentum, eps,
                       training, save_mean, save_rstd);
  FEDFR_LAUNCH_CHECK("bn1d_fwd");
  return FEDFR_OK;
}

template <bool CACHE>
__global__ __launch_bounds__(64 * BN1D_RG) void bn1d_bwd_kernel(const float* dy, const float* x, float* dx, int B, int C, const float* gamma,
                                const float* mean, const float* rstd, float* dbeta, float* dx_colsum, bf16_t* dxb,
                                bf16_t* dxbt, int ldt, int frozen) {
  __shared__ double red[BN1D_RG][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = min(blockIdx.x * 64 + cl, C - 1);
  const bool valid = blockIdx.x * 64 + cl < C, own = valid && rg == 0;
  float dv[CACHE ? BN1D_NR : 1], xv[CACHE ? BN1D_NR : 1];
  if constexpr (CACHE) {
#pragma unroll
    for (int j = 0; j < BN1D_NR; ++j) {
      const size_t o = (size_t)min(rg + j * BN1D_RG, B - 1) * C + c;
      dv[j] = dy[o];
      xv[j] = x[o];
    }
  }
  const float mu = mean[c], rs = rstd[c], g = gamma ? gamma[c] : 1.f;
  double s1 = 0.0, s2 = 0.0;
  if constexpr (CACHE) {
#pragma unroll
    for (int j = 0; j < BN1D_NR; ++j)
      if (rg + j * BN1D_RG < B) {
        s1 += dv[j];
        s2 += (double)dv[j] * (double)((xv[j] - mu) * rs);
      }
  } else {
    for (int b = rg; b < B; b += BN1D_RG) {
      const float d = dy[(size_t)b * C + c];
      s1 += d;
      s2 += (double)d * (double)((x[(size_t)b * C + c] - mu) * rs);
    }
  }
  s1 = bn1d_rg_sum(s1, red, rg, cl);
  s2 = bn1d_rg_sum(s2, red, rg, cl);
  const float m1 = frozen ? 0.f : (float)(s1 / B), m2 = frozen ? 0.f : (float)(s2 / B);     // frozen: statistics were constants, dx = g rstd dy
  if (dbeta && own) dbeta[c] = (float)s1;
  double cs = 0.0;
  auto one = [&](int b, float d, float xx) {
    const float xh = (xx - mu) * rs;
    const float v = g * rs * (d - m1 - xh * m2);
    dx[(size_t)b * C + c] = v;
    cs += v;
    if (dxb) dxb[(size_t)b * C + c] = f2bf(v);
    if (dxbt) dxbt[(size_t)c * ldt + b] = f2bf(v);
  };
  if (valid) {
    if constexpr (CACHE) {
#pragma unroll
      for (int j = 0; j < BN1D_NR; ++j) if (rg + j * BN1D_RG < B) one(rg + j * BN1D_RG, dv[j], xv[j]);
    } else {
      for (int b = rg; b < B; b += BN1D_RG) one(b, dy[(size_t)b * C + c], x[(size_t)b * C + c]);
    }
  }
  cs = bn1d_rg_sum(cs, red, rg, cl);
  if (dx_colsum && own) dx_colsum[c] = (float)cs;
}

int ew_bn1d_bwd(const float* dy, const float* x, float* dx, int B, int C, const float* gamma, const float* mean,
                const float* rstd, float* dbeta, float* dx_colsum, bf16_t* dxb, bf16_t* dxbt, int ldt, hipStream_t st, int frozen) {
  FEDFR_REQUIRE(dy && x && dx && B > 0 && C > 0 && mean && rstd, "bn1d_bwd: bad args");
  if (B <= BN1D_NR * BN1D_RG)
    hipLaunchKernelGGL(bn1d_bwd_kernel<true>, dim3(ceil_div(C, 64)), dim3(64 * BN1D_RG), 0, st, dy, x, dx, B, C, gamma, mean, rstd, dbeta,
                       dx_colsum, dxb, dxbt, ldt, frozen);
  else
    hipLaunchKernelGGL(bn1d_bwd_kernel<false>, dim3(ceil_div(C, 64)), dim3(64 * BN1D_RG), 0, st, dy, x, dx, B, C, gamma, mean, rstd, dbeta,
                       dx_colsum, dxb, dxbt, ldt, frozen);
  FEDFR_LAUNCH_CHECK("bn1d_bwd");
  return FEDFR_OK;
}

// =====================================================================================================
// split-K slab reductions, casts, transposes
// =====================================================================================================
// blockIdx.y = 1 (ew_reduce_slabs2): the second (dst, slabs) pair of the launch, same geometry
__global__ void reduce_slabs_kernel(float* dst, const float* slabs, int nsplit, size_t n4, const float* bias, int bias_n, float* dst1,
                                    const float* slabs1) {
  if (blockIdx.y) { dst = dst1; slabs = slabs1; }
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 a = reinterpret_cast<const float4*>(slabs)[i];
    for (int s = 1; s < nsplit; ++s) {
      const float4 b = reinterpret_cast<const float4*>(slabs + (size_t)s * n4 * 4)[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (bias) {
      const int c = (int)((i * 4) % (size_t)bias_n);
      a.x += bias[c]; a.y += bias[c + 1]; a.z += bias[c + 2]; a.w += bias[c + 3];
    }
    reinterpret_cast<float4*>(dst)[i] = a;
  }
}
// many slabs of a small tensor (the 64-channel layers' weight gradients arrive as 128 slabs of 147 KB): one thread per output walking
// all slabs is a serial chain of 128 loads on 9 K threads (93 us).  Here 8 lanes share an output (slab s goes to lane s % 8, four loads
// in flight each) and meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void reduce_slabs_wide_kernel(float* dst, const float* slabs, int nsplit, size_t n4, const float* bias, int bias_n,
                                                                float* dst1, const float* slabs1) {
  if (blockIdx.y) { dst = dst1; slabs = slabs1; }
  __shared__ float4 red[8][32];
  const int o = threadIdx.x & 31, q = threadIdx.x >> 5;
  const size_t i = (size_t)blockIdx.x * 32 + o;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const float4* src = reinterpret_cast<const float4*>(slabs) + i;
    int sidx = q;
    for (; sidx + 24 < nsplit; sidx += 32) {
      const float4 b0 = src[(size_t)sidx * n4], b1 = src[(size_t)(sidx + 8) * n4], b2 = src[(size_t)(sidx + 16) * n4], b3 = src[(size_t)(sidx + 24) * n4];
      a.x += (b0.x + b1.x) + (b2.x + b3.x); a.y += (b0.y + b1.y) + (b2.y + b3.y);
      a.z += (b0.z + b1.z) + (b2.z + b3.z); a.w += (b0.w + b1.w) + (b2.w + b3.w);
    }
    for (; sidx < nsplit; sidx += 8) {
      const float4 b = src[(size_t)sidx * n4];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
  }
  red[q][o] = a;
  __syncthreads();
  if (q == 0 && i < n4) {
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const float4 b = red[k][o];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (bias) {
      const int c = (int)((i * 4) % (size_t)bias_n);
      a.x += bias[c]; a.y += bias[c + 1]; a.z += bias[c + 2]; a.w += bias[c + 3];
    }
    reinterpret_cast<float4*>(dst)[i] = a;
  }
}
static int reduce_slabs_launch(float* dst, const float* slabs, float* dst1, const float* slabs1, int nsplit, size_t n, const float* bias, int bias_n,
                               hipStream_t st) {
  FEDFR_REQUIRE(dst && slabs && nsplit > 0 && n > 0 && (n & 3) == 0, "reduce_slabs: bad args (n%%4)");
  if (bias) FEDFR_REQUIRE((bias_n & 3) == 0 && bias_n > 0, "reduce_slabs: bias_n%%4");
  const size_t n4 = n / 4;
  const unsigned ny = dst1 ? 2 : 1;
  ProfScope prof(25, (double)n * 4 * (nsplit + 1) * ny, st);
  if (nsplit >= 16 && n4 <= 65536) {
    hipLaunchKernelGGL(reduce_slabs_wide_kernel, dim3((unsigned)((n4 + 31) / 32), ny), dim3(256), 0, st, dst, slabs, nsplit, n4, bias, bias_n, dst1, slabs1);
    FEDFR_LAUNCH_CHECK("reduce_slabs_wide");
    return FEDFR_OK;
  }
  const int grid = (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(grid, ny), dim3(256), 0, st, dst, slabs, nsplit, n4, bias, bias_n, dst1, slabs1);
  FEDFR_LAUNCH_CHECK("reduce_slabs");
  return FEDFR_OK;
}
int ew_reduce_slabs(float* dst, const float* slabs, int nsplit, size_t n, const float* bias, int bias_n, hipStream_t st) {
  return reduce_slabs_launch(dst, slabs, nullptr, nullptr, nsplit, n, bias, bias_n, st);
}
// two slab sets of the same geometry (the two 3x3 weight gradients of a residual block) in one launch
int ew_reduce_slabs2(float* dst0, const float* slabs0, float* dst1, const float* slabs1, int nsplit, size_t n, hipStream_t st) {
  FEDFR_REQUIRE(dst1 && slabs1, "reduce_slabs2: null second pair");
  return reduce_slabs_launch(dst0, slabs0, dst1, slabs1, nsplit, n, nullptr, 0, st);
}

__global__ void reduce_slabs_bf16_kernel(bf16_t* dst, const float* slabs, int nsplit, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 a = reinterpret_cast<const float4*>(slabs)[i];
    for (int s = 1; s < nsplit; ++s) {
      const float4 b = reinterpret_cast<const float4*>(slabs + (size_t)s * n4 * 4)[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    uint2 o;
    o.x = pack_bf2(a.x, a.y);
    o.y = pack_bf2(a.z, a.w);
    reinterpret_cast<uint2*>(dst)[i] = o;
  }
}
int ew_reduce_slabs_bf16(bf16_t* dst, const float* slabs, int nsplit, size_t n, hipStream_t st) {
  FEDFR_REQUIRE(dst && slabs && nsplit > 0 && n > 0 && (n & 3) == 0, "reduce_slabs_bf16: bad args (n%%4)");
  const size_t n4 = n / 4;
  const int grid = (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
  hipLaunchKernelGGL(reduce_slabs_bf16_kernel, dim3(grid), dim3(256), 0, st, dst, slabs, nsplit, n4);
  FEDFR_LAUNCH_CHECK("reduce_slabs_bf16");
  return FEDFR_OK;
}

__global__ void cast_f32_bf16_kernel(const float* src, bf16_t* dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t n8 = n / 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
    uint4 o;
    o.x = pack_bf2(a.x, a.y); o.y = pack_bf2(a.z, a.w); o.z = pack_bf2(b.x, b.y); o.w = pack_bf2(b.z, b.w);
    reinterpret_cast<uint4*>(dst)[i] = o;
  }
  for (size_t i = n8 * 8 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = f2bf(src[i]);
}
int ew_cast_f32_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t st) {
  FEDFR_REQUIRE(src && dst && n > 0, "cast_f32_bf16: bad args");
  FEDFR_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "cast_f32_bf16: 16-byte alignment");
  const size_t work = n / 8 + 1;
  const int grid = (int)((work + 255) / 256 > 4096 ? 4096 : (work + 255) / 256);
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid), dim3(256), 0, st, src, dst, n);
  FEDFR_LAUNCH_CHECK("cast_f32_bf16");
  return FEDFR_OK;
}

// [Cout][RS][Cin] fp32 -> [Cin][RS(flipped)][Cout] bf16, 64x64 LDS-tiled transpose
__global__ __launch_bounds__(256) void weight_dgrad_shadow_kernel(const float* __restrict__ w, bf16_t* __restrict__ dst,
                                                                  int Cout, int RS, int Cin) {
  __shared__ float tile[64][65];
  const int co0 = blockIdx.x * 64, ci0 = blockIdx.y * 64, tap = blockIdx.z;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (t >> 4) + 16 * i, col = (t & 15) * 4;
    const float4 v = *reinterpret_cast<const float4*>(w + ((size_t)(co0 + row) * RS + tap) * Cin + ci0 + col);
    tile[row][col] = v.x; tile[row][col + 1] = v.y; tile[row][col + 2] = v.z; tile[row][col + 3] = v.w;
  }
  __syncthreads();
  const int tapf = RS - 1 - tap;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ci = (t >> 3) + 32 * i, co8 = (t & 7) * 8;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = tile[co8 + j][ci];
    *reinterpret_cast<uint4*>(dst + ((size_t)(ci0 + ci) * RS + tapf) * Cout + co0 + co8) = pack8(f);
  }
}
// the same transpose for every conv of a network: workgroup -> (layer, co tile, ci tile, tap) through the by-value table
__global__ __launch_bounds__(256) void weight_dgrad_shadow_multi_kernel(const float* __restrict__ params, bf16_t* __restrict__ shadow, ShadowTable tb) {
  __shared__ float tile[64][65];
  int lo = 0, hi = tb.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tb.e[mid].first_blk <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const ShadowEntry e = tb.e[lo];
  int b = (int)(blockIdx.x - e.first_blk);
  const int cob = b % e.cout64; b /= e.cout64;
  const int cib = b % e.cin64;
  const int tap = b / e.cin64;
  const int Cout = e.cout64 * 64, Cin = e.cin64 * 64, RS = e.rs;
  const float* w = params + e.src;
  bf16_t* dst = shadow + e.dst;
  const int co0 = cob * 64, ci0 = cib * 64, t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (t >> 4) + 16 * i, col = (t & 15) * 4;
    const float4 v = *reinterpret_cast<const float4*>(w + ((size_t)(co0 + row) * RS + tap) * Cin + ci0 + col);
    tile[row][col] = v.x; tile[row][col + 1] = v.y; tile[row][col + 2] = v.z; tile[row][col + 3] = v.w;
  }
  __syncthreads();
  const int tapf = RS - 1 - tap;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ci = (t >> 3) + 32 * i, co8 = (t & 7) * 8;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = tile[co8 + j][ci];
    *reinterpret_cast<uint4*>(dst + ((size_t)(ci0 + ci) * RS + tapf) * Cout + co0 + co8) = pack8(f);
  }
}
int ew_weight_dgrad_shadow_multi(const float* params, bf16_t* shadow, ShadowTable& t, hipStream_t st) {
  FEDFR_REQUIRE(params && shadow && t.n > 0 && t.n <= kMaxShadowEntries, "weight_dgrad_shadow_multi: bad table");
  unsigned blk = 0;
  for (int i = 0; i < t.n; ++i) {
    t.e[i].first_blk = blk;
    blk += (unsigned)t.e[i].cout64 * t.e[i].cin64 * t.e[i].rs;
  }
  hipLaunchKernelGGL(weight_dgrad_shadow_multi_kernel, dim3(blk), dim3(256), 0, st, params, shadow, t);
  FEDFR_LAUNCH_CHECK("weight_dgrad_shadow_multi");
  return FEDFR_OK;
}

int ew_weight_dgrad_shadow(const float* w, bf16_t* dst, int Cout, int R, int S, int Cin, hipStream_t st) {
  FEDFR_REQUIRE(w && dst && (Cout & 63) == 0 && (Cin & 63) == 0 && R > 0 && S > 0, "weight_dgrad_shadow: need Cout,Cin %%64==0");
  hipLaunchKernelGGL(weight_dgrad_shadow_kernel, dim3(Cout / 64, Cin / 64, R * S), dim3(256), 0, st, w, dst, Cout, R * S, Cin);
  FEDFR_LAUNCH_CHECK("weight_dgrad_shadow");
  return FEDFR_OK;
}

// [B][C][HW] fp32 -> [B][HW][C] bf16 through a 64 x 64 LDS tile: coalesced on both sides (round 4: the one-element-per-thread gather read with a
// stride of HW floats took 26 us for the 6.4 MB gradient of the flattened 7x7 map at the top of every backward pass)
__global__ __launch_bounds__(256) void nchw_f32_to_nhwc_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int C, int HW) {
  __shared__ float tile[64][65];
  const int t = threadIdx.x, b = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
  const float* s = src + (size_t)b * C * HW;
#pragma unroll 4
  for (int r = t >> 6; r < 64; r += 4) {
    const int c = c0 + r, p = p0 + (t & 63);
    tile[r][t & 63] = (c < C && p < HW) ? s[(size_t)c * HW + p] : 0.f;
  }
  __syncthreads();
  bf16_t* d = dst + (size_t)b * HW * C;
#pragma unroll 4
  for (int pp = t >> 5; pp < 64; pp += 8) {
    const int p = p0 + pp, c = c0 + (t & 31) * 2;
    if (p < HW && c + 1 < C) *reinterpret_cast<unsigned*>(d + (size_t)p * C + c) = pack_bf2(tile[(t & 31) * 2][pp], tile[(t & 31) * 2 + 1][pp]);      // (C is even: launcher)
  }
}
int ew_nchw_f32_to_nhwc_bf16(const float* src, bf16_t* dst, int B, int C, int HW, hipStream_t st) {
  FEDFR_REQUIRE(src && dst && B > 0 && C > 0 && HW > 0 && (C & 1) == 0 && B <= 65535 && (C + 63) / 64 <= 65535, "nchw_f32_to_nhwc_bf16: bad args (C even)");
  hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_kernel, dim3((HW + 63) / 64, (C + 63) / 64, B), dim3(256), 0, st, src, dst, C, HW);
  FEDFR_LAUNCH_CHECK("nchw_f32_to_nhwc_bf16");
  return FEDFR_OK;
}

__global__ void transpose_bf16_kernel(const bf16_t* src, bf16_t* dst, int R, int C) {
  const size_t n = (size_t)R * C;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int r = (int)(i % R), c = (int)(i / R);
    dst[i] = src[(size_t)r * C + c];
  }
}
int ew_transpose_bf16(const bf16_t* src, bf16_t* dst, int R, int C, hipStream_t st) {
  FEDFR_REQUIRE(src && dst && R > 0 && C > 0, "transpose_bf16: bad args");
  const size_t n = (size_t)R * C;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3(grid), dim3(256), 0, st, src, dst, R, C);
  FEDFR_LAUNCH_CHECK("transpose_bf16");
  return FEDFR_OK;
}

// =====================================================================================================
// stem conv 3 -> 64, 3x3 s1 p1, fp32 NCHW input, one 16x16x32 MFMA per 16 pixels x 16 channels (K = 27 -> 32)
// =====================================================================================================
#if FEDFR_FP16
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#else
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#endif

__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       bf16_t* __restrict__ y, float* __restrict__ stats, int B, int H,
                                                       int W, int ntiles) {
  __shared__ __attribute__((aligned(16))) unsigned char sC[256 * 144];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int M = B * H * W;
  // weight fragments (A operand: row n = ni*16 + l15, k = 8*lg + j); KRSC index = n*27 + k
  bf16x8_t wf[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    s16x8_t v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * lg + j;
      v[j] = (short)f2bf(k < 27 ? w[(ni * 16 + l15) * 27 + k] : 0.f);
    }
    wf[ni] = __builtin_bit_cast(bf16x8_t, v);
  }
  // this lane's eight im2col columns k = 8 lg + j: tap offsets and the element offset relative to (img, channel 0, h, w).  The gather is
  // branch-free (raw buffer loads, an out-of-image tap reads beyond the buffer = 0): predicated loads made hipcc wait for each one in turn,
  // and the kernel ran at 1.3 TB/s of the 224 MB it moves.  All 32 loads of a tile are in flight before the first conversion.
  int dr[8], ds[8], koff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * lg + j;
    const int tap = k / 3, ci = k - tap * 3;
    const int r = tap / 3, s_ = tap - r * 3;
    dr[j] = k < 27 ? r - 1 : (1 << 20);                  // k >= 27: never inside the image
    ds[j] = s_ - 1;
    koff[j] = (ci * H + (r - 1)) * W + (s_ - 1);
  }
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)((size_t)B * 3 * H * W * 4), 0x00020000);
  // a workgroup walks tiles of 256 pixels (the weight fragments above are set up once); the statistics rows keep the per-tile layout
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const int mblk = tile * 256 + wave * 64;
  f32x4_t acc[4][4];
  unsigned raw[4][8];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = mblk + mi * 16 + l15;
    const bool okm = m < M;
    const int mm = okm ? m : 0;
    const int img = mm / (H * W), rem = mm - img * H * W;
    const int h = rem / W, wq = rem - h * W;
    const int base = (img * 3 * H + h) * W + wq;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = okm & ((unsigned)(h + dr[j]) < (unsigned)H) & ((unsigned)(wq + ds[j]) < (unsigned)W);
      raw[mi][j] = __builtin_amdgcn_raw_buffer_load_b32(rsX, ok ? (base + koff[j]) * 4 : (int)0xfffffff0u, 0, 0);
    }
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    s16x8_t v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (short)f2bf(__uint_as_float(raw[mi][j]));
    const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, v);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const f32x4_t z4 = {0.f, 0.f, 0.f, 0.f};
      acc[ni][mi] = MFMA16(wf[ni], xf, z4);
    }
  }
  // D: n = ni*16 + lg*4 + reg ; m(local) = wave*64 + mi*16 + l15
  float ssum[4][4], ssq[4][4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int q = 0; q < 4; ++q) ssum[ni][q] = ssq[ni][q] = 0.f;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      bf16_t hh[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        hh[q] = f2bf(acc[ni][mi][q]);
        const float v = bf2f(hh[q]);
        ssum[ni][q] += v;
        ssq[ni][q] += v * v;
      }
      uint2 pk;
      pk.x = (unsigned)hh[0] | ((unsigned)hh[1] << 16);
      pk.y = (unsigned)hh[2] | ((unsigned)hh[3] << 16);
      *reinterpret_cast<uint2*>(sC + (wave * 64 + mi * 16 + l15) * 144 + (ni * 16 + lg * 4) * 2) = pk;
    }
  if (stats) {
    float* prow = stats + (size_t)(tile * 4 + wave) * 128;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float a = row16_sum(ssum[ni][q]), b = row16_sum(ssq[ni][q]);      // DPP adds (the shuffle form was 128 ds_bpermute per tile)
        if (l15 == 0) {
          prow[ni * 16 + lg * 4 + q] = a;
          prow[64 + ni * 16 + lg * 4 + q] = b;
        }
      }
  }
  __syncthreads();
  for (int idx = tid; idx < 256 * 8; idx += 256) {
    const int row = idx >> 3, c = idx & 7;
    const int m = tile * 256 + row;
    if (m < M) *reinterpret_cast<uint4*>(y + (size_t)m * 64 + c * 8) = *reinterpret_cast<const uint4*>(sC + row * 144 + c * 16);
  }
  __syncthreads();                                     // the staged tile has been read: the next one may be written
  }
}

int ew_stem_stat_rows(int B, int H, int W) { return ceil_div((long long)B * H * W, 256) * 4; }

int ew_stem_fwd(const float* x, const float* w, bf16_t* y, float* stats, int B, int H, int W, hipStream_t st) {
  FEDFR_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0, "stem_fwd: bad args");
  const int M = B * H * W;
  FEDFR_REQUIRE((size_t)B * 3 * H * W * 4 < (1ull << 31), "stem_fwd: input larger than 2 GiB (32-bit buffer offsets)");
  const int ntiles = ceil_div(M, 256);
  hipLaunchKernelGGL(stem_fwd_kernel, dim3(ntiles < 2048 ? ntiles : 2048), dim3(256), 0, st, x, w, y, stats, B, H, W, ntiles);
  FEDFR_LAUNCH_CHECK("stem_fwd");
  return FEDFR_OK;
}

// stem wgrad: dw[co][k] = sum_m dy[m][co] * col[m][k]; VALU, LDS-staged (5.5 GFLOP at B=128: not worth MFMA staging)
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const bf16_t* __restrict__ dy,
                                                         float* __restrict__ tmp, int B, int H, int W, int px_per_block) {
  __shared__ float sdy[64][64];
  __shared__ __attribute__((aligned(16))) float scol[64][32];
  const int tid = threadIdx.x;
  const int co = tid & 63, kq = tid >> 6;
  const int M = B * H * W;
  const int mbeg = blockIdx.x * px_per_block, mend = min(M, mbeg + px_per_block);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int mc = mbeg; mc < mend; mc += 64) {
    // dy tile: 64 px x 64 co bf16 = 512 chunks of 16 B
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i;
      const int px = idx >> 3, c8 = (idx & 7) * 8;
      const int m = mc + px;
      float f[8];
      if (m < mend) {
        unpack8(*reinterpret_cast<const uint4*>(dy + (size_t)m * 64 + c8), f);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) sdy[px][c8 + j] = f[j];
    }
    {
      const int px = tid & 63;
      const int m = mc + px;
      const bool okm = m < mend;
      const int mm = okm ? m : 0;
      const int img = mm / (H * W), rem = mm - img * H * W;
      const int h = rem / W, wq = rem - h * W;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int k = (tid >> 6) + 4 * jj;
        const int tap = k / 3, ci = k - tap * 3;
        const int r = tap / 3, s = tap - r * 3;
        const int hp = h + r - 1, wp = wq + s - 1;
        float f = 0.f;
        if (okm && k < 27 && (unsigned)hp < (unsigned)H && (unsigned)wp < (unsigned)W)
          f = x[(((size_t)img * 3 + ci) * H + hp) * W + wp];
        scol[px][k] = f;
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int px = 0; px < 64; ++px) {
      const float d = sdy[px][co];
      const float4 c0 = *reinterpret_cast<const float4*>(&scol[px][kq * 8]);
      const float4 c1 = *reinterpret_cast<const float4*>(&scol[px][kq * 8 + 4]);
      acc[0] += d * c0.x; acc[1] += d * c0.y; acc[2] += d * c0.z; acc[3] += d * c0.w;
      acc[4] += d * c1.x; acc[5] += d * c1.y; acc[6] += d * c1.z; acc[7] += d * c1.w;
    }
    __syncthreads();
  }
  float* o = tmp + (size_t)blockIdx.x * 2048 + co * 32 + kq * 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = acc[j];
}

__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ tmp, int nblk, float* __restrict__ dw) {
  __shared__ double red[8][32];
  const int co = blockIdx.x, k = threadIdx.x & 31, g = threadIdx.x >> 5;
  double s = 0.0;
  int b = g;
  for (; b + 24 < nblk; b += 32) {                  // four independent loads per trip: the walk over ~1000 blocks is a latency chain
    const float v0 = tmp[(size_t)b * 2048 + co * 32 + k], v1 = tmp[(size_t)(b + 8) * 2048 + co * 32 + k];
    const float v2 = tmp[(size_t)(b + 16) * 2048 + co * 32 + k], v3 = tmp[(size_t)(b + 24) * 2048 + co * 32 + k];
    s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
  }
  for (; b < nblk; b += 8) s += (double)tmp[(size_t)b * 2048 + co * 32 + k];
  red[g][k] = s;
  __syncthreads();
  if (g == 0 && k < 27) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += red[i][k];
    dw[co * 27 + k] = (float)t;
  }
}

// ---- stem wgrad on MFMA --------------------------------------------------------------------------------------------------
// dw[co][k] = sum_px dy[px][co] * col[px][k] as a GEMM with the pixels as the reduction: A = dy^T (transpose reads of the [px][co] tile,
// gemm_tn_dev.h), B = col kept TRANSPOSED in LDS ([k][px], built by the threads from the fp32 NCHW input), split into a bf16 high and
// low part (x = hi + lo to 2^-17: two MFMAs per tile keep the fp32-input accuracy the VALU kernel had).  The VALU kernel above was
// compute-bound (244 us for 224 MB of traffic at B = 128); this one streams.
#ifndef STEM_WGRAD_MFMA
#define STEM_WGRAD_MFMA 1
#endif
#ifndef STEM_WGRAD_PX
#define STEM_WGRAD_PX 128
#endif
constexpr int SW_PX = STEM_WGRAD_PX;          // pixels per stage (256 threads: SW_PX pixels x 256 / SW_PX slices of the 32 im2col columns)
constexpr int SW_CPITCH = 2 * SW_PX + 16;     // colT row pitch in bytes: +16 B keeps the 16 k-rows of a b128 fragment read on distinct banks
// FUSE (round 6): dy is the gradient wrt the stem's ACTIVATION and the BatchNorm + PReLU backward is applied to the tile on its way into LDS — the same
// expression, in the same order, rounded to the same 16 bits as bn_bwd_apply_kernel<true, 0, false> would have stored it (dz = z <= 0 ? dy alpha : dy with
// z = G x0 + H, o = a dz + (A x0 + B)): the 205 MB tensor d(conv output) that only this kernel ever read is neither written nor read back.
struct StemFuse {
  const bf16_t* x0;       // the stem conv's raw output [px][64]
  const float* coef;      // [3][64] from bn_bwd_finalize8_kernel
  const float *sc, *sh;   // the forward's (scale, shift): the PReLU mask is the forward's
  const float* alpha;
};
template <bool FUSE>
__global__ __launch_bounds__(256) void stem_wgrad_mfma_kernel(const float* __restrict__ x, const bf16_t* __restrict__ dy, StemFuse fz,
                                                              float* __restrict__ tmp, int B, int H, int W, int px_per_block) {
  __shared__ __attribute__((aligned(16))) unsigned char sdy[SW_PX * 128];          // [px][64 co] bf16, chunk-swizzled (tn_swz<128>)
  __shared__ __attribute__((aligned(16))) unsigned char scol[2][32 * SW_CPITCH];   // hi / lo: [k 0..31][px] bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = B * H * W;
  const int mbeg = blockIdx.x * px_per_block, mend = min(M, mbeg + px_per_block);
  f32x4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};                  // wave = co block of 16; two k blocks of 16
  const int foff = tn_frag_off<128>(wave * 16, lane);
  const int kg = lane >> 4, kn = lane & 15;
  // this thread's im2col columns k0 .. k0 + KSL - 1 of pixel pxl: tap offsets and element offsets relative to (img, channel 0, h, w).
  // Both operands are fetched with branch-free buffer loads (out of range = 0), all of a stage's loads issued before the first is used:
  // the predicated form made hipcc wait for every load in turn (stem_fwd_kernel has the same history).
  constexpr int KSL = 32 * SW_PX / 256;                 // im2col columns per thread
  const int pxl = tid % SW_PX, k0 = (tid / SW_PX) * KSL;
  int dr[KSL], ds[KSL], koff[KSL];
#pragma unroll
  for (int kk = 0; kk < KSL; ++kk) {
    const int k = k0 + kk;
    const int tap = k / 3, ci = k - tap * 3;
    const int r = tap / 3, sx = tap - r * 3;
    dr[kk] = k < 27 ? r - 1 : (1 << 20);
    ds[kk] = sx - 1;
    koff[kk] = (ci * H + (r - 1)) * W + (sx - 1);
  }
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)((size_t)B * 3 * H * W * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dy) + (size_t)mbeg * 64, 0, (int)((size_t)(mend - mbeg) * 128), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(FUSE ? fz.x0 : dy) + (size_t)mbeg * 64, 0, (int)((size_t)(mend - mbeg) * 128), 0x00020000);
  float ca[8], cA[8], cB[8], G[8], Hs[8], al[8];      // FUSE: this thread's 16-byte chunk is the same eight channels in every stage (256 % 8 == 0)
  if (FUSE) {
    const int c0 = (tid & 7) * 8;
    load8f(fz.coef, c0, ca, 1.f);
    load8f(fz.coef + 64, c0, cA, 0.f);
    load8f(fz.coef + 128, c0, cB, 0.f);
    load8f(fz.sc, c0, G, 1.f);
    load8f(fz.sh, c0, Hs, 0.f);
    load8f(fz.alpha, c0, al, 1.f);
  }
  for (int mc = mbeg; mc < mend; mc += SW_PX) {
    uint4 dv[SW_PX / 32], xv[SW_PX / 32];
    unsigned raw[KSL];
    // dy tile: SW_PX px x 8 chunks of 16 B (rows at or beyond mend lie beyond this workgroup's descriptor: zeros)
#pragma unroll
    for (int i = 0; i < SW_PX / 32; ++i) {
      const int idx = tid + 256 * i;
      const int px = idx >> 3, c = idx & 7;
      const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsD, (mc - mbeg + px) * 128 + c * 16, 0, 0);
      dv[i] = make_uint4(v[0], v[1], v[2], v[3]);
      if (FUSE) {
        const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rsX0, (mc - mbeg + px) * 128 + c * 16, 0, 0);
        xv[i] = make_uint4(q[0], q[1], q[2], q[3]);
      }
    }
    {   // im2col of this thread's pixel
      const int m = mc + pxl;
      const bool okm = m < mend;
      const int mm = okm ? m : 0;
      const int img = mm / (H * W), rem = mm - img * H * W;
      const int h = rem / W, wq = rem - h * W;
      const int base = (img * 3 * H + h) * W + wq;
#pragma unroll
      for (int kk = 0; kk < KSL; ++kk) {
        const bool ok = okm & ((unsigned)(h + dr[kk]) < (unsigned)H) & ((unsigned)(wq + ds[kk]) < (unsigned)W);
        raw[kk] = __builtin_amdgcn_raw_buffer_load_b32(rsX, ok ? (base + koff[kk]) * 4 : (int)0xfffffff0u, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < SW_PX / 32; ++i) {
      const int idx = tid + 256 * i;
      const int px = idx >> 3, c = idx & 7;
      if (FUSE) {
        float d[8], x0[8], o[8];
        unpack8(dv[i], d);
        unpack8(xv[i], x0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float dz = d[j];
          const float z = x0[j] * G[j] + Hs[j];
          if (z <= 0.f) dz = d[j] * al[j];
          o[j] = ca[j] * dz + (cA[j] * x0[j] + cB[j]);
        }
        dv[i] = mc + px < mend ? pack8(o) : make_uint4(0, 0, 0, 0);      // rows beyond this workgroup's range read as zeros, and B is not zero
      }
      *reinterpret_cast<uint4*>(sdy + px * 128 + ((c ^ tn_swz<128>(px)) << 4)) = dv[i];
    }
    // transposed: colT[k][px], k = (tap, ci), rows 27..31 zero
#pragma unroll
    for (int kk = 0; kk < KSL; ++kk) {
      const int k = k0 + kk;
      const float f = __uint_as_float(raw[kk]);
      const bf16_t hi = f2bf(f);
      const bf16_t lo = f2bf(f - bf2f(hi));
      *reinterpret_cast<bf16_t*>(scol[0] + k * SW_CPITCH + pxl * 2) = hi;
      *reinterpret_cast<bf16_t*>(scol[1] + k * SW_CPITCH + pxl * 2) = lo;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < SW_PX / 32; ++ks) {
      const bf16x8_t a = tn_frag_tr<128>(sdy + (ks >> 1) * 64 * 128, foff, ks & 1);      // dy^T fragment: 16 co x 32 px
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int boff = (kb * 16 + kn) * SW_CPITCH + (ks * 32 + kg * 8) * 2;            // colT[k][px .. px+7]
        const bf16x8_t bh = *reinterpret_cast<const bf16x8_t*>(scol[0] + boff);
        const bf16x8_t bl = *reinterpret_cast<const bf16x8_t*>(scol[1] + boff);
        acc[kb] = MFMA16(a, bh, acc[kb]);
        acc[kb] = MFMA16(a, bl, acc[kb]);
      }
    }
    __syncthreads();
  }
  // D[m = co][n = k]: lane holds co = wave 16 + (lane >> 4) 4 + r, k = kb 16 + (lane & 15)
  float* o = tmp + (size_t)blockIdx.x * 2048;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) o[(wave * 16 + (lane >> 4) * 4 + r) * 32 + kb * 16 + (lane & 15)] = acc[kb][r];
}

static int stem_px_per_block(int M) {
  int ppb = ceil_div(M, 1024);
  const int q = STEM_WGRAD_MFMA ? SW_PX : 64;           // whole stages
  ppb = (ppb + q - 1) / q * q;
  if (ppb < q) ppb = q;
  return ppb;
}
int ew_stem_wgrad_blocks(int B, int H, int W) {
  const int M = B * H * W;
  return ceil_div(M, stem_px_per_block(M));
}
int ew_stem_wgrad(const float* x, const bf16_t* dy, float* dw, float* tmp, int B, int H, int W, hipStream_t st, const bf16_t* x0, const float* coef,
                  const float* sc, const float* sh, const float* alpha) {
  FEDFR_REQUIRE(x && dy && dw && tmp && B > 0 && H > 0 && W > 0, "stem_wgrad: bad args");
  FEDFR_REQUIRE(!x0 || (STEM_WGRAD_MFMA && coef && sc && sh && alpha), "stem_wgrad: the fused BatchNorm + PReLU backward needs coefficients, (scale, shift) and slopes");
  const int M = B * H * W;
  const int ppb = stem_px_per_block(M);
  const int nblk = ceil_div(M, ppb);
  const StemFuse fz{x0, coef, sc, sh, alpha};
  if (x0) hipLaunchKernelGGL(stem_wgrad_mfma_kernel<true>, dim3(nblk), dim3(256), 0, st, x, dy, fz, tmp, B, H, W, ppb);
  else if (STEM_WGRAD_MFMA) hipLaunchKernelGGL(stem_wgrad_mfma_kernel<false>, dim3(nblk), dim3(256), 0, st, x, dy, fz, tmp, B, H, W, ppb);
  else hipLaunchKernelGGL(stem_wgrad_kernel, dim3(nblk), dim3(256), 0, st, x, dy, tmp, B, H, W, ppb);
  FEDFR_LAUNCH_CHECK("stem_wgrad");
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(64), dim3(256), 0, st, tmp, nblk, dw);
  FEDFR_LAUNCH_CHECK("stem_wgrad_reduce");
  return FEDFR_OK;
}

// =====================================================================================================
// input pipeline on the device (reference dataset.py:81-92: ToPILImage -> RandomHorizontalFlip -> ToTensor ->
// Normalize(0.5, 0.5)): uint8 HWC images (as decoded) + one flip flag per image -> fp32 NCHW in [-1, 1], the backbone's input.
// Same fp32 operations in the same order as torchvision (x / 255, then (t - 0.5) / 0.5): bit-exact.  The host uploads 1 byte
// per pixel-channel instead of 4.
// =====================================================================================================
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char* __restrict__ src, const unsigned char* __restrict__ flip,
                                                           float* __restrict__ dst, int B, int H, int W) {
  const long long total = (long long)B * H * W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int w = (int)(i % W);
    const long long t = i / W;
    const int h = (int)(t % H), b = (int)(t / H);
    const int ws = (flip && flip[b]) ? W - 1 - w : w;
    const unsigned char* sp = src + (((long long)b * H + h) * W + ws) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = __fdiv_rn((float)sp[c], 255.f);
      dst[(((long long)b * 3 + c) * H + h) * W + w] = __fdiv_rn(__fsub_rn(v, 0.5f), 0.5f);
    }
  }
}
int ew_preprocess_u8(const unsigned char* src, const unsigned char* flip, float* dst, int B, int H, int W, hipStream_t st) {
  FEDFR_REQUIRE(src && dst && B > 0 && H > 0 && W > 0, "preprocess_u8: bad args");
  const long long total = (long long)B * H * W;
  const int grid = (int)std::min<long long>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(preprocess_u8_kernel, dim3(grid), dim3(256), 0, st, src, flip, dst, B, H, W);
  FEDFR_LAUNCH_CHECK("preprocess_u8");
  return FEDFR_OK;
}

// =====================================================================================================
// bias + PReLU backward without normalisation (sphnet: conv(+bias) -> PReLU, reference backbones/sphnet.py:4-13, :53-60):
// z = x + bias, dz = dy * (z > 0 ? 1 : alpha), dbias = sum dz, dalpha = sum dy * z [z <= 0], dx = dz (+ add).
// Runs on the BN-backward kernels with mean = 0, rstd = 1, gamma = 1, beta = bias and the coefficients (1, 0, 0).
// =====================================================================================================
__global__ __launch_bounds__(256) void prelu_bwd_finalize_kernel(const float* __restrict__ part, int P, int C, float* dbias,
                                                                float* dalpha, float* coef) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s3 = 0.0;
  for (int r = 0; r < P; ++r) {
    s1 += (double)part[(size_t)r * 3 * C + c];
    s3 += (double)part[(size_t)r * 3 * C + 2 * C + c];
  }
  if (dbias) dbias[c] = (float)s1;
  if (dalpha) dalpha[c] = (float)s3;
  coef[c] = 1.f;
  coef[C + c] = 0.f;
  coef[2 * C + c] = 0.f;
}
int ew_bias_prelu_bwd(const bf16_t* dy, const bf16_t* x, const float* bias, const float* alpha, int M, int C, float* partials,
                      float* coef, float* dbias, float* dalpha, const bf16_t* add, bf16_t* dx, hipStream_t st) {
  FEDFR_REQUIRE(alpha && coef, "bias_prelu_bwd: alpha and coef are required");
  BnBwd p{};
  p.dy = dy; p.x = x; p.beta = bias; p.alpha = alpha; p.M = M; p.C = C; p.partials = partials; p.coef = coef; p.add = add; p.dx = dx;
  FEDFR_TRY(ew_bn_bwd_reduce(p, st));
  hipLaunchKernelGGL(prelu_bwd_finalize_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, partials, ew_bn_bwd_grid(M, C), C, dbias, dalpha, coef);
  FEDFR_LAUNCH_CHECK("prelu_bwd_finalize");
  return ew_bn_bwd_apply(p, st);
}

// The same backward as ONE streaming pass (round 3; the sphnet plan, net_sph.inc): unlike a BatchNorm's, a PReLU's parameter sums do not
// feed its input gradient, so dz is written while the sums are gathered and the finalize runs off the critical path.
//   g = dy (+ add);  z = x + bias;  dz = g * (z > 0 ? 1 : alpha);  rows[blk] = (sum dz | sum g z over z <= 0);  optional gsum = g (bf16)
struct PreluBwdP {
  const bf16_t *dy, *add, *x;
  const float *bias, *alpha;
  int M, C;
  bf16_t *gsum, *dz;
  float* rows;
};
template <bool ADD>
__global__ __launch_bounds__(EW_THREADS) void prelu_bwd_pass_kernel(PreluBwdP p, int slab, int shfl) {
  extern __shared__ float red[];
  const int tpr = p.C >> 3, rpp = EW_THREADS / tpr;
  const int cl = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  const bool active = rl < rpp;
  const int c0 = cl * 8;
  float bi[8], al[8];
  load8f(p.bias, c0, bi, 0.f);
  load8f(p.alpha, c0, al, 1.f);
  float acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = 0.f;
  const int bid = ew_block_id();
  const int mbeg = bid * slab, mend = min(p.M, mbeg + slab);
  auto one = [&](int m, const uint4& vd, const uint4& va, const uint4& vx) {
    const size_t off = (size_t)m * p.C + c0;
    float g[8], x[8], o[8];
    unpack8(vd, g);
    unpack8(vx, x);
    if (ADD) {
      float a[8];
      unpack8(va, a);
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] += a[j];
      const uint4 gs = pack8(g);                     // the identity-path gradient the next block adds to is the bf16-rounded sum ...
      *reinterpret_cast<uint4*>(p.gsum + off) = gs;
      unpack8(gs, g);                                // ... and so is what this PReLU sees (as the two-pass form: add pass, then PReLU pass)
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float z = x[j] + bi[j];
      float dz = g[j];
      if (z <= 0.f) {
        acc[1][j] += g[j] * z;
        dz = g[j] * al[j];
      }
      acc[0][j] += dz;
      o[j] = dz;
    }
    const uint4 ov = pack8(o);
    *reinterpret_cast<uint4*>(p.dz + off) = ov;
  };
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  constexpr int UNR = 4;
  int m = active ? mbeg + rl : mend;
  for (; m + (UNR - 1) * rpp < mend; m += UNR * rpp) {
    uint4 vd[UNR], va[UNR], vx[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const size_t off = (size_t)(m + u * rpp) * p.C + c0;
      vd[u] = *reinterpret_cast<const uint4*>(p.dy + off);
      va[u] = ADD ? *reinterpret_cast<const uint4*>(p.add + off) : zero4;
      vx[u] = ew_ld16(p.x + off);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      one(m + u * rpp, vd[u], va[u], vx[u]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  for (; m < mend; m += rpp) {
    const size_t off = (size_t)m * p.C + c0;
    one(m, *reinterpret_cast<const uint4*>(p.dy + off), ADD ? *reinterpret_cast<const uint4*>(p.add + off) : zero4,
        *reinterpret_cast<const uint4*>(p.x + off));
  }
  // NOTE dbias of the two-pass form sums the UNROUNDED dz; so does this (acc[0] adds dz before the bf16 pack)
  ew_block_colsum<2>(acc, p.C, tpr, rpp, cl, rl, active, shfl != 0, red, p.rows + (size_t)bid * 2 * p.C);
}
int ew_prelu_bwd_rows(int M, int C) { return ceil_div(M, slab_rows(M, C, 512)); }
int ew_prelu_bwd_pass(const bf16_t* dy, const bf16_t* add, const bf16_t* x, const float* bias, const float* alpha, int M, int C, bf16_t* gsum,
                      bf16_t* dz, float* rows, hipStream_t st) {
  FEDFR_TRY(check_mc(M, C, "prelu_bwd_pass"));
  FEDFR_REQUIRE(dy && x && alpha && dz && rows && (!add || gsum), "prelu_bwd_pass: null tensor");
  const int slab = slab_rows(M, C, 512);
  const int grid = ceil_div(M, slab);
  const size_t lds = ew_colsum_lds(C, 2);
  const int shfl = ew_shfl_ok(C) ? 1 : 0;
  PreluBwdP p{dy, add, x, bias, alpha, M, C, gsum, dz, rows};
  ProfScope prof(22, (double)M * C * 2 * (add ? 5.0 : 3.0), st);
  if (add) hipLaunchKernelGGL(prelu_bwd_pass_kernel<true>, dim3(grid), dim3(EW_THREADS), lds, st, p, slab, shfl);
  else hipLaunchKernelGGL(prelu_bwd_pass_kernel<false>, dim3(grid), dim3(EW_THREADS), lds, st, p, slab, shfl);
  FEDFR_LAUNCH_CHECK("prelu_bwd_pass");
  return FEDFR_OK;
}
// 8 channels per workgroup, every row load of a trip in flight (fin8_accumulate): the first version walked the rows one thread per channel
// from 1-2 workgroups and took 173 us per PReLU on the sphnet plan's weight-gradient stream (62 launches = 10.7 ms per step)
__global__ __launch_bounds__(256) void prelu_rows_finalize_kernel(const float* __restrict__ rows, int P, int C, float* dbias, float* dalpha) {
  __shared__ double tot[2][8];
  const int c0 = blockIdx.x * 8;
  fin8_accumulate<2>(rows, P, C, c0, 2, tot);
  if (threadIdx.x < 8) {
    const int c = c0 + threadIdx.x;
    if (dbias) dbias[c] = (float)tot[0][threadIdx.x];
    if (dalpha) dalpha[c] = (float)tot[1][threadIdx.x];
  }
}
__global__ __launch_bounds__(256) void prelu_rows_finalize_multi_kernel(const unsigned char* __restrict__ base, float* __restrict__ grads, PreluFinTable t) {
  __shared__ double tot[3][8];
  int i = 0;
  while (i + 1 < t.n && (int)blockIdx.x >= t.e[i + 1].blk0) ++i;        // (workgroup-uniform: a scan of at most 63 scalars)
  const PreluFinEntry e = t.e[i];
  const int c0 = ((int)blockIdx.x - e.blk0) * 8;
  if (e.nv_row == 3) fin8_accumulate<3>(reinterpret_cast<const float*>(base + e.rows_off), e.P, e.C, c0, 2, tot);
  else fin8_accumulate<2>(reinterpret_cast<const float*>(base + e.rows_off), e.P, e.C, c0, 2, tot);
  if (threadIdx.x < 8) {
    const int c = c0 + threadIdx.x;
    if (e.dbias_off >= 0) grads[e.dbias_off + c] = (float)tot[0][threadIdx.x];
    grads[e.dalpha_off + c] = (float)tot[1][threadIdx.x];
  }
}
int ew_prelu_bwd_finalize_multi(const unsigned char* base, float* grads, const PreluFinTable& t, hipStream_t st) {
  FEDFR_REQUIRE(base && grads && t.n > 0 && t.n <= kMaxPreluFin && t.blocks > 0, "prelu_bwd_finalize_multi: bad table");
  hipLaunchKernelGGL(prelu_rows_finalize_multi_kernel, dim3(t.blocks), dim3(256), 0, st, base, grads, t);
  FEDFR_LAUNCH_CHECK("prelu_bwd_finalize_multi");
  return FEDFR_OK;
}
int ew_prelu_bwd_finalize(const float* rows, int P, int C, float* dbias, float* dalpha, hipStream_t st) {
  FEDFR_REQUIRE(rows && P > 0 && C > 0, "prelu_bwd_finalize: bad args");
  FEDFR_REQUIRE((C & 7) == 0, "prelu_bwd_finalize: C %% 8");
  hipLaunchKernelGGL(prelu_rows_finalize_kernel, dim3(C / 8), dim3(256), 0, st, rows, P, C, dbias, dalpha);
  FEDFR_LAUNCH_CHECK("prelu_bwd_finalize");
  return FEDFR_OK;
}

// fp32 NCHW [B][C][HW] -> bf16 NHWC [B][HW][Cpad], channels >= C zero (sphnet's 3-channel input feeds the 64-channel-granular conv)
__global__ __launch_bounds__(256) void pad_input_nhwc_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int B, int C, int HW, int Cpad) {
  const long long total = (long long)B * HW * (Cpad / 8);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ch = (int)(i % (Cpad / 8));
    const long long px = i / (Cpad / 8);
    const int hw = (int)(px % HW), b = (int)(px / HW);
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = ch * 8 + j;
      f[j] = c < C ? src[((long long)b * C + c) * HW + hw] : 0.f;
    }
    *reinterpret_cast<uint4*>(dst + px * Cpad + ch * 8) = pack8(f);
  }
}
int ew_pad_input_nhwc(const float* src, bf16_t* dst, int B, int C, int HW, int Cpad, hipStream_t st) {
  FEDFR_REQUIRE(src && dst && B > 0 && C > 0 && HW > 0 && Cpad >= C && (Cpad & 7) == 0, "pad_input_nhwc: bad args");
  const long long total = (long long)B * HW * (Cpad / 8);
  hipLaunchKernelGGL(pad_input_nhwc_kernel, dim3((int)std::min<long long>((total + 255) / 256, 16384)), dim3(256), 0, st, src, dst, B, C, HW, Cpad);
  FEDFR_LAUNCH_CHECK("pad_input_nhwc");
  return FEDFR_OK;
}


// =====================================================================================================
// Dropout on the flattened bn2 output (reference backbones/iresnet.py:96,169: nn.Dropout(p, inplace=True) between bn2 and fc)
// =====================================================================================================
// Counter-based mask: element i of training step `step` is kept iff a 16-bit hash of (seed, step, i) >= p * 65536, so the mask is a pure
// function of (seed, step, index) — reproducible, no RNG state on the device.  Kept values are scaled by 1 / (1 - p) (torch semantics).
__device__ __forceinline__ unsigned long long dropout_hash(unsigned long long x) {   // splitmix64 finaliser
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void dropout_fwd_kernel(bf16_t* __restrict__ t, unsigned char* __restrict__ mask, size_t n8, unsigned thr,
                                                          float inv_keep, unsigned long long seed, unsigned long long step) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const unsigned long long key = seed * 0x100000001B3ull + step * 0x9E3779B1ull;
    const unsigned long long h0 = dropout_hash(key ^ (2 * i)), h1 = dropout_hash(key ^ (2 * i + 1));
    uint4 v = reinterpret_cast<uint4*>(t)[i];
    float f[8];
    unpack8(v, f);
    unsigned char m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned r = (unsigned)(((j < 4 ? h0 : h1) >> (16 * (j & 3))) & 0xffffu);
      m[j] = r >= thr ? 1 : 0;
      f[j] = m[j] ? f[j] * inv_keep : 0.f;
    }
    reinterpret_cast<uint4*>(t)[i] = pack8(f);
    uint2 mk;
    mk.x = (unsigned)m[0] | ((unsigned)m[1] << 8) | ((unsigned)m[2] << 16) | ((unsigned)m[3] << 24);
    mk.y = (unsigned)m[4] | ((unsigned)m[5] << 8) | ((unsigned)m[6] << 16) | ((unsigned)m[7] << 24);
    reinterpret_cast<uint2*>(mask)[i] = mk;
  }
}
__global__ __launch_bounds__(256) void dropout_bwd_kernel(float* __restrict__ dx, const unsigned char* __restrict__ mask, size_t n4, float inv_keep) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const unsigned mk = reinterpret_cast<const unsigned*>(mask)[i];
    float4 d = reinterpret_cast<float4*>(dx)[i];
    d.x = (mk & 0xffu) ? d.x * inv_keep : 0.f;
    d.y = (mk & 0xff00u) ? d.y * inv_keep : 0.f;
    d.z = (mk & 0xff0000u) ? d.z * inv_keep : 0.f;
    d.w = (mk & 0xff000000u) ? d.w * inv_keep : 0.f;
    reinterpret_cast<float4*>(dx)[i] = d;
  }
}
int ew_dropout_fwd(bf16_t* t, unsigned char* mask, size_t n, float p, unsigned long long seed, unsigned long long step, hipStream_t st) {
  FEDFR_REQUIRE(t && mask && n > 0 && (n & 7) == 0 && p > 0.f && p < 1.f, "dropout_fwd: bad args (n%%8, 0 < p < 1)");
  const size_t n8 = n / 8;
  const int grid = (int)std::min<size_t>((n8 + 255) / 256, 4096);
  hipLaunchKernelGGL(dropout_fwd_kernel, dim3(grid), dim3(256), 0, st, t, mask, n8, (unsigned)(p * 65536.f), 1.f / (1.f - p), seed, step);
  FEDFR_LAUNCH_CHECK("dropout_fwd");
  return FEDFR_OK;
}
int ew_dropout_bwd(float* dx, const unsigned char* mask, size_t n, float p, hipStream_t st) {
  FEDFR_REQUIRE(dx && mask && n > 0 && (n & 3) == 0 && p > 0.f && p < 1.f, "dropout_bwd: bad args (n%%4, 0 < p < 1)");
  const size_t n4 = n / 4;
  const int grid = (int)std::min<size_t>((n4 + 255) / 256, 4096);
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3(grid), dim3(256), 0, st, dx, mask, n4, 1.f / (1.f - p));
  FEDFR_LAUNCH_CHECK("dropout_bwd");
  return FEDFR_OK;
}
