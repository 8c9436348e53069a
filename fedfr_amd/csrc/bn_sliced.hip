// Train-mode BatchNorm passes WITHOUT the finalize launch between the statistics and their consumer (round 2).
//
// A BatchNorm is a grid-wide dependency: statistics over all pixels, then a per-pixel pass.  The row-slab kernels of ew.hip hand the
// statistics over through a tiny kernel (bn_finalize8 / bn_bwd_finalize8: P partial rows -> per-channel coefficients) because a
// workgroup that streams whole 512-B pixel rows would have to reduce P x C partials itself.  In an iresnet100 step that is 306
// launches of 5-6 us in a dependent chain (1.7 ms of 18.5).  Here the consumer does the reduction itself, which is cheap once a
// workgroup owns a CHANNEL SLICE:
//   * workgroup = (32-channel slice s, pixel group g): C / 32 slices x G groups = 256 workgroups; 4 lanes x 16 B cover a pixel's 64-B
//     piece of the slice, a wave-load covers 16 pixels (the same 16 L1 cycles as one coalesced 1-KiB row load);
//   * logical id = g * NS + s through xcd_remap: the slices of a pixel group run on one XCD, so the 128-B lines two slices share are
//     fetched into that L2 once;
//   * prologue: the workgroup sums the P partial rows of ITS 32 channels in fp64 (P x NV x 128 B, <= 64 KB, 16-B loads, all issued
//     before the first add) while its first tensor loads are already in flight, derives the coefficients, and group 0 of every slice
//     writes what later passes need (saved scale / shift / mean / rstd + running statistics; dgamma / dbeta / dalpha);
//   * partial rows it produces (statistics of its output for the next BatchNorm; the next BatchNorm-backward's sums) are G rows of the
//     same [row][statistic][C] layout the ew.hip kernels and the conv epilogues write, so fused and unfused passes mix freely.
// Rows are read and written by different workgroups of one launch with no ordering between them: the output rows must not alias the
// input rows (net.hip alternates between two partial-row buffers).
#include "ew.h"
#include "gemm_dev.h"   // ProfScope

#ifndef BNS_PRIO
#define BNS_PRIO 3   // wave priority of the backward passes: they share CUs with the weight-gradient kernels of the aux stream
#endif
namespace {
constexpr int SW = 32;          // channels per slice
constexpr int NTH = 256;
constexpr int PXP = NTH / 4;    // pixels per pass

__device__ __forceinline__ uint4 ld16_nt(const bf16_t* p) {
  typedef __attribute__((ext_vector_type(4))) unsigned u4v;
  const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint4 ld16(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }

// Reduction of the P partial rows of this workgroup's 32 channels, in two halves so that the tensor loads can sit between them: the
// loads of the rows are issued FIRST (vmcnt retires loads in order: rows issued behind the tensor would only be usable once the whole
// tensor share has arrived), FL per thread with the row index clamped (branch-free: predicated loads make hipcc drain vmcnt), then the
// caller issues its tensor loads, then fan_in_finish sums — the compiler's counted vmcnt leaves the tensor loads in flight.
//   tot[v * 32 + c] = sum over rows of statistic v, channel cs + c  (fp64; rows are [P][nv_row][C], NV <= nv_row of them read)
template <int NV> struct FanIn {
  static constexpr int COLS4 = NV * 8, RG = NTH / COLS4, COLS = NV * 32;
  static constexpr int red_doubles = RG * (COLS + 1);
};
template <int NV, int FL>
__device__ __forceinline__ void fan_in_issue(const float* __restrict__ part, int P, int C, int nv_row, int cs, float4* v) {
  constexpr int COLS4 = FanIn<NV>::COLS4, RG = FanIn<NV>::RG;
  const int tid = threadIdx.x;
  const int c4 = tid % COLS4, rg = min(tid / COLS4, RG - 1);
  const float* src = part + (size_t)(c4 >> 3) * C + cs + (c4 & 7) * 4;
  const size_t rs = (size_t)nv_row * C;
#pragma unroll
  for (int i = 0; i < FL; ++i) v[i] = *reinterpret_cast<const float4*>(src + (size_t)min(rg + i * RG, P - 1) * rs);
  __builtin_amdgcn_sched_barrier(0);                    // the scheduler would move these behind the tensor loads (their use comes first)
}
template <int NV, int FL>
__device__ __forceinline__ void fan_in_finish(const float4* v, int P, double* tot, double* red) {
  constexpr int COLS4 = FanIn<NV>::COLS4, RG = FanIn<NV>::RG, COLS = FanIn<NV>::COLS;
  const int tid = threadIdx.x;
  const int c4 = tid % COLS4, rg = tid / COLS4;
  if (rg < RG) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int i = 0; i < FL; ++i) {
      const double w = rg + i * RG < P ? 1.0 : 0.0;     // clamped repeats of the last row count for nothing
      a0 += w * (double)v[i].x; a1 += w * (double)v[i].y; a2 += w * (double)v[i].z; a3 += w * (double)v[i].w;
    }
    double* d = red + rg * (COLS + 1) + c4 * 4;
    d[0] = a0; d[1] = a1; d[2] = a2; d[3] = a3;
  }
  __syncthreads();
  if (tid < COLS) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < RG; ++i) t += red[i * (COLS + 1) + tid];
    tot[tid] = t;
  }
  __syncthreads();
}
constexpr int kFwdFL = 16;                  // forward: up to 16 x 16 = 256 rows (a 14x14 conv's epilogue leaves 256)
constexpr int kBwdFL2 = 8, kBwdFL3 = 13;    // backward: up to 128 rows (two statistics) / 130 (three, PReLU)

// v[NV][8] (this thread's 8 channels) summed over the workgroup's 64 pixel lanes -> row[v * C + cs + c]
template <int NV>
__device__ __forceinline__ void slice_rowsum(float (*v)[8], int C, int cs, float* sred /*[4][NV*32]*/, float* row) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cl = tid & 3;
#pragma unroll
  for (int o = 4; o < 64; o <<= 1)
#pragma unroll
    for (int s = 0; s < NV; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[s][j] += __shfl_xor(v[s][j], o, 64);
  if (lane < 4) {
#pragma unroll
    for (int s = 0; s < NV; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) sred[wave * (NV * 32) + s * 32 + cl * 8 + j] = v[s][j];
  }
  __syncthreads();
  if (tid < NV * 32) {
    const float t = (sred[tid] + sred[NV * 32 + tid]) + (sred[2 * NV * 32 + tid] + sred[3 * NV * 32 + tid]);
    row[(size_t)(tid >> 5) * C + cs + (tid & 31)] = t;
  }
}

__device__ __forceinline__ void lds8(const float* p, float* v) {
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = p[j];
}
__device__ __forceinline__ void glb8(const float* p, int c0, float* v, float dflt) {
  if (p) {
    const float4 a = *reinterpret_cast<const float4*>(p + c0), b = *reinterpret_cast<const float4*>(p + c0 + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = dflt;
  }
}
// branch-free forms for optional per-channel vectors (a null pointer reads `safe`, any readable fp32 array of >= C elements, and the
// value is replaced): a branch per optional load would put one memory round trip after another at the top of the kernel
__device__ __forceinline__ float opt1(const float* p, const float* safe, int c, float dflt) {
  const float v = (p ? p : safe)[c];
  return p ? v : dflt;
}
__device__ __forceinline__ void ld8(const float* p, float* v) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void opt8(const float* p, const float* safe, int c0, float* v, float dflt) {
  ld8((p ? p : safe) + c0, v);
  const float m = p ? 1.f : 0.f, d = p ? 0.f : dflt;     // arithmetic select: a branch here would wait for the load on the spot
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = v[j] * m + d;
}

// =====================================================================================================
// forward: y = prelu?(bn(x1)) (+ x2), statistics of x1 from partial rows; optional statistics of y
// =====================================================================================================
// NP = passes of 64 pixels a workgroup makes: ALL its tensor loads are issued before the prologue, so the reduction of the partial rows
// (a ~2 us chain of dependent latencies) overlaps the whole fetch instead of delaying it
template <int NP, bool X2>
__global__ __launch_bounds__(NTH) void bn_apply_s_kernel(BnApplyS p) {
  __shared__ double red[FanIn<2>::red_doubles];
  __shared__ double tot[64];
  __shared__ float cf[2][SW];
  __shared__ float sred[4 * 64];
  const int tid = threadIdx.x;
  const int NS = p.C >> 5;
  const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int g = lid / NS, s = lid - g * NS;
  const int cl = tid & 3, pl = tid >> 2;
  const int cs = s * SW, c0 = cs + cl * 8;
  const int m0 = g * p.ppg, m1 = min(p.M, m0 + p.ppg);
  // every small per-channel load first: a load issued behind the tensor loads could only be waited for with the tensor (in-order vmcnt)
  const int cc = cs + (tid & 31);
  const float ga = opt1(p.gamma, p.part, cc, 1.f), be = opt1(p.beta, p.part, cc, 0.f);
  const float rm0 = opt1(p.rm, p.part, cc, 0.f), rv0 = opt1(p.rv, p.part, cc, 0.f);
  const bool has_alpha = p.alpha != nullptr;
  float al[8];
  opt8(p.alpha, p.part, c0, al, 1.f);
  float4 fv[kFwdFL];
  fan_in_issue<2, kFwdFL>(p.part, p.P, p.C, 2, cs, fv);
  const double ic = 1.0 / p.count;                      // (ew.h: bn_rsqrt; formed here, under the loads)
  asm volatile("" ::"v"(ic));
  uint4 a1[NP], a2[X2 ? NP : 1];
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    const int m = min(m0 + u * PXP + pl, m1 - 1);
    const size_t off = (size_t)m * p.C + c0;
    a1[u] = ld16_nt(p.x1 + off);
    if (X2) a2[u] = ld16(p.x2 + off);
  }

  fan_in_finish<2, kFwdFL>(fv, p.P, tot, red);
  if (tid < SW) {
    const int c = cs + tid;
    const double mean = tot[tid] * ic;
    double var = tot[SW + tid] * ic - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = bn_rsqrt(var + (double)p.eps);
    const float sc = (float)((double)ga * rstd), sh = (float)((double)be - mean * (double)ga * rstd);
    cf[0][tid] = sc;
    cf[1][tid] = sh;
    if (g == 0) {                                       // one workgroup per slice leaves what the backward pass / the caller reads
      p.scale[c] = sc; p.shift[c] = sh; p.mean[c] = (float)mean; p.rstd[c] = (float)rstd;
      if (p.rm) {
        const double unb = p.count > 1.0 ? var * p.count / (p.count - 1.0) : var;
        p.rm[c] = (float)((1.0 - p.momentum) * (double)rm0 + p.momentum * mean);
        p.rv[c] = (float)((1.0 - p.momentum) * (double)rv0 + p.momentum * unb);
      }
    }
  }
  __syncthreads();
  float sc[8], sh[8];
  lds8(&cf[0][cl * 8], sc);
  lds8(&cf[1][cl * 8], sh);
  float st[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) st[0][j] = st[1][j] = 0.f;
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    const int m = m0 + u * PXP + pl;
    if (m < m1) {
      float f[8];
      unpack8(a1[u], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = f[j] * sc[j] + sh[j];
        if (has_alpha) v = v > 0.f ? v : al[j] * v;
        f[j] = v;
      }
      if (X2) {
        float h[8];
        unpack8(a2[u], h);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] += h[j];
      }
      const uint4 o = pack8(f);
      if (p.stats) {
        float r[8];
        unpack8(o, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          st[0][j] += r[j];
          st[1][j] += r[j] * r[j];
        }
      }
      *reinterpret_cast<uint4*>(p.y + (size_t)m * p.C + c0) = o;
    }
  }
  if (p.stats) slice_rowsum<2>(st, p.C, cs, sred, p.stats + (size_t)g * 2 * p.C);
}

// =====================================================================================================
// forward, round 3: out = bn(x1) + x2 and y2 = bn_next(out) in ONE pass (BnApply2S, ew.h): the statistics of `out` follow from the raw
// moments (sum x1, sum x1 x2, sum x1 x1) the conv left and the saved statistics of x2 — the next block's bn1 pass disappears
// =====================================================================================================
template <int NP, int FL>
__global__ __launch_bounds__(NTH) void bn_apply2_s_kernel(BnApply2S p) {
  __shared__ double red[FanIn<3>::red_doubles];
  __shared__ double tot[96];
  __shared__ float cf[4][SW];
  const int tid = threadIdx.x;
  const int NS = p.C >> 5;
  const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int g = lid / NS, s = lid - g * NS;
  const int cl = tid & 3, pl = tid >> 2;
  const int cs = s * SW, c0 = cs + cl * 8;
  const int m0 = g * p.ppg, m1 = min(p.M, m0 + p.ppg);
  // every small per-channel load first (in-order vmcnt: see bn_apply_s_kernel)
  const int cc = cs + (tid & 31);
  const float ga = p.gamma[cc], be = p.beta[cc], rm0 = p.rm[cc], rv0 = p.rv[cc];
  const float nga = p.ngamma[cc], nbe = p.nbeta[cc], nrm0 = p.nrm[cc], nrv0 = p.nrv[cc];
  const float xm = p.xmean[cc], xr = p.xrstd[cc];
  float4 fv[FL];
  fan_in_issue<3, FL>(p.part, p.P, p.C, 3, cs, fv);
  uint4 a1[NP], a2[NP];
  const double ic = 1.0 / p.count;                      // (ew.h: bn_rsqrt) the count's reciprocal and the identity path's variance: formed under the loads
  double vx = 1.0 / ((double)xr * (double)xr) - (double)p.eps;
  if (vx < 0.0) vx = 0.0;
  asm volatile("" ::"v"(ic), "v"(vx));
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    const int m = min(m0 + u * PXP + pl, m1 - 1);
    const size_t off = (size_t)m * p.C + c0;
    a1[u] = ld16_nt(p.x1 + off);
    a2[u] = ld16(p.x2 + off);
  }
  fan_in_finish<3, FL>(fv, p.P, tot, red);
  if (tid < SW) {
    const int c = cs + tid;
    const double mean = tot[tid] * ic;
    double var = tot[2 * SW + tid] * ic - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = bn_rsqrt(var + (double)p.eps);
    const float sc = (float)((double)ga * rstd), sh = (float)((double)be - mean * (double)ga * rstd);
    // statistics of out = sc x1 + sh + x2 (with the fp32 coefficients the elements are computed with)
    const double mx = (double)xm;
    const double cov = tot[SW + tid] * ic - mean * mx;
    const double omean = (double)sc * mean + (double)sh + mx;
    double ovar = (double)sc * (double)sc * var + vx + 2.0 * (double)sc * cov;
    if (ovar < 0.0) ovar = 0.0;
    const double orstd = bn_rsqrt(ovar + (double)p.eps);
    const float nsc = (float)((double)nga * orstd), nsh = (float)((double)nbe - omean * (double)nga * orstd);
    cf[0][tid] = sc; cf[1][tid] = sh; cf[2][tid] = nsc; cf[3][tid] = nsh;
    if (g == 0) {
      p.scale[c] = sc; p.shift[c] = sh; p.mean[c] = (float)mean; p.rstd[c] = (float)rstd;
      p.nscale[c] = nsc; p.nshift[c] = nsh; p.nmean[c] = (float)omean; p.nrstd[c] = (float)orstd;
      const double k = p.count > 1.0 ? p.count / (p.count - 1.0) : 1.0;
      p.rm[c] = (float)((1.0 - p.momentum) * (double)rm0 + p.momentum * mean);
      p.rv[c] = (float)((1.0 - p.momentum) * (double)rv0 + p.momentum * var * k);
      p.nrm[c] = (float)((1.0 - p.momentum) * (double)nrm0 + p.momentum * omean);
      p.nrv[c] = (float)((1.0 - p.momentum) * (double)nrv0 + p.momentum * ovar * k);
    }
  }
  __syncthreads();
  float sc[8], sh[8], nsc[8], nsh[8];
  lds8(&cf[0][cl * 8], sc);
  lds8(&cf[1][cl * 8], sh);
  lds8(&cf[2][cl * 8], nsc);
  lds8(&cf[3][cl * 8], nsh);
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    const int m = m0 + u * PXP + pl;
    if (m < m1) {
      float f[8], h[8];
      unpack8(a1[u], f);
      unpack8(a2[u], h);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f[j] = f[j] * sc[j] + sh[j];
        f[j] += h[j];
      }
      const uint4 o = pack8(f);
      *reinterpret_cast<uint4*>(p.y + (size_t)m * p.C + c0) = o;
      float r[8];
      unpack8(o, r);                                      // the next BatchNorm sees the stored (bf16) sum, as its own pass would
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = r[j] * nsc[j] + nsh[j];
      *reinterpret_cast<uint4*>(p.y2 + (size_t)m * p.C + c0) = pack8(r);
    }
  }
}

// =====================================================================================================
// backward, pass 1: partial sums (sum dz | sum dz xhat | sum dy z over z <= 0) -> G rows [3][C]
// =====================================================================================================
template <bool ALPHA, int U>
__global__ __launch_bounds__(NTH) void bn_bwd_reduce_s_kernel(BnBwdS p) {
  __builtin_amdgcn_s_setprio(BNS_PRIO);
  __shared__ float sred[4 * 96];
  const int tid = threadIdx.x;
  const int NS = p.C >> 5;
  const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int g = lid / NS, s = lid - g * NS;
  const int cl = tid & 3, pl = tid >> 2;
  const int cs = s * SW, c0 = cs + cl * 8;
  const int m0 = g * p.ppg, m1 = min(p.M, m0 + p.ppg);
  float mean[8], G[8], H[8], al[8];
  glb8(p.mean, c0, mean, 0.f);
  if (ALPHA) {
    glb8(p.sc, c0, G, 1.f);
    glb8(p.sh, c0, H, 0.f);
    glb8(p.alpha, c0, al, 1.f);
  }
  float acc[3][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = acc[2][j] = 0.f;
  auto issue = [&](int base, uint4* vd, uint4* vx) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = min(base + u * PXP + pl, m1 - 1);
      const size_t off = (size_t)m * p.C + c0;
      vd[u] = ld16(p.dy + off);
      vx[u] = ld16(p.x + off);
    }
  };
  auto one = [&](int m, const uint4& vd, const uint4& vx) {
    if (m >= m1) return;
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
  uint4 ad[U], ax[U];
  issue(m0, ad, ax);
  int base = m0;
  while (true) {
    const int nb = base + U * PXP;
    const bool more = nb < m1;
    uint4 bd[U], bx[U];
    if (more) issue(nb, bd, bx);
#pragma unroll
    for (int u = 0; u < U; ++u) one(base + u * PXP + pl, ad[u], ax[u]);
    if (!more) break;
#pragma unroll
    for (int u = 0; u < U; ++u) { ad[u] = bd[u]; ax[u] = bx[u]; }
    base = nb;
  }
  {
    float rstd[8];
    glb8(p.rstd, c0, rstd, 1.f);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[1][j] *= rstd[j];
  }
  slice_rowsum<3>(acc, p.C, cs, sred, p.partials + (size_t)g * 3 * p.C);
}

// =====================================================================================================
// backward, pass 2: coefficients from the partial rows, dx = a dz + A x + B (+ addend) (+ the next BatchNorm-backward's sums of dx)
// =====================================================================================================
// NPRE passes of every tensor are fetched before the prologue (their latency hides it); what is left of the workgroup's pixels follows in
// double-buffered chunks of U passes.  NPRE >= the passes a workgroup makes = everything in flight at once: fastest alone (7.4 vs 9.7 us
// on a 14x14x256 tensor) but 170-230 VGPRs — beside wgrad9's two waves per SIMD (2 x 144 registers) such a wave only fits an unfragmented
// register file, and in the dual-stream backward pass it waited (40 us per launch for the 230-register variant); the leaner profiles
// trade a microsecond alone for co-residency.
// FX = 2: twice the fan-in (256 / 260 partial rows: what the two-tiles 28x28 dgrad with the reduction in its epilogue leaves, round 3)
template <bool ALPHA, bool NX, bool ADD, int NPRE, int U, int FX = 1>
__global__ __launch_bounds__(NTH) void bn_bwd_apply_s_kernel(BnBwdS p) {
  __builtin_amdgcn_s_setprio(BNS_PRIO);
  constexpr int NV = ALPHA ? 3 : 2, FL = FX * (ALPHA ? kBwdFL3 : kBwdFL2);
  __shared__ double red[FanIn<NV>::red_doubles];
  __shared__ double tot[NV * 32];
  __shared__ float cf[3][SW];
  __shared__ float sred[NX ? 4 * 64 : 1];
  const int tid = threadIdx.x;
  const int NS = p.C >> 5;
  const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int g = lid / NS, s = lid - g * NS;
  const int cl = tid & 3, pl = tid >> 2;
  const int cs = s * SW, c0 = cs + cl * 8;
  const int m0 = g * p.ppg, m1 = min(p.M, m0 + p.ppg);
  // every small per-channel load first (see bn_apply_s_kernel)
  const int cc = cs + (tid & 31);
  const float ga_ = opt1(p.gamma, p.part_in, cc, 1.f), r_ = opt1(p.rstd, p.part_in, cc, 1.f), mu_ = opt1(p.mean, p.part_in, cc, 0.f);
  float G[8], H[8], al[8], nmean[8], nrstd[8];
  if (ALPHA) {                                          // non-null: checked by the launcher
    ld8(p.sc + c0, G);
    ld8(p.sh + c0, H);
    ld8(p.alpha + c0, al);
  }
  if (NX) {
    ld8(p.nmean + c0, nmean);
    ld8(p.nrstd + c0, nrstd);
  }
  float4 fv[FL];
  fan_in_issue<NV, FL>(p.part_in, p.P, p.C, 3, cs, fv);
  struct Px { uint4 d, x, a, n; };
  auto fetch = [&](int m_raw) {
    Px r;
    const int m = min(m_raw, m1 - 1);
    const size_t off = (size_t)m * p.C + c0;
    r.d = ld16_nt(p.dy + off);
    r.x = ld16_nt(p.x + off);
    if (ADD) r.a = ld16(p.add + off);
    if (NX) r.n = ld16(p.nx + off);
    return r;
  };
  Px pre[NPRE];
#pragma unroll
  for (int u = 0; u < NPRE; ++u) pre[u] = fetch(m0 + u * PXP + pl);

  const double ic = 1.0 / p.count;                      // (ew.h: bn_rsqrt; formed under the loads)
  asm volatile("" ::"v"(ic));
  fan_in_finish<NV, FL>(fv, p.P, tot, red);
  if (tid < SW) {
    const int c = cs + tid;
    const double t1 = tot[tid], t2 = tot[SW + tid];
    if (g == 0) {
      if (p.dgamma) p.dgamma[c] = (float)t2;
      if (p.dbeta) p.dbeta[c] = (float)t1;
      if (ALPHA && p.dalpha) p.dalpha[c] = (float)tot[(NV - 1) * SW + tid];
    }
    const double ga = (double)ga_, r = (double)r_, mu = (double)mu_;
    const double a = (double)(float)(ga * r), cb = t1 * ic, cq = t2 * ic;
    cf[0][tid] = (float)a;
    cf[1][tid] = (float)(-a * cq * r);
    cf[2][tid] = (float)(a * (cq * r * mu - cb));
  }
  __syncthreads();
  float ca[8], cA[8], cB[8];
  lds8(&cf[0][cl * 8], ca);
  lds8(&cf[1][cl * 8], cA);
  lds8(&cf[2][cl * 8], cB);
  float nacc[2][8];
  if (NX) {
#pragma unroll
    for (int j = 0; j < 8; ++j) nacc[0][j] = nacc[1][j] = 0.f;
  }
  auto one = [&](int m, const Px& v) {
    if (m >= m1) return;
    float dy[8], x[8], o[8];
    unpack8(v.d, dy);
    unpack8(v.x, x);
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
      unpack8(v.a, a);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += a[j];
    }
    const uint4 ov = pack8(o);
    *reinterpret_cast<uint4*>(p.dx + (size_t)m * p.C + c0) = ov;
    if (NX) {                                           // the next BN sees the bf16-rounded dx, exactly as its own reduce pass would
      float dn[8], xn[8];
      unpack8(ov, dn);
      unpack8(v.n, xn);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        nacc[0][j] += dn[j];
        nacc[1][j] += dn[j] * (xn[j] - nmean[j]);
      }
    }
  };
  int base = m0 + NPRE * PXP;
  const bool rest = base < m1;
  Px cur[U];
  if (rest) {
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = fetch(base + u * PXP + pl);
  }
#pragma unroll
  for (int u = 0; u < NPRE; ++u) {
    one(m0 + u * PXP + pl, pre[u]);
    if (NPRE > 4) __builtin_amdgcn_sched_barrier(0);    // one pixel's temporaries at a time
  }
  if (rest) {
    while (true) {
      const int nb = base + U * PXP;
      const bool more = nb < m1;
      Px nxt[U];
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u) nxt[u] = fetch(nb + u * PXP + pl);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        one(base + u * PXP + pl, cur[u]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!more) break;
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = nxt[u];
      base = nb;
    }
  }
  if (NX) {
#pragma unroll
    for (int j = 0; j < 8; ++j) nacc[1][j] *= nrstd[j];
    float* row = p.npart + (size_t)g * 3 * p.C;
    slice_rowsum<2>(nacc, p.C, cs, sred, row);
    if (tid < SW) row[2 * (size_t)p.C + cs + tid] = 0.f;
  }
}

}  // namespace

int g_bn_sliced = 1;   // option "bn_sliced": channel-sliced BatchNorm passes that reduce their partial rows themselves (no finalize launches)
// (round 6: the alternative prefetch profiles of the backward apply pass — everything in flight / six passes, option "bn_sliced_pre" 1 / 2 — lost every
// sweep of rounds 3-5 (+0.6 / +0.8 ms: such a wave does not fit beside a weight-gradient wave) and were removed with their 32 instantiations)

// geometry: C / 32 slices x G pixel groups, ~256 workgroups, at most 13 passes of 64 pixels per workgroup (more groups on larger maps)
constexpr int kMaxPasses = 13;
static const int g_bn_sliced_bwd_passes = 26;   // most passes a workgroup of the BACKWARD kernels makes (they stream in chunks, any count works): 26 keeps a
                                   // 28x28 map at 256 workgroups instead of 484 that are dispatched in two rounds between the weight-gradient workgroups (17.20 -> 17.11 ms/step)
static void sliced_geometry(int M, int C, int* G, int* ppg, bool backward = false) {
  const int NS = C / SW;
  int g = 256 / NS;
  if (g < 1) g = 1;
  int per = (M + g - 1) / g;
  per = (per + 7) / 8 * 8;
  const int maxp = backward ? g_bn_sliced_bwd_passes : kMaxPasses;
  if (per > maxp * PXP) per = maxp * PXP;
  *ppg = per;
  *G = (M + per - 1) / per;
}
int ew_bn_sliced_rows(int M, int C, bool backward) {
  int G, ppg;
  sliced_geometry(M, C, &G, &ppg, backward);
  return G;
}
// a BatchNorm over [M][C] whose statistics arrive as P_in partial rows can run sliced: the tensor is small enough that a few hundred
// workgroups hold it in flight (larger maps keep the row-slab kernels and their finalize: the launch is noise beside 50+ us of
// streaming), and the per-workgroup reduction of the rows stays a fraction of the tensor share it streams
bool ew_bn_sliced_ok(int M, int C, int P_in, bool backward) {
  if (!g_bn_sliced || C < 64 || C > 1024 || (C % SW) != 0 || M < 256) return false;
  if ((long long)M * C > 14ll * 1000 * 1000 || M > 128 * kMaxPasses * PXP) return false;
  return P_in > 0 && P_in <= (backward ? 2 * kBwdFL2 * FanIn<2>::RG : kFwdFL * FanIn<2>::RG);     // backward: the apply pass has a double fan-in variant
}

bool ew_bn_apply2_sliced_ok(int M, int C, int P) {
  return ew_bn_sliced_ok(M, C, 1, false) && P > 0 && P <= 2 * kBwdFL3 * FanIn<3>::RG;
}
int ew_bn_apply2_sliced(BnApply2S p, hipStream_t st) {
  FEDFR_REQUIRE(p.part && p.x1 && p.x2 && p.y && p.y2 && p.gamma && p.beta && p.rm && p.rv && p.scale && p.shift && p.mean && p.rstd && p.xmean &&
                    p.xrstd && p.ngamma && p.nbeta && p.nrm && p.nrv && p.nscale && p.nshift && p.nmean && p.nrstd, "bn_apply2_sliced: null argument");
  FEDFR_REQUIRE(ew_bn_apply2_sliced_ok(p.M, p.C, p.P), "bn_apply2_sliced: unsupported shape M=%d C=%d P=%d", p.M, p.C, p.P);
  sliced_geometry(p.M, p.C, &p.G, &p.ppg);
  ProfScope prof(20, (double)p.M * p.C * 2 * 4, st);
  const dim3 grid((p.C / SW) * p.G);
  const bool small = p.ppg <= 7 * PXP, wide = p.P > kBwdFL3 * FanIn<3>::RG;
  if (small) {
    if (wide) hipLaunchKernelGGL((bn_apply2_s_kernel<7, 2 * kBwdFL3>), grid, dim3(NTH), 0, st, p);
    else hipLaunchKernelGGL((bn_apply2_s_kernel<7, kBwdFL3>), grid, dim3(NTH), 0, st, p);
  } else {
    if (wide) hipLaunchKernelGGL((bn_apply2_s_kernel<kMaxPasses, 2 * kBwdFL3>), grid, dim3(NTH), 0, st, p);
    else hipLaunchKernelGGL((bn_apply2_s_kernel<kMaxPasses, kBwdFL3>), grid, dim3(NTH), 0, st, p);
  }
  FEDFR_LAUNCH_CHECK("bn_apply2_sliced");
  return FEDFR_OK;
}

int ew_bn_apply_sliced(BnApplyS p, hipStream_t st) {
  FEDFR_REQUIRE(p.x1 && p.y && p.part && p.scale && p.shift && p.mean && p.rstd && p.P > 0 && p.M > 0 && p.C > 0 && (p.C % SW) == 0,
                "bn_apply_sliced: bad args");
  sliced_geometry(p.M, p.C, &p.G, &p.ppg);
  FEDFR_REQUIRE(!p.stats || p.stats + (size_t)p.G * 2 * p.C <= p.part || p.part + (size_t)p.P * 2 * p.C <= p.stats,
                "bn_apply_sliced: output rows alias the input rows");
  ProfScope prof(20, (double)p.M * p.C * 2 * (p.x2 ? 3 : 2), st);
  const dim3 grid((p.C / SW) * p.G);
  const bool small = p.ppg <= 7 * PXP;
  if (p.x2) {
    if (small) hipLaunchKernelGGL((bn_apply_s_kernel<7, true>), grid, dim3(NTH), 0, st, p);
    else hipLaunchKernelGGL((bn_apply_s_kernel<kMaxPasses, true>), grid, dim3(NTH), 0, st, p);
  } else {
    if (small) hipLaunchKernelGGL((bn_apply_s_kernel<7, false>), grid, dim3(NTH), 0, st, p);
    else hipLaunchKernelGGL((bn_apply_s_kernel<kMaxPasses, false>), grid, dim3(NTH), 0, st, p);
  }
  FEDFR_LAUNCH_CHECK("bn_apply_sliced");
  return FEDFR_OK;
}

int ew_bn_bwd_reduce_sliced(BnBwdS p, hipStream_t st) {
  FEDFR_REQUIRE(p.dy && p.x && p.partials && p.M > 0 && p.C > 0 && (p.C % SW) == 0, "bn_bwd_reduce_sliced: bad args");
  FEDFR_REQUIRE(!p.alpha || (p.sc && p.sh), "bn_bwd_reduce_sliced: the PReLU mask needs the forward's (scale, shift)");
  sliced_geometry(p.M, p.C, &p.G, &p.ppg, true);
  ProfScope prof(21, (double)p.M * p.C * 2 * 2, st);
  const dim3 grid((p.C / SW) * p.G);
  if (p.alpha) hipLaunchKernelGGL((bn_bwd_reduce_s_kernel<true, 2>), grid, dim3(NTH), 0, st, p);
  else hipLaunchKernelGGL((bn_bwd_reduce_s_kernel<false, 4>), grid, dim3(NTH), 0, st, p);
  FEDFR_LAUNCH_CHECK("bn_bwd_reduce_sliced");
  return FEDFR_OK;
}

int ew_bn_bwd_apply_sliced(BnBwdS p, hipStream_t st) {
  FEDFR_REQUIRE(p.dy && p.x && p.dx && p.part_in && p.P > 0 && p.M > 0 && p.C > 0 && (p.C % SW) == 0, "bn_bwd_apply_sliced: bad args");
  FEDFR_REQUIRE(!p.alpha || (p.sc && p.sh), "bn_bwd_apply_sliced: the PReLU mask needs the forward's (scale, shift)");
  if (p.nx) FEDFR_REQUIRE(p.nmean && p.nrstd && p.npart, "bn_bwd_apply_sliced: next-BN reduction needs mean / rstd / partials");
  sliced_geometry(p.M, p.C, &p.G, &p.ppg, true);
  FEDFR_REQUIRE(!p.nx || p.npart + (size_t)p.G * 3 * p.C <= p.part_in || p.part_in + (size_t)p.P * 3 * p.C <= p.npart,
                "bn_bwd_apply_sliced: output rows alias the input rows");
  ProfScope prof(22, (double)p.M * p.C * 2 * (3.0 + (p.add ? 1.0 : 0.0) + (p.nx ? 1.0 : 0.0)), st);
  const dim3 grid((p.C / SW) * p.G);
  const int variant = (p.alpha ? 4 : 0) | (p.nx ? 2 : 0) | (p.add ? 1 : 0);
  // ONE prefetch profile (three passes in flight, chunks of two): see the kernel comment (registers beside the weight-gradient waves)
  const bool wide = p.P > (p.alpha ? kBwdFL3 * FanIn<3>::RG : kBwdFL2 * FanIn<2>::RG);
  FEDFR_REQUIRE(p.P <= 2 * (p.alpha ? kBwdFL3 * FanIn<3>::RG : kBwdFL2 * FanIn<2>::RG), "bn_bwd_apply_sliced: %d partial rows", p.P);
#define BWD_S(A, N, D)                                                                                        \
  do {                                                                                                        \
    if (wide) hipLaunchKernelGGL((bn_bwd_apply_s_kernel<A, N, D, 3, 2, 2>), grid, dim3(NTH), 0, st, p);            \
    else hipLaunchKernelGGL((bn_bwd_apply_s_kernel<A, N, D, 3, 2>), grid, dim3(NTH), 0, st, p);                    \
  } while (0)
  switch (variant) {
    case 0: BWD_S(false, false, false); break;
    case 1: BWD_S(false, false, true); break;
    case 2: BWD_S(false, true, false); break;
    case 3: BWD_S(false, true, true); break;
    case 4: BWD_S(true, false, false); break;
    case 5: BWD_S(true, false, true); break;
    case 6: BWD_S(true, true, false); break;
    default: BWD_S(true, true, true); break;
  }
#undef BWD_S
  FEDFR_LAUNCH_CHECK("bn_bwd_apply_sliced");
  return FEDFR_OK;
}
