// HBM-bound flat-buffer kernels: fused momentum-SGD (+bf16 weight shadow), FedAvg scale/accumulate,
// PartialFC sampling (counter-based RNG, radix top-k select, ordered compaction), row gather/scatter.
// Reference semantics: torch.optim.SGD as used in client.py:335,527-529; server.py:25-46; partial_fc.py:89-116.
#include "optim.h"
#include "gemm_dev.h"   // ProfScope

// ---------------------------------------------------------------------------------------------------------
// SGD: g += wd*p ; buf = first ? g : mu*buf + g ; p -= lr*buf ; optional bf16 shadow of the new p.
// Op order / fusion mirrors torch's vectorised CPU kernels (alpha-adds are fmadd, mul_ then add_ are two roundings).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sgd_one(float p, float g, float& buf, float lr, float mu, float wd, int first) {
  const float d = fmaf(wd, p, g);                                  // grad.add(param, alpha=wd)   (fmadd in torch)
  const float b = first ? d : __fadd_rn(__fmul_rn(buf, mu), d);    // buf.mul_(mu).add_(d)        (two roundings)
  buf = b;
  return fmaf(-lr, b, p);                                          // param.add_(buf, alpha=-lr)  (fmadd in torch)
}

// UNSCALE: the gradient buffer holds gs^-1 x the gradient (static loss scale of the fp16-storage build); the kernel multiplies by gs
// (a power of two: exact) and stores the true gradient back, so the buffer reads like p.grad afterwards.  Overflow guard (what
// torch.cuda.amp.GradScaler.step does for the reference's fp16 path, client.py:394-396, without its host synchronisation): an element whose
// gradient is not finite is NOT updated (parameter, momentum and mirror keep their values; on the first step its momentum starts at 0) and
// *ovf is set, so one overflowing pass cannot poison the weights; the host reads the word when it next synchronises and lowers the scale.
template <bool UNSCALE>
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ buf,
                                                  bf16_t* __restrict__ shadow, size_t n, float lr, float mu, float wd, int first, float gs,
                                                  unsigned* __restrict__ ovf) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t n4 = n / 4;
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    float4 gv = reinterpret_cast<const float4*>(g)[i];
    if (UNSCALE) {
      gv.x *= gs; gv.y *= gs; gv.z *= gs; gv.w *= gs;
      reinterpret_cast<float4*>(g)[i] = gv;
    }
    float4 bv = first ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<float4*>(buf)[i];
    if (UNSCALE) {
      const float4 p0 = pv, b0 = bv;
      pv.x = sgd_one(pv.x, gv.x, bv.x, lr, mu, wd, first);
      pv.y = sgd_one(pv.y, gv.y, bv.y, lr, mu, wd, first);
      pv.z = sgd_one(pv.z, gv.z, bv.z, lr, mu, wd, first);
      pv.w = sgd_one(pv.w, gv.w, bv.w, lr, mu, wd, first);
      if (!isfinite(gv.x)) { pv.x = p0.x; bv.x = b0.x; bad = true; }
      if (!isfinite(gv.y)) { pv.y = p0.y; bv.y = b0.y; bad = true; }
      if (!isfinite(gv.z)) { pv.z = p0.z; bv.z = b0.z; bad = true; }
      if (!isfinite(gv.w)) { pv.w = p0.w; bv.w = b0.w; bad = true; }
    } else {
      pv.x = sgd_one(pv.x, gv.x, bv.x, lr, mu, wd, first);
      pv.y = sgd_one(pv.y, gv.y, bv.y, lr, mu, wd, first);
      pv.z = sgd_one(pv.z, gv.z, bv.z, lr, mu, wd, first);
      pv.w = sgd_one(pv.w, gv.w, bv.w, lr, mu, wd, first);
    }
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(buf)[i] = bv;
    if (shadow) {
      uint2 o;
      o.x = pack_bf2(pv.x, pv.y);
      o.y = pack_bf2(pv.z, pv.w);
      reinterpret_cast<uint2*>(shadow)[i] = o;
    }
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float b = first ? 0.f : buf[i];
    float gi = g[i];
    if (UNSCALE) { gi *= gs; g[i] = gi; }
    const float p0 = p[i], b0 = b;
    float v = sgd_one(p0, gi, b, lr, mu, wd, first);
    if (UNSCALE && !isfinite(gi)) { v = p0; b = b0; bad = true; }
    p[i] = v;
    buf[i] = b;
    if (shadow) shadow[i] = f2bf(v);
  }
  if (UNSCALE && bad && ovf) *ovf = 1u;          // (every writer stores the same value)
}

int optim_sgd(float* p, float* g, float* buf, bf16_t* shadow, size_t n, float lr, float mu, float wd, int first,
              hipStream_t st, float gscale, unsigned* overflow) {
  FEDFR_REQUIRE(p && g && buf && n > 0, "sgd: bad args");
  FEDFR_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)buf) & 15) == 0, "sgd: buffers must be 16-byte aligned");
  const size_t work = n / 4 + 1;
  const int grid = (int)((work + 255) / 256 > 4096 ? 4096 : (work + 255) / 256);
  const bool unscale = gscale != 1.f || overflow != nullptr;
  ProfScope prof(26, (double)n * (first ? 16.0 : 20.0) + (shadow ? 2.0 * n : 0.0) + (unscale ? 4.0 * n : 0.0), st);      // p, g (, buf) read; p, buf (, bf16 mirror, unscaled g) written
  if (unscale)
    hipLaunchKernelGGL(sgd_kernel<true>, dim3(grid), dim3(256), 0, st, p, g, buf, shadow, n, lr, mu, wd, first, gscale, overflow);
  else
    hipLaunchKernelGGL(sgd_kernel<false>, dim3(grid), dim3(256), 0, st, p, g, buf, shadow, n, lr, mu, wd, first, 1.f, nullptr);
  FEDFR_LAUNCH_CHECK("sgd");
  return FEDFR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// FedAvg: dst = (accumulate ? dst : 0) + w * src, fp32, explicit op order (server.py:27-33: tmp += w_i * m_i)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fedavg_axpy_kernel(float* __restrict__ dst, const float* __restrict__ src, float w,
                                                          size_t n, int accumulate) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 s = reinterpret_cast<const float4*>(src)[i];
    float4 d = accumulate ? reinterpret_cast<float4*>(dst)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    d.x = __fadd_rn(d.x, __fmul_rn(w, s.x));
    d.y = __fadd_rn(d.y, __fmul_rn(w, s.y));
    d.z = __fadd_rn(d.z, __fmul_rn(w, s.z));
    d.w = __fadd_rn(d.w, __fmul_rn(w, s.w));
    reinterpret_cast<float4*>(dst)[i] = d;
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float d = accumulate ? dst[i] : 0.f;
    dst[i] = __fadd_rn(d, __fmul_rn(w, src[i]));
  }
}
int optim_fedavg_axpy(float* dst, const float* src, float w, size_t n, int accumulate, hipStream_t st) {
  FEDFR_REQUIRE(dst && src && n > 0, "fedavg_axpy: bad args");
  FEDFR_REQUIRE((((uintptr_t)dst | (uintptr_t)src) & 15) == 0, "fedavg_axpy: buffers must be 16-byte aligned");
  const size_t work = n / 4 + 1;
  const int grid = (int)((work + 255) / 256 > 4096 ? 4096 : (work + 255) / 256);
  hipLaunchKernelGGL(fedavg_axpy_kernel, dim3(grid), dim3(256), 0, st, dst, src, w, n, accumulate);
  FEDFR_LAUNCH_CHECK("fedavg_axpy");
  return FEDFR_OK;
}

// FedAvg over up to 8 client states in ONE pass: dst = (accumulate ? dst : 0) + w_0 * src_0 + w_1 * src_1 + ... in ascending client order with
// the same two roundings per term as fedavg_axpy_kernel, so the result is bit-identical to k sequential axpy launches (server.py:27-33) while
// every state is read once and the aggregate is written once: (k + 1) x n x 4 B instead of (3k - 1) x n x 4 B.
struct FedavgMulti {
  const float* src[8];
  float w[8];
};
template <int K>
__global__ __launch_bounds__(256) void fedavg_multi_kernel(float* __restrict__ dst, FedavgMulti p, size_t n, int accumulate) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 s[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      typedef float f4v __attribute__((ext_vector_type(4)));
      const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p.src[k]) + i);      // each state is read once
      s[k] = make_float4(v.x, v.y, v.z, v.w);
    }
    float4 d = accumulate ? reinterpret_cast<float4*>(dst)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      d.x = __fadd_rn(d.x, __fmul_rn(p.w[k], s[k].x));
      d.y = __fadd_rn(d.y, __fmul_rn(p.w[k], s[k].y));
      d.z = __fadd_rn(d.z, __fmul_rn(p.w[k], s[k].z));
      d.w = __fadd_rn(d.w, __fmul_rn(p.w[k], s[k].w));
    }
    reinterpret_cast<float4*>(dst)[i] = d;
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float d = accumulate ? dst[i] : 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) d = __fadd_rn(d, __fmul_rn(p.w[k], p.src[k][i]));
    dst[i] = d;
  }
}
int optim_fedavg_multi(float* dst, const float* const* srcs, const float* ws, int k, size_t n, int accumulate, hipStream_t st) {
  FEDFR_REQUIRE(dst && srcs && ws && n > 0 && k >= 1 && k <= 8, "fedavg_multi: bad args (k=%d)", k);
  FedavgMulti p{};
  uintptr_t al = (uintptr_t)dst;
  for (int i = 0; i < k; ++i) {
    FEDFR_REQUIRE(srcs[i] != nullptr, "fedavg_multi: source %d is null", i);
    p.src[i] = srcs[i];
    p.w[i] = ws[i];
    al |= (uintptr_t)srcs[i];
  }
  FEDFR_REQUIRE((al & 15) == 0, "fedavg_multi: buffers must be 16-byte aligned");
  const size_t work = n / 4 + 1;
  const int grid = (int)((work + 255) / 256 > 2048 ? 2048 : (work + 255) / 256);
  switch (k) {
#define FM_CASE(K_) case K_: hipLaunchKernelGGL(fedavg_multi_kernel<K_>, dim3(grid), dim3(256), 0, st, dst, p, n, accumulate); break;
    FM_CASE(1) FM_CASE(2) FM_CASE(3) FM_CASE(4) FM_CASE(5) FM_CASE(6) FM_CASE(7) FM_CASE(8)
#undef FM_CASE
  }
  FEDFR_LAUNCH_CHECK("fedavg_multi");
  return FEDFR_OK;
}

// int64 counters (num_batches_tracked): acc_f32 (+)= w * float(src) ; optional final truncation back to int64
__global__ void fedavg_i64_kernel(float* acc, const long long* src, float w, int n, int accumulate, long long* out_trunc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float d = accumulate ? acc[i] : 0.f;
  const float v = __fadd_rn(d, __fmul_rn(w, (float)src[i]));
  acc[i] = v;
  if (out_trunc) out_trunc[i] = (long long)v;
}
int optim_fedavg_i64(float* acc, const long long* src, float w, int n, int accumulate, long long* out_trunc, hipStream_t st) {
  FEDFR_REQUIRE(acc && src && n > 0, "fedavg_i64: bad args");
  hipLaunchKernelGGL(fedavg_i64_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, acc, src, w, n, accumulate, out_trunc);
  FEDFR_LAUNCH_CHECK("fedavg_i64");
  return FEDFR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// PartialFC sampling
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned hash_u32(unsigned long long x) {   // splitmix64 finaliser
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (unsigned)(x >> 32);
}
__global__ void pfc_rand_kernel(float* perm, int n, unsigned long long seed, unsigned long long step) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) perm[i] = (float)(hash_u32(seed * 0x100000001B3ull + step * 0x9E3779B1ull + (unsigned long long)i * 0xD6E8FEB86659FD93ull) >> 8) * (1.0f / 16777216.0f);
}
int optim_pfc_rand(float* perm, int n, unsigned long long seed, unsigned long long step, hipStream_t st) {
  FEDFR_REQUIRE(perm && n > 0, "pfc_rand: bad args");
  hipLaunchKernelGGL(pfc_rand_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, perm, n, seed, step);
  FEDFR_LAUNCH_CHECK("pfc_rand");
  return FEDFR_OK;
}

// labels -> local ids (or -1); perm[local] = 2.0 for positives (partial_fc.py:91-98)
__global__ void pfc_localize_kernel(long long* label, int n, long long class_start, int num_local, float* perm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long l = label[i] - class_start;
  if (l >= 0 && l < num_local) {
    label[i] = l;
    if (perm) perm[l] = 2.0f;
  } else {
    label[i] = -1;
  }
}
int optim_pfc_localize(long long* label, int n, long long class_start, int num_local, float* perm, hipStream_t st) {
  FEDFR_REQUIRE(label && n > 0 && num_local > 0, "pfc_localize: bad args");
  hipLaunchKernelGGL(pfc_localize_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, label, n, class_start, num_local, perm);
  FEDFR_LAUNCH_CHECK("pfc_localize");
  return FEDFR_OK;
}

// Top-k of non-negative floats as an index set in ascending index order == sort(topk(perm,k).indices).
// One 1024-thread block: 3-pass radix select (11+11+10 bits) for the k-th largest value, then ordered compaction.
__device__ __forceinline__ int block_excl_scan_1024(int v, int* sh, int& total) {
  // sh: 17 ints.  returns exclusive prefix of v over the block; total = block sum
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) sh[w] = x;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < 16; ++i) {
      const int t = sh[i];
      sh[i] = run;
      run += t;
    }
    sh[16] = run;
  }
  __syncthreads();
  const int res = sh[w] + x - v;
  total = sh[16];
  __syncthreads();
  return res;
}

// The digit of the kk-th largest candidate: the highest bin whose count, added to the counts of the bins above it, reaches kk (bin 0 if none
// does).  Block-parallel (1024 threads, nb = 1024 or 2048 bins, descending bin order through one block scan): a single thread walking up to 2 048
// LDS words in a dependent chain was ~70 us per pass at n = 85 000 (round 5 trace).  Ends with a barrier; *s_prefix / *s_k are then visible.
__device__ __forceinline__ void pfc_select_bin(const int* hist, int nb, int kk, unsigned prefix, int shift, int* sh, unsigned* s_prefix, int* s_k) {
  const int per = nb >> 10;                              // bins per thread (1 or 2), thread t owns bins nb - 1 - per t ... downwards
  const int top = nb - 1 - per * (int)threadIdx.x;
  int mine = 0;
  for (int q = 0; q < per; ++q) mine += hist[top - q];
  int total;
  int above = block_excl_scan_1024(mine, sh, total);     // candidates in the bins above this thread's
  for (int q = 0; q < per; ++q) {
    const int bin = top - q, h = hist[bin];
    if (above < kk && (above + h >= kk || bin == 0)) {
      *s_prefix = prefix | ((unsigned)bin << shift);
      *s_k = kk - above;
    }
    above += h;
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void pfc_topk_kernel(const float* __restrict__ perm, int n, int k, long long* __restrict__ index,
                                                        int* __restrict__ npos_out) {
  __shared__ int hist[2048];
  __shared__ int sh[17];
  __shared__ unsigned s_prefix;
  __shared__ int s_k;
  const unsigned* bits = reinterpret_cast<const unsigned*>(perm);
  const int tid = threadIdx.x;
  unsigned prefix = 0;     // selected high bits so far
  int kk = k;              // how many still to take from the current candidate set
  const int shifts[3] = {21, 10, 0};
  const int widths[3] = {11, 11, 10};
  unsigned mask_hi = 0;    // mask of bits already fixed
  for (int pass = 0; pass < 3; ++pass) {
    const int nb = 1 << widths[pass];
    for (int i = tid; i < nb; i += 1024) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
      const unsigned b = bits[i];
      if ((b & mask_hi) == prefix) atomicAdd(&hist[(b >> shifts[pass]) & (nb - 1)], 1);
    }
    __syncthreads();
    pfc_select_bin(hist, nb, kk, prefix, shifts[pass], sh, &s_prefix, &s_k);
    prefix = s_prefix;
    kk = s_k;
    mask_hi |= (unsigned)(nb - 1) << shifts[pass];
    __syncthreads();
  }
  // prefix == bit pattern T of the k-th largest value; take all > T and the first kk of == T (index order)
  const unsigned T = prefix;
  int base = 0, eq_taken = 0, npos = 0;
  for (int c0 = 0; c0 < n; c0 += 1024) {
    const int i = c0 + tid;
    const unsigned b = i < n ? bits[i] : 0u;
    const int gt = (i < n && b > T) ? 1 : 0;
    const int eq = (i < n && b == T) ? 1 : 0;
    int tot_eq, tot_sel;
    const int eq_rank = block_excl_scan_1024(eq, sh, tot_eq);
    const int sel = gt | (eq && (eq_taken + eq_rank) < kk ? 1 : 0);
    const int pos = block_excl_scan_1024(sel, sh, tot_sel);
    if (sel) index[base + pos] = i;
    if (i < n && b == 0x40000000u) ++npos;     // 2.0f marks a positive class
    base += tot_sel;
    eq_taken += tot_eq;
  }
  if (npos_out) {
    int tot;
    (void)block_excl_scan_1024(npos, sh, tot);
    if (tid == 0) *npos_out = tot;
  }
}
// The same selection with WIDE, prefetched loads (round 5).  The kernel above walks the n values four times (three histogram passes + the
// compaction) with ONE dependent 4-byte global load per thread and iteration: at n = 85 000 that is 4 x 83 L2 round trips = 319 us of the 930 us
// PartialFC head (profiles/r05_pfc_head_trace_v1.txt).  Here a thread owns NPT CONSECUTIVE values of every chunk of 1024 NPT (fetched as 16-byte
// loads, the next chunk's in flight while this one is counted), so a pass is n / (1024 NPT) round trips and the ordered compaction needs one
// pair of block scans per chunk instead of one per 1024 values: same radix select, same index set.  The histogram is kept in NCOPY copies
// (lane l adds to copy l % NCOPY, rows padded to 2049 words so that the copies of a bin sit in different banks): the values are uniform draws,
// so half of them share ONE 11-bit digit in the first pass, and 64 lanes adding to one LDS word are served one after the other (the
// single-copy form of this kernel: 256 us).
template <int NPT, int NCOPY>
__global__ __launch_bounds__(1024) void pfc_topk_wide_kernel(const float* __restrict__ perm, int n, int k, long long* __restrict__ index,
                                                             int* __restrict__ npos_out) {
  extern __shared__ int hcopies[];                      // [NCOPY][2049]
  __shared__ int hist[2048];
  __shared__ int sh[17];
  int* const myhist = hcopies + (threadIdx.x % NCOPY) * 2049;
  __shared__ unsigned s_prefix;
  __shared__ int s_k;
  static_assert(NPT % 4 == 0, "16-byte loads");
  const int tid = threadIdx.x;
  constexpr int CH = 1024 * NPT;
  const int nch = (n + CH - 1) / CH;
  auto load = [&](int c, unsigned (&v)[NPT]) {            // values [c CH + tid NPT, + NPT); beyond n: 0 with ok() false
#pragma unroll
    for (int j = 0; j < NPT / 4; ++j) {
      const int i = c * CH + tid * NPT + 4 * j;
      uint4 q = make_uint4(0u, 0u, 0u, 0u);
      if (i + 3 < n) q = *reinterpret_cast<const uint4*>(perm + i);
      else {
        const unsigned* b = reinterpret_cast<const unsigned*>(perm);
        if (i < n) q.x = b[i];
        if (i + 1 < n) q.y = b[i + 1];
        if (i + 2 < n) q.z = b[i + 2];
      }
      v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
    }
  };
  unsigned prefix = 0, mask_hi = 0;
  int kk = k;
  const int shifts[3] = {21, 10, 0};
  const int widths[3] = {11, 11, 10};
  unsigned cur[NPT], nxt[NPT];
  for (int pass = 0; pass < 3; ++pass) {
    const int nb = 1 << widths[pass];
    for (int i = tid; i < NCOPY * 2049; i += 1024) hcopies[i] = 0;
    load(0, cur);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
      if (c + 1 < nch) load(c + 1, nxt);
      const int nv = n - (c * CH + tid * NPT);
#pragma unroll
      for (int j = 0; j < NPT; ++j)
        if (j < nv && (cur[j] & mask_hi) == prefix) atomicAdd(&myhist[(cur[j] >> shifts[pass]) & (nb - 1)], 1);
#pragma unroll
      for (int j = 0; j < NPT; ++j) cur[j] = nxt[j];
    }
    __syncthreads();
    for (int i = tid; i < nb; i += 1024) {              // the copies of a bin, summed
      int t = 0;
#pragma unroll
      for (int q = 0; q < NCOPY; ++q) t += hcopies[q * 2049 + i];
      hist[i] = t;
    }
    __syncthreads();
    pfc_select_bin(hist, nb, kk, prefix, shifts[pass], sh, &s_prefix, &s_k);
    prefix = s_prefix;
    kk = s_k;
    mask_hi |= (unsigned)(nb - 1) << shifts[pass];
    __syncthreads();
  }
  // prefix == bit pattern T of the k-th largest value; take all > T and the first kk of == T (index order)
  const unsigned T = prefix;
  int base = 0, eq_taken = 0, npos = 0;
  load(0, cur);
  for (int c = 0; c < nch; ++c) {
    if (c + 1 < nch) load(c + 1, nxt);
    const int i0 = c * CH + tid * NPT, nv = n - i0;
    int n_gt = 0, n_eq = 0;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      n_gt += (j < nv && cur[j] > T) ? 1 : 0;
      n_eq += (j < nv && cur[j] == T) ? 1 : 0;
      npos += (j < nv && cur[j] == 0x40000000u) ? 1 : 0;     // 2.0f marks a positive class
    }
    int tot_eq, tot_gt;
    int eq_before = eq_taken + block_excl_scan_1024(n_eq, sh, tot_eq);
    const int gt_before = block_excl_scan_1024(n_gt, sh, tot_gt);
    // selected elements in front of this thread: `base` from earlier chunks (their > T and taken == T), this chunk's predecessors' > T, and of
    // all == T in front of it the first kk — minus those already counted in `base`
    int pos = base + gt_before + (min(eq_before, kk) - min(eq_taken, kk));
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const bool gt = j < nv && cur[j] > T, eq = j < nv && cur[j] == T;
      if (gt || (eq && eq_before < kk)) index[pos++] = i0 + j;
      eq_before += eq ? 1 : 0;
    }
    base += tot_gt + (min(eq_taken + tot_eq, kk) - min(eq_taken, kk));
    eq_taken += tot_eq;
#pragma unroll
    for (int j = 0; j < NPT; ++j) cur[j] = nxt[j];
  }
  if (npos_out) {
    int tot;
    (void)block_excl_scan_1024(npos, sh, tot);
    if (tid == 0) *npos_out = tot;
  }
}
int optim_pfc_topk(const float* perm, int n, int k, long long* index, int* npos_out, hipStream_t st) {
  FEDFR_REQUIRE(perm && index && n > 0 && k > 0 && k <= n, "pfc_topk: bad args (k=%d n=%d)", k, n);
  if ((reinterpret_cast<size_t>(perm) & 15) == 0 && n >= 8192) {
    constexpr int NCOPY = 16;
    constexpr size_t lds = (size_t)NCOPY * 2049 * sizeof(int);
    static PerDeviceOnce attr_once;     // hipFuncSetAttribute is per device
    attr_once.run([&] {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&pfc_topk_wide_kernel<16, NCOPY>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    hipLaunchKernelGGL((pfc_topk_wide_kernel<16, NCOPY>), dim3(1), dim3(1024), lds, st, perm, n, k, index, npos_out);
  } else hipLaunchKernelGGL(pfc_topk_kernel, dim3(1), dim3(1024), 0, st, perm, n, k, index, npos_out);      // (small or unaligned shards: the 4-byte form)
  FEDFR_LAUNCH_CHECK("pfc_topk");
  return FEDFR_OK;
}

// positives only (num_sample < #positives branch, partial_fc.py:101-102): ordered compaction of perm == 2.0
__global__ __launch_bounds__(1024) void pfc_positive_kernel(const float* __restrict__ perm, int n, long long* __restrict__ index,
                                                            int* __restrict__ count) {
  __shared__ int sh[17];
  int base = 0;
  for (int c0 = 0; c0 < n; c0 += 1024) {
    const int i = c0 + threadIdx.x;
    const int sel = (i < n && perm[i] == 2.0f) ? 1 : 0;
    int tot;
    const int pos = block_excl_scan_1024(sel, sh, tot);
    if (sel) index[base + pos] = i;
    base += tot;
  }
  if (threadIdx.x == 0) *count = base;
}
int optim_pfc_positive(const float* perm, int n, long long* index, int* count, hipStream_t st) {
  FEDFR_REQUIRE(perm && index && count && n > 0, "pfc_positive: bad args");
  hipLaunchKernelGGL(pfc_positive_kernel, dim3(1), dim3(1024), 0, st, perm, n, index, count);
  FEDFR_LAUNCH_CHECK("pfc_positive");
  return FEDFR_OK;
}

// label[i] = searchsorted(index, label[i]) for label != -1 (partial_fc.py:104)
__global__ void pfc_remap_kernel(long long* label, int n, const long long* __restrict__ index, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long l = label[i];
  if (l < 0) return;
  int lo = 0, hi = k;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (index[mid] < l) lo = mid + 1; else hi = mid;
  }
  label[i] = lo;
}
int optim_pfc_remap(long long* label, int n, const long long* index, int k, hipStream_t st) {
  FEDFR_REQUIRE(label && index && n > 0 && k > 0, "pfc_remap: bad args");
  hipLaunchKernelGGL(pfc_remap_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, label, n, index, k);
  FEDFR_LAUNCH_CHECK("pfc_remap");
  return FEDFR_OK;
}

// row gather / scatter (fp32 rows of D floats, D % 4 == 0), one wave per row
__global__ __launch_bounds__(256) void rows_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                   const long long* __restrict__ index, int k, int D, int scatter, int nrows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= k) return;
  const long long r = index[row];
  if (r < 0 || r >= nrows) return;                     // never touch memory outside the indexed table
  const float4* s = reinterpret_cast<const float4*>(src + (scatter ? (size_t)row : (size_t)r) * D);
  float4* d = reinterpret_cast<float4*>(dst + (scatter ? (size_t)r : (size_t)row) * D);
  for (int i = lane; i < D / 4; i += 64) d[i] = s[i];
}
int optim_rows(float* dst, const float* src, const long long* index, int k, int D, int scatter, int nrows, hipStream_t st) {
  FEDFR_REQUIRE(dst && src && index && k > 0 && D > 0 && (D & 3) == 0 && nrows > 0, "rows gather/scatter: bad args");
  hipLaunchKernelGGL(rows_kernel, dim3(ceil_div(k, 4)), dim3(256), 0, st, dst, src, index, k, D, scatter, nrows);
  FEDFR_LAUNCH_CHECK("rows");
  return FEDFR_OK;
}
