// fp32 head kernels: row L2-normalisation (fwd/bwd), strided fp32 GEMM on v_mfma_f32_16x16x4_f32 (exact fp32 FMA
// chain), CosFace/ArcFace margin + (optionally distributed) softmax-CE with hand-written gradient
// (reference: client.py:69-74, losses.py:17-45, partial_fc.py:130-176), BCE personalised head elementwise part
// (client.py:45-58, losses.py:4-15).  The head stays fp32: s=30..64 amplifies cosine error.
#include "head.h"

// ---------------------------------------------------------------------------------------------------------
// normalisation: one wave per row
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ x, float* __restrict__ xn,
                                                             float* __restrict__ inv, int R, int D, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const float* xr = x + (size_t)row * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += xr[i] * xr[i];
  s = wave_sum(s);
  const float iv = 1.f / fmaxf(sqrtf(s), eps);
  if (lane == 0 && inv) inv[row] = iv;
  if (xn)
    for (int i = lane; i < D; i += 64) xn[(size_t)row * D + i] = xr[i] * iv;
}
int head_normalize_rows(const float* x, float* xn, float* inv, int R, int D, float eps, hipStream_t st) {
  FEDFR_REQUIRE(x && R > 0 && D > 0, "normalize_rows: bad args");
  hipLaunchKernelGGL(normalize_rows_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, st, x, xn, inv, R, D, eps);
  FEDFR_LAUNCH_CHECK("normalize_rows");
  return FEDFR_OK;
}

// dx = inv * (dxn - xn * <xn, dxn>)     (F.normalize backward; the eps clamp branch has zero measure)
// dxn may arrive as `nslab` split-K slabs of the GEMM that produced it (head_sgemm_splitk): summed here, slab 0 first
// WIDE (D <= 64 * NRB_J): a lane's elements (lane, lane + 64, ...) are fetched once per slab with all loads of a slab in flight together and stay in
// registers for both passes (round 4: with 7 split-K slabs the element-at-a-time form was a chain of ~110 dependent round trips = 21 us on the
// serial head chain).  Same additions in the same order.
constexpr int NRB_J = 8;
template <bool WIDE>
__global__ __launch_bounds__(256) void normalize_rows_bwd_kernel(const float* __restrict__ xn, const float* __restrict__ inv,
                                                                 const float* __restrict__ dxn, float* __restrict__ dx,
                                                                 int R, int D, float beta, int nslab, long long slab_stride) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const size_t o = (size_t)row * D;
  if constexpr (WIDE) {
    float gv[NRB_J], xv[NRB_J];
#pragma unroll
    for (int j = 0; j < NRB_J; ++j) {
      const int i = min(lane + 64 * j, D - 1);
      xv[j] = xn[o + i];
      gv[j] = dxn[o + i];
    }
    for (int k = 1; k < nslab; ++k) {
      float t[NRB_J];
#pragma unroll
      for (int j = 0; j < NRB_J; ++j) t[j] = dxn[(size_t)k * slab_stride + o + min(lane + 64 * j, D - 1)];
#pragma unroll
      for (int j = 0; j < NRB_J; ++j) gv[j] += t[j];
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NRB_J; ++j) if (lane + 64 * j < D) s += xv[j] * gv[j];
    s = wave_sum(s);
    const float iv = inv[row];
#pragma unroll
    for (int j = 0; j < NRB_J; ++j) {
      const int i = lane + 64 * j;
      if (i < D) {
        const float v = iv * (gv[j] - xv[j] * s);
        dx[o + i] = beta != 0.f ? beta * dx[o + i] + v : v;
      }
    }
  } else {
  auto g = [&](int i) {
    float v = dxn[o + i];
    for (int k = 1; k < nslab; ++k) v += dxn[(size_t)k * slab_stride + o + i];
    return v;
  };
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += xn[o + i] * g(i);
  s = wave_sum(s);
  const float iv = inv[row];
  for (int i = lane; i < D; i += 64) {
    const float v = iv * (g(i) - xn[o + i] * s);
    dx[o + i] = beta != 0.f ? beta * dx[o + i] + v : v;
  }
  }
}
int head_normalize_rows_bwd_slabs(const float* xn, const float* inv, const float* dxn, int nslab, long long slab_stride, float* dx, int R, int D,
                                  float beta, hipStream_t st) {
  FEDFR_REQUIRE(xn && inv && dxn && dx && R > 0 && D > 0 && nslab >= 1 && (nslab == 1 || slab_stride >= (long long)R * D), "normalize_rows_bwd: bad args");
  if (D <= 64 * NRB_J)
    hipLaunchKernelGGL(normalize_rows_bwd_kernel<true>, dim3(ceil_div(R, 4)), dim3(256), 0, st, xn, inv, dxn, dx, R, D, beta, nslab, slab_stride);
  else
    hipLaunchKernelGGL(normalize_rows_bwd_kernel<false>, dim3(ceil_div(R, 4)), dim3(256), 0, st, xn, inv, dxn, dx, R, D, beta, nslab, slab_stride);
  FEDFR_LAUNCH_CHECK("normalize_rows_bwd");
  return FEDFR_OK;
}
int head_normalize_rows_bwd(const float* xn, const float* inv, const float* dxn, float* dx, int R, int D, float beta,
                            hipStream_t st) {
  return head_normalize_rows_bwd_slabs(xn, inv, dxn, 1, 0, dx, R, D, beta, st);
}

// ---------------------------------------------------------------------------------------------------------
// strided fp32 GEMM  C[m][n] = alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn] (+ bias[n]) (+ beta*C)
// block tile 64x64, 4 waves (2x2) of 32x32, BK = 16, v_mfma_f32_16x16x4_f32
// ---------------------------------------------------------------------------------------------------------
#ifndef SGEMM_COLFLAG_BK
#define SGEMM_COLFLAG_BK 16      // k depth of the hard-negative-mining GEMM (throughput-bound: measured per value below)
#endif
struct SgemmP {
  const float* A; const float* B; float* C;
  int M, N, K;
  long long sam, sak, sbk, sbn;
  int ldc;
  float alpha, beta;
  const float* bias;
  unsigned char* colflag;   // != null: store nothing, set colflag[n] = 1 for every column with some alpha * (A B)[m][n] > thr
  float thr;
  int kchunk;               // split-K (gridDim.z > 1): block z sums k in [z * kchunk, min(K, (z + 1) * kchunk)) into slab C + z * slab_stride
  long long slab_stride;
};

template <int BK>      // k depth of a stage: 16, or 32 (half as many global-load round trips on the K loop: the head's 128 x 1000 x 512 GEMMs are a latency chain)
__global__ __launch_bounds__(256) void sgemm_kernel(SgemmP p) {
  constexpr int BM = 64, BN = 64, LD = 80, NL = BK / 4;   // k-major LDS rows; LD%32==16 + column XOR (k>>1)<<1: reads and writes conflict-free
  __shared__ float sA[2][BK][LD], sB[2][BK][LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  float ra[NL], rb[NL];
  auto load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int e = tid + 256 * i;
      int m, k;
      if (p.sak == 1) { k = e & (BK - 1); m = e / BK; } else { m = e & 63; k = e >> 6; }
      const int gm = m0 + m, gk = k0 + k;
      ra[i] = (gm < p.M && gk < p.K) ? p.A[(long long)gm * p.sam + (long long)gk * p.sak] : 0.f;
      int n, kb;
      if (p.sbn == 1) { n = e & 63; kb = e >> 6; } else { kb = e & (BK - 1); n = e / BK; }
      const int gn = n0 + n, gkb = k0 + kb;
      rb[i] = (gn < p.N && gkb < p.K) ? p.B[(long long)gkb * p.sbk + (long long)gn * p.sbn] : 0.f;
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int e = tid + 256 * i;
      int m, k;
      if (p.sak == 1) { k = e & (BK - 1); m = e / BK; } else { m = e & 63; k = e >> 6; }
      sA[buf][k][m ^ ((k >> 1) << 1)] = ra[i];
      int n, kb;
      if (p.sbn == 1) { n = e & 63; kb = e >> 6; } else { kb = e & (BK - 1); n = e / BK; }
      sB[buf][kb][n ^ ((kb >> 1) << 1)] = rb[i];
    }
  };
  f32x4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // split-K: this block's k range (the loads mask k >= p.K, so the range end is made the problem's K for them)
  const int kbeg = gridDim.z > 1 ? (int)blockIdx.z * p.kchunk : 0;
  if (gridDim.z > 1) {
    p.K = min(p.K, kbeg + p.kchunk);
    p.C += (size_t)blockIdx.z * p.slab_stride;
  }
  const int nk = ceil_div(p.K - kbeg, BK);
  load(kbeg);
  store(0);
  __syncthreads();
  const int l15 = lane & 15, lg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load(kbeg + (kt + 1) * BK);
#pragma unroll
    for (int k4 = 0; k4 < BK; k4 += 4) {
      float fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kk = k4 + lg, sw = (kk >> 1) << 1;
        fa[i] = sA[buf][kk][(wm * 32 + i * 16 + l15) ^ sw];
        fb[i] = sB[buf][kk][(wn * 32 + i * 16 + l15) ^ sw];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store(buf ^ 1);
    __syncthreads();
  }
  // D[row = m][col = n]: m = wm*32 + i*16 + lg*4 + reg, n = wn*32 + j*16 + l15
  if (p.colflag) {            // threshold + column-OR epilogue (hard-negative mining): every hit stores the same 1 -> order-free
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 32 + j * 16 + l15;
      bool hit = false;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = m0 + wm * 32 + i * 16 + lg * 4 + q;
          hit |= m < p.M && p.alpha * acc[i][j][q] > p.thr;
        }
      if (hit && n < p.N) p.colflag[n] = 1;
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + wm * 32 + i * 16 + lg * 4 + q, n = n0 + wn * 32 + j * 16 + l15;
        if (m < p.M && n < p.N) {
          float v = p.alpha * acc[i][j][q];
          if (p.bias) v += p.bias[n];
          float* c = p.C + (size_t)m * p.ldc + n;
          if (p.beta != 0.f) v += p.beta * *c;
          *c = v;
        }
      }
}

// Same GEMM with the products accumulated in fp64 (v_mfma_f64_16x16x4_f64 on the fp32 operands widened exactly): the result is the
// correctly rounded fp32 value of the exact sum for all practical purposes.  Used by the fp32 validation path of the backbone (net_f32.hip),
// whose weight-gradient GEMMs sum over up to 10^5 positions: a sequential fp32 accumulation of that length put it 3x further from the fp64
// evaluation of a training step than the fp32 reference is.
template <int BK>
__global__ __launch_bounds__(256) void sgemm_f64acc_kernel(SgemmP p) {
  constexpr int BM = 64, BN = 64, LD = 80, NL = BK / 4;   // k-major LDS rows; LD%32==16 + column XOR (k>>1)<<1: reads and writes conflict-free
  __shared__ float sA[2][BK][LD], sB[2][BK][LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  float ra[NL], rb[NL];
  auto load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int e = tid + 256 * i;
      int m, k;
      if (p.sak == 1) { k = e & (BK - 1); m = e / BK; } else { m = e & 63; k = e >> 6; }
      const int gm = m0 + m, gk = k0 + k;
      ra[i] = (gm < p.M && gk < p.K) ? p.A[(long long)gm * p.sam + (long long)gk * p.sak] : 0.f;
      int n, kb;
      if (p.sbn == 1) { n = e & 63; kb = e >> 6; } else { kb = e & (BK - 1); n = e / BK; }
      const int gn = n0 + n, gkb = k0 + kb;
      rb[i] = (gn < p.N && gkb < p.K) ? p.B[(long long)gkb * p.sbk + (long long)gn * p.sbn] : 0.f;
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int e = tid + 256 * i;
      int m, k;
      if (p.sak == 1) { k = e & (BK - 1); m = e / BK; } else { m = e & 63; k = e >> 6; }
      sA[buf][k][m ^ ((k >> 1) << 1)] = ra[i];
      int n, kb;
      if (p.sbn == 1) { n = e & 63; kb = e >> 6; } else { kb = e & (BK - 1); n = e / BK; }
      sB[buf][kb][n ^ ((kb >> 1) << 1)] = rb[i];
    }
  };
  typedef __attribute__((ext_vector_type(4))) double f64x4_acc_t;
  f64x4_acc_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f64x4_acc_t){0.0, 0.0, 0.0, 0.0};
  const int nk = ceil_div(p.K, BK);
  load(0);
  store(0);
  __syncthreads();
  const int l15 = lane & 15, lg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load((kt + 1) * BK);
#pragma unroll
    for (int k4 = 0; k4 < BK; k4 += 4) {
      float fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kk = k4 + lg, sw = (kk >> 1) << 1;
        fa[i] = sA[buf][kk][(wm * 32 + i * 16 + l15) ^ sw];
        fb[i] = sB[buf][kk][(wn * 32 + i * 16 + l15) ^ sw];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa[i], (double)fb[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store(buf ^ 1);
    __syncthreads();
  }
  // f64 16x16x4 accumulator layout (differs from the f32 form): register q of lane l holds D[row = 4 q + (l >> 4)][col = l & 15]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + wm * 32 + i * 16 + 4 * q + lg, n = n0 + wn * 32 + j * 16 + l15;
        if (m < p.M && n < p.N) {
          double v = (double)p.alpha * acc[i][j][q];
          if (p.bias) v += (double)p.bias[n];
          float* c = p.C + (size_t)m * p.ldc + n;
          if (p.beta != 0.f) v += (double)p.beta * (double)*c;
          *c = (float)v;
        }
      }
}

int head_sgemm(const float* A, const float* B, float* C, int M, int N, int K, long long sam, long long sak,
               long long sbk, long long sbn, int ldc, float alpha, float beta, const float* bias, hipStream_t st) {
  FEDFR_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && ldc >= N, "sgemm: bad args");
  SgemmP p{A, B, C, M, N, K, sam, sak, sbk, sbn, ldc, alpha, beta, bias, nullptr, 0.f};
  if (K >= 128) hipLaunchKernelGGL(sgemm_kernel<32>, dim3(ceil_div(N, 64), ceil_div(M, 64)), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(sgemm_kernel<16>, dim3(ceil_div(N, 64), ceil_div(M, 64)), dim3(256), 0, st, p);
  FEDFR_LAUNCH_CHECK("sgemm");
  return FEDFR_OK;
}
// split-K form for the head's small GEMMs (128 x 1000 x 512 and 128 x 512 x 1000 are latency chains of 16 / 32 dependent k-steps on 16 - 32
// workgroups): `splits` slabs C + z * slab_stride, each the sum over one k range; the consumer adds the slabs in order (deterministic)
int head_sgemm_splitk(const float* A, const float* B, float* C, int M, int N, int K, long long sam, long long sak, long long sbk, long long sbn,
                      int ldc, float alpha, int splits, long long slab_stride, hipStream_t st) {
  FEDFR_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && ldc >= N && splits >= 1 && splits <= 64 && (splits == 1 || slab_stride >= (long long)M * ldc),
                "sgemm_splitk: bad args");
  SgemmP p{A, B, C, M, N, K, sam, sak, sbk, sbn, ldc, alpha, 0.f, nullptr, nullptr, 0.f, 0, slab_stride};
  p.kchunk = ceil_div(ceil_div(K, splits), 32) * 32;
  FEDFR_REQUIRE((long long)p.kchunk * (splits - 1) < K, "sgemm_splitk: %d splits leave an empty k range at K = %d", splits, K);
  hipLaunchKernelGGL(sgemm_kernel<32>, dim3(ceil_div(N, 64), ceil_div(M, 64), splits), dim3(256), 0, st, p);
  FEDFR_LAUNCH_CHECK("sgemm_splitk");
  return FEDFR_OK;
}
int head_sgemm_f64acc(const float* A, const float* B, float* C, int M, int N, int K, long long sam, long long sak, long long sbk, long long sbn,
                      int ldc, float alpha, float beta, const float* bias, hipStream_t st) {
  FEDFR_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && ldc >= N, "sgemm_f64acc: bad args");
  SgemmP p{A, B, C, M, N, K, sam, sak, sbk, sbn, ldc, alpha, beta, bias, nullptr, 0.f};
  hipLaunchKernelGGL(sgemm_f64acc_kernel<32>, dim3(ceil_div(N, 64), ceil_div(M, 64)), dim3(256), 0, st, p);
  FEDFR_LAUNCH_CHECK("sgemm_f64acc");
  return FEDFR_OK;
}
int head_sgemm_colflag(const float* A, const float* B, int M, int N, int K, long long sam, long long sak, long long sbk,
                       long long sbn, float alpha, float thr, unsigned char* flags, hipStream_t st) {
  FEDFR_REQUIRE(A && B && flags && M > 0 && N > 0 && K > 0, "sgemm_colflag: bad args");
  SgemmP p{A, B, nullptr, M, N, K, sam, sak, sbk, sbn, N, alpha, 0.f, nullptr, flags, thr};
  if (SGEMM_COLFLAG_BK == 32) hipLaunchKernelGGL(sgemm_kernel<32>, dim3(ceil_div(N, 64), ceil_div(M, 64)), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(sgemm_kernel<16>, dim3(ceil_div(N, 64), ceil_div(M, 64)), dim3(256), 0, st, p);
  FEDFR_LAUNCH_CHECK("sgemm_colflag");
  return FEDFR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// per-class feature sums of one batch (class-centre initialisation, reference client.py:171-178, server.py:213-222):
// sums[c] += sum over rows b with label[b] == c of x[b] (rows in batch order -> deterministic), counts[c] += #rows.
// One block per class; labels outside [0, C) are ignored.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void class_accumulate_kernel(const float* __restrict__ x, const long long* __restrict__ label, int B,
                                                               int D, int C, float* __restrict__ sums, float* __restrict__ counts) {
  const int c = blockIdx.x;
  __shared__ int rows[1024];
  __shared__ int nrows;
  for (int b0 = 0; b0 < B; b0 += 1024) {
    __syncthreads();
    if (threadIdx.x == 0) {                            // ordered compaction of the matching rows of this slice (B is a batch size)
      int n = 0;
      const int e = min(B, b0 + 1024);
      for (int b = b0; b < e; ++b)
        if (label[b] == (long long)c) rows[n++] = b;
      nrows = n;
    }
    __syncthreads();
    const int n = nrows;
    if (n == 0) continue;
    for (int d = threadIdx.x; d < D; d += 256) {
      float s = 0.f;
      for (int i = 0; i < n; ++i) s += x[(size_t)rows[i] * D + d];
      sums[(size_t)c * D + d] += s;
    }
    if (threadIdx.x == 0) counts[c] += (float)n;
  }
}
int head_class_accumulate(const float* x, const long long* label, int B, int D, int C, float* sums, float* counts, hipStream_t st) {
  FEDFR_REQUIRE(x && label && sums && counts && B > 0 && D > 0 && C > 0, "class_accumulate: bad args");
  hipLaunchKernelGGL(class_accumulate_kernel, dim3(C), dim3(256), 0, st, x, label, B, D, C, sums, counts);
  FEDFR_LAUNCH_CHECK("class_accumulate");
  return FEDFR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// margin + softmax CE in three steps so the row max / row sum can be all-reduced between them (PartialFC
// C3/C4/C5, partial_fc.py:142-161).  One 256-thread block per row.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return r;
}

// step 1: logits = margin(cos) * s in place; row_max; dmul[row] = d logit_target / d cos_target
__global__ __launch_bounds__(256) void margin_rowmax_kernel(float* __restrict__ z, const long long* __restrict__ label, int C,
                                                            int ldz, float s, float m, int arc, float* __restrict__ row_max,
                                                            float* __restrict__ dmul) {
  __shared__ float sh[4];
  const int row = blockIdx.x;
  float* zr = z + (size_t)row * ldz;
  const long long y = label[row];
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < C; c += 256) {
    float x = zr[c], v;
    if (arc) {
      float th = acosf(x);                     // unclamped, as losses.py:42
      if (c == y) {
        const float st = sinf(th);
        if (dmul) dmul[row] = s * sinf(th + m) / st;
        th += m;
      }
      v = cosf(th) * s;
    } else {
      if (c == y) {
        x -= m;
        if (dmul) dmul[row] = s;
      }
      v = x * s;
    }
    zr[c] = v;
    mx = fmaxf(mx, v);
  }
  mx = block_max(mx, sh);
  if (threadIdx.x == 0) {
    row_max[row] = mx;
    if (dmul && (y < 0 || y >= C)) dmul[row] = s;
  }
}

// step 2: z = exp(z - max) in place; row_sum
__global__ __launch_bounds__(256) void exp_rowsum_kernel(float* __restrict__ z, int C, int ldz, const float* __restrict__ row_max,
                                                         float* __restrict__ row_sum) {
  __shared__ float sh[4];
  const int row = blockIdx.x;
  float* zr = z + (size_t)row * ldz;
  const float mx = row_max[row];
  float sum = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float e = expf(zr[c] - mx);
    zr[c] = e;
    sum += e;
  }
  sum = block_sum(sum, sh);
  if (threadIdx.x == 0) row_sum[row] = sum;
}

// step 2 of the class-sharded softmax (PartialFC): as exp_rowsum, and the target's numerator rides along — sums2[row] = sum_c e,
// sums2[R + row] = e[label] (0 when this shard does not hold the row's class), so that ONE sum all-reduce of the [2][R] tensor replaces
// the reference's two (partial_fc.py:147 sum of exponentials, :161 target probability)
__global__ __launch_bounds__(256) void exp_rowsum_target_kernel(float* __restrict__ z, const long long* __restrict__ label, int R, int C, int ldz,
                                                                const float* __restrict__ row_max, float* __restrict__ sums2) {
  __shared__ float sh[4];
  const int row = blockIdx.x;
  float* zr = z + (size_t)row * ldz;
  const float mx = row_max[row];
  const long long y = label[row];
  float sum = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float e = expf(zr[c] - mx);
    zr[c] = e;
    sum += e;
    if (c == y) sums2[R + row] = e;
  }
  sum = block_sum(sum, sh);
  if (threadIdx.x == 0) {
    sums2[row] = sum;
    if (y < 0 || y >= C) sums2[R + row] = 0.f;
  }
}
// loss = -mean_r log(max(num[r] / den[r], floor))
__global__ __launch_bounds__(256) void nll_mean_ratio_kernel(const float* __restrict__ num, const float* __restrict__ den, int R, float floor_,
                                                             float* __restrict__ loss) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) s += -logf(fmaxf(num[r] / den[r], floor_));
  s = block_sum(s, sh);
  if (threadIdx.x == 0) *loss = s / R;
}

// step 3: p = z / sum; prob_t[row] = p[label] (0 when label == -1); grad wrt cos in place:
//   g[c] = (p[c] - [c == y]) * inv_batch * (c == y ? dmul[row] : s)
__global__ __launch_bounds__(256) void softmax_grad_kernel(float* __restrict__ z, const long long* __restrict__ label, int C, int ldz,
                                                           const float* __restrict__ row_sum, const float* __restrict__ dmul,
                                                           float s, float inv_batch, float* __restrict__ prob_t) {
  const int row = blockIdx.x;
  float* zr = z + (size_t)row * ldz;
  const float inv = 1.f / row_sum[row];
  const long long y = label[row];
  if (threadIdx.x == 0 && (y < 0 || y >= C)) prob_t[row] = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float pr = zr[c] * inv;
    float g = pr;
    float mul = s;
    if (c == y) {
      prob_t[row] = pr;
      g -= 1.f;
      mul = dmul[row];
    }
    zr[c] = g * inv_batch * mul;
  }
}

// steps 1 - 3 in ONE launch for rows of at most 256 * NPT classes (dense heads): the row stays in registers between the three passes, the
// block reductions and the per-element expressions are the three kernels' own (same thread -> column mapping: bit-identical results).  The
// cosines may arrive as `nslab` split-K slabs (head_sgemm_splitk), summed slab 0 first; the gradient is written over slab 0.
template <int NPT>
__global__ __launch_bounds__(256) void softmax_ce_fused_kernel(float* __restrict__ z, const long long* __restrict__ label, int C, int ldz, float s,
                                                               float m, int arc, float inv_batch, float* __restrict__ prob_t, int nslab,
                                                               long long slab_stride) {
  __shared__ float sh[4];
  const int row = blockIdx.x;
  float* zr = z + (size_t)row * ldz;
  const long long y = label[row];
  float v[NPT];
  float mx = -INFINITY, dm = s;
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int c = threadIdx.x + 256 * k;
    v[k] = -INFINITY;
    if (c < C) {
      float x = zr[c];
      for (int q = 1; q < nslab; ++q) x += zr[(size_t)q * slab_stride + c];
      float t;
      if (arc) {
        float th = acosf(x);
        if (c == y) {
          const float st = sinf(th);
          dm = s * sinf(th + m) / st;
          th += m;
        }
        t = cosf(th) * s;
      } else {
        if (c == y) x -= m;
        t = x * s;
      }
      v[k] = t;
      mx = fmaxf(mx, t);
    }
  }
  mx = block_max(mx, sh);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int c = threadIdx.x + 256 * k;
    if (c < C) {
      v[k] = expf(v[k] - mx);
      sum += v[k];
    }
  }
  __syncthreads();                                       // sh is reused
  sum = block_sum(sum, sh);
  const float inv = 1.f / sum;
  if (threadIdx.x == 0 && (y < 0 || y >= C)) prob_t[row] = 0.f;
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int c = threadIdx.x + 256 * k;
    if (c < C) {
      const float pr = v[k] * inv;
      float g = pr, mul = s;
      if (c == y) {
        prob_t[row] = pr;
        g -= 1.f;
        mul = dm;
      }
      zr[c] = g * inv_batch * mul;
    }
  }
}
int head_softmax_ce_fused(float* z, const long long* label, int R, int C, int ldz, float s, float m, int arc, float inv_batch, float* prob_t,
                          int nslab, long long slab_stride, hipStream_t st) {
  FEDFR_REQUIRE(z && label && prob_t && R > 0 && C > 0 && C <= 16384 && ldz >= C && nslab >= 1 && (nslab == 1 || slab_stride >= (long long)R * ldz),
                "softmax_ce_fused: bad args (rows of at most 16384 classes)");
  if (C <= 1024) hipLaunchKernelGGL(softmax_ce_fused_kernel<4>, dim3(R), dim3(256), 0, st, z, label, C, ldz, s, m, arc, inv_batch, prob_t, nslab, slab_stride);
  else if (C <= 4096) hipLaunchKernelGGL(softmax_ce_fused_kernel<16>, dim3(R), dim3(256), 0, st, z, label, C, ldz, s, m, arc, inv_batch, prob_t, nslab, slab_stride);
  else hipLaunchKernelGGL(softmax_ce_fused_kernel<64>, dim3(R), dim3(256), 0, st, z, label, C, ldz, s, m, arc, inv_batch, prob_t, nslab, slab_stride);      // (round 5: the sampled PartialFC head, 8 500 classes)
  FEDFR_LAUNCH_CHECK("softmax_ce_fused");
  return FEDFR_OK;
}

// loss = -mean_r log(max(prob_t[r], floor))   (floor = 1e-30 for PartialFC, 0 for F.cross_entropy)
__global__ __launch_bounds__(256) void nll_mean_kernel(const float* __restrict__ prob_t, int R, float floor_, float* __restrict__ loss) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) s += -logf(fmaxf(prob_t[r], floor_));
  s = block_sum(s, sh);
  if (threadIdx.x == 0) *loss = s / R;
}

int head_margin_rowmax(float* z, const long long* label, int R, int C, int ldz, float s, float m, int arc, float* row_max,
                       float* dmul, hipStream_t st) {
  FEDFR_REQUIRE(z && label && row_max && R > 0 && C > 0 && ldz >= C, "margin_rowmax: bad args");
  hipLaunchKernelGGL(margin_rowmax_kernel, dim3(R), dim3(256), 0, st, z, label, C, ldz, s, m, arc, row_max, dmul);
  FEDFR_LAUNCH_CHECK("margin_rowmax");
  return FEDFR_OK;
}
int head_exp_rowsum(float* z, int R, int C, int ldz, const float* row_max, float* row_sum, hipStream_t st) {
  FEDFR_REQUIRE(z && row_max && row_sum && R > 0 && C > 0, "exp_rowsum: bad args");
  hipLaunchKernelGGL(exp_rowsum_kernel, dim3(R), dim3(256), 0, st, z, C, ldz, row_max, row_sum);
  FEDFR_LAUNCH_CHECK("exp_rowsum");
  return FEDFR_OK;
}
int head_exp_rowsum_target(float* z, const long long* label, int R, int C, int ldz, const float* row_max, float* sums2, hipStream_t st) {
  FEDFR_REQUIRE(z && label && row_max && sums2 && R > 0 && C > 0, "exp_rowsum_target: bad args");
  hipLaunchKernelGGL(exp_rowsum_target_kernel, dim3(R), dim3(256), 0, st, z, label, R, C, ldz, row_max, sums2);
  FEDFR_LAUNCH_CHECK("exp_rowsum_target");
  return FEDFR_OK;
}
int head_nll_mean_ratio(const float* num, const float* den, int R, float floor_, float* loss, hipStream_t st) {
  FEDFR_REQUIRE(num && den && loss && R > 0, "nll_mean_ratio: bad args");
  hipLaunchKernelGGL(nll_mean_ratio_kernel, dim3(1), dim3(256), 0, st, num, den, R, floor_, loss);
  FEDFR_LAUNCH_CHECK("nll_mean_ratio");
  return FEDFR_OK;
}
int head_softmax_grad(float* z, const long long* label, int R, int C, int ldz, const float* row_sum, const float* dmul, float s,
                      float inv_batch, float* prob_t, hipStream_t st) {
  FEDFR_REQUIRE(z && label && row_sum && dmul && prob_t && R > 0 && C > 0, "softmax_grad: bad args");
  hipLaunchKernelGGL(softmax_grad_kernel, dim3(R), dim3(256), 0, st, z, label, C, ldz, row_sum, dmul, s, inv_batch, prob_t);
  FEDFR_LAUNCH_CHECK("softmax_grad");
  return FEDFR_OK;
}
int head_nll_mean(const float* prob_t, int R, float floor_, float* loss, hipStream_t st) {
  FEDFR_REQUIRE(prob_t && loss && R > 0, "nll_mean: bad args");
  hipLaunchKernelGGL(nll_mean_kernel, dim3(1), dim3(256), 0, st, prob_t, R, floor_, loss);
  FEDFR_LAUNCH_CHECK("nll_mean");
  return FEDFR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// BCE personalised head (client.py:45-58, losses.py:4-15), two elementwise kernels:
//   bce_logits: cos [B][C] -> z = r*(g(cos) -/+ m) + bias, gt[b][c] = (label[b] == c), dzdcos = r*t*((cos+1)/2)^(t-1)
//               with g(x) = 2((x+1)/2)^t - 1
//   bce_loss:   z, gt -> row_loss[b] = sum_c (gt ? (lam/r) log(1+e^-z+1e-8) : ((1-lam)/r) log(1+e^z+1e-8)),
//               dz = dL/dz for L = loss_scale * mean_b row_loss, and optionally dcos = dz * dzdcos
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ cosv, const long long* __restrict__ label,
                                                         const float* __restrict__ bias, int C, float m, float r, float t,
                                                         float* __restrict__ z, unsigned char* __restrict__ gt,
                                                         float* __restrict__ dzdcos) {
  const int row = blockIdx.x;
  const long long y = label[row];
  for (int c = threadIdx.x; c < C; c += 256) {
    const size_t i = (size_t)row * C + c;
    const float x = cosv[i];
    const float hb = (x + 1.f) * 0.5f;
    const float pw1 = powf(hb, t - 1.f);
    const float g = 2.f * pw1 * hb - 1.f;
    const bool pos = (c == y);
    z[i] = r * (pos ? g - m : g + m) + bias[c];
    if (gt) gt[i] = pos ? 1 : 0;
    if (dzdcos) dzdcos[i] = r * t * pw1;
  }
}
int head_bce_logits(const float* cosv, const long long* label, const float* bias, int B, int C, float m, float r, float t,
                    float* z, unsigned char* gt, float* dzdcos, hipStream_t st) {
  FEDFR_REQUIRE(cosv && label && bias && z && B > 0 && C > 0, "bce_logits: bad args");
  hipLaunchKernelGGL(bce_logits_kernel, dim3(B), dim3(256), 0, st, cosv, label, bias, C, m, r, t, z, gt, dzdcos);
  FEDFR_LAUNCH_CHECK("bce_logits");
  return FEDFR_OK;
}

__global__ __launch_bounds__(256) void bce_loss_kernel(const float* __restrict__ z, const unsigned char* __restrict__ gt,
                                                       const float* __restrict__ dzdcos, int C, float r, float lam,
                                                       float loss_scale, float inv_batch, float* __restrict__ dz,
                                                       float* __restrict__ dcos, float* __restrict__ row_loss) {
  __shared__ float sh[4];
  const int row = blockIdx.x;
  float ls = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const size_t i = (size_t)row * C + c;
    const float zz = z[i];
    float le, dl;
    if (gt[i]) {
      const float e = expf(-zz);
      le = (lam / r) * logf(1.f + e + 1e-8f);
      dl = (lam / r) * (-e) / (1.f + e + 1e-8f);
    } else {
      const float e = expf(zz);
      le = ((1.f - lam) / r) * logf(1.f + e + 1e-8f);
      dl = ((1.f - lam) / r) * e / (1.f + e + 1e-8f);
    }
    ls += le;
    const float gz = dl * inv_batch * loss_scale;
    if (dz) dz[i] = gz;
    if (dcos) dcos[i] = gz * dzdcos[i];
  }
  ls = block_sum(ls, sh);
  if (threadIdx.x == 0) row_loss[row] = ls;
}
int head_bce_loss(const float* z, const unsigned char* gt, const float* dzdcos, int B, int C, float r, float lam, float loss_scale,
                  float* dz, float* dcos, float* row_loss, hipStream_t st) {
  FEDFR_REQUIRE(z && gt && row_loss && B > 0 && C > 0 && (!dcos || dzdcos), "bce_loss: bad args");
  hipLaunchKernelGGL(bce_loss_kernel, dim3(B), dim3(256), 0, st, z, gt, dzdcos, C, r, lam, loss_scale, 1.f / B, dz, dcos, row_loss);
  FEDFR_LAUNCH_CHECK("bce_loss");
  return FEDFR_OK;
}

// margin backward: dcos = dlogits * (c == label ? dmul[row] : s)     (autograd bridge of losses.CosFace/ArcFace)
__global__ __launch_bounds__(256) void margin_bwd_kernel(const float* __restrict__ dlogits, const long long* __restrict__ label,
                                                         const float* __restrict__ dmul, float s, int C, float* __restrict__ dcos) {
  const int row = blockIdx.x;
  const long long y = label[row];
  const float dm = dmul[row];
  for (int c = threadIdx.x; c < C; c += 256) {
    const size_t i = (size_t)row * C + c;
    dcos[i] = dlogits[i] * (c == y ? dm : s);
  }
}
int head_margin_bwd(const float* dlogits, const long long* label, const float* dmul, float s, int R, int C, float* dcos,
                    hipStream_t st) {
  FEDFR_REQUIRE(dlogits && label && dmul && dcos && R > 0 && C > 0, "margin_bwd: bad args");
  hipLaunchKernelGGL(margin_bwd_kernel, dim3(R), dim3(256), 0, st, dlogits, label, dmul, s, C, dcos);
  FEDFR_LAUNCH_CHECK("margin_bwd");
  return FEDFR_OK;
}

// out[c] = sum_r x[r][c]  (fp32, small R)   and   *out = scale * sum_i x[i]
__global__ void colsum_f32_kernel(const float* __restrict__ x, int R, int C, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int r = 0; r < R; ++r) s += (double)x[(size_t)r * C + c];
  out[c] = (float)s;
}
int head_colsum_f32(const float* x, int R, int C, float* out, hipStream_t st) {
  FEDFR_REQUIRE(x && out && R > 0 && C > 0, "colsum_f32: bad args");
  hipLaunchKernelGGL(colsum_f32_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, st, x, R, C, out);
  FEDFR_LAUNCH_CHECK("colsum_f32");
  return FEDFR_OK;
}
__global__ __launch_bounds__(256) void sum_scale_kernel(const float* __restrict__ x, int n, float scale, float* __restrict__ out) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) *out = s * scale;
}
int head_sum_scale(const float* x, int n, float scale, float* out, hipStream_t st) {
  FEDFR_REQUIRE(x && out && n > 0, "sum_scale: bad args");
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, st, x, n, scale, out);
  FEDFR_LAUNCH_CHECK("sum_scale");
  return FEDFR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// model-contrastive term (reference client.py:372-375 / :415-418): per row, pos = cos(x, g)/T, neg = cos(x, l)/T with
// nn.CosineSimilarity(dim=1, eps=1e-8) (each norm clamped from below by eps), loss_row = CE([pos, neg], label 0)
// = logsumexp(pos, neg) - pos; dx = d(mean_b loss_row)/dx.  One wave per row.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void contrastive_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                          const float* __restrict__ l, int B, int D, float inv_t,
                                                          float* __restrict__ row_loss, float* __restrict__ dx) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  const size_t o = (size_t)row * D;
  float xx = 0.f, gg = 0.f, ll = 0.f, xg = 0.f, xl = 0.f;
  for (int i = lane; i < D; i += 64) {
    const float a = x[o + i], b = g[o + i], c = l[o + i];
    xx += a * a; gg += b * b; ll += c * c; xg += a * b; xl += a * c;
  }
  xx = wave_sum(xx); gg = wave_sum(gg); ll = wave_sum(ll); xg = wave_sum(xg); xl = wave_sum(xl);
  const float eps = 1e-8f;
  const float nx = fmaxf(sqrtf(xx), eps), ng = fmaxf(sqrtf(gg), eps), nl = fmaxf(sqrtf(ll), eps);
  const float cg = xg / (nx * ng), cl = xl / (nx * nl);
  const float pos = cg * inv_t, neg = cl * inv_t;
  const float mx = fmaxf(pos, neg);
  const float lse = mx + logf(expf(pos - mx) + expf(neg - mx));
  if (lane == 0) row_loss[row] = lse - pos;
  if (dx) {
    const float p1 = expf(neg - lse);                  // softmax prob of the negative pair
    const float w = p1 * inv_t / (float)B;             // d mean-loss / d neg-cos = +w ; / d pos-cos = -w
    const float kg = -w / (nx * ng), kl = w / (nx * nl), kx = (w * cg - w * cl) / (nx * nx);
    for (int i = lane; i < D; i += 64) dx[o + i] = kg * g[o + i] + kl * l[o + i] + kx * x[o + i];
  }
}
int head_contrastive(const float* x, const float* g, const float* l, int B, int D, float temperature, float* row_loss, float* dx,
                     hipStream_t st) {
  FEDFR_REQUIRE(x && g && l && row_loss && B > 0 && D > 0 && temperature > 0.f, "contrastive: bad args");
  hipLaunchKernelGGL(contrastive_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, st, x, g, l, B, D, 1.f / temperature, row_loss, dx);
  FEDFR_LAUNCH_CHECK("contrastive");
  return FEDFR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// pairwise-similarity ROC histogram (reference roc_cuda.py:14-30 calc_ROC, the reference's only hand-written GPU
// kernel: one thread per pair, a 512-long scalar dot product and fp64 global atomics).  Here: 64 x 64 pair tiles,
// v_mfma_f64_16x16x4_f64 on the fp32 features widened to fp64 (the reference accumulates in float64 too, so the bin
// index int((dot + 1) * 1000) agrees to the last pair), an LDS-private 4002-counter histogram per workgroup, flushed with
// integer atomics (order-free -> deterministic).  Pairs (a, b): a < b, a < T (the T target rows come first), b < N.
// hist[2 * bin] counts same-label pairs, hist[2 * bin + 1] different-label pairs.
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) double f64x4_t;
__global__ __launch_bounds__(256) void roc_hist_kernel(const float* __restrict__ feat, const long long* __restrict__ label, int N, int D,
                                                       int T, unsigned long long* __restrict__ hist) {
  constexpr int BK = 16, LD = 80, NBIN = 4002;
  __shared__ float sA[2][BK][LD], sB[2][BK][LD];
  __shared__ unsigned lh[NBIN];
  const int a0 = blockIdx.y * 64, b0 = blockIdx.x * 64;
  if (b0 + 63 <= a0) return;                                  // tile entirely on / below the diagonal: no pair with a < b
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  for (int i = tid; i < NBIN; i += 256) lh[i] = 0u;
  float ra[4], rb[4];
  auto load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i, k = e & 15, m = e >> 4;   // features are row-major: k fastest
      const int ga = a0 + m, gb = b0 + m, gk = k0 + k;
      ra[i] = (ga < T && gk < D) ? feat[(size_t)ga * D + gk] : 0.f;
      rb[i] = (gb < N && gk < D) ? feat[(size_t)gb * D + gk] : 0.f;
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i, k = e & 15, m = e >> 4;
      sA[buf][k][m ^ ((k >> 1) << 1)] = ra[i];
      sB[buf][k][m ^ ((k >> 1) << 1)] = rb[i];
    }
  };
  f64x4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f64x4_t){0.0, 0.0, 0.0, 0.0};
  const int nk = ceil_div(D, BK);
  load(0);
  store(0);
  __syncthreads();
  const int l15 = lane & 15, lg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load((kt + 1) * BK);
#pragma unroll
    for (int k4 = 0; k4 < BK; k4 += 4) {
      double fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kk = k4 + lg, sw = (kk >> 1) << 1;
        fa[i] = (double)sA[buf][kk][(wm * 32 + i * 16 + l15) ^ sw];
        fb[i] = (double)sB[buf][kk][(wn * 32 + i * 16 + l15) ^ sw];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store(buf ^ 1);
    __syncthreads();
  }
  // f64 16x16x4 accumulator layout (differs from the f32 form): register q of lane l holds D[row = 4 q + (l >> 4)][col = l & 15]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int b = b0 + wn * 32 + j * 16 + l15;
    const long long lb = b < N ? label[b] : 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int a = a0 + wm * 32 + i * 16 + q * 4 + lg;
        if (a < b && a < T && b < N) {
          int bin = (int)((acc[i][j][q] + 1.0) * 1000.0);           // truncation, as int() in the reference
          bin = bin < 0 ? 0 : (bin > 2000 ? 2000 : bin);             // (the reference would write out of bounds instead)
          atomicAdd(&lh[2 * bin + (label[a] == lb ? 0 : 1)], 1u);
        }
      }
  }
  __syncthreads();
  for (int i = tid; i < NBIN; i += 256)
    if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
}
int head_roc_histogram(const float* feat, const long long* label, int N, int D, int T, unsigned long long* hist, hipStream_t st) {
  FEDFR_REQUIRE(feat && label && hist && N > 0 && D > 0 && T > 0 && T <= N, "roc_histogram: bad args");
  hipLaunchKernelGGL(roc_hist_kernel, dim3(ceil_div(N, 64), ceil_div(T, 64)), dim3(256), 0, st, feat, label, N, D, T, hist);
  FEDFR_LAUNCH_CHECK("roc_histogram");
  return FEDFR_OK;
}
