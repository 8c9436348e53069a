// fedfr_amd — shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// wave = 64 lanes; MFMA 16x16x32 bf16; LDS 160 KiB/CU.
#pragma once
#include <atomic>
#include <mutex>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

// 16-bit storage type of activations, activation gradients and MFMA weight operands.  Default (libfedfr_hip.so): IEEE fp16 — the reference's own
// AMP type (backbones/iresnet.py:159), 10 mantissa bits: whole-network embeddings / logits / gradients inside north_star's 1e-2 of the fp32
// reference; its range needs a loss scale for the gradients, which the host side applies and guards (fedfr_sgd_step_scaled).  -DFEDFR_FP16=0
// (`make bf16` -> libfedfr_hip_bf16.so) builds the SAME kernels on bf16 storage: 7 mantissa bits, no loss scale, ~1 % faster, 1.5-2.5e-2 on
// whole-network outputs.  The names below keep "bf16": they denote "the 16-bit storage type".
#ifndef FEDFR_FP16
#define FEDFR_FP16 1
#endif
typedef unsigned short bf16_t;   // raw 16-bit storage bits in memory
#if FEDFR_FP16
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8_t;
#else
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
#endif
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

#define FEDFR_OK 0
#define FEDFR_ERR_ARG (-1)
#define FEDFR_ERR_HIP (-2)
#define FEDFR_ERR_WORKSPACE (-3)
#define FEDFR_ERR_UNSUPPORTED (-4)

// ---- error plumbing (never throw across the C ABI) ---------------------------------
void fedfr_set_error(const char* fmt, ...);
int fedfr_check_launch(const char* what);

#define FEDFR_REQUIRE(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      fedfr_set_error(__VA_ARGS__);              \
      return FEDFR_ERR_ARG;                      \
    }                                            \
  } while (0)

#define FEDFR_LAUNCH_CHECK(what)                 \
  do {                                           \
    int _rc = fedfr_check_launch(what);          \
    if (_rc != FEDFR_OK) return _rc;             \
  } while (0)

#define FEDFR_TRY(expr)                          \
  do {                                           \
    int _rc = (expr);                            \
    if (_rc != FEDFR_OK) return _rc;             \
  } while (0)

// ---- 16-bit storage <-> f32 ---------------------------------------------------------------------------
#if FEDFR_FP16
__device__ __forceinline__ float bf2f(bf16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }      // RNE (v_cvt_f16_f32), saturates to inf
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = bf2f((bf16_t)(v.x & 0xffffu)); f[1] = bf2f((bf16_t)(v.x >> 16));
  f[2] = bf2f((bf16_t)(v.y & 0xffffu)); f[3] = bf2f((bf16_t)(v.y >> 16));
  f[4] = bf2f((bf16_t)(v.z & 0xffffu)); f[5] = bf2f((bf16_t)(v.z >> 16));
  f[6] = bf2f((bf16_t)(v.w & 0xffffu)); f[7] = bf2f((bf16_t)(v.w >> 16));
}
#else
__device__ __forceinline__ float bf2f(bf16_t u) {
  return __builtin_bit_cast(float, (unsigned)u << 16);
}
__device__ __forceinline__ bf16_t f2bf(float f) {      // RNE, NaN-preserving (v_cvt_pk_bf16_f32)
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __builtin_bit_cast(float, v.x << 16); f[1] = __builtin_bit_cast(float, v.x & 0xffff0000u);
  f[2] = __builtin_bit_cast(float, v.y << 16); f[3] = __builtin_bit_cast(float, v.y & 0xffff0000u);
  f[4] = __builtin_bit_cast(float, v.z << 16); f[5] = __builtin_bit_cast(float, v.z & 0xffff0000u);
  f[6] = __builtin_bit_cast(float, v.w << 16); f[7] = __builtin_bit_cast(float, v.w & 0xffff0000u);
}
#endif
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack_bf2(f[0], f[1]); v.y = pack_bf2(f[2], f[3]);
  v.z = pack_bf2(f[4], f[5]); v.w = pack_bf2(f[6], f[7]);
  return v;
}

// ---- 16-byte global store of an OUTPUT tile that writes through the L2 (sc0 sc1): the kernel boundary then finds no dirty lines to write back
// before the dependent launch behind it may start (tools/probe/launch_tail_probe.hip: 12.8 MB stored per launch -> 18.9 vs 18.1 us per
// dependent launch).  Used by the 3x3 LDS-DMA convs' epilogue (FEDFR_STORE_WT = 1; step -0.04 / -0.08 ms same-box on two boxes); the BatchNorm
// passes LOSE with it (+0.22 ms: profiles/r05_ab_store_wt_conv_and_bn_v2.txt) and keep plain stores.
// The s_nop 1: on gfx940+ a VMEM store of more than 64 bits needs TWO wait states before a VALU write of its data registers, and the compiler's
// hazard pass does not look inside inline asm (without any the two-tile 28x28 dgrad stored garbage; with one wait state a BatchNorm pass did).
#ifndef FEDFR_STORE_WT
#define FEDFR_STORE_WT 1
#endif
__device__ __forceinline__ void st16_out(void* dst, const uint4& v) {
#if FEDFR_STORE_WT
  typedef __attribute__((ext_vector_type(4))) unsigned st16_u4_t;
  const st16_u4_t vv = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(vv) : "memory");
#else
  *reinterpret_cast<uint4*>(dst) = v;
#endif
}

// ---- wave / block reductions (64-lane wavefront) -----------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// exact n / d for n*d < 2^40 via one 64-bit multiply: magic = ceil(2^40 / d)
struct FastDiv {
  unsigned long long magic;
  unsigned d;
};
static inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  f.d = d;
  f.magic = ((1ull << 40) + d - 1) / d;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
  return (unsigned)(((unsigned long long)n * f.magic) >> 40);
}

// bijective XCD remap (blocks b and b+8 share an XCD): gives every XCD a contiguous range of logical ids
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = id & 7, loc = id >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
}

// "once per device" guard of a call site that configures a kernel (hipFuncSetAttribute state lives per device; one process may drive several
// GPUs: a Server training clients on cuda:0..3, and several host threads: Server.train with parallel_clients, ThreadComm.run).  run(f) calls f
// once per device; the device's bit is published (release) only AFTER f has returned, under a mutex, so a racing thread either waits for the
// first caller's hipFuncSetAttribute to finish or finds the bit set — it can never launch a kernel that needs > 64 KB of dynamic LDS before the
// attribute is in place (the round-2 form set the bit first: ADVICE r2).
struct PerDeviceOnce {
  std::atomic<unsigned long long> done{0};
  std::mutex mu;
  template <class F>
  void run(F&& f) {
    int d = 0;
    (void)hipGetDevice(&d);
    const unsigned long long bit = 1ull << (d & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    std::lock_guard<std::mutex> lock(mu);
    if (done.load(std::memory_order_relaxed) & bit) return;
    f();
    done.fetch_or(bit, std::memory_order_release);
  }
};

__host__ __device__ static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }
