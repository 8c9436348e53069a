// Hardware probe (diagnostics, not product): what COLD-from-HBM streaming passes of the BatchNorm kind can reach on MI355X.
//
// The BatchNorm passes of the large maps (51 / 205 MB bf16 tensors written a whole forward pass earlier: cold in every cache) run at
// 2.3-3.8 TB/s in the network.  This probe times the three access shapes with the caches flushed before EVERY launch (a 1 GiB memset),
// over grid size, 16-B loads in flight per thread and load / store cache policy:
//   read2      sum of two tensors               (bn_bwd_reduce: dy, x)
//   r1w1       y = a x + b                      (bn_apply)
//   r2w1       dx = a dy + b x + c              (bn_bwd_apply)
// [M][64]-channel bf16 rows; a thread owns 16 B of a row; a workgroup streams a contiguous slab of rows.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe/hbm_stream_probe.hip -o tools/probe/hbm_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned u4v;

template <bool NT>
__device__ __forceinline__ u4v ld(const u4v* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT>
__device__ __forceinline__ void st(u4v* p, u4v v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

__device__ __forceinline__ u4v mix(u4v a, u4v b) {      // a little arithmetic per element so that nothing is optimised away
  u4v r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = (a[i] & 0xffff0000u) + (b[i] >> 16);
  return r;
}

// MODE 0 read2, 1 r1w1, 2 r2w1.  n16 = 16-byte vectors per tensor; the grid splits them into contiguous slabs
template <int MODE, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void stream(const u4v* __restrict__ a, const u4v* __restrict__ b, u4v* __restrict__ o, unsigned* sink, size_t n16) {
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const size_t beg = (size_t)blockIdx.x * per, end = min(n16, beg + per);
  u4v acc = {0, 0, 0, 0};
  size_t i = beg + threadIdx.x;
  for (; i + (size_t)(U - 1) * 256 < end; i += (size_t)U * 256) {
    u4v va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      va[u] = ld<NTL>(a + i + u * 256);
      if (MODE != 1) vb[u] = ld<NTL>(b + i + u * 256);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const u4v r = MODE == 1 ? mix(va[u], va[u]) : mix(va[u], vb[u]);
      if (MODE == 0) acc = mix(acc, r);
      else st<NTS>(o + i + u * 256, r);
    }
  }
  for (; i < end; i += 256) {
    const u4v r = MODE == 1 ? mix(a[i], a[i]) : mix(a[i], b[i]);
    if (MODE == 0) acc = mix(acc, r);
    else o[i] = r;
  }
  if (MODE == 0 && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

static char* g_flush;
static const size_t kFlush = (size_t)1 << 30;
// flush = a plain-load read-modify-write sweep over 1 GiB (hipMemset's stores left the probe's tensors in the 256 MB Infinity Cache: a
// "cold" read2 of 103 MB took 6.6 us = 15 TB/s)
__global__ __launch_bounds__(256) void flush_kernel(u4v* buf, size_t n16, unsigned k) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    u4v v = buf[i];
    v[0] += k;
    buf[i] = v;
  }
}
template <int MODE, int U, bool NTL, bool NTS>
static double run(int grid, const u4v* a, const u4v* b, u4v* o, unsigned* sink, size_t n16, bool cold) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ts;
  for (int r = 0; r < 7; ++r) {
    if (cold) hipLaunchKernelGGL(flush_kernel, dim3(2048), dim3(256), 0, 0, reinterpret_cast<u4v*>(g_flush), kFlush / 16, (unsigned)r);
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream<MODE, U, NTL, NTS>), dim3(grid), dim3(256), 0, 0, a, b, o, sink, n16);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2] * 1e3;      // median, us
}

template <int MODE>
static void sweep(const char* name, int ntens, const u4v* a, const u4v* b, u4v* o, unsigned* sink, size_t n16) {
  const double mb = (double)n16 * 16 * ntens / 1e6;
  const int grids[] = {256, 512, 1024, 2048, 4096, 8192};
  for (int cold = 1; cold >= 0; --cold) {
    printf("%s  %.0f MB moved  (%s)\n", name, mb, cold ? "COLD: 1 GiB read-modify-write sweep before every launch" : "warm: back to back");
    printf("  grid     U=2 plain      U=4 plain      U=8 plain      U=4 nt-load    U=8 nt-load    U=8 nt-load+nt-store\n");
    for (int g : grids) {
      const double t[6] = {run<MODE, 2, false, false>(g, a, b, o, sink, n16, cold), run<MODE, 4, false, false>(g, a, b, o, sink, n16, cold),
                           run<MODE, 8, false, false>(g, a, b, o, sink, n16, cold), run<MODE, 4, true, false>(g, a, b, o, sink, n16, cold),
                           run<MODE, 8, true, false>(g, a, b, o, sink, n16, cold), run<MODE, 8, true, true>(g, a, b, o, sink, n16, cold)};
      printf("  %5d", g);
      for (double x : t) printf("  %6.1f us %4.2f", x, mb / x);   // MB / us = TB/s
      printf("\n");
    }
  }
}

int main(int argc, char** argv) {
  const int hw = argc > 1 ? atoi(argv[1]) : 56;
  const size_t M = (size_t)128 * hw * hw, n16 = M * 64 * 2 / 16;
  u4v *a, *b, *o; unsigned* sink;
  hipMalloc(&a, n16 * 16); hipMalloc(&b, n16 * 16); hipMalloc(&o, n16 * 16); hipMalloc(&sink, 64);
  hipMalloc(&g_flush, kFlush);
  hipMemset(a, 0x3c, n16 * 16); hipMemset(b, 0x3d, n16 * 16); hipMemset(o, 0, n16 * 16);
  printf("tensor [%zu][64] bf16 = %.1f MB  (columns: median of 5 launches, us and TB/s)\n", M, n16 * 16 / 1e6);
  sweep<0>("read2 (bn_bwd_reduce)", 2, a, b, o, sink, n16);
  sweep<1>("r1w1  (bn_apply)     ", 2, a, b, o, sink, n16);
  sweep<2>("r2w1  (bn_bwd_apply) ", 3, a, b, o, sink, n16);
  return 0;
}
