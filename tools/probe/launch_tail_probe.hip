// Hardware probe (diagnostics, not product): what does a dependent launch of a conv-shaped kernel cost OUTSIDE its waves' lifetime, and how much
// of that is the end-of-kernel write-back of the tile it stored?
//
// conv3x3_glds<14,14> runs 31 us per launch while its waves live 21.7 us (SQ counters, profiles/r04_pmc_sq_conv14_fwd_v2.txt): 6-9 us per launch
// are dispatch, ramp and drain.  This probe times chains of dependent launches (one stream, HIP events around N launches) of a stand-in with the
// conv's launch shape — 256 workgroups x 512 threads, 128 KB of dynamic LDS — whose workgroups spin for a fixed time and then store a
// 50 KB tile each (12.8 MB per launch, the 14x14x256 activation), in four variants:
//   none      no stores                                    -> launch + ramp + drain of the shape alone
//   plain     16-B global stores, default policy           -> + stores + whatever the kernel boundary does with the dirty L2 lines
//   nt        __builtin_nontemporal_store                  -> streaming stores
//   sc1       global_store_dwordx4 ... sc0 sc1 (write-through at system scope)
// and with the spin at 0 / 15 us, so that (time - spin) is the fixed cost.  Ping-pong output buffers (a launch never rewrites the lines the
// previous one left dirty).
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe/launch_tail_probe.hip -o tools/probe/launch_tail_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u4v;

template <int MODE>
__global__ __launch_bounds__(512) void tail_kernel(u4v* out, long long spin_ticks, unsigned seed) {
  extern __shared__ unsigned char smem[];
  if (threadIdx.x == 0) smem[0] = (unsigned char)seed;             // (the LDS allocation is real)
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
  if (MODE == 0) return;
  // 50 KB per workgroup = 3200 x 16 B: 512 threads x 6.25 -> 7 rounds, the last partly masked
  u4v* dst = out + (size_t)blockIdx.x * 3200;
  const u4v v = {seed, threadIdx.x, blockIdx.x, 7u};
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const int i = r * 512 + threadIdx.x;
    if (i < 3200) {
      if (MODE == 1) dst[i] = v;
      else if (MODE == 2) __builtin_nontemporal_store(v, dst + i);
      else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst + i), "v"(v) : "memory");
    }
  }
}

template <int MODE>
static float chain(u4v* a, u4v* b, long long ticks, int n) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(tail_kernel<MODE>, dim3(256), dim3(512), 128 * 1024, 0, (i & 1) ? a : b, ticks, (unsigned)i);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tail_kernel<MODE>, dim3(256), dim3(512), 128 * 1024, 0, (i & 1) ? a : b, ticks, (unsigned)i);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms * 1e3f / n;
}

int main() {
  u4v *a, *b;
  const size_t bytes = (size_t)256 * 3200 * 16;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes);
  const long long hz = 100000000ll;                       // wall_clock64: 100 MHz
  const int n = 400;
  for (int spin_us : {0, 15}) {
    const long long ticks = hz * spin_us / 1000000;
    const float t0 = chain<0>(a, b, ticks, n), t1 = chain<1>(a, b, ticks, n), t2 = chain<2>(a, b, ticks, n), t3 = chain<3>(a, b, ticks, n);
    printf("spin %2d us: per dependent launch  none %6.2f us | plain stores %6.2f | nontemporal %6.2f | sc0 sc1 %6.2f   (12.8 MB stored per launch)\n", spin_us, t0, t1, t2,
           t3);
  }
  hipFree(a); hipFree(b);
  return 0;
}
