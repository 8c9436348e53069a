// Hardware probe (diagnostics): semantics of `buffer_load_dwordx4 ... offen lds` on gfx950 —
//  (1) LDS destination = wave-uniform M0 base + lane * 16, (2) what an out-of-range lane writes (expected: zeros).
// Build: hipcc -O2 --offload-arch=gfx950 tools/probe/glds_probe.hip -o tools/probe/glds_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned char* src, unsigned nbytes, unsigned* out, const unsigned* voffs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, (int)nbytes, 0x00020000);
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = 0xdeadbeefu;
  __syncthreads();
  unsigned char* dst = smem + __builtin_amdgcn_readfirstlane(wave) * 1024;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, (int)voffs[threadIdx.x], 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)");
  __syncthreads();
  for (int i = threadIdx.x; i < 4096 / 4; i += blockDim.x) out[i] = reinterpret_cast<unsigned*>(smem)[i];
}
int main() {
  const unsigned nbytes = 8192;
  std::vector<unsigned> h(nbytes / 4);
  for (unsigned i = 0; i < h.size(); ++i) h[i] = i;                       // word i holds i
  std::vector<unsigned> voffs(256);
  for (int t = 0; t < 256; ++t) voffs[t] = ((t * 37) % 512) * 16;        // scattered 16-B source chunks
  voffs[5] = nbytes;          // exactly out of range
  voffs[70] = 0xfffffff0u;    // far out of range
  voffs[130] = nbytes - 8;    // straddles the end
  unsigned char* d; unsigned *o, *v;
  hipMalloc(&d, nbytes); hipMalloc(&o, 4096); hipMalloc(&v, 1024);
  hipMemcpy(d, h.data(), nbytes, hipMemcpyHostToDevice);
  hipMemcpy(v, voffs.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, d, nbytes, o, v);
  std::vector<unsigned> res(1024);
  hipMemcpy(res.data(), o, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 256; ++t) {
    const bool oob = (t == 5 || t == 70);
    for (int w = 0; w < 4; ++w) {
      const unsigned got = res[t * 4 + w];
      unsigned exp = oob ? 0u : voffs[t] / 4 + w;
      if (t == 130) exp = (w < 2) ? voffs[t] / 4 + w : 0u;
      if (got != exp) { if (bad < 12) printf("lane %d word %d: got %08x expected %08x\n", t, w, got, exp); ++bad; }
    }
  }
  printf("glds probe: %s (%d mismatches); oob lane 5 wrote %08x %08x, lane 130 (straddle) %08x %08x %08x %08x\n", bad ? "MISMATCH" : "OK", bad,
         res[20], res[21], res[520], res[521], res[522], res[523]);
  return bad != 0;
}
