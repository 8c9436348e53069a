// Hardware probe (diagnostics, not product): what does a kernel pay for fetching its arguments?
//
// Every kernel of the library takes one descriptor struct by value: its first instructions are s_load's from the kernarg segment (device memory
// since HIP_FORCE_DEV_KERNARG=1 is the runtime's default here; =0 costs the step 0.78 ms, profiles/r05_ab_runtime_env_v1.txt) and nothing can be
// addressed before they return.  gfx950 can PRELOAD the first kernel arguments into SGPRs while the wave is launched
// (-mllvm -amdgpu-kernarg-preload-count=N), but only scalar / pointer arguments in front of the first by-value struct.
// This probe times chains of dependent launches (256 workgroups x 256 threads, one dependent global load + store per thread) of
//   struct    kernel(S s)                         : pointers and sizes inside a 320-byte struct, as the library passes them
//   scalars   kernel(const float* in, float* out, int n, S s)  built with preload: the three leading arguments arrive in SGPRs
// Build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=8 tools/probe/kernarg_probe.hip -o tools/probe/kernarg_probe
#include <hip/hip_runtime.h>
#include <cstdio>

struct S {
  const float* in;
  float* out;
  int n;
  int pad[75];      // (the conv descriptor is ~320 bytes)
};

__global__ __launch_bounds__(256) void k_struct(S s) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < s.n) s.out[i] = s.in[i] + 1.f;
}
__global__ __launch_bounds__(256) void k_scalar(const float* in, float* out, int n, S s) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[i] + (float)(s.pad[3] + 1);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const int n = 256 * 256;
  float *a, *b;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
  CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  S s{};
  s.n = n;
  const int reps = 2000;
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) {
        s.in = (i & 1) ? b : a; s.out = (i & 1) ? a : b;          // each launch reads what the previous one wrote
        if (mode == 0) hipLaunchKernelGGL(k_struct, dim3(256), dim3(256), 0, st, s);
        else hipLaunchKernelGGL(k_scalar, dim3(256), dim3(256), 0, st, s.in, s.out, n, s);
      }
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("%-8s %7.3f us per dependent launch\n", mode == 0 ? "struct" : "scalars", ms * 1e3f / reps);
    }
  return 0;
}
