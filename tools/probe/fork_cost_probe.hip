// Hardware probe (diagnostics, not product): what does a cross-stream FORK cost the stream that releases it?
//
// The dual-stream backward pass releases a block's weight gradients to the auxiliary stream with hipEventRecord(main) + hipStreamWaitEvent(aux):
// the kernel trace shows ~5 us of idle main stream behind every record (49 per step).  This probe times chains of
//     main:  A (spins 20 us)  [fork]  C (spins 20 us)        aux:  B (spins 10 us), released by the fork
// per iteration, for four fork forms:
//   none     no fork, no B                                        -> the floor: two dependent launches
//   event    hipEventRecord(e, main); hipStreamWaitEvent(aux, e)  (events with hipEventDisableTiming | hipEventDisableSystemFence, as net.hip)
//   write    hipStreamWriteValue32(main, flag, i); hipStreamWaitValue32(aux, flag, i, GEQ)
//   kernel   A's LAST workgroup stores the flag itself (agent-scope counter + system-scope store); hipStreamWaitValue32(aux, ...): NOTHING is
//            enqueued on main between A and C
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe/fork_cost_probe.hip -o tools/probe/fork_cost_probe      (run under `timeout`: a wait that is never
// satisfied would hang the stream)
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void spin_kernel(long long ticks, unsigned* counter, unsigned* flag, unsigned value) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (flag && threadIdx.x == 0) {
    __threadfence();
    const unsigned old = atomicAdd(counter, 1u);
    if (old == gridDim.x - 1) {
      *counter = 0;
      __threadfence_system();
      __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t mainS, auxS;
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&mainS, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithPriority(&auxS, hipStreamNonBlocking, lo));
  unsigned *flag, *counter;
  CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
  CK(hipMalloc(&counter, 64));
  CK(hipMemset(counter, 0, 64));
  *flag = 0;
  const long long hz = 100000000ll;
  const long long tA = hz * 20 / 1000000, tB = hz * 10 / 1000000;
  const int N = 300;
  hipEvent_t ev[N];
  for (int i = 0; i < N; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming | hipEventDisableSystemFence));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned epoch = 0;
  for (int mode = 0; mode < (can ? 4 : 2); ++mode) {
    for (int rep = 0; rep < 2; ++rep) {                    // rep 0: warm-up
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, mainS));
      for (int i = 0; i < N; ++i) {
        ++epoch;
        if (mode == 3) hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, mainS, tA, counter, flag, epoch);
        else hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, mainS, tA, (unsigned*)nullptr, (unsigned*)nullptr, 0u);
        if (mode == 1) { CK(hipEventRecord(ev[i], mainS)); CK(hipStreamWaitEvent(auxS, ev[i], 0)); }
        if (mode == 2) { CK(hipStreamWriteValue32(mainS, flag, epoch, 0)); CK(hipStreamWaitValue32(auxS, flag, epoch, hipStreamWaitValueGte, 0xffffffffu)); }
        if (mode == 3) CK(hipStreamWaitValue32(auxS, flag, epoch, hipStreamWaitValueGte, 0xffffffffu));
        if (mode != 0) hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, auxS, tB, (unsigned*)nullptr, (unsigned*)nullptr, 0u);
        hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, mainS, tA, (unsigned*)nullptr, (unsigned*)nullptr, 0u);
      }
      CK(hipEventRecord(e1, mainS));
      CK(hipStreamSynchronize(mainS));
      CK(hipStreamSynchronize(auxS));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("%-7s main-stream time per iteration (A + fork + C): %7.2f us\n", mode == 0 ? "none" : mode == 1 ? "event" : mode == 2 ? "write" : "kernel", ms * 1e3f / N);
    }
  }
  return 0;
}
