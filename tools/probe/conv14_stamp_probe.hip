// Hardware probe (diagnostics, not product): where does a launch of the 14x14 LDS-DMA conv kernel spend its time?
//
// Compiles conv3x3_glds_kernel<14, 14, 32, 4, false> (fedfr_amd/csrc/conv_glds_impl.h) with its in-kernel phase stamps (FEDFR_HALO2_STAMPS:
// wave 0 of every workgroup stores wall_clock64() — the 100 MHz constant clock, comparable across CUs — at kernel entry, behind the prologue's
// first barrier, behind the K loop and at the end of the epilogue), runs the headline layer (128 images, 256 -> 256 channels) back to back and
// prints, over the 256 workgroups of the LAST launch, the phase durations and the launch's own timeline (first entry -> last exit).
//
// Stands alone (the library's host-side helpers the launcher calls are stubbed below: linking libfedfr_hip.so as well would register the same
// kernel symbol from two code objects).  Build from the repo root:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -DFEDFR_HALO2_STAMPS -Wno-unused-value -Ifedfr_amd/csrc -Iinclude \
//         tools/probe/conv14_stamp_probe.hip -o tools/probe/conv14_stamp_probe            (-DPROBE_W=28 -o tools/probe/conv28_stamp_probe: 28x28, two tiles)
#include "conv_glds_impl.h"
#include <cstdio>
#include <vector>
#include <algorithm>

#include <cstdarg>
void fedfr_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }
int fedfr_check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { printf("%s: %s\n", what, hipGetErrorString(e)); return 1; } return 0; }
double gemm_nt_alg_bytes(const GemmNT&, int) { return 0.0; }
int gemm_nt_stat_rows(int M, int) { return (M + 127) / 128 * 2; }
ProfScope::ProfScope(int, double, hipStream_t st, double) : st_(st) {}
ProfScope::~ProfScope() {}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#ifndef PROBE_W
#define PROBE_W 14
#endif
#ifndef PROBE_FUSED
#define PROBE_FUSED 0      // 1: the dgrad instantiation with the BatchNorm-backward reduction in its epilogue (sums of dy and dy * x; no PReLU), 2: the same with PReLU slopes
#endif

int main(int argc, char** argv) {
  const int imgs = 128, C = argc > 1 ? atoi(argv[1]) : 256, N = C, W = PROBE_W;
  const int M = imgs * W * W, K = 9 * C;
  bf16_t *A, *B, *Cb;
  float* stats;
  unsigned long long* dbg;
  const int srows = gemm_nt_stat_rows(M, N);
  CK(hipMalloc(&A, (size_t)M * C * 2));
  CK(hipMalloc(&B, (size_t)N * K * 2));
  CK(hipMalloc(&Cb, (size_t)M * N * 2));
  CK(hipMalloc(&stats, (size_t)srows * 2 * N * 4));
  const int nwg = 4096;
  CK(hipMalloc(&dbg, (size_t)nwg * 16 * 8));
  // small normal-range fp16 / bf16 patterns (0x2c00.. ~ 0.06 in fp16): the values do not matter, their being finite does
  std::vector<unsigned short> ha((size_t)M * C), hb((size_t)N * K);
  unsigned s = 12345u;
  for (auto& v : ha) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x2c00u + ((s >> 16) & 0x3ffu) + ((s >> 31) << 15)); }
  for (auto& v : hb) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x2800u + ((s >> 16) & 0x3ffu) + ((s >> 31) << 15)); }
  CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(dbg, 0, (size_t)nwg * 16 * 8));
  GemmNT p{};
  p.A = A; p.B = B; p.M = M; p.N = N; p.K = K; p.mode = 1; p.H = W; p.W = W; p.C = C; p.Ho = W; p.Wo = W; p.S = 3; p.stride = 1; p.pad = 1; p.up = 1;
  p.cpt = C / 64; p.Cb = Cb; p.ldc = N; p.stats = stats; p.ksteps_total = K / 64; p.ksteps_per_split = K / 64;
  p.a_bytes = (unsigned)((size_t)M * C * 2); p.b_bytes = (unsigned)((size_t)N * K * 2);
  p.dbg = dbg;
#if PROBE_FUSED
  bf16_t* X;
  float *coef, *bpart;
  CK(hipMalloc(&X, (size_t)M * N * 2));
  CK(hipMemcpy(X, ha.data(), (size_t)M * N * 2, hipMemcpyHostToDevice));     // (C == N: the image pattern serves as the BatchNorm input)
  CK(hipMalloc(&coef, (size_t)5 * N * 4));
  std::vector<float> hc((size_t)5 * N, 0.5f);
  CK(hipMemcpy(coef, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&bpart, (size_t)(M / 196) * 3 * N * 4));
  p.stats = nullptr; p.bx = X; p.bmean = coef; p.brstd = coef + N; p.bgamma = coef + 2 * N; p.bbeta = coef + 3 * N; p.bpart = bpart;
  p.balpha = PROBE_FUSED == 2 ? coef + 4 * N : nullptr;
#endif
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 50;
  auto launch = [&]() {
#if PROBE_W == 14
    return launch_glds<14, 14, 32, 4, PROBE_FUSED != 0>(p, st);
#else
    return launch_glds<28, 7, 40, 4, PROBE_FUSED != 0, 128, false, 2>(p, st);
#endif
  };
  for (int i = 0; i < 10; ++i) if (launch()) { printf("launch refused\n"); return 1; }
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("conv3x3_glds<%d, fused %d, GLDS_MFMA_STATS %d> %d -> %d channels, %d images: %.2f us per launch back to back (HIP events over %d launches)\n", W, PROBE_FUSED, GLDS_MFMA_STATS, C, N, imgs, ms * 1e3f / reps, reps);
  std::vector<unsigned long long> h((size_t)nwg * 16);
  CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
  int used = 0;
  while (used < nwg && h[(size_t)used * 16 + 1]) ++used;
  unsigned long long t_first = ~0ull, t_last = 0;
  std::vector<double> ph[3], ep[4], entry, exit_, mhz, fin;
  for (int b = 0; b < used; ++b) {
    const unsigned long long* r = &h[(size_t)b * 16];
    t_first = std::min(t_first, r[1]);
    t_last = std::max(t_last, r[7]);
  }
  for (int b = 0; b < used; ++b) {
    const unsigned long long* r = &h[(size_t)b * 16];
    for (int i = 0; i < 3; ++i) ph[i].push_back((double)(r[2 * i + 3] - r[2 * i + 1]) * 0.01);
    // epilogue split: loop end (2) -> staged (4) -> statistics rows written (5) -> behind the barrier (6) -> tile stored (3)
    const int order[5] = {2, 4, 5, 6, 3};
    for (int i = 0; i < 4; ++i) ep[i].push_back((double)(r[2 * order[i + 1] + 1] - r[2 * order[i] + 1]) * 0.01);
    fin.push_back((double)(r[7] - r[15]) * 0.01);
    mhz.push_back((double)(r[4] - r[2]) / ((double)(r[5] - r[3]) * 0.01));      // shader-clock ticks per microsecond over the K loop
    entry.push_back((double)(r[1] - t_first) * 0.01);
    exit_.push_back((double)(t_last - r[7]) * 0.01);
  }
  auto stat = [&](const char* name, std::vector<double>& v) {
    std::sort(v.begin(), v.end());
    printf("  %-42s min %6.2f  median %6.2f  p90 %6.2f  max %6.2f us\n", name, v.front(), v[v.size() / 2], v[v.size() * 9 / 10], v.back());
  };
  printf("last launch: %d workgroups stamped, first entry -> last exit %.2f us\n", used, (double)(t_last - t_first) * 0.01);
  stat("entry after the launch's first entry", entry);
  stat("prologue (entry -> first barrier)", ph[0]);
  stat("K loop", ph[1]);
  stat("epilogue", ph[2]);
  stat("  staging (convert, sums, LDS writes)", ep[0]);
  stat("  statistics rows (row sums, stores)", ep[1]);
  stat("  barrier", ep[2]);
  stat("  tile LDS -> global", ep[3]);
  stat("  (of it: behind the tile loop, final rows)", fin);
  stat("exit before the launch's last exit", exit_);
  stat("s_memtime ticks per us over the K loop", mhz);
  return 0;
}
