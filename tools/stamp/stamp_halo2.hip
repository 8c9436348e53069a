// Diagnostics only (not part of libfedfr_hip.so): the halo2 3x3 conv kernel rebuilt with in-kernel clock stamps, to split a
// launch into prologue / main loop / epilogue per workgroup and to read the shader clock the chip holds under this load.
// Build: make -C tools/stamp ; run: python tools/stamp_halo2.py
#define FEDFR_HALO2_STAMPS 1
#include "conv_halo2_impl.h"
#include "conv_glds_impl.h"

extern "C" int stamp_conv3x3(const void* x, const void* w, void* y, float* stats, int batch, int h, int cin, int cout,
                             unsigned long long* dbg, void* stream, int which) {
  GemmNT p{};
  p.A = (const bf16_t*)x; p.B = (const bf16_t*)w; p.M = batch * h * h; p.N = cout; p.K = 9 * cin;
  p.mode = 1; p.H = h; p.W = h; p.C = cin; p.Ho = h; p.Wo = h; p.S = 3; p.stride = 1; p.pad = 1; p.up = 1;
  p.Cb = (bf16_t*)y; p.ldc = cout; p.stats = stats; p.cpt = cin / 64; p.ksteps_total = p.K / 64;
  p.a_bytes = (unsigned)(2ull * batch * h * h * cin);
  p.b_bytes = (unsigned)(2ull * cout * p.K);
  p.dbg = dbg;
  if (which == 3) {
    if (h == 14) return launch_glds<14, 14, 32, 2, false>(p, (hipStream_t)stream);
    if (h == 28) return launch_glds<28, 7, 36, 2, false>(p, (hipStream_t)stream);
    return -1;
  }
  if (h == 14) return launch_halo2<128, 14, 2, false>(p, (hipStream_t)stream);
  if (h == 28) return launch_halo2<128, 28, 2, false>(p, (hipStream_t)stream);
  return -1;
}
