#!/usr/bin/env python3
"""Phase breakdown of the halo2 3x3 conv kernel from in-kernel clock stamps (diagnostics; build with make -C tools/stamp).
Per workgroup: t0 entry, t1 prologue done (first tiles in LDS), t2 main loop done, t3 epilogue stores issued.
Prints shader-clock cycles per phase, the shader clock rate (s_memtime vs the 100 MHz wall clock) and the wall-clock span."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fedfr_amd import _C

_C.lib()
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "stamp", os.environ.get("STAMP_LIB", "libstamp.so")))
vp = ctypes.c_void_p
lib.stamp_conv3x3.argtypes = [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_int]
WHICH = [int(a) for a in sys.argv[1:]] or [2, 3]
dev = torch.device("cuda:0")
B = 128
for H, Cin, Cout, which in [(h, a, b, w_) for (h, a, b) in [(14, 256, 256), (28, 128, 128)] for w_ in WHICH]:
    x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05).to(torch.bfloat16)
    y = torch.empty(B, H, H, Cout, dtype=torch.bfloat16, device=dev)
    stats = torch.empty(_C.lib().fedfr_conv2d_stat_rows(B, H, Cout), 2, Cout, device=dev)
    nblk = ((B * H * H + 127) // 128) * ((Cout + 127) // 128) if which == 2 else (B * H * H // 196) * (Cout // 128)
    dbg = torch.zeros(nblk, 16, dtype=torch.int64, device=dev)
    for it in range(30):
        rc = lib.stamp_conv3x3(x.data_ptr(), w.data_ptr(), y.data_ptr(), None if os.environ.get("STAMP_NOSTATS") else stats.data_ptr(), B, H, Cin, Cout, dbg.data_ptr(), None, which)
        assert rc == 0
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype(np.int64)
    cyc, wall = d[:, 0:8:2], d[:, 1:8:2]
    span_wall = (wall[:, 3].max() - wall[:, 0].min()) / 100.0          # us (100 MHz)
    dur_c = cyc[:, 3] - cyc[:, 0]
    dur_w = (wall[:, 3] - wall[:, 0]) / 100.0
    ghz = np.median(dur_c / np.maximum(dur_w, 1e-3)) / 1e3
    start = (wall[:, 0] - wall[:, 0].min()) / 100.0
    first = start < 0.5 * np.median(dur_w)
    print("== kernel %s" % {2: "halo2 (register-staged)", 3: "glds (LDS-DMA)"}[which])
    print("== %dx%d C%d->%d: %d workgroups, launch span %.1f us, shader clock ~%.2f GHz" % (H, H, Cin, Cout, nblk, span_wall, ghz))
    for name, a, b in (("prologue", 0, 1), ("main loop", 1, 2), ("epilogue", 2, 3)):
        c = cyc[:, b] - cyc[:, a]
        print("   %-9s median %7d cyc (%.2f us)  p10 %7d  p90 %7d" % (name, np.median(c), np.median(c) / ghz / 1e3, np.percentile(c, 10), np.percentile(c, 90)))
    print("   workgroup total median %.2f us; first-round WGs %d (median %.2f us), later WGs %d (start median %.2f us, dur median %.2f us)"
          % (np.median(dur_w), first.sum(), np.median(dur_w[first]), (~first).sum(), np.median(start[~first]) if (~first).any() else 0,
             np.median(dur_w[~first]) if (~first).any() else 0))
    ideal = (16 if which == 2 else 28) * (9 * Cin // 32) * 16
    print("   ideal MFMA cycles per workgroup (per SIMD): %d -> main loop efficiency %.1f%%" % (ideal, 100.0 * ideal / np.median(cyc[:, 2] - cyc[:, 1])))
