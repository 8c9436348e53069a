#!/usr/bin/env python3
"""Micro-benchmark of the MFMA conv kernels on the iresnet layer shapes (B=128): fwd / dgrad / wgrad TFLOP/s with
HIP events.  Usage: python tools/conv_bench.py [iters] [shape-filter]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fedfr_amd import _C
dev = torch.device("cuda:0")
T16 = _C.storage_dtype()                      # the loaded library's 16-bit storage type
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
filt = sys.argv[2] if len(sys.argv) > 2 else ""
which = sys.argv[3] if len(sys.argv) > 3 else "fwd,dgrad,wgrad"
for opt in sys.argv[4:]:
    k, v = opt.split("=")
    _C.call("fedfr_set_option", k.encode(), int(v))
B = 128
SHAPES = [  # name, H, Cin, Cout, k, s, count in r100
    ("s1_64x64@112", 112, 64, 64, 3, 1, 1), ("s1_64x64@112s2", 112, 64, 64, 3, 2, 1), ("s1_64x64@56", 56, 64, 64, 3, 1, 4),
    ("s2_64x128@56", 56, 64, 128, 3, 1, 1), ("s2_128x128@56s2", 56, 128, 128, 3, 2, 1), ("s2_128x128@28", 28, 128, 128, 3, 1, 24),
    ("s3_128x256@28", 28, 128, 256, 3, 1, 1), ("s3_256x256@28s2", 28, 256, 256, 3, 2, 1), ("s3_256x256@14", 14, 256, 256, 3, 1, 58),
    ("s4_256x512@14", 14, 256, 512, 3, 1, 1), ("s4_512x512@14s2", 14, 512, 512, 3, 2, 1), ("s4_512x512@7", 7, 512, 512, 3, 1, 4),
    ("ds_64x128@56", 56, 64, 128, 1, 2, 1), ("ds_128x256@28", 28, 128, 256, 1, 2, 1), ("ds_256x512@14", 14, 256, 512, 1, 2, 1),
]
def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
for name, H, Cin, Cout, k, s, cnt in SHAPES:
    if filt and filt not in name: continue
    Ho = H // s
    x = torch.randn(B, H, H, Cin, device=dev).to(T16)
    dy = torch.randn(B, Ho, Ho, Cout, device=dev).to(T16)
    w = (torch.randn(Cout, k, k, Cin, device=dev) * 0.05)
    wb = torch.empty(Cout, k, k, Cin, dtype=T16, device=dev); wdb = torch.empty(Cin, k, k, Cout, dtype=T16, device=dev)
    _C.call("fedfr_weight_shadows", w.data_ptr(), wb.data_ptr(), wdb.data_ptr(), Cout, k, Cin, _C.stream())
    y = torch.empty(B, Ho, Ho, Cout, dtype=T16, device=dev)
    dx = torch.empty(B, H, H, Cin, dtype=T16, device=dev)
    dw = torch.empty(Cout, k, k, Cin, device=dev)
    stats = torch.empty(_C.lib().fedfr_conv2d_stat_rows(B, Ho, Cout), 2, Cout, device=dev)
    nb = _C.lib().fedfr_conv2d_wgrad_ws_bytes(B, H, Cin, Cout, k, s); ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    st = _C.stream()
    flop = 2.0 * B * Ho * Ho * Cout * k * k * Cin
    res = []
    if "fwd" in which:
        t = timeit(lambda: _C.call("fedfr_conv2d_fwd", x.data_ptr(), wb.data_ptr(), y.data_ptr(), stats.data_ptr(), B, H, Cin, Cout, k, s, st)); tot["fwd"] += t * cnt; res.append("fwd %7.1f us %6.0f TF" % (t * 1e3, flop / t / 1e9))
    if "dgrad" in which:
        t = timeit(lambda: _C.call("fedfr_conv2d_dgrad", dy.data_ptr(), wdb.data_ptr(), dx.data_ptr(), B, H, Cin, Cout, k, s, st)); tot["dgrad"] += t * cnt; res.append("dgrad %7.1f us %6.0f TF" % (t * 1e3, flop / t / 1e9))
    if "fdgrad" in which and k == 3 and s == 1:      # dgrad with the fused BN-backward reduction epilogue (PReLU variant)
        import ctypes
        bnx = torch.randn(B, H, H, Cin, device=dev).to(T16)
        cv = [torch.rand(Cin, device=dev) + 0.5 for _ in range(5)]
        part = torch.empty((B * H * H + 127) // 128, 3, Cin, device=dev); rows = ctypes.c_int(0)
        t = timeit(lambda: _C.call("fedfr_conv2d_dgrad_bnbwd", dy.data_ptr(), wdb.data_ptr(), dx.data_ptr(), B, H, Cin, Cout, k, s, bnx.data_ptr(),
                                   cv[0].data_ptr(), cv[1].data_ptr(), cv[2].data_ptr(), cv[3].data_ptr(), cv[4].data_ptr(), part.data_ptr(),
                                   ctypes.byref(rows), st)); res.append("fdgrad %7.1f us (rows %d)" % (t * 1e3, rows.value))
    if "wgrad" in which:
        t = timeit(lambda: _C.call("fedfr_conv2d_wgrad", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, B, H, Cin, Cout, k, s, st)); tot["wgrad"] += t * cnt; res.append("wgrad %7.1f us %6.0f TF" % (t * 1e3, flop / t / 1e9))
    if "wpair" in which and k == 3 and s == 1 and Cin == Cout:      # the block's two same-shape weight gradients in one call (wgrad9p.hip when it applies)
        x2 = torch.randn(B, H, H, Cin, device=dev).to(T16); dy2 = torch.randn(B, Ho, Ho, Cout, device=dev).to(T16)
        dw2 = torch.empty(Cout, k, k, Cin, device=dev); ws2 = torch.empty(max(2 * nb, 16), dtype=torch.uint8, device=dev)
        t = timeit(lambda: _C.call("fedfr_conv2d_wgrad_pair", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), x2.data_ptr(), dy2.data_ptr(), dw2.data_ptr(),
                                   ws2.data_ptr(), 2 * nb, B, H, Cin, Cout, k, s, st)); res.append("wpair %7.1f us %6.0f TF" % (t * 1e3, 2 * flop / t / 1e9))
    print("%-18s x%-2d %s" % (name, cnt, " | ".join(res)))
print("r100 totals (ms): fwd %.2f dgrad %.2f wgrad %.2f" % (tot["fwd"], tot["dgrad"], tot["wgrad"]))
