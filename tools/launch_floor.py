#!/usr/bin/env python3
"""Per-launch cost of the short kernels of the BatchNorm chain, measured as dependent chains on one stream (HIP events around N
back-to-back launches): what a tiny kernel (fedfr_sum_scale on 64 floats), a BN finalize (P partial rows x C channels) and a
bn_apply of one 14x14 layer cost per launch, alone and alternating.  Output: one JSON object on stdout."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fedfr_amd import _C

dev = torch.device("cuda:0")
f32 = torch.float32


def timed(fn, n=400, warm=50):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n      # us per call


def main():
    st = _C.stream()
    out = {}
    x = torch.randn(64, device=dev)
    o = torch.zeros(1, device=dev)
    out["tiny_kernel_us"] = timed(lambda: _C.call("fedfr_sum_scale", x.data_ptr(), 64, 1.0, o.data_ptr(), st))
    for (P, Cc, M) in ((128, 256, 25088), (512, 128, 100352), (256, 64, 401408)):
        part = torch.randn(P, 2, Cc, device=dev).abs()
        g, b = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
        rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
        sc, sh, mean, rstd = (torch.empty(Cc, device=dev) for _ in range(4))
        tmp = torch.empty(64 * 2 * 512, device=dev)

        def fin():
            _C.call("fedfr_bn_finalize", part.data_ptr(), P, Cc, float(M), g.data_ptr(), b.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5,
                    sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), rstd.data_ptr(), tmp.data_ptr(), st)
        xa = torch.randn(M, Cc, device=dev).to(torch.bfloat16)
        ya = torch.empty_like(xa)
        rows = _C.lib().fedfr_bn_apply_stat_rows(M, Cc)
        stats = torch.empty(rows, 2, Cc, device=dev)

        def app():
            _C.call("fedfr_bn_apply", xa.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, None, None, None, ya.data_ptr(), M, Cc, 0,
                    stats.data_ptr(), st)

        def both():
            fin()
            app()
        key = "P%d_C%d_M%d" % (P, Cc, M)
        out[key] = {"finalize_us": timed(fin), "bn_apply_us": timed(app, 200, 20), "finalize_then_apply_us": timed(both, 200, 20),
                    "apply_GBps": None}
        out[key]["apply_GBps"] = round(2 * M * Cc * 2 / out[key]["bn_apply_us"] / 1e3, 1)
        # the channel-sliced pass that reduces the P rows itself (one launch instead of the two above), with and without output statistics
        for Pin in (P, 2 * P, 32):
            if not _C.lib().fedfr_bn_sliced_ok(M, Cc, Pin, 0):
                continue
            part_s = torch.randn(Pin, 2, Cc, device=dev).abs()
            srows = _C.lib().fedfr_bn_sliced_rows(M, Cc, 0)
            sstats = torch.empty(srows, 2, Cc, device=dev)

            def sliced(with_stats):
                _C.call("fedfr_bn_apply_sliced", part_s.data_ptr(), Pin, float(M), g.data_ptr(), b.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                        0.1, 1e-5, sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), rstd.data_ptr(), xa.data_ptr(), None, None, ya.data_ptr(),
                        M, Cc, sstats.data_ptr() if with_stats else None, st)
            out[key]["sliced_P%d_us" % Pin] = timed(lambda: sliced(False), 200, 20)
            out[key]["sliced_P%d_stats_us" % Pin] = timed(lambda: sliced(True), 200, 20)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
