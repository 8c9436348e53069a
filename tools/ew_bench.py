#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound BatchNorm kernels on the iresnet activation shapes (B=128): bn_apply (fwd) and the
three bn_bwd kernels, HIP-event timed, with algorithmic GB/s (bf16 tensors read + written once)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fedfr_amd import _C
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = 128
SHAPES = [("56x56x64", 56, 64, 10), ("28x28x128", 28, 128, 39), ("14x14x256", 14, 256, 90), ("7x7x512", 7, 512, 9)]   # name, H, C, ~BN sites
def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us
tot = {"apply": 0.0, "apply2": 0.0, "bwd": 0.0}
for name, H, C, cnt in SHAPES:
    M = B * H * H
    x = torch.randn(M, C, device=dev).to(torch.bfloat16)
    x2 = torch.randn(M, C, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x)
    dy = torch.randn(M, C, device=dev).to(torch.bfloat16)
    dx = torch.empty_like(x)
    v = [torch.rand(C, device=dev) + 0.5 for _ in range(8)]
    stats = torch.empty(_C.lib().fedfr_bn_apply_stat_rows(M, C), 2, C, device=dev)
    part = torch.empty(_C.lib().fedfr_bn_bwd_rows(M, C), 3, C, device=dev)
    coef = torch.empty(3, C, device=dev)
    g = [torch.empty(C, device=dev) for _ in range(3)]
    st = _C.stream()
    nbytes = M * C * 2
    t1 = timeit(lambda: _C.call("fedfr_bn_apply", x.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), None, None, None,
                                y.data_ptr(), M, C, 0, None, st))
    t2 = timeit(lambda: _C.call("fedfr_bn_apply", x.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), None, x2.data_ptr(), v[3].data_ptr(), v[4].data_ptr(),
                                y.data_ptr(), M, C, 0, stats.data_ptr(), st))
    t3 = timeit(lambda: _C.call("fedfr_bn_bwd", dy.data_ptr(), x.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(),
                                v[4].data_ptr(), M, C, part.data_ptr(), coef.data_ptr(), g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(),
                                None, None, H, dx.data_ptr(), st))
    print("%-10s M=%7d  bn_apply+prelu %6.1f us %5.0f GB/s | bn_apply+residual+stats %6.1f us %5.0f GB/s | bn_bwd (reduce+finalize+apply) %6.1f us %5.0f GB/s"
          % (name, M, t1, 2 * nbytes / t1 / 1e3, t2, 3 * nbytes / t2 / 1e3, t3, 5 * nbytes / t3 / 1e3))
    tot["apply"] += t1 * cnt; tot["bwd"] += t3 * cnt
print("weighted r100 estimate (ms/step): bn_apply %.2f  bn_bwd %.2f" % (tot["apply"] / 1e3, tot["bwd"] / 1e3))
