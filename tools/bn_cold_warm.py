#!/usr/bin/env python3
"""Are the BatchNorm passes bandwidth-bound or latency-bound, and is what rocprofv3's FETCH_SIZE / WRITE_SIZE count DRAM traffic?  

FETCH_SIZE / WRITE_SIZE are the L2's fabric-side request counters (MI355X_MICROARCH.md, "HBM"): Infinity-Cache (MALL, 256 MiB) hits are counted, so
"47.9 GB per step" is L2 <-> fabric traffic, an upper bound of the DRAM traffic.  gfx950 exposes no MALL hit / DRAM-side counter through rocprofv3, so the
split is measured by timing: each forward BatchNorm pass of the step's shapes (batch 128) is run

  warm: on ONE (x, y) tensor pair, back to back           -> from the second launch on the operands sit in the Infinity Cache if they fit
  cold: over a RING of distinct pairs larger than 256 MiB  -> every launch reads bytes that were evicted since their last use (DRAM)

as dependent launches on one stream (HIP events around N launches).  In the training step a pass's input was written by the kernel in front of it
(warm if it fits), its other operand (the saved activation of the forward pass, in backward passes) is cold.  Output: one line per shape and kernel.
usage: python tools/bn_cold_warm.py [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fedfr_amd import _C

d = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = 128
T16 = _C.storage_dtype()


def timed(fn, ring):
    for i in range(min(len(ring), 8) + 2):
        fn(ring[i % len(ring)])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(N):
        fn(ring[i % len(ring)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / N


def main():
    st = _C.stream()
    print("# %d dependent launches per number; tensors %s; 'cold' ring > 256 MiB of distinct operands" % (N, str(T16).split(".")[-1]))
    for name, H, C in (("14x14x256", 14, 256), ("28x28x128", 28, 128), ("56x56x64", 56, 64), ("112x112x64", 112, 64)):
        M = B * H * H
        nbytes = M * C * 2
        nring = max(2, int(640e6 // (2 * nbytes)) + 1)
        ring = [(torch.randn(M, C, device=d).to(T16), torch.empty(M, C, device=d, dtype=T16)) for _ in range(nring)]
        g, b, rm, rv = torch.ones(C, device=d), torch.zeros(C, device=d), torch.zeros(C, device=d), torch.ones(C, device=d)
        sc, sh, mu, rs = (torch.rand(C, device=d) + 0.5 for _ in range(4))
        al = torch.full((C,), 0.25, device=d)
        # row-slab pass (ew.hip): y = prelu(x * sc + sh)
        def slab(p):
            _C.call("fedfr_bn_apply", p[0].data_ptr(), sc.data_ptr(), sh.data_ptr(), al.data_ptr(), None, None, None, p[1].data_ptr(), M, C, 0, None, st)
        rows = [("row-slab bn_apply + PReLU", slab)]
        rows_in = {14: 128, 28: 256}.get(H)
        if rows_in and _C.lib().fedfr_bn_sliced_ok(M, C, rows_in, 0):
            part = torch.rand(rows_in, 2, C, device=d) + 1.0
            def sliced(p):
                _C.call("fedfr_bn_apply_sliced", part.data_ptr(), rows_in, float(M), g.data_ptr(), b.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5,
                        sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), p[0].data_ptr(), al.data_ptr(), None, p[1].data_ptr(), M, C, None, st)
            rows.append(("channel-sliced bn_apply_s + PReLU (reduces %d partial rows itself)" % rows_in, sliced))
        for label, fn in rows:
            warm = timed(fn, ring[:1])
            cold = timed(fn, ring)
            print("%-11s %-72s tensor %6.1f MB  warm %6.2f us (%5.0f GB/s)  cold %6.2f us (%5.0f GB/s)  cold/warm %.2f" %
                  (name, label, nbytes / 1e6, warm, 2 * nbytes / warm / 1e3, cold, 2 * nbytes / cold / 1e3, cold / warm))
        del ring
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
