#!/usr/bin/env python3
"""Channel-sliced BatchNorm forward pass (bn_sliced.hip) against the number of partial rows it has to reduce itself, in a dependent chain of
launches on one stream: what would fewer rows per producer buy?  usage: python tools/bn_rows_bench.py [M C]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fedfr_amd import _C
d = torch.device("cuda:0")
M, C = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (25088, 256)
x = torch.randn(M, C, device=d).to(torch.bfloat16); y = torch.empty_like(x); x2 = torch.randn(M, C, device=d).to(torch.bfloat16)
g, b, rm, rv = torch.ones(C, device=d), torch.zeros(C, device=d), torch.zeros(C, device=d), torch.ones(C, device=d)
sc, sh, mu, rs = (torch.empty(C, device=d) for _ in range(4))
al = torch.full((C,), 0.25, device=d)
rows_out = _C.lib().fedfr_bn_sliced_rows(M, C, 0)
stats = torch.empty(rows_out, 2, C, device=d)
for rows_in in (32, 64, 128, 256):
    if not _C.lib().fedfr_bn_sliced_ok(M, C, rows_in, 0): continue
    part = torch.rand(rows_in, 2, C, device=d) + 1.0
    for name, a_, x2_, st_ in (("plain", None, None, None), ("prelu+stats", al, None, stats), ("add+stats", None, x2, stats)):
        def run():
            _C.call("fedfr_bn_apply_sliced", part.data_ptr(), rows_in, float(M), g.data_ptr(), b.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5,
                    sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), x.data_ptr(), _C.ptr(a_), _C.ptr(x2_), y.data_ptr(), M, C, _C.ptr(st_), _C.stream())
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): run()
        e1.record(); torch.cuda.synchronize()
        print("M %d C %d rows_in %3d %-12s %6.2f us per launch" % (M, C, rows_in, name, e0.elapsed_time(e1) * 5))
