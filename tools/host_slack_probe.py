#!/usr/bin/env python3
"""Is the host ever on the step's critical path?  A one-step kernel trace shows 50-100 us holes on BOTH queues at the top of the backward pass (the 7x7
stage, right behind the head's Python-side launches).  Under rocprofv3 every launch costs the host more, so the holes may be the tracer's.  This probe
answers without a tracer: the headline step is timed with the host DELAYED by d microseconds between the head and the backward pass (a busy wait in
front of FusedTrainer._backward).  If the host runs ahead of the GPU by more than d, the step time does not move; if it moves by ~d, the GPU was
waiting for the host at that point.  usage: python tools/host_slack_probe.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fedfr_amd import backbones, client

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B, NC = 128, 1000
torch.manual_seed(100)
model = backbones.iresnet100(False, dropout=0, fp16=True).to(dev)
fc = (torch.randn(NC, 512) * 0.01).to(dev)
tr = client.FusedTrainer(model, fc, "CosFace", 30.0, 0.4, lr=1e-3)
g = torch.Generator().manual_seed(100)
imgs = [(torch.rand(B, 3, 112, 112, generator=g) * 2 - 1).to(dev) for _ in range(4)]
labs = [torch.randint(0, NC, (B,), generator=g).to(dev) for _ in range(4)]
lo_p, hi_p = torch.cuda.Stream.priority_range()
hi = torch.cuda.Stream(device=dev, priority=hi_p)
hi.wait_stream(torch.cuda.current_stream())
torch.cuda.set_stream(hi)
orig = tr._backward
delay = [0.0]


def delayed(plan, imgs_, dfeats, st):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e6 < delay[0]:
        pass
    return orig(plan, imgs_, dfeats, st)


tr._backward = delayed
for i in range(8):
    tr.step(imgs[i % 4], labs[i % 4])
torch.cuda.synchronize()
print("# host delayed by d us between the head and the backward pass; %d steps per number, 2 repetitions" % steps)
for rep in range(2):
    for d in (0.0, 100.0, 300.0, 1000.0, 3000.0):
        delay[0] = d
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            tr.step(imgs[i % 4], labs[i % 4])
        te = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("delay %6.0f us: step %.3f ms   (host enqueue %.3f ms per step)" % (d, dt * 1e3 / steps, te * 1e3 / steps))
