#!/usr/bin/env python3
"""K independent FL clients training CONCURRENTLY on one GPU (K threads, one HIP stream pair each): the clients of a round are
independent (the reference trains them one after another, server.py:283), and one client's step is a dependent chain of ~1250 short
kernels that leaves CUs idle between launches — a second client's chain fills them.  Prints aggregate images/s for K = 1, 2, (3).

  python tools/multi_client_bench.py [--arch iresnet100] [--batch 128] [--steps 20] [--clients 1,2]
"""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fedfr_amd import backbones, client


def run(arch, B, steps, warmup, K, dev):
    NC = 1000
    lo_p, hi_p = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    ctx = []
    for k in range(K):
        torch.manual_seed(100 + k)
        model = getattr(backbones, arch)(False, dropout=0, fp16=True).to(dev)
        fc = (torch.randn(NC, 512) * 0.01).to(dev)
        st = torch.cuda.Stream(device=dev, priority=hi_p)
        g = torch.Generator().manual_seed(100 + k)
        imgs = [(torch.rand(B, 3, 112, 112, generator=g) * 2 - 1).to(dev) for _ in range(2)]
        labs = [torch.randint(0, NC, (B,), generator=g).to(dev) for _ in range(2)]
        with torch.cuda.stream(st):
            tr = client.FusedTrainer(model, fc, "CosFace", 30.0, 0.4, lr=1e-3, aux_slot=k)
        ctx.append((tr, st, imgs, labs))
    torch.cuda.synchronize()
    bar = threading.Barrier(K + 1)
    losses = [None] * K

    def worker(k):
        tr, st, imgs, labs = ctx[k]
        torch.cuda.set_device(dev)
        with torch.cuda.stream(st):
            for i in range(warmup):
                tr.step(imgs[i % 2], labs[i % 2])
            st.synchronize()
            bar.wait()
            for i in range(steps):
                loss = tr.step(imgs[i % 2], labs[i % 2])
            tr.finish()
            st.synchronize()
            losses[k] = float(loss)
            bar.wait()
    ts = [threading.Thread(target=worker, args=(k,), daemon=True) for k in range(K)]
    for t in ts:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    dt = time.perf_counter() - t0
    for t in ts:
        t.join()
    return {"clients": K, "images_per_sec": round(K * B * steps / dt, 1), "ms_per_step_per_client": round(dt * 1e3 / steps, 3), "final_loss": losses}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="iresnet100")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--clients", default="1,2")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from fedfr_amd import _C
    out = []
    for k in a.clients.split(","):
        if "wgrad9p" not in os.environ.get("FEDFR_OPTIONS", ""):      # what Server.train selects: the paired weight-gradient kernel when clients share the GPU
            _C.call("fedfr_set_option", b"wgrad9p", 1 if int(k) > 1 else 0)
        out.append(run(a.arch, a.batch, a.steps, a.warmup, int(k), dev))
    print(json.dumps({"arch": a.arch, "batch": a.batch, "steps": a.steps, "results": out}))


if __name__ == "__main__":
    main()
