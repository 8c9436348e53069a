import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fedfr_amd import backbones, client, _C
dev = torch.device("cuda:0")
m = backbones.iresnet100(False, dropout=0, fp16=True).to(dev)
fc = (torch.randn(1000, 512) * 0.01).to(dev)
tr = client.FusedTrainer(m, fc, "CosFace", 30.0, 0.4, lr=1e-3)
x = (torch.rand(128, 3, 112, 112) * 2 - 1).to(dev); y = torch.randint(0, 1000, (128,)).to(dev)
for _ in range(5): tr.step(x, y)
torch.cuda.synchronize()
import fedfr_amd.client as C
orig = _C.call
acc = {}
def timed(name, *a):
    t0 = time.perf_counter(); r = orig(name, *a); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
_C.call = timed; C._C.call = timed
N = 10
t0 = time.perf_counter()
for _ in range(N): tr.step(x, y)
host = time.perf_counter() - t0
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("host enqueue time per step %.2f ms, wall per step %.2f ms" % (host / N * 1e3, tot / N * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:6]: print("  %-28s %.3f ms/step (host)" % (k, v / N * 1e3))
