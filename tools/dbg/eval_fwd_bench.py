import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fedfr_amd import backbones, _C
dev = torch.device("cuda:0")
m = backbones.iresnet100(False, dropout=0, fp16=True).to(dev).eval()
x = (torch.rand(128, 3, 112, 112) * 2 - 1).to(dev)
for opt in (0, 1, 0, 1):
    _C.call("fedfr_set_option", b"fuse_bnapply", opt)
    with torch.no_grad():
        for _ in range(3): m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("fuse_bnapply=%d eval forward B=128: %.3f ms" % (opt, dt * 1e3))
