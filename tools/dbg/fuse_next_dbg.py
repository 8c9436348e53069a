import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from fedfr_amd import _C
import test_e2e_gpu as T
R = T.R
DEV = T.DEV
def run(sliced, fuse):
    _C.call("fedfr_set_option", b"bn_sliced", sliced)
    _C.call("fedfr_set_option", b"fuse_bnred_next", fuse)
    m, sd, _ = T.make_model("iresnet18", tag=3.0)
    m.train()
    f = m(R.closed_form_images(64).to(DEV))
    (f * R.closed_form((64, 512), 0.37, 0.9, 1.0).to(DEV)).sum().backward()
    torch.cuda.synchronize()
    return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
ref = run(0, 1)
for sl, fu in ((0, 0), (1, 1), (1, 0)):
    o = run(sl, fu)
    errs = sorted(((T.rel(o[k], ref[k]), k) for k in ref if float(ref[k].norm()) > 0), reverse=True)
    print("sliced", sl, "fuse", fu, ["%s %.2e" % (k, e) for e, k in errs[:8]])
for k in ("layer1.1.bn3.bias", "layer1.1.bn3.weight", "layer2.1.bn3.bias", "layer2.1.bn3.weight", "layer3.0.downsample.1.bias", "layer3.0.downsample.1.weight"):
    print(k, float(ref[k].norm()), float((o[k] - ref[k]).norm()))
