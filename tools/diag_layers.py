#!/usr/bin/env python3
"""Diagnostic (GPU box): compare every saved activation of the HIP forward with the bf16-storage emulating oracle,
layer by layer, to localise discrepancies."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from oracle import ref_cpu as R, bf16_emul as E
from fedfr_amd import backbones, _C
DEV = torch.device("cuda:0")
arch = sys.argv[1] if len(sys.argv) > 1 else "iresnet18"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
training = (sys.argv[3] == "train") if len(sys.argv) > 3 else False
layers = R.IRESNET_LAYERS[arch]
sd = R.closed_form_state_dict(layers)
m = getattr(backbones, arch)().to(DEV); m.load_state_dict(sd)
x = R.closed_form_images(B)
m.train(training)
with torch.no_grad():
    f = m._run_forward(x.to(DEV), training=training)
torch.cuda.synchronize()
plan = m._plan(B)
def act(block, which):
    off, rows, ch = C.c_longlong(), C.c_int(), C.c_int()
    _C.call("fedfr_net_act_info", plan.handle, block, which, C.byref(off), C.byref(rows), C.byref(ch))
    if off.value < 0: return None
    a = plan.act[off.value * 2: (off.value + rows.value * ch.value) * 2].view(torch.bfloat16).view(rows.value, ch.value)
    return a.float().cpu()
def nhwc(t): return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
def rel(a, b): return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
def mx(a, b): return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
# emulated forward, capturing the same tensors
sdc = {k: v.clone() for k, v in sd.items()}
q = E.q
with torch.no_grad():
    c0 = q(F.conv2d(q(x), q(sdc["conv1.weight"]), None, 1, 1))
    h = q(F.prelu(R._bn(sdc, "bn1", c0, training), sdc["prelu.weight"]))
    print("stem c0   rel %.2e max %.2e" % (rel(act(-1, 0), nhwc(c0)), mx(act(-1, 0), nhwc(c0))))
    print("stem a0   rel %.2e max %.2e" % (rel(act(-1, 1), nhwc(h)), mx(act(-1, 1), nhwc(h))))
    bi_global = 0
    for si, nblk in enumerate(layers):
        for bi in range(nblk):
            p = "layer%d.%d" % (si + 1, bi); stride = 2 if bi == 0 else 1
            # feed the emulator the HIP block input so errors do not accumulate across blocks
            xin_h = act(bi_global, 0)
            a1 = q(R._bn(sdc, p + ".bn1", h, training))
            c1 = q(F.conv2d(a1, q(sdc[p + ".conv1.weight"]), None, 1, 1))
            a2 = q(F.prelu(R._bn(sdc, p + ".bn2", c1, training), sdc[p + ".prelu.weight"]))
            c2 = q(F.conv2d(a2, q(sdc[p + ".conv2.weight"]), None, stride, 1))
            out = R._bn(sdc, p + ".bn3", c2, training)
            d = None
            if (p + ".downsample.0.weight") in sdc:
                d = q(F.conv2d(h, q(sdc[p + ".downsample.0.weight"]), None, stride, 0))
                idn = R._bn(sdc, p + ".downsample.1", d, training)
            else:
                idn = h
            o = q(out + idn)
            row = "%-10s x %.1e a1 %.1e c1 %.1e a2 %.1e c2 %.1e" % (p, rel(xin_h, nhwc(h)), rel(act(bi_global, 1), nhwc(a1)), rel(act(bi_global, 2), nhwc(c1)),
                                                                  rel(act(bi_global, 3), nhwc(a2)), rel(act(bi_global, 4), nhwc(c2)))
            if d is not None: row += " d %.1e" % rel(act(bi_global, 5), nhwc(d))
            row += " out %.1e (max %.1e)" % (rel(act(bi_global, 6), nhwc(o)), mx(act(bi_global, 6), nhwc(o)))
            print(row)
            # continue the emulator from the HIP output: per-block error only
            ho = act(bi_global, 6)
            h = ho.view(o.shape[0], o.shape[2], o.shape[3], o.shape[1]).permute(0, 3, 1, 2).contiguous()
            bi_global += 1
    t = q(R._bn(sdc, "bn2", h, training))
    print("tail t    rel %.2e" % rel(act(-1, 2), torch.flatten(t, 1)))
    y = F.linear(torch.flatten(t, 1), q(sdc["fc.weight"]), sdc["fc.bias"])
    fe = R._bn(sdc, "features", y, training)
    print("feats     rel %.2e (emulator continued from HIP activations)" % rel(f.cpu(), fe))
