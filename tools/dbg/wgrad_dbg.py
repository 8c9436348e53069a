import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from fedfr_amd import _C
dev = torch.device("cuda:0")
def run(B, H, Cin, Cout, k, s, glds):
    _C.call("fedfr_set_option", b"tn_glds", glds)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Cin, H, H, generator=g).bfloat16().float()
    Ho = H // s
    dy = torch.randn(B, Cout, Ho, Ho, generator=g).bfloat16().float()
    w = torch.zeros(Cout, Cin, k, k, requires_grad=True)
    F.conv2d(x, w, None, s, 1 if k == 3 else 0).backward(dy)
    ref = w.grad.permute(0, 2, 3, 1).contiguous()
    xd = x.permute(0, 2, 3, 1).contiguous().bfloat16().to(dev); dyd = dy.permute(0, 2, 3, 1).contiguous().bfloat16().to(dev)
    dw = torch.full((Cout, k, k, Cin), float("nan"), device=dev)
    nb = _C.lib().fedfr_conv2d_wgrad_ws_bytes(B, H, Cin, Cout, k, s)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    _C.call("fedfr_conv2d_wgrad", xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, B, H, Cin, Cout, k, s, _C.stream())
    torch.cuda.synchronize()
    d = (dw.cpu() - ref)
    err = float(d.norm() / ref.norm())
    print("B%d H%d %d->%d k%d s%d glds=%d ws=%d relerr %.3e nan %d" % (B, H, Cin, Cout, k, s, glds, nb, err, int(torch.isnan(dw).sum())))
    if err > 1e-3:
        e = d.abs().reshape(Cout, k * k, Cin)
        print("  err by tap:", [round(float(e[:, t].mean()), 3) for t in range(k * k)], " ref mean abs", float(ref.abs().mean()))
        print("  err by co block of 16:", [round(float(e[c:c+16].mean()), 2) for c in range(0, min(Cout, 128), 16)])
        print("  err by ci block of 16:", [round(float(e[:, :, c:c+16].mean()), 2) for c in range(0, min(Cin, 128), 16)])
for case in [(3, 14, 128, 256, 1, 2), (2, 28, 128, 128, 3, 1), (1, 14, 256, 256, 3, 1), (40, 14, 256, 256, 3, 1)]:
    for g in (0, 1):
        run(*case, g)
