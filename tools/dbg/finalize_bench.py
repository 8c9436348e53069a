import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fedfr_amd import _C
dev = torch.device("cuda:0")
def timeit(fn, iters=200):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for P, C in [(392, 256), (1568, 128), (6272, 64), (98, 512)]:
    part = torch.rand(P, 2, C, device=dev)
    v = [torch.rand(C, device=dev) + 0.5 for _ in range(10)]
    tmp = torch.empty(64 * 2 * 512 + 4096, device=dev)
    st = _C.stream()
    t = timeit(lambda: _C.call("fedfr_bn_finalize", part.data_ptr(), P, C, float(P * 64), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(),
                               0.1, 1e-5, v[4].data_ptr(), v[5].data_ptr(), v[6].data_ptr(), v[7].data_ptr(), tmp.data_ptr(), st))
    # empty-ish kernel for reference: sum_scale on 1 element
    t0 = timeit(lambda: _C.call("fedfr_sum_scale", v[0].data_ptr(), 1, 1.0, v[9].data_ptr(), st))
    print("bn_finalize P=%5d C=%3d: %.2f us per call (back-to-back launches; trivial kernel %.2f us)" % (P, C, t, t0))
