#!/usr/bin/env python3
"""Where does the bf16 path's distance to the fp32 reference come from, and can 1e-2 be reached by storing SOME tensors wider?
CPU only (oracle/bf16_emul-style rounding inserted at chosen points of the fp32 restatement), iresnet50 batch 8, train-mode BN, against the
reference-generated golden embeddings (tests/golden/r50_b8.npz).  Round-2 result (relative L2 of the [8, 512] embeddings):

  bf16 everywhere the HIP path stores bf16 ........................ 1.7e-2
  + residual stream (block outputs) in fp16 / fp32 ................. 1.4e-2 / 1.3e-2
  + conv OUTPUTS (c1, c2, d: BatchNorm inputs, never MFMA operands) in fp16 / fp32 ... 1.6e-2 / 1.6e-2
  + tail (bn2 output, fc weight) in fp32 ........................... 1.7e-2   (no gain: the verdict's "cheap fix")
  conv outputs AND residual stream in fp16 ......................... 1.1e-2
  everything but the MFMA operands (a1, a2, weights) in fp32 ....... 1.1e-2   <- floor of bf16 GEMM operands on this 50-layer fixture

=> north_star's 1e-2 on whole-network embeddings is not reachable with bf16 MFMA operands on this fixture, whatever else is widened.
Per block the same model is inside 1e-2 on every forward output (tests/test_oracle_golden.py:test_block_bf16_storage_floor).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from oracle import ref_cpu as R, bf16_emul as E

torch.set_num_threads(8)
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "r50_b8.npz"))
layers = R.IRESNET_LAYERS["iresnet50"]
x = R.closed_form_images(8)
ref = torch.from_numpy(g["feat_train"]).double()
q = E.q


def q16(t):
    return t + (t.detach().to(torch.float16).float() - t.detach())


def ident(t):
    return t


def fwd(sd, x, qout, qtail, qc):
    c0 = qc(F.conv2d(q(x), q(sd["conv1.weight"]), None, 1, 1))
    h = qout(F.prelu(R._bn(sd, "bn1", c0, True), sd["prelu.weight"]))
    for si, nblk in enumerate(layers):
        for bi in range(nblk):
            p, stride = "layer%d.%d" % (si + 1, bi), (2 if bi == 0 else 1)
            a1 = q(R._bn(sd, p + ".bn1", h, True))
            c1 = qc(F.conv2d(a1, q(sd[p + ".conv1.weight"]), None, 1, 1))
            a2 = q(F.prelu(R._bn(sd, p + ".bn2", c1, True), sd[p + ".prelu.weight"]))
            c2 = qc(F.conv2d(a2, q(sd[p + ".conv2.weight"]), None, stride, 1))
            out = R._bn(sd, p + ".bn3", c2, True)
            if (p + ".downsample.0.weight") in sd:
                d = qc(F.conv2d(q(h), q(sd[p + ".downsample.0.weight"]), None, stride, 0))
                idn = R._bn(sd, p + ".downsample.1", d, True)
            else:
                idn = h
            h = qout(out + idn)
    t = qtail(R._bn(sd, "bn2", h, True))
    y = F.linear(torch.flatten(t, 1), qtail(sd["fc.weight"]), sd["fc.bias"])
    return R._bn(sd, "features", y, True)


with torch.no_grad():
    for name, (qo, qt, qc) in {"bf16 everywhere": (q, q, q), "residual stream fp16": (q16, q, q), "residual stream fp32": (ident, q, q),
                               "conv outputs fp16": (q, q, q16), "conv outputs fp32": (q, q, ident), "tail fp32": (q, ident, q),
                               "conv outputs + residual stream fp16": (q16, q, q16),
                               "all but the MFMA operands fp32": (ident, ident, ident)}.items():
        f = fwd(R.closed_form_state_dict(layers), x, qo, qt, qc)
        print("%-40s %.3e" % (name, float((f.double() - ref).norm() / ref.norm())), flush=True)
