"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The fp32 oracle (ref_cpu.py) with bf16 *storage* rounding inserted at exactly the points where the HIP
pipeline stores a bf16 tensor (conv inputs/outputs, BN/PReLU outputs, block outputs, bf16 weight copies);
all arithmetic stays fp32 (MFMA accumulates in fp32, BN statistics / normalisation are fp32).  It separates
"bf16 storage noise" (inherent to the north_star's bf16 configs, bounded by the 1e-2 tolerance against the
fp32 reference) from kernel bugs: the HIP path must agree with THIS model to accumulation-order accuracy.
Rounding uses a straight-through gradient so autograd yields the matching backward.
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.nn.functional as F

from . import ref_cpu as R

STORAGE = torch.bfloat16      # the 16-bit storage type being modelled; the GPU tests set it to the loaded library's (set_storage): bfloat16 for the
                              # product build, float16 for the fp16-storage build (same kernels, 10 mantissa bits)


def set_storage(dtype: torch.dtype) -> None:
    global STORAGE
    assert dtype in (torch.bfloat16, torch.float16)
    STORAGE = dtype


def q(x: torch.Tensor) -> torch.Tensor:
    """round to bf16 storage, identity gradient."""
    return x + (x.detach().to(STORAGE).float() - x.detach())


class _RoundGrad(torch.autograd.Function):
    """identity forward; the gradient flowing back through this point is rounded to bf16 — where the HIP backward stores a bf16 tensor."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(STORAGE).float()


def qg(x: torch.Tensor) -> torch.Tensor:
    return _RoundGrad.apply(x)


def _bn(sd, prefix, x, training):
    return R._bn(sd, prefix, x, training)


def block(sd: Dict[str, torch.Tensor], p: str, x: torch.Tensor, stride: int, training: bool, grad_round: bool = False) -> torch.Tensor:
    """``grad_round``: also round the gradients at the points where the HIP backward stores them in bf16 (the gradient wrt the block
    input, bn1 / bn2 outputs, conv outputs and the compact down-sample-path gradient: net.hip gin, da1, dc1, da2, dc2, dd, dxd)."""
    r = qg if grad_round else (lambda t: t)
    x = r(x)
    a1 = r(q(_bn(sd, p + ".bn1", x, training)))
    c1 = r(q(F.conv2d(a1, q(sd[p + ".conv1.weight"]), None, 1, 1)))
    a2 = r(q(F.prelu(_bn(sd, p + ".bn2", c1, training), sd[p + ".prelu.weight"])))
    c2 = r(q(F.conv2d(a2, q(sd[p + ".conv2.weight"]), None, stride, 1)))
    out = _bn(sd, p + ".bn3", c2, training)
    if (p + ".downsample.0.weight") in sd:
        d = r(q(F.conv2d(r(x), q(sd[p + ".downsample.0.weight"]), None, stride, 0)))
        idn = _bn(sd, p + ".downsample.1", d, training)
    else:
        idn = x
    return q(out + idn)


def iresnet_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, layers: Sequence[int], training: bool = True,
                    return_taps: bool = False):
    taps = {}
    c0 = q(F.conv2d(q(x), q(sd["conv1.weight"]), None, 1, 1))
    h = q(F.prelu(_bn(sd, "bn1", c0, training), sd["prelu.weight"]))
    taps["stem"] = h
    for si, nblk in enumerate(layers):
        for bi in range(nblk):
            h = block(sd, "layer%d.%d" % (si + 1, bi), h, 2 if bi == 0 else 1, training)
        taps["layer%d" % (si + 1)] = h
    t = q(_bn(sd, "bn2", h, training))
    y = F.linear(torch.flatten(t, 1), q(sd["fc.weight"]), sd["fc.bias"])
    f = _bn(sd, "features", y, training)
    return (f, taps) if return_taps else f


def train_step_grads(sd, fc, imgs, labels, layers, loss_name="CosFace", s=30.0, m=0.4):
    """Same contract as ref_cpu.train_step_grads, through the bf16-storage model."""
    keys = R.trainable_keys(sd)
    ps = [sd[k].requires_grad_(True) for k in keys]
    fc.requires_grad_(True)
    for p in ps + [fc]:
        p.grad = None
    feats = iresnet_forward(sd, imgs, layers, training=True)
    cosine = R.fc_module_forward(feats, fc)
    logits = R.MARGINS[loss_name](cosine.clone(), labels, s, m)
    loss = F.cross_entropy(logits, labels)
    loss.backward()
    grads = {k: sd[k].grad.detach().clone() for k in keys}
    fcg = fc.grad.detach().clone()
    for p in ps + [fc]:
        p.requires_grad_(False)
        p.grad = None
    return feats.detach(), cosine.detach(), float(loss.detach()), grads, fcg
