"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement, in plain fp32 PyTorch ops, of the reference algorithms on the
FedFR per-client training hot path.  Written from the semantics listed in
SURVEY.md App. B; every function cites the reference file:line it follows.
Pinned against the imported reference by tests/golden/*.npz
(tools/make_golden.py), checked in tests/test_oracle_golden.py.

Everything is *functional*: a network is just an ordered ``dict`` of tensors with
the reference's state_dict key names, so the same code drives parity checks,
golden generation and the timed CPU baseline.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# network description  (reference: backbones/iresnet.py:182-204 factories)
# --------------------------------------------------------------------------------------
IRESNET_LAYERS = {
    "iresnet18": (2, 2, 2, 2),
    "iresnet34": (3, 4, 6, 3),
    "iresnet50": (3, 4, 14, 3),
    "iresnet100": (3, 13, 30, 3),
    "iresnet200": (6, 26, 60, 6),
}
STAGE_PLANES = (64, 128, 256, 512)
BN_EPS = 1e-5          # iresnet.py:37,39,42,77,95,98,123
BN_MOMENTUM = 0.1      # torch default, used everywhere in the reference


def _bn_keys(prefix: str, c: int) -> List[Tuple[str, Tuple[int, ...], str]]:
    return [
        (prefix + ".weight", (c,), "bn_w"),
        (prefix + ".bias", (c,), "bn_b"),
        (prefix + ".running_mean", (c,), "bn_rm"),
        (prefix + ".running_var", (c,), "bn_rv"),
        (prefix + ".num_batches_tracked", (), "bn_nbt"),
    ]


def iresnet_spec(layers: Sequence[int], num_features: int = 512,
                 in_hw: int = 112) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Ordered (key, shape, kind) list == reference ``state_dict()`` order.

    Mirrors module registration order in IResNet.__init__ (iresnet.py:76-98) and
    IBasicBlock.__init__ (iresnet.py:37-43).
    """
    spec: List[Tuple[str, Tuple[int, ...], str]] = []
    spec.append(("conv1.weight", (64, 3, 3, 3), "conv"))
    spec += _bn_keys("bn1", 64)
    spec.append(("prelu.weight", (64,), "prelu"))
    inpl = 64
    for si, (planes, nblk) in enumerate(zip(STAGE_PLANES, layers)):
        for bi in range(nblk):
            p = "layer%d.%d" % (si + 1, bi)
            cin = inpl if bi == 0 else planes
            spec += _bn_keys(p + ".bn1", cin)
            spec.append((p + ".conv1.weight", (planes, cin, 3, 3), "conv"))
            spec += _bn_keys(p + ".bn2", planes)
            spec.append((p + ".prelu.weight", (planes,), "prelu"))
            spec.append((p + ".conv2.weight", (planes, planes, 3, 3), "conv"))
            spec += _bn_keys(p + ".bn3", planes)
            if bi == 0:  # stride 2 in every stage => always a downsample (iresnet.py:120-125)
                spec.append((p + ".downsample.0.weight", (planes, cin, 1, 1), "conv"))
                spec += _bn_keys(p + ".downsample.1", planes)
        inpl = planes
    spec += _bn_keys("bn2", 512)
    fc_in = 512 * (in_hw // 16) ** 2
    spec.append(("fc.weight", (num_features, fc_in), "fc_w"))
    spec.append(("fc.bias", (num_features,), "fc_b"))
    spec += _bn_keys("features", num_features)
    return spec


def block_spec(cin: int, cout: int, downsample: bool) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Ordered (key, shape, kind) of a lone IBasicBlock == its reference ``state_dict()`` order (iresnet.py:37-43)."""
    spec: List[Tuple[str, Tuple[int, ...], str]] = []
    spec += _bn_keys("bn1", cin)
    spec.append(("conv1.weight", (cout, cin, 3, 3), "conv"))
    spec += _bn_keys("bn2", cout)
    spec.append(("prelu.weight", (cout,), "prelu"))
    spec.append(("conv2.weight", (cout, cout, 3, 3), "conv"))
    spec += _bn_keys("bn3", cout)
    if downsample:
        spec.append(("downsample.0.weight", (cout, cin, 1, 1), "conv"))
        spec += _bn_keys("downsample.1", cout)
    return spec


# name -> (cin, cout, stride, hw, batch, prelu slope fixed to one?)   (tests/golden/block.npz, tools/make_golden.py:gen_block)
# The "_lin" variants set the PReLU slope to 1: the block is then free of the derivative discontinuity at z = 0, and every output of a
# bf16-storage implementation — gradients included — stays within the 1e-2 class of the fp32 reference.  With the real slopes a
# rounding of the PReLU input flips the derivative of the elements next to zero, which alone moves the gradients behind it by 2-5e-2
# under bf16 storage and 0.5-2.5e-2 under the reference's own fp16 autocast (oracle/bf16_emul.py reproduces both).
BLOCK_FIXTURES = {
    "s1": (64, 64, 1, 14, 3, False), "s2": (64, 128, 2, 16, 3, False), "s3": (256, 256, 1, 14, 4, False),
    "s4": (128, 128, 1, 28, 2, False), "s5": (256, 512, 2, 14, 4, False),
    "s1_lin": (64, 64, 1, 14, 3, True), "s3_lin": (256, 256, 1, 14, 4, True),
}
FIXTURE_SAMPLE = 8192


def fixture_sample(t: torch.Tensor) -> torch.Tensor:
    """What a fixture keeps of a large tensor: every (numel // 8192)-th element of the flattened tensor (all of a small one)."""
    t = t.detach().reshape(-1)
    if t.numel() <= FIXTURE_SAMPLE:
        return t
    return t[:: t.numel() // FIXTURE_SAMPLE][:FIXTURE_SAMPLE]


def block_fixture(name: str):
    """Inputs of the IBasicBlock fixture ``name`` of tests/golden/block.npz: (state_dict, x, dy, stride) — RNG-free tensors that
    tools/make_golden.py:gen_block loads into the reference block (He-scaled hashed Gaussian filters and inputs: sinusoidal filters are
    low-rank and cancel so strongly that bf16 operand rounding alone moves the block output by 6-11 %)."""
    cin, cout, stride, hw, batch, lin = BLOCK_FIXTURES[name]
    seed = 7000 + 37 * sorted(BLOCK_FIXTURES).index(name)
    sd = OrderedDict()
    for i, (k, shape, kind) in enumerate(block_spec(cin, cout, stride != 1 or cin != cout)):
        a, b = 0.3 + 0.01 * i, 0.2 * i
        if kind == "bn_nbt":
            sd[k] = torch.tensor(2)
        elif kind == "bn_rv":
            sd[k] = closed_form(shape, a, b, 0.2, 1.0)
        elif kind == "bn_w":
            sd[k] = closed_form(shape, a, b, 0.25, 1.0)
        elif kind == "prelu":
            sd[k] = torch.ones(shape) if lin else closed_form(shape, a, b, 0.1, 0.25)
        elif kind == "conv":
            fan = shape[1] * shape[2] * shape[3]
            sd[k] = hash_normal(shape, seed + i) * math.sqrt(2.0 / fan)
        else:
            sd[k] = closed_form(shape, a, b, 0.1)
    x = hash_normal((batch, cin, hw, hw), seed + 100)
    dy = hash_normal((batch, cout, hw // stride, hw // stride), seed + 101)
    return sd, x, dy, stride


def block_fixture_run(name: str, block_fn=None):
    """The fixture through the restated block (``ibasic_block``, or ``block_fn(sd, prefix, x, stride, training)``): returns
    (y, dx, {param: grad}, state_dict after the forward)."""
    sd, x, dy, stride = block_fixture(name)
    sd = OrderedDict(("blk." + k, v.clone()) for k, v in sd.items())
    pk = [k for k, v in sd.items() if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var"))]
    for k in pk:
        sd[k].requires_grad_(True)
    x = x.clone().requires_grad_(True)
    y = (block_fn or ibasic_block)(sd, "blk", x, stride, True)
    y.backward(dy)
    grads = {k[4:]: sd[k].grad.detach().clone() for k in pk}
    out = OrderedDict((k[4:], v.detach()) for k, v in sd.items())
    return y.detach(), x.grad.detach(), grads, out


# --------------------------------------------------------------------------------------
# closed-form (RNG-free) tensors shared by the golden generator and the tests
# --------------------------------------------------------------------------------------
def closed_form(shape: Sequence[int], a: float, b: float, scale: float = 1.0,
                offset: float = 0.0) -> torch.Tensor:
    """``offset + scale * sin(a*i + b)`` over the flattened index, computed in fp64."""
    n = 1
    for s in shape:
        n *= int(s)
    i = torch.arange(n, dtype=torch.float64)
    v = offset + scale * torch.sin(a * i + b)
    return v.to(torch.float32).reshape(tuple(shape))


_M64 = (1 << 64) - 1


def _s64(v: int) -> int:
    v &= _M64
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(x: torch.Tensor, k: int) -> torch.Tensor:
    return (x >> k) & ((1 << (64 - k)) - 1)


def _splitmix64(x: torch.Tensor) -> torch.Tensor:
    """splitmix64 finaliser on int64 tensors (two's-complement wrap-around == uint64 arithmetic)."""
    x = x + _s64(0x9E3779B97F4A7C15)
    x = (x ^ _lsr(x, 30)) * _s64(0xBF58476D1CE4E5B9)
    x = (x ^ _lsr(x, 27)) * _s64(0x94D049BB133111EB)
    return x ^ _lsr(x, 31)


def hash_normal(shape: Sequence[int], seed: int) -> torch.Tensor:
    """N(0,1) samples defined by integer hashing + Box-Muller (portable: no library RNG involved)."""
    n = 1
    for s in shape:
        n *= int(s)
    i = torch.arange(n, dtype=torch.int64) + _s64(seed * 0x100000001B3 + 0x1234567)
    h1 = _splitmix64(i)
    h2 = _splitmix64(h1 ^ _s64(0xD6E8FEB86659FD93))
    u1 = (_lsr(h1, 11).double() + 0.5) / float(1 << 53)
    u2 = (_lsr(h2, 11).double() + 0.5) / float(1 << 53)
    z = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * math.pi * u2)
    return z.to(torch.float32).reshape(tuple(shape))


def head_fc(num_classes: int, dim: int = 512, seed: int = 7) -> torch.Tensor:
    """cosine-head class weights ~ N(0, 0.01) (reference init client.py:66)."""
    return hash_normal((num_classes, dim), 900 + seed) * 0.01


def closed_form_state_dict(layers: Sequence[int], num_features: int = 512, in_hw: int = 112, tag: float = 0.0,
                           residual_gain: float = 0.25) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic iresnet state (no RNG; ``tag`` decorrelates variants).

    ``residual_gain`` scales every block's last BN weight (bn3): like a trained ResNet, residual branches are
    small perturbations of the identity path, which keeps the 50-100-layer train-mode-BN network well conditioned
    (an untrained net with unit gains amplifies single bf16 roundings chaotically: emulated bf16 storage alone
    moves iresnet100's embeddings by 20 %, which would make any end-to-end tolerance meaningless)."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for idx, (key, shape, kind) in enumerate(iresnet_spec(layers, num_features, in_hw)):
        a = 0.37 + 0.011 * (idx % 89) + 0.0007 * tag
        b = 0.13 * idx + tag
        if kind == "conv":      # He-scaled hashed Gaussians (sinusoidal filters are low-rank and collapse deep features)
            fan_in = shape[1] * shape[2] * shape[3]
            t = hash_normal(shape, idx + 1000 * int(tag)) * math.sqrt(2.0 / fan_in)
        elif kind == "fc_w":
            t = hash_normal(shape, idx + 1000 * int(tag)) / math.sqrt(shape[1])
        elif kind == "fc_b":
            t = closed_form(shape, a, b, scale=0.05)
        elif kind == "bn_w":
            if key == "features.weight":  # frozen at 1 (iresnet.py:99-100)
                t = torch.ones(shape)
            else:
                t = closed_form(shape, a, b, scale=0.25, offset=1.0)
                if key.endswith(".bn3.weight"):
                    t = t * residual_gain
        elif kind == "bn_b":
            t = closed_form(shape, a, b, scale=0.1)
        elif kind == "bn_rm":
            t = closed_form(shape, a, b, scale=0.05)
        elif kind == "bn_rv":
            t = closed_form(shape, a, b, scale=0.2, offset=1.0)
        elif kind == "bn_nbt":
            t = torch.tensor(3 + (idx % 5), dtype=torch.int64)
        elif kind == "prelu":
            t = closed_form(shape, a, b, scale=0.1, offset=0.25)
        else:  # pragma: no cover
            raise AssertionError(kind)
        sd[key] = t
    return sd


def closed_form_images(batch: int, hw: int = 112, tag: float = 0.0) -> torch.Tensor:
    """Synthetic faces in [-1, 1] (reference normalisation dataset.py:81-86): every image is its own mix of
    two plane waves + a radial blob, so a batch is as diverse as distinct identities (well-conditioned batch
    statistics even at batch 4-8), still RNG-free."""
    ys = torch.arange(hw, dtype=torch.float64).view(1, 1, hw, 1) / hw
    xs = torch.arange(hw, dtype=torch.float64).view(1, 1, 1, hw) / hw
    i = torch.arange(batch, dtype=torch.float64).view(batch, 1, 1, 1) + 1.7 * tag
    c = torch.arange(3, dtype=torch.float64).view(1, 3, 1, 1)
    f1, f2 = 3.0 + 1.3 * ((i * 0.618) % 1.0) * 7.0, 2.0 + ((i * 0.414) % 1.0) * 9.0
    ph = 2.399963 * i + 0.7 * c
    w1 = torch.sin(2 * math.pi * (f1 * xs + 0.5 * f2 * ys) + ph)
    w2 = torch.sin(2 * math.pi * (f2 * ys - 0.3 * f1 * xs) + 1.3 * ph + c)
    cx, cy = 0.3 + 0.4 * ((i * 0.7548) % 1.0), 0.3 + 0.4 * ((i * 0.5698) % 1.0)
    blob = torch.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) * (8.0 + 4.0 * c))
    img = 0.45 * w1 + 0.35 * w2 + 0.6 * blob - 0.2
    return img.clamp(-1.0, 1.0).to(torch.float32)


def closed_form_labels(batch: int, num_classes: int, tag: int = 0) -> torch.Tensor:
    i = torch.arange(batch, dtype=torch.int64)
    return (i * 7919 + 13 * tag + 5) % num_classes


# --------------------------------------------------------------------------------------
# backbone  (reference: backbones/iresnet.py:46-57 block, :158-172 forward)
# --------------------------------------------------------------------------------------
def _bn(sd: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor, training: bool, frozen: bool = False) -> torch.Tensor:
    """nn.BatchNorm{1,2}d semantics: biased batch var for normalisation, unbiased for the
    running update, momentum 0.1, num_batches_tracked += 1 in training.  ``frozen``: the module was put into eval() inside a training
    net (IResNet.freeze_BN(test_mode=True), iresnet.py:140-147): running statistics normalise, nothing is updated."""
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    training = training and not frozen
    if training:
        sd[prefix + ".num_batches_tracked"] += 1
    return F.batch_norm(x, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"],
                        training, BN_MOMENTUM, BN_EPS)


def ibasic_block(sd: Dict[str, torch.Tensor], p: str, x: torch.Tensor, stride: int,
                 training: bool, bn_frozen: bool = False, prelu_hook=None) -> torch.Tensor:
    """iresnet.py:46-57: BN→conv3x3(s1)→BN→PReLU→conv3x3(stride)→BN, (+1x1 conv+BN shortcut), add.
    ``prelu_hook(name, z, weight)`` (tests only) stands in for F.prelu — e.g. a PReLU whose BACKWARD uses an injected sign pattern."""
    out = _bn(sd, p + ".bn1", x, training, bn_frozen)
    out = F.conv2d(out, sd[p + ".conv1.weight"], None, 1, 1)
    out = _bn(sd, p + ".bn2", out, training, bn_frozen)
    out = prelu_hook(p + ".prelu", out, sd[p + ".prelu.weight"]) if prelu_hook else F.prelu(out, sd[p + ".prelu.weight"])
    out = F.conv2d(out, sd[p + ".conv2.weight"], None, stride, 1)
    out = _bn(sd, p + ".bn3", out, training, bn_frozen)
    if (p + ".downsample.0.weight") in sd:
        idn = F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride, 0)
        idn = _bn(sd, p + ".downsample.1", idn, training, bn_frozen)
    else:
        idn = x
    return out + idn


def iresnet_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, layers: Sequence[int],
                    training: bool = True, return_taps: bool = False, dropout_p: float = 0.0,
                    dropout_mask: Optional[torch.Tensor] = None, bn_frozen: bool = False, prelu_hook=None):
    """iresnet.py:158-172 with fp16=False (CPU path).  ``bn_frozen``: after IResNet.freeze_BN(test_mode=True) (iresnet.py:140-147) — every
    BatchNorm in eval mode while the net trains (dropout stays on).  ``dropout_p`` > 0 with an injected keep-``dropout_mask`` [B, 25088] (0/1): the
    nn.Dropout(p, inplace=True) of iresnet.py:169 with that mask (torch's RNG stream is not part of the contract; FL configs use p = 0,
    client.py:142)."""
    taps = {}
    h = F.conv2d(x, sd["conv1.weight"], None, 1, 1)
    h = _bn(sd, "bn1", h, training, bn_frozen)
    h = prelu_hook("prelu", h, sd["prelu.weight"]) if prelu_hook else F.prelu(h, sd["prelu.weight"])
    taps["stem"] = h
    for si, nblk in enumerate(layers):
        for bi in range(nblk):
            h = ibasic_block(sd, "layer%d.%d" % (si + 1, bi), h, 2 if bi == 0 else 1, training, bn_frozen, prelu_hook)
        taps["layer%d" % (si + 1)] = h
    h = _bn(sd, "bn2", h, training, bn_frozen)
    h = torch.flatten(h, 1)
    if training and dropout_p > 0.0:
        h = h * dropout_mask.to(h.dtype) / (1.0 - dropout_p)
    h = F.linear(h, sd["fc.weight"], sd["fc.bias"])
    h = _bn(sd, "features", h, training, bn_frozen)
    return (h, taps) if return_taps else h


def trainable_keys(sd: Dict[str, torch.Tensor]) -> List[str]:
    """Keys ``model.parameters()`` yields with requires_grad (features.weight is frozen, iresnet.py:100)."""
    out = []
    for k, v in sd.items():
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            continue
        if k == "features.weight":
            continue
        out.append(k)
    return out


# --------------------------------------------------------------------------------------
# heads and losses
# --------------------------------------------------------------------------------------
def fc_module_forward(x: torch.Tensor, fc: torch.Tensor, normalize_feat: bool = True) -> torch.Tensor:
    """client.py:69-74: cosine logits, F.normalize p=2 dim=1 eps=1e-12 on both sides."""
    w = F.normalize(fc)
    return (F.normalize(x) if normalize_feat else x) @ w.t()


def cosface(cosine: torch.Tensor, label: torch.Tensor, s: float, m: float) -> torch.Tensor:
    """losses.py:23-29: subtract m at the target column of rows with label != -1, then scale by s."""
    rows = torch.nonzero(label != -1).flatten()
    shift = torch.zeros_like(cosine)
    shift[rows, label[rows]] = m
    return (cosine - shift) * s


def arcface(cosine: torch.Tensor, label: torch.Tensor, s: float, m: float) -> torch.Tensor:
    """losses.py:38-45: theta = acos(c) (unclamped); theta += m at target; cos(theta) * s."""
    rows = torch.nonzero(label != -1).flatten()
    theta = torch.acos(cosine)
    add = torch.zeros_like(cosine)
    add[rows, label[rows]] = m
    return torch.cos(theta + add) * s


MARGINS = {"CosFace": cosface, "ArcFace": arcface}


def bce_module_forward(x: torch.Tensor, labels: torch.Tensor, conv_w: torch.Tensor, conv_b: torch.Tensor,
                       weight: torch.Tensor, bias: torch.Tensor, m: float = 0.4, r: float = 30.0,
                       t: int = 3) -> Tuple[torch.Tensor, torch.Tensor]:
    """client.py:45-58 with converter_layer == 1 (config.py:31)."""
    feat = F.linear(x, conv_w, conv_b)
    cos = F.normalize(feat) @ F.normalize(weight).t()
    n_class = weight.shape[0]
    gt = torch.zeros(x.shape[0], n_class, dtype=torch.bool)
    inr = labels < n_class                        # labels >= n_class => all-negative row (client.py:48-52)
    gt[torch.nonzero(inr).flatten(), labels[inr]] = True
    g = 2.0 * ((cos + 1.0) / 2.0).pow(t) - 1.0
    z = torch.where(gt, r * (g - m), r * (g + m)) + bias.unsqueeze(0)
    return z, gt


def bce_loss(z: torch.Tensor, gt: torch.Tensor, r: float = 30.0, lambda_: float = 0.7) -> torch.Tensor:
    """losses.py:11-15, reduction 'sum_mean'."""
    pos = (lambda_ / r) * torch.log(1 + torch.exp(-z) + 1e-8)
    neg = ((1 - lambda_) / r) * torch.log(1 + torch.exp(z) + 1e-8)
    return torch.where(gt, pos, neg).sum(dim=1).mean()


def contrastive_loss(feats: torch.Tensor, global_feats: torch.Tensor, last_feats: torch.Tensor,
                     temperature: float = 0.5) -> torch.Tensor:
    """client.py:372-375: 2-way CE over cosine-sim(feats, global)/T vs cosine-sim(feats, last)/T."""
    pos = F.cosine_similarity(feats, global_feats, dim=1) / temperature
    neg = F.cosine_similarity(feats, last_feats, dim=1) / temperature
    return F.cross_entropy(torch.stack([pos, neg], dim=1), torch.zeros(len(feats), dtype=torch.long))


# --------------------------------------------------------------------------------------
# optimiser  (torch.optim.SGD semantics; client.py:335,527-529; config.py:8-9)
# --------------------------------------------------------------------------------------
def sgd_step(params: List[torch.Tensor], grads: List[Optional[torch.Tensor]],
             bufs: List[Optional[torch.Tensor]], lr: float, momentum: float = 0.9,
             weight_decay: float = 5e-4) -> None:
    """g += wd*p ; buf = g (first step) | mu*buf + g ; p -= lr*buf.  In place; bufs list is filled."""
    with torch.no_grad():
        for i, (p, g) in enumerate(zip(params, grads)):
            if g is None:
                continue
            d = g + weight_decay * p
            if bufs[i] is None:
                bufs[i] = d.clone()
            else:
                bufs[i].mul_(momentum).add_(d)
            p.sub_(lr * bufs[i])


def lr_step_func(epoch: int, steps=(6, 14)) -> float:
    """config.py:22-25."""
    if epoch < -1:
        return ((epoch + 1) / (4 + 1)) ** 2
    return 0.1 ** len([m for m in steps if m - 1 <= epoch])


# --------------------------------------------------------------------------------------
# federated aggregation (server.py:25-46)
# --------------------------------------------------------------------------------------
def fedpavg(models: List[Dict[str, torch.Tensor]], weights: Sequence[float]) -> "OrderedDict[str, torch.Tensor]":
    """server.py:25-34: w_i = n_i / sum(n) as Python floats; ascending-client accumulation from 0."""
    tot = sum(weights)
    ws = [w / tot for w in weights]
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name in models[0]:
        acc = 0
        for w, m in zip(ws, models):
            acc = acc + w * m[name]
        out[name] = acc
    return out


def fedavg_on_fc(pretrain_fc: torch.Tensor, models: List[torch.Tensor], weights: Sequence[float],
                 p: float) -> torch.Tensor:
    """server.py:36-46."""
    tot = sum(weights)
    ws = [w / tot for w in weights]
    acc = models[0].clone() * ws[0]
    for i in range(1, len(models)):
        acc = acc + models[i] * ws[i]
    return acc if p == 1 else (1 - p) * pretrain_fc + p * acc


# --------------------------------------------------------------------------------------
# one client local-training run  (client.py:511-571 ``Client.train``)
# --------------------------------------------------------------------------------------
def client_train(sd: Dict[str, torch.Tensor], fc: torch.Tensor, batches, layers: Sequence[int],
                 loss_name: str = "CosFace", s: float = 30.0, m: float = 0.4, lr: float = 0.1,
                 momentum: float = 0.9, weight_decay: float = 5e-4):
    """state_dict in -> N x (zero_grad, fwd, margin, CE, bwd, SGD step) -> state_dict out.

    ``batches`` is an iterable of (imgs, labels).  Fresh optimiser (momentum reset, F8).
    Returns (per-step losses, sd, fc); sd/fc are updated in place.
    """
    keys = trainable_keys(sd)
    params = [sd[k] for k in keys] + [fc]
    bufs: List[Optional[torch.Tensor]] = [None] * len(params)
    margin = MARGINS[loss_name]
    losses = []
    for imgs, labels in batches:
        if len(imgs) == 1:                       # client.py:538-540
            imgs, labels = torch.cat([imgs, imgs]), torch.cat([labels, labels])
        for p in params:
            p.requires_grad_(True)
            p.grad = None
        feats = iresnet_forward(sd, imgs, layers, training=True)
        logits = margin(fc_module_forward(feats, fc), labels, s, m)
        loss = F.cross_entropy(logits, labels)
        loss.backward()
        grads = [p.grad for p in params]
        for p in params:
            p.requires_grad_(False)
        sgd_step(params, grads, bufs, lr, momentum, weight_decay)
        losses.append(float(loss.detach()))
    return losses, sd, fc


def client_train_public(sd: Dict[str, torch.Tensor], fc: torch.Tensor, bce: Optional[Dict[str, torch.Tensor]], batches,
                        layers: Sequence[int], *, loss_name: str = "CosFace", s: float = 30.0, m: float = 0.4, lr: float = 0.05,
                        momentum: float = 0.9, weight_decay: float = 5e-4, bce_detach: bool = False,
                        global_sd: Optional[Dict[str, torch.Tensor]] = None, last_sd: Optional[Dict[str, torch.Tensor]] = None,
                        temperature: float = 0.5, mu: float = 5.0, reweight: Optional[Tuple[int, int]] = None):
    """One local epoch of ``Client.train_with_public_data`` (client.py:354-441) as a function:
    ``fc`` = [local | public] class centres (client.py:312); ``bce`` = {'converter.0.weight','converter.0.bias','weight','bias'}
    or None (args.BCE_local); ``global_sd``/``last_sd`` given => model-contrastive term (args.contrastive_bb) with the frozen
    global / last-round backbones in eval mode (client.py:326-329); ``reweight`` = (num_classes, num_client) => client.py:269-285
    (quirk kept: the reference concatenates under no_grad, so the re-weighted CosFace loss is reported but not differentiated).
    loss = cos + 10*bce + mu*con.  Fresh SGD over backbone + fc + bce parameters (client.py:335).
    Returns (rows of (loss, cos, con, bce) per step, sd, fc, bce) — updated in place."""
    keys = trainable_keys(sd)
    bkeys = ["converter.0.weight", "converter.0.bias", "weight", "bias"] if bce is not None else []
    params = [sd[k] for k in keys] + [fc] + [bce[k] for k in bkeys]
    bufs: List[Optional[torch.Tensor]] = [None] * len(params)
    margin = MARGINS[loss_name]
    rows = []
    for imgs, labels in batches:
        for p in params:
            p.requires_grad_(True)
            p.grad = None
        con = bl = None
        if global_sd is not None:
            with torch.no_grad():
                gfe = iresnet_forward(global_sd, imgs, layers, training=False)
                lfe = iresnet_forward(last_sd, imgs, layers, training=False)
        feats = iresnet_forward(sd, imgs, layers, training=True)
        logits = margin(fc_module_forward(feats, fc), labels, s, m)
        if reweight is not None:
            ncls, nclient = reweight
            with torch.no_grad():
                keep = torch.ones(logits.shape, dtype=torch.bool)
                keep[torch.arange(len(labels)), labels] = False
                tmp = logits.detach().clone()[keep].reshape(len(labels), logits.shape[1] - 1)[:, :ncls].repeat(1, nclient - 1)
                logits = torch.cat([logits, tmp], dim=1)     # INSIDE no_grad, as client.py:276: the CosFace term then carries no gradient
        cos = F.cross_entropy(logits, labels)
        loss = cos
        if bce is not None:
            z, gt = bce_module_forward(feats.detach() if bce_detach else feats, labels, bce["converter.0.weight"],
                                       bce["converter.0.bias"], bce["weight"], bce["bias"])
            bl = bce_loss(z, gt)
            loss = loss + 10 * bl
        if global_sd is not None:
            con = contrastive_loss(feats, gfe, lfe, temperature)
            loss = loss + mu * con
        loss.backward()
        grads = [p.grad for p in params]
        for p in params:
            p.requires_grad_(False)
        sgd_step(params, grads, bufs, lr, momentum, weight_decay)
        rows.append((float(loss.detach()), float(cos.detach()), None if con is None else float(con.detach()),
                     None if bl is None else float(bl.detach())))
    return rows, sd, fc, bce


# --------------------------------------------------------------------------------------
# sphnet (reference backbones/sphnet.py:4-73) — functional restatement
# --------------------------------------------------------------------------------------
SPHERE_LAYERS = {20: (1, 2, 4, 1), 64: (3, 7, 16, 3)}


def sphere_state_dict(type_: int = 20, tag: float = 0.0) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic sphnet state in reference key order (hash-generated Gaussians; fan-in scaled so activations stay O(1))."""
    filt = [3, 64, 128, 256, 512]
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    seed = [7000 + int(tag * 17)]

    def nxt(shape, scale):
        seed[0] += 1
        return hash_normal(shape, seed[0]) * scale

    for L, nblk in enumerate(SPHERE_LAYERS[type_]):
        cin, c = filt[L], filt[L + 1]
        pre = "layer%d." % (L + 1)
        sd[pre + "0.weight"] = nxt((c, cin, 3, 3), (2.0 / (9 * cin)) ** 0.5)
        sd[pre + "0.bias"] = nxt((c,), 0.1)
        sd[pre + "1.weight"] = 0.25 + nxt((c,), 0.05)
        for b in range(nblk):
            bp = pre + "%d." % (2 + b)
            sd[bp + "conv1.weight"] = nxt((c, c, 3, 3), (1.0 / (9 * c)) ** 0.5)
            sd[bp + "prelu1.weight"] = 0.25 + nxt((c,), 0.05)
            sd[bp + "conv2.weight"] = nxt((c, c, 3, 3), (1.0 / (9 * c)) ** 0.5)
            sd[bp + "prelu2.weight"] = 0.25 + nxt((c,), 0.05)
    sd["fc.weight"] = nxt((512, 512 * 7 * 7), (1.0 / (512 * 49)) ** 0.5)
    sd["fc.bias"] = nxt((512,), 0.05)
    return sd


def sphere_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, type_: int = 20) -> torch.Tensor:
    """sphnet.py:62-71 (fp32): 4 stages of [conv3x3 s2 + bias -> PReLU -> n x (x + prelu(conv(prelu(conv(x)))))], flatten, fc."""
    for L, nblk in enumerate(SPHERE_LAYERS[type_]):
        pre = "layer%d." % (L + 1)
        x = F.prelu(F.conv2d(x, sd[pre + "0.weight"], sd[pre + "0.bias"], 2, 1), sd[pre + "1.weight"])
        for b in range(nblk):
            bp = pre + "%d." % (2 + b)
            t = F.prelu(F.conv2d(x, sd[bp + "conv1.weight"], None, 1, 1), sd[bp + "prelu1.weight"])
            x = x + F.prelu(F.conv2d(t, sd[bp + "conv2.weight"], None, 1, 1), sd[bp + "prelu2.weight"])
    return F.linear(x.reshape(x.shape[0], -1), sd["fc.weight"], sd["fc.bias"])


def sphere_step_grads(sd: Dict[str, torch.Tensor], x: torch.Tensor, dfeats: torch.Tensor, type_: int = 20):
    """(feats, {key: grad}) for the scalar <feats, dfeats>."""
    ps = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    feats = sphere_forward(ps, x, type_)
    (feats * dfeats).sum().backward()
    return feats.detach(), {k: v.grad for k, v in ps.items()}


# --------------------------------------------------------------------------------------
# inference sweeps + hard-negative mining
# --------------------------------------------------------------------------------------
def embed(sd: Dict[str, torch.Tensor], batches, layers: Sequence[int], normalize: bool = True) -> torch.Tensor:
    """server.py:242-263 Generate_pretrain_feats / client.py:197-205: eval-mode backbone, F.normalize, concatenated."""
    out = []
    with torch.no_grad():
        for imgs, _ in batches:
            f = iresnet_forward(sd, imgs, layers, training=False)
            out.append(F.normalize(f) if normalize else f)
    return torch.cat(out, dim=0)


def class_centers(sd: Dict[str, torch.Tensor], batches, layers: Sequence[int], num_classes: int, norm_before_avg: bool) -> torch.Tensor:
    """client.py:159-188 data_update_fc / server.py:182-240 Initialize_pretrain_FC: per-batch per-class feature sums, divided by the
    per-class sample counts (a class without samples gives 0/0 = NaN, as in the reference)."""
    init_fc = torch.zeros(num_classes, 512)
    num = torch.zeros(num_classes)
    with torch.no_grad():
        for imgs, label in batches:
            f = iresnet_forward(sd, imgs, layers, training=False)
            if norm_before_avg:
                f = F.normalize(f)
            for l in torch.unique(label):
                init_fc[l:l + 1, :] += torch.sum(f[label == l, :], dim=0)
                num[l] += torch.sum(label == l)
    return init_fc / num.unsqueeze(1)


def hard_negative_indices(local_feats: torch.Tensor, pretrained_feats: torch.Tensor, threshold: float) -> torch.Tensor:
    """client.py:208-226 choose_hard_negative_2: sorted union over local rows of the public columns with similarity > threshold."""
    sim = local_feats @ pretrained_feats.t()
    return torch.unique(torch.where(sim > threshold)[1])


def roc_histogram(feature, label, target_size: int):
    """roc_cuda.py:14-30 + the batching of :36-58 collapsed: pairs (a, b), a < b, a < target_size; float64 dot products of the
    float32 features; bin int((dot + 1) * 1000); column 0 same label, column 1 different.  numpy int64 [2001, 2]."""
    import numpy as np
    f = np.asarray(feature, dtype=np.float32).astype(np.float64)
    lab = np.asarray(label).astype(np.int64)
    out = np.zeros((2001, 2), dtype=np.int64)
    for a in range(int(target_size)):
        if a + 1 >= len(f):
            break
        d = f[a + 1:] @ f[a]
        b = ((d + 1.0) * 1000.0).astype(np.int64)              # truncation towards zero; the argument is >= 0
        same = lab[a + 1:] == lab[a]
        np.add.at(out[:, 0], b[same], 1)
        np.add.at(out[:, 1], b[~same], 1)
    return out


def roc_tpr_at_fpr(hist):
    """roc_cuda.py:61-78 plot_ROC: TPR (%) at FPR = 1e-1 ... 1e-6."""
    import numpy as np
    from scipy.interpolate import interp1d
    data = np.cumsum(np.asarray(hist, dtype=np.int64), axis=0)
    tpr, fpr = [1.0], [1.0]
    for i in range(data.shape[0]):
        tpr.append((data[-1, 0] - data[i, 0]) / data[-1, 0])
        fpr.append((data[-1, 1] - data[i, 1]) / data[-1, 1])
    tpr, fpr = np.array(tpr), np.array(fpr)
    idx = np.argsort(fpr)
    roc = interp1d(fpr[idx], tpr[idx])
    return [float("%.2f" % (100 * roc(10 ** i))) for i in range(-1, -7, -1)]


def mining_fixture_state(g):
    """inputs of tests/golden/mining_r18.npz rebuilt from the closed forms: (sd, local batches, public batches)."""
    layers = IRESNET_LAYERS["iresnet18"]
    sd = closed_form_state_dict(layers, tag=float(g["tag"]))
    B, nb = int(g["B"]), int(g["nb"])
    local = [(closed_form_images(B, tag=10.0 + i), closed_form_labels(B, int(g["n_local"]), tag=i)) for i in range(nb)]
    # every third public image is a blend dominated by a local image (a true hard negative, cosine ~0.9 to it); the others are
    # unrelated closed-form images (cosine ~0.5): the selected set is then stable under 1e-2-class embedding noise
    allloc = torch.cat([x for x, _ in local], dim=0)
    public = []
    for i in range(nb):
        x = closed_form_images(B, tag=20.0 + i).clone()
        for s_ in range(B):
            k = i * B + s_
            if k % 3 == 0:
                x[s_] = 0.85 * allloc[(5 * k + 1) % len(allloc)] + 0.15 * x[s_]
        public.append((x, closed_form_labels(B, int(g["n_public"]), tag=5 + i)))
    return sd, local, public


def public_fixture_state(g, variant: str):
    """Initial state + batches of the tests/golden/client_public_*.npz fixtures (tools/make_golden.py gen_public), rebuilt from
    the closed forms: returns (sd, fc, bce-or-None, batches, kwargs for client_train_public)."""
    layers = IRESNET_LAYERS["iresnet18"]
    nl, npub, B, steps = int(g["n_local"]), int(g["n_public"]), int(g["B"]), int(g["steps"])
    C = nl + npub
    sd = closed_form_state_dict(layers, tag=float(g["tag"]))
    fc = torch.cat([head_fc(nl, seed=11), head_fc(npub, seed=12)], dim=0)
    bce = None
    if variant in ("full", "bce_rw"):
        bce = {"converter.0.weight": torch.eye(512), "converter.0.bias": torch.zeros(512), "weight": head_fc(nl, seed=13),
               "bias": torch.zeros(nl)}
    kw = dict(lr=float(g["lr"]), mu=float(g["mu"]), temperature=float(g["temperature"]))
    if variant == "full":
        kw["global_sd"] = closed_form_state_dict(layers, tag=float(g["tag"]))
        kw["last_sd"] = closed_form_state_dict(layers, tag=float(g["last_tag"]))
    if variant == "bce_rw":
        kw["reweight"] = (nl, 4)
    batches = [(closed_form_images(B, tag=float(st)), closed_form_labels(B, C, tag=st)) for st in range(steps)]
    return sd, fc, bce, batches, kw


def train_step_grads(sd: Dict[str, torch.Tensor], fc: torch.Tensor, imgs: torch.Tensor,
                     labels: torch.Tensor, layers: Sequence[int], loss_name: str = "CosFace",
                     s: float = 30.0, m: float = 0.4, prelu_hook=None):
    """One fwd+bwd; returns (feats, cosine, loss, {key: grad}, fc_grad).  BN buffers in sd are updated.
    ``prelu_hook``: see ibasic_block (tests inject the HIP path's PReLU sign pattern into this backward pass)."""
    keys = trainable_keys(sd)
    ps = [sd[k].requires_grad_(True) for k in keys]
    fc.requires_grad_(True)
    for p in ps + [fc]:
        p.grad = None
    feats = iresnet_forward(sd, imgs, layers, training=True, prelu_hook=prelu_hook)
    cosine = fc_module_forward(feats, fc)
    logits = MARGINS[loss_name](cosine.clone(), labels, s, m)
    loss = F.cross_entropy(logits, labels)
    loss.backward()
    grads = {k: sd[k].grad.detach().clone() for k in keys}
    fcg = fc.grad.detach().clone()
    for p in ps + [fc]:
        p.requires_grad_(False)
        p.grad = None
    return feats.detach(), cosine.detach(), float(loss.detach()), grads, fcg


# --------------------------------------------------------------------------------------
# PartialFC  (partial_fc.py:19-176) — functional restatement, collectives via callables
# --------------------------------------------------------------------------------------
class SingleRankComm:
    """world_size == 1 stand-in for the six torch.distributed call sites (partial_fc.py:122-173)."""
    world_size = 1
    rank = 0

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        return t.clone()

    def all_reduce_max(self, t: torch.Tensor) -> torch.Tensor:
        return t

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        return t

    def reduce_scatter_sum(self, t: torch.Tensor) -> torch.Tensor:
        return t.clone()


class DistComm:
    """gloo/RCCL-backed comm with the same four verbs (gloo lacks reduce_scatter: all_reduce + chunk)."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.world_size = dist.get_world_size()
        self.rank = dist.get_rank()

    def all_gather(self, t):
        outs = [torch.zeros_like(t) for _ in range(self.world_size)]
        self.dist.all_gather(outs, t.contiguous())
        return torch.cat(outs, dim=0)

    def all_reduce_max(self, t):
        t = t.clone()
        self.dist.all_reduce(t, self.dist.ReduceOp.MAX)
        return t

    def all_reduce_sum(self, t):
        t = t.clone()
        self.dist.all_reduce(t, self.dist.ReduceOp.SUM)
        return t

    def reduce_scatter_sum(self, t):
        t = t.clone()
        self.dist.all_reduce(t, self.dist.ReduceOp.SUM)
        return t.chunk(self.world_size, dim=0)[self.rank].clone()


def pfc_shard(num_classes: int, world_size: int, rank: int) -> Tuple[int, int]:
    """partial_fc.py:34-35 → (num_local, class_start)."""
    num_local = num_classes // world_size + int(rank < num_classes % world_size)
    class_start = num_classes // world_size * rank + min(rank, num_classes % world_size)
    return num_local, class_start


def pfc_sample(total_label: torch.Tensor, num_local: int, class_start: int, num_sample: int,
               sample_rate: float, perm: Optional[torch.Tensor] = None):
    """partial_fc.py:89-106.  ``perm`` = the uniform draw (torch.rand(num_local)); injected so
    parity does not depend on RNG streams.  Returns (remapped label, index or None)."""
    lab = total_label.clone()
    pos_mask = (class_start <= lab) & (lab < class_start + num_local)
    lab[~pos_mask] = -1
    lab[pos_mask] -= class_start
    if int(sample_rate) == 1:
        return lab, None
    positive = torch.unique(lab[pos_mask], sorted=True)
    if num_sample - positive.numel() >= 0:
        pr = perm.clone()
        pr[positive] = 2.0
        index = torch.topk(pr, k=num_sample)[1].sort()[0]
    else:
        index = positive
    lab[pos_mask] = torch.searchsorted(index, lab[pos_mask])
    return lab, index


def pfc_forward_backward(label: torch.Tensor, features: torch.Tensor, weight: torch.Tensor,
                         weight_mom: torch.Tensor, comm, batch_size: int, num_classes: int,
                         sample_rate: float, margin_name: str, s: float, m: float,
                         perm: Optional[torch.Tensor] = None):
    """partial_fc.py:118-176.  Returns dict with x_grad, loss_v, index, total_label,
    sub_weight, sub_weight_grad (grad wrt the *un-normalised* sampled rows)."""
    W, rank = comm.world_size, comm.rank
    num_local, class_start = pfc_shard(num_classes, W, rank)
    num_sample = int(sample_rate * num_local)
    total_label = comm.all_gather(label)                               # C1 :122
    total_label, index = pfc_sample(total_label, num_local, class_start, num_sample, sample_rate, perm)
    sub_weight = (weight if index is None else weight[index]).clone().requires_grad_(True)
    norm_weight = F.normalize(sub_weight)                              # :127
    total_features = comm.all_gather(features.detach()).requires_grad_(True)   # C2 :134
    logits = F.linear(total_features, norm_weight)                     # :110
    logits = MARGINS[margin_name](logits, total_label, s, m)           # :138
    with torch.no_grad():
        mx = comm.all_reduce_max(logits.max(dim=1, keepdim=True)[0])   # C3 :142
        ex = torch.exp(logits - mx)
        sm = comm.all_reduce_sum(ex.sum(dim=1, keepdim=True))          # C4 :147
        prob = ex / sm
        rows = torch.nonzero(total_label != -1).flatten()
        loss = torch.zeros(prob.shape[0], 1)
        loss[rows] = prob[rows].gather(1, total_label[rows, None])
        loss = comm.all_reduce_sum(loss)                               # C5 :161
        loss_v = -loss.clamp_min(1e-30).log().mean()
        grad = prob.clone()
        grad[rows, total_label[rows]] -= 1.0
        grad = grad / (batch_size * W)
    logits.backward(grad)
    x_grad = comm.reduce_scatter_sum(total_features.grad) * W          # C6 :173-174
    return {
        "x_grad": x_grad, "loss_v": loss_v, "index": index, "total_label": total_label,
        "sub_weight": sub_weight.detach(), "sub_weight_grad": sub_weight.grad.detach(),
        "num_local": num_local, "class_start": class_start,
    }


def pfc_sgd_update(weight: torch.Tensor, weight_mom: torch.Tensor, index: Optional[torch.Tensor],
                   sub_weight_grad: torch.Tensor, lr: float, momentum: float = 0.9,
                   weight_decay: float = 5e-4) -> None:
    """Caller protocol (upstream convention): opt.step() on sub_weight with the aliased
    momentum rows (partial_fc.py:124-126, buffer already exists => buf = mu*buf + g), then
    ``update()`` scatters rows back (partial_fc.py:113-116)."""
    with torch.no_grad():
        if index is None:
            sw, sm = weight, weight_mom
        else:
            sw, sm = weight[index], weight_mom[index]
        d = sub_weight_grad + weight_decay * sw
        sm = momentum * sm + d
        sw = sw - lr * sm
        if index is None:
            weight.copy_(sw)
            weight_mom.copy_(sm)
        else:
            weight[index] = sw
            weight_mom[index] = sm


# --------------------------------------------------------------------------------------
# BASELINE config 5 — build-defined hybrid, composed from the restated reference pieces:
# per-client backbone (iresnet.py) + ONE class-sharded PartialFC over all ranks (partial_fc.py:118-176) + a private BCE_module per
# client (client.py:25-60, weight 10 as client.py:383), momentum-SGD on everything, FedPavg of the backbones at round end.
# --------------------------------------------------------------------------------------
def config5_client_steps(sd: Dict[str, torch.Tensor], pfc_weight: torch.Tensor, pfc_mom: torch.Tensor, bce: Dict[str, torch.Tensor],
                         batches, layers: Sequence[int], comm, batch_size: int, num_classes: int, sample_rate: float, id_base: int,
                         lr: float, perms=None, margin_name: str = "CosFace", s: float = 30.0, m: float = 0.4, momentum: float = 0.9,
                         weight_decay: float = 5e-4, bce_weight: float = 10.0):
    """One rank's local steps.  ``batches``: [(imgs, GLOBAL labels)]; ``bce``: {conv_w, conv_b, weight, bias}; all state updated in place.
    Returns per-step (loss, cos_loss, bce_loss)."""
    keys = trainable_keys(sd)
    params = [sd[k] for k in keys]
    bufs: List[Optional[torch.Tensor]] = [None] * len(params)
    bkeys = ["conv_w", "conv_b", "weight", "bias"]
    bbufs: List[Optional[torch.Tensor]] = [None] * len(bkeys)
    n_local_ids = bce["weight"].shape[0]
    out = []
    for st, (imgs, labels) in enumerate(batches):
        for p in params:
            p.requires_grad_(True)
            p.grad = None
        feats = iresnet_forward(sd, imgs, layers, training=True)
        fdet = feats.detach()
        # shared sharded head on the normalised embeddings (upstream PartialFC protocol)
        fn_in = fdet.clone().requires_grad_(True)
        fn = F.normalize(fn_in)
        r = pfc_forward_backward(labels, fn.detach(), pfc_weight, pfc_mom, comm, batch_size, num_classes, sample_rate, margin_name, s, m,
                                 None if perms is None else perms[st])
        fn.backward(r["x_grad"])
        dfeats = fn_in.grad.clone()
        # private personalised head
        bp = [bce[k].requires_grad_(True) for k in bkeys]
        for p in bp:
            p.grad = None
        leaf = fdet.clone().requires_grad_(True)
        lab = labels - id_base
        lab = torch.where((lab < 0) | (lab >= n_local_ids), torch.full_like(lab, n_local_ids), lab)
        z, gt = bce_module_forward(leaf, lab, *bp)
        bl = bce_loss(z, gt)
        (bce_weight * bl).backward()
        dfeats = dfeats + leaf.grad
        feats.backward(dfeats)
        grads = [p.grad for p in params]
        bgrads = [p.grad for p in bp]
        for p in params + bp:
            p.requires_grad_(False)
        sgd_step(params, grads, bufs, lr, momentum, weight_decay)
        sgd_step(bp, bgrads, bbufs, lr, momentum, weight_decay)
        pfc_sgd_update(pfc_weight, pfc_mom, r["index"], r["sub_weight_grad"], lr, momentum, weight_decay)
        for p in params + bp:
            p.grad = None
        out.append((float(r["loss_v"]) + bce_weight * float(bl.detach()), float(r["loss_v"]), float(bl.detach())))
    return out
