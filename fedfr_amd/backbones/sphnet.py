"""SphereFace backbone (reference backbones/sphnet.py:4-73 — the network `run.sh` trains) on the MI355X kernels.

Same constructor / factory / state_dict keys as the reference (`sphere(type=20|64)`, `sphnet(pretrained, dropout, fp16, type)`;
keys `layerL.0.weight|bias`, `layerL.1.weight`, `layerL.K.conv1|prelu1|conv2|prelu2.weight`, `fc.weight|bias`).  There is no
normalisation layer: every unit is conv3x3 (+bias on the stride-2 stage heads) -> PReLU, residual blocks add their input.

Round 3: the network is a C++ plan like iresnet's (`csrc/net_sph.inc`, `fedfr_net_create_sphere`) behind the SAME entry points
(`fedfr_net_forward` / `_backward2` / `_prepare_weights`), so this class is `IResNet`'s storage and dispatch machinery with another
module tree: one flat fp32 parameter tensor (conv weights KRSC = channels_last OIHW views), bf16 shadows, one call per pass — and
`client.FusedTrainer`, `client.shared_backbone`, the flat FedAvg path and `Client.train` take it like an iresnet.  (Round 2
sequenced it from Python over the per-op ABI: a few hundred ctypes calls and as many allocations per step.)
Per pass: conv kernels of the iresnet plan; y = prelu(conv + bias) (+ identity) as one streaming pass; the PReLU backward as ONE
pass that also folds the block's identity add in and leaves partial sums for dbias / dalpha; weight gradients on the second stream."""
from __future__ import annotations

import torch
from torch import nn

from .. import _C
from . import iresnet as _ir
from .iresnet import IResNet, _Holder


class _Conv(_Holder):
    def __init__(self, cin, cout, bias):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.has_bias = cin, cout, (3, 3), bias


class _PReLU(_Holder):
    def __init__(self, c):
        super().__init__()
        self.num_parameters = c


class Block(_Holder):
    """reference sphnet.py:4-13"""

    def __init__(self, planes):
        super().__init__()
        self.conv1, self.prelu1 = _Conv(planes, planes, False), _PReLU(planes)
        self.conv2, self.prelu2 = _Conv(planes, planes, False), _PReLU(planes)


class _Linear(_Holder):
    def __init__(self, i, o):
        super().__init__()
        self.in_features, self.out_features = i, o


class sphere(IResNet):
    def __init__(self, type=20, is_gray=False, fp16=False):
        nn.Module.__init__(self)
        if type == 20:
            layers = [1, 2, 4, 1]
        elif type == 64:
            layers = [3, 7, 16, 3]
        else:
            raise ValueError("sphere" + str(type) + " IS NOT SUPPORTED! (sphere20 or sphere64)")
        if is_gray:
            raise NotImplementedError("fedfr_amd sphnet: is_gray is not built")
        self.fp16 = fp16
        self.sphere_type = int(type)
        self.layers_cfg = tuple(layers)
        self.filters = [3, 64, 128, 256, 512]
        self.num_features = 512
        self.in_hw = 112
        self.dropout_p = 0.0
        self.dropout_seed = 100
        self._dropout_step = 0
        counts, table = _ir._tensor_table(self.layers_cfg, self.in_hw, 512, sphere_type=self.sphere_type)
        self._counts, self._table = counts, table
        self._flat_state = torch.zeros(self._state_len(counts), dtype=torch.float32)
        self._slice_state()
        self._flat_nbt = torch.zeros(0, dtype=torch.int64)
        self._flat_grads = None
        self._shadow = None
        self._plans = {}
        self._shadow_dirty = True
        self._grads_live = False
        self._fwd_generation = 0
        self._bn_frozen = False
        self.validation_fp32 = False
        self._anchor = torch.zeros(1, requires_grad=True)
        filt = self.filters
        for L in range(4):                                      # nn.Sequential(conv, PReLU, Block, ...) per stage (sphnet.py:49-56)
            mods = [_Conv(filt[L], filt[L + 1], True), _PReLU(filt[L + 1])] + [Block(filt[L + 1]) for _ in range(layers[L])]
            setattr(self, "layer%d" % (L + 1), nn.Sequential(*mods))
        self.fc = _Linear(512 * 7 * 7, 512)
        self._bind(create=True)
        self._init_weights(False)

    def _init_weights(self, zero_init_residual=False):
        """reference sphnet.py:39-46: conv / linear WITH a bias: xavier_uniform weight, zero bias; conv without: N(0, 0.01); PReLU 0.25."""
        mods = dict(self.named_modules())
        with torch.no_grad():
            for owner, attr, v, is_param, kind, name in self._views():
                if kind in (_ir.KIND_CONV, _ir.KIND_FCW):
                    has_bias = kind == _ir.KIND_FCW or mods[name.rsplit(".", 1)[0]].has_bias
                    if has_bias:
                        nn.init.xavier_uniform_(v)
                    else:
                        v.normal_(0, 0.01)
                elif kind == _ir.KIND_PRELU:
                    v.fill_(0.25)
                else:
                    v.zero_()

    def freeze_BN(self, *a, **k):                               # no BatchNorm in this network
        pass

    def unfreeze_BN(self, *a, **k):
        pass

    def save(self, file_path):
        with open(file_path, "wb") as f:
            torch.save(self.state_dict(), f)


def sphnet(pretrained=False, dropout=None, fp16=False, type=64):
    """reference sphnet.py:72-73"""
    return sphere(type, fp16=fp16)
