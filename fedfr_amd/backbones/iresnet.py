"""IResNet on MI355X — drop-in for the reference's ``backbones/iresnet.py``.

Same factories and constructor signature (reference iresnet.py:60-63, :182-204), same module tree
and therefore the same ``state_dict`` keys, same ``forward(x[B,3,112,112]) -> [B,512]``.
What differs is underneath:

* every parameter / buffer is a *view* into one flat fp32 tensor (conv weights are stored KRSC, i.e.
  they are channels_last OIHW tensors), so SGD, FedAvg and the 16-bit weight shadows are single flat
  kernels / collectives;
* forward and backward are one call each into libfedfr_hip.so (``fedfr_net_forward`` /
  ``fedfr_net_backward``): NHWC 16-bit activations (fp16 storage in the product library), MFMA implicit-GEMM convolutions, fused BN
  statistics, fp32 master weights.  There is no PyTorch fallback: without the library (or on a CPU
  tensor) ``forward`` raises.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from .. import _C

__all__ = ["iresnet18", "iresnet34", "iresnet50", "iresnet100", "iresnet200", "IResNet", "IBasicBlock", "BlockPlan"]

KIND_CONV, KIND_BNW, KIND_BNB, KIND_PRELU, KIND_FCW, KIND_FCB, KIND_RM, KIND_RV, KIND_NBT = range(9)


# ---------------------------------------------------------------------------------------------
# holder modules: they only carry Parameters/buffers under the reference's attribute names
# ---------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("fedfr_amd: sub-modules are parameter holders; call the IResNet itself "
                           "(the whole network runs as fused HIP kernels)")


class Conv2d(_Holder):
    def __init__(self, cin, cout, k, stride):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.stride = cin, cout, (k, k), (stride, stride)


class BatchNorm2d(_Holder):     # class name contains 'BatchNorm' (reference freeze_BN, iresnet.py:142)
    def __init__(self, c):
        super().__init__()
        self.num_features, self.eps, self.momentum = c, 1e-5, 0.1


class BatchNorm1d(BatchNorm2d):
    pass


class PReLU(_Holder):
    def __init__(self, c):
        super().__init__()
        self.num_parameters = c


class Linear(_Holder):
    def __init__(self, i, o):
        super().__init__()
        self.in_features, self.out_features = i, o


class Downsample(nn.Sequential):
    pass


class IBasicBlock(_Holder):
    """Holder with the reference's attribute names (iresnet.py:37-43)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.bn1 = BatchNorm2d(inplanes)
        self.conv1 = Conv2d(inplanes, planes, 3, 1)
        self.bn2 = BatchNorm2d(planes)
        self.prelu = PReLU(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride)
        self.bn3 = BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


class _Plan:
    """One C-side plan (batch-size specific) + its activation arena and workspace."""

    def __init__(self, layers, batch, in_hw, nfeat, device, sphere_type=0):
        l = _C.lib()
        if sphere_type:                    # sphnet plan (fedfr_net_create_sphere): same entry points, same buffers
            self.handle = l.fedfr_net_create_sphere(sphere_type, batch)
        else:
            arr = (C.c_int * 4)(*layers)
            self.handle = l.fedfr_net_create(arr, batch, in_hw, nfeat)
        if not self.handle:
            raise RuntimeError("fedfr_net_create failed: " + _C.last_error())
        self.batch = batch
        self.dropout_seed = None          # seed the C-side plan was last given (IResNet._run_forward)
        self.mask_off = 0
        self.act = torch.empty(self.query(_C.Q_ACT_BYTES), dtype=torch.uint8, device=device)
        self.ws = torch.empty(self.query(_C.Q_WS_BYTES), dtype=torch.uint8, device=device)

    def query(self, what) -> int:
        q = C.c_longlong()
        _C.call("fedfr_net_query", self.handle, what, C.byref(q))
        return q.value

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _C.lib().fedfr_net_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def _read_table(h):
    q = C.c_longlong()
    counts = {}
    for key in (_C.Q_PARAM_COUNT, _C.Q_TRAINABLE_COUNT, _C.Q_BUFFER_COUNT, _C.Q_NBT_COUNT, _C.Q_SHADOW_COUNT,
                _C.Q_NUM_TENSORS, _C.Q_FC_IN, _C.Q_ACT_BYTES, _C.Q_WS_BYTES):
        _C.call("fedfr_net_query", h, key, C.byref(q))
        counts[key] = q.value
    table = []
    name = C.create_string_buffer(128)
    kind, region, ndim = C.c_int(), C.c_int(), C.c_int()
    off = C.c_longlong()
    shape = (C.c_int * 4)()
    for i in range(counts[_C.Q_NUM_TENSORS]):
        _C.call("fedfr_net_tensor_info", h, i, name, 128, C.byref(kind), C.byref(region), C.byref(off), C.byref(ndim), shape)
        table.append((name.value.decode(), kind.value, region.value, off.value, tuple(shape[: ndim.value])))
    return counts, table


class BlockPlan:
    """A lone ``IBasicBlock(inplanes, planes, stride, downsample)`` (reference backbones/iresnet.py:28-57) on the GPU: the same C++
    block code the whole-network plan runs, driven through ``fedfr_block_create``.  ``forward`` / ``backward`` exchange fp32 NCHW
    tensors like the reference block; parameters use the reference block's state_dict keys.  Exists for block-level parity checks
    (tests/golden/block.npz); networks run as one plan (``IResNet``)."""

    def __init__(self, inplanes, planes, stride, hin, batch, device):
        l = _C.lib()
        self.handle = l.fedfr_block_create(inplanes, planes, stride, hin, batch)
        if not self.handle:
            raise RuntimeError("fedfr_block_create failed: " + _C.last_error())
        self.cin, self.cout, self.stride, self.hin, self.hout, self.batch = inplanes, planes, stride, hin, hin // stride, batch
        self.device = torch.device(device)
        self.counts, self.table = _read_table(self.handle)
        c = self.counts
        self.params = torch.zeros(c[_C.Q_PARAM_COUNT], dtype=torch.float32, device=device)
        self.grads = torch.zeros(c[_C.Q_PARAM_COUNT], dtype=torch.float32, device=device)
        self.bufs = torch.zeros(c[_C.Q_BUFFER_COUNT], dtype=torch.float32, device=device)
        self.nbt = torch.zeros(c[_C.Q_NBT_COUNT], dtype=torch.int64, device=device)
        self.shadow = torch.empty(c[_C.Q_SHADOW_COUNT], dtype=_C.storage_dtype(), device=device)
        self.act = torch.empty(c[_C.Q_ACT_BYTES], dtype=torch.uint8, device=device)
        self.ws = torch.empty(c[_C.Q_WS_BYTES], dtype=torch.uint8, device=device)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _C.lib().fedfr_net_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def _view(self, flat, kind, off, shape):
        if kind == KIND_CONV:
            o, i, r, _ = shape
            return flat[off: off + o * i * r * r].view(o, r, r, i).permute(0, 3, 1, 2)
        n = 1
        for s in shape:
            n *= s
        return flat[off: off + n].view(shape)

    def load_state_dict(self, sd):
        for name, kind, region, off, shape in self.table:
            v = sd[name].to(self.device)
            if region == 0:
                self._view(self.params, kind, off, shape).copy_(v)
            elif region == 1:
                self.bufs[off: off + shape[0]].copy_(v)
            else:
                self.nbt[off] = int(v)

    def state_dict(self):
        out = OrderedDict()
        for name, kind, region, off, shape in self.table:
            if region == 0:
                out[name] = self._view(self.params, kind, off, shape).clone()
            elif region == 1:
                out[name] = self.bufs[off: off + shape[0]].clone()
            else:
                out[name] = self.nbt[off].clone()
        return out

    def grad_dict(self):
        return OrderedDict((name, self._view(self.grads, kind, off, shape)) for name, kind, region, off, shape in self.table if region == 0)

    def _act(self, which):
        off, rows, ch = C.c_longlong(), C.c_int(), C.c_int()
        _C.call("fedfr_net_act_info", self.handle, 0, which, C.byref(off), C.byref(rows), C.byref(ch))
        a = self.act[off.value * 2: (off.value + rows.value * ch.value) * 2].view(_C.storage_dtype()).view(rows.value, ch.value)
        h = int(round((rows.value // self.batch) ** 0.5))
        return a.float().view(self.batch, h, h, ch.value).permute(0, 3, 1, 2).contiguous()

    def forward(self, x, training=True):
        x = _C.require_gpu_tensor(x.contiguous(), torch.float32, "block input")
        if tuple(x.shape) != (self.batch, self.cin, self.hin, self.hin):
            raise RuntimeError("BlockPlan expects %s, got %s" % ((self.batch, self.cin, self.hin, self.hin), tuple(x.shape)))
        st = _C.stream()
        _C.call("fedfr_net_prepare_weights", self.handle, self.params.data_ptr(), self.shadow.data_ptr(), 1, st)
        _C.call("fedfr_net_forward", self.handle, x.data_ptr(), self.params.data_ptr(), self.bufs.data_ptr(), self.shadow.data_ptr(),
                self.act.data_ptr(), self.ws.data_ptr(), None, 1 if training else 0, st)
        if training:
            self.nbt += 1
        return self._act(6)

    def backward(self, dy, aux_stream=None):
        dy = _C.require_gpu_tensor(dy.contiguous(), torch.float32, "block output gradient")
        if tuple(dy.shape) != (self.batch, self.cout, self.hout, self.hout):
            raise RuntimeError("BlockPlan.backward expects %s, got %s" % ((self.batch, self.cout, self.hout, self.hout), tuple(dy.shape)))
        _C.call("fedfr_net_backward2", self.handle, None, dy.data_ptr(), self.params.data_ptr(), self.shadow.data_ptr(),
                self.act.data_ptr(), self.ws.data_ptr(), self.grads.data_ptr(), _C.stream(),
                aux_stream.cuda_stream if aux_stream is not None else None)
        return self._act(7)


def _tensor_table(layers, in_hw, nfeat, sphere_type=0):
    """state_dict layout from the C plan (batch-independent part)."""
    l = _C.lib()
    if sphere_type:
        h = l.fedfr_net_create_sphere(sphere_type, 8)
    else:
        arr = (C.c_int * 4)(*layers)
        h = l.fedfr_net_create(arr, 8, in_hw, nfeat)
    if not h:
        raise RuntimeError("fedfr_net_create failed: " + _C.last_error())
    try:
        q = C.c_longlong()
        counts = {}
        for key in (_C.Q_PARAM_COUNT, _C.Q_TRAINABLE_COUNT, _C.Q_BUFFER_COUNT, _C.Q_NBT_COUNT, _C.Q_SHADOW_COUNT,
                    _C.Q_NUM_TENSORS, _C.Q_FC_IN):
            _C.call("fedfr_net_query", h, key, C.byref(q))
            counts[key] = q.value
        table = []
        name = C.create_string_buffer(128)
        kind, region, ndim = C.c_int(), C.c_int(), C.c_int()
        off = C.c_longlong()
        shape = (C.c_int * 4)()
        for i in range(counts[_C.Q_NUM_TENSORS]):
            _C.call("fedfr_net_tensor_info", h, i, name, 128, C.byref(kind), C.byref(region), C.byref(off), C.byref(ndim), shape)
            table.append((name.value.decode(), kind.value, region.value, off.value, tuple(shape[: ndim.value])))
        return counts, table
    finally:
        l.fedfr_net_destroy(h)


class _IResNetFn(torch.autograd.Function):
    """autograd bridge: forward = fedfr_net_forward, backward = fedfr_net_backward (hand-written adjoint)."""

    @staticmethod
    def forward(ctx, x, anchor, net):
        feats = net._run_forward(x, training=True)
        ctx.net = net
        ctx.x = x
        ctx.plan = net._plan(x.shape[0])
        ctx.gen = net._fwd_generation
        return feats

    @staticmethod
    def backward(ctx, dfeats):
        net = ctx.net
        if ctx.gen != net._fwd_generation:
            raise RuntimeError("fedfr_amd: backward() after a newer forward() on the same IResNet — saved "
                               "activations live in a per-module arena and were overwritten")
        net._run_backward(ctx.plan, ctx.x, dfeats.contiguous())
        return None, None, None


class IResNet(nn.Module):
    fc_scale = 7 * 7
    _instances = 0          # models created in this process (default dropout seeds differ per model)

    def __init__(self, block, layers, dropout=0, num_features=512, zero_init_residual=False, groups=1,
                 width_per_group=64, replace_stride_with_dilation=None, fp16=False):
        super().__init__()
        if groups != 1 or width_per_group != 64:
            raise ValueError("BasicBlock only supports groups=1 and base_width=64")       # iresnet.py:33-34
        if replace_stride_with_dilation is not None and any(replace_stride_with_dilation):
            raise NotImplementedError("Dilation > 1 not supported in BasicBlock")          # iresnet.py:35-36
        self.fp16 = fp16          # accepted for signature parity; this backend always stores 16-bit (fp16 in the product library, bf16 in libfedfr_hip_bf16.so) / accumulates fp32
        self.layers_cfg = tuple(int(v) for v in layers)
        self.num_features = num_features
        self.in_hw = 112
        self.dropout_p = float(dropout)
        # seed of the counter-based dropout mask: the reference's RNG seed (train.py:35) for the first model of the process, a different
        # one for every later model — the reference's clients draw from ONE global RNG stream, so no two of them ever see the same masks;
        # the step counter lives HERE (not on the per-batch-size plan, which is re-created after an eviction / release_workspace)
        self.dropout_seed = 100 + 7919 * IResNet._instances
        IResNet._instances += 1
        self._dropout_step = 0
        counts, table = _tensor_table(self.layers_cfg, self.in_hw, num_features)
        self._counts, self._table = counts, table
        # flat storage (CPU until .to(device), like any nn.Module).  Parameters and BN running statistics are slices of ONE fp32
        # tensor, followed by a float image of the num_batches_tracked counters: the FedAvg exchange of a round is then a single
        # in-place all-reduce of `_flat_state` (server.fedavg_all_reduce), with no packing copies
        self._flat_state = torch.zeros(self._state_len(counts), dtype=torch.float32)
        self._slice_state()
        self._flat_nbt = torch.zeros(counts[_C.Q_NBT_COUNT], dtype=torch.int64)
        self._flat_grads: Optional[torch.Tensor] = None
        self._shadow: Optional[torch.Tensor] = None
        self._plans: Dict[int, _Plan] = {}
        self._shadow_dirty = True
        self._grads_live = False
        self._fwd_generation = 0
        self._bn_frozen = False                 # freeze_BN(test_mode=True): BatchNorms in eval mode inside a training net
        # True: forward / backward run the fp32 VALIDATION path (csrc/net_f32.hip: fp32 activations, exact-fp32 GEMMs, fp64 statistics;
        # ~50x slower) — same parameters, buffers and gradients; what the "1e-3 fp32" tolerance of the parity tests is checked with
        self.validation_fp32 = False
        self._anchor = torch.zeros(1, requires_grad=True)     # keeps the autograd graph connected
        # module tree with the reference's names (iresnet.py:76-98)
        self.conv1 = Conv2d(3, 64, 3, 1)
        self.bn1 = BatchNorm2d(64)
        self.prelu = PReLU(64)
        inpl = 64
        for si, (planes, nblk) in enumerate(zip((64, 128, 256, 512), self.layers_cfg)):
            blocks = []
            for bi in range(nblk):
                cin = inpl if bi == 0 else planes
                ds = Downsample(Conv2d(cin, planes, 1, 2), BatchNorm2d(planes)) if bi == 0 else None
                blocks.append(IBasicBlock(cin, planes, 2 if bi == 0 else 1, ds))
            setattr(self, "layer%d" % (si + 1), nn.Sequential(*blocks))
            inpl = planes
        self.bn2 = BatchNorm2d(512)
        self.dropout = nn.Dropout(p=dropout, inplace=True)
        self.fc = Linear(512 * self.fc_scale, num_features)
        self.features = BatchNorm1d(num_features)
        self._bind(create=True)
        self._init_weights(zero_init_residual)

    # ------------------------------------------------------------------ storage plumbing
    @staticmethod
    def _state_len(counts):
        return counts[_C.Q_PARAM_COUNT] + counts[_C.Q_BUFFER_COUNT] + (counts[_C.Q_NBT_COUNT] + 3) // 4 * 4

    def _slice_state(self):
        P, Bc, N = self._counts[_C.Q_PARAM_COUNT], self._counts[_C.Q_BUFFER_COUNT], self._counts[_C.Q_NBT_COUNT]
        self._flat_params = self._flat_state[:P]
        self._flat_bufs = self._flat_state[P: P + Bc]
        self._nbt_f32 = self._flat_state[P + Bc: P + Bc + N]          # scratch: float image of num_batches_tracked during an exchange

    def _views(self):
        """yield (owner module, attr, view tensor, is_param, kind) for every state_dict entry."""
        mods = dict(self.named_modules())
        for name, kind, region, off, shape in self._table:
            mod_name, attr = name.rsplit(".", 1)
            owner = mods[mod_name]
            if region == 0:
                if kind == KIND_CONV:
                    o, i, r, _ = shape
                    v = self._flat_params[off: off + o * i * r * r].view(o, r, r, i).permute(0, 3, 1, 2)
                else:
                    n = 1
                    for s in shape:
                        n *= s
                    v = self._flat_params[off: off + n].view(shape)
                yield owner, attr, v, True, kind, name
            elif region == 1:
                yield owner, attr, self._flat_bufs[off: off + shape[0]], False, kind, name
            else:
                yield owner, attr, self._flat_nbt[off], False, kind, name

    def _bind(self, create=False):
        for owner, attr, v, is_param, kind, name in self._views():
            if is_param:
                if create:
                    owner.register_parameter(attr, nn.Parameter(v, requires_grad=(name != "features.weight")))
                else:
                    getattr(owner, attr).data = v
            else:
                if create:
                    owner.register_buffer(attr, v)
                else:
                    owner._buffers[attr] = v
        self._grad_views = None

    def _init_weights(self, zero_init_residual):
        """Reference init (iresnet.py:97-112): conv N(0, .1); BN 1/0; fc = nn.Linear default; PReLU .25."""
        with torch.no_grad():
            for owner, attr, v, is_param, kind, name in self._views():
                if kind == KIND_CONV:
                    v.normal_(0, 0.1)
                elif kind == KIND_BNW or kind == KIND_RV:
                    v.fill_(1.0)
                elif kind == KIND_PRELU:
                    v.fill_(0.25)
                elif kind == KIND_FCW or kind == KIND_FCB:
                    bound = 1.0 / (512 * self.fc_scale) ** 0.5
                    v.uniform_(-bound, bound)
                elif kind in (KIND_BNB, KIND_RM):
                    v.zero_()
            if zero_init_residual:
                for m in self.modules():
                    if isinstance(m, IBasicBlock):
                        m.bn2.weight.zero_()          # bn2, as the reference does (iresnet.py:109-112)

    def _apply(self, fn, recurse=True):
        new = fn(self._flat_state)
        if new.dtype != torch.float32:
            raise RuntimeError("fedfr_amd.IResNet keeps fp32 master weights (bf16 compute copies are internal); "
                               "dtype conversion to %s is not supported" % new.dtype)
        self._flat_state = new
        self._slice_state()
        self._flat_nbt = fn(self._flat_nbt) if not self._flat_nbt.is_floating_point() else self._flat_nbt
        if self._flat_nbt.device != self._flat_params.device:
            self._flat_nbt = self._flat_nbt.to(self._flat_params.device)
        self._anchor = torch.zeros(1, device=self._flat_params.device, requires_grad=True)
        self._flat_grads = None
        self._shadow = None
        self._plans = {}
        self._shadow_dirty = True
        self._grads_live = False
        self._bind(create=False)
        return self

    def __getstate__(self):
        st = self.__dict__.copy()
        for k in ("_flat_params", "_flat_bufs", "_nbt_f32"):      # slices of _flat_state: rebuilt, never copied on their own
            st.pop(k, None)
        st["_plans"] = {}
        st["_shadow"] = None
        st["_flat_grads"] = None
        st["_grad_views"] = None
        st["_shadow_dirty"] = True
        st["_grads_live"] = False
        return st

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__getstate__().items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        new._slice_state()
        new._bind(create=False)
        return new

    def __setstate__(self, st):
        super().__setstate__(st)
        self._slice_state()
        self._bind(create=False)

    def load_state_dict(self, state_dict, strict=True, **kw):
        # a FlatStateDict nobody has looked into (views never built, so no entry can have been replaced) of the same architecture: three
        # flat copies instead of 925 (FedAvg's hand-back, server.py:330-335; int64 counters take the float average truncated, F9)
        flat = getattr(state_dict, "flat", None)
        if (flat is not None and getattr(state_dict, "_built", True) is False and list(getattr(state_dict, "layers", ())) == list(self.layers_cfg)
                and flat[0].numel() == self._flat_params.numel() and flat[1].numel() == self._flat_bufs.numel()
                and flat[2].numel() == self._flat_nbt.numel() and flat[0].dtype == torch.float32):
            with torch.no_grad():
                self._flat_params.copy_(flat[0])
                self._flat_bufs.copy_(flat[1])
                self._flat_nbt.copy_(flat[2])
            self._shadow_dirty = True
            from torch.nn.modules.module import _IncompatibleKeys
            return _IncompatibleKeys([], [])
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._shadow_dirty = True
        return out

    def release_workspace(self):
        """Drop everything but the weights: activation arenas / workspaces of every batch size, bf16 shadows, the flat gradient buffer
        (GBs for iresnet100); they are rebuilt on the next forward.  For models that stay resident but idle between rounds."""
        self._plans = {}
        self._shadow = None
        self._flat_grads = None
        self._grad_views = None
        self._shadow_dirty = True
        self._grads_live = False
        for p in self.parameters():
            p.grad = None

    def mark_weights_dirty(self):
        """Call after modifying parameters in place outside of this package's optimiser while in eval mode."""
        self._shadow_dirty = True

    # reference API (iresnet.py:140-156): put every BatchNorm module into eval() — running statistics normalise, nothing is updated —
    # while the net keeps training (dropout on, gradients flow; dgamma / dbeta are still produced unless fix_affine).  As in the
    # reference, only model.train() brings the BatchNorms back (nn.Module.train() resets every submodule); unfreeze_BN() just restores
    # requires_grad and, with test_mode=True, calls .eval() on them again (iresnet.py:149-156).
    def freeze_BN(self, test_mode=True, fix_affine=False):
        if fix_affine:
            for m in self.modules():
                if m.__class__.__name__.find("BatchNorm") != -1:
                    for p in m.parameters():
                        p.requires_grad = False
        if test_mode:
            self._bn_frozen = True
            for m in self.modules():
                if m.__class__.__name__.find("BatchNorm") != -1:
                    m.training = False

    def unfreeze_BN(self, test_mode=False, affine=True):
        if affine:
            for n, m in self.named_modules():
                if m.__class__.__name__.find("BatchNorm") != -1:
                    for pn, p in m.named_parameters():
                        p.requires_grad = True
        if test_mode:
            self.freeze_BN(True, False)

    def train(self, mode: bool = True):
        self._bn_frozen = False                 # nn.Module.train() sets .training on every submodule, BatchNorms included
        return super().train(mode)

    def _f32_buffers(self, plan):
        """Activation arena + scratch of the fp32 validation path for this plan (allocated on first use; released with the plan)."""
        if getattr(plan, "f32_arena", None) is None:
            na, nw = _C.lib().fedfr_net_f32_arena_floats(plan.handle), _C.lib().fedfr_net_f32_ws_floats(plan.handle)
            plan.f32_arena = torch.empty(na, dtype=torch.float32, device=self.device)
            plan.f32_ws = torch.empty(nw, dtype=torch.float32, device=self.device)
        return plan.f32_arena, plan.f32_ws

    def _pre_forward(self, plan, mode: int):
        """Before every fedfr_net_forward on ``plan``: hand the model's dropout seed / mask index to the C-side plan, so that the mask
        sequence survives plan re-creation and differs between models (the plan's own counter restarts at 0 with every new plan)."""
        if mode and self.dropout_p > 0:
            if plan.dropout_seed != self.dropout_seed:
                off = C.c_longlong()
                _C.call("fedfr_net_set_dropout", plan.handle, self.dropout_p, self.dropout_seed, C.byref(off))
                plan.mask_off, plan.dropout_seed = off.value, self.dropout_seed
            _C.call("fedfr_net_set_dropout_step", plan.handle, self._dropout_step)
            self._dropout_step += 1

    def require_all_trainable(self, who: str):
        """The fused trainers update the whole trainable region with ONE flat SGD kernel: a parameter frozen with requires_grad = False
        (``freeze_BN(fix_affine=True)``) would still receive gradient, momentum and weight decay there — refuse instead of ignoring it."""
        frozen = [n for n, p_ in self.named_parameters() if not p_.requires_grad and n != "features.weight"]
        if frozen:
            raise NotImplementedError("fedfr_amd.%s updates every backbone parameter with one flat SGD kernel and cannot honour requires_grad = "
                                      "False on %d parameter(s) (%s, ...): train such a model through the autograd path (IResNet.forward + "
                                      "torch.optim.SGD), or unfreeze them" % (who, len(frozen), frozen[0]))

    def _fwd_mode(self) -> int:
        """``training`` argument of fedfr_net_forward: 0 eval, 1 train, 2 train with the BatchNorms frozen in eval mode."""
        if not self.training:
            return 0
        return 2 if self._bn_frozen else 1

    # ------------------------------------------------------------------ device side
    @property
    def device(self):
        return self._flat_params.device

    def _plan(self, batch) -> _Plan:
        p = self._plans.get(batch)
        if p is None:
            if len(self._plans) >= 2:       # arenas are GBs: keep at most two batch sizes alive
                self._plans.pop(next(iter(self._plans)))
            p = _Plan(self.layers_cfg, batch, self.in_hw, self.num_features, self.device, getattr(self, "sphere_type", 0))
            if self.dropout_p > 0:
                off = C.c_longlong()
                _C.call("fedfr_net_set_dropout", p.handle, self.dropout_p, self.dropout_seed, C.byref(off))
                p.mask_off = off.value
                p.dropout_seed = self.dropout_seed
            self._plans[batch] = p
        return p

    def _ensure_device_state(self):
        if not self._flat_params.is_cuda:
            raise RuntimeError("fedfr_amd.IResNet runs only on an MI355X device: move the module with .to('cuda') "
                               "(there is no CPU / PyTorch fallback path)")
        if self._shadow is None:
            self._shadow = torch.empty(self._counts[_C.Q_SHADOW_COUNT], dtype=_C.storage_dtype(), device=self.device)
            self._shadow_dirty = True
        if self._flat_grads is None:
            self._flat_grads = torch.zeros(self._counts[_C.Q_PARAM_COUNT], dtype=torch.float32, device=self.device)
            self._grad_views = None

    @_C.on_device(lambda self: self._flat_params.device)
    def refresh_shadows(self, fwd_shadow_too=True):
        self._ensure_device_state()
        plan = next(iter(self._plans.values())) if self._plans else self._plan(8)
        _C.call("fedfr_net_prepare_weights", plan.handle, self._flat_params.data_ptr(), self._shadow.data_ptr(),
                1 if fwd_shadow_too else 0, _C.stream())
        self._shadow_dirty = False

    def _check_input(self, x):
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != self.in_hw or x.shape[3] != self.in_hw:
            raise RuntimeError("fedfr_amd.IResNet expects [B,3,%d,%d] input, got %s" % (self.in_hw, self.in_hw, tuple(x.shape)))
        _C.require_gpu_tensor(x, torch.float32, "input batch")
        if x.device != self.device:
            raise RuntimeError("input on %s but model on %s" % (x.device, self.device))

    @_C.on_device(lambda self: self._flat_params.device)
    def _run_forward(self, x, training: bool):
        self._ensure_device_state()
        plan = self._plan(x.shape[0])
        if training or self._shadow_dirty:
            self.refresh_shadows(True)
        feats = torch.empty(x.shape[0], self.num_features, dtype=torch.float32, device=self.device)
        mode = (2 if self._bn_frozen else 1) if training else 0
        if self.validation_fp32:
            if mode == 2 or (training and self.dropout_p > 0):
                raise NotImplementedError("fedfr_amd: the fp32 validation path covers train / eval forward + backward with dropout 0")
            arena, ws = self._f32_buffers(plan)
            _C.call("fedfr_net_f32_forward", plan.handle, x.data_ptr(), self._flat_params.data_ptr(), self._flat_bufs.data_ptr(),
                    arena.data_ptr(), ws.data_ptr(), feats.data_ptr(), mode, _C.stream())
            if mode == 1:
                self._flat_nbt += 1
            self._fwd_generation += 1
            return feats
        self._pre_forward(plan, mode)
        _C.call("fedfr_net_forward", plan.handle, x.data_ptr(), self._flat_params.data_ptr(), self._flat_bufs.data_ptr(),
                self._shadow.data_ptr(), plan.act.data_ptr(), plan.ws.data_ptr(), feats.data_ptr(), mode, _C.stream())
        if mode == 1:                           # frozen BatchNorms track nothing (num_batches_tracked included)
            self._flat_nbt += 1
        self._fwd_generation += 1
        return feats

    def _grad_view_list(self):
        if self._grad_views is None:
            mods = dict(self.named_modules())
            views = []
            tc = self._counts[_C.Q_TRAINABLE_COUNT]
            for name, kind, region, off, shape in self._table:
                if region != 0 or off >= tc:
                    continue
                mod_name, attr = name.rsplit(".", 1)
                p = getattr(mods[mod_name], attr)
                if kind == KIND_CONV:
                    o, i, r, _ = shape
                    g = self._flat_grads[off: off + o * i * r * r].view(o, r, r, i).permute(0, 3, 1, 2)
                else:
                    n = 1
                    for s in shape:
                        n *= s
                    g = self._flat_grads[off: off + n].view(shape)
                views.append((p, g))
            self._grad_views = views
        return self._grad_views

    @_C.on_device(lambda self: self._flat_params.device)
    def _run_backward(self, plan, x, dfeats):
        """Writes parameter gradients into the flat grad buffer and exposes them as ``p.grad`` views
        (accumulating if the caller did not zero them — torch.optim semantics)."""
        views = self._grad_view_list()
        accumulate = self._grads_live and any(p.grad is not None for p, _ in views[:1])
        target = self._flat_grads
        if accumulate:
            target = torch.empty_like(self._flat_grads)
        if self.validation_fp32:
            arena, ws = self._f32_buffers(plan)
            _C.call("fedfr_net_f32_backward", plan.handle, dfeats.data_ptr(), self._flat_params.data_ptr(), arena.data_ptr(), ws.data_ptr(),
                    target.data_ptr(), _C.stream())
        else:
            S = _C.loss_scale_state(self._flat_params.device).scale      # fp16-storage library: the device's loss scale (1 for the bf16 library)
            if S != 1.0:
                dfeats = dfeats * S
            _C.call("fedfr_net_backward", plan.handle, x.data_ptr(), dfeats.data_ptr(), self._flat_params.data_ptr(),
                    self._shadow.data_ptr(), plan.act.data_ptr(), plan.ws.data_ptr(), target.data_ptr(), _C.stream())
            if S != 1.0:
                target.mul_(1.0 / S)
        if accumulate:
            self._flat_grads.add_(target)
        for p, g in views:
            if p.requires_grad:
                p.grad = g
        self._grads_live = True

    @_C.on_device(lambda self: self._flat_params.device)
    def forward(self, x):
        self._check_input(x)
        x = x.contiguous()
        if self.training and torch.is_grad_enabled():
            if self._anchor.device != x.device:
                self._anchor = torch.zeros(1, device=x.device, requires_grad=True)
            return _IResNetFn.apply(x, self._anchor, self)
        return self._run_forward(x, training=self.training)

    # flat accessors used by the fused trainer / FedAvg
    def flat_state(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        return self._flat_params, self._flat_bufs, self._flat_nbt

    def exchange_buffer(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """(whole fp32 state [params | running stats | float image of num_batches_tracked (+pad)], that float image, the int64
        counters): the one tensor a FedAvg round all-reduces in place."""
        return self._flat_state, self._nbt_f32, self._flat_nbt

    def trainable_count(self) -> int:
        return self._counts[_C.Q_TRAINABLE_COUNT]


def _iresnet(arch, block, layers, pretrained, progress, **kwargs):
    model = IResNet(block, layers, **kwargs)
    if pretrained:
        raise ValueError()          # reference iresnet.py:176-178
    return model


def iresnet18(pretrained=False, progress=True, **kwargs):
    return _iresnet("iresnet18", IBasicBlock, [2, 2, 2, 2], pretrained, progress, **kwargs)


def iresnet34(pretrained=False, progress=True, **kwargs):
    return _iresnet("iresnet34", IBasicBlock, [3, 4, 6, 3], pretrained, progress, **kwargs)


def iresnet50(pretrained=False, progress=True, **kwargs):
    return _iresnet("iresnet50", IBasicBlock, [3, 4, 14, 3], pretrained, progress, **kwargs)


def iresnet100(pretrained=False, progress=True, **kwargs):
    return _iresnet("iresnet100", IBasicBlock, [3, 13, 30, 3], pretrained, progress, **kwargs)


def iresnet200(pretrained=False, progress=True, **kwargs):
    return _iresnet("iresnet200", IBasicBlock, [6, 26, 60, 6], pretrained, progress, **kwargs)
