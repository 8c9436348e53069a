"""Client side of the federated loop on MI355X — mirrors the reference's ``client.py`` surface:
``FC_module``, ``BCE_module``, ``Sequential_model``, ``Branch_model`` (client.py:25-113) and ``Client`` with
``train`` / ``get_model`` / ``get_data_size`` / ``get_train_loss`` (client.py:116-157, :511-582).

The hot loop (reference client.py:536-551: zero_grad → fwd → margin → CE → bwd → SGD step) runs through
``FusedTrainer``: one ``fedfr_net_forward``, the fp32 head kernels, one ``fedfr_net_backward`` and the flat
``fedfr_sgd_step`` — no per-layer Python, no PyTorch autograd on the hot path.
"""
from __future__ import annotations

import logging
from collections import OrderedDict
from typing import Iterable, Optional, Tuple

import torch
from torch import nn

from . import _C, backbones, losses, ops
from .config import config as cfg

f32 = torch.float32


# ------------------------------------------------------------------------------------------------
# heads
# ------------------------------------------------------------------------------------------------
class FC_module(nn.Module):
    """Dense cosine classifier (reference client.py:63-83)."""

    def __init__(self, hidden, n_class, output_dir):
        super().__init__()
        self.fc = nn.Parameter(torch.normal(0, 0.01, (n_class, hidden)))
        self.output_dir = output_dir
        self.n_class = n_class

    def forward(self, x, normalize_feat=True):
        return ops.cosine_linear(x, self.fc, normalize_feat)

    def update_from_tensor(self, fc):
        self.fc.data = fc.clone()

    def update_with_pretrain(self, pretrain_fc):
        self.fc = nn.Parameter(torch.cat([self.fc.data, pretrain_fc.to(self.fc.device)], dim=0))

    def remove_pretrain(self):
        self.fc.data = self.fc.data[0:self.n_class]

    def get_pretrain_fc(self):
        return self.fc.data[self.n_class:]


class _BceLogitsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cosine, labels, bias, m, r, t):
        cosine = ops._chk(cosine.detach(), "cosine")
        labels = ops._chk(labels, "labels", torch.int64)
        B, C = cosine.shape
        z = torch.empty_like(cosine)
        gt = torch.empty(B, C, dtype=torch.bool, device=cosine.device)
        dzdcos = torch.empty_like(cosine)
        _C.call("fedfr_bce_logits", cosine.data_ptr(), labels.data_ptr(), bias.detach().data_ptr(), B, C, m, r, float(t),
                z.data_ptr(), gt.data_ptr(), dzdcos.data_ptr(), _C.stream())
        ctx.save_for_backward(dzdcos)
        ctx.mark_non_differentiable(gt)
        return z, gt

    @staticmethod
    def backward(ctx, dz, _dgt):
        (dzdcos,) = ctx.saved_tensors
        dz = dz.contiguous()
        return dz * dzdcos, None, ops.colsum(dz), None, None, None


class _BceLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, gt, r, lam):
        z = ops._chk(z.detach(), "bce logits")
        gt = ops._chk(gt, "gts", torch.bool)
        B, C = z.shape
        dz = torch.empty_like(z)
        row_loss = torch.empty(B, dtype=f32, device=z.device)
        _C.call("fedfr_bce_loss", z.data_ptr(), gt.data_ptr(), None, B, C, r, lam, 1.0, dz.data_ptr(), None, row_loss.data_ptr(),
                _C.stream())
        loss = torch.empty((), dtype=f32, device=z.device)
        _C.call("fedfr_sum_scale", row_loss.data_ptr(), B, 1.0 / B, loss.data_ptr(), _C.stream())
        ctx.save_for_backward(dz)
        return loss

    @staticmethod
    def backward(ctx, dl):
        (dz,) = ctx.saved_tensors
        return dz * dl, None, None, None


def bce_loss_from_logits(z, gt, r, lam):
    return _BceLossFn.apply(z, gt, float(r), float(lam))


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b in exact fp32 (the 512x512 'converter' of the personalised head, client.py:29-33)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w = ops._chk(x.detach(), "x"), ops._chk(w.detach(), "w")
        ctx.save_for_backward(x, w)
        return ops.sgemm(x, w, trans_b=True, bias=b.detach())

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = ops._chk(dy, "dy")
        return ops.sgemm(dy, w), ops.sgemm(dy, x, trans_a=True), ops.colsum(dy)


class _Converter(nn.Module):
    def __init__(self, hidden):
        super().__init__()
        self.weight = nn.Parameter(torch.eye(hidden))
        self.bias = nn.Parameter(torch.zeros(hidden))

    def forward(self, x):
        return _LinearFn.apply(x, self.weight, self.bias)


class BCE_module(nn.Module):
    """Personalised transform head (reference client.py:25-60), converter_layer == 1 only (config.py:31)."""

    def __init__(self, hidden, n_class, converter_layer=1, m=0.4, r=30.0, t=3):
        super().__init__()
        if converter_layer != 1:
            raise NotImplementedError("fedfr_amd: BottleBlock converter (converter_layer != 1) is out of scope (SURVEY §2.1)")
        self.converter = nn.Sequential(_Converter(hidden))      # keeps the reference key 'converter.0.weight'
        self.weight = nn.Parameter(torch.normal(0, 0.01, (n_class, hidden)))
        self.bias = nn.Parameter(torch.zeros(n_class))
        self.n_class, self.hidden, self.m, self.r, self.t = n_class, hidden, m, r, t

    def forward(self, x, labels):
        feat = self.converter(x)
        cosine = ops.cosine_linear(feat, self.weight)
        z, gt = _BceLogitsFn.apply(cosine, labels, self.bias, float(self.m), float(self.r), self.t)
        return z, gt

    def initialize(self, fc):
        self.weight.data = fc.clone()


class Branch_model(nn.Module):
    """reference client.py:85-100."""

    def __init__(self, backbone, fc_module, bce_module):
        super().__init__()
        self.backbone, self.fc_module, self.bce_module = backbone, fc_module, bce_module

    def forward(self, imgs, labels, contrastive=False, detach=False):
        feature = self.backbone(imgs)
        cosface_logits = self.fc_module(feature)
        bce_logits, bce_gts = self.bce_module(feature.detach() if detach else feature, labels)
        if contrastive:
            return cosface_logits, bce_logits, bce_gts, feature
        return cosface_logits, bce_logits, bce_gts


class Sequential_model(nn.Module):
    """reference client.py:102-113."""

    def __init__(self, backbone, fc_module):
        super().__init__()
        self.backbone, self.fc_module = backbone, fc_module

    def forward(self, imgs, contrastive=False):
        feature = self.backbone(imgs)
        logits = self.fc_module(feature)
        return (logits, feature) if contrastive else logits


# ------------------------------------------------------------------------------------------------
# fused train step
# ------------------------------------------------------------------------------------------------
_AUX_STREAMS = {}      # device index -> the one auxiliary stream every trainer on that device shares (created once, lives with the process)
_AUX_LOCK = __import__("threading").Lock()


def _make_aux_stream(device, slot: int = 0):
    """Second HIP stream of the dual-stream backward (None with FEDFR_DUAL_STREAM=0): created at the LOWEST priority the device offers
    (through the C ABI: torch clamps stream priorities to [-1, 0], HIP has +1), so that workgroups of the critical path on the caller's
    stream are dispatched first whenever both streams have work.  FEDFR_AUX_PRIORITY=0 keeps a default-priority torch stream.
    ONE stream per (device, slot) for the life of the process: trainers are re-created every FL round for every client (the reference
    re-creates its optimiser the same way), a stream per trainer would leak a HIP stream each time.  ``slot`` > 0: further streams
    for clients that train CONCURRENTLY on one device (one slot per concurrent client)."""
    import os
    if os.environ.get("FEDFR_DUAL_STREAM", "1") == "0":
        return None
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    with _AUX_LOCK:
        return _aux_stream_locked(idx, int(slot))


def _aux_stream_locked(idx, slot):
    import os
    key = idx if slot == 0 else (idx, slot)
    s = _AUX_STREAMS.get(key)
    if s is None:
        if os.environ.get("FEDFR_AUX_PRIORITY", "1") != "0":
            import ctypes
            h = ctypes.c_void_p()
            with torch.cuda.device(idx):
                _C.call("fedfr_stream_create_low_priority", ctypes.byref(h))
            s = torch.cuda.ExternalStream(h.value, device=torch.device("cuda", idx))
        else:
            s = torch.cuda.Stream(device=torch.device("cuda", idx))
        _AUX_STREAMS[key] = s
    return s


_HEAD_OFF_PATH = __import__("os").environ.get("FEDFR_HEAD_OFF_PATH", "1") != "0"
_HEAD_SPLITK = __import__("os").environ.get("FEDFR_HEAD_SPLITK", "1") != "0"


def _split_for(k: int) -> int:
    """split-K degree of a head GEMM: k ranges of about 128 (four k-steps of 32 per workgroup), at most 8.

    The kernel side (``head_sgemm_splitk``, csrc/head.hip) cuts k into chunks of ceil32(ceil(k / splits)) and refuses a split count whose
    last chunk would be empty (k = 800 with 6 splits: 5 chunks of 160 already cover it), so the count is derived from the chunk."""
    s = max(1, min(8, k // 128))
    chunk = -(-(-(-k // s)) // 32) * 32
    return -(-k // chunk)


class _LossScaleGuard:
    """fp16-storage library (``_C.loss_scale() != 1``): a trainer's view of its device's loss scale (``_C.LossScaleState``: it outlives the
    trainer, so a back-off carries over to the next FL round) and of the device word the update kernels set when they skipped a non-finite
    gradient element (fedfr_sgd_step_scaled).  The word is read in ``check_overflow()``: in ``finish()``, every ``overflow_check_every`` steps
    and — through ``Client.train*`` — wherever the host drains a loss value anyway (every step at the reference's ``loss.item()`` cadence).
    An overflow halves the scale and warns; ``growth_interval`` clean steps double it again up to its initial value: torch.cuda.amp.GradScaler's
    schedule (client.py:301,394-396) without a host sync per step.  What differs from GradScaler is stated in INTEGRATION.md: the skip is per
    ELEMENT (a parameter element whose gradient is not finite keeps its value, momentum and mirror), not per step."""

    def _init_loss_scale(self, device):
        self._ls = _C.loss_scale_state(device)
        self.overflow_check_every = 100
        self.overflows = 0                       # overflowing intervals THIS trainer has seen (the device total: self._ls.overflows)
        self._steps_since_check = 0

    @property
    def loss_scale(self) -> float:
        return self._ls.scale

    @loss_scale.setter
    def loss_scale(self, v: float):
        self._ls.scale = float(v)

    @property
    def guarded(self) -> bool:
        """True for the fp16-storage library: every parameter update goes through the overflow-guarded kernel, whatever the scale is now."""
        return self._ls.enabled

    @property
    def _overflow(self) -> torch.Tensor:
        return self._ls.word

    def _count_step(self):
        self._ls.count_step()
        self._steps_since_check += 1
        if self._steps_since_check >= self.overflow_check_every:
            self.check_overflow()

    def check_overflow(self) -> bool:
        """Did an update kernel skip non-finite gradient elements since the last check?  (bf16 library: no, nothing is read; fp16 library:
        synchronises the stream.)  If so the loss scale is halved for the following steps and a warning is issued."""
        self._steps_since_check = 0
        hit = self._ls.poll()
        if hit:
            self.overflows += 1
        return hit


class FusedTrainer(_LossScaleGuard):
    """One optimiser lifetime (= one FL round for one client: the reference re-creates SGD every round, F8).

    step(imgs, labels) == the body of the reference hot loop (client.py:537-550) for the Sequential model:
    zero_grad; logits = fc(backbone(imgs)); logits = margin(logits, labels); loss = CE; backward; SGD step.
    """

    def __init__(self, backbone: "backbones.IResNet", fc, loss_name: str = "CosFace", s: float = 30.0,
                 m: float = 0.4, lr: float = 0.1, momentum: float = 0.9, weight_decay: float = 5e-4, aux_slot: int = 0):
        """``fc``: a dense class-weight tensor [C, 512] (FC_module.fc.data), or a ``PartialFC`` instance — then the head is the
        sampled / class-sharded softmax (its margin comes from the PartialFC's own ``margin_softmax``)."""
        from .partial_fc import PartialFC
        self.pfc = fc if isinstance(fc, PartialFC) else None
        if self.pfc is not None:
            fc = self.pfc.weight
        if loss_name not in ("CosFace", "ArcFace"):
            raise ValueError("loss must be CosFace or ArcFace")
        self.bb = backbone
        self.fc = _C.require_gpu_tensor(fc, f32, "fc weight")
        self.arc = loss_name == "ArcFace"
        self.s, self.m, self.lr, self.mu, self.wd = float(s), float(m), float(lr), float(momentum), float(weight_decay)
        bb = backbone
        bb._ensure_device_state()
        bb.train()
        bb.require_all_trainable("FusedTrainer")
        self.n_train = bb.trainable_count()
        self.mom = torch.empty(self.n_train, dtype=f32, device=bb.device)
        if self.pfc is None:
            self.fc_mom = torch.empty_like(self.fc)
            self.fc_grad = torch.empty_like(self.fc)
        self.first = True
        # weight-gradient GEMMs run on a second HIP stream (fedfr_net_backward2) unless FEDFR_DUAL_STREAM=0
        self.aux_stream = _make_aux_stream(bb.device, aux_slot)
        self._shadows_pending = None
        if self.aux_stream is not None:       # the shared aux stream may still carry the previous trainer's shadow rebuild of this backbone
            torch.cuda.current_stream().wait_stream(self.aux_stream)
        bb.refresh_shadows(True)
        # step(): SGD folded into the backward pass (fedfr_net_backward2_sgd) unless FEDFR_FUSE_SGD=0; forward_backward() + optimizer_step()
        # called separately keep the gradients-then-update contract
        import os
        self.fuse_sgd = os.environ.get("FEDFR_FUSE_SGD", "1") != "0"
        self._fuse_sgd = False
        self._sgd_done_from = None
        self._grads_scaled = False          # fp16-storage build: [0, _sgd_done_from) of the gradient buffer still carries the loss scale
        self._scale_used = 1.0              # ... namely this one (the scale the last backward pass multiplied in)
        self._init_loss_scale(bb.device)

    def set_lr(self, lr: float):
        self.lr = float(lr)

    def finish(self):
        """Order the aux-stream shadow rebuild of the last step before anything the caller does next on the current stream
        (state_dict snapshot, load_state_dict of the next round, a new trainer)."""
        if self._shadows_pending is not None:
            torch.cuda.current_stream().wait_stream(self._shadows_pending)
            self._shadows_pending = None
        self.check_overflow()

    @_C.on_device(lambda self: self.bb.device)
    def forward_backward(self, imgs: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        bb = self.bb
        bb._check_input(imgs)
        labels = _C.require_gpu_tensor(labels, torch.int64, "labels")
        B = imgs.shape[0]
        plan = bb._plan(B)
        st = _C.stream()
        feats = torch.empty(B, bb.num_features, dtype=f32, device=bb.device)
        bb._pre_forward(plan, bb._fwd_mode())
        _C.call("fedfr_net_forward", plan.handle, imgs.data_ptr(), bb._flat_params.data_ptr(), bb._flat_bufs.data_ptr(),
                bb._shadow.data_ptr(), plan.act.data_ptr(), plan.ws.data_ptr(), feats.data_ptr(), bb._fwd_mode(), st)
        bb._fwd_generation += 1
        if self.pfc is not None:
            if not bb._bn_frozen:
                bb._flat_nbt += 1
            # upstream PartialFC protocol: normalised embeddings in, d(embedding) out
            fn, finv = ops.normalize_rows(feats)
            if self.pfc.world_size == 1:
                x_slabs, loss = self.pfc.forward_backward(labels, fn, None, x_grad_slabs=True)      # split-K slabs, added by the pass below
                dfeats = ops.normalize_rows_bwd_slabs(fn, finv, x_slabs.contiguous())
            else:
                x_grad, loss = self.pfc.forward_backward(labels, fn, None)
                dfeats = ops.normalize_rows_bwd(fn, finv, x_grad.contiguous())
            self._backward(plan, imgs, dfeats, st)
            return loss
        # head: cosine logits -> margin -> softmax CE, gradient wrt cosine written in place
        fn, finv = ops.normalize_rows(feats)
        wn, winv = ops.normalize_rows(self.fc)
        C_, D_ = wn.shape
        if _HEAD_SPLITK and C_ <= 4096 and D_ >= 256 and C_ >= 256:
            # round 3: the two GEMMs on the path are latency chains (16 / 32 dependent k-steps on 32 / 16 workgroups): split-K slabs that
            # the consumer adds in order, and margin -> softmax -> gradient as one launch
            ks, kd = _split_for(D_), _split_for(C_)
            prob_t, g = ops.softmax_ce_fused(ops.sgemm(fn, wn, trans_b=True, splits=ks), labels, self.s, self.m, self.arc, 1.0 / B)
            dfeats = ops.normalize_rows_bwd_slabs(fn, finv, ops.sgemm(g, wn, splits=kd))
        else:
            cos = ops.sgemm(fn, wn, trans_b=True)
            prob_t, g = ops.softmax_ce_grad(cos, labels, self.s, self.m, self.arc, 1.0 / B)
            dfn = ops.sgemm(g, wn)
            dfeats = ops.normalize_rows_bwd(fn, finv, dfn)

        def off_path():
            # what the backbone's backward pass does not wait for: the loss value, d(loss)/d(class weights), the step counters
            loss = ops.nll_mean(prob_t, 0.0)
            dwn = ops.sgemm(g, fn, trans_a=True)
            _C.call("fedfr_normalize_rows_bwd", wn.data_ptr(), winv.data_ptr(), dwn.data_ptr(), self.fc_grad.data_ptr(),
                    wn.shape[0], wn.shape[1], 0.0, _C.stream())
            if not bb._bn_frozen:
                bb._flat_nbt += 1
            return loss

        if self.aux_stream is None or not _HEAD_OFF_PATH:
            loss = off_path()
            self._backward(plan, imgs, dfeats, st)
            return loss
        # round 3: those ~25 us of launches go to the weight-gradient stream BEHIND the backward pass's own work there (that stream is
        # ordered after the head kernels above by the pass's first fork) and the main stream joins it again before anything else runs
        self._backward(plan, imgs, dfeats, st)
        main = torch.cuda.current_stream()
        try:
            with torch.cuda.stream(self.aux_stream):
                loss = off_path()
        finally:
            # the join is unconditional: prob_t / g / fn / wn live in the main stream's pool, and a raise inside off_path must not leave the
            # main stream free to reuse them while the aux stream still reads
            main.wait_stream(self.aux_stream)
        loss.record_stream(main)
        return loss

    def _backward(self, plan, imgs, dfeats, st):
        bb = self.bb
        if self._shadows_pending is not None:
            torch.cuda.current_stream().wait_stream(self._shadows_pending)   # dgrad shadows rebuilt on aux after the last SGD step
            self._shadows_pending = None
        aux = self.aux_stream.cuda_stream if self.aux_stream is not None else None
        S = self._scale_used = self.loss_scale      # (the device's scale may be lowered by another trainer's poll before optimizer_step(): remember ours)
        self._grads_scaled = False
        if self.guarded:
            # fp16-storage library: the gradient enters the backbone multiplied by the loss scale S
            ds = dfeats * S
            if self._fuse_sgd:
                # step(): the update kernels undo the scale themselves (fedfr_sgd_step_scaled: g * 1/S, stored back), so the update rides inside
                # the backward pass exactly as in the bf16 build; optimizer_step() finishes [0, done_from) the same way
                import ctypes
                done = ctypes.c_longlong(0)
                _C.call("fedfr_net_backward2_sgd_scaled", plan.handle, imgs.data_ptr(), ds.data_ptr(), bb._flat_params.data_ptr(),
                        bb._shadow.data_ptr(), plan.act.data_ptr(), plan.ws.data_ptr(), bb._flat_grads.data_ptr(), self.mom.data_ptr(),
                        self.lr, self.mu, self.wd, 1 if self.first else 0, 1.0 / S, self._overflow.data_ptr(), ctypes.byref(done), st, aux)
                self._sgd_done_from = int(done.value)
                self._grads_scaled = True
                return
            # forward_backward(): the parameter gradients are unscaled before anything reads them
            _C.call("fedfr_net_backward2", plan.handle, imgs.data_ptr(), ds.data_ptr(), bb._flat_params.data_ptr(),
                    bb._shadow.data_ptr(), plan.act.data_ptr(), plan.ws.data_ptr(), bb._flat_grads.data_ptr(), st, aux)
            bb._flat_grads.mul_(1.0 / S)         # (inf / NaN stay what they are: optimizer_step()'s guarded kernel skips those elements)
            self._sgd_done_from = None
            return
        if self._fuse_sgd:
            # step(): the optimiser update of every parameter range whose gradient is final rides on the weight-gradient stream inside the
            # backward pass (bn2 / fc / features first, then stage by stage); optimizer_step() finishes [0, done_from)
            import ctypes
            done = ctypes.c_longlong(0)
            _C.call("fedfr_net_backward2_sgd", plan.handle, imgs.data_ptr(), dfeats.data_ptr(), bb._flat_params.data_ptr(),
                    bb._shadow.data_ptr(), plan.act.data_ptr(), plan.ws.data_ptr(), bb._flat_grads.data_ptr(), self.mom.data_ptr(),
                    self.lr, self.mu, self.wd, 1 if self.first else 0, ctypes.byref(done), st, aux)
            self._sgd_done_from = int(done.value)
            return
        _C.call("fedfr_net_backward2", plan.handle, imgs.data_ptr(), dfeats.data_ptr(), bb._flat_params.data_ptr(),
                bb._shadow.data_ptr(), plan.act.data_ptr(), plan.ws.data_ptr(), bb._flat_grads.data_ptr(), st, aux)

    @_C.on_device(lambda self: self.bb.device)
    def optimizer_step(self):
        bb = self.bb
        st = _C.stream()
        first = 1 if self.first else 0
        n_rest = self.n_train if self._sgd_done_from is None else self._sgd_done_from      # the rest was updated inside the backward pass
        self._sgd_done_from = None
        scaled, self._grads_scaled = self._grads_scaled, False
        if self.guarded:
            # fp16-storage library: EVERY update is the guarded kernel — the rest of a fused step still carries the scale (gscale = 1/S),
            # the gradients of forward_backward() were unscaled there already (gscale = 1); non-finite elements are skipped and flagged either way
            ovf = self._overflow.data_ptr()
            if n_rest > 0:
                _C.call("fedfr_sgd_step_scaled", bb._flat_params.data_ptr(), bb._flat_grads.data_ptr(), self.mom.data_ptr(),
                        bb._shadow.data_ptr(), n_rest, self.lr, self.mu, self.wd, first, 1.0 / self._scale_used if scaled else 1.0, ovf, st)
            if self.pfc is not None:
                self.pfc.fused_sgd_update(self.lr, self.mu, self.wd, overflow=self._overflow)      # sampled rows: guarded SGD + scatter back
            else:
                # (fp32 head: no scale to undo, but a forward pass that overflowed fp16 hands it non-finite gradients too: same guard)
                _C.call("fedfr_sgd_step_scaled", self.fc.data_ptr(), self.fc_grad.data_ptr(), self.fc_mom.data_ptr(), None, self.fc.numel(),
                        self.lr, self.mu, self.wd, first, 1.0, ovf, st)
            self._count_step()
        else:
            if n_rest > 0:
                _C.call("fedfr_sgd_step", bb._flat_params.data_ptr(), bb._flat_grads.data_ptr(), self.mom.data_ptr(),
                        bb._shadow.data_ptr(), n_rest, self.lr, self.mu, self.wd, first, st)
            if self.pfc is not None:
                self.pfc.fused_sgd_update(self.lr, self.mu, self.wd)      # sampled rows: SGD + scatter back
            else:
                _C.call("fedfr_sgd_step", self.fc.data_ptr(), self.fc_grad.data_ptr(), self.fc_mom.data_ptr(), None, self.fc.numel(),
                        self.lr, self.mu, self.wd, first, st)
        # dgrad-layout weight copies (only needed by the NEXT backward; the forward mirror was written by the SGD kernel):
        # rebuilt on the aux stream so they overlap the next forward pass
        if self.aux_stream is not None:
            main = torch.cuda.current_stream()
            self.aux_stream.wait_stream(main)
            with torch.cuda.stream(self.aux_stream):
                bb.refresh_shadows(False)
            self._shadows_pending = self.aux_stream
        else:
            bb.refresh_shadows(False)
        self.first = False

    def step(self, imgs: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        self._fuse_sgd = self.fuse_sgd
        try:
            loss = self.forward_backward(imgs, labels)
        finally:
            self._fuse_sgd = False
        self.optimizer_step()
        return loss


class FusedHeadTrainer(_LossScaleGuard):
    """Fused backbone step with an arbitrary differentiable head — the training body of ``train_with_public_data``
    (reference client.py:354-441): the backbone runs as ``fedfr_net_forward`` / ``fedfr_net_backward2`` + the flat SGD
    kernel exactly as in ``FusedTrainer``; the head (FC_module / BCE_module / margin / CE / contrastive — each a HIP-backed
    autograd Function of this package) is evaluated on the embedding leaf, and its few parameters get the same
    ``fedfr_sgd_step`` (torch.optim.SGD semantics: coupled weight decay, momentum buffer created on first use)."""

    def __init__(self, backbone: "backbones.IResNet", head_params: Iterable[nn.Parameter], lr: float = 0.1,
                 momentum: float = 0.9, weight_decay: float = 5e-4, aux_slot: int = 0):
        import os
        self.bb = backbone
        self.head_params = list(head_params)
        self.lr, self.mu, self.wd = float(lr), float(momentum), float(weight_decay)
        bb = backbone
        bb._ensure_device_state()
        bb.train()
        bb.require_all_trainable("FusedHeadTrainer")
        for hp in self.head_params:
            _C.require_gpu_tensor(hp.data, f32, "head parameter")
        self.n_train = bb.trainable_count()
        self.mom = torch.empty(self.n_train, dtype=f32, device=bb.device)
        self.head_mom = {}
        self.first = True
        self.fuse_sgd = os.environ.get("FEDFR_FUSE_SGD", "1") != "0"
        self._init_loss_scale(bb.device)
        self.aux_stream = _make_aux_stream(bb.device, aux_slot)
        self._shadows_pending = None
        if self.aux_stream is not None:
            torch.cuda.current_stream().wait_stream(self.aux_stream)
        bb.refresh_shadows(True)

    def set_lr(self, lr: float):
        self.lr = float(lr)

    def _forward(self, imgs, labels):
        bb = self.bb
        bb._check_input(imgs)
        labels = _C.require_gpu_tensor(labels, torch.int64, "labels")
        plan = bb._plan(imgs.shape[0])
        feats = torch.empty(imgs.shape[0], bb.num_features, dtype=f32, device=bb.device)
        bb._pre_forward(plan, bb._fwd_mode())
        _C.call("fedfr_net_forward", plan.handle, imgs.data_ptr(), bb._flat_params.data_ptr(), bb._flat_bufs.data_ptr(),
                bb._shadow.data_ptr(), plan.act.data_ptr(), plan.ws.data_ptr(), feats.data_ptr(), bb._fwd_mode(), _C.stream())
        if not bb._bn_frozen:
            bb._flat_nbt += 1
        bb._fwd_generation += 1
        return plan, feats, labels

    def _backward_and_update(self, plan, imgs, dfeats):
        """backbone backward from d(loss)/d(features), then opt.step() for the backbone (flat) and every head parameter with a grad."""
        bb, st = self.bb, _C.stream()
        dfeats = _C.require_gpu_tensor(dfeats.contiguous(), f32, "d(loss)/d(features)")
        if self._shadows_pending is not None:
            torch.cuda.current_stream().wait_stream(self._shadows_pending)
            self._shadows_pending = None
        S = self.loss_scale                               # the fp16-storage library scales the incoming gradient (bf16 library: no scale)
        guarded = self.guarded
        if guarded:
            dfeats = dfeats * S
        aux = self.aux_stream.cuda_stream if self.aux_stream is not None else None
        ovf = self._overflow.data_ptr()
        first = 1 if self.first else 0
        n_rest = self.n_train
        # ---- backward + opt.step() of the backbone.  Round 5: as in FusedTrainer.step(), the update of every parameter range whose gradient is final
        # rides on the weight-gradient stream INSIDE the backward pass (fedfr_net_backward2_sgd: the same kernels, the same elementwise arithmetic,
        # enqueued earlier — bit-identical to backward + one flat update, FEDFR_FUSE_SGD=0); the flat kernel finishes [0, done_from).
        # (fp16-storage library: the kernels undo the scale in place and skip non-finite elements, _LossScaleGuard)
        if self.fuse_sgd:
            import ctypes
            done = ctypes.c_longlong(0)
            if guarded:
                _C.call("fedfr_net_backward2_sgd_scaled", plan.handle, imgs.data_ptr(), dfeats.data_ptr(), bb._flat_params.data_ptr(),
                        bb._shadow.data_ptr(), plan.act.data_ptr(), plan.ws.data_ptr(), bb._flat_grads.data_ptr(), self.mom.data_ptr(),
                        self.lr, self.mu, self.wd, first, 1.0 / S, ovf, ctypes.byref(done), st, aux)
            else:
                _C.call("fedfr_net_backward2_sgd", plan.handle, imgs.data_ptr(), dfeats.data_ptr(), bb._flat_params.data_ptr(),
                        bb._shadow.data_ptr(), plan.act.data_ptr(), plan.ws.data_ptr(), bb._flat_grads.data_ptr(), self.mom.data_ptr(),
                        self.lr, self.mu, self.wd, first, ctypes.byref(done), st, aux)
            n_rest = int(done.value)
        else:
            _C.call("fedfr_net_backward2", plan.handle, imgs.data_ptr(), dfeats.data_ptr(), bb._flat_params.data_ptr(),
                    bb._shadow.data_ptr(), plan.act.data_ptr(), plan.ws.data_ptr(), bb._flat_grads.data_ptr(), st, aux)
        if n_rest > 0 and guarded:
            _C.call("fedfr_sgd_step_scaled", bb._flat_params.data_ptr(), bb._flat_grads.data_ptr(), self.mom.data_ptr(),
                    bb._shadow.data_ptr(), n_rest, self.lr, self.mu, self.wd, first, 1.0 / S, ovf, st)
        elif n_rest > 0:
            _C.call("fedfr_sgd_step", bb._flat_params.data_ptr(), bb._flat_grads.data_ptr(), self.mom.data_ptr(),
                    bb._shadow.data_ptr(), n_rest, self.lr, self.mu, self.wd, first, st)
        for hp in self.head_params:
            if hp.grad is None:                                       # torch.optim.SGD skips parameters without a gradient
                continue
            if not hp.data.is_contiguous() or not hp.grad.is_contiguous():
                raise RuntimeError("fedfr_amd: head parameters and their gradients must be contiguous")
            buf = self.head_mom.get(hp)
            first = buf is None
            if first:
                buf = self.head_mom[hp] = torch.empty_like(hp.data)
            if guarded:
                _C.call("fedfr_sgd_step_scaled", hp.data.data_ptr(), hp.grad.data_ptr(), buf.data_ptr(), None, hp.numel(), self.lr, self.mu,
                        self.wd, 1 if first else 0, 1.0, ovf, st)
            else:
                _C.call("fedfr_sgd_step", hp.data.data_ptr(), hp.grad.data_ptr(), buf.data_ptr(), None, hp.numel(), self.lr, self.mu,
                        self.wd, 1 if first else 0, st)
        if self.aux_stream is not None:
            main = torch.cuda.current_stream()
            self.aux_stream.wait_stream(main)
            with torch.cuda.stream(self.aux_stream):
                bb.refresh_shadows(False)
            self._shadows_pending = self.aux_stream
        else:
            bb.refresh_shadows(False)
        self.first = False
        if guarded:
            self._count_step()

    @_C.on_device(lambda self: self.bb.device)
    def step(self, imgs: torch.Tensor, labels: torch.Tensor, head_loss):
        """``head_loss(feats, labels) -> (loss, *extras)``; returns that tuple (loss detached)."""
        plan, feats, labels = self._forward(imgs, labels)
        for hp in self.head_params:                                   # opt.zero_grad()
            hp.grad = None
        feats.requires_grad_(True)
        with torch.enable_grad():
            out = head_loss(feats, labels)
            loss = out[0] if isinstance(out, tuple) else out
            loss.backward()
        self._backward_and_update(plan, imgs, feats.grad)
        if isinstance(out, tuple):
            return tuple(o.detach() if torch.is_tensor(o) else o for o in out)
        return loss.detach()

    def finish(self):
        """Order the aux-stream shadow rebuild before anything the caller does next on the current stream."""
        if self._shadows_pending is not None:
            torch.cuda.current_stream().wait_stream(self._shadows_pending)
            self._shadows_pending = None
        self.check_overflow()


class ShardedHeadTrainer(FusedHeadTrainer):
    """BASELINE config 5 — "iresnet100 + CosFace + personalized transform head, 8 clients = 8 MI355X, PartialFC class-sharded across
    GPUs".  The reference never runs this combination (its clients own private dense heads, client.py:149; its PartialFC is dead
    code), so the definition is this build's:

    * every rank is one FL client with its OWN backbone (weights diverge during the round, no gradient all-reduce) and its own data
      shard: identities [id_base, id_base + n_local_ids) of the global label space;
    * the identity head is ONE ``PartialFC(world_size=W)`` shared by all ranks: per step the normalised embeddings and labels of all
      W clients are all-gathered (global batch B*W), every rank computes the logits of its class shard against all of them, the
      softmax statistics are all-reduced, and d(embedding) comes back by reduce-scatter (partial_fc.py:118-176, four packed
      collectives per step) — this couples the clients step by step, unlike pure FL;
    * each client also trains its private ``BCE_module`` (client.py:25-60) on its own embeddings with local labels
      ``label - id_base`` (rows of other clients' identities are all-negative, client.py:48-52), weight 10 (client.py:383);
    * loss_i = CosFace-PartialFC loss (identical on all ranks) + 10 * BCE_i; backbone + BCE parameters + the rank's PartialFC rows
      get momentum-SGD (coupled weight decay); at round end the backbones are averaged with ``server.fedavg_all_reduce``.
    """

    def __init__(self, backbone, pfc, bce_module=None, id_base: int = 0, lr: float = 0.1, momentum: float = 0.9,
                 weight_decay: float = 5e-4, bce_weight: float = 10.0, aux_slot: int = 0):
        super().__init__(backbone, list(bce_module.parameters()) if bce_module is not None else [], lr, momentum, weight_decay, aux_slot=aux_slot)
        self.pfc, self.bce_module, self.id_base, self.bce_weight = pfc, bce_module, int(id_base), float(bce_weight)
        self.bce_loss = losses.BCE_loss() if bce_module is not None else None
        if bce_module is not None:
            bce_module.train()

    @_C.on_device(lambda self: self.bb.device)
    def step(self, imgs: torch.Tensor, labels: torch.Tensor, perm=None):
        """labels: GLOBAL identity ids.  Returns (loss, cos_loss, bce_loss or None) as device scalars."""
        plan, feats, labels = self._forward(imgs, labels)
        # ---- shared class-sharded head (the step's collectives live in here)
        fn, finv = ops.normalize_rows(feats)
        x_grad, cos_loss = self.pfc.forward_backward(labels, fn, None, perm=perm)
        dfeats = ops.normalize_rows_bwd(fn, finv, x_grad.contiguous())
        # ---- private personalised head on this client's own embeddings
        bce = None
        if self.bce_module is not None:
            for hp in self.head_params:
                hp.grad = None
            leaf = feats.detach().requires_grad_(True)
            with torch.enable_grad():
                lab = labels - self.id_base                   # local identity index; other clients' identities -> n_class = the
                lab = torch.where((lab < 0) | (lab >= self.bce_module.n_class), torch.full_like(lab, self.bce_module.n_class), lab)
                z, gt = self.bce_module(leaf, lab)            # all-negative row of client.py:48-52
                bce = self.bce_loss(z, gt)
                (self.bce_weight * bce).backward()
            ops.axpy_(dfeats, leaf.grad, 1.0)
            bce = bce.detach()
        self._backward_and_update(plan, imgs, dfeats)
        self.pfc.fused_sgd_update(self.lr, self.mu, self.wd, overflow=self._overflow if self.guarded else None)
        loss = cos_loss if bce is None else ops.axpy_(ops.scale(bce.reshape(1), self.bce_weight), cos_loss.reshape(1), 1.0).reshape(())
        return loss, cos_loss, bce

    def end_round(self, data_size: float, total_size: float):
        """FedAvg of the backbones over the ranks (ONE all-reduce of the flat state)."""
        from .server import fedavg_all_reduce
        self.finish()
        w = fedavg_all_reduce(self.bb, data_size, total_size, self.pfc.comm)
        self.bb.refresh_shadows(True)
        return w


# ------------------------------------------------------------------------------------------------
# Client
# ------------------------------------------------------------------------------------------------
class AverageMeter:
    """reference utils/utils_logging.py:6-27."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


class CombineDataset(torch.utils.data.Dataset):
    """Local identities followed by the (hard-negative subset of the) public identities, public labels shifted behind the
    local classes (reference dataset.py:170-187, MXFaceDataset_Combine).  Both datasets expose ``num_classes``."""

    def __init__(self, first_dataset, second_dataset):
        super().__init__()
        self.first_dataset, self.second_dataset = first_dataset, second_dataset
        self.first_nclass = first_dataset.num_classes
        self.first_len, self.second_len = len(first_dataset), len(second_dataset)
        self.num_class = first_dataset.num_classes + second_dataset.num_classes

    def __getitem__(self, idx):
        if idx < self.first_len:
            return self.first_dataset[idx]
        img, label = self.second_dataset[idx - self.first_len]
        return img, label + self.first_nclass

    def __len__(self):
        return self.first_len + self.second_len


class SubsetDataset(torch.utils.data.Dataset):
    """The images of ``dataset`` selected by ``indices``, labels and ``num_classes`` unchanged — what the reference gets by
    overwriting ``dataset.imgidx`` of a deep-copied loader (client.py:228, :258)."""

    def __init__(self, dataset, indices):
        super().__init__()
        self.dataset, self.indices = dataset, [int(i) for i in indices]
        self.num_classes = dataset.num_classes
        self.ID_base = getattr(dataset, "ID_base", None)

    def __getitem__(self, i):
        return self.dataset[self.indices[i]]

    def __len__(self):
        return len(self.indices)


@torch.no_grad()
def embed_dataset(backbone, loader, device, normalize=True):
    """eval-mode embeddings of every image of ``loader`` (the inference sweep of server.py:242-263 / client.py:197-205):
    returns ([N, 512] fp32 on ``device``, [N] int64 labels)."""
    backbone.eval()
    feats, labels = [], []
    for img, label in loader:
        img, label = to_device_batch(img, label, device, train=False)
        f = backbone(img)
        if normalize:
            f, _ = ops.normalize_rows(f)
        feats.append(f)
        labels.append(label)
    return torch.cat(feats, dim=0), torch.cat(labels, dim=0).to(torch.int64)


@torch.no_grad()
def class_centers(backbone, loader, num_classes, device, norm_before_avg):
    """per-class mean embedding (client.py:159-188 data_update_fc, server.py:182-240 Initialize_pretrain_FC): eval forward,
    optional F.normalize, per-batch per-class sums accumulated on the GPU, divided by the sample counts."""
    backbone.eval()
    sums = torch.zeros(num_classes, backbone.num_features, dtype=f32, device=device)
    counts = torch.zeros(num_classes, dtype=f32, device=device)
    labels = []
    for img, label in loader:
        img, lab = to_device_batch(img, label, device, train=False)
        f = backbone(img)
        if norm_before_avg:
            f, _ = ops.normalize_rows(f)
        ops.class_accumulate(f, lab, sums, counts)
        labels.append(lab)
    return sums / counts.unsqueeze(1), torch.cat(labels)


def to_device_batch(imgs, labels, device, train: bool):
    """Host batch -> (fp32 [B,3,H,W] in [-1,1], int64 labels) on ``device``.  Already-normalised fp32 NCHW batches (what the
    reference's DataLoader yields after its CPU transform, dataset.py:81-92) pass through; uint8 [B,H,W,3] batches take the
    on-device transform instead (``ops.preprocess_u8``: ToTensor + Normalize(0.5, 0.5), with RandomHorizontalFlip(p=0.5) when
    ``train``), so the host uploads one byte per pixel-channel."""
    labels = torch.as_tensor(labels).to(device, non_blocking=True).to(torch.int64)
    if imgs.dtype == torch.uint8:
        flip = (torch.rand(imgs.shape[0]) < 0.5).to(torch.uint8).to(device) if train else None
        return ops.preprocess_u8(imgs.to(device, non_blocking=True).contiguous(), flip), labels
    return imgs.to(device, non_blocking=True).contiguous(), labels


_BACKBONE_POOL = {}     # (network, device, dropout) -> the one resident backbone (+ activation arenas) all Clients of this process share


def shared_backbone(network: str, device, dropout=0, slot: int = 0):
    """The reference builds a client's backbone inside ``train()`` and deletes it afterwards (client.py:513, :568-570), so 40 clients
    fit one GPU.  Here a backbone owns GBs of activation arena + workspace per batch size (iresnet100 at B=128: 5.8 + 3 GB), so all
    Clients of a process train through ONE resident instance per (arch, device, slot): each call loads its own state_dict into it.
    ``slot`` > 0: the instances of clients that train concurrently on one device (``Server.train`` with ``args.parallel_clients``)."""
    device = torch.device(device)
    key = (network, str(device), float(dropout), int(slot))
    with _AUX_LOCK:
        bb = _BACKBONE_POOL.get(key)
        if bb is None:
            bb = _BACKBONE_POOL[key] = getattr(backbones, network)(False, dropout=dropout, fp16=cfg.fp16).to(device)
    return bb


class Client(object):
    """Local trainer of one FL participant (reference client.py:116-157, :511-582).

    ``data`` must expose ``train_class_sizes``, ``train_dataset_sizes`` and ``train_loaders`` (indexable by
    cid; each loader iterates ``(imgs[B,3,112,112] fp32 in [-1,1], labels int64)`` and has ``.dataset.ID_base``),
    exactly what ``All_Client_Dataset`` provides in the reference (dataset.py:73-142).
    """

    def __init__(self, cid, args, data, device: Optional[torch.device] = None):
        self.cid = cid
        self.args = args
        self.num_classes = data.train_class_sizes[self.cid]
        self.local_epoch = args.local_epoch
        self.dataset_size = data.train_dataset_sizes[self.cid]
        self.train_loader = data.train_loaders[self.cid]
        ds = getattr(self.train_loader, "dataset", None)
        self.ID_base = getattr(ds, "ID_base", 0)
        self.target_ID = list(range(self.ID_base, self.ID_base + self.num_classes))
        self.loss_name = args.loss
        self.margin_softmax = getattr(losses, args.loss)(s=30, m=0.4)          # client.py:133
        if getattr(self.args, "BCE_local", False):
            self.bce_module = BCE_module(512, self.num_classes, cfg.converter_layer)
            self.bce_loss = losses.BCE_loss()
        self.rank = 0
        self.local_rank = 0
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.dropout = 0.4 if cfg.dataset == "webface" else 0
        self.backbone_state_dict = None
        self.fc_module = FC_module(512, self.num_classes, getattr(args, "output_dir", "."))
        if getattr(self.args, "contrastive_bb", False):                          # client.py:151-155
            self.last_model = getattr(backbones, self.args.network)(False, dropout=self.dropout, fp16=cfg.fp16)
            self.temperature = 0.5
        if hasattr(data, "public_train_loader"):
            self.public_num_classes = data.public_train_loader.dataset.num_classes
        if hasattr(data, "test_loaders"):
            self.test_loaders = data.test_loaders[self.cid]                      # client.py:128-129
        self.logger = logging.getLogger("FL_face.client")
        self.loss_meter = AverageMeter()
        self.sync_every = getattr(args, "loss_sync_every", 1)      # reference syncs (loss.item()) every step

    def _get_backbone(self):
        return shared_backbone(self.args.network, self.device, self.dropout, getattr(self, "slot", 0))

    @_C.on_device(lambda self: self.device)
    def train(self, start_epoch=0, callback_verification=None):
        """reference client.py:511-571."""
        backbone = self._get_backbone()
        backbone.load_state_dict(self.backbone_state_dict)
        backbone.train()
        self.fc_module.to(self.device)
        self.fc_module.train()
        trainer = FusedTrainer(backbone, self.fc_module.fc.data, self.loss_name, 30.0, 0.4,
                               lr=cfg.lr_func(start_epoch) * cfg.lr, momentum=cfg.momentum, weight_decay=cfg.weight_decay,
                               aux_slot=getattr(self, "slot", 0))
        loss_meter = AverageMeter()
        pending = []
        for epoch in range(start_epoch, start_epoch + self.local_epoch):
            for step, (imgs, labels) in enumerate(self.train_loader):
                if len(imgs) == 1:                                              # client.py:538-540
                    imgs = torch.cat([imgs, imgs], dim=0)
                    labels = torch.cat([labels, labels])
                imgs, labels = to_device_batch(imgs, labels, self.device, train=True)
                pending.append(trainer.step(imgs, labels))
                if len(pending) >= self.sync_every:
                    for l in pending:
                        loss_meter.update(l.item(), 1)
                    pending = []
                    trainer.check_overflow()         # fp16 library: the host has just synchronised for the loss — read the overflow word too
        for l in pending:
            loss_meter.update(l.item(), 1)
        trainer.finish()
        self.loss_meter = loss_meter
        self.backbone_state_dict = flat_state_dict(backbone)
        self.fc_module.cpu()

    @_C.on_device(lambda self: self.device)
    def data_update_fc(self, fed_model_state_dict, norm_before_avg, fc_name="center_features", save_to_disk=False):
        """reference client.py:159-188: class centres of the client's own identities under the incoming global model become the
        local rows of the cosine head."""
        backbone = self._get_backbone()
        backbone.load_state_dict(fed_model_state_dict)
        init_fc, _ = class_centers(backbone, self.test_loaders, self.num_classes, self.device, norm_before_avg)
        init_fc = init_fc.cpu()
        if save_to_disk:
            import os
            torch.save(init_fc, os.path.join(getattr(self, "client_output", "."), fc_name + ".pth"))
        self.fc_module.update_from_tensor(init_fc)

    @_C.on_device(lambda self: self.device)
    def choose_hard_negative_2(self, public_train_loader, pretrained_label, pretrained_feats, threshold=0.2):
        """reference client.py:191-236 (feature-based hard negatives): embed the local images with the current backbone, keep
        every public image whose normalised embedding has cosine similarity > threshold with ANY local embedding.  The
        [N_local, N_public] similarity matrix is never built: one fp32 MFMA GEMM whose epilogue ORs `> threshold` down each
        column.  Returns a loader over the selected public images (same labels / num_classes); ``self.HN_index`` keeps the
        sorted image indices."""
        backbone = self._get_backbone()
        backbone.load_state_dict(self.backbone_state_dict)
        local_feats, _ = embed_dataset(backbone, self.test_loaders, self.device, normalize=True)
        pretrained_feats = _C.require_gpu_tensor(pretrained_feats.to(self.device), f32, "pretrained_feats")
        flags = ops.similarity_column_flags(local_feats, pretrained_feats, float(threshold))
        unique_idx = torch.nonzero(flags, as_tuple=False).flatten()               # ascending == sorted(union of the row hits)
        self.HN_index = unique_idx.cpu()
        num_id = int(torch.unique(torch.as_tensor(pretrained_label)[self.HN_index]).numel()) if len(self.HN_index) else 0
        self.logger.info("%d imgs (%d ID) are hard negative with similarity > %.2f" % (len(self.HN_index), num_id, threshold))
        subset = SubsetDataset(public_train_loader.dataset, self.HN_index.tolist())
        return torch.utils.data.DataLoader(subset, batch_size=getattr(public_train_loader, "batch_size", None) or cfg.public_batch_size,
                                           shuffle=True, drop_last=False)

    def reweight_cosface(self, logits, labels):
        """reference client.py:269-285: append (num_client-1) copies of the first ``num_classes`` non-target logits of every
        row, so the softmax denominator weighs the local negatives as if every client contributed them.  Quirk kept on
        purpose: the reference concatenates INSIDE torch.no_grad() (client.py:272-276), so the returned logits are detached —
        the re-weighted CosFace loss is reported but contributes no gradient (and the non-BCE, non-contrastive branch then
        fails in loss.backward(), here as there)."""
        B, C = logits.shape
        with torch.no_grad():
            keep = torch.ones(B, C, dtype=torch.bool, device=logits.device)
            keep[torch.arange(B, device=logits.device), labels] = False
            tmp = logits.detach()[keep].reshape(B, C - 1)[:, :self.num_classes].repeat(1, self.args.num_client - 1)
            logits = torch.cat([logits, tmp], dim=1)
        return logits

    @_C.on_device(lambda self: self.device)
    def train_with_public_data(self, start_epoch=0, callback_verification=None, public_train_loader=None, pretrained_fc=None,
                               choose_hard_negative=False, pretrained_label=None, pretrained_feats=None, combine_loader=None):
        """reference client.py:287-508 — local + public identities, CosFace over [local | public] class centres, optional
        personalised BCE branch (``args.BCE_local``, weight 10) and model-contrastive term (``args.contrastive_bb``,
        weight cfg.mu), StepLR(cfg.train_decay, 0.1) re-created per call.

        ``combine_loader`` (build extension, used by tests and synthetic benchmarks) supplies the combined batches directly;
        otherwise the combined loader is built from ``self.train_loader.dataset`` + the (hard-negative subset of the) public
        dataset as the reference does.  The reference's non-BCE contrastive branch unpacks ``model(imgs)`` into two values (client.py:413),
        which only works if the model is called with ``contrastive=True``; that is what this method does."""
        if combine_loader is None:
            if choose_hard_negative:                                               # client.py:291-293
                public_loader_subset = self.choose_hard_negative_2(public_train_loader, pretrained_label, pretrained_feats,
                                                                   threshold=cfg.HN_threshold)
            else:
                public_loader_subset = public_train_loader
            if not getattr(self.args, "combine_dataset", False):
                raise NotImplementedError()                                        # client.py:303-304
            combine_dataset = CombineDataset(self.train_loader.dataset, public_loader_subset.dataset)
            combine_loader = torch.utils.data.DataLoader(combine_dataset, batch_size=cfg.com_batch_size, shuffle=True, num_workers=0,
                                                         pin_memory=True, drop_last=True)
            self.dataset_size = len(combine_dataset)                               # for FedAvg (client.py:302)
        elif hasattr(combine_loader, "dataset"):
            self.dataset_size = len(combine_loader.dataset)
        backbone = self._get_backbone()
        backbone.load_state_dict(self.backbone_state_dict)
        backbone.train()
        self.fc_module.update_with_pretrain(pretrained_fc)                         # [local | public] rows (client.py:312)
        self.fc_module.train()
        self.fc_module.to(self.device)
        use_bce = bool(getattr(self.args, "BCE_local", False))
        use_con = bool(getattr(self.args, "contrastive_bb", False))
        detach = bool(getattr(self.args, "BCE_detach", False))
        reweight = bool(getattr(self.args, "reweight_cosface", False))
        head_params = list(self.fc_module.parameters())
        if use_bce:
            self.bce_module.train()
            self.bce_module.to(self.device)
            head_params += list(self.bce_module.parameters())
        if use_con:
            import copy
            global_model = copy.deepcopy(backbone).eval()                          # frozen copy of the incoming global model
            self.last_model = self.last_model.to(self.device).eval()
        trainer = FusedHeadTrainer(backbone, head_params, lr=cfg.lr, momentum=cfg.momentum, weight_decay=cfg.weight_decay,
                                   aux_slot=getattr(self, "slot", 0))
        margin, fc_module = self.margin_softmax, self.fc_module
        state = {}

        def head_loss(feats, labels):
            cos_logits = margin(fc_module(feats), labels)
            if reweight:
                cos_logits = self.reweight_cosface(cos_logits, labels)
            cos_loss = ops.cross_entropy(cos_logits, labels)
            loss = cos_loss
            bce = con = None
            if use_bce:
                bce_logits, bce_gts = self.bce_module(feats.detach() if detach else feats, labels)
                bce = self.bce_loss(bce_logits, bce_gts)
                loss = loss + 10 * bce
            if use_con:
                con = ops.contrastive_loss(feats, state["global_feats"], state["last_feats"], self.temperature)
                loss = loss + cfg.mu * con
            return loss, cos_loss, con, bce

        loss_meter, cos_meter, con_meter, bce_meter = AverageMeter(), AverageMeter(), AverageMeter(), AverageMeter()
        pending = []

        def drain():
            for l, c, k, b in pending:
                loss_meter.update(l.item(), 1)
                cos_meter.update(c.item(), 1)
                if k is not None:
                    con_meter.update(k.item(), 1)
                if b is not None:
                    bce_meter.update(b.item(), 1)
            pending.clear()
            trainer.check_overflow()                 # fp16 library: rides on the synchronisation the loss values have just paid for

        for epoch in range(start_epoch, start_epoch + self.local_epoch):
            trainer.set_lr(cfg.lr * 0.1 ** ((epoch - start_epoch) // cfg.train_decay))     # StepLR (client.py:348,443)
            for step, (imgs, labels) in enumerate(combine_loader):
                imgs, labels = to_device_batch(imgs, labels, self.device, train=True)
                if use_con:
                    with torch.no_grad():
                        state["global_feats"] = global_model(imgs)
                        state["last_feats"] = self.last_model(imgs)
                pending.append(trainer.step(imgs, labels, head_loss))
                if len(pending) >= self.sync_every:
                    drain()
        drain()
        trainer.finish()
        self.cos_meter, self.con_meter, self.bce_meter = cos_meter, con_meter, bce_meter
        self.loss_meter = loss_meter
        self.backbone_state_dict = flat_state_dict(backbone)
        self.fc_module.cpu()
        if use_bce:
            self.bce_module.cpu()
        if use_con:
            self.last_model.load_state_dict(self.backbone_state_dict)               # client.py:499-501
            self.last_model.release_workspace()      # keep only its 260 MB of weights resident between rounds, not its arenas
            del global_model

    def get_train_loss(self):
        return self.loss_meter.avg

    def get_model(self):
        return self.backbone_state_dict

    def get_global_fc(self):
        return self.fc_module.get_pretrain_fc()

    def get_data_size(self):
        return self.dataset_size


class FlatStateDict(OrderedDict):
    """state_dict whose tensors are views of three flat tensors (kept as ``.flat``) so that FedPavg can run as
    a few fused kernels instead of a Python loop over 925 keys (reference server.py:25-34).

    The 925 views are built LAZILY, on the first dictionary access: the flat consumers (``FedPavg``'s fast path, ``IResNet.load_state_dict``
    of a FlatStateDict) never touch them, and building them costs ~2 ms of host time per state — more than the aggregation kernels
    themselves (measured round 3: FedPavg over 8 iresnet100 states 1.5 ms with eager views, of which the kernels are 0.5)."""
    flat: Tuple[torch.Tensor, torch.Tensor, torch.Tensor] = None
    table = None
    layers = None
    _built = True           # plain construction (no flat tensors): an ordinary OrderedDict

    @classmethod
    def from_flat(cls, flat, table, layers) -> "FlatStateDict":
        sd = cls()
        sd.flat, sd.table, sd.layers = tuple(flat), table, layers
        sd._built = False
        return sd

    def _ensure(self):
        if self._built:
            return
        self._built = True
        p, b, n = self.flat
        put = OrderedDict.__setitem__
        for name, kind, region, off, shape in self.table:
            if region == 0:
                if kind == backbones.iresnet.KIND_CONV:
                    o, i, r, _ = shape
                    put(self, name, p[off: off + o * i * r * r].view(o, r, r, i).permute(0, 3, 1, 2))
                else:
                    num = 1
                    for s_ in shape:
                        num *= s_
                    put(self, name, p[off: off + num].view(shape))
            elif region == 1:
                put(self, name, b[off: off + shape[0]])
            else:
                put(self, name, n[off])

    def __getitem__(self, k):
        self._ensure()
        return OrderedDict.__getitem__(self, k)

    def __setitem__(self, k, v):
        self._ensure()
        OrderedDict.__setitem__(self, k, v)

    def __delitem__(self, k):
        self._ensure()
        OrderedDict.__delitem__(self, k)

    def __iter__(self):
        self._ensure()
        return OrderedDict.__iter__(self)

    def __reversed__(self):
        self._ensure()
        return OrderedDict.__reversed__(self)

    def __len__(self):
        self._ensure()
        return OrderedDict.__len__(self)

    def __contains__(self, k):
        self._ensure()
        return OrderedDict.__contains__(self, k)

    def __eq__(self, other):
        self._ensure()
        return OrderedDict.__eq__(self, other)

    __hash__ = None

    def keys(self):
        self._ensure()
        return OrderedDict.keys(self)

    def values(self):
        self._ensure()
        return OrderedDict.values(self)

    def items(self):
        self._ensure()
        return OrderedDict.items(self)

    def get(self, k, default=None):
        self._ensure()
        return OrderedDict.get(self, k, default)

    def pop(self, *a, **kw):
        self._ensure()
        return OrderedDict.pop(self, *a, **kw)

    def popitem(self, last=True):
        self._ensure()
        return OrderedDict.popitem(self, last)

    def setdefault(self, k, default=None):
        self._ensure()
        return OrderedDict.setdefault(self, k, default)

    def move_to_end(self, k, last=True):
        self._ensure()
        OrderedDict.move_to_end(self, k, last)

    def update(self, *a, **kw):
        self._ensure()
        OrderedDict.update(self, *a, **kw)

    def clear(self):
        # an emptied dictionary must stay empty: no lazy rebuild afterwards, and no flat tensors for load_state_dict's fast path to pick up
        self._built = True
        self.flat = None
        OrderedDict.clear(self)

    def copy(self):
        self._ensure()
        return OrderedDict(self.items())

    def __repr__(self):
        self._ensure()
        return OrderedDict.__repr__(self)


def flat_state_dict(backbone, clone=True) -> FlatStateDict:
    """Snapshot of a backbone's state as a FlatStateDict (device tensors; keys == reference state_dict keys)."""
    p, b, n = backbone.flat_state()
    if clone:
        p, b, n = p.clone(), b.clone(), n.clone()
    return FlatStateDict.from_flat((p, b, n), backbone._table, backbone.layers_cfg)
