"""PartialFC on MI355X — class-sharded, negatively-sampled margin softmax with hand-written gradient.

Same constructor / attributes / ``forward_backward(label, features, optimizer) -> (x_grad, loss_v)`` /
``update()`` / ``save_params()`` as the reference (reference partial_fc.py:19-176).  The six collectives
(partial_fc.py:122,134,142,147,161,173) go through ``torch.distributed`` (backend "nccl" == RCCL over xGMI);
sampling (RNG, top-k select, ordered compaction, label remap), row gather/scatter, normalisation, the cosine
GEMMs and the margin/softmax/grad are HIP kernels (fedfr_pfc_*, fedfr_rows_*, fedfr_sgemm, fedfr_margin_*).
"""
from __future__ import annotations

import logging
import os
from typing import Optional

import torch
from torch.nn import Module
from torch.nn.parameter import Parameter

from . import _C, ops
from .comm import default_comm

f32 = torch.float32


class PartialFC(Module):
    @torch.no_grad()
    def __init__(self, rank, local_rank, world_size, batch_size, resume, margin_softmax, num_classes, sample_rate=1.0,
                 embedding_size=512, prefix="./", seed: int = 100, comm=None):
        """Reference signature (partial_fc.py:20-23) + two build extensions: ``seed`` of the counter-based sampling RNG and ``comm``
        (fedfr_amd.comm: the exchange layer; default = the torch.distributed process group when world_size > 1)."""
        super().__init__()
        self.comm = comm if comm is not None else default_comm(world_size)
        if self.comm.world_size != world_size or self.comm.rank != rank:
            raise ValueError("PartialFC: rank/world_size (%d/%d) disagree with the communicator (%d/%d)"
                             % (rank, world_size, self.comm.rank, self.comm.world_size))
        # public attributes of the reference object (partial_fc.py:24-39,63-69): callers and checkpoints rely on these names
        self.rank, self.local_rank, self.world_size = int(rank), int(local_rank), int(world_size)
        self.num_classes, self.batch_size, self.embedding_size = int(num_classes), int(batch_size), int(embedding_size)
        self.margin_softmax, self.sample_rate, self.prefix = margin_softmax, float(sample_rate), str(prefix)
        self.device = torch.device("cuda", self.local_rank)
        base, extra = divmod(self.num_classes, self.world_size)          # class c lives on the rank whose [class_start, +num_local) holds it:
        self.num_local = base + (1 if self.rank < extra else 0)           # the first `extra` ranks hold one class more (partial_fc.py:34-35)
        self.class_start = base * self.rank + min(self.rank, extra)
        self.num_sample = int(self.sample_rate * self.num_local)
        self.weight_name, self.weight_mom_name = (os.path.join(self.prefix, "rank:%d_softmax_weight%s.pt" % (self.rank, sfx)) for sfx in ("", "_mom"))
        log = logging.getLogger("FL_face.partial")
        self.weight = self._restore(self.weight_name, "softmax weight", log) if resume else None
        if self.weight is None:
            self.weight = torch.normal(0, 0.01, (self.num_local, self.embedding_size), device=self.device)          # partial_fc.py:45,56
        self.weight_mom = self._restore(self.weight_mom_name, "softmax weight mom", log) if resume else None
        if self.weight_mom is None or self.weight_mom.shape != self.weight.shape:
            self.weight_mom = torch.zeros_like(self.weight)
        # The reference overlaps label gather + sampling on a side stream (partial_fc.py:61,119).  Here sampling is ~30 us of
        # kernels, and side-stream allocations consumed by main-stream kernels would need record_stream() bookkeeping, so
        # everything is enqueued on the caller's current stream; the attribute is kept for API compatibility.
        self.stream = torch.cuda.current_stream(self.device)
        self.index = None
        self._seed, self._step = int(seed) * 1000003 + self.rank, 0
        self._perm = torch.empty(self.num_local, dtype=f32, device=self.device)
        self._npos = torch.zeros(1, dtype=torch.int32, device=self.device)
        # margin parameters for the fused softmax kernels
        name = margin_softmax.__class__.__name__
        if name not in ("CosFace", "ArcFace"):
            raise ValueError("margin_softmax must be a fedfr_amd.losses.CosFace/ArcFace instance")
        self._arc, self._s, self._m = name == "ArcFace", float(margin_softmax.s), float(margin_softmax.m)
        self._sampling = int(self.sample_rate) != 1
        if self._sampling:
            self.sub_weight = Parameter(torch.empty((0, 0), device=self.device))          # filled by sample() every step
        else:                                                                            # sample_rate 1: the whole shard is the "sample"
            self._alias_all_rows()
            self.update = lambda: 0                                                       # nothing to scatter back (partial_fc.py:64-67)

    def _restore(self, path, what, log):
        """one tensor of a resumed run (partial_fc.py:41-53), or None when the file is missing / unreadable."""
        try:
            t = torch.load(path).to(self.device)
        except (FileNotFoundError, KeyError, IndexError):
            log.info("%s resume fail!", what)
            return None
        log.info("%s resume successfully!", what)
        return t

    def _alias_all_rows(self):
        self.sub_weight = Parameter(self.weight)
        self.sub_weight_mom = self.weight_mom

    # ------------------------------------------------------------------ persistence: the reference's file names (partial_fc.py:71-87)
    def _fc_path(self):
        return os.path.join(self.prefix, "FC_rank_%d.pth" % self.local_rank)

    def save_params(self):
        for t, path in ((self.weight.data, self.weight_name), (self.weight_mom, self.weight_mom_name)):
            torch.save(t, path)

    def save_FC(self):
        torch.save(self.weight.data, self._fc_path())

    def update_from_tensor(self, tensor):
        self.weight.data = tensor.to(self.device)
        self.sub_weight = Parameter(self.weight)

    def update_FC(self):
        self.update_from_tensor(torch.load(self._fc_path()))

    # ------------------------------------------------------------------ sampling (partial_fc.py:89-106)
    @torch.no_grad()
    def sample(self, total_label, perm: Optional[torch.Tensor] = None):
        """In place on ``total_label``: labels outside this shard -> -1, inside -> local (then sampled) ids.
        ``perm`` injects the uniform draw (parity tests); by default a counter-based HIP RNG fills it."""
        st = _C.stream()
        n = total_label.numel()
        sampling = self._sampling
        if sampling:
            if perm is not None:
                self._perm.copy_(perm)
            else:
                _C.call("fedfr_pfc_rand", self._perm.data_ptr(), self.num_local, self._seed, self._step, st)
            self._step += 1
        _C.call("fedfr_pfc_localize", total_label.data_ptr(), n, self.class_start, self.num_local,
                self._perm.data_ptr() if sampling else None, st)
        if not sampling:
            return
        index = torch.empty(self.num_sample, dtype=torch.int64, device=self.device)
        _C.call("fedfr_pfc_topk", self._perm.data_ptr(), self.num_local, self.num_sample, index.data_ptr(), self._npos.data_ptr(), st)
        # more positives than samples -> index = the positives (:101-102), a data-dependent SHAPE.  It cannot happen while the global
        # batch is no larger than num_sample (npos <= B*W), which is every BASELINE config: then no host sync at all
        npos = int(self._npos.item()) if n > self.num_sample else 0
        if npos > self.num_sample:
            index = torch.empty(npos, dtype=torch.int64, device=self.device)
            _C.call("fedfr_pfc_positive", self._perm.data_ptr(), self.num_local, index.data_ptr(), self._npos.data_ptr(), st)
        self.index = index
        _C.call("fedfr_pfc_remap", total_label.data_ptr(), n, index.data_ptr(), index.numel(), st)
        k = index.numel()
        sub_w = torch.empty(k, self.embedding_size, dtype=f32, device=self.device)
        sub_m = torch.empty(k, self.embedding_size, dtype=f32, device=self.device)
        _C.call("fedfr_rows_gather", sub_w.data_ptr(), self.weight.data_ptr(), index.data_ptr(), k, self.embedding_size, self.num_local, st)
        _C.call("fedfr_rows_gather", sub_m.data_ptr(), self.weight_mom.data_ptr(), index.data_ptr(), k, self.embedding_size, self.num_local, st)
        self.sub_weight = Parameter(sub_w)
        self.sub_weight_mom = sub_m

    def forward(self, total_features, norm_weight):
        return ops.sgemm(total_features, norm_weight, trans_b=True)

    @torch.no_grad()
    @_C.on_device(lambda self: self.device)
    def update(self):
        """scatter the sampled rows back (partial_fc.py:113-116)."""
        st = _C.stream()
        k = self.index.numel()
        sw = self.sub_weight.data.contiguous()
        _C.call("fedfr_rows_scatter", self.weight_mom.data_ptr(), self.sub_weight_mom.data_ptr(), self.index.data_ptr(), k,
                self.embedding_size, self.num_local, st)
        _C.call("fedfr_rows_scatter", self.weight.data_ptr(), sw.data_ptr(), self.index.data_ptr(), k, self.embedding_size,
                self.num_local, st)

    # ------------------------------------------------------------------ exchange points
    # The reference has six (partial_fc.py:122,134,142,147,161,173); packed here into four collectives per step:
    #   G  all-gather of [features | label bits]            (C1 + C2: one [B, D+2] fp32 tensor, the int64 label bit-cast into 2 lanes)
    #   M  max all-reduce of the row maxima                 (C3)
    #   S  sum all-reduce of [row sums | target numerators] (C4 + C5: one [2, B*W] tensor)
    #   R  reduce-scatter of d(total_features)              (C6)
    def _gather_inputs(self, label, features):
        W = self.world_size
        if W == 1:
            return label.to(torch.int64).clone(), features
        if not getattr(self.comm, "bitwise_gather", True):       # gloo debug path: its device "all-gather" is an arithmetic all-reduce
            return self.comm.all_gather(label.to(torch.int64).contiguous()), self.comm.all_gather(features)
        B, D = features.shape
        packed = torch.empty(B, D + 2, dtype=f32, device=features.device)
        packed[:, :D] = features
        packed[:, D:] = label.to(torch.int64).contiguous().view(torch.int32).view(B, 2).view(f32)    # exact bit copy
        tot = self.comm.all_gather(packed)                                                    # G
        total_label = tot[:, D:].contiguous().view(torch.int32).view(-1).view(torch.int64).clone()
        return total_label, tot[:, :D].contiguous()

    def prepare(self, label, optimizer, perm=None, total_label=None):
        """partial_fc.py:118-128: gather labels, sample, alias the sampled rows into the optimiser's last group."""
        if total_label is None:
            total_label = self.comm.all_gather(label.to(torch.int64).contiguous())           # C1 alone (callers without features)
        self.sample(total_label, perm)
        if optimizer is not None:
            optimizer.state.pop(optimizer.param_groups[-1]["params"][0], None)
            optimizer.param_groups[-1]["params"][0] = self.sub_weight
            optimizer.state[self.sub_weight]["momentum_buffer"] = self.sub_weight_mom
        norm_weight, winv = ops.normalize_rows(self.sub_weight.data)
        return total_label, norm_weight, winv

    @_C.on_device(lambda self: self.device)
    def forward_backward(self, label, features, optimizer, perm=None, x_grad_slabs=False):
        """partial_fc.py:130-176.  Returns (x_grad [B, D], loss_v scalar); ``self.sub_weight.grad`` is set.  ``x_grad_slabs`` (build extension, world
        size 1): x_grad comes back as the split-K slabs [S, B, D] of its GEMM for a consumer that adds them itself (``ops.normalize_rows_bwd_slabs``).
        (Moving the loss value and d(sampled class weights) behind the backbone's backward pass onto the weight-gradient stream, as the dense head
        does, measured neutral in round 5 — 15.51 vs 15.52 ms — and was not kept.)"""
        features = _C.require_gpu_tensor(features.detach().contiguous(), f32, "features")
        W = self.world_size
        total_label, total_features = self._gather_inputs(label, features)                  # G
        total_label, norm_weight, winv = self.prepare(label, optimizer, perm, total_label)
        logits = self.forward(total_features, norm_weight)
        inv_batch = 1.0 / (self.batch_size * W)
        if W > 1:
            loss_v, grad = ops.sharded_softmax_ce_grad(logits, total_label, self._s, self._m, self._arc, inv_batch,
                                                       self.comm.all_reduce, 1e-30)                  # M, S
        else:
            if logits.shape[1] <= 4096:               # margin -> softmax -> gradient with the row in registers: one launch, the same bits as the three
                # (longer rows: one workgroup per row leaves half the CUs idle at batch 128 — 46 us fused vs 38 for the three at 8 500 classes)
                prob_t, grad = ops.softmax_ce_fused(logits.unsqueeze(0), total_label, self._s, self._m, self._arc, inv_batch)
            else:
                prob_t, grad = ops.softmax_ce_grad(logits, total_label, self._s, self._m, self._arc, inv_batch)
            loss_v = ops.nll_mean(prob_t, 1e-30)
        # logits.backward(grad): d total_features = grad @ norm_weight ; d norm_weight = grad^T @ total_features
        # d total_features [B W, D] reduces over the SAMPLED CLASSES (8 500 at 85 000 x 0.1): on 64 x 64 tiles that is 16 workgroups walking 8 500
        # k-steps each (405 us in the round-5 trace) — split K like the dense head's GEMMs (round 3), here into up to 32 slabs
        K_ = grad.shape[1]
        ks = max(1, min(32, K_ // 256))
        chunk = -(-(-(-K_ // ks)) // 32) * 32
        ks = -(-K_ // chunk)                              # (no empty last chunk: head_sgemm_splitk refuses one)
        if ks > 1:
            dslabs = ops.sgemm(grad, norm_weight, splits=ks)
            dfeat = dslabs if (x_grad_slabs and W == 1) else ops.sum_slabs(dslabs)
        else:
            dfeat = ops.sgemm(grad, norm_weight)
            if x_grad_slabs and W == 1:
                dfeat = dfeat.unsqueeze(0)
        dwn = ops.sgemm(grad, total_features, trans_a=True)
        self.sub_weight.grad = ops.normalize_rows_bwd(norm_weight, winv, dwn)
        if W > 1:
            x_grad = self.comm.reduce_scatter(dfeat)                                          # R
            x_grad = ops.scale(x_grad, float(W))                                              # partial_fc.py:174
        else:
            x_grad = dfeat
        return x_grad, loss_v

    @torch.no_grad()
    @_C.on_device(lambda self: self.device)
    def fused_sgd_update(self, lr, momentum=0.9, weight_decay=5e-4, overflow=None):
        """Caller-side ``opt.step(); pfc.update()`` for the sampled rows as two HIP calls (momentum rows already
        exist, so this is never a 'first' step — partial_fc.py:124-126).  ``overflow``: the fp16-storage library's device word
        (client._LossScaleGuard) — the update then skips and flags non-finite gradient elements like every other update of that library."""
        sw = self.sub_weight.data
        if overflow is not None:
            _C.call("fedfr_sgd_step_scaled", sw.data_ptr(), self.sub_weight.grad.data_ptr(), self.sub_weight_mom.data_ptr(), None, sw.numel(),
                    float(lr), float(momentum), float(weight_decay), 0, 1.0, overflow.data_ptr(), _C.stream())
        else:
            _C.call("fedfr_sgd_step", sw.data_ptr(), self.sub_weight.grad.data_ptr(), self.sub_weight_mom.data_ptr(), None, sw.numel(),
                    float(lr), float(momentum), float(weight_decay), 0, _C.stream())
        if int(self.sample_rate) != 1:
            self.update()
