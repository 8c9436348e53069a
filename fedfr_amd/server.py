"""Server side of the federated loop (reference server.py:25-46, :265-338): dataset-size-weighted
averaging of client models, as HIP kernels over flat buffers (single process) or one RCCL all-reduce over
xGMI when every client is its own rank (one client = one MI355X)."""
from __future__ import annotations

import copy
import logging
import random
from collections import OrderedDict
from typing import List, Sequence

import numpy as np
import torch

from . import _C, backbones
from .client import FlatStateDict, flat_state_dict

f32 = torch.float32


def _axpy(dst: torch.Tensor, src: torch.Tensor, w: float, accumulate: bool):
    _C.call("fedfr_fedavg_axpy", dst.data_ptr(), src.data_ptr(), float(np.float32(w)), dst.numel(), 1 if accumulate else 0,
            _C.stream())


def _multi(dst: torch.Tensor, srcs: Sequence[torch.Tensor], ws: Sequence[float], accumulate: bool):
    """dst (+)= Σ_i ws[i]·srcs[i] over ≤ 8 same-shape contiguous fp32 tensors in one kernel (csrc/optim.hip: fedavg_multi_kernel)."""
    import ctypes as C
    k = len(srcs)
    for t in srcs:
        if t.numel() != dst.numel() or t.dtype != f32 or not t.is_contiguous() or t.device != dst.device:
            raise RuntimeError("fedfr_amd.FedPavg: client states must be same-shape contiguous fp32 tensors on one device")
    ptrs = (C.c_void_p * k)(*[t.data_ptr() for t in srcs])
    wv = (C.c_float * k)(*[float(np.float32(w)) for w in ws])
    _C.call("fedfr_fedavg_multi", dst.data_ptr(), ptrs, wv, k, dst.numel(), 1 if accumulate else 0, _C.stream())


def FedPavg(models: List[dict], weights: Sequence[float]):
    """Σ_i (n_i/Σn)·sd_i[k] for every key k (reference server.py:25-34).

    Same op order as the reference (ascending client index, fp32 mul then add), so float entries are
    bit-identical to it.  int64 ``num_batches_tracked`` entries come back as float32, like the reference (F9).
    Fast path: FlatStateDicts → one pass over up to 8 clients' flat states (3 launches per 8 clients + one per counter vector); generic path: one kernel per key per client.
    """
    tot = sum(weights)
    ws = [w / tot for w in weights]
    if all(isinstance(m, FlatStateDict) and m.flat is not None for m in models):
        p0, b0, n0 = models[0].flat
        dev = p0.device
        if not p0.is_cuda:
            raise RuntimeError("fedfr_amd.FedPavg: state tensors must be on the GPU (no CPU fallback)")
        P, Bf = torch.empty_like(p0), torch.empty_like(b0)
        N = torch.empty(n0.numel(), dtype=f32, device=dev)
        # up to 8 client states per pass (fedfr_fedavg_multi: every state read once, the aggregate written once; same op order and
        # roundings as one axpy per client, so still bit-identical to the reference loop)
        for c0 in range(0, len(models), 8):
            grp, wgrp = models[c0:c0 + 8], ws[c0:c0 + 8]
            _multi(P, [m.flat[0] for m in grp], wgrp, c0 > 0)
            if Bf.numel():                          # (sphnet has no BatchNorm: no running statistics, no counters)
                _multi(Bf, [m.flat[1] for m in grp], wgrp, c0 > 0)
        for i, (m, w) in enumerate(zip(models, ws)):
            n = m.flat[2]
            if n.numel():
                _C.call("fedfr_fedavg_i64", N.data_ptr(), n.data_ptr(), float(np.float32(w)), n.numel(), 1 if i else 0, None, _C.stream())
        return FlatStateDict.from_flat((P, Bf, N), models[0].table, models[0].layers)
    aggr = OrderedDict()
    for name in models[0]:
        t0 = models[0][name]
        if not t0.is_cuda:
            raise RuntimeError("fedfr_amd.FedPavg: tensor '%s' is on %s; move state_dicts to the GPU" % (name, t0.device))
        if t0.is_floating_point():
            acc = torch.empty(t0.shape, dtype=f32, device=t0.device)
            for i, (m, w) in enumerate(zip(models, ws)):
                _axpy(acc, m[name].contiguous(), w, i > 0)
        else:
            acc = torch.empty(t0.shape, dtype=f32, device=t0.device)
            for i, (m, w) in enumerate(zip(models, ws)):
                src = m[name].contiguous().view(-1)
                _C.call("fedfr_fedavg_i64", acc.data_ptr(), src.data_ptr(), float(np.float32(w)), src.numel(), 1 if i else 0, None,
                        _C.stream())
        aggr[name] = acc
    return aggr


def FedAvg_on_FC(pretrain_fc, models, weights, p):
    """reference server.py:36-46."""
    tot = sum(weights)
    ws = [w / tot for w in weights]
    m0 = models[0].contiguous()
    if not m0.is_cuda:
        raise RuntimeError("fedfr_amd.FedAvg_on_FC: tensors must be on the GPU")
    aggr = torch.empty_like(m0)
    for i, (m, w) in enumerate(zip(models, ws)):
        _axpy(aggr, m.contiguous(), w, i > 0)
    if p == 1:
        return aggr
    out = torch.empty_like(aggr)
    _axpy(out, pretrain_fc.contiguous(), 1 - p, False)
    _axpy(out, aggr, p, True)
    return out


def _i64_scale(acc: torch.Tensor, src: torch.Tensor, w: float):
    _C.call("fedfr_fedavg_i64", acc.data_ptr(), src.data_ptr(), float(np.float32(w)), src.numel(), 0, None, _C.stream())


def _i64_trunc(acc: torch.Tensor, dst: torch.Tensor):
    """dst (int64) = trunc(acc) — load_state_dict's float -> int64 copy of the averaged num_batches_tracked (F9)."""
    _C.call("fedfr_fedavg_i64", acc.data_ptr(), dst.data_ptr(), 0.0, dst.numel(), 1, dst.data_ptr(), _C.stream())


def exchange_data_sizes(data_size: float, comm) -> float:
    """Σ n_j over the ranks (one tiny all-reduce + host read).  Run it when the sizes become known — at round start, off the exchange
    path — and hand the result to ``fedavg_all_reduce``; ranks that know every client's size in advance skip it."""
    dev = torch.device("cuda", torch.cuda.current_device()) if getattr(comm, "needs_device_tensors", False) else torch.device("cpu")
    t = torch.tensor([float(data_size)], dtype=torch.float64, device=dev)
    t = comm.all_reduce(t, "sum")
    return float(t.item())


def fedavg_all_reduce(backbone, data_size: float, total_size: float, comm=None, _axpy=_axpy, _i64=_i64_scale, _trunc=_i64_trunc):
    """One client per rank: the FedAvg of a round as ONE collective (replaces the CPU loop of server.py:25-34 and the state_dict
    hand-offs of server.py:286,311).  The local state — parameters, BN running statistics and a float image of the
    ``num_batches_tracked`` counters, which are slices of one fp32 tensor (``IResNet.exchange_buffer``) — is scaled in place by the
    pre-agreed weight n_i / Σn (the reference's ``weights[i] / sum(weights)``, server.py:27) and SUM-all-reduced in place over
    RCCL/xGMI: 261 MB for iresnet100, no packing copies, no host synchronisation.  The counters are truncated back to int64 as
    ``load_state_dict`` does (F9).  The summation ORDER is the collective's (a ring), not the reference's ascending client index: results
    agree with ``FedPavg`` to fp32 rounding, not bit for bit.
    ``comm``: fedfr_amd.comm communicator (default: the torch.distributed world).  ``_axpy`` / ``_i64`` / ``_trunc`` are the HIP
    kernels (injectable only so the plumbing can be exercised by the gloo CPU test)."""
    from .comm import TorchDistComm
    if comm is None:
        comm = TorchDistComm()
    state, nbt_f, nbt = backbone.exchange_buffer()
    w = float(data_size) / float(total_size)
    # params + running stats = everything in front of the counter image.  Storage offsets, not pointer differences: a backbone without
    # BatchNorm counters (sphnet) has an EMPTY image, and data_ptr() of an empty tensor is 0 on torch >= 2.x
    n_float = nbt_f.storage_offset() - state.storage_offset()
    assert 0 <= n_float <= state.numel(), "exchange_buffer: the counter image is not a slice of the state tensor"
    fl = state[:n_float]
    _axpy(fl, fl, w, False)
    if nbt.numel():
        _i64(nbt_f, nbt, w)
    comm.all_reduce(state, "sum")                  # THE exchange of the round
    if nbt.numel():
        _trunc(nbt_f, nbt)
    backbone.mark_weights_dirty()
    return w


class Server(object):
    """Round driver (reference server.py:68-133, :265-338): clients are trained sequentially in one process (as the
    reference does, server.py:283) and averaged with ``FedPavg``; with ``args.add_pretrained_data`` every client trains on
    local + public identities (``Client.train_with_public_data``) and, with ``args.return_all``, the public class centres
    are averaged with ``FedAvg_on_FC`` (server.py:316-327).  With ``server.pretrained_label`` set (``Initialize_pretrain_FC``)
    every round starts with the public-set embedding sweep (``Generate_pretrain_feats``) and the clients mine hard negatives
    from it (``Client.choose_hard_negative_2``), as server.py:272-275 / :294-304 do."""

    def __init__(self, clients, data, args, device=None):
        self.data = data
        self.clients = clients
        self.args = args
        self.num_client = len(clients)
        self.local_epoch = args.local_epoch
        self.global_epoch = 0
        self.global_round = 0
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.federated_model = getattr(backbones, args.network)(False, dropout=0, fp16=True).to(self.device)
        self.current_client_list = list(range(self.num_client))
        self.logger = logging.getLogger("FL_face.server")
        self.public_train_loader = getattr(data, "public_train_loader", None)
        self.public_test_loader = getattr(data, "public_test_loader", self.public_train_loader)
        self.pretrained_label = None                 # [N_public] identity of every public image
        self.pretrained_feats = None                 # [N_public, 512] normalised embeddings (hard-negative mining)
        self.pretrained_fc = None                    # [n_public, 512] class centres of the public identities (server.py:182-240)
        self.pretrain_fc = None                      # where the reference stores the FedAvg'd public centres (server.py:325, see train())

    # ---- public-set inference sweeps (reference server.py:182-263)
    def _eval_backbone(self):
        from .client import shared_backbone
        bb = shared_backbone(self.args.network, self.device, 0)       # the process-wide resident instance (one set of arenas)
        bb.load_state_dict(flat_state_dict(self.federated_model))
        return bb.eval()

    @torch.no_grad()
    @_C.on_device(lambda self: self.device)
    def Generate_pretrain_feats(self):
        """normalised embeddings of the whole public set under the current global model (server.py:242-263); stays on the GPU."""
        from .client import embed_dataset
        feats, _ = embed_dataset(self._eval_backbone(), self.public_test_loader, self.device, normalize=True)
        return feats

    @torch.no_grad()
    @_C.on_device(lambda self: self.device)
    def Initialize_pretrain_FC(self, only_labels=False):
        """(init_matrix [n_public_ID, 512], raw_labels [N]) — per-identity mean embedding of the public set (server.py:182-240).
        The reference's optional .pth cache (load_pth / save_pth) is checkpoint I/O, outside this path."""
        from .client import class_centers
        if only_labels:
            return None, torch.cat([torch.as_tensor(l) for _, l in self.public_test_loader]).to(torch.int64)
        n_id = self.public_test_loader.dataset.num_classes
        init_matrix, raw_labels = class_centers(self._eval_backbone(), self.public_test_loader, n_id, self.device,
                                                getattr(self, "norm_before_avg", getattr(self.args, "norm_before_avg", True)))
        return init_matrix, raw_labels.cpu()

    @_C.on_device(lambda self: self.device)
    def train(self):
        from .config import config as cfg
        models, models_fc, losses_, data_sizes = [], [], [], []
        public = bool(getattr(self.args, "add_pretrained_data", False))
        return_all = bool(getattr(self.args, "return_all", False))
        mine = public and bool(getattr(self.args, "choose_hard_negative", True)) and self.public_test_loader is not None \
            and self.pretrained_label is not None
        if mine:                                                                                 # server.py:272-275
            self.pretrained_feats = self.Generate_pretrain_feats()
        if getattr(self.args, "adaptive_local_epoch", False) and self.global_round != 0:        # server.py:277-280
            self.local_epoch = max(4, self.local_epoch - 2)
            cfg.train_decay = max(1, int(3 / 4 * self.local_epoch))
        def run_client(i, slot=0):
            c = self.clients[i]
            c.slot = slot
            c.backbone_state_dict = flat_state_dict(self.federated_model)       # "server sends backbone"
            c.local_epoch = self.local_epoch
            if public:
                if self.pretrained_fc is None:
                    raise RuntimeError("Server.train: add_pretrained_data needs server.pretrained_fc ([n_public, 512] class centres)")
                c.train_with_public_data(self.global_epoch, public_train_loader=self.public_train_loader,
                                         pretrained_fc=self.pretrained_fc, choose_hard_negative=mine,
                                         pretrained_label=self.pretrained_label, pretrained_feats=self.pretrained_feats)
            else:
                c.train(self.global_epoch)

        par = max(1, int(getattr(self.args, "parallel_clients", 1)))
        order = list(self.current_client_list)
        if par == 1:
            for i in order:
                run_client(i)
        else:
            # The clients of a round are independent (the reference trains them one after another, server.py:283); one client's step
            # is a dependent chain of ~1250 short kernels that leaves CUs idle between launches.  `par` clients train CONCURRENTLY on
            # this GPU, each on its own HIP stream pair and resident backbone (measured: 2 clients = +20 % images/s on iresnet100 at
            # B = 128; 3 regress).  Kernels are deterministic and clients share no state, so the round's result is identical to the sequential
            # round with the same kernel selection (the paired weight-gradient kernel below: otherwise equal up to fp32 summation order).
            import threading
            # Kernel selection is the lone client's (the paired weight-gradient kernel has been the default since round 3).  No kernel of the library
            # waits for another workgroup of its own launch (round 5 removed the one option that did, bn_fuse_bwd), so grids of several clients may
            # compete for the CUs freely.
            main = torch.cuda.current_stream(self.device)
            streams = getattr(self, "_client_streams", None)
            if streams is None or len(streams) < par:
                streams = self._client_streams = [torch.cuda.Stream(device=self.device, priority=-1) for _ in range(par)]
            for w0 in range(0, len(order), par):
                errs = []

                def target(i, slot):
                    try:
                        torch.cuda.set_device(self.device)
                        streams[slot].wait_stream(main)
                        with torch.cuda.stream(streams[slot]):
                            run_client(i, slot)
                        streams[slot].synchronize()
                    except BaseException as e:      # noqa: BLE001 — re-raised in the caller's thread
                        errs.append(e)
                ts = [threading.Thread(target=target, args=(i, k), daemon=True) for k, i in enumerate(order[w0: w0 + par])]
                for t in ts:
                    t.start()
                for t in ts:
                    t.join()
                if errs:
                    raise errs[0]
        for i in order:
            losses_.append(self.clients[i].get_train_loss())
            models.append(self.clients[i].get_model())
            if return_all:
                models_fc.append(self.clients[i].get_global_fc().to(self.device))
                self.clients[i].fc_module.remove_pretrain()
            data_sizes.append(self.clients[i].get_data_size())
        self.avg_loss = sum(losses_) / len(losses_)
        if return_all:                                                                           # server.py:316-327
            # Reference quirk kept: the averaged public centres are assigned to `self.pretrain_fc` — a misspelling of
            # `self.pretrained_fc` (server.py:325 vs :121-124, :295) — so the clients of the NEXT round still receive the initial
            # centres.  `args.feedback_public_fc = True` (a build extension, off by default) feeds the average back.
            self.pretrain_fc = FedAvg_on_FC(self.pretrained_fc, models_fc, data_sizes, p=1.0)
            if getattr(self.args, "feedback_public_fc", False):
                self.pretrained_fc = self.pretrain_fc
        if getattr(self.args, "aggr_alg", "FedAvg") in ("FedAvg", "FedProx"):
            aggr_state_dict = FedPavg(models, data_sizes)
            self.federated_model.load_state_dict(aggr_state_dict)
        # the round / epoch counters belong to the driver, as in the reference (train.py:87-88): call step_round() after train()
        return self.avg_loss

    def step_round(self):
        """What the reference driver does after every ``server.train()`` (train.py:87-88)."""
        self.global_epoch += self.local_epoch
        self.global_round += 1
