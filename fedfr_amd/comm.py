"""The collectives of the path, behind one small interface.

The reference calls ``torch.distributed`` directly at six points of ``PartialFC`` (partial_fc.py:122,134,142,147,161,173) and
exchanges models through Python lists (server.py:25-34).  Here every exchange goes through a ``Comm``:

* ``TorchDistComm`` — one process per GPU, ``torch.distributed`` process group (backend "nccl" == RCCL over xGMI).  The gloo backend is
  kept as a debug / CPU-test path (device tensors move through padded all-reduces there: gloo has no device all-gather).
* ``ThreadComm``    — W simulated ranks as W Python threads of ONE process on ONE device.  Deterministic rank-ordered reductions.
  This is what lets a single-GPU box run the 8-rank configurations (BASELINE config 5) end to end: the GPU box admits at most six
  GPU processes, and RCCL refuses two ranks on one device.  All ranks enqueue on the same HIP stream, so a tensor deposited before the
  thread barrier is ordered before every kernel another rank enqueues after it.
* ``SingleComm``    — world size 1.

Verbs (all the path needs): ``all_gather(t) -> [W * t.shape[0], ...]``, ``all_reduce(t, op)`` in place with op in {"sum", "max"},
``reduce_scatter(full) -> this rank's [full.shape[0] / W, ...] slice of the sum``, ``barrier()``.
"""
from __future__ import annotations

import threading
from typing import List, Optional

import torch

__all__ = ["SingleComm", "TorchDistComm", "ThreadComm", "default_comm"]


class SingleComm:
    rank, world_size = 0, 1

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        return t.clone()

    def all_reduce(self, t: torch.Tensor, op: str = "sum") -> torch.Tensor:
        return t

    def reduce_scatter(self, full: torch.Tensor) -> torch.Tensor:
        return full

    def barrier(self) -> None:
        pass


class TorchDistComm:
    """torch.distributed-backed (RCCL over xGMI with backend "nccl")."""

    def __init__(self, group=None):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("TorchDistComm needs an initialised torch.distributed process group")
        self.dist, self.group = dist, group
        self.rank, self.world_size = dist.get_rank(group), dist.get_world_size(group)
        self._gloo = dist.get_backend(group) == "gloo"
        self.bitwise_gather = not self._gloo       # all_gather is a pure byte copy (callers may pack bit-cast integers into float lanes)
        self.needs_device_tensors = not self._gloo  # RCCL moves device memory only

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        t = t.contiguous()
        if self._gloo and t.is_cuda:      # gloo moves device tensors only through broadcast / all_reduce (debug + test path)
            out = torch.zeros((self.world_size,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            out[self.rank] = t
            self.dist.all_reduce(out, group=self.group)
        else:                             # concatenated along dim 0: the output form every backend accepts
            out = torch.empty((self.world_size * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            self.dist.all_gather_into_tensor(out, t, group=self.group)
        return out.view((-1,) + tuple(t.shape[1:]))

    def all_reduce(self, t: torch.Tensor, op: str = "sum") -> torch.Tensor:
        self.dist.all_reduce(t, self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM, group=self.group)
        return t

    def reduce_scatter(self, full: torch.Tensor) -> torch.Tensor:
        full = full.contiguous()
        rows = full.shape[0] // self.world_size
        if self._gloo:
            self.dist.all_reduce(full, group=self.group)
            return full[self.rank * rows: (self.rank + 1) * rows].clone()
        out = torch.empty((rows,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
        self.dist.reduce_scatter_tensor(out, full, group=self.group)
        return out

    def barrier(self) -> None:
        self.dist.barrier(group=self.group)


class _ThreadWorld:
    def __init__(self, world_size: int):
        self.world_size = world_size
        self.slots: List[Optional[torch.Tensor]] = [None] * world_size
        self.result: Optional[torch.Tensor] = None
        self.bar = threading.Barrier(world_size)


class ThreadComm:
    """Rank ``rank`` of a W-rank world simulated by W threads of this process (``ThreadComm.world(W)`` makes the W handles; run each
    rank's code in its own thread, e.g. with ``ThreadComm.run``).  Reductions add the ranks' tensors in ascending rank order."""

    def __init__(self, world: _ThreadWorld, rank: int):
        self._w, self.rank, self.world_size = world, rank, world.world_size

    @staticmethod
    def world(world_size: int) -> List["ThreadComm"]:
        w = _ThreadWorld(world_size)
        return [ThreadComm(w, r) for r in range(world_size)]

    @staticmethod
    def run(world_size: int, fn, device=None, stream=None, timeout: float = 600.0):
        """fn(comm) on world_size threads; returns the list of results by rank, re-raises the first failure.  Every thread makes
        ``device`` / ``stream`` current (default: the caller's current stream), so all ranks enqueue on ONE stream."""
        comms = ThreadComm.world(world_size)
        out, err = [None] * world_size, [None] * world_size
        if device is not None and torch.device(device).type == "cuda" and stream is None:
            stream = torch.cuda.current_stream(device)

        def target(r):
            try:
                if device is not None and torch.device(device).type == "cuda":
                    torch.cuda.set_device(device)
                    torch.cuda.set_stream(stream)
                out[r] = fn(comms[r])
            except BaseException as e:          # noqa: BLE001 — handed to the caller
                err[r] = e
                comms[r]._w.bar.abort()
        ts = [threading.Thread(target=target, args=(r,), daemon=True) for r in range(world_size)]
        for t in ts:
            t.start()
        import time
        deadline = time.monotonic() + timeout       # ONE deadline for the whole world, not `timeout` per rank
        for t in ts:
            t.join(max(0.0, deadline - time.monotonic()))
        if any(t.is_alive() for t in ts):
            # release every rank still parked in a collective (its barrier wait raises BrokenBarrierError and the thread ends) before
            # raising: a live rank thread must not keep enqueueing GPU work on the caller's stream
            comms[0]._w.bar.abort()
            for t in ts:
                t.join(5.0)
            raise TimeoutError("ThreadComm.run: a rank did not finish within %.0f s" % timeout)
        real = [e for e in err if e is not None and not isinstance(e, threading.BrokenBarrierError)]
        if real or any(e is not None for e in err):
            raise (real[0] if real else [e for e in err if e is not None][0])
        return out

    def _exchange(self, t: torch.Tensor) -> List[torch.Tensor]:
        w = self._w
        w.slots[self.rank] = t
        w.bar.wait()
        parts = list(w.slots)
        w.bar.wait()                 # nobody overwrites a slot before everyone has read all of them
        return parts

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        return torch.cat([p.reshape((-1,) + tuple(t.shape[1:])) for p in self._exchange(t.contiguous())], dim=0)

    def all_reduce(self, t: torch.Tensor, op: str = "sum") -> torch.Tensor:
        w = self._w
        parts = self._exchange(t)
        if self.rank == 0:           # one rank reduces (ascending rank order), everyone copies the result
            acc = parts[0].clone()
            for p in parts[1:]:
                acc = torch.maximum(acc, p) if op == "max" else acc + p
            w.result = acc
        w.bar.wait()
        t.copy_(w.result)
        w.bar.wait()
        return t

    def reduce_scatter(self, full: torch.Tensor) -> torch.Tensor:
        parts = self._exchange(full.contiguous())
        rows = full.shape[0] // self.world_size
        sl = slice(self.rank * rows, (self.rank + 1) * rows)
        acc = parts[0][sl].clone()
        for p in parts[1:]:
            acc = acc + p[sl]
        self._w.bar.wait()           # every rank has read its slices before anyone reuses `full`
        return acc

    def barrier(self) -> None:
        self._w.bar.wait()


def default_comm(world_size: int):
    """The comm a reference-shaped constructor gets when none is passed: torch.distributed if a group is up, else single."""
    import torch.distributed as dist
    if world_size > 1:
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("world_size %d needs an initialised torch.distributed process group (or pass comm=...)" % world_size)
        c = TorchDistComm()
        if c.world_size != world_size:
            raise RuntimeError("world_size %d != process group size %d" % (world_size, c.world_size))
        return c
    return SingleComm()
