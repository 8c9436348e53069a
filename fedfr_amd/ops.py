"""Thin tensor-level wrappers over the fp32 head kernels + autograd bridges.

Everything here runs in libfedfr_hip.so; torch is used only to own device memory and to connect the
hand-written backward passes to ``loss.backward()`` for callers that use the reference's eager style
(``logits = margin(fc(feat), y); F.cross_entropy(logits, y).backward()``).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _C

f32 = torch.float32


def _chk(t: torch.Tensor, name: str, dtype=f32) -> torch.Tensor:
    t = t if t.is_contiguous() else t.contiguous()
    return _C.require_gpu_tensor(t, dtype, name)


# ------------------------------------------------------------------------------------------ raw ops
def normalize_rows(x: torch.Tensor, eps: float = 1e-12) -> Tuple[torch.Tensor, torch.Tensor]:
    """F.normalize(x) (p=2, dim=1) -> (x_hat, 1/max(||x||, eps))."""
    x = _chk(x, "x")
    R, D = x.shape
    xn = torch.empty_like(x)
    inv = torch.empty(R, dtype=f32, device=x.device)
    _C.call("fedfr_normalize_rows", x.data_ptr(), xn.data_ptr(), inv.data_ptr(), R, D, eps, _C.stream())
    return xn, inv


def normalize_rows_bwd(xn: torch.Tensor, inv: torch.Tensor, dxn: torch.Tensor) -> torch.Tensor:
    dxn = _chk(dxn, "dxn")
    dx = torch.empty_like(xn)
    _C.call("fedfr_normalize_rows_bwd", xn.data_ptr(), inv.data_ptr(), dxn.data_ptr(), dx.data_ptr(), xn.shape[0], xn.shape[1],
            0.0, _C.stream())
    return dx


def normalize_rows_bwd_slabs(xn: torch.Tensor, inv: torch.Tensor, dxn_slabs: torch.Tensor) -> torch.Tensor:
    """``normalize_rows_bwd`` on a d(xn) that arrives as split-K slabs [S, R, D] (``sgemm(..., splits=S)``), added slab 0 first."""
    dxn_slabs = _chk(dxn_slabs, "dxn slabs")
    S, R, D = dxn_slabs.shape
    dx = torch.empty_like(xn)
    _C.call("fedfr_normalize_rows_bwd_slabs", xn.data_ptr(), inv.data_ptr(), dxn_slabs.data_ptr(), S, R * D, dx.data_ptr(), R, D, 0.0,
            _C.stream())
    return dx


def softmax_ce_fused(cos_slabs: torch.Tensor, label: torch.Tensor, s: float, m: float, arcface: bool, inv_batch: float):
    """``softmax_ce_grad`` (no collectives) as ONE launch; ``cos_slabs`` [S, R, C] (C <= 16384) are split-K slabs of the cosine matrix.
    Returns (prob_target [R], grad = cos_slabs[0], overwritten)."""
    cos_slabs = _chk(cos_slabs, "cosine slabs")
    label = _chk(label, "label", torch.int64)
    S, R, Cc = cos_slabs.shape
    prob_t = torch.empty(R, dtype=f32, device=cos_slabs.device)
    _C.call("fedfr_softmax_ce_fused", cos_slabs.data_ptr(), label.data_ptr(), R, Cc, Cc, s, m, 1 if arcface else 0, inv_batch,
            prob_t.data_ptr(), S, R * Cc, _C.stream())
    return prob_t, cos_slabs[0]


def _auto_splits(M: int, N: int, K: int) -> int:
    """split-K degree ``sgemm`` picks by itself: none while the output has >= 128 tiles of 64 x 64 or K is short; else enough slabs of >= 256 k
    (a multiple of 32) to put ~512 workgroups on the chip, at most 64, none of them empty."""
    tiles = -(-M // 64) * -(-N // 64)
    if tiles >= 128 or K < 1024 or (M * N) % 4:           # (sum_slabs adds 16-byte vectors: every slab must start on one)
        return 1
    s = max(1, min(64, K // 256, -(-512 // tiles)))
    chunk = -(-(-(-K // s)) // 32) * 32
    return -(-K // chunk)


def sgemm(a: torch.Tensor, b: torch.Tensor, trans_a: bool = False, trans_b: bool = False,
          bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, splits: int = 0) -> torch.Tensor:
    """C = op(A) @ op(B) (+bias) in exact fp32 (v_mfma_f32_16x16x4_f32); A, B row-major contiguous.  ``splits`` > 0: split-K, returns
    the slabs [splits, M, N] (their sum is C; consumers: ``softmax_ce_fused``, ``normalize_rows_bwd_slabs``)."""
    a, b = _chk(a, "A"), _chk(b, "B")
    if trans_a:
        K, M = a.shape
        sam, sak = 1, M
    else:
        M, K = a.shape
        sam, sak = K, 1
    if trans_b:
        N, Kb = b.shape
        sbk, sbn = 1, Kb
    else:
        Kb, N = b.shape
        sbk, sbn = N, 1
    if K != Kb:
        raise RuntimeError("sgemm: inner dimensions differ (%d vs %d)" % (K, Kb))
    if splits == 0 and bias is None and out is None:
        # A long reduction over few 64 x 64 output tiles (d(features) = d(cosine) @ class weights: 128 x 512 over 10 000 ... 85 000 classes is 16
        # workgroups walking K alone — 4.0 ms at 85 000 in the round-5 trace of config 5): split K into slabs and add them (ascending), as the
        # dense head's GEMMs have done since round 3.  Same products, another fp32 summation order.
        auto = _auto_splits(M, N, K)
        if auto > 1:
            return sum_slabs(sgemm(a, b, trans_a, trans_b, splits=auto))
    if splits > 0:
        if bias is not None or out is not None:
            raise RuntimeError("sgemm: split-K slabs take neither bias nor out")
        slabs = torch.empty(splits, M, N, dtype=f32, device=a.device)
        _C.call("fedfr_sgemm_splitk", a.data_ptr(), b.data_ptr(), slabs.data_ptr(), M, N, K, sam, sak, sbk, sbn, N, 1.0, splits, M * N,
                _C.stream())
        return slabs
    if out is None:
        out = torch.empty(M, N, dtype=f32, device=a.device)
    _C.call("fedfr_sgemm", a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, sam, sak, sbk, sbn, N, 1.0, 0.0,
            _C.ptr(bias), _C.stream())
    return out


def softmax_ce_grad(cosine: torch.Tensor, label: torch.Tensor, s: float, m: float, arcface: bool,
                    inv_batch: float, all_reduce=None):
    """In place on ``cosine`` [R,C]: margin -> softmax -> gradient wrt the cosine matrix.
    Returns (prob_target [R], grad == cosine storage).  ``all_reduce(t, op)`` (op in {"max","sum"}) is
    applied to the row max / row sum / target prob between the three kernels (PartialFC C3-C5)."""
    cosine = _chk(cosine, "cosine")
    label = _chk(label, "label", torch.int64)
    R, Cc = cosine.shape
    dev = cosine.device
    row_max = torch.empty(R, dtype=f32, device=dev)
    row_sum = torch.empty(R, dtype=f32, device=dev)
    dmul = torch.empty(R, dtype=f32, device=dev)
    prob_t = torch.empty(R, dtype=f32, device=dev)
    st = _C.stream()
    _C.call("fedfr_margin_rowmax", cosine.data_ptr(), label.data_ptr(), R, Cc, Cc, s, m, 1 if arcface else 0,
            row_max.data_ptr(), dmul.data_ptr(), st)
    if all_reduce is not None:
        all_reduce(row_max, "max")
    _C.call("fedfr_exp_rowsum", cosine.data_ptr(), R, Cc, Cc, row_max.data_ptr(), row_sum.data_ptr(), st)
    if all_reduce is not None:
        all_reduce(row_sum, "sum")
    _C.call("fedfr_softmax_grad", cosine.data_ptr(), label.data_ptr(), R, Cc, Cc, row_sum.data_ptr(), dmul.data_ptr(), s,
            inv_batch, prob_t.data_ptr(), st)
    if all_reduce is not None:
        all_reduce(prob_t, "sum")
    return prob_t, cosine


def sharded_softmax_ce_grad(cosine: torch.Tensor, label: torch.Tensor, s: float, m: float, arcface: bool, inv_batch: float,
                            all_reduce, floor: float = 1e-30):
    """The class-sharded form (PartialFC, partial_fc.py:138-166) with its three reductions packed into two collectives: in place on
    this rank's ``cosine`` [R, C_local] (labels already localised, -1 = class lives elsewhere): margin -> global row max (MAX
    all-reduce) -> exponentials; ONE SUM all-reduce of [row sums | target numerators]; gradient wrt the cosine matrix.
    Returns (loss_v scalar = -mean log max(p_target, floor), grad == cosine storage)."""
    cosine = _chk(cosine, "cosine")
    label = _chk(label, "label", torch.int64)
    R, Cc = cosine.shape
    dev = cosine.device
    row_max = torch.empty(R, dtype=f32, device=dev)
    sums2 = torch.empty(2, R, dtype=f32, device=dev)
    dmul = torch.empty(R, dtype=f32, device=dev)
    prob_t = torch.empty(R, dtype=f32, device=dev)
    loss = torch.empty((), dtype=f32, device=dev)
    st = _C.stream()
    _C.call("fedfr_margin_rowmax", cosine.data_ptr(), label.data_ptr(), R, Cc, Cc, s, m, 1 if arcface else 0,
            row_max.data_ptr(), dmul.data_ptr(), st)
    all_reduce(row_max, "max")                                                                     # partial_fc.py:142
    _C.call("fedfr_exp_rowsum_target", cosine.data_ptr(), label.data_ptr(), R, Cc, Cc, row_max.data_ptr(), sums2.data_ptr(), st)
    all_reduce(sums2, "sum")                                                                       # partial_fc.py:147 + :161
    _C.call("fedfr_softmax_grad", cosine.data_ptr(), label.data_ptr(), R, Cc, Cc, sums2.data_ptr(), dmul.data_ptr(), s,
            inv_batch, prob_t.data_ptr(), st)
    _C.call("fedfr_nll_mean_ratio", sums2[1].data_ptr(), sums2[0].data_ptr(), R, floor, loss.data_ptr(), st)
    return loss, cosine


def sum_slabs(slabs: torch.Tensor) -> torch.Tensor:
    """Σ_s slabs[s] over the split-K slabs [S, ...] of ``sgemm(..., splits=S)``, ascending s (fedfr_fedavg_multi: ≤ 8 tensors per pass)."""
    import ctypes as C
    slabs = _chk(slabs, "slabs")
    S = slabs.shape[0]
    out = torch.empty_like(slabs[0])
    for s0 in range(0, S, 8):
        k = min(8, S - s0)
        ptrs = (C.c_void_p * k)(*[slabs[s0 + i].data_ptr() for i in range(k)])
        wv = (C.c_float * k)(*([1.0] * k))
        _C.call("fedfr_fedavg_multi", out.data_ptr(), ptrs, wv, k, out.numel(), 1 if s0 else 0, _C.stream())
    return out


def scale(x: torch.Tensor, w: float) -> torch.Tensor:
    """w * x (fp32) as a HIP kernel."""
    x = _chk(x, "x")
    out = torch.empty_like(x)
    _C.call("fedfr_fedavg_axpy", out.data_ptr(), x.data_ptr(), float(w), x.numel(), 0, _C.stream())
    return out


def axpy_(dst: torch.Tensor, src: torch.Tensor, w: float) -> torch.Tensor:
    """dst += w * src in place (fp32, HIP kernel)."""
    dst, src = _chk(dst, "dst"), _chk(src, "src")
    if dst.numel() != src.numel():
        raise RuntimeError("axpy_: size mismatch")
    _C.call("fedfr_fedavg_axpy", dst.data_ptr(), src.data_ptr(), float(w), dst.numel(), 1, _C.stream())
    return dst


def nll_mean(prob_t: torch.Tensor, floor: float = 0.0) -> torch.Tensor:
    loss = torch.empty((), dtype=f32, device=prob_t.device)
    _C.call("fedfr_nll_mean", prob_t.data_ptr(), prob_t.numel(), floor, loss.data_ptr(), _C.stream())
    return loss


def colsum(x: torch.Tensor) -> torch.Tensor:
    x = _chk(x, "x")
    out = torch.empty(x.shape[1], dtype=f32, device=x.device)
    _C.call("fedfr_colsum_f32", x.data_ptr(), x.shape[0], x.shape[1], out.data_ptr(), _C.stream())
    return out


# ------------------------------------------------------------------------------------------ autograd bridges
class CosineLinearFn(torch.autograd.Function):
    """normalize(x) @ normalize(w).T  (reference FC_module.forward, client.py:69-74)."""

    @staticmethod
    def forward(ctx, x, w, normalize_feat=True):
        x, w = _chk(x.detach(), "features"), _chk(w.detach(), "fc")
        if normalize_feat:
            xn, xinv = normalize_rows(x)
        else:
            xn, xinv = x, None
        wn, winv = normalize_rows(w)
        ctx.save_for_backward(xn, wn, winv, xinv if xinv is not None else torch.empty(0, device=x.device))
        ctx.normalize_feat = normalize_feat
        return sgemm(xn, wn, trans_b=True)

    @staticmethod
    def backward(ctx, dcos):
        xn, wn, winv, xinv = ctx.saved_tensors
        dcos = _chk(dcos, "dcos")
        dxn = sgemm(dcos, wn)                       # [B,C] @ [C,D]
        dwn = sgemm(dcos, xn, trans_a=True)         # [C,B] @ [B,D]
        dx = normalize_rows_bwd(xn, xinv, dxn) if ctx.normalize_feat else dxn
        dw = normalize_rows_bwd(wn, winv, dwn)
        return dx, dw, None


class MarginFn(torch.autograd.Function):
    """CosFace / ArcFace on a cosine matrix (losses.py:23-29, :38-45); rows with label == -1 get no margin."""

    @staticmethod
    def forward(ctx, cosine, label, s, m, arcface):
        z = _chk(cosine.detach().clone(), "cosine")
        label = _chk(label, "label", torch.int64)
        R, Cc = z.shape
        row_max = torch.empty(R, dtype=f32, device=z.device)
        dmul = torch.empty(R, dtype=f32, device=z.device)
        _C.call("fedfr_margin_rowmax", z.data_ptr(), label.data_ptr(), R, Cc, Cc, s, m, 1 if arcface else 0, row_max.data_ptr(),
                dmul.data_ptr(), _C.stream())
        ctx.save_for_backward(label, dmul)
        ctx.s = s
        return z

    @staticmethod
    def backward(ctx, dz):
        label, dmul = ctx.saved_tensors
        dz = _chk(dz, "dlogits")
        out = torch.empty_like(dz)
        _C.call("fedfr_margin_bwd", dz.data_ptr(), label.data_ptr(), dmul.data_ptr(), ctx.s, dz.shape[0], dz.shape[1],
                out.data_ptr(), _C.stream())
        return out, None, None, None, None


class CrossEntropyFn(torch.autograd.Function):
    """mean softmax cross-entropy (F.cross_entropy as used in client.py:545) with the gradient from the fused kernels."""

    @staticmethod
    def forward(ctx, logits, label):
        z = _chk(logits.detach().clone(), "logits")
        label = _chk(label, "label", torch.int64)
        prob_t, grad = softmax_ce_grad(z, label, 1.0, 0.0, False, 1.0 / z.shape[0])
        ctx.save_for_backward(grad)
        return nll_mean(prob_t, 0.0)

    @staticmethod
    def backward(ctx, dl):
        (grad,) = ctx.saved_tensors
        return grad * dl, None


class ContrastiveFn(torch.autograd.Function):
    """model-contrastive term of FedFR's local objective (reference client.py:372-375, :415-418):
    CE([cos(f, f_global)/T, cos(f, f_last)/T], 0) averaged over the batch; only ``feats`` gets a gradient."""

    @staticmethod
    def forward(ctx, feats, global_feats, last_feats, temperature):
        x = _chk(feats.detach(), "feats")
        g, l = _chk(global_feats.detach(), "global_feats"), _chk(last_feats.detach(), "last_feats")
        if g.shape != x.shape or l.shape != x.shape:
            raise RuntimeError("contrastive: feats / global_feats / last_feats must have the same [B, D] shape")
        B, D = x.shape
        row_loss = torch.empty(B, dtype=f32, device=x.device)
        dx = torch.empty_like(x)
        _C.call("fedfr_contrastive", x.data_ptr(), g.data_ptr(), l.data_ptr(), B, D, float(temperature), row_loss.data_ptr(),
                dx.data_ptr(), _C.stream())
        loss = torch.empty((), dtype=f32, device=x.device)
        _C.call("fedfr_sum_scale", row_loss.data_ptr(), B, 1.0 / B, loss.data_ptr(), _C.stream())
        ctx.save_for_backward(dx)
        return loss

    @staticmethod
    def backward(ctx, dl):
        (dx,) = ctx.saved_tensors
        return dx * dl, None, None, None


def contrastive_loss(feats, global_feats, last_feats, temperature=0.5):
    return ContrastiveFn.apply(feats, global_feats, last_feats, temperature)


@torch.no_grad()
def preprocess_u8(images_hwc: torch.Tensor, flip: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uint8 [B, H, W, 3] (decoded images) -> fp32 [B, 3, H, W] in [-1, 1], optionally mirrored per image: the reference's
    train / test transform (dataset.py:81-92) on the device, bit-exact with torchvision's ToTensor + Normalize(0.5, 0.5)."""
    x = _chk(images_hwc, "images", torch.uint8)
    if x.dim() != 4 or x.shape[3] != 3:
        raise RuntimeError("preprocess_u8: expected uint8 [B, H, W, 3]")
    B, H, W, _ = x.shape
    f = None
    if flip is not None:
        f = _chk(flip.to(torch.uint8), "flip", torch.uint8)
        if f.shape != (B,):
            raise RuntimeError("preprocess_u8: flip must be [B]")
    out = torch.empty(B, 3, H, W, dtype=f32, device=x.device)
    _C.call("fedfr_preprocess_u8", x.data_ptr(), f.data_ptr() if f is not None else None, out.data_ptr(), B, H, W, _C.stream())
    return out


@torch.no_grad()
def class_accumulate(feats: torch.Tensor, label: torch.Tensor, sums: torch.Tensor, counts: torch.Tensor) -> None:
    """sums[c] += sum of feats rows labelled c, counts[c] += their number (reference client.py:171-178, server.py:213-222)."""
    feats, label = _chk(feats, "feats"), _chk(label, "label", torch.int64)
    sums, counts = _chk(sums, "sums"), _chk(counts, "counts")
    if sums.shape != (counts.shape[0], feats.shape[1]) or label.shape[0] != feats.shape[0]:
        raise RuntimeError("class_accumulate: shape mismatch")
    _C.call("fedfr_class_accumulate", feats.data_ptr(), label.data_ptr(), feats.shape[0], feats.shape[1], sums.shape[0],
            sums.data_ptr(), counts.data_ptr(), _C.stream())


@torch.no_grad()
def similarity_column_flags(a: torch.Tensor, b: torch.Tensor, threshold: float) -> torch.Tensor:
    """uint8 [N]: 1 where some row m has (a @ b.T)[m, n] > threshold — `torch.where(a @ b.T > thr)[1]` as a set, without the
    [M, N] matrix (reference client.py:208-226).  a [M, K], b [N, K], fp32, exact fp32 MFMA."""
    a, b = _chk(a, "a"), _chk(b, "b")
    if a.shape[1] != b.shape[1]:
        raise RuntimeError("similarity_column_flags: inner dimensions differ")
    flags = torch.zeros(b.shape[0], dtype=torch.uint8, device=a.device)
    _C.call("fedfr_sgemm_colflag", a.data_ptr(), b.data_ptr(), a.shape[0], b.shape[0], a.shape[1], a.stride(0), 1, 1, b.stride(0),
            1.0, float(threshold), flags.data_ptr(), _C.stream())
    return flags


def cosine_linear(x, w, normalize_feat=True):
    return CosineLinearFn.apply(x, w, normalize_feat)


def cross_entropy(logits, label):
    return CrossEntropyFn.apply(logits, label)
