"""SphereFace backbone (reference backbones/sphnet.py:4-73 — the network `run.sh` trains) on the MI355X kernels.

Same constructor / factory / state_dict keys as the reference (`sphere(type=20|64)`, `sphnet(pretrained, dropout, fp16, type)`;
keys `layerL.0.weight|bias`, `layerL.1.weight`, `layerL.K.conv1|prelu1|conv2|prelu2.weight`, `fc.weight|bias`).  There is no
normalisation layer: every unit is conv3x3 (+bias on the stride-2 stage heads) -> PReLU, residual blocks add their input.

Unlike iresnet (one C++ plan, `csrc/net.hip`) this second model is sequenced from Python over the per-op C ABI — the same
`fedfr_conv2d_fwd / _dgrad / _wgrad` MFMA kernels, `fedfr_bn_apply` with scale 1 / shift = bias as the fused bias + PReLU
(+ residual) pass, `fedfr_bias_prelu_bwd`, and bf16 GEMMs for the fc — so it costs a few hundred ctypes calls per step.
Activations are NHWC bf16, weights are stored KRSC (the OIHW parameters are channels_last views of that storage), accumulation
is fp32; the 3-channel input is zero-padded to 64 channels for the first conv (`fedfr_pad_input_nhwc`)."""
from __future__ import annotations

import torch
from torch import nn

from .. import _C

f32, bf16 = torch.float32, torch.bfloat16


def _conv_param(cout, cin, std=0.01):
    """OIHW-shaped parameter whose storage is KRSC ([Cout][kh][kw][Cin]): what the conv kernels read."""
    w = torch.empty(cout, 3, 3, cin).normal_(0, std)
    return nn.Parameter(w.permute(0, 3, 1, 2))


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("fedfr_amd: sphnet sub-modules are parameter holders; call the network itself")


class _Conv(_Holder):
    def __init__(self, cin, cout, bias):
        super().__init__()
        self.weight = _conv_param(cout, cin)
        if bias:                                                    # sphnet.py:44-47: xavier_uniform weight, zero bias
            nn.init.xavier_uniform_(self.weight)
            self.bias = nn.Parameter(torch.zeros(cout))


class _PReLU(_Holder):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.full((c,), 0.25))


class Block(_Holder):
    """reference sphnet.py:4-13"""

    def __init__(self, planes):
        super().__init__()
        self.conv1, self.prelu1 = _Conv(planes, planes, False), _PReLU(planes)
        self.conv2, self.prelu2 = _Conv(planes, planes, False), _PReLU(planes)


class _Linear(_Holder):
    def __init__(self, i, o):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i))
        self.bias = nn.Parameter(torch.zeros(o))
        nn.init.xavier_uniform_(self.weight)


def _p(t):
    return t.data_ptr() if t is not None else None


class _SphereFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, net):
        feats, saved = net._forward_impl(x, keep=True)
        ctx.net, ctx.saved = net, saved
        return feats

    @staticmethod
    def backward(ctx, dfeats):
        ctx.net._backward_impl(ctx.saved, dfeats.contiguous())
        return None, None, None


class sphere(nn.Module):
    def __init__(self, type=20, is_gray=False, fp16=False):
        super().__init__()
        self.fp16 = fp16
        if type == 20:
            layers = [1, 2, 4, 1]
        elif type == 64:
            layers = [3, 7, 16, 3]
        else:
            raise ValueError("sphere" + str(type) + " IS NOT SUPPORTED! (sphere20 or sphere64)")
        if is_gray:
            raise NotImplementedError("fedfr_amd sphnet: is_gray is not built")
        filt = [3, 64, 128, 256, 512]
        self.layers_cfg, self.filters = layers, filt
        for L in range(4):
            mods = [_Conv(filt[L], filt[L + 1], True), _PReLU(filt[L + 1])] + [Block(filt[L + 1]) for _ in range(layers[L])]
            setattr(self, "layer%d" % (L + 1), nn.Sequential(*mods))
        self.fc = _Linear(512 * 7 * 7, 512)
        self.num_features = 512
        self._anchor = torch.zeros(1, requires_grad=True)
        self._cache = {}

    # ------------------------------------------------------------------ helpers
    @property
    def device(self):
        return self.fc.weight.device

    def _stage(self, L):
        seq = getattr(self, "layer%d" % (L + 1))
        return seq[0], seq[1], list(seq)[2:]

    def _shadows(self, conv: _Conv, cin_pad=None):
        """(fwd bf16 [Cout][3][3][Cin'], dgrad bf16 [Cin'][3][3][Cout] flipped) of a conv's current weights."""
        w = conv.weight.data
        cout, cin = w.shape[0], w.shape[1]
        wk = w.permute(0, 2, 3, 1)                                  # KRSC view of the parameter's own storage
        if not wk.is_contiguous():
            raise RuntimeError("fedfr_amd sphnet: conv weights must keep their KRSC (channels_last) storage")
        if cin_pad and cin_pad != cin:
            pad = torch.zeros(cout, 3, 3, cin_pad, dtype=f32, device=w.device)
            pad[..., :cin] = wk
            wk, cin = pad, cin_pad
        wb = torch.empty(cout, 3, 3, cin, dtype=bf16, device=w.device)
        wd = torch.empty(cin, 3, 3, cout, dtype=bf16, device=w.device)
        _C.call("fedfr_weight_shadows", wk.data_ptr(), wb.data_ptr(), wd.data_ptr(), cout, 3, cin, _C.stream())
        return wb, wd

    def _ones_zeros(self, c):
        key = ("oz", c, str(self.device))
        if key not in self._cache:
            self._cache[key] = (torch.ones(c, device=self.device), torch.zeros(c, device=self.device))
        return self._cache[key]

    def _act(self, x1, c, M, shift, alpha, x2=None, nchw_hw=0):
        """y = prelu(x1 + shift) (+ x2): fedfr_bn_apply with unit scales."""
        ones, zeros = self._ones_zeros(c)
        y = torch.empty_like(x1)
        _C.call("fedfr_bn_apply", x1.data_ptr(), ones.data_ptr(), _p(shift) if shift is not None else zeros.data_ptr(), _p(alpha),
                _p(x2), ones.data_ptr() if x2 is not None else None, zeros.data_ptr() if x2 is not None else None, y.data_ptr(), M, c,
                nchw_hw, None, _C.stream())
        return y

    def _conv_fwd(self, x, wb, B, hin, cin, cout, stride):
        ho = hin // stride
        y = torch.empty(B, ho, ho, cout, dtype=bf16, device=x.device)
        _C.call("fedfr_conv2d_fwd", x.data_ptr(), wb.data_ptr(), y.data_ptr(), None, B, hin, cin, cout, 3, stride, _C.stream())
        return y

    # ------------------------------------------------------------------ forward / backward
    def _forward_impl(self, x, keep):
        x = _C.require_gpu_tensor(x.contiguous(), f32, "input")
        if x.dim() != 4 or x.shape[1:] != (3, 112, 112):
            raise RuntimeError("fedfr_amd sphnet: expected input [B, 3, 112, 112]")
        B, st = x.shape[0], _C.stream()
        xin = torch.empty(B, 112, 112, 64, dtype=bf16, device=x.device)
        _C.call("fedfr_pad_input_nhwc", x.data_ptr(), xin.data_ptr(), B, 3, 112 * 112, 64, st)
        saved = {"B": B, "stages": []}
        a, hin, cin = xin, 112, 64
        for L in range(4):
            head, pre, blocks = self._stage(L)
            c, h = self.filters[L + 1], hin // 2
            M = B * h * h
            wb, wd = self._shadows(head, cin_pad=64 if L == 0 else None)
            c0 = self._conv_fwd(a, wb, B, hin, cin, c, 2)
            last = L == 3 and not blocks
            a0 = self._act(c0, c, M, head.bias.data, pre.weight.data)
            rec = {"x": a, "c0": c0, "wd": wd, "blocks": [], "hin": hin, "cin": cin, "c": c, "h": h}
            a = a0
            for bi, blk in enumerate(blocks):
                w1b, w1d = self._shadows(blk.conv1)
                w2b, w2d = self._shadows(blk.conv2)
                c1 = self._conv_fwd(a, w1b, B, h, c, c, 1)
                t1 = self._act(c1, c, M, None, blk.prelu1.weight.data)
                c2 = self._conv_fwd(t1, w2b, B, h, c, c, 1)
                flat = L == 3 and bi == len(blocks) - 1                      # last activation leaves NCHW-flat for x.view(B, -1)
                a_new = self._act(c2, c, M, None, blk.prelu2.weight.data, x2=a, nchw_hw=h * h if flat else 0)
                rec["blocks"].append({"a": a, "c1": c1, "t1": t1, "c2": c2, "w1d": w1d, "w2d": w2d})
                a = a_new
            saved["stages"].append(rec)
            hin, cin = h, c
        flat = a.view(B, 512 * 7 * 7)                                            # NCHW-flat bf16 (see `flat` above)
        wfc = self.fc.weight.data.to(bf16)
        feats = torch.empty(B, 512, dtype=f32, device=x.device)
        nb = 128 * B * 512 * 4 + 64
        ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
        _C.call("fedfr_gemm_nt", flat.data_ptr(), wfc.data_ptr(), feats.data_ptr(), ws.data_ptr(), nb, B, 512, 512 * 7 * 7, st)
        ones_col = torch.ones(B, 1, dtype=f32, device=x.device)
        bias_row = self.fc.bias.data.view(1, 512)
        _C.call("fedfr_sgemm", ones_col.data_ptr(), bias_row.data_ptr(), feats.data_ptr(), B, 512, 1, 1, 1, 512, 1, 512, 1.0, 1.0, None, st)
        if keep:
            saved["flat"], saved["wfc"] = flat, wfc
            return feats, saved
        return feats, None

    def _grad_view(self, conv: _Conv, cin_store=None):
        """fp32 KRSC gradient buffer of a conv + the OIHW view assigned to .grad"""
        cout, cin = conv.weight.shape[0], conv.weight.shape[1]
        g = torch.empty(cout, 3, 3, cin_store or cin, dtype=f32, device=self.device)
        return g

    def _wgrad(self, x, dy, conv, B, hin, cin, cout, stride):
        g = torch.empty(cout, 3, 3, cin, dtype=f32, device=x.device)
        nb = _C.lib().fedfr_conv2d_wgrad_ws_bytes(B, hin, cin, cout, 3, stride)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=x.device)
        _C.call("fedfr_conv2d_wgrad", x.data_ptr(), dy.data_ptr(), g.data_ptr(), ws.data_ptr(), nb, B, hin, cin, cout, 3, stride, _C.stream())
        real_cin = conv.weight.shape[1]
        if real_cin != cin:
            g = g[..., :real_cin].contiguous()
        conv.weight.grad = g.permute(0, 3, 1, 2)

    def _dgrad(self, dy, wd, B, hin, cin, cout, stride):
        dx = torch.empty(B, hin, hin, cin, dtype=bf16, device=dy.device)
        _C.call("fedfr_conv2d_dgrad", dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), B, hin, cin, cout, 3, stride, _C.stream())
        return dx

    def _prelu_bwd(self, dy, x, bias, prelu: _PReLU, M, c, add=None, bias_mod=None):
        rows = _C.lib().fedfr_bn_bwd_rows(M, c)
        part = torch.empty(rows, 3, c, dtype=f32, device=dy.device)
        coef = torch.empty(3, c, dtype=f32, device=dy.device)
        dalpha = torch.empty(c, dtype=f32, device=dy.device)
        dbias = torch.empty(c, dtype=f32, device=dy.device) if bias is not None else None
        dx = torch.empty_like(dy)
        _C.call("fedfr_bias_prelu_bwd", dy.data_ptr(), x.data_ptr(), _p(bias), prelu.weight.data.data_ptr(), M, c, part.data_ptr(),
                coef.data_ptr(), _p(dbias), dalpha.data_ptr(), _p(add), dx.data_ptr(), _C.stream())
        prelu.weight.grad = dalpha
        if bias_mod is not None:
            bias_mod.bias.grad = dbias
        return dx

    def _backward_impl(self, saved, dfeats):
        B, st = saved["B"], _C.stream()
        dfeats = _C.require_gpu_tensor(dfeats, f32, "dfeats")
        dyb = dfeats.to(bf16)
        # fc: dW = dy^T x, db = colsum(dy), dx = dy W
        K = 512 * 7 * 7
        dw = torch.empty(512, K, dtype=f32, device=dfeats.device)
        _C.call("fedfr_gemm_tn", dyb.data_ptr(), saved["flat"].data_ptr(), dw.data_ptr(), B, 512, K, st)
        self.fc.weight.grad = dw
        db = torch.empty(512, dtype=f32, device=dfeats.device)
        _C.call("fedfr_colsum_f32", dfeats.data_ptr(), B, 512, db.data_ptr(), st)
        self.fc.bias.grad = db
        dflat = torch.empty(B, K, dtype=f32, device=dfeats.device)
        dyt = dyb.t().contiguous()
        _C.call("fedfr_gemm_tn", dyt.data_ptr(), saved["wfc"].data_ptr(), dflat.data_ptr(), 512, B, K, st)
        # the flattened activation was NCHW-flat: back to NHWC bf16
        g = dflat.view(B, 512, 7, 7).permute(0, 2, 3, 1).contiguous().to(bf16)
        for L in (3, 2, 1, 0):
            head, pre, blocks = self._stage(L)
            rec = saved["stages"][L]
            c, h, M = rec["c"], rec["h"], B * rec["h"] * rec["h"]
            for bi in range(len(blocks) - 1, -1, -1):
                blk, r = blocks[bi], rec["blocks"][bi]
                dz2 = self._prelu_bwd(g, r["c2"], None, blk.prelu2, M, c)
                self._wgrad(r["t1"], dz2, blk.conv2, B, h, c, c, 1)
                dt1 = self._dgrad(dz2, r["w2d"], B, h, c, c, 1)
                dz1 = self._prelu_bwd(dt1, r["c1"], None, blk.prelu1, M, c)
                self._wgrad(r["a"], dz1, blk.conv1, B, h, c, c, 1)
                da = self._dgrad(dz1, r["w1d"], B, h, c, c, 1)
                g = self._act(da, c, M, None, None, x2=g)                        # identity path + branch
            dz0 = self._prelu_bwd(g, rec["c0"], head.bias.data, pre, M, c, bias_mod=head)
            self._wgrad(rec["x"], dz0, head, B, rec["hin"], rec["cin"], c, 2)
            if L > 0:
                g = self._dgrad(dz0, rec["wd"], B, rec["hin"], rec["cin"], c, 2)

    def forward(self, x):
        if self.training and torch.is_grad_enabled():
            if self._anchor.device != x.device:
                self._anchor = torch.zeros(1, device=x.device, requires_grad=True)
            return _SphereFn.apply(x, self._anchor, self)
        return self._forward_impl(x, keep=False)[0]

    def save(self, file_path):
        with open(file_path, "wb") as f:
            torch.save(self.state_dict(), f)


def sphnet(pretrained=False, dropout=None, fp16=False, type=64):
    """reference sphnet.py:72-73"""
    return sphere(type, fp16=fp16)
