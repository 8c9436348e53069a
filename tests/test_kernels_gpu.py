"""Per-kernel parity of the HIP library (through the C ABI) against plain fp32 PyTorch CPU references
of the same op on identical seeded inputs.  bf16 kernels are fed bf16-rounded inputs so the only
differences are accumulation order and the final bf16 rounding of the output (tolerances state that)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from fedfr_amd import _C, ops  # noqa: E402


def dev():
    return torch.device("cuda:0")


def S16():
    """the loaded library's 16-bit storage type: float16 (the product library) or bfloat16 (FEDFR_HIP_LIB_NAME=libfedfr_hip_bf16.so)"""
    return _C.storage_dtype()


def bf(t):
    return t.to(S16())


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def nhwc(t):  # NCHW -> NHWC contiguous
    return t.permute(0, 2, 3, 1).contiguous()


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


CONV_CASES = [
    # B, H, Cin, Cout, k, s
    (2, 14, 64, 64, 3, 1),
    (2, 16, 64, 128, 3, 2),
    (3, 14, 128, 256, 1, 2),
    (2, 28, 128, 128, 3, 1),
    (1, 14, 256, 256, 3, 1),     # M = 196: ragged M tile
    (5, 8, 512, 512, 3, 2),
    (40, 14, 256, 256, 3, 1),    # M = 7840 -> 128-row tiles with tail
    # large-M 3x3/s1 shapes take the LDS-halo kernel (fwd + dgrad): every halo size, BN = 64 / 128, ragged M
    (131, 14, 256, 256, 3, 1),
    (17, 56, 64, 64, 3, 1),
    (4, 112, 64, 128, 3, 1),
    (9, 112, 64, 64, 3, 1),
    (65, 28, 128, 128, 3, 1),
    (260, 14, 64, 64, 3, 1),     # zero-padded-image kernel, BN = 64 instantiation
    (131, 28, 128, 256, 3, 1),
    (9, 56, 64, 128, 3, 1),      # LDS-DMA kernel, single input chunk -> 128-wide tile (fwd) / two chunks -> 64-wide tile (dgrad)
    (3, 56, 128, 64, 3, 1),
    # 64 -> 64 channel layers: persistent kernel with the filter bank in registers (conv_c64p.hip): 1, 2 and 5 tiles per workgroup
    (1, 56, 64, 64, 3, 1),
    (20, 112, 64, 64, 3, 1),
    (37, 56, 64, 64, 3, 1),
]


NT_GLDS_DEFAULT = 4        # the library's default for option nt_glds


def _conv_inputs(B, H, Cin, Cout, k, s, seed=0):
    x = bf(rnd((B, Cin, H, H), seed + 1)).float()
    w = bf(rnd((Cout, Cin, k, k), seed + 2, 0.1)).float()
    return x, w


@pytest.mark.parametrize("B,H,Cin,Cout,k,s", CONV_CASES)
def test_conv_fwd_and_stats(B, H, Cin, Cout, k, s):
    x, w = _conv_inputs(B, H, Cin, Cout, k, s)
    ref = F.conv2d(x, w, None, s, 1 if k == 3 else 0)
    Ho = H // s
    xd = bf(nhwc(x)).to(dev())
    wd = bf(w.permute(0, 2, 3, 1).contiguous()).to(dev())          # KRSC
    y = torch.empty(B, Ho, Ho, Cout, dtype=S16(), device=dev())
    rows = _C.lib().fedfr_conv2d_stat_rows(B, Ho, Cout)
    stats = torch.full((rows, 2, Cout), float("nan"), device=dev())
    _C.call("fedfr_conv2d_fwd", xd.data_ptr(), wd.data_ptr(), y.data_ptr(), stats.data_ptr(), B, H, Cin, Cout, k, s, _C.stream())
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    # fp32 accumulate, bf16-rounded output: |err| <= 2^-8 |y| (+ tiny accumulation-order noise)
    assert relerr(got, ref) < 6e-3
    yr = y.float().cpu().reshape(-1, Cout)
    st = stats.cpu().double().sum(0)
    np.testing.assert_allclose(st[0].numpy(), yr.double().sum(0).numpy(), rtol=1e-4, atol=1e-2)
    np.testing.assert_allclose(st[1].numpy(), (yr.double() ** 2).sum(0).numpy(), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("B,H,Cin,Cout,k,s", CONV_CASES)
def test_conv_dgrad(B, H, Cin, Cout, k, s):
    x, w = _conv_inputs(B, H, Cin, Cout, k, s)
    Ho = H // s
    dy = bf(rnd((B, Cout, Ho, Ho), 7)).float()
    x.requires_grad_(True)
    F.conv2d(x, w, None, s, 1 if k == 3 else 0).backward(dy)
    ref = x.grad
    wk = w.permute(0, 2, 3, 1).contiguous().to(dev())            # KRSC fp32
    wb = torch.empty(wk.shape, dtype=S16(), device=dev())
    wdb = torch.empty(Cin, k, k, Cout, dtype=S16(), device=dev())
    _C.call("fedfr_weight_shadows", wk.data_ptr(), wb.data_ptr(), wdb.data_ptr(), Cout, k, Cin, _C.stream())
    dyd = bf(nhwc(dy)).to(dev())
    if k == 1:
        dx = torch.empty(B, Ho, Ho, Cin, dtype=S16(), device=dev())
    else:
        dx = torch.empty(B, H, H, Cin, dtype=S16(), device=dev())
    _C.call("fedfr_conv2d_dgrad", dyd.data_ptr(), wdb.data_ptr(), dx.data_ptr(), B, H, Cin, Cout, k, s, _C.stream())
    torch.cuda.synchronize()
    got = dx.float().cpu().permute(0, 3, 1, 2)
    if k == 1:   # compact result lives at the even positions of the input grid
        full = torch.zeros_like(ref)
        full[:, :, ::s, ::s] = got
        got = full
    assert relerr(got, ref) < 6e-3
    assert torch.equal(wb.float().cpu(), w.permute(0, 2, 3, 1))   # shadow cast is exact for bf16-valued weights


@pytest.mark.parametrize("B,H,Cin,Cout,prelu", [(131, 14, 256, 256, True), (65, 28, 128, 128, False), (2, 14, 64, 64, True),
                                                (5, 112, 64, 64, False), (5, 112, 64, 64, True), (19, 56, 64, 64, False), (19, 56, 64, 64, True),
                                                (3, 56, 64, 64, True), (64, 28, 128, 128, True), (33, 28, 256, 128, False),
                                                (131, 14, 256, 256, "signs"), (64, 28, 128, 128, "signs"), (19, 56, 64, 64, "signs")])
def test_conv_dgrad_fused_bn_bwd_reduction(B, H, Cin, Cout, prelu):
    """dgrad epilogue also reduces (sum dz, sum dz*xhat, sum dx*min(z,0)) of the BN that precedes the conv.  The 64 -> 64 layers of the
    56x56 / 112x112 maps run on the persistent kernel (conv_c64p.hip): one partial row per workgroup, 1 or 2 tiles each here."""
    x, w = _conv_inputs(B, H, Cin, Cout, 3, 1)
    dy = bf(rnd((B, Cout, H, H), 7))
    d = dev()
    wk = w.permute(0, 2, 3, 1).contiguous().to(d)
    wdb = torch.empty(Cin, 3, 3, Cout, dtype=S16(), device=d)
    _C.call("fedfr_weight_shadows", wk.data_ptr(), None, wdb.data_ptr(), Cout, 3, Cin, _C.stream())
    dyd = nhwc(dy).to(d)
    bnx = bf(rnd((B * H * H, Cin), 9) * 1.5 + 0.2).to(d)
    mean, rstd = (rnd((Cin,), 10) * 0.2).to(d), (rnd((Cin,), 11) * 0.2 + 1.0).to(d)
    gamma, beta, alpha = (rnd((Cin,), 12) * 0.2 + 1).to(d), (rnd((Cin,), 13) * 0.3).to(d), (rnd((Cin,), 14) * 0.1 + 0.25).to(d)
    if prelu == "signs":
        # (round 5: the matrix-core epilogues take the PReLU mask from a per-channel THRESHOLD on x — z <= 0 <=> sgn(sc) x <= -sh / |sc| — so the
        # sign of gamma, gamma == 0 (z = beta everywhere: always / never masked), beta == 0 and x sitting exactly ON a representable threshold
        # are cases of their own)
        gamma = gamma.clone(); beta = beta.clone()
        gamma[::3] *= -1.0
        gamma[5::16] = 0.0
        beta[7::16] = 0.0
        beta[5::32] = -0.25
        gamma[9::16] = 1.0; beta[9::16] = -0.5; mean[9::16] = 0.0; rstd[9::16] = 1.0       # threshold x = 0.5 exactly ...
        bnx[::7, 9::16] = 0.5                                                                 # ... and x on it: z == 0 counts as masked
    dx = torch.empty(B, H, H, Cin, dtype=S16(), device=d)
    part = torch.full((((B * H * H + 127) // 128), 3, Cin), float("nan"), device=d)
    rows = C.c_int(0)
    _C.call("fedfr_conv2d_dgrad_bnbwd", dyd.data_ptr(), wdb.data_ptr(), dx.data_ptr(), B, H, Cin, Cout, 3, 1, bnx.data_ptr(),
            mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), alpha.data_ptr() if prelu else None,
            part.data_ptr(), C.byref(rows), _C.stream())
    torch.cuda.synchronize()
    # plain dgrad result is unchanged by the fusion
    dx_ref = torch.empty_like(dx)
    _C.call("fedfr_conv2d_dgrad", dyd.data_ptr(), wdb.data_ptr(), dx_ref.data_ptr(), B, H, Cin, Cout, 3, 1, _C.stream())
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_ref)
    if Cin == 64 and H in (56, 112):        # persistent kernel: 224-pixel tiles dealt out to at most one workgroup per CU
        ntiles = B * H * H // 224
        cus = torch.cuda.get_device_properties(d).multi_processor_count
        per_wg = -(-ntiles // min(ntiles, cus))
        assert rows.value == -(-ntiles // per_wg)
    elif B * H * H < 384 * 128 // 2:        # small problems take the generic kernel: no fusion, caller reduces itself
        assert rows.value == 0
        return
    else:
        # partial rows = M tiles of the kernel that ran: 196-pixel image tiles (LDS-DMA kernel) or 128-row tiles (halo2 kernel)
        # ... or one row per PAIR of 196-pixel tiles (the two-tiles 28x28 kernel, when the tile count is even)
        assert rows.value in ((B * H * H + 127) // 128, B * H * H // 196) or (H == 28 and (B * 4) % 2 == 0 and rows.value == B * 2)
    g = dx.float().reshape(-1, Cin).double()
    xh = (bnx.float().double() - mean.double()) * rstd.double()
    dz = g.clone()
    s3 = torch.zeros(Cin, dtype=torch.float64, device=d)
    if prelu:
        z = gamma.double() * xh + beta.double()
        neg = z <= 0
        s3 = (g * z * neg).sum(0)
        dz = torch.where(neg, g * alpha.double(), g)
    got = part[: rows.value].double().sum(0)
    for i, ref in enumerate((dz.sum(0), (dz * xh).sum(0), s3)):
        scale = float(ref.abs().max()) + 1e-6
        assert float((got[i] - ref).abs().max()) < 2e-4 * scale + 1e-2, i


@pytest.mark.parametrize("B,H,Ch", [(128, 14, 256), (64, 28, 128), (200, 14, 256), (40, 14, 256)])
def test_forward_moment_conv_and_bn_apply2(B, H, Ch):
    """Round 3 forward moment pass, kernel level.  (1) fedfr_conv2d_fwd_moments: same output as the plain conv, rows of raw moments
    (sum y, sum y * other, sum y * y) that add up to torch's.  (2) fedfr_bn_apply2_sliced on those rows: out = bn(y) + other and
    y2 = bn_next(out) with the statistics of `out` DERIVED from the moments — against torch BatchNorms (training mode) that MEASURE them:
    saved mean / rstd, running statistics, both outputs.  Shapes: one row per 14x14 image tile, one per pair of 28x28 band tiles, the
    double fan-in variant (200 rows), and a problem too small for a kernel with that epilogue (rows == 0, plain output)."""
    d = dev()
    M = B * H * H
    x, w = _conv_inputs(B, H, Ch, Ch, 3, 1)
    wb = bf(w.permute(0, 2, 3, 1).contiguous()).to(d)                   # KRSC bf16
    xd = bf(nhwc(x)).to(d)
    other = bf(rnd((M, Ch), 21) * 1.3 + 0.25).to(d)
    y = torch.empty(B, H, H, Ch, dtype=S16(), device=d)
    part = torch.full((M // 196 + 1, 3, Ch), float("nan"), device=d)
    rows = C.c_int(0)
    _C.call("fedfr_conv2d_fwd_moments", xd.data_ptr(), wb.data_ptr(), y.data_ptr(), B, H, Ch, Ch, other.data_ptr(), part.data_ptr(),
            C.byref(rows), _C.stream())
    y_ref = torch.empty_like(y)
    _C.call("fedfr_conv2d_fwd", xd.data_ptr(), wb.data_ptr(), y_ref.data_ptr(), None, B, H, Ch, Ch, 3, 1, _C.stream())
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref)
    if -(-M // 128) * -(-Ch // 128) < 384:           # (gemm.hip: nt_bm) small problems take the generic kernel: no epilogue of this kind
        assert rows.value == 0
        return
    assert rows.value == (M // 196 if H == 14 else M // 392)
    yf, of = y.float().reshape(M, Ch).double(), other.float().double()
    got = part[: rows.value].double().sum(0)
    for i, ref in enumerate((yf.sum(0), (yf * of).sum(0), (yf * yf).sum(0))):
        assert float((got[i] - ref).abs().max()) < 2e-4 * (float(ref.abs().max()) + 1e-6) + 1e-2, i
    # ---- the pass on those rows
    assert _C.lib().fedfr_bn_apply2_sliced_ok(M, Ch, rows.value) == 1
    eps, mom = 1e-5, 0.1
    g3, b3 = (rnd((Ch,), 31) * 0.2 + 1).to(d), (rnd((Ch,), 32) * 0.3).to(d)
    g1, b1 = (rnd((Ch,), 33) * 0.2 + 1).to(d), (rnd((Ch,), 34) * 0.3).to(d)
    rm3, rv3, rm1, rv1 = [(rnd((Ch,), 35 + i) * 0.1 + (1.0 if i % 2 else 0.0)).to(d) for i in range(4)]
    rm3_0, rv3_0, rm1_0, rv1_0 = rm3.clone(), rv3.clone(), rm1.clone(), rv1.clone()
    xmean = of.mean(0).float()
    xrstd = (1.0 / torch.sqrt(of.var(0, unbiased=False) + eps)).float()
    sv = [torch.full((Ch,), float("nan"), device=d) for _ in range(8)]
    out = torch.empty(M, Ch, dtype=S16(), device=d)
    y2 = torch.empty(M, Ch, dtype=S16(), device=d)
    _C.call("fedfr_bn_apply2_sliced", part.data_ptr(), rows.value, float(M), mom, eps, g3.data_ptr(), b3.data_ptr(), rm3.data_ptr(), rv3.data_ptr(),
            sv[0].data_ptr(), sv[1].data_ptr(), sv[2].data_ptr(), sv[3].data_ptr(), xmean.data_ptr(), xrstd.data_ptr(), g1.data_ptr(), b1.data_ptr(),
            rm1.data_ptr(), rv1.data_ptr(), sv[4].data_ptr(), sv[5].data_ptr(), sv[6].data_ptr(), sv[7].data_ptr(), y.data_ptr(), other.data_ptr(),
            out.data_ptr(), y2.data_ptr(), M, Ch, _C.stream())
    torch.cuda.synchronize()
    my, vy = yf.mean(0), yf.var(0, unbiased=False)
    out_ref = (yf - my) / torch.sqrt(vy + eps) * g3.double() + b3.double() + of
    assert relerr(out.float(), out_ref.float()) < 4e-3                    # one bf16 rounding
    assert float((sv[2].double() - my).abs().max()) < 1e-5 * (1 + float(my.abs().max()))
    assert float((sv[3].double() * torch.sqrt(vy + eps) - 1).abs().max()) < 1e-5
    og = out.float().double()                                            # what a measuring bn1 pass would see
    mo, vo = og.mean(0), og.var(0, unbiased=False)
    # derived vs measured statistics of the STORED tensor: they differ by what its rounding to bf16 adds (per element 2^-9 relative, zero mean:
    # ~1e-4 of a mean over 2.5e4 elements at the worst channel, 7.3e-5 measured), nothing else
    assert float((sv[6].double() - mo).abs().max()) < 2e-4 * (1 + float(mo.abs().max()))
    assert float((sv[7].double() * torch.sqrt(vo + eps) - 1).abs().max()) < 2e-4
    y2_ref = (og - mo) / torch.sqrt(vo + eps) * g1.double() + b1.double()
    assert relerr(y2.float(), y2_ref.float()) < 4e-3
    k = M / (M - 1.0)
    for got_, ref_, tol in ((rm3, 0.9 * rm3_0.double() + 0.1 * my, 1e-5), (rv3, 0.9 * rv3_0.double() + 0.1 * vy * k, 1e-5),
                            (rm1, 0.9 * rm1_0.double() + 0.1 * mo, 3e-5), (rv1, 0.9 * rv1_0.double() + 0.1 * vo * k, 3e-5)):
        assert float((got_.double() - ref_).abs().max()) < tol * (1 + float(ref_.abs().max()))


@pytest.mark.parametrize("variant", ["default", "no_wgrad9", "scalar_frags"])
@pytest.mark.parametrize("B,H,Cin,Cout,k,s", CONV_CASES + [(128, 14, 256, 256, 3, 1), (19, 14, 256, 512, 3, 1), (33, 28, 128, 256, 3, 1)])
def test_conv_wgrad(B, H, Cin, Cout, k, s, variant):
    """default: the nine-tap kernel (wgrad9.hip) on 3x3/s1 14x14 and 28x28 layers, the LDS-DMA / register-staged TN GEMMs elsewhere;
    no_wgrad9: the TN GEMMs everywhere; scalar_frags: their validation fallback without transpose reads."""
    use_tr, w9 = (0 if variant == "scalar_frags" else 1), (0 if variant == "no_wgrad9" else 1)
    x, w = _conv_inputs(B, H, Cin, Cout, k, s)
    Ho = H // s
    dy = bf(rnd((B, Cout, Ho, Ho), 7)).float()
    w.requires_grad_(True)
    F.conv2d(x, w, None, s, 1 if k == 3 else 0).backward(dy)
    ref = w.grad.permute(0, 2, 3, 1)                                 # KRSC
    xd, dyd = bf(nhwc(x)).to(dev()), bf(nhwc(dy)).to(dev())
    dw = torch.full((Cout, k, k, Cin), float("nan"), device=dev())
    nbytes = _C.lib().fedfr_conv2d_wgrad_ws_bytes(B, H, Cin, Cout, k, s)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev())
    _C.call("fedfr_set_option", b"tn_use_tr", use_tr)
    _C.call("fedfr_set_option", b"wgrad9", w9)
    try:
        _C.call("fedfr_conv2d_wgrad", xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), ws.data_ptr(), nbytes, B, H, Cin, Cout, k, s,
                _C.stream())
        torch.cuda.synchronize()
    finally:
        _C.call("fedfr_set_option", b"tn_use_tr", 1)
        _C.call("fedfr_set_option", b"wgrad9", 1)
    # bf16 operands are exact, fp32 accumulation: only summation order differs
    assert relerr(dw, ref) < 2e-4


@pytest.mark.parametrize("B,H,C", [(128, 14, 256), (32, 28, 128), (24, 14, 128), (2, 14, 128)])
def test_conv_wgrad_pair(B, H, C):
    """fedfr_conv2d_wgrad_pair (the two same-shape 3x3 weight gradients of a residual block in one call) == autograd of F.conv2d on the same
    bf16 operands, whichever kernel serves the shape."""
    outs = []
    for seed in (11, 23):
        x = bf(rnd((B, C, H, H), seed)).float()
        dy = bf(rnd((B, C, H, H), seed + 1)).float()
        w = torch.zeros(C, C, 3, 3, requires_grad=True)
        F.conv2d(x, w, None, 1, 1).backward(dy)
        outs.append((bf(nhwc(x)).to(dev()), bf(nhwc(dy)).to(dev()), w.grad.permute(0, 2, 3, 1)))
    nbytes = 2 * _C.lib().fedfr_conv2d_wgrad_ws_bytes(B, H, C, C, 3, 1)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev())
    dwa = torch.full((C, 3, 3, C), float("nan"), device=dev())
    dwb = torch.full((C, 3, 3, C), float("nan"), device=dev())
    _C.call("fedfr_conv2d_wgrad_pair", outs[0][0].data_ptr(), outs[0][1].data_ptr(), dwa.data_ptr(), outs[1][0].data_ptr(),
            outs[1][1].data_ptr(), dwb.data_ptr(), ws.data_ptr(), nbytes, B, H, C, C, 3, 1, _C.stream())
    torch.cuda.synchronize()
    assert relerr(dwa, outs[0][2]) < 2e-4 and relerr(dwb, outs[1][2]) < 2e-4


@pytest.mark.parametrize("B,H,C", [(128, 14, 256), (16, 28, 128), (3, 56, 64), (1, 112, 64), (2, 14, 512), (5, 14, 128), (1, 14, 64)])
def test_conv_wgrad9_pair(B, H, C):
    """The paired 64 x 64 nine-tap kernel (wgrad9p.hip: the two 3x3 / stride-1 weight gradients of a residual block in one launch) ==
    autograd of F.conv2d on the same bf16 operands, and == the single-layer nine-tap kernel (option wgrad9p off) up to fp32 summation
    order; odd batch sizes (ragged last K-split) and every map width it serves."""
    outs = []
    for seed in (31, 47):
        x = bf(rnd((B, C, H, H), seed)).float()
        dy = bf(rnd((B, C, H, H), seed + 1)).float()
        w = torch.zeros(C, C, 3, 3, requires_grad=True)
        F.conv2d(x, w, None, 1, 1).backward(dy)
        outs.append((bf(nhwc(x)).to(dev()), bf(nhwc(dy)).to(dev()), w.grad.permute(0, 2, 3, 1)))
    nbytes = 2 * _C.lib().fedfr_conv2d_wgrad_ws_bytes(B, H, C, C, 3, 1)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev())
    res = {}
    for pair in (1, 0):
        dwa = torch.full((C, 3, 3, C), float("nan"), device=dev())
        dwb = torch.full((C, 3, 3, C), float("nan"), device=dev())
        with _C.option_scope("wgrad9p", pair):
            _C.call("fedfr_conv2d_wgrad_pair", outs[0][0].data_ptr(), outs[0][1].data_ptr(), dwa.data_ptr(), outs[1][0].data_ptr(),
                    outs[1][1].data_ptr(), dwb.data_ptr(), ws.data_ptr(), nbytes, B, H, C, C, 3, 1, _C.stream())
            torch.cuda.synchronize()
        assert relerr(dwa, outs[0][2]) < 2e-4 and relerr(dwb, outs[1][2]) < 2e-4, (pair, relerr(dwa, outs[0][2]), relerr(dwb, outs[1][2]))
        res[pair] = (dwa.cpu(), dwb.cpu())
    assert relerr(res[1][0], res[0][0]) < 1e-5 and relerr(res[1][1], res[0][1]) < 1e-5


@pytest.mark.parametrize("M,N,K", [(128, 512, 25088), (32, 512, 1024), (4, 512, 25088), (200, 1000, 512), (128, 64, 192)])
def test_gemm_nt_plain(M, N, K):
    a, b = bf(rnd((M, K), 1)), bf(rnd((N, K), 2, 0.05))
    ref = a.float() @ b.float().t()
    c = torch.full((M, N), float("nan"), device=dev())
    ws = torch.empty(128 * M * N * 4 + 16, dtype=torch.uint8, device=dev())
    ad, bd = a.to(dev()), b.to(dev())            # keep device copies alive until the kernel has run
    _C.call("fedfr_gemm_nt", ad.data_ptr(), bd.data_ptr(), c.data_ptr(), ws.data_ptr(), ws.numel(), M, N, K, _C.stream())
    torch.cuda.synchronize()
    assert relerr(c, ref) < 2e-4


@pytest.mark.parametrize("use_tr", [1, 0])
@pytest.mark.parametrize("Kp,NI,NJ", [(128, 512, 25088), (512, 128, 25088), (4, 512, 1024), (512, 8, 2048), (100, 64, 64)])
def test_gemm_tn_plain(Kp, NI, NJ, use_tr):
    p, q = bf(rnd((Kp, NI), 1)), bf(rnd((Kp, NJ), 2, 0.05))
    ref = p.float().t() @ q.float()
    c = torch.full((NI, NJ), float("nan"), device=dev())
    pd, qd = p.to(dev()), q.to(dev())
    _C.call("fedfr_set_option", b"tn_use_tr", use_tr)
    try:
        _C.call("fedfr_gemm_tn", pd.data_ptr(), qd.data_ptr(), c.data_ptr(), Kp, NI, NJ, _C.stream())
        torch.cuda.synchronize()
    finally:
        _C.call("fedfr_set_option", b"tn_use_tr", 1)
    assert relerr(c, ref) < 2e-4


@pytest.mark.parametrize("B,HW", [(2, 16), (3, 112), (1, 32)])
def test_stem_fwd_wgrad(B, HW):
    x = rnd((B, 3, HW, HW), 3)
    w = rnd((64, 3, 3, 3), 4, 0.2)
    xb, wb = bf(x).float(), bf(w).float()        # the kernel rounds x and w to bf16 for the MFMA
    ref = F.conv2d(xb, wb, None, 1, 1)
    wk = w.permute(0, 2, 3, 1).contiguous().to(dev())
    y = torch.empty(B, HW, HW, 64, dtype=S16(), device=dev())
    rows = _C.lib().fedfr_stem_stat_rows(B, HW)
    stats = torch.full((rows, 2, 64), float("nan"), device=dev())
    xd = x.to(dev())
    _C.call("fedfr_stem_fwd", xd.data_ptr(), wk.data_ptr(), y.data_ptr(), stats.data_ptr(), B, HW, _C.stream())
    torch.cuda.synchronize()
    assert relerr(y.float().cpu().permute(0, 3, 1, 2), ref) < 6e-3
    yr = y.float().cpu().reshape(-1, 64).double()
    st = stats.cpu().double().sum(0)
    np.testing.assert_allclose(st[0].numpy(), yr.sum(0).numpy(), rtol=1e-4, atol=1e-2)
    np.testing.assert_allclose(st[1].numpy(), (yr ** 2).sum(0).numpy(), rtol=1e-4, atol=1e-2)
    # wgrad: fp32 x (exact), bf16 dy
    dy = bf(rnd((B, 64, HW, HW), 5))
    wv = w.clone().requires_grad_(True)
    F.conv2d(x, wv, None, 1, 1).backward(dy.float())
    refw = wv.grad.permute(0, 2, 3, 1)
    dw = torch.full((64, 3, 3, 3), float("nan"), device=dev())
    ws = torch.empty(_C.lib().fedfr_stem_wgrad_ws_bytes(B, HW), dtype=torch.uint8, device=dev())
    dyd = nhwc(dy).to(dev())
    _C.call("fedfr_stem_wgrad", xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), ws.data_ptr(), B, HW, _C.stream())
    torch.cuda.synchronize()
    assert relerr(dw, refw) < 1e-4


@pytest.mark.parametrize("M,C,prelu,second", [(3000, 64, True, 0), (25088, 256, False, 1), (777, 512, False, 2), (40000, 128, True, 0)])
def test_bn_forward_chain(M, C, prelu, second):
    """stats(conv-epilogue emulated via bn_apply identity) -> finalize -> apply, vs F.batch_norm on the same bf16 tensor."""
    x = bf(rnd((M, C), 1) * 2 + 0.3)
    gamma, beta = rnd((C,), 2) * 0.2 + 1, rnd((C,), 3) * 0.1
    rm, rv = rnd((C,), 4) * 0.1, rnd((C,), 5) * 0.1 + 1
    alpha = rnd((C,), 6) * 0.1 + 0.25
    x2 = bf(rnd((M, C), 7))
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.batch_norm(x.float(), rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    if prelu:
        ref = F.prelu(ref, alpha)
    sc2 = sh2 = None
    if second == 1:
        ref = ref + x2.float()
    elif second == 2:
        sc2, sh2 = rnd((C,), 8) + 1.5, rnd((C,), 9)
        ref = ref + x2.float() * sc2 + sh2
    d = dev()
    xd = x.to(d)
    ones, zeros = torch.ones(C, device=d), torch.zeros(C, device=d)
    rows = _C.lib().fedfr_bn_apply_stat_rows(M, C)
    stats = torch.full((rows, 2, C), float("nan"), device=d)
    tmpy = torch.empty_like(xd)
    # identity apply just to produce the column statistics of x
    _C.call("fedfr_bn_apply", xd.data_ptr(), ones.data_ptr(), zeros.data_ptr(), None, None, None, None, tmpy.data_ptr(), M, C, 0,
            stats.data_ptr(), _C.stream())
    assert torch.equal(tmpy.cpu(), x)
    g_, b_, rm_, rv_ = gamma.to(d), beta.to(d), rm.to(d), rv.to(d)
    scale, shift, mean, rstd = (torch.empty(C, device=d) for _ in range(4))
    tmp = torch.empty(64 * 2 * C, device=d)
    _C.call("fedfr_bn_finalize", stats.data_ptr(), rows, C, float(M), g_.data_ptr(), b_.data_ptr(), rm_.data_ptr(), rv_.data_ptr(),
            0.1, 1e-5, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), tmp.data_ptr(), _C.stream())
    y = torch.empty_like(xd)
    al = alpha.to(d) if prelu else None
    x2d = x2.to(d) if second else None
    sc2d, sh2d = (sc2.to(d), sh2.to(d)) if second == 2 else (None, None)
    _C.call("fedfr_bn_apply", xd.data_ptr(), scale.data_ptr(), shift.data_ptr(), _C.ptr(al), _C.ptr(x2d), _C.ptr(sc2d), _C.ptr(sh2d),
            y.data_ptr(), M, C, 0, None, _C.stream())
    torch.cuda.synchronize()
    torch.testing.assert_close(mean.cpu(), x.float().mean(0), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rm_.cpu(), rm_ref, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rv_.cpu(), rv_ref, rtol=1e-4, atol=1e-5)
    # bf16 output rounding: 2^-8 relative
    assert relerr(y.float(), ref) < 6e-3


@pytest.mark.parametrize("M,C,prelu,addmode", [(3000, 64, True, 0), (6272, 256, False, 1), (4 * 16 * 16, 128, False, 2)])
def test_bn_backward(M, C, prelu, addmode):
    x = bf(rnd((M, C), 1) * 2 + 0.3)
    dy = bf(rnd((M, C), 2))
    gamma, beta = rnd((C,), 3) * 0.2 + 1, rnd((C,), 4) * 0.1
    alpha = rnd((C,), 5) * 0.1 + 0.25
    xg = x.float().requires_grad_(True)
    gg, bb, aa = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True), alpha.clone().requires_grad_(True)
    out = F.batch_norm(xg, None, None, gg, bb, True, 0.1, 1e-5)
    if prelu:
        out = F.prelu(out, aa)
    out.backward(dy.float())
    ref = xg.grad.clone()
    add = add_up = None
    H = 0
    if addmode == 1:
        add = bf(rnd((M, C), 6))
        ref = ref + add.float()
    elif addmode == 2:
        H = 16
        add_up = bf(rnd((4, 8, 8, C), 7))
        up = torch.zeros(4, 16, 16, C)
        up[:, ::2, ::2, :] = add_up.float()
        ref = ref + up.reshape(M, C)
    d = dev()
    mean = x.float().mean(0)
    rstd = 1.0 / torch.sqrt(x.float().var(0, unbiased=False) + 1e-5)
    rows = _C.lib().fedfr_bn_bwd_rows(M, C)
    part = torch.full((rows, 3, C), float("nan"), device=d)
    coef = torch.empty(3, C, device=d)
    dg, db, da = (torch.full((C,), float("nan"), device=d) for _ in range(3))
    dx = torch.empty(M, C, dtype=S16(), device=d)
    t = lambda v: None if v is None else v.to(d)   # noqa: E731
    keep = [t(dy), t(x), t(mean), t(rstd), t(gamma), t(beta), t(alpha) if prelu else None, t(add), t(add_up)]
    _C.call("fedfr_bn_bwd", keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(),
            keep[5].data_ptr(), _C.ptr(keep[6]), M, C, part.data_ptr(), coef.data_ptr(), dg.data_ptr(), db.data_ptr(),
            da.data_ptr() if prelu else None, _C.ptr(keep[7]), _C.ptr(keep[8]), H, dx.data_ptr(), _C.stream())
    torch.cuda.synchronize()
    assert relerr(dx.float(), ref) < 8e-3
    torch.testing.assert_close(dg.cpu(), gg.grad, rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(db.cpu(), bb.grad, rtol=2e-3, atol=2e-3)
    if prelu:
        torch.testing.assert_close(da.cpu(), aa.grad, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("M,C,prelu,second,rows_in", [(25088, 256, False, 1, 256), (6272, 512, True, 0, 98), (3001, 64, True, 1, 7),
                                                      (100352, 128, False, 0, 64), (1000, 256, False, 1, 1), (784, 256, True, 1, 8)])
def test_bn_apply_sliced(M, C, prelu, second, rows_in):
    """channel-sliced apply pass that reduces the partial rows itself vs F.batch_norm (+PReLU, + identity) on the same bf16 tensor; the
    statistics of its OUTPUT (rows for the next BatchNorm) vs the output's column sums; ragged M and a single partial row included."""
    assert _C.lib().fedfr_bn_sliced_ok(M, C, rows_in, 0) == 1
    x = bf(rnd((M, C), 1) * 2 + 0.3)
    gamma, beta = rnd((C,), 2) * 0.2 + 1, rnd((C,), 3) * 0.1
    rm, rv = rnd((C,), 4) * 0.1, rnd((C,), 5) * 0.1 + 1
    alpha = rnd((C,), 6) * 0.1 + 0.25
    x2 = bf(rnd((M, C), 7))
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.batch_norm(x.float(), rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    if prelu:
        ref = F.prelu(ref, alpha)
    if second:
        ref = ref + x2.float()
    d = dev()
    # partial rows as a producer would leave them: rows_in slabs of the pixels, (sum, sumsq) per channel
    xf = x.float()
    bounds = np.linspace(0, M, rows_in + 1).astype(int)
    part = torch.stack([torch.stack([xf[a:b].sum(0), (xf[a:b] ** 2).sum(0)]) for a, b in zip(bounds[:-1], bounds[1:])]).to(d)
    xd, g_, b_, rm_, rv_ = x.to(d), gamma.to(d), beta.to(d), rm.to(d), rv.to(d)
    scale, shift, mean, rstd = (torch.full((C,), float("nan"), device=d) for _ in range(4))
    rows = _C.lib().fedfr_bn_sliced_rows(M, C, 0)
    stats = torch.full((rows, 2, C), float("nan"), device=d)
    y = torch.empty_like(xd)
    al = alpha.to(d) if prelu else None
    x2d = x2.to(d) if second else None
    _C.call("fedfr_bn_apply_sliced", part.data_ptr(), rows_in, float(M), g_.data_ptr(), b_.data_ptr(), rm_.data_ptr(), rv_.data_ptr(), 0.1, 1e-5,
            scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), xd.data_ptr(), _C.ptr(al), _C.ptr(x2d), y.data_ptr(), M, C,
            stats.data_ptr(), _C.stream())
    torch.cuda.synchronize()
    torch.testing.assert_close(mean.cpu(), xf.mean(0), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rstd.cpu(), 1 / torch.sqrt(xf.var(0, unbiased=False) + 1e-5), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(scale.cpu(), gamma * rstd.cpu(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(shift.cpu(), beta - mean.cpu() * gamma * rstd.cpu(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rm_.cpu(), rm_ref, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rv_.cpu(), rv_ref, rtol=1e-4, atol=1e-5)
    assert relerr(y.float(), ref) < 6e-3
    yf = y.float().cpu()
    torch.testing.assert_close(stats[:, 0].sum(0).cpu(), yf.sum(0), rtol=1e-4, atol=2e-2)
    torch.testing.assert_close(stats[:, 1].sum(0).cpu(), (yf ** 2).sum(0), rtol=1e-4, atol=2e-2)
    # an error, not a wrong answer, for the shapes the sliced passes do not serve
    assert _C.lib().fedfr_bn_sliced_ok(128 * 112 * 112, 64, 64, 0) == 0 and _C.lib().fedfr_bn_sliced_ok(M, C, 4096, 0) == 0
    with pytest.raises(RuntimeError):
        _C.call("fedfr_bn_apply_sliced", part.data_ptr(), 4096, float(M), g_.data_ptr(), b_.data_ptr(), None, None, 0.1, 1e-5, scale.data_ptr(),
                shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), xd.data_ptr(), None, None, y.data_ptr(), M, C, None, _C.stream())


@pytest.mark.parametrize("M,C,prelu,with_add,with_next", [(25088, 256, True, False, False), (25088, 256, False, True, True),
                                                          (6272, 512, False, False, True), (3001, 64, True, True, True),
                                                          (100352, 128, False, True, False), (784, 256, False, True, True),
                                                          (784, 256, True, False, False), (1176, 512, False, True, True), (12544, 256, False, True, False),
                                                          (50176, 128, False, True, False), (3136, 512, False, True, False),
                                                          (12544, 256, False, False, False)])
def test_bn_bwd_sliced(M, C, prelu, with_add, with_next):
    """channel-sliced BatchNorm(+PReLU) backward (reduce pass + apply pass that reduces the rows itself) vs autograd; the rows it leaves for
    the NEXT BatchNorm backward (sum dx, sum dx * xhat_next) vs the same sums of its bf16 output; and vs the row-slab path (fedfr_bn_bwd)."""
    x = bf(rnd((M, C), 1) * 2 + 0.3)
    dy = bf(rnd((M, C), 2))
    gamma, beta = rnd((C,), 3) * 0.2 + 1, rnd((C,), 4) * 0.1
    alpha = rnd((C,), 5) * 0.1 + 0.25
    xg = x.float().requires_grad_(True)
    gg, bb, aa = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True), alpha.clone().requires_grad_(True)
    out = F.batch_norm(xg, None, None, gg, bb, True, 0.1, 1e-5)
    if prelu:
        out = F.prelu(out, aa)
    out.backward(dy.float())
    ref = xg.grad.clone()
    add = bf(rnd((M, C), 6)) if with_add else None
    if with_add:
        ref = ref + add.float()
    d = dev()
    mean = x.float().mean(0)
    rstd = 1.0 / torch.sqrt(x.float().var(0, unbiased=False) + 1e-5)
    sc, sh = gamma * rstd, beta - mean * gamma * rstd
    nx = bf(rnd((M, C), 8) * 1.5 - 0.2) if with_next else None
    nmean = nx.float().mean(0) if with_next else None
    nrstd = 1.0 / torch.sqrt(nx.float().var(0, unbiased=False) + 1e-5) if with_next else None
    rows = _C.lib().fedfr_bn_sliced_rows(M, C, 1)
    part = torch.full((rows, 3, C), float("nan"), device=d)
    npart = torch.full((rows, 3, C), float("nan"), device=d)
    dg, db, da = (torch.full((C,), float("nan"), device=d) for _ in range(3))
    dx = torch.empty(M, C, dtype=S16(), device=d)
    t = lambda v: None if v is None else v.to(d)   # noqa: E731
    k = dict(dy=t(dy), x=t(x), mean=t(mean), rstd=t(rstd), gamma=t(gamma), alpha=t(alpha) if prelu else None, sc=t(sc), sh=t(sh), add=t(add),
             nx=t(nx), nmean=t(nmean), nrstd=t(nrstd))
    _C.call("fedfr_bn_bwd_sliced", k["dy"].data_ptr(), k["x"].data_ptr(), k["mean"].data_ptr(), k["rstd"].data_ptr(), k["gamma"].data_ptr(),
            _C.ptr(k["alpha"]), k["sc"].data_ptr(), k["sh"].data_ptr(), M, C, part.data_ptr(), 0, dg.data_ptr(), db.data_ptr(),
            da.data_ptr() if prelu else None, _C.ptr(k["add"]), dx.data_ptr(), _C.ptr(k["nx"]), _C.ptr(k["nmean"]), _C.ptr(k["nrstd"]),
            npart.data_ptr() if with_next else None, _C.stream())
    torch.cuda.synchronize()
    assert relerr(dx.float(), ref) < 8e-3
    torch.testing.assert_close(dg.cpu(), gg.grad, rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(db.cpu(), bb.grad, rtol=2e-3, atol=2e-3)
    if prelu:
        torch.testing.assert_close(da.cpu(), aa.grad, rtol=2e-3, atol=2e-3)
    if with_next:
        dxf = dx.float().cpu()
        torch.testing.assert_close(npart[:, 0].sum(0).cpu(), dxf.sum(0), rtol=1e-3, atol=2e-2)
        torch.testing.assert_close(npart[:, 1].sum(0).cpu(), (dxf * (nx.float() - nmean) * nrstd).sum(0), rtol=1e-3, atol=5e-2)
        assert float(npart[:, 2].abs().max()) == 0.0
    # same numbers as the row-slab kernels (reduce / finalize / apply), up to the summation order
    rows_o = _C.lib().fedfr_bn_bwd_rows(M, C)
    part_o = torch.empty(rows_o, 3, C, device=d)
    coef = torch.empty(3, C, device=d)
    dg2, db2, da2 = (torch.empty(C, device=d) for _ in range(3))
    dx2 = torch.empty_like(dx)
    beta_d = t(beta)
    _C.call("fedfr_bn_bwd", k["dy"].data_ptr(), k["x"].data_ptr(), k["mean"].data_ptr(), k["rstd"].data_ptr(), k["gamma"].data_ptr(),
            beta_d.data_ptr(), _C.ptr(k["alpha"]), M, C, part_o.data_ptr(), coef.data_ptr(), dg2.data_ptr(), db2.data_ptr(),
            da2.data_ptr() if prelu else None, _C.ptr(k["add"]), None, 0, dx2.data_ptr(), _C.stream())
    torch.cuda.synchronize()
    torch.testing.assert_close(dg.cpu(), dg2.cpu(), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(db.cpu(), db2.cpu(), rtol=1e-4, atol=1e-3)
    assert relerr(dx.float(), dx2.float()) < 4e-3      # bf16 outputs of coefficients that differ in the last fp32 bits


@pytest.mark.parametrize("M,N,K,ta,tb", [(128, 1000, 512, False, True), (128, 512, 1000, False, False), (1000, 512, 128, True, False),
                                        (7, 33, 19, False, True), (65, 130, 40, True, False)])
def test_sgemm(M, N, K, ta, tb):
    a = rnd((K, M) if ta else (M, K), 1)
    b = rnd((N, K) if tb else (K, N), 2)
    A = a.t() if ta else a
    Bm = b.t() if tb else b
    ref = (A.double() @ Bm.double()).float()
    d = dev()
    ad, bd = a.to(d), b.to(d)
    c = torch.full((M, N), float("nan"), device=d)
    sam, sak = (1, M) if ta else (K, 1)
    sbk, sbn = (1, K) if tb else (N, 1)
    _C.call("fedfr_sgemm", ad.data_ptr(), bd.data_ptr(), c.data_ptr(), M, N, K, sam, sak, sbk, sbn, N, 1.0, 0.0, None, _C.stream())
    torch.cuda.synchronize()
    assert relerr(c, ref) < 1e-5


@pytest.mark.parametrize("M,N,K,tb,splits", [(128, 1000, 512, True, 4), (128, 512, 1000, False, 7), (37, 130, 300, True, 3), (128, 512, 1000, False, 1)])
def test_sgemm_splitk_slabs(M, N, K, tb, splits):
    """split-K slabs of the head GEMMs: their sum is the product; every slab is the product over its own k range."""
    from fedfr_amd import ops
    a, b = rnd((M, K), 1), rnd((N, K) if tb else (K, N), 2)
    Bm = b.t() if tb else b
    d = dev()
    slabs = ops.sgemm(a.to(d), b.to(d), trans_b=tb, splits=splits)
    torch.cuda.synchronize()
    assert slabs.shape == (splits, M, N)
    assert relerr(slabs.sum(0), (a.double() @ Bm.double()).float()) < 1e-5
    kc = -(-(-(-K // splits)) // 32) * 32
    for z in range(splits):
        lo, hi = z * kc, min(K, (z + 1) * kc)
        assert relerr(slabs[z], (a[:, lo:hi].double() @ Bm[lo:hi].double()).float()) < 1e-5, z


@pytest.mark.parametrize("M,N,K,tb", [(128, 512, 8500, False), (256, 512, 85000, False), (128, 1000, 512, True), (64, 64, 1024, True), (37, 130, 3000, False), (36, 130, 3000, False)])
def test_sgemm_auto_split_and_sum_slabs(M, N, K, tb):
    """ops.sgemm with splits=0 owns the output: few tiles over a long reduction are split over K and summed in ascending slab order — equal to
    sum_slabs() of the explicit slabs bit for bit, and to the fp64 product within the fp32 head tolerance; short / wide products stay unsplit."""
    from fedfr_amd import ops
    a, b = rnd((M, K), 11), rnd((N, K) if tb else (K, N), 12)
    Bm = b.t() if tb else b
    d = dev()
    ad, bd = a.to(d), b.to(d)
    got = ops.sgemm(ad, bd, trans_b=tb)
    ns = ops._auto_splits(M, N, K)
    assert got.shape == (M, N)
    assert relerr(got, (a.double() @ Bm.double()).float()) < 1e-5
    if ns > 1:
        slabs = ops.sgemm(ad, bd, trans_b=tb, splits=ns)
        assert slabs.shape[0] == ns
        assert torch.equal(got, ops.sum_slabs(slabs))
        ref = slabs[0].clone()
        for z in range(1, ns):
            ref += slabs[z]
        assert relerr(got, ref) < 1e-6
    else:
        assert torch.equal(got, ops.sgemm(ad, bd, trans_b=tb, splits=1).reshape(M, N))
    assert (ns > 1) == (K >= 1024 and -(-M // 64) * -(-N // 64) < 128 and M * N % 4 == 0)      # (37 x 130: slabs would not start on 16 bytes)


@pytest.mark.parametrize("R,C,arc,nslab", [(128, 1000, False, 1), (128, 1000, True, 1), (33, 1000, False, 4), (5, 3000, True, 3), (9, 257, False, 2),
                                           (128, 8500, True, 1), (7, 16384, False, 2)])      # (round 5: rows of up to 16 384 classes — the sampled PartialFC head)
def test_softmax_ce_fused_equals_three_kernels(R, C, arc, nslab):
    """margin -> softmax -> gradient in one launch: bit-identical to the three-kernel chain on the summed slabs (labels incl. -1)."""
    from fedfr_amd import ops
    d = dev()
    parts = (rnd((nslab, R, C), 3) * (0.9 / nslab)).to(d)                       # cosines in (-0.9, 0.9) after the sum
    total = parts[0].clone()
    for q in range(1, nslab):
        total += parts[q]
    lab = torch.randint(0, C, (R,), generator=torch.Generator().manual_seed(5))
    lab[::7] = -1
    lab = lab.to(d)
    p_ref, g_ref = ops.softmax_ce_grad(total.clone(), lab, 30.0, 0.4, arc, 1.0 / R)
    p_got, g_got = ops.softmax_ce_fused(parts.clone(), lab, 30.0, 0.4, arc, 1.0 / R)
    torch.cuda.synchronize()
    assert torch.equal(p_got, p_ref) and torch.equal(g_got, g_ref)


def test_normalize_rows_bwd_slabs():
    from fedfr_amd import ops
    d = dev()
    x = rnd((128, 512), 1).to(d)
    xn, inv = ops.normalize_rows(x)
    parts = rnd((5, 128, 512), 2).to(d)
    total = parts[0].clone()
    for q in range(1, 5):
        total += parts[q]
    ref = ops.normalize_rows_bwd(xn, inv, total)
    got = ops.normalize_rows_bwd_slabs(xn, inv, parts)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)


def test_sgd_matches_golden():
    from conftest import load_golden
    from oracle import ref_cpu as R
    g = load_golden("sgd")
    d = dev()
    shapes = [(7, 5), (33,), (4, 3, 3, 3)]
    ps = [R.closed_form(s, 0.2 + 0.1 * i, 0.3 * i, 0.5) for i, s in enumerate(shapes)]
    # one flat buffer, each tensor padded to 4 floats — exactly how the product lays parameters out
    offs, n = [], 0
    for p in ps:
        offs.append(n)
        n += (p.numel() + 3) // 4 * 4
    flat, grad, buf = torch.zeros(n, device=d), torch.zeros(n, device=d), torch.zeros(n, device=d)
    for o, p in zip(offs, ps):
        flat[o:o + p.numel()] = p.flatten().to(d)
    for step in range(3):
        for i, (o, p) in enumerate(zip(offs, ps)):
            gr = R.closed_form(tuple(p.shape), 0.15 + 0.05 * i + 0.01 * step, 0.7 * step, 0.3)
            grad[o:o + p.numel()] = gr.flatten().to(d)
        _C.call("fedfr_sgd_step", flat.data_ptr(), grad.data_ptr(), buf.data_ptr(), None, n, 0.1, 0.9, 5e-4, 1 if step == 0 else 0,
                _C.stream())
        torch.cuda.synchronize()
        for i, (o, p) in enumerate(zip(offs, ps)):
            got_p = flat[o:o + p.numel()].cpu().reshape(p.shape)
            got_m = buf[o:o + p.numel()].cpu().reshape(p.shape)
            torch.testing.assert_close(got_p, torch.from_numpy(g["p%d_s%d" % (i, step)]), rtol=1e-6, atol=1e-7)
            torch.testing.assert_close(got_m, torch.from_numpy(g["m%d_s%d" % (i, step)]), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("first", [1, 0])
def test_sgd_step_scaled_unscales_in_place_and_skips_non_finite_gradients(first):
    """fedfr_sgd_step_scaled (the GradScaler.step of client.py:394-396 with a static scale, no host sync): the buffer holds gradient / grad_scale;
    finite elements are updated bit-identically to `unscale, then fedfr_sgd_step` and the buffer holds the true gradient afterwards; an element
    whose gradient is inf / nan keeps parameter, momentum and mirror (momentum 0 on the first step) and the overflow word is set."""
    d = dev()
    n = 4099                                                    # float4 body + scalar tail
    p0 = rnd((n,), 1).to(d)
    g_true = rnd((n,), 2, 0.3).to(d)
    m0 = rnd((n,), 3, 0.1).to(d)
    S = 256.0
    bad = torch.tensor([0, 5, 1023, 4096, 4098], device=d)
    vals = torch.tensor([float("inf"), float("nan"), -float("inf"), float("nan"), float("inf")], device=d)
    # reference: plain kernel on the true gradient
    pr, gr, mr = p0.clone(), g_true.clone(), m0.clone()
    shr = torch.zeros(n, dtype=S16(), device=d)
    _C.call("fedfr_sgd_step", pr.data_ptr(), gr.data_ptr(), mr.data_ptr(), shr.data_ptr(), n, 0.1, 0.9, 5e-4, first, _C.stream())
    # scaled kernel, clean gradient: identical, flag clear
    ps, gs, ms = p0.clone(), g_true * S, m0.clone()
    shs = torch.zeros(n, dtype=S16(), device=d)
    ovf = torch.zeros(1, dtype=torch.int32, device=d)
    _C.call("fedfr_sgd_step_scaled", ps.data_ptr(), gs.data_ptr(), ms.data_ptr(), shs.data_ptr(), n, 0.1, 0.9, 5e-4, first, 1.0 / S, ovf.data_ptr(), _C.stream())
    torch.cuda.synchronize()
    assert torch.equal(ps, pr) and torch.equal(ms, mr) and torch.equal(gs, g_true) and torch.equal(shs.view(torch.int16), shr.view(torch.int16))
    assert int(ovf.item()) == 0
    # poisoned gradient: those elements are skipped, everything else as before
    pq, gq, mq = p0.clone(), g_true * S, m0.clone()
    gq[bad] = vals
    shq = torch.full((n,), 7.0, dtype=S16(), device=d)
    _C.call("fedfr_sgd_step_scaled", pq.data_ptr(), gq.data_ptr(), mq.data_ptr(), shq.data_ptr(), n, 0.1, 0.9, 5e-4, first, 1.0 / S, ovf.data_ptr(), _C.stream())
    torch.cuda.synchronize()
    assert int(ovf.item()) == 1
    keep = torch.ones(n, dtype=torch.bool, device=d)
    keep[bad] = False
    assert torch.equal(pq[keep], pr[keep]) and torch.equal(mq[keep], mr[keep])
    assert torch.equal(pq[bad], p0[bad])
    assert torch.equal(mq[bad], torch.zeros_like(m0[bad]) if first else m0[bad])
    assert torch.equal(shq[bad].float(), p0[bad].to(S16()).float())                   # the mirror of a skipped element = its (unchanged) parameter
    assert bool(torch.isfinite(pq).all()) and bool(torch.isfinite(mq).all())


def test_fedavg_axpy_bit_exact():
    ws = [1200 / 5100, 800 / 5100, 3100 / 5100]
    xs = [rnd((100003,), i) for i in range(3)]
    ref = 0
    for w, x in zip(ws, xs):
        ref = ref + w * x                      # server.py:27-33 op order
    d = dev()
    out = torch.empty(100003, device=d)
    xds = [x.to(d) for x in xs]
    for i, (w, x) in enumerate(zip(ws, xds)):
        _C.call("fedfr_fedavg_axpy", out.data_ptr(), x.data_ptr(), float(np.float32(w)), 100003, 1 if i else 0, _C.stream())
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("n,k,npos", [(85000, 8500, 128), (2000, 200, 8), (10625, 1062, 600), (1000, 1000, 5), (70000, 3, 2)])
def test_pfc_topk_matches_torch(n, k, npos):
    perm = torch.rand(n, generator=torch.Generator().manual_seed(n))
    pos = torch.randperm(n, generator=torch.Generator().manual_seed(k))[:npos]
    perm[pos] = 2.0
    ref = torch.topk(perm, k)[1].sort()[0]
    d = dev()
    idx = torch.full((k,), -1, dtype=torch.int64, device=d)
    cnt = torch.zeros(1, dtype=torch.int32, device=d)
    permd = perm.to(d)
    _C.call("fedfr_pfc_topk", permd.data_ptr(), n, k, idx.data_ptr(), cnt.data_ptr(), _C.stream())
    torch.cuda.synchronize()
    assert int(cnt) == npos
    assert torch.equal(idx.cpu(), ref)


@pytest.mark.parametrize("n,k,levels", [(85000, 8500, 50), (40000, 7777, 16), (20001, 20001, 3), (16384, 1, 2), (33000, 32999, 7)])
def test_pfc_topk_with_ties(n, k, levels):
    """Heavily tied values (a handful of distinct levels): the selection takes every value above the k-th largest and the FIRST ones in index
    order of those equal to it — across the chunk boundaries of the wide-load kernel (round 5) exactly as torch's stable ordering does."""
    g = torch.Generator().manual_seed(n + k)
    perm = torch.randint(0, levels, (n,), generator=g).float() / levels
    order = torch.argsort(perm, descending=True, stable=True)                       # ties: ascending index
    ref = order[:k].sort()[0]
    d = dev()
    idx = torch.full((k,), -1, dtype=torch.int64, device=d)
    cnt = torch.zeros(1, dtype=torch.int32, device=d)
    permd = perm.to(d)
    _C.call("fedfr_pfc_topk", permd.data_ptr(), n, k, idx.data_ptr(), cnt.data_ptr(), _C.stream())
    torch.cuda.synchronize()
    assert int(cnt) == 0
    assert torch.equal(idx.cpu(), ref)


def test_rows_gather_scatter_and_remap():
    d = dev()
    w = rnd((1000, 512), 1).to(d)
    index = torch.tensor(sorted(np.random.RandomState(0).choice(1000, 100, replace=False)), dtype=torch.int64, device=d)
    sub = torch.empty(100, 512, device=d)
    _C.call("fedfr_rows_gather", sub.data_ptr(), w.data_ptr(), index.data_ptr(), 100, 512, 1000, _C.stream())
    torch.cuda.synchronize()
    assert torch.equal(sub, w[index])
    w2 = w.clone()
    sub2 = (sub * 2).contiguous()
    _C.call("fedfr_rows_scatter", w2.data_ptr(), sub2.data_ptr(), index.data_ptr(), 100, 512, 1000, _C.stream())
    torch.cuda.synchronize()
    ref = w.clone()
    ref[index] = sub * 2
    assert torch.equal(w2, ref)
    # out-of-range indices are ignored (never a wild access)
    bad = torch.tensor([5, -3, 10 ** 9, 7], dtype=torch.int64, device=d)
    w3 = w.clone()
    _C.call("fedfr_rows_scatter", w3.data_ptr(), sub2.data_ptr(), bad.data_ptr(), 4, 512, 1000, _C.stream())
    torch.cuda.synchronize()
    assert torch.equal(w3[5], sub2[0]) and torch.equal(w3[7], sub2[3]) and torch.equal(w3[6], w[6])
    lab = torch.tensor([int(index[3]), -1, int(index[99]), int(index[0])], dtype=torch.int64, device=d)
    _C.call("fedfr_pfc_remap", lab.data_ptr(), 4, index.data_ptr(), 100, _C.stream())
    torch.cuda.synchronize()
    assert lab.tolist() == [3, -1, 99, 0]


@pytest.mark.parametrize("B,D,temp", [(8, 512, 0.5), (37, 512, 0.5), (5, 96, 0.07)])
def test_contrastive_vs_torch(B, D, temp):
    """fedfr_contrastive == CE([cos(x,g)/T, cos(x,l)/T], 0) built from nn.CosineSimilarity + F.cross_entropy (client.py:372-375)."""
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(B * 7 + D)
    x = torch.randn(B, D, generator=gen, dtype=torch.float64, requires_grad=True)
    g_ = x.detach() + 0.5 * torch.randn(B, D, generator=gen, dtype=torch.float64)
    l_ = torch.randn(B, D, generator=gen, dtype=torch.float64)
    cs = torch.nn.CosineSimilarity(dim=1)
    ref = F.cross_entropy(torch.stack([cs(x, g_) / temp, cs(x, l_) / temp], dim=1), torch.zeros(B, dtype=torch.long))
    ref.backward()
    xd = x.detach().float().to(dev()).requires_grad_(True)
    gd, ld = g_.float().to(dev()), l_.float().to(dev())
    loss = ops.contrastive_loss(xd, gd, ld, temp)
    (loss * 3.0).backward()
    assert abs(float(loss) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    err = (xd.grad.double().cpu() / 3.0 - x.grad).abs().max() / x.grad.abs().max()
    assert float(err) < 1e-4, float(err)


@pytest.mark.parametrize("opt,val", [("tn_glds", 0), ("tn_glds", 1)])
def test_wgrad_kernel_variants(opt, val):
    """the non-default weight-gradient kernel variants kept as validation fallbacks (register-staged / 4-wave LDS-DMA GEMM form) stay
    parity-correct: fwd + dgrad + wgrad of one 14x14 and one 28x28 layer."""
    default = {"tn_glds": 2}[opt]
    _C.call("fedfr_set_option", opt.encode(), val)
    try:
        for case in [(131, 14, 256, 256, 3, 1), (65, 28, 128, 128, 3, 1)]:
            test_conv_fwd_and_stats(*case)
            test_conv_dgrad(*case)
            test_conv_wgrad(*case, 1)
    finally:
        _C.call("fedfr_set_option", opt.encode(), default)


NT_GLDS_CASES = [
    (128, 7, 512, 512, 3, 1),    # the 7x7 stage's convs at the bench batch: M = 6272 -> 64-row tiles
    (3, 7, 512, 512, 3, 1),      # ragged M
    (16, 28, 256, 256, 3, 2),    # stride-2 conv: fwd, and dgrad by output-parity class
    (40, 56, 128, 128, 3, 2),    # ... M = 31360 -> 128-row tiles (384 tiles and more)
    (5, 8, 512, 512, 3, 2),
    (16, 28, 128, 256, 1, 2),    # 1x1 / stride-2 downsample
    (70, 56, 64, 128, 1, 2),
    (4, 14, 192, 128, 3, 1),     # three channel chunks per tap
    (1, 14, 256, 256, 3, 1),
    (2, 16, 64, 128, 3, 2),
]


@pytest.mark.parametrize("val", [0, 9, 10, 11, 12])
def test_nt_glds_variants(val):
    """gemm_nt_glds.hip (option nt_glds: the register-staged NT kernel's shapes on an LDS-DMA operand ring, 4 or 8 waves per 128-row tile,
    64-row shapes on 64- or 128-row tiles; + 8 = also where the default policy keeps the register-staged kernel; 0 = that kernel everywhere):
    forward + BatchNorm partial rows, dgrad (stride-2: parity classes in one launch and one launch per class, and the masked single
    GEMM), plain GEMMs with split K."""
    _C.call("fedfr_set_option", b"nt_glds", val)
    try:
        # BASELINE's full sizes (batch 128) of the stride-2 layers once, on the variant the default policy uses
        full = [(128, 28, 256, 256, 3, 2), (128, 14, 512, 512, 3, 2)] if val == 12 else []
        for case in NT_GLDS_CASES + full:
            test_conv_fwd_and_stats(*case)
            test_conv_dgrad(*case)
        for par in (1, 0):
            _C.call("fedfr_set_option", b"dgrad_parity", par)
            test_conv_dgrad(16, 28, 256, 256, 3, 2)
            test_conv_dgrad(5, 8, 512, 512, 3, 2)
        for mnk in [(128, 512, 25088), (32, 512, 1024), (4, 512, 25088), (200, 1000, 512), (300, 136, 200)]:
            test_gemm_nt_plain(*mnk)
    finally:
        _C.call("fedfr_set_option", b"dgrad_parity", 2)
        _C.call("fedfr_set_option", b"nt_glds", NT_GLDS_DEFAULT)


@pytest.mark.parametrize("M,N,K,thr", [(37, 1000, 512, 0.12), (130, 4099, 512, 0.15), (5, 70, 96, 0.2)])
def test_sgemm_colflag_vs_torch(M, N, K, thr):
    """threshold + column-OR epilogue == set(torch.where(a @ b.T > thr)[1]) (client.py:213-222); entries within 1e-5 of the
    threshold are excluded from the comparison (fp32 summation order)."""
    gen = torch.Generator().manual_seed(M + N)
    a = F.normalize(torch.randn(M, K, generator=gen))
    b = F.normalize(torch.randn(N, K, generator=gen))
    sim = (a.double() @ b.double().t())
    flags = ops.similarity_column_flags(a.to(dev()), b.to(dev()), thr).cpu().bool()
    sure_hit = (sim > thr + 1e-5).any(dim=0)
    sure_miss = (sim < thr - 1e-5).all(dim=0)
    assert bool(flags[sure_hit].all()) and not bool(flags[sure_miss].any())
    assert 0 < int(flags.sum()) < N


@pytest.mark.parametrize("B,D,C", [(512, 512, 6000), (7, 512, 5), (1500, 64, 33)])
def test_class_accumulate_vs_torch(B, D, C):
    gen = torch.Generator().manual_seed(B)
    x = torch.randn(B, D, generator=gen)
    lab = torch.randint(0, C, (B,), generator=gen)
    sums = torch.randn(C, D, generator=gen)
    cnt = torch.randint(0, 5, (C,), generator=gen).float()
    ref_s = sums.double().clone().index_add_(0, lab, x.double())
    ref_c = cnt + torch.bincount(lab, minlength=C).float()
    sd_, cd_ = sums.to(dev()), cnt.to(dev())
    ops.class_accumulate(x.to(dev()), lab.to(dev()), sd_, cd_)
    assert torch.equal(cd_.cpu(), ref_c)
    assert float((sd_.cpu().double() - ref_s).abs().max()) < 1e-4


@pytest.mark.parametrize("N,D,T", [(72, 64, 30), (300, 512, 131), (1000, 512, 1000), (65, 48, 1)])
def test_roc_histogram_vs_oracle(N, D, T):
    """fp64-MFMA pair histogram == the oracle (float64 dots of the float32 features, roc_cuda.py:14-30), bin for bin."""
    from oracle import ref_cpu as R
    from fedfr_amd import eval_roc
    gen = torch.Generator().manual_seed(N + T)
    nid = 11
    lab = torch.randint(0, nid, (N,), generator=gen)
    cen = F.normalize(torch.randn(nid, D, generator=gen))
    f = F.normalize(cen[lab] + 0.3 * torch.randn(N, D, generator=gen))
    ref = R.roc_histogram(f.numpy(), lab.numpy(), T)
    got = eval_roc.roc_histogram(f.to(dev()), lab.to(dev()), T).cpu().numpy()
    assert int(got.sum()) == T * (T - 1) // 2 + T * (N - T)
    assert np.array_equal(got, ref)


def test_preprocess_u8_bit_exact():
    """dataset.py:81-92 on the device: ToTensor (x / 255) + Normalize(0.5, 0.5) (+ horizontal flip) — bit-exact with the same
    fp32 torch ops; flipped images equal torch.flip of the unflipped result."""
    gen = torch.Generator().manual_seed(3)
    u8 = torch.randint(0, 256, (5, 112, 112, 3), generator=gen, dtype=torch.uint8)
    u8[0] = torch.arange(112 * 112 * 3, dtype=torch.int64).remainder(256).to(torch.uint8).reshape(112, 112, 3)    # every byte value
    flip = torch.tensor([0, 1, 0, 1, 1], dtype=torch.uint8)
    ref = u8.permute(0, 3, 1, 2).float().div(255).sub(0.5).div(0.5)
    ref = torch.where(flip.bool()[:, None, None, None], torch.flip(ref, dims=[3]), ref)
    got = ops.preprocess_u8(u8.to(dev()), flip.to(dev())).cpu()
    assert torch.equal(got, ref)
    assert torch.equal(ops.preprocess_u8(u8.to(dev())).cpu(), u8.permute(0, 3, 1, 2).float().div(255).sub(0.5).div(0.5))


@pytest.mark.parametrize("M,C,with_bias,with_add", [(3000, 64, True, False), (777, 256, False, True)])
def test_bias_prelu_bwd_vs_torch(M, C, with_bias, with_add):
    """fedfr_bias_prelu_bwd == autograd of prelu(x + bias) (sphnet.py:53-60)."""
    gen = torch.Generator().manual_seed(M)
    x = bf(torch.randn(M, C, generator=gen)).float()
    dy = bf(torch.randn(M, C, generator=gen)).float()
    add = bf(torch.randn(M, C, generator=gen)).float() if with_add else None
    bias = (torch.randn(C, generator=gen) * 0.3).requires_grad_(True)
    alpha = (0.25 + 0.1 * torch.randn(C, generator=gen)).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    F.prelu((xr + (bias if with_bias else 0.0)).t().reshape(1, C, M), alpha).backward(dy.t().reshape(1, C, M))
    xd, dyd = bf(x).to(dev()), bf(dy).to(dev())
    addd = bf(add).to(dev()) if with_add else None
    rows = _C.lib().fedfr_bn_bwd_rows(M, C)
    part = torch.empty(rows, 3, C, device=dev()); coef = torch.empty(3, C, device=dev())
    db = torch.empty(C, device=dev()); da = torch.empty(C, device=dev()); dx = torch.empty(M, C, dtype=S16(), device=dev())
    bd, ad = bias.detach().to(dev()), alpha.detach().to(dev())
    _C.call("fedfr_bias_prelu_bwd", dyd.data_ptr(), xd.data_ptr(), bd.data_ptr() if with_bias else None, ad.data_ptr(), M, C, part.data_ptr(),
            coef.data_ptr(), db.data_ptr() if with_bias else None, da.data_ptr(), addd.data_ptr() if with_add else None, dx.data_ptr(), _C.stream())
    torch.cuda.synchronize()
    ref_dx = xr.grad + (add if with_add else 0.0)
    assert relerr(dx.float(), ref_dx) < 6e-3                    # bf16 output rounding
    assert relerr(da, alpha.grad) < 1e-4
    if with_bias:
        assert relerr(db, bias.grad) < 1e-4


def test_pad_input_nhwc():
    x = torch.randn(3, 3, 20, 20)
    out = torch.empty(3, 20, 20, 64, dtype=S16(), device=dev())
    xd = x.to(dev())
    _C.call("fedfr_pad_input_nhwc", xd.data_ptr(), out.data_ptr(), 3, 3, 400, 64, _C.stream())
    torch.cuda.synchronize()
    assert torch.equal(out[..., :3].cpu(), x.permute(0, 2, 3, 1).to(S16())) and float(out[..., 3:].float().abs().max()) == 0.0


@pytest.mark.parametrize("k", [1, 2, 3, 5, 8, 11])
def test_fedavg_multi_equals_sequential_axpy(k):
    """fedfr_fedavg_multi (one pass over up to 8 client states; FedPavg's flat path chains passes for more) is bit-identical to one
    fedfr_fedavg_axpy per client in ascending order = the reference loop server.py:27-33 (fp32 multiply, then fp32 add, per client);
    odd length exercises the scalar tail."""
    import numpy as np
    from fedfr_amd import server
    DEV = dev()
    n = 4 * 50_001 + 3
    g = torch.Generator().manual_seed(5 + k)
    srcs = [(torch.randn(n, generator=g) * (1 + i)).to(DEV) for i in range(k)]
    ws = [float(np.float32((1000.0 + 7 * i) / sum(1000.0 + 7 * j for j in range(k)))) for i in range(k)]
    ref = torch.empty(n, device=DEV)
    for i in range(k):
        _C.call("fedfr_fedavg_axpy", ref.data_ptr(), srcs[i].data_ptr(), ws[i], n, 1 if i else 0, _C.stream())
    out = torch.full((n,), float("nan"), device=DEV)
    for c0 in range(0, k, 8):
        server._multi(out, srcs[c0:c0 + 8], ws[c0:c0 + 8], c0 > 0)
    assert torch.equal(out, ref)
    cpu = torch.zeros(n)
    for i in range(k):
        cpu = cpu + torch.tensor(ws[i], dtype=torch.float32) * srcs[i].cpu()
    assert torch.equal(out.cpu(), cpu)                     # and to the reference's own torch expression
