"""Block-level parity of the HIP path, forward AND backward (reference backbones/iresnet.py:46-57):

* ``test_block_vs_reference``: the lone-block plan (``fedfr_block_create`` — the same C++ block code a whole network runs) against
  tests/golden/block.npz = y, dx, every parameter gradient and the BN buffers of the imported reference block;
* ``test_backward_layerwise_vs_bf16_oracle``: every block of a whole network's backward pass (gradient wrt the block input and all its
  parameter gradients) against the bf16-storage oracle's autograd, each block fed with the HIP block input and the HIP gradient
  entering it, so rounding flips do not compound.  The backward twin of test_forward_layerwise_vs_bf16_oracle (test_e2e_gpu.py).

Tolerances: on the product library (fp16 storage) everything in front of the PReLU kink — and, for the slope-1 fixtures, EVERYTHING — is asserted
at north_star's 1e-2 itself; the gradients behind the kink at the stated exception (tests/test_e2e_gpu.py docstring: a rounding of the PReLU's
input flips the derivative of the elements next to zero, relative L2 ~ sqrt(flipped fraction)).  The bf16 bounds (second argument of T16) serve
the child-process subset on libfedfr_hip_bf16.so (round 4's measurements x 1.25).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402
from oracle import bf16_emul as E  # noqa: E402

from fedfr_amd import backbones, _C  # noqa: E402
from fedfr_amd.backbones.iresnet import BlockPlan  # noqa: E402


@pytest.fixture(autouse=True)
def _oracle_models_the_loaded_librarys_storage():
    """oracle/bf16_emul.py rounds where the HIP path stores 16-bit tensors: to the loaded library's type (float16; bfloat16 under
    FEDFR_HIP_LIB_NAME=libfedfr_hip_bf16.so), back to bfloat16 for the CPU tests that may follow in the same session."""
    from oracle import bf16_emul
    bf16_emul.set_storage(_C.storage_dtype())
    yield
    bf16_emul.set_storage(torch.bfloat16)

DEV = torch.device("cuda:0")
SPEC = 1e-2                 # north_star: outputs within 1e-2 of the reference for 16-bit storage


def T16(fp16_bound, bf16_bound):
    """The bound an assertion uses on the loaded library: the product library stores fp16; libfedfr_hip_bf16.so runs the child-process subset."""
    return fp16_bound if _C.storage_dtype() == torch.float16 else bf16_bound


# gradients BEHIND the PReLU derivative in the block's backward pass (tests/test_oracle_golden.py:test_block_bf16_storage_floor)
POST_MASK = ("dx", "g_bn1.weight", "g_bn1.bias", "g_conv1.weight", "g_bn2.bias")


def T(a):
    return torch.from_numpy(np.asarray(a))


def srel(a, ref):
    """relative L2 error on the fixture's sample of a tensor."""
    a = R.fixture_sample(a.detach().cpu()).double()
    r = T(ref).double().reshape(-1)
    return float((a - r).norm() / (r.norm() + 1e-30))


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _run_block(name, dual_stream):
    cin, cout, stride, hw, batch, lin = R.BLOCK_FIXTURES[name]
    sd, x, dy, _ = R.block_fixture(name)
    plan = BlockPlan(cin, cout, stride, hw, batch, DEV)
    assert [t[0] for t in plan.table] == list(sd.keys())                 # reference block state_dict order
    plan.load_state_dict(sd)
    y = plan.forward(x.to(DEV), training=True)
    aux = torch.cuda.Stream(device=DEV) if dual_stream else None
    dx = plan.backward(dy.to(DEV), aux_stream=aux)
    torch.cuda.synchronize()
    return plan, y, dx


@pytest.mark.parametrize("dual_stream", [False, True])
@pytest.mark.parametrize("name", sorted(R.BLOCK_FIXTURES))
def test_block_vs_reference(name, dual_stream):
    """One IBasicBlock through the product path against the imported reference (fp32).  Everything in front of the PReLU kink — and,
    for the slope-1 fixtures, every output — is held to north_star's 1e-2.  The gradients behind the kink are held to what bf16
    storage allows (the emulator's own distance to the reference, measured 2-6e-2: see test_block_bf16_storage_floor) and, tightly,
    to the bf16-storage oracle itself."""
    g = load_golden("block")
    lin = R.BLOCK_FIXTURES[name][5]
    plan, y, dx = _run_block(name, dual_stream)
    grads = plan.grad_dict()
    gmax = max(float(g[name + "_gn_" + k]) for k in grads)
    errs = {"y": srel(y, g[name + "_y"]), "dx": srel(dx, g[name + "_dx"])}
    for k, v in grads.items():
        if float(g[name + "_gn_" + k]) > 1e-6 * gmax:                      # bn3 / downsample.1 bias: analytically zero
            errs["g_" + k] = srel(v, g[name + "_g_" + k])
            assert abs(float(v.double().norm()) - float(g[name + "_gn_" + k])) < 2e-2 * float(g[name + "_gn_" + k]), k
        else:
            assert float(v.abs().max()) < 1e-3 * gmax, k
    front = {k: e for k, e in errs.items() if k not in POST_MASK}
    behind = {k: e for k, e in errs.items() if k in POST_MASK}
    assert max(front.values()) < T16(SPEC, 7.5e-3), front        # fp16 measured <= 1.1e-3, bf16 <= 5.95e-3 on all seven fixtures; north_star: 1e-2
    # bn1.bias / bn2.bias gradients are column sums of a tensor whose channel means the BatchNorm behind it has just removed: the exact
    # value is a border effect of the 3x3 window, i.e. a nearly cancelling sum of bf16-rounded terms (measured 1.0e-2 on the slope-1 blocks)
    sums = {k: e for k, e in behind.items() if k in ("g_bn1.bias", "g_bn2.bias")}
    rest = {k: e for k, e in behind.items() if k not in sums}
    print("block %s dual=%d: front %.2e (%s); behind the PReLU kink %.2e (%s), cancelling sums %.2e" %
          (name, dual_stream, max(front.values()), max(front, key=front.get), max(rest.values()), max(rest, key=rest.get), max(sums.values())))
    # slope 1: inside north_star's 1e-2 like everything else (fp16: 6.5e-4, sums 1.4e-3).  With the kink: fp16 measured <= 1.7e-2 / sums 2.5e-2,
    # bf16 <= 3.6e-2 / 5.9e-2 — the exception class
    assert max(rest.values()) < (T16(SPEC, 7.5e-3) if lin else T16(3e-2, 4e-2)), rest
    assert max(sums.values()) < (T16(SPEC, 1.35e-2) if lin else T16(5e-2, 7.5e-2)), sums
    # BN buffers after one training forward (momentum 0.1, unbiased running variance) and the batch counters
    out = plan.state_dict()
    for k, v in out.items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(g[name + "_b_" + k]), k
        elif "running" in k:
            assert rel(v, T(g[name + "_b_" + k])) < 5e-3, (k, rel(v, T(g[name + "_b_" + k])))
    # the same block through the bf16-storage oracle (gradient rounding points included): accumulation order + rare ulp flips only
    ye, dxe, ge, _ = R.block_fixture_run(name, lambda sd, p, x, s, t: E.block(sd, p, E.q(x), s, t, grad_round=True))
    emu = {"y": rel(y, ye), "dx": rel(dx, dxe)}
    for k, v in grads.items():
        if float(g[name + "_gn_" + k]) > 1e-6 * gmax:
            emu["g_" + k] = rel(v, ge[k])
    esum = {k: e for k, e in emu.items() if k in ("g_bn1.bias", "g_bn2.bias")}       # the nearly cancelling column sums (see above)
    erest = {k: e for k, e in emu.items() if k not in esum}
    print("   vs bf16 oracle: worst %.2e %s; cancelling sums %.2e" % (max(erest.values()), max(erest, key=erest.get), max(esum.values())))
    assert max(erest.values()) < T16(2e-3, 6e-3), erest          # kernel vs storage-emulating oracle: fp16 measured <= 6.0e-4, bf16 <= 4.6e-3
    assert max(esum.values()) < T16(SPEC, 1.9e-2), esum          # fp16 measured <= 1.9e-3, bf16 <= 1.5e-2
    assert float(np.median(list(emu.values()))) < 4e-3, emu


def _nchw(a, B):
    rows, ch = a.shape
    h = int(round((rows // B) ** 0.5))
    return a.view(B, h, h, ch).permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("arch,batch", [("iresnet18", 8), ("iresnet50", 4), ("iresnet100", 6)])
def test_backward_layerwise_vs_bf16_oracle(arch, batch):
    """Whole-network backward, checked block by block: the HIP gradient leaving every block (= entering the block below it) and all of
    the block's parameter gradients against autograd of the bf16-storage oracle block evaluated on the HIP block input and the HIP
    gradient entering the block."""
    import ctypes as C
    layers = R.IRESNET_LAYERS[arch]
    m = getattr(backbones, arch)(False, dropout=0, fp16=True)
    sd = R.closed_form_state_dict(layers)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    x = R.closed_form_images(batch).to(DEV)
    dfeats = (R.hash_normal((batch, 512), 4242) * 0.05).to(DEV)
    plan = m._plan(batch)
    # capture buffer: gradient entering every block (last first) + gradient wrt the first block's input
    blocks = []
    hw, cin = 112, 64
    for si, nblk in enumerate(layers):
        for bi in range(nblk):
            stride, cout = (2 if bi == 0 else 1), (64, 128, 256, 512)[si]
            blocks.append(("layer%d.%d" % (si + 1, bi), cin, cout, stride, hw))
            hw, cin = hw // stride, cout
    sizes = [batch * (b[4] // b[3]) ** 2 * b[2] for b in blocks]
    total = sum(sizes) + batch * 112 * 112 * 64
    cap = torch.zeros(total, dtype=_C.storage_dtype(), device=DEV)
    _C.call("fedfr_net_debug_capture", cap.data_ptr(), cap.numel())
    try:
        feats = m(x)
        feats.backward(dfeats)
        torch.cuda.synchronize()
    finally:
        _C.call("fedfr_net_debug_capture", None, 0)
    # slice the capture: order = last block first
    gin, off = {}, 0
    for bi in range(len(blocks) - 1, -1, -1):
        name, ci, co, stride, h = blocks[bi]
        ho = h // stride
        gin[bi] = _nchw(cap[off: off + sizes[bi]].float().view(batch * ho * ho, co), batch).cpu()
        off += sizes[bi]
    gstem = _nchw(cap[off: off + batch * 112 * 112 * 64].float().view(batch * 112 * 112, 64), batch).cpu()

    def act(bi, which):
        o, rows, ch = C.c_longlong(), C.c_int(), C.c_int()
        _C.call("fedfr_net_act_info", plan.handle, bi, which, C.byref(o), C.byref(rows), C.byref(ch))
        a = plan.act[o.value * 2: (o.value + rows.value * ch.value) * 2].view(_C.storage_dtype()).view(rows.value, ch.value)
        return _nchw(a.float(), batch).cpu()

    params = dict(m.named_parameters())
    # the captured activation gradients carry the library's static loss scale (1 for bf16 storage, 256 for the fp16 build: _C.loss_scale());
    # the oracle block is run on them as captured, so its parameter gradients carry it too — the module's p.grad are already unscaled
    S = _C.loss_scale()
    errs = []
    for bi, (name, ci, co, stride, h) in enumerate(blocks):
        bsd = {k: v.clone() for k, v in sd.items() if k.startswith(name + ".")}
        pk = [k for k, v in bsd.items() if v.dtype.is_floating_point and "running" not in k]
        for k in pk:
            bsd[k].requires_grad_(True)
        xin = act(bi, 0).requires_grad_(True)
        yb = E.block(bsd, name, xin, stride, True, grad_round=True)
        yb.backward(gin[bi])
        dx_hip = gin[bi - 1] if bi > 0 else gstem
        errs.append((name + ".dx", rel(dx_hip, xin.grad)))
        scale = max(float(bsd[k].grad.norm()) for k in pk)
        for k in pk:
            if float(bsd[k].grad.norm()) < 1e-4 * scale:                 # biases in front of a BatchNorm: analytically zero
                continue
            errs.append((k, rel(params[k].grad * S, bsd[k].grad)))
    # bn1.bias / bn2.bias gradients are column sums of tensors whose channel means a BatchNorm backward has just removed (the exact
    # value is a border effect of the 3x3 window): nearly cancelling sums of bf16-rounded terms, held to a looser bar
    sums = [e for e in errs if e[0].endswith(("bn1.bias", "bn2.bias"))]
    rest = [e for e in errs if not e[0].endswith(("bn1.bias", "bn2.bias"))]
    worst, worst_sum = max(rest, key=lambda e: e[1]), max(sums, key=lambda e: e[1])
    vals = np.array([e for _, e in errs])
    print("layerwise bwd %s: worst %.2e (%s) cancelling sums %.2e (%s) median %.2e p90 %.2e" %
          (arch, worst[1], worst[0], worst_sum[1], worst_sum[0], np.median(vals), np.percentile(vals, 90)))
    print("   top: " + ", ".join("%s %.2e" % e for e in sorted(rest, key=lambda e: -e[1])[:6]))
    # measured 2.4e-3 / 5.7e-3 / 3.7e-3 (iresnet18 / 50 / 100; 2.3e-3 / 2.7e-3 / 4.3e-3 with option bn_sliced=0).  The outliers come in pairs
    # (bn1.weight, conv1.weight) of ONE block: both read the gradient behind that block's PReLU, where a bf16 rounding that lands on the
    # other side of the kink changes the derivative of an element by (1 - slope); which block draws the outlier moves with the last
    # fp32 bits of the BatchNorm coefficients (summation order), its size does not
    # fp16 (product library): worst 8.6e-4 / 2.4e-3 / 5.9e-3, sums <= 8.6e-3, median 1.9e-4: everything inside north_star's 1e-2
    # (round 5, statistics on the matrix cores = another summation order: the bf16 build's outlier moved to layer4.1.bn1.weight at 7.3e-3 —
    # the bf16 bound follows it, still inside the 1e-2 the product library is held to)
    assert worst[1] < T16(SPEC, 8.5e-3), worst
    assert worst_sum[1] < T16(SPEC, 2.5e-2), worst_sum  # bf16 measured <= 1.9e-2
    assert np.median(vals) < T16(5e-4, 1.3e-3), np.median(vals)    # fp16 measured 1.9e-4; bf16 0.5e-3 / 1.0e-3 / 0.9e-3
