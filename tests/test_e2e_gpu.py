"""End-to-end parity of the HIP path (through the reference-shaped Python surface) against the golden vectors
captured from the imported reference and, where finer detail is needed, the CPU oracle on identical
closed-form inputs.

Tolerances.  The product library (libfedfr_hip.so) stores IEEE fp16 — the reference's own AMP type, backbones/iresnet.py:159.  On it every
whole-network OUTPUT (embeddings, cosine logits, loss, running statistics, gradient norms) is asserted at north_star's 1e-2 itself (SPEC), not
at a measured value.  One class cannot meet 1e-2 on any 16-bit storage and is asserted at a stated exception (KINK_*): the per-parameter
DIRECTION of gradients that passed a PReLU.  Rounding the PReLU's input moves elements across zero; each flipped element changes its
derivative from 1 to the slope, so a flipped fraction f of the elements changes the gradient by ~ (1 - slope) sqrt(f) in relative L2 —
sqrt(1e-3) = 3e-2 for fp16's 2^-11, sqrt(8e-3) = 9e-2 for bf16's 2^-8, 8e-4 for the fp32 validation path (measured: 2.7-3.5e-2, 8.5e-2-1.2e-1,
2.5-8e-4).  The same kernels with slope 1 (no kink) are inside 1e-2 on every gradient (test_gradients_without_the_prelu_kink_meet_the_spec,
tests/test_block_gpu.py `*_lin`), and on the REAL networks with their real slopes the law is checked element by element
(test_prelu_kink_law_and_mask_injected_gradients: flipped inputs counted per PReLU, every parameter's error <= 1.5 x the prediction; with the HIP
path's sign pattern injected into the fp32 oracle's backward pass every gradient norm and direction is inside 1e-2).  T16(fp16_bound, bf16_bound) picks the bound of the loaded library; the bf16 bounds (what 7 mantissa bits
allow, measured in round 4) are used by the child-process subset that loads libfedfr_hip_bf16.so."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from conftest import load_golden  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402

import fedfr_amd  # noqa: E402
from fedfr_amd import backbones, losses, client, server, ops, _C  # noqa: E402
from fedfr_amd.partial_fc import PartialFC  # noqa: E402

DEV = torch.device("cuda:0")
SPEC = 1e-2                 # north_star: embeddings, logits, grads within 1e-2 of the reference for 16-bit storage
# stated exceptions on fp16 storage (see the module docstring): per-parameter gradient direction behind a PReLU kink
# (median / worst tensor over a network), worst single parameter tensor's gradient NORM (a bn3 scale: a sum of kink-affected products)
# = measured x 1.25 (iresnet100 b6: median 3.87e-2, max 6.24e-2, worst norm 1.63e-2).  The LAW behind them is an assertion of its own:
# test_prelu_kink_law_and_mask_injected_gradients counts the flipped PReLU inputs and bounds every parameter's error by the prediction, and shows
# every gradient inside 1e-2 (measured 6.8e-3) of the fp32 oracle once both sides differentiate with the same sign pattern
KINK_DIR_MEDIAN, KINK_DIR_MAX, KINK_NORM_MAX = 4.9e-2, 8e-2, 2.1e-2


def T16(fp16_bound, bf16_bound):
    """The bound an assertion uses on the loaded library: the product library stores fp16; libfedfr_hip_bf16.so runs the child-process subset."""
    return fp16_bound if _C.storage_dtype() == torch.float16 else bf16_bound



@pytest.fixture(autouse=True)
def _oracle_models_the_loaded_librarys_storage():
    """oracle/bf16_emul.py rounds where the HIP path stores 16-bit tensors: to the loaded library's type (float16; bfloat16 under
    FEDFR_HIP_LIB_NAME=libfedfr_hip_bf16.so), back to bfloat16 for the CPU tests that may follow in the same session."""
    from oracle import bf16_emul
    bf16_emul.set_storage(_C.storage_dtype())
    yield
    bf16_emul.set_storage(torch.bfloat16)

def T(a):
    return torch.from_numpy(np.asarray(a))


def rel(a, b):
    a, b = a.detach().double().cpu(), T(b).double() if not isinstance(b, torch.Tensor) else b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def maxrel(a, b):
    a, b = a.detach().double().cpu(), T(b).double() if not isinstance(b, torch.Tensor) else b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def make_model(arch, tag=0.0):
    layers = R.IRESNET_LAYERS[arch]
    m = getattr(backbones, arch)(False, dropout=0, fp16=True)
    sd = R.closed_form_state_dict(layers, tag=tag)
    assert list(m.state_dict().keys()) == list(sd.keys())          # reference key order
    m.load_state_dict(sd)
    return m.to(DEV), sd, layers


def test_state_dict_roundtrip_and_layout():
    m, sd, layers = make_model("iresnet18")
    out = m.state_dict()
    for k, v in sd.items():
        assert out[k].shape == v.shape and out[k].dtype == v.dtype, k
        assert torch.equal(out[k].cpu(), v), k
    assert not m.features.weight.requires_grad                    # iresnet.py:99-100
    n_train = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert n_train == sum(v.numel() for k, v in sd.items() if k in R.trainable_keys(sd))
    m2 = getattr(backbones, "iresnet18")().to(DEV)
    m2.load_state_dict({k: v.to(DEV) for k, v in out.items()})
    assert torch.equal(m2._flat_params, m._flat_params)


def _hip_act(plan, block, which):
    import ctypes as C
    off, rows, ch = C.c_longlong(), C.c_int(), C.c_int()
    _C.call("fedfr_net_act_info", plan.handle, block, which, C.byref(off), C.byref(rows), C.byref(ch))
    if off.value < 0:
        return None
    a = plan.act[off.value * 2: (off.value + rows.value * ch.value) * 2].view(_C.storage_dtype()).view(rows.value, ch.value)
    return a.float().cpu()


@pytest.mark.parametrize("arch,batch,training", [("iresnet18", 8, False), ("iresnet18", 8, True), ("iresnet50", 4, True),
                                                 ("iresnet100", 6, True)])
def test_forward_layerwise_vs_bf16_oracle(arch, batch, training):
    """Every saved activation of the HIP forward against the bf16-storage oracle (oracle/bf16_emul.py), block by
    block, each block fed with the HIP block input so rounding flips do not compound.  Differences are
    accumulation order + rare single-ulp bf16 flips: <= 1e-2 per tensor (measured <= 6e-3, typically 5e-4)."""
    from oracle import bf16_emul as E
    m, sd, layers = make_model(arch)
    x = R.closed_form_images(batch)
    m.train(training)
    _C.call("fedfr_set_option", b"eval_fuse", 0)       # every activation is inspected: the fused eval epilogues do not store c1 / c2
    try:
        with torch.no_grad():
            f = m._run_forward(x.to(DEV), training=training)
        torch.cuda.synchronize()
    finally:
        _C.call("fedfr_set_option", b"eval_fuse", 1)
    plan = m._plan(batch)
    nhwc = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])    # noqa: E731
    sdc = {k: v.clone() for k, v in sd.items()}
    q = E.q
    errs = []
    with torch.no_grad():
        c0 = q(F.conv2d(q(x), q(sdc["conv1.weight"]), None, 1, 1))
        h = q(F.prelu(R._bn(sdc, "bn1", c0, training), sdc["prelu.weight"]))
        errs += [("stem.c0", rel(_hip_act(plan, -1, 0), nhwc(c0))), ("stem.a0", rel(_hip_act(plan, -1, 1), nhwc(h)))]
        bi_g = 0
        for si, nblk in enumerate(layers):
            for bi in range(nblk):
                p, stride = "layer%d.%d" % (si + 1, bi), (2 if bi == 0 else 1)
                a1 = q(R._bn(sdc, p + ".bn1", h, training))
                c1 = q(F.conv2d(a1, q(sdc[p + ".conv1.weight"]), None, 1, 1))
                a2 = q(F.prelu(R._bn(sdc, p + ".bn2", c1, training), sdc[p + ".prelu.weight"]))
                c2 = q(F.conv2d(a2, q(sdc[p + ".conv2.weight"]), None, stride, 1))
                out = R._bn(sdc, p + ".bn3", c2, training)
                if (p + ".downsample.0.weight") in sdc:
                    d = q(F.conv2d(h, q(sdc[p + ".downsample.0.weight"]), None, stride, 0))
                    idn = R._bn(sdc, p + ".downsample.1", d, training)
                    errs.append((p + ".d", rel(_hip_act(plan, bi_g, 5), nhwc(d))))
                else:
                    idn = h
                o = q(out + idn)
                for nm, which, ref in (("a1", 1, a1), ("c1", 2, c1), ("a2", 3, a2), ("c2", 4, c2), ("out", 6, o)):
                    errs.append((p + "." + nm, rel(_hip_act(plan, bi_g, which), nhwc(ref))))
                ho = _hip_act(plan, bi_g, 6)
                h = ho.view(o.shape[0], o.shape[2], o.shape[3], o.shape[1]).permute(0, 3, 1, 2).contiguous()
                bi_g += 1
        t = q(R._bn(sdc, "bn2", h, training))
        errs.append(("tail.t", rel(_hip_act(plan, -1, 2), torch.flatten(t, 1))))
        y = F.linear(torch.flatten(t, 1), q(sdc["fc.weight"]), sdc["fc.bias"])
        fe = R._bn(sdc, "features", y, training)
        errs.append(("feats", rel(f, fe)))
    worst = max(errs, key=lambda e: e[1])
    print("layerwise fwd %s: worst %.2e (%s) median %.2e feats %.2e" % (arch, worst[1], worst[0], float(np.median([e for _, e in errs])),
                                                                       dict(errs)["feats"]))
    assert worst[1] < 1e-2, worst
    assert float(np.median([e for _, e in errs])) < 2e-3
    assert dict(errs)["feats"] < 2e-3, dict(errs)["feats"]


# fp16 (product library): north_star's 1e-2 for every whole-network output (measured: embeddings 1.5e-3 / 2.2e-3 (iresnet50 eval / train),
# 2.0e-3 / 3.1e-3 (iresnet100); running statistics 2e-4 early layers, 1.3-1.9e-3 bn2 / features).
# bf16 (libfedfr_hip_bf16.so, child-process subset): 7 mantissa bits do not reach 1e-2 on 50-100 layers of train-mode BatchNorm
# (tools/precision_study.py); bounds = round 4's measurements (1.17e-2 / 1.71e-2, 1.55e-2 / 2.59e-2; statistics 1.7e-3 / 1.5e-2) x 1.25
EMB_TOL_BF16 = {"iresnet50": (1.5e-2, 2.2e-2), "iresnet100": (2e-2, 3.3e-2)}
STAT_TOL_BF16 = (2.5e-3, 2e-2)      # early layers, bn2 / features


@pytest.mark.parametrize("arch,batch,fname", [("iresnet50", 8, "r50_b8"), ("iresnet100", 6, "r100_b6")])
def test_backbone_forward_vs_reference(arch, batch, fname):
    """Embeddings (eval- and train-mode BatchNorm) and running statistics against the imported fp32 reference: north_star's 1e-2 on the
    product (fp16-storage) library.  (bf16 STORAGE alone moves the embeddings of these 50/100-layer train-mode-BN nets by 1.2e-2 / 1.6e-2 —
    oracle/bf16_emul.py vs the fp32 oracle, same inputs — which is where the bf16 build sits.)"""
    g = load_golden(fname)
    m, sd, layers = make_model(arch)
    x = R.closed_form_images(batch).to(DEV)
    m.eval()
    with torch.no_grad():
        fe = m(x)
    m.train()
    ft = m(x)
    print("MEASURED %s embeddings: eval %.3e train %.3e" % (arch, rel(fe, g["feat_eval"]), rel(ft, g["feat_train"])))
    lim_e, lim_t = T16((SPEC, SPEC), EMB_TOL_BF16[arch])
    assert rel(fe, g["feat_eval"]) < lim_e, rel(fe, g["feat_eval"])
    assert rel(ft, g["feat_train"]) < lim_t, rel(ft, g["feat_train"])
    # running stats: reference momentum / unbiased-variance rule (fp32 statistics of bf16 tensors)
    sd_out = m.state_dict()
    worst_early, worst_late = 0.0, 0.0
    for k in ("bn1", "layer1.0.bn1", "layer2.0.downsample.1", "layer4.2.bn3", "bn2", "features"):
        e = max(rel(sd_out[k + ".running_mean"], g["rm_" + k]), rel(sd_out[k + ".running_var"], g["rv_" + k]))
        if k in ("bn2", "features"):
            worst_late = max(worst_late, e)
        else:
            worst_early = max(worst_early, e)
    print("MEASURED %s running stats: early layers %.3e, bn2/features %.3e" % (arch, worst_early, worst_late))
    for k in ("bn1", "layer1.0.bn1", "layer2.0.downsample.1", "layer4.2.bn3", "bn2", "features"):
        tol = T16((SPEC, SPEC), STAT_TOL_BF16)[1 if k in ("bn2", "features") else 0]     # (bf16: late layers carry the accumulated storage noise)
        assert rel(sd_out[k + ".running_mean"], g["rm_" + k]) < tol, (k, rel(sd_out[k + ".running_mean"], g["rm_" + k]))
        assert rel(sd_out[k + ".running_var"], g["rv_" + k]) < tol, (k, rel(sd_out[k + ".running_var"], g["rv_" + k]))
        assert int(sd_out[k + ".num_batches_tracked"]) == int(g["nbt_" + k])


@pytest.mark.parametrize("arch,batch,fname", [("iresnet50", 8, "r50_b8"), ("iresnet100", 6, "r100_b6")])
def test_fp32_validation_path_vs_reference(arch, batch, fname):
    """north_star's fp32 clause (outputs within 1e-3 of the reference PyTorch CPU path): the SAME plan, parameters and buffers run through
    the fp32 validation path (IResNet.validation_fp32 = True -> csrc/net_f32.hip: fp32 activations, exact-fp32 MFMA GEMMs, BatchNorm
    statistics in fp64) — eval and train embeddings, the reference-style eager step (client.py:543-549) with every gradient, running
    statistics.  Whatever the bf16 product path shows beyond these numbers (tests above) is bf16 storage, not the algorithm."""
    g = load_golden(fname)
    C = int(g["num_classes"])
    m, sd, layers = make_model(arch)
    m.validation_fp32 = True
    x = R.closed_form_images(batch).to(DEV)
    lab = R.closed_form_labels(batch, C).to(DEV)
    m.eval()
    with torch.no_grad():
        fe = m(x)
    m.train()
    fcm = client.FC_module(512, C, "/tmp").to(DEV)
    fcm.fc.data = R.head_fc(C).to(DEV)
    model = client.Sequential_model(m, fcm)
    cosine = model(x)
    feats_err = None
    logits = losses.CosFace(s=30, m=0.4)(cosine, lab)
    loss = ops.cross_entropy(logits, lab)
    loss.backward()
    e_eval, e_cos = rel(fe, g["feat_eval"]), rel(cosine, g["cosine"])
    names = [str(n) for n in g["grad_names"]]
    params = dict(m.named_parameters())
    norms = np.array([float(params[k].grad.norm()) for k in names])
    ref = g["grad_norms"]
    big = ref > 1e-6 * ref.max()
    nerr = np.abs(norms[big] - ref[big]) / ref[big]
    gmax = max(float(T(g[k]).double().norm()) for k in g.files if k.startswith("g_") and k[2:] in params)
    dirs = []
    for k in g.files:
        if k.startswith("g_") and k[2:] in params:
            r = T(g[k]).double()
            if float(r.norm()) < 1e-3 * gmax:          # biases in front of a BatchNorm: analytically zero, pure rounding noise in any precision
                continue
            dirs.append((k[2:], rel(params[k[2:]].grad.reshape(r.shape), r)))
    dirs.append(("layer3.1.conv1.weight[:4,:16]", rel(params["layer3.1.conv1.weight"].grad[:4, :16], g["g_layer3.1.conv1.weight_slice"])))
    dirs.append(("fc.weight[:4,:2048]", rel(params["fc.weight"].grad[:4, :2048], g["g_fc.weight_slice"])))
    dirs.append(("head fc[:8]", rel(fcm.fc.grad[:8], g["g_fc_head_rows"])))
    # bn1.bias / bn2.bias gradients are column sums that cancel up to a border effect of the 3x3 window (test_block_gpu.py): their relative
    # error is fp32 summation-order noise of the cancelled part, in the reference as much as here (the CPU oracle meets the same goldens at 2e-3)
    sums = [d for d in dirs if d[0].endswith(("bn1.bias", "bn2.bias"))]
    rest = [d for d in dirs if not d[0].endswith(("bn1.bias", "bn2.bias"))]
    vals, svals = np.array([d for _, d in rest]), np.array([d for _, d in sums])
    sd_out = m.state_dict()
    stat = max(max(rel(sd_out[k + ".running_mean"], g["rm_" + k]), rel(sd_out[k + ".running_var"], g["rv_" + k]))
               for k in ("bn1", "layer1.0.bn1", "layer2.0.downsample.1", "layer4.2.bn3", "bn2", "features"))
    print("MEASURED fp32 path %s: eval embeddings %.2e, train cosines %.2e, loss %.2e; grad norms median %.2e max %.2e; directions median %.2e "
          "max %.2e (%s), cancelling sums max %.2e (%s); running stats %.2e" %
          (arch, e_eval, e_cos, abs(float(loss) - float(g["loss"])) / abs(float(g["loss"])), np.median(nerr), nerr.max(), np.median(vals), vals.max(),
           max(rest, key=lambda d: d[1])[0], svals.max(), max(sums, key=lambda d: d[1])[0], stat))
    assert e_eval < 1e-3 and e_cos < 1e-3, (e_eval, e_cos)          # measured 1e-6
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))      # measured: equal to the last printed digit
    assert nerr.max() < 1e-3, (nerr.max(), names[int(np.argmax(nerr))])      # measured: median 3e-5, max 7e-4
    assert stat < 1e-3, stat
    # Gradient DIRECTIONS: 8e-4 median, 1.1e-3 / 1.8e-3 at worst — not arithmetic noise (the GEMMs accumulate in fp64; the numbers did not move
    # by a digit when they were switched from fp32 accumulation) but PReLU derivative flips: of the ~10^7 PReLU inputs of a pass a handful
    # lie within an ulp of the kink, and two fp32 evaluations put one of them on different sides (here: one element of layer3.9, which moves
    # that layer's parameter gradients by 1e-3 and everything upstream of it by 1e-4).  The check that isolates the arithmetic is below.
    assert vals.max() < 2.5e-3, max(rest, key=lambda d: d[1])
    assert svals.max() < 5e-3, max(sums, key=lambda d: d[1])
    if arch != "iresnet50":
        return
    # ---- the same network with every PReLU slope set to 1 (no kink): forward + backward of sum(feats * w) against the fp64 evaluation of the
    # oracle, next to the fp32 evaluation of the same oracle (= what the reference computes)
    sd1 = {k: (torch.ones_like(v) if k.endswith("prelu.weight") else v.clone()) for k, v in R.closed_form_state_dict(layers).items()}
    m.load_state_dict(sd1)
    m.train()
    w = R.closed_form((batch, 512), 0.37, 0.9, 1.0)
    f = m(x)
    for p_ in m.parameters():
        p_.grad = None
    (f * w.to(DEV)).sum().backward()
    hip = {k: p_.grad.detach().double().cpu() for k, p_ in m.named_parameters() if p_.grad is not None}
    keys = R.trainable_keys(sd1)
    ev = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        work = {k: (v.to(dt).clone().requires_grad_(True) if k in keys else (v.to(dt).clone() if v.dtype.is_floating_point else v.clone()))
                for k, v in sd1.items()}
        ff = R.iresnet_forward(work, R.closed_form_images(batch).to(dt), layers, training=True)
        (ff * w.to(dt)).sum().backward()
        ev[name] = ({k: work[k].grad.double() for k in keys}, ff.detach().double())
    e_hip = [(rel(hip[k], ev["f64"][0][k]), k) for k in keys if float(ev["f64"][0][k].norm()) > 1e-9]
    e_ref = [(rel(ev["f32"][0][k], ev["f64"][0][k]), k) for k in keys if float(ev["f64"][0][k].norm()) > 1e-9]
    print("   slope-1 net vs the fp64 evaluation: embeddings HIP %.2e / fp32 oracle %.2e; gradients median %.2e / %.2e, worst %.2e (%s) / %.2e (%s)" %
          ((rel(f.double().cpu(), ev["f64"][1]), rel(ev["f32"][1], ev["f64"][1]), np.median([e for e, _ in e_hip]), np.median([e for e, _ in e_ref]))
           + max(e_hip) + max(e_ref)))
    assert rel(f.double().cpu(), ev["f64"][1]) < 1e-5
    assert np.median([e for e, _ in e_hip]) < 1e-5, np.median([e for e, _ in e_hip])           # measured 4.4e-7 (fp32 oracle: 1.5e-6)
    assert max(e_hip)[0] < max(1e-3, 2.0 * max(e_ref)[0]), (max(e_hip), max(e_ref))              # measured 9.9e-5 (fp32 oracle: 1.1e-4): cancelling sums


def test_freeze_bn_vs_reference():
    """IResNet.freeze_BN(test_mode=True) (iresnet.py:140-147) through the product path: every BatchNorm normalises with its running
    statistics inside a TRAINING net (fedfr_net_forward training = 2 + the BatchNorm-backward passes with the mean terms switched off),
    against the imported reference (tests/golden/freeze_bn_r18.npz).  Nothing is tracked; model.train() undoes it, as in the reference."""
    g = load_golden("freeze_bn_r18")
    B = int(g["B"])
    m, sd, layers = make_model("iresnet18", tag=7.0)
    x = R.closed_form_images(B, tag=3.0).to(DEV)
    w = R.closed_form((B, 512), 0.37, 0.9, 1.0).to(DEV)
    m.train()
    m.freeze_BN()
    assert m._fwd_mode() == 2 and not m.bn1.training and m.training
    feats = m(x)
    (feats * w).sum().backward()
    e_f = rel(feats, g["feats"])
    names = [str(n) for n in g["grad_names"]]
    params = dict(m.named_parameters())
    norms = np.array([float(params[k].grad.norm()) for k in names])
    nerr = np.abs(norms - g["grad_norms"]) / (g["grad_norms"] + 1e-12)
    dirs = [(k[2:], rel(params[k[2:]].grad, g[k])) for k in g.files if k.startswith("g_") and k[2:] in params]
    dirs.append(("layer3.1.conv1.weight[:4,:16]", rel(params["layer3.1.conv1.weight"].grad[:4, :16], g["g_layer3.1.conv1.weight_slice"])))
    dirs.append(("fc.weight[:4,:2048]", rel(params["fc.weight"].grad[:4, :2048], g["g_fc.weight_slice"])))
    vals = np.array([d for _, d in dirs])
    print("MEASURED freeze_BN iresnet18: embeddings %.3e; grad norms median %.3e max %.3e; directions median %.3e max %.3e (%s)" %
          (e_f, np.median(nerr), nerr.max(), np.median(vals), vals.max(), max(dirs, key=lambda d: d[1])[0]))
    # fp16: embeddings 1.0e-3, gradient norms median 6e-4 / max 4.9e-3 (all at the spec); directions median 1.1e-2 / max 2.2e-2 = the PReLU-kink class
    # bf16 subset: embeddings 8.6e-3, norms 2.0e-3 / 2.4e-2, directions 3.5e-2 / 5.8e-2 (round 4, x 1.25)
    assert e_f < SPEC, e_f
    assert np.median(nerr) < T16(SPEC, 2.5e-3) and nerr.max() < T16(SPEC, 3.1e-2), (np.median(nerr), nerr.max())
    assert np.median(vals) < T16(KINK_DIR_MEDIAN, 4.4e-2) and vals.max() < T16(KINK_DIR_MAX, 7.3e-2), max(dirs, key=lambda d: d[1])
    out = m.state_dict()
    for k in ("bn1", "layer2.0.downsample.1", "layer4.1.bn3", "bn2", "features"):      # nothing tracked
        assert torch.equal(out[k + ".running_mean"].cpu(), sd[k + ".running_mean"]) and torch.equal(out[k + ".running_var"].cpu(), sd[k + ".running_var"])
        assert int(out[k + ".num_batches_tracked"]) == int(g["nbt_" + k])
    m.train()                                   # nn.Module.train() resets the BatchNorm submodules
    assert m._fwd_mode() == 1 and m.bn1.training
    again = m(x)
    assert rel(again, g["feats_after_train_call"]) < T16(SPEC, 2.5e-2)
    assert int(m.state_dict()["bn1.num_batches_tracked"]) == int(g["nbt_bn1"]) + 1


# whole-network gradients against the fp32 reference.
# fp16 (product library): cosine logits, loss and gradient NORMS at north_star's 1e-2 (measured: cosines 2.2e-3 / 3.1e-3; norms median 9e-4 / 1.2e-3);
# the worst single tensor's norm (a bn3 scale, 1.0e-2 / 1.9e-2) and the per-parameter DIRECTIONS (median 2.7e-2 / 3.5e-2, max 6.1e-2 / 6.5e-2) are
# the PReLU-kink class (module docstring): stated exceptions KINK_*, not kernel error — per block the backward pass is within 6e-3 of the
# storage-emulating oracle (test_block_gpu.py), and with slope 1 the whole network is inside 1e-2 (next test but one).
# bf16 (libfedfr_hip_bf16.so subset): round 4's measurements x 1.25 (cosines 1.7e-2 / 2.6e-2; norms 3.2e-3 / 4.2e-3, max 6.7e-2 / 4.7e-2; directions
# median 8.5e-2 / 1.18e-1, max 0.175 / 0.33)
GRAD_TOL_BF16 = {"iresnet50": {"cosine": 2.2e-2, "norm_median": 4.5e-3, "norm_max": 8.5e-2, "dir_median": 0.11, "dir_max": 0.22},
                 "iresnet100": {"cosine": 3.3e-2, "norm_median": 5.5e-3, "norm_max": 6e-2, "dir_median": 0.15, "dir_max": 0.42}}
GRAD_TOL_FP16 = {"cosine": SPEC, "norm_median": SPEC, "norm_max": KINK_NORM_MAX, "dir_median": KINK_DIR_MEDIAN, "dir_max": KINK_DIR_MAX}


def grad_tol(arch):
    return T16(GRAD_TOL_FP16, GRAD_TOL_BF16[arch])


@pytest.mark.parametrize("arch,batch,fname", [("iresnet50", 8, "r50_b8"), ("iresnet100", 6, "r100_b6")])
def test_train_step_grads_vs_reference(arch, batch, fname):
    """reference-style eager step (client.py:543-549): logits = fc(backbone(x)); margin; cross-entropy; backward — cosine logits, loss,
    every parameter gradient's norm and direction against the imported reference (bounds: grad_tol above)."""
    g = load_golden(fname)
    C = int(g["num_classes"])
    m, sd, layers = make_model(arch)
    m.train()
    fcm = client.FC_module(512, C, "/tmp").to(DEV)
    fcm.fc.data = R.head_fc(C).to(DEV)
    x = R.closed_form_images(batch).to(DEV)
    lab = R.closed_form_labels(batch, C).to(DEV)
    model = client.Sequential_model(m, fcm)
    cosine = model(x)
    logits = losses.CosFace(s=30, m=0.4)(cosine, lab)
    loss = ops.cross_entropy(logits, lab)
    loss.backward()
    print("MEASURED %s cosine %.3e loss %.3e" % (arch, rel(cosine, g["cosine"]), abs(float(loss) - float(g["loss"])) / abs(float(g["loss"]))))
    GT = grad_tol(arch)
    assert rel(cosine, g["cosine"]) < GT["cosine"], rel(cosine, g["cosine"])
    assert abs(float(loss) - float(g["loss"])) < 5e-3 * abs(float(g["loss"]))
    names = [str(n) for n in g["grad_names"]]
    params = dict(m.named_parameters())
    norms = np.array([float(params[k].grad.norm()) for k in names])
    ref = g["grad_norms"]
    big = ref > 1e-6 * ref.max()
    relerr = np.abs(norms[big] - ref[big]) / ref[big]
    print("MEASURED %s grad norms: median %.3e max %.3e (%s)" % (arch, np.median(relerr), relerr.max(), names[int(np.argmax(relerr))]))
    assert np.median(relerr) < GT["norm_median"], np.median(relerr)
    assert relerr.max() < GT["norm_max"], (relerr.max(), names[int(np.argmax(relerr))])
    dirs = []
    gmax = max(float(T(g[k]).double().norm()) for k in g.files if k.startswith("g_") and k[2:] in params)
    for k in g.files:
        if k.startswith("g_") and k[2:] in params and params[k[2:]].grad is not None:
            r = T(g[k]).double()
            if float(r.norm()) < 1e-3 * gmax:  # biases in front of a BatchNorm (bn3 / downsample.1 / fc.bias): analytically zero
                continue
            dirs.append((k, rel(params[k[2:]].grad.reshape(r.shape), r)))
    dirs.append(("layer3.1.conv1.weight[:4,:16]", rel(params["layer3.1.conv1.weight"].grad[:4, :16], g["g_layer3.1.conv1.weight_slice"])))
    dirs.append(("fc.weight[:4,:2048]", rel(params["fc.weight"].grad[:4, :2048], g["g_fc.weight_slice"])))
    dirs.append(("head fc[:8]", rel(fcm.fc.grad[:8], g["g_fc_head_rows"])))
    vals = np.array([d for _, d in dirs])
    print("MEASURED %s grad directions: median %.3e max %.3e (%s)" % (arch, np.median(vals), vals.max(), max(dirs, key=lambda d: d[1])[0]))
    assert np.median(vals) < GT["dir_median"], np.median(vals)
    assert vals.max() < GT["dir_max"], max(dirs, key=lambda d: d[1])


@pytest.mark.parametrize("arch,batch", [("iresnet50", 8), ("iresnet100", 6)])
def test_gradients_without_the_prelu_kink_meet_the_spec(arch, batch):
    """The exception above is the kink, not the kernels: the SAME networks with every PReLU slope set to 1 (the activation becomes the
    identity, nothing else changes: same kernels, same BatchNorm statistics, same 16-bit storage) — one eager train step against the fp32
    oracle on the same inputs (oracle/ref_cpu.py restates the reference and is pinned to it by tests/golden): cosine logits, loss, every
    parameter gradient's norm AND direction inside north_star's 1e-2 on the product library (the nearly cancelling BatchNorm-bias column sums
    are listed apart, as everywhere)."""
    C = 32
    m, sd, layers = make_model(arch)
    for k in sd:
        if k.endswith("prelu.weight"):
            sd[k] = torch.ones_like(sd[k])
    m.load_state_dict(sd)
    m.train()
    fcm = client.FC_module(512, C, "/tmp").to(DEV)
    fc0 = R.head_fc(C)
    fcm.fc.data = fc0.clone().to(DEV)
    x, lab = R.closed_form_images(batch), R.closed_form_labels(batch, C)
    cosine = client.Sequential_model(m, fcm)(x.to(DEV))
    loss = ops.cross_entropy(losses.CosFace(s=30, m=0.4)(cosine, lab.to(DEV)), lab.to(DEV))
    loss.backward()
    f_ref, c_ref, l_ref, g_ref, fcg_ref = R.train_step_grads(sd, fc0.clone(), x, lab, layers)
    params = dict(m.named_parameters())
    gmax = max(float(v.norm()) for v in g_ref.values())
    names = [k for k in R.trainable_keys(sd) if float(g_ref[k].norm()) > 1e-3 * gmax and not k.endswith("prelu.weight")]
    sums = [k for k in names if k.endswith(("bn1.bias", "bn2.bias"))]              # column sums of tensors whose channel means a BatchNorm has just removed
    rest = [k for k in names if k not in sums]
    nerr = np.array([abs(float(params[k].grad.norm()) - float(g_ref[k].norm())) / float(g_ref[k].norm()) for k in rest])
    derr = np.array([rel(params[k].grad, g_ref[k]) for k in rest])
    serr = np.array([rel(params[k].grad, g_ref[k]) for k in sums] or [0.0])      # (may be empty: those sums are tiny and fall under the 1e-3 norm filter)
    print("MEASURED %s slope-1 network: cosine %.3e loss %.3e; grad norms median %.3e max %.3e; directions median %.3e max %.3e (%s); "
          "cancelling sums median %.3e max %.3e; head grad %.3e" %
          (arch, rel(cosine, c_ref), abs(float(loss) - l_ref) / abs(l_ref), np.median(nerr), nerr.max(), np.median(derr), derr.max(),
           rest[int(np.argmax(derr))], np.median(serr), serr.max(), rel(fcm.fc.grad, fcg_ref)))
    assert rel(cosine, c_ref) < T16(SPEC, 3.3e-2)
    assert abs(float(loss) - l_ref) < 5e-3 * abs(l_ref)
    assert np.median(nerr) < T16(SPEC, 2e-2) and nerr.max() < T16(SPEC, 6e-2), (np.median(nerr), nerr.max())
    assert np.median(derr) < T16(SPEC, 5e-2) and derr.max() < T16(SPEC, 0.15), (np.median(derr), derr.max(), rest[int(np.argmax(derr))])
    assert rel(fcm.fc.grad, fcg_ref) < T16(SPEC, 3.3e-2)


class _MaskedPReLU(torch.autograd.Function):
    """F.prelu in the forward direction; the BACKWARD pass uses an injected sign pattern (`pos` = where the HIP path's PReLU input was
    positive) instead of the sign of its own input: d/dz = 1 where pos, the slope elsewhere; d/dslope = sum of dy z over the rest."""

    @staticmethod
    def forward(ctx, z, w, pos):
        ctx.save_for_backward(z, w, pos)
        return F.prelu(z, w)

    @staticmethod
    def backward(ctx, dy):
        z, w, pos = ctx.saved_tensors
        wv = w.view(1, -1, 1, 1)
        dz = torch.where(pos, dy, dy * wv)
        dw = torch.where(pos, torch.zeros_like(dy), dy * z).sum(dim=(0, 2, 3))
        return dz, dw, None


@pytest.mark.parametrize("arch,batch", [("iresnet50", 8), ("iresnet100", 6)])
def test_prelu_kink_law_and_mask_injected_gradients(arch, batch):
    """Closes the one parity exception with a CHECK on the real network with its real slopes:
    (b) the fp32 oracle's backward pass run with the HIP path's PReLU sign pattern (read back from the saved activations, injected through
        oracle.ref_cpu's prelu_hook — the way the dropout test injects the HIP mask): every parameter gradient of the HIP step, norm AND direction,
        is inside north_star's 1e-2 of it — what is left of the kink class once both sides differentiate the same piecewise-linear function;
    (a) the law itself: per PReLU the fraction f of inputs whose sign differs between the HIP activations and the fp32 oracle is counted, and
        the observed direction error of a parameter against the plain fp32 oracle (= the reference, tests/test_oracle_golden.py) is at most
        1.5 x sqrt(sum over the PReLUs its gradient passed of (1 - slope)^2 f) plus its mask-injected error (triangle inequality)."""
    C = 32
    m, sd, layers = make_model(arch)
    m.train()
    fcm = client.FC_module(512, C, "/tmp").to(DEV)
    fc0 = R.head_fc(C)
    fcm.fc.data = fc0.clone().to(DEV)
    x, lab = R.closed_form_images(batch), R.closed_form_labels(batch, C)
    cosine = client.Sequential_model(m, fcm)(x.to(DEV))
    loss = ops.cross_entropy(losses.CosFace(s=30, m=0.4)(cosine, lab.to(DEV)), lab.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    params = dict(m.named_parameters())
    # the HIP path's sign pattern: a = prelu(z) has the sign of z (every closed-form slope is positive)
    assert all(float(v.min()) > 0 for k, v in sd.items() if k.endswith("prelu.weight"))
    plan = m._plan(batch)
    hip_pos = {}

    def nchw(a, C_):
        hw = int(round((a.shape[0] // batch) ** 0.5))
        return (a.view(batch, hw, hw, C_) > 0).permute(0, 3, 1, 2).contiguous()
    hip_pos["prelu"] = nchw(_hip_act(plan, -1, 1), 64)
    order = ["prelu"]
    bi_g = 0
    for si, nblk in enumerate(layers):
        for bi in range(nblk):
            a2 = _hip_act(plan, bi_g, 3)
            name = "layer%d.%d.prelu" % (si + 1, bi)
            hip_pos[name] = nchw(a2, a2.shape[1])
            order.append(name)
            bi_g += 1
    flips, slope = {}, {}

    def hook(name, z, w):
        own = z.detach() > 0
        flips[name] = float((own != hip_pos[name]).float().mean())
        slope[name] = float(w.detach().mean())
        return _MaskedPReLU.apply(z, w, hip_pos[name])
    sd_a = {k: v.clone() for k, v in sd.items()}
    _, c_ref, l_ref, g_ref, fcg_ref = R.train_step_grads(sd_a, fc0.clone(), x, lab, layers)                       # the reference's own derivative
    sd_b = {k: v.clone() for k, v in sd.items()}
    _, c_m, l_m, g_m, fcg_m = R.train_step_grads(sd_b, fc0.clone(), x, lab, layers, prelu_hook=hook)            # ... with the HIP sign pattern
    assert torch.equal(c_ref, c_m)                                                                                # (forward values are untouched)
    gmax = max(float(v.norm()) for v in g_ref.values())
    names = [k for k in R.trainable_keys(sd) if float(g_ref[k].norm()) > 1e-3 * gmax]
    sums = [k for k in names if k.endswith(("bn1.bias", "bn2.bias"))]       # nearly cancelling column sums: listed apart, as everywhere
    rest = [k for k in names if k not in sums]
    # (b) HIP vs the mask-injected oracle
    nerr = np.array([abs(float(params[k].grad.norm()) - float(g_m[k].norm())) / float(g_m[k].norm()) for k in rest])
    derr = {k: rel(params[k].grad, g_m[k]) for k in rest}
    dv = np.array(list(derr.values()))
    wk = max(derr, key=derr.get)
    print("MEASURED %s mask-injected oracle: grad norms median %.3e max %.3e; directions median %.3e max %.3e (%s); head grad %.3e; flipped "
          "fraction per PReLU median %.3e max %.3e" % (arch, np.median(nerr), nerr.max(), np.median(dv), dv.max(), wk, rel(fcm.fc.grad, fcg_m),
                                                      np.median(list(flips.values())), max(flips.values())))
    # (a) the law: prediction per parameter from the flips of the PReLUs its gradient passed through
    idx = {n: i for i, n in enumerate(order)}

    def first_prelu(k):
        """index (forward order) of the first PReLU the gradient of parameter k has passed on its way back from the loss"""
        if not k.startswith("layer"):
            return 0 if k in ("conv1.weight", "bn1.weight", "bn1.bias", "prelu.weight") else len(order)      # stem: all of them; tail: none
        blk = ".".join(k.split(".")[:2])
        own = idx[blk + ".prelu"]
        behind = k.split(".")[2] in ("bn1", "conv1", "bn2", "prelu")
        return own if behind else own + 1
    obs = {k: rel(params[k].grad, g_ref[k]) for k in rest}
    kink = {k: rel(g_m[k], g_ref[k]) for k in rest}                         # the kink's share alone (CPU, both sides fp32)
    worst = (0.0, None)
    for k in rest:
        pred = float(np.sqrt(sum(((1.0 - slope[n]) ** 2) * flips[n] for n in order[first_prelu(k):])))
        bound = 1.5 * pred + 1.05 * derr[k] + 1e-4
        worst = max(worst, (obs[k] / bound, k))
        assert obs[k] <= bound, (k, obs[k], pred, derr[k])
        assert kink[k] <= 1.5 * pred + 1e-4, (k, kink[k], pred)
    ov, kv = np.array(list(obs.values())), np.array(list(kink.values()))
    print("MEASURED %s kink law: observed directions vs fp32 oracle median %.3e max %.3e; kink share (oracle vs mask-injected oracle) median %.3e "
          "max %.3e; worst observed / bound %.2f (%s)" % (arch, np.median(ov), ov.max(), np.median(kv), kv.max(), worst[0], worst[1]))
    assert np.median(nerr) < T16(SPEC, 2e-2) and nerr.max() < T16(SPEC, 6e-2), (np.median(nerr), nerr.max())
    assert np.median(dv) < T16(SPEC, 5e-2) and dv.max() < T16(SPEC, 0.15), (np.median(dv), dv.max(), wk)
    assert rel(fcm.fc.grad, fcg_m) < T16(SPEC, 3.3e-2)


def test_fused_client_loop_vs_reference():
    """FusedTrainer == the reference hot loop (client.py:537-550): iresnet18, 3 SGD steps (lr 0.01, momentum 0.9,
    wd 5e-4), closed-form data.  Losses track the fp32 reference to < 0.5 % (measured 0.07 %)."""
    g = load_golden("client_r18")
    B, C, steps, lr = int(g["B"]), int(g["C"]), int(g["steps"]), float(g["lr"])
    m, sd, layers = make_model("iresnet18", tag=2.0)
    fc = R.head_fc(C).to(DEV)
    tr = client.FusedTrainer(m, fc, "CosFace", 30.0, 0.4, lr=lr, momentum=0.9, weight_decay=5e-4)
    ls = []
    for st in range(steps):
        imgs = R.closed_form_images(B, tag=float(st)).to(DEV)
        lab = R.closed_form_labels(B, C, tag=st).to(DEV)
        ls.append(float(tr.step(imgs, lab)))
    np.testing.assert_allclose(np.array(ls), g["losses"], rtol=5e-3)
    out = m.state_dict()
    for k in ("bn1.running_mean", "bn1.running_var", "layer4.1.bn3.running_var", "features.running_mean"):
        assert rel(out[k], g["sd_" + k]) < T16(SPEC, 3e-2), (k, rel(out[k], g["sd_" + k]))      # measured 5.2e-3 on fp16
    assert int(out["bn1.num_batches_tracked"]) == int(g["sd_bn1.num_batches_tracked"])
    for k in ("conv1.weight", "layer2.0.downsample.0.weight", "bn1.weight", "prelu.weight", "features.bias"):
        assert rel(out[k], g["sd_" + k]) < 1e-2, (k, rel(out[k], g["sd_" + k]))
    assert rel(out["fc.weight"][:4, :2048], g["sd_fc.weight_slice"]) < 1e-2
    assert rel(fc, g["head_fc"]) < 5e-2


@pytest.mark.parametrize("arch,dual,B", [("iresnet18", True, 8), ("iresnet50", True, 8), ("iresnet18", False, 8), ("iresnet50", True, 128)])
def test_sgd_inside_backward_is_bit_identical(arch, dual, B, monkeypatch):
    """step() folds torch.optim.SGD's update into the backward pass (fedfr_net_backward2_sgd: the bn2 / fc / features tail and every finished
    stage are updated on the weight-gradient stream while the main stream is still in the earlier stages).  Same kernels, same element-wise
    arithmetic, only enqueued earlier: parameters, momentum buffers, bf16 mirrors and losses after 3 steps are bit-identical to
    backward + one flat fedfr_sgd_step (FEDFR_FUSE_SGD=0), with one stream and with two; at batch 128 too (round 6: the size at which the paired
    weight-gradient launches carry each other's slab reductions)."""
    monkeypatch.setenv("FEDFR_DUAL_STREAM", "1" if dual else "0")
    C = 40
    res = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("FEDFR_FUSE_SGD", fuse)
        m, sd, layers = make_model(arch, tag=5.0)
        fc = R.head_fc(C).to(DEV)
        tr = client.FusedTrainer(m, fc, "CosFace", 30.0, 0.4, lr=0.05, momentum=0.9, weight_decay=5e-4)
        assert tr.fuse_sgd == (fuse == "1")
        ls = []
        for st in range(3):
            imgs = R.closed_form_images(B, tag=float(st)).to(DEV)
            lab = R.closed_form_labels(B, C, tag=st).to(DEV)
            ls.append(float(tr.step(imgs, lab)))
        tr.finish()
        torch.cuda.synchronize()
        res.append((ls, m._flat_params.clone(), tr.mom.clone(), m._shadow[: m.trainable_count()].clone(), fc.clone(), m._flat_grads.clone()))
    a, b = res
    assert a[0] == b[0], (a[0], b[0])
    # gradient buffer: in the fp16-storage build the fused update kernels undo the loss scale in place (fedfr_sgd_step_scaled)
    for i, name in ((1, "parameters"), (2, "momentum"), (3, "bf16 mirror"), (4, "head fc"), (5, "gradient buffer")):
        assert torch.equal(a[i], b[i]), name
    # and the update really happened inside the pass: only stem + stage 1 are left to the flat kernel
    tr._fuse_sgd = True
    imgs = R.closed_form_images(B, tag=9.0).to(DEV)
    tr.forward_backward(imgs, R.closed_form_labels(B, C, tag=9).to(DEV))
    tr._fuse_sgd = False
    first_s2 = dict(m.named_parameters())["layer2.0.bn1.weight"]
    off = (first_s2.data_ptr() - m._flat_params.data_ptr()) // 4
    assert tr._sgd_done_from == off, (tr._sgd_done_from, off)
    tr.optimizer_step()
    tr.finish()


@pytest.mark.parametrize("arch", ["iresnet18", "iresnet50"])
def test_head_trainer_sgd_inside_backward_is_bit_identical(arch, monkeypatch):
    """FusedHeadTrainer (the body of train_with_public_data and of the config-5 trainer) folds the backbone's SGD update into the backward pass
    like FusedTrainer.step() (round 5): parameters, momentum, 16-bit mirrors, head parameters and losses after 3 steps are bit-identical to
    backward + one flat update (FEDFR_FUSE_SGD=0)."""
    B, C = 8, 24
    res = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("FEDFR_FUSE_SGD", fuse)
        m, sd, layers = make_model(arch, tag=6.0)
        fcm = client.FC_module(512, C, "/tmp").to(DEV)
        fcm.fc.data = R.head_fc(C).to(DEV)
        margin = losses.CosFace(s=30, m=0.4)
        tr = client.FusedHeadTrainer(m, list(fcm.parameters()), lr=0.05, momentum=0.9, weight_decay=5e-4)
        assert tr.fuse_sgd == (fuse == "1")

        def head_loss(feats, labels):
            return ops.cross_entropy(margin(fcm(feats), labels), labels)
        ls = []
        for st in range(3):
            imgs = R.closed_form_images(B, tag=float(st)).to(DEV)
            lab = R.closed_form_labels(B, C, tag=st).to(DEV)
            ls.append(float(tr.step(imgs, lab, head_loss)))
        tr.finish()
        torch.cuda.synchronize()
        res.append((ls, m._flat_params.clone(), tr.mom.clone(), m._shadow[: m.trainable_count()].clone(), fcm.fc.data.clone()))
    a, b = res
    assert a[0] == b[0], (a[0], b[0])
    for i, name in ((1, "parameters"), (2, "momentum"), (3, "16-bit mirror"), (4, "head fc")):
        assert torch.equal(a[i], b[i]), name


@pytest.mark.parametrize("variant", ["full", "seq", "bce_rw"])
def test_train_with_public_data_vs_reference(variant):
    """Client.train_with_public_data (client.py:287-508) through the fused head trainer: iresnet18, 6 local + 14 public
    classes, 3 SGD steps.  'full' = Branch_model + 10*BCE + mu*contrastive (frozen global / last-round backbones in eval
    mode), 'seq' = plain Sequential model, 'bce_rw' = BCE + reweight_cosface (detached CosFace term, reference quirk).
    Compared with the values captured from the imported reference modules."""
    g = load_golden("client_public_" + variant)
    nl, npub, B, steps = int(g["n_local"]), int(g["n_public"]), int(g["B"]), int(g["steps"])
    layers = R.IRESNET_LAYERS["iresnet18"]

    class Args:
        network, loss, local_epoch, output_dir, aggr_alg, num_client = "iresnet18", "CosFace", 1, "/tmp", "FedAvg", 4
        BCE_local = variant in ("full", "bce_rw")
        contrastive_bb = variant == "full"
        reweight_cosface = variant == "bce_rw"
        BCE_detach, combine_dataset = False, True

    class DS:
        ID_base, num_classes = 0, nl

    class Loader(list):
        dataset = DS()

    class Data:
        train_class_sizes, train_dataset_sizes, train_loaders = [nl], [B * steps], [Loader()]

    from fedfr_amd.config import config as cfg
    saved = (cfg.lr, cfg.mu)
    cfg.lr, cfg.mu = float(g["lr"]), float(g["mu"])
    try:
        cl = client.Client(0, Args, Data, device=DEV)
        cl.backbone_state_dict = R.closed_form_state_dict(layers, tag=float(g["tag"]))
        cl.fc_module.fc.data = R.head_fc(nl, seed=11)
        if Args.BCE_local:
            cl.bce_module.weight.data = R.head_fc(nl, seed=13)
        if Args.contrastive_bb:
            cl.last_model.load_state_dict(R.closed_form_state_dict(layers, tag=float(g["last_tag"])))
        batches = [(R.closed_form_images(B, tag=float(st)), R.closed_form_labels(B, nl + npub, tag=st)) for st in range(steps)]
        cl.train_with_public_data(pretrained_fc=R.head_fc(npub, seed=12), combine_loader=batches)
    finally:
        cfg.lr, cfg.mu = saved
    rows = g["rows"]
    assert abs(cl.get_train_loss() - rows[:, 0].mean()) < 1e-2 * abs(rows[:, 0].mean()), (cl.get_train_loss(), rows[:, 0].mean())
    assert abs(cl.cos_meter.avg - rows[:, 1].mean()) < 1e-2 * abs(rows[:, 1].mean())
    if variant == "full":
        assert abs(cl.con_meter.avg - rows[:, 2].mean()) < 2e-2 * abs(rows[:, 2].mean()), (cl.con_meter.avg, rows[:, 2].mean())
    if Args.BCE_local:
        assert abs(cl.bce_meter.avg - rows[:, 3].mean()) < 1e-2 * abs(rows[:, 3].mean())
    out = cl.get_model()
    assert int(out["bn1.num_batches_tracked"]) == int(g["sd_bn1.num_batches_tracked"])
    for k in ("bn1.running_mean", "layer4.1.bn3.running_var", "features.running_mean"):
        assert rel(out[k], g["sd_" + k]) < 3e-2, (k, rel(out[k], g["sd_" + k]))
    for k in ("conv1.weight", "layer2.0.downsample.0.weight", "bn1.weight", "prelu.weight", "fc.bias"):
        assert rel(out[k], g["sd_" + k]) < 1e-2, (k, rel(out[k], g["sd_" + k]))
    assert rel(cl.fc_module.fc.data, g["head_fc"]) < 5e-2
    assert cl.fc_module.fc.shape[0] == nl + npub and cl.get_global_fc().shape[0] == npub
    if Args.BCE_local:
        assert rel(cl.bce_module.weight.data, g["bce_weight"]) < 5e-2
        assert rel(cl.bce_module.converter[0].weight.data[:8, :64], g["bce_conv_w_slice"]) < 1e-2
        assert float((cl.bce_module.bias.data - T(g["bce_bias"]).cpu()).abs().max()) < 2e-3
    if variant == "full":       # last_model <- this round's local model (client.py:499-501)
        assert torch.equal(cl.last_model.state_dict()["conv1.weight"].cpu(), out["conv1.weight"].cpu())


def test_client_server_round():
    """Two clients, one FedAvg round through the reference-shaped Client / Server objects (server.py:265-338):
    the aggregate equals the data-size-weighted mean of the two locally trained models (bit exact, flat path)."""
    class Args:
        network, loss, local_epoch, output_dir, BCE_local, aggr_alg = "iresnet18", "CosFace", 1, "/tmp", False, "FedAvg"

    class DS:
        ID_base = 0

    class Loader(list):
        dataset = DS()

    class Data:
        train_class_sizes = [10, 10]
        train_dataset_sizes = [300, 100]
        train_loaders = [Loader([(R.closed_form_images(4, tag=float(c * 2 + s)), R.closed_form_labels(4, 10, tag=c + s))
                                 for s in range(2)]) for c in range(2)]

    from fedfr_amd.config import config as cfg
    cfg.lr = 0.01
    clients = [client.Client(c, Args, Data, device=DEV) for c in range(2)]
    srv = server.Server(clients, Data, Args, device=DEV)
    srv.federated_model.load_state_dict(R.closed_form_state_dict(R.IRESNET_LAYERS["iresnet18"], tag=2.0))
    avg_loss = srv.train()
    assert np.isfinite(avg_loss)
    m0, m1 = clients[0].get_model(), clients[1].get_model()
    agg = srv.federated_model.state_dict()
    for k in ("conv1.weight", "layer3.1.conv2.weight", "bn2.running_var", "fc.bias"):
        exp = np.float32(0.75) * m0[k] + np.float32(0.25) * m1[k]
        assert torch.equal(agg[k], exp), k
    assert int(agg["bn1.num_batches_tracked"]) == int(0.75 * float(m0["bn1.num_batches_tracked"]) + 0.25 * float(m1["bn1.num_batches_tracked"]))
    assert not torch.equal(m0["conv1.weight"], m1["conv1.weight"])


def test_sweeps_and_hard_negative_mining_vs_reference():
    """SURVEY §8f N1/N2 through the reference-shaped API: Client.data_update_fc (client.py:159-188), Server.Generate_pretrain_feats /
    Initialize_pretrain_FC (server.py:182-263), Client.choose_hard_negative_2 (client.py:191-236) on iresnet18 against values
    captured from the imported reference (bf16 backbone: 1e-2-class embeddings; the selected index set is exact)."""
    g = load_golden("mining_r18")
    layers = R.IRESNET_LAYERS["iresnet18"]
    sd, local, public = R.mining_fixture_state(g)
    nl, npub = int(g["n_local"]), int(g["n_public"])

    class Args:
        network, loss, local_epoch, output_dir, aggr_alg, num_client = "iresnet18", "CosFace", 1, "/tmp", "FedAvg", 2
        BCE_local = contrastive_bb = False
        norm_before_avg = True

    class DS:
        def __init__(self, n, idb=0):
            self.num_classes, self.ID_base = n, idb

    class Loader(list):
        pass

    def loader(batches, ds):
        l = Loader(batches)
        l.dataset = ds
        return l

    class Data:
        train_class_sizes, train_dataset_sizes = [nl], [12]
        train_loaders = [loader(local, DS(nl))]
        test_loaders = [loader(local, DS(nl))]
        public_train_loader = loader(public, DS(npub))
        public_test_loader = loader(public, DS(npub))

    cl = client.Client(0, Args, Data, device=DEV)
    for nba in (True, False):
        cl.data_update_fc(sd, nba)
        assert rel(cl.fc_module.fc.data, g["local_centers_nba%d" % int(nba)]) < 2e-2, nba
    srv = server.Server([cl], Data, Args, device=DEV)
    srv.federated_model.load_state_dict(sd)
    feats = srv.Generate_pretrain_feats()
    assert rel(feats, g["public_feats"]) < 2e-2
    assert float((feats.norm(dim=1) - 1).abs().max()) < 1e-5
    init_matrix, raw_labels = srv.Initialize_pretrain_FC()
    assert torch.equal(raw_labels, T(g["public_labels"]))
    assert rel(init_matrix, g["public_centers"]) < 2e-2
    cl.backbone_state_dict = sd
    sub = cl.choose_hard_negative_2(Data.public_train_loader, raw_labels, feats, threshold=float(g["hn_threshold"]))
    assert torch.equal(cl.HN_index, T(g["hn_index"]))                       # same hard-negative image set as the reference
    assert len(sub.dataset) == len(g["hn_index"]) and sub.dataset.num_classes == npub


# per type: (embeddings, gradient-norm median, gradient-norm max, direction median, direction max).
# fp16: embeddings 7.7e-4 / 7.0e-4 and norm medians 2.4e-3 / 3.2e-3 at the spec; worst norm 2.0e-2 / 1.4e-2 and directions 3.4e-2 / 3.0e-2 median,
# 5.4e-2 / 8.5e-2 max: the PReLU-kink class (every sphnet unit is conv -> PReLU)
# bf16 subset, round 4 x 1.25: sphere20 6.3e-3, 4.1e-3 / 5.8e-2, 0.106 / 0.160; sphere64 5.5e-3, 5.8e-3 / 3.3e-2, 0.075 / 0.273
SPH_TOL_BF16 = {20: (2e-2, 1e-2, 0.12, 0.15, 0.4), 64: (7e-3, 7.5e-3, 4.2e-2, 0.095, 0.345)}
SPH_TOL_FP16 = (SPEC, SPEC, 2.5e-2, 4.3e-2, 1.07e-1)      # kink class: measured (2.0e-2, 3.4e-2, 8.5e-2) x 1.25


@pytest.mark.parametrize("type_", [20, 64])
def test_sphnet_vs_reference(type_):
    """backbones.sphnet — sphere20 and sphere64 (the reference's default, sphnet.py:72; what run.sh trains and bench.py --arch
    sphnet times) — reference state_dict keys / shapes, forward and all parameter gradients against the imported reference module
    (bf16 activations: 1e-2-class), eval == train forward (no normalisation layers)."""
    g = load_golden("sphnet%d" % type_)
    B = int(g["B"])
    sd = R.sphere_state_dict(type_, tag=1.0)
    net = backbones.sphnet(False, dropout=0, fp16=True, type=type_).to(DEV)
    assert list(net.state_dict().keys()) == [str(k) for k in g["keys"]]
    assert all(tuple(v.shape) == tuple(sd[k].shape) for k, v in net.state_dict().items())
    net.load_state_dict(sd)
    back = net.state_dict()
    assert all(torch.equal(back[k].cpu(), sd[k]) for k in sd)                        # KRSC storage is invisible through state_dict
    net.train()
    x = R.closed_form_images(B, tag=4.0).to(DEV)
    dfe = R.closed_form((B, 512), 0.37, 0.9, 1.0).to(DEV)
    feats = net(x)
    tol_f, tol_nm, tol_nx, tol_dm, tol_dx = T16(SPH_TOL_FP16, SPH_TOL_BF16[type_])
    print("MEASURED sphere%d embeddings %.3e" % (type_, rel(feats, g["feats"])))
    assert rel(feats, g["feats"]) < tol_f, rel(feats, g["feats"])
    (feats * dfe).sum().backward()
    nerr, derr = [], []
    grads = dict(net.named_parameters())
    for k, p in grads.items():
        assert p.grad is not None and p.grad.shape == p.shape, k
        nerr.append(abs(float(p.grad.norm()) - float(g["gnorm_" + k])) / float(g["gnorm_" + k]))
        key = "g_" + k
        if key in g.files:
            derr.append((k, rel(p.grad, g[key])))
    for key in [f for f in g.files if f.endswith("_slice")]:          # slices of large tensors: "g_<param>_slice" = grad[:a, :b]
        ref = T(g[key])
        derr.append((key[2:-6] + "[slice]", rel(grads[key[2:-6]].grad[: ref.shape[0], : ref.shape[1]], ref)))
    dvals = [e for _, e in derr]
    print("MEASURED sphere%d gradient norms median %.3e max %.3e; directions median %.3e max %.3e (%s)" %
          (type_, np.median(nerr), max(nerr), np.median(dvals), max(dvals), max(derr, key=lambda t: t[1])[0]))
    # per-tensor gradient norms: median 0.4 %; the PReLU-slope / bias gradients are sign-filtered sums of bf16-rounded values -> up to ~6 %
    assert np.median(nerr) < tol_nm and max(nerr) < tol_nx, (np.median(nerr), max(nerr))
    # direction errors: bf16 storage noise through the whole backward chain; worst on the first layer's bias, a heavily cancelling
    # sum over 4e5 pixels (same policy as the iresnet gradient test: median small, max bounded)
    assert np.median(dvals) < tol_dm and max(dvals) < tol_dx, (np.median(dvals), max(derr, key=lambda t: t[1]))   # sphere20 measured 0.106 / 0.16
    net.eval()
    with torch.no_grad():
        assert torch.equal(net(x), feats.detach())


def test_sphnet_backward_fusions_at_full_tile_batch():
    """Round 3: at batches whose 14x14 / 28x28 layers run on the LDS-DMA kernels with a fused epilogue, sphnet's backward pass (a) lets the
    dgrad of a block's conv2 apply the backward of the PReLU in front of it (option sph_fuse_prelu_bwd: output = dz, rows = the PReLU's
    parameter sums), (b) runs the block's two weight gradients as one paired launch, (c) finalizes all PReLU sums in one launch.  Against
    the plain sequence on the same model and inputs: identical embeddings, conv weight gradients from identical operands (paired kernel:
    another K split order -> fp32 noise), PReLU slope / bias gradients equal up to the summation order of their partial rows."""
    B = 64
    sd = R.sphere_state_dict(20, tag=1.0)
    x = R.closed_form_images(B, tag=4.0).to(DEV)
    dfe = R.closed_form((B, 512), 0.37, 0.9, 1.0).to(DEV)
    res = []
    for on in (0, 1):
        with _C.option_scope("sph_fuse_prelu_bwd", on), _C.option_scope("sph_pair_wgrad", on), _C.option_scope("sph_fin_multi", on):
            net = backbones.sphnet(False, dropout=0, fp16=True, type=20).to(DEV)
            net.load_state_dict(sd)
            net.train()
            f = net(x)
            (f * dfe).sum().backward()
            res.append((f.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()}))
    (f0, g0), (f1, g1) = res
    assert torch.equal(f0, f1)
    for k in g0:
        a, b = g0[k].double().flatten(), g1[k].double().flatten()
        e = float((a - b).norm() / (a.norm() + 1e-30))
        assert e < 2e-5, (k, e)
    # round 4: a paired launch sums the PREVIOUS pair's split-K slabs itself where the shape allows (sphere64's 16-unit 14x14 stage at this
    # batch: 8 slabs per layer) — same summation order as the stand-alone reduction launches, so every gradient bit is the same
    res64 = []
    sd64 = R.sphere_state_dict(64, tag=1.0)
    for bg in (1, 0):
        with _C.option_scope("wgrad9p_bg", bg):
            net = backbones.sphnet(False, dropout=0, fp16=True, type=64).to(DEV)
            net.load_state_dict(sd64)
            net.train()
            f = net(x)
            (f * dfe).sum().backward()
            res64.append({k: p.grad.clone() for k, p in net.named_parameters()})
            del net
    assert all(torch.equal(res64[0][k], res64[1][k]) for k in res64[0])


def test_roc_vs_reference():
    """roc_cuda.py end to end: histogram and TPR@FPR read-out equal the values produced by the reference's own kernel body."""
    from fedfr_amd import eval_roc
    g = load_golden("roc")
    hist = eval_roc.roc_histogram(T(g["features"]).to(DEV), T(g["labels"]).to(DEV), int(g["target_size"]))
    assert np.array_equal(hist.cpu().numpy(), g["hist"])
    assert eval_roc.tpr_at_fpr(hist) == [float(v) for v in g["tpr"]]
    f, l, t = eval_roc.order_targets(T(g["features"]).to(DEV), T(g["labels"]).to(DEV), [0, 1, 2])
    assert t == int(g["target_size"]) and torch.equal(l.cpu(), T(g["labels"]))     # already target-first: a stable partition keeps it


def test_paired_weight_gradient_kernel_in_the_network():
    """option wgrad9p (default since round 3): every block's two same-shape 3x3 weight gradients from the paired
    64 x 64 nine-tap kernel — same operands, same K order per split, so the network's gradients agree with the single-layer kernel's to fp32
    summation-order level (activations and activation gradients do not depend on the choice at all: bit-identical embeddings)."""
    outs = []
    for opt in (0, 1):
        with _C.option_scope("wgrad9p", opt):
            m, sd, _ = make_model("iresnet50", tag=3.0)
            m.train()
            f = m(R.closed_form_images(16).to(DEV))
            (f * R.closed_form((16, 512), 0.37, 0.9, 1.0).to(DEV)).sum().backward()
            outs.append((f.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert torch.equal(outs[0][0], outs[1][0])
    worst = max((rel(outs[1][1][k], outs[0][1][k]), k) for k in outs[0][1] if float(outs[0][1][k].norm()) > 0)
    print("wgrad9p on vs off: worst %.2e (%s)" % worst)
    assert worst[0] < 1e-5, worst
    for k in outs[0][1]:
        if "conv" not in k:                       # everything that is not a paired conv weight gradient is untouched
            assert torch.equal(outs[0][1][k], outs[1][1][k]), k


def test_bn_backward_next_reduction_fusion_equivalence():
    """option fuse_bnred_next (a BN-backward apply pass also reduces its output for the BatchNorm that consumes it): same gradients as
    the separate reduce kernel up to fp32 summation order."""
    outs = []
    for opt in (0, 1):
        _C.call("fedfr_set_option", b"fuse_bnred_next", opt)
        try:
            m, sd, _ = make_model("iresnet18", tag=3.0)
            m.train()
            f = m(R.closed_form_images(64).to(DEV))
            (f * R.closed_form((64, 512), 0.37, 0.9, 1.0).to(DEV)).sum().backward()
            outs.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        finally:
            _C.call("fedfr_set_option", b"fuse_bnred_next", 1)
    # biases in front of a BatchNorm (bn3 / downsample BN feed the next bn1 through the residual sum) have an analytically zero gradient:
    # what is computed is the rounding noise of a cancelling column sum, different for every summation order -> those are held to an
    # ABSOLUTE bar (a fraction of the same layer's weight gradient), everything else to a relative one
    def zero_mean_bias(k):
        return k.endswith(("bn3.bias", "downsample.1.bias"))
    errs = {k: rel(outs[1][k], outs[0][k]) for k in outs[0] if not zero_mean_bias(k) and float(outs[0][k].norm()) > 0}
    # bn1.bias / bn2.bias: a conv and a BatchNorm follow, only the zero-padded border keeps the sum from cancelling (tests/test_block_gpu.py)
    sums = {k: e for k, e in errs.items() if k.endswith(("bn1.bias", "bn2.bias"))}
    rest = {k: e for k, e in errs.items() if k not in sums}
    print("fuse_bnred_next on vs off: worst %.2e (%s); border sums %.2e (%s)" % (max(rest.values()), max(rest, key=rest.get), max(sums.values()),
                                                                                 max(sums, key=sums.get)))
    assert max(rest.values()) < 2e-2, max(rest, key=rest.get)       # bf16 activations downstream of a differently-ordered fp32 sum: last-bit flips propagate
    assert max(sums.values()) < 6e-2, max(sums, key=sums.get)
    for k in outs[0]:
        if zero_mean_bias(k):
            wn = float(outs[0][k[:-4] + "weight"].norm())
            assert float((outs[1][k] - outs[0][k]).norm()) < 2e-2 * wn, (k, float(outs[0][k].norm()), wn)


def test_public_data_server_round():
    """One FedFR round with public data (server.py:265-338, add_pretrained_data + return_all): both clients train the
    [local | public] CosFace head + BCE branch, the server averages backbones (FedPavg) and public class centres (FedAvg_on_FC)."""
    nl, npub, B = 5, 9, 4

    class Args:
        network, loss, local_epoch, output_dir, aggr_alg, num_client = "iresnet18", "CosFace", 1, "/tmp", "FedAvg", 2
        BCE_local, contrastive_bb, reweight_cosface, BCE_detach, combine_dataset = True, False, False, False, True
        add_pretrained_data, return_all = True, True

    class DS(torch.utils.data.Dataset):
        def __init__(self, n, ncls, tag, id_base=0):
            self.num_classes, self.ID_base = ncls, id_base
            self.x, self.y = R.closed_form_images(n, tag=tag), R.closed_form_labels(n, ncls, tag=int(tag))

        def __len__(self):
            return len(self.x)

        def __getitem__(self, i):
            return self.x[i], self.y[i]

    def loader(ds):
        return torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False)

    class Data:
        train_class_sizes = [nl, nl]
        train_dataset_sizes = [8, 4]
        train_loaders = [loader(DS(8, nl, 1.0)), loader(DS(4, nl, 2.0, id_base=nl))]
        test_loaders = train_loaders
        public_train_loader = loader(DS(8, npub, 3.0))
        public_test_loader = public_train_loader

    from fedfr_amd.config import config as cfg
    saved = (cfg.lr, cfg.com_batch_size, cfg.HN_threshold)
    cfg.lr, cfg.com_batch_size = 0.01, B
    try:
        clients = [client.Client(c, Args, Data, device=DEV) for c in range(2)]
        srv = server.Server(clients, Data, Args, device=DEV)
        srv.federated_model.load_state_dict(R.closed_form_state_dict(R.IRESNET_LAYERS["iresnet18"], tag=2.0))
        pre = R.head_fc(npub, seed=12).to(DEV)
        srv.pretrained_fc = pre.clone()
        _, srv.pretrained_label = srv.Initialize_pretrain_FC(only_labels=True)      # enables the per-round sweep + hard-negative mining
        cfg.HN_threshold = -1.0                                                      # every public image qualifies
        avg_loss = srv.train()
    finally:
        cfg.lr, cfg.com_batch_size, cfg.HN_threshold = saved
    assert np.isfinite(avg_loss)
    assert srv.pretrained_feats.shape == (8, 512) and all(len(c.HN_index) == 8 for c in clients)
    assert [c.get_data_size() for c in clients] == [16, 12]                    # combined dataset sizes drive FedAvg (client.py:302)
    assert all(c.fc_module.fc.shape[0] == nl for c in clients)                 # remove_pretrain() after the round
    # reference quirk (server.py:325): the averaged centres land in `pretrain_fc`; `pretrained_fc` (what clients receive) is unchanged
    assert srv.pretrain_fc.shape == (npub, 512) and not torch.equal(srv.pretrain_fc, pre)
    assert torch.equal(srv.pretrained_fc, pre)
    assert srv.global_round == 0 and srv.global_epoch == 0            # counters belong to the driver (train.py:87-88)
    srv.step_round()
    assert srv.global_round == 1 and srv.global_epoch == srv.local_epoch
    m0, m1 = clients[0].get_model(), clients[1].get_model()
    w0, w1 = np.float32(16 / 28), np.float32(12 / 28)
    agg = srv.federated_model.state_dict()
    for k in ("conv1.weight", "layer3.1.conv2.weight", "fc.bias"):
        assert torch.equal(agg[k], w0 * m0[k] + w1 * m1[k]), k


@pytest.mark.parametrize("fuse_bnbwd", [0, 1, 2])
def test_large_batch_train_step_vs_oracle(fuse_bnbwd):
    """iresnet18 at batch 128: M = 25088..1.6M rows, so the conv layers run on the LDS-halo kernels that the small-batch
    goldens never reach — with the BN-backward reduction as its own kernel (default) and fused into the dgrad epilogue
    (option fuse_bnbwd).  Compared with the fp32 oracle on the same inputs."""
    with _C.option_scope("fuse_bnbwd", fuse_bnbwd):
        _large_batch_step()


# fp16: cosine 1.5e-3, loss 1e-6, norms median 8.4e-4 (spec), head gradient 1.6e-3 (spec); worst norm 1.02e-2 and directions 1.6e-2 / 4.7e-2: kink class
# bf16 subset: round 4 x 1.25 (cosine 1.19e-2, norms 2.6e-3 / 9.3e-2, directions 4.8e-2 / 0.34, head gradient 1.2e-2)
LB_TOL_BF16 = {"cosine": 1.5e-2, "norm_median": 3.2e-3, "norm_max": 0.12, "dir_median": 6e-2, "dir_max": 0.43, "head": 1.6e-2}
LB_TOL_FP16 = {"cosine": SPEC, "norm_median": SPEC, "norm_max": KINK_NORM_MAX, "dir_median": KINK_DIR_MEDIAN, "dir_max": KINK_DIR_MAX, "head": SPEC}


def _large_batch_step():
    layers = R.IRESNET_LAYERS["iresnet18"]
    LB_TOL = T16(LB_TOL_FP16, LB_TOL_BF16)
    B, C = 128, 100
    m, sd, _ = make_model("iresnet18", tag=3.0)
    m.train()
    fcm = client.FC_module(512, C, "/tmp").to(DEV)
    fc0 = R.head_fc(C)
    fcm.fc.data = fc0.clone().to(DEV)
    x, lab = R.closed_form_images(B), R.closed_form_labels(B, C)
    cosine = client.Sequential_model(m, fcm)(x.to(DEV))
    loss = ops.cross_entropy(losses.CosFace(s=30, m=0.4)(cosine, lab.to(DEV)), lab.to(DEV))
    loss.backward()
    torch.set_num_threads(min(64, torch.get_num_threads()))
    f_ref, c_ref, l_ref, g_ref, fcg_ref = R.train_step_grads(sd, fc0.clone(), x, lab, layers)
    print("MEASURED r18 b128 cosine %.3e loss %.3e" % (rel(cosine, c_ref), abs(float(loss) - l_ref) / abs(l_ref)))
    assert rel(cosine, c_ref) < LB_TOL["cosine"], rel(cosine, c_ref)
    assert abs(float(loss) - l_ref) < 5e-3 * abs(l_ref)
    params = dict(m.named_parameters())
    names = [k for k in R.trainable_keys(sd) if float(g_ref[k].norm()) > 1e-6 * max(float(v.norm()) for v in g_ref.values())]
    nerr = np.array([abs(float(params[k].grad.norm()) - float(g_ref[k].norm())) / float(g_ref[k].norm()) for k in names])
    derr = np.array([rel(params[k].grad, g_ref[k]) for k in names])
    print("MEASURED r18 b128 grad norms median %.3e max %.3e; directions median %.3e max %.3e (%s); head grad %.3e" %
          (np.median(nerr), nerr.max(), np.median(derr), derr.max(), names[int(np.argmax(derr))], rel(fcm.fc.grad, fcg_ref)))
    assert np.median(nerr) < LB_TOL["norm_median"] and nerr.max() < LB_TOL["norm_max"], (np.median(nerr), nerr.max(), names[int(np.argmax(nerr))])
    assert np.median(derr) < LB_TOL["dir_median"] and derr.max() < LB_TOL["dir_max"], (np.median(derr), derr.max(), names[int(np.argmax(derr))])
    assert rel(fcm.fc.grad, fcg_ref) < LB_TOL["head"]


@pytest.mark.parametrize("arch", ["iresnet100", "iresnet50"])
def test_full_size_step_invariants_r100_b128(monkeypatch, arch):
    """BASELINE.json's metric configuration (iresnet100 + CosFace, batch 128, 112x112) and its config 2 (iresnet50 + CosFace, batch 128) at
    their own size, too large for the CPU oracle in a test:
    size-independent properties instead.  (1) the step is deterministic: two runs give bit-identical loss and gradients (no atomics
    anywhere); (2) kernel and scheduling choices only change fp32 summation order: gradients of the nine-tap weight-gradient kernel
    vs the GEMM-form kernels, and of the dual-stream vs the single-stream backward, agree to 1e-4 of the gradient norm (bf16
    operands are exact in the MFMA, so nothing else may differ); (3) every BatchNorm's num_batches_tracked advances once per forward."""
    torch.manual_seed(7)
    g = torch.Generator().manual_seed(100)
    B, C = 128, 1000
    x = (torch.rand(B, 3, 112, 112, generator=g) * 2 - 1).to(DEV)
    lab = torch.randint(0, C, (B,), generator=g).to(DEV)
    m = getattr(backbones, arch)(False, dropout=0, fp16=True).to(DEV)
    m.train()
    fc0 = (torch.randn(C, 512, generator=g) * 0.01).to(DEV)

    def run(dual=True, **opts):
        monkeypatch.setenv("FEDFR_DUAL_STREAM", "1" if dual else "0")
        prev = {k: _C.get_option(k) for k in opts}
        for k, v in opts.items():
            _C.call("fedfr_set_option", k.encode(), v)
        try:
            tr = client.FusedTrainer(m, fc0.clone(), "CosFace", 30.0, 0.4, lr=0.0)
            loss = float(tr.forward_backward(x, lab))
            torch.cuda.synchronize()
            return loss, m._flat_grads.clone(), tr.fc_grad.clone()
        finally:
            for k in opts:
                _C.call("fedfr_set_option", k.encode(), prev[k])

    l0, g0, f0 = run()
    l1, g1, f1 = run()
    assert l0 == l1 and torch.equal(g0, g1) and torch.equal(f0, f1)                 # (1)
    assert np.isfinite(l0) and float(g0.norm()) > 0
    for kw in (dict(wgrad9=0), dict(dual=False), dict(wgrad9=0, tn_glds=0)):          # (2)
        l2, g2, f2 = run(**kw)
        assert abs(l2 - l0) <= 1e-6 * abs(l0), kw
        assert float((g2 - g0).norm()) <= 1e-4 * float(g0.norm()), (kw, float((g2 - g0).norm() / g0.norm()))
        assert torch.equal(f2, f0), kw
    # (2b) round 4: a paired weight-gradient launch sums the PREVIOUS pair's split-K slabs beside its own work (csrc/wgrad9p.hip, W9PJob) in the
    # stand-alone reduction kernels' own summation order: with the launches put back (wgrad9p_bg = 0) every gradient bit is the same
    l3, g3, f3 = run(wgrad9p_bg=0)
    assert l3 == l0 and torch.equal(g3, g0) and torch.equal(f3, f0)
    # (3)
    nbt = [v for k, v in m.state_dict().items() if k.endswith("num_batches_tracked")]
    assert len(nbt) == {"iresnet100": 154, "iresnet50": 79}[arch] and len({int(v) for v in nbt}) == 1


@pytest.mark.parametrize("arch", ["iresnet100", "iresnet50"])
def test_full_size_product_path_vs_fp32_validation_path(arch):
    """BASELINE.json's metric configuration at FULL size (iresnet100, batch 128, 112x112) and config 2's (iresnet50, batch 128) — too large for the CPU oracle, not for the fp32
    validation path (csrc/net_f32.hip, itself checked against the reference at 1e-6 on the small fixtures): the bf16 product path's
    embeddings (eval and train mode), running statistics and parameter gradients of one step against it, same weights, same inputs.
    The numbers are the small fixtures' numbers: bf16 storage noise does not grow with the batch."""
    B = 128
    layers = R.IRESNET_LAYERS[arch]
    m, sd, _ = make_model(arch)
    x = R.closed_form_images(B).to(DEV)
    w = R.closed_form((B, 512), 0.37, 0.9, 1.0).to(DEV)
    res = {}
    for name, f32 in (("bf16", False), ("fp32", True)):
        m.load_state_dict(sd)
        m.validation_fp32 = f32
        m.eval()
        with torch.no_grad():
            fe = m(x).clone()
        m.train()
        for p_ in m.parameters():
            p_.grad = None
        ft = m(x)
        (ft * w).sum().backward()
        torch.cuda.synchronize()
        out = m.state_dict()
        res[name] = (fe, ft.detach().clone(), {k: p_.grad.detach().clone() for k, p_ in m.named_parameters() if p_.grad is not None},
                     {k: out[k].clone() for k in ("bn1.running_var", "layer3.10.bn2.running_mean", "bn2.running_var", "features.running_mean")})      # (both nets have a layer3.10)
        if f32:
            m._plans = {}                       # tens of GB of fp32 activations: release before the next test
    m.validation_fp32 = False
    a, b = res["bf16"], res["fp32"]
    e_eval, e_train = rel(a[0], b[0]), rel(a[1], b[1])
    gn = {k: float(v.norm()) for k, v in b[2].items()}
    gmax = max(gn.values())
    nerr = np.array([abs(float(a[2][k].norm()) - gn[k]) / gn[k] for k in gn if gn[k] > 1e-3 * gmax])
    derr = np.array([rel(a[2][k], b[2][k]) for k in gn if gn[k] > 1e-3 * gmax])
    stat = max(rel(a[3][k], b[3][k]) for k in a[3])
    print("MEASURED %s b128, 16-bit product path vs fp32 validation path: embeddings eval %.3e train %.3e; gradient norms median %.3e max %.3e; "
          "directions median %.3e max %.3e; running statistics %.3e" % (arch, e_eval, e_train, np.median(nerr), nerr.max(), np.median(derr), derr.max(), stat))
    # fp16: measured 2.0e-3 / 2.6e-3 (iresnet100), 1.5e-3 / 2.0e-3 (iresnet50); norms 1e-3 / 1.0e-2; directions 3.4e-2 / 6.0e-2; statistics 7e-4
    lim_e, lim_t = T16((SPEC, SPEC), EMB_TOL_BF16[arch])
    gt = grad_tol(arch)
    assert e_eval < lim_e and e_train < lim_t, (e_eval, e_train)
    assert np.median(nerr) < gt["norm_median"] and nerr.max() < gt["norm_max"], (np.median(nerr), nerr.max())
    assert np.median(derr) < gt["dir_median"] and derr.max() < T16(KINK_DIR_MAX, 0.5), (np.median(derr), derr.max())
    assert stat < T16(SPEC, STAT_TOL_BF16[1]), stat


@pytest.mark.parametrize("arch,batch", [("iresnet18", 128), ("iresnet50", 64)])
def test_eval_forward_fused_epilogues_match_separate_passes(arch, batch):
    """Eval-mode forward at batches where the LDS-DMA conv kernels run: BatchNorm (+PReLU, + identity, + the next block's bn1) applied in
    the conv epilogues (option eval_fuse, default) against the separate bn_apply passes.  The fused path keeps the conv output in fp32
    until after the affine (one bf16 rounding fewer) and rounds once more after the identity add: agreement at the bf16-storage level, and both
    against the fp32 oracle's embedding at the usual bf16-storage tolerance."""
    m, sd, layers = make_model(arch, tag=5.0)
    m.eval()
    x = R.closed_form_images(batch).to(DEV)
    outs = {}
    for fuse in (1, 0):
        _C.call("fedfr_set_option", b"eval_fuse", fuse)
        try:
            with torch.no_grad():
                outs[fuse] = m(x).float().clone()
            torch.cuda.synchronize()
        finally:
            _C.call("fedfr_set_option", b"eval_fuse", 1)
    assert torch.isfinite(outs[1]).all()
    # two bf16-storage pipelines with different rounding points: each sits ~1.2e-2 (iresnet50) from the fp32 embedding
    assert rel(outs[1], outs[0]) < 3e-2, rel(outs[1], outs[0])
    torch.set_num_threads(min(64, torch.get_num_threads()))
    with torch.no_grad():
        ref = R.iresnet_forward(sd, R.closed_form_images(min(batch, 16)), layers, training=False)
    ref = ref[0] if isinstance(ref, tuple) else ref
    assert rel(outs[1][: ref.shape[0]], ref) < 4e-2, rel(outs[1][: ref.shape[0]], ref)
    assert rel(outs[0][: ref.shape[0]], ref) < 4e-2


def test_heads_vs_reference():
    g = load_golden("heads")
    B, C = int(g["B"]), int(g["C"])
    x = R.closed_form((B, 512), 0.113, 0.2, 1.0).to(DEV)
    w = R.closed_form((C, 512), 0.071, 1.1, 0.01).to(DEV)
    lab, lab_m1 = T(g["labels"]).to(DEV), T(g["labels_m1"]).to(DEV)
    for nm, cls, s, m in (("cos", losses.CosFace, 30.0, 0.4), ("arc", losses.ArcFace, 30.0, 0.4),
                          ("cos64", losses.CosFace, 64.0, 0.4), ("arc64", losses.ArcFace, 64.0, 0.5)):
        xx = x.clone().requires_grad_(True)
        fcm = client.FC_module(512, C, "/tmp").to(DEV)
        fcm.fc.data = w.clone()
        cosine = fcm(xx)
        logits = cls(s=s, m=m)(cosine, lab)
        loss = ops.cross_entropy(logits, lab)
        loss.backward()
        # fp32 head: 1e-4-class agreement with the fp32 reference
        assert maxrel(cosine, g[nm + "_cosine"]) < 1e-5
        assert maxrel(logits, g[nm + "_logits"]) < 1e-5
        assert abs(float(loss) - float(g[nm + "_loss"])) < 1e-5 * max(1.0, abs(float(g[nm + "_loss"])))
        assert maxrel(xx.grad, g[nm + "_dx"]) < 1e-4
        assert maxrel(fcm.fc.grad, g[nm + "_dw"]) < 1e-4
        with torch.no_grad():
            assert maxrel(cls(s=s, m=m)(fcm(x), lab_m1), g[nm + "_logits_m1"]) < 1e-5
    with torch.no_grad():
        assert maxrel(fcm(x, normalize_feat=False), g["nonorm_cosine"]) < 1e-5


def test_bce_head_vs_reference():
    g = load_golden("bce")
    B, C = int(g["B"]), int(g["C"])
    x = R.closed_form((B, 512), 0.113, 0.2, 1.0).to(DEV).requires_grad_(True)
    mod = client.BCE_module(512, C, 1).to(DEV)
    mod.weight.data = R.closed_form((C, 512), 0.071, 1.1, 0.05).to(DEV)
    mod.bias.data = R.closed_form((C,), 0.5, 0.1, 0.1).to(DEV)
    mod.converter[0].weight.data = (torch.eye(512) + R.closed_form((512, 512), 0.013, 0.7, 0.01)).to(DEV)
    mod.converter[0].bias.data = R.closed_form((512,), 0.3, 0.2, 0.01).to(DEV)
    assert [k for k, _ in mod.state_dict().items()] == ["weight", "bias", "converter.0.weight", "converter.0.bias"]
    lab = T(g["labels"]).to(DEV)
    z, gt = mod(x, lab)
    loss = losses.BCE_loss()(z, gt)
    loss.backward()
    assert maxrel(z, g["z"]) < 1e-4
    assert bool((gt.cpu() == T(g["gt"])).all())
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    assert maxrel(x.grad, g["dx"]) < 1e-3
    assert maxrel(mod.weight.grad, g["d_weight"]) < 1e-3
    assert maxrel(mod.bias.grad, g["d_bias"]) < 1e-3
    assert maxrel(mod.converter[0].weight.grad[:8, :64], g["d_conv_w_slice"]) < 1e-3
    assert maxrel(mod.converter[0].bias.grad, g["d_conv_b"]) < 1e-3


def test_fedpavg_vs_reference_bit_exact():
    g = load_golden("fedavg")
    layers = R.IRESNET_LAYERS["iresnet18"]
    sizes = [int(v) for v in g["sizes"]]
    # generic path: plain dicts of GPU tensors (reference calling convention)
    models = []
    for i in range(3):
        sd = R.closed_form_state_dict(layers, tag=float(i + 1))
        models.append({k: v.to(DEV) for k, v in sd.items() if not k.startswith("fc.weight")})
    agg = server.FedPavg(models, sizes)
    for k in g.files:
        if k.startswith("agg_") and k[4:] in agg:
            a, b = agg[k[4:]].cpu(), T(g[k])
            assert a.dtype == b.dtype, k
            assert torch.equal(a, b), k
    assert sum(float(v.double().sum()) for v in agg.values()) == float(g["agg_checksum"])
    fcs = [R.closed_form((60, 512), 0.1 + 0.01 * i, 0.2 * i, 0.02).to(DEV) for i in range(3)]
    pre = R.closed_form((60, 512), 0.31, 0.5, 0.02).to(DEV)
    assert torch.equal(server.FedAvg_on_FC(pre, fcs, sizes, 1).cpu(), T(g["fc_p1"]))
    assert torch.equal(server.FedAvg_on_FC(pre, fcs, sizes, 0.5).cpu(), T(g["fc_p05"]))
    # flat fast path gives the same numbers and loads back into a model
    ms = []
    for i in range(3):
        m = backbones.iresnet18().to(DEV)
        m.load_state_dict(R.closed_form_state_dict(layers, tag=float(i + 1)))
        ms.append(client.flat_state_dict(m))
    flat = server.FedPavg(ms, sizes)
    for k in ("conv1.weight", "bn1.running_var", "layer2.0.downsample.0.weight", "layer4.1.prelu.weight", "fc.bias"):
        assert torch.equal(flat[k].cpu(), T(g["agg_" + k])), k
    assert flat["bn1.num_batches_tracked"].dtype == torch.float32          # F9
    tgt = backbones.iresnet18().to(DEV)
    tgt.load_state_dict(flat)
    assert int(tgt.state_dict()["bn1.num_batches_tracked"]) == int(float(g["agg_bn1.num_batches_tracked"]))


@pytest.mark.parametrize("name", ["pfc_w1_arc_r01", "pfc_w1_cos_r1", "pfc_w1_cos_r03"])
def test_partial_fc_w1_vs_reference(name):
    g = load_golden(name)
    B, C, rate = int(g["B"]), int(g["C"]), float(g["rate"])
    s, m, steps, mn = float(g["s"]), float(g["m"]), int(g["steps"]), str(g["margin"])
    margin = getattr(losses, mn)(s=s, m=m)
    pfc = PartialFC(rank=0, local_rank=0, world_size=1, batch_size=B, resume=False, margin_softmax=margin, num_classes=C,
                    sample_rate=rate, embedding_size=512, prefix="/tmp")
    num_local = C
    pfc.weight.copy_(R.closed_form((num_local, 512), 0.071, 1.1, 0.01).to(DEV))
    pfc.weight_mom.zero_()
    for st in range(steps):
        feats = F.normalize(R.closed_form((B, 512), 0.113 + 0.001 * st, 0.2 + st, 1.0)).to(DEV)
        lab = ((R.closed_form_labels(B, C, tag=st) * 31) % C).to(DEV)
        perm = R.closed_form((num_local,), 0.77 + 0.1 * st, 0.3, 0.5, 0.5).to(DEV)
        x_grad, loss_v = pfc.forward_backward(lab, feats, None, perm=perm)
        pre = "r0_s%d_" % st
        if (pre + "index") in g.files:
            assert torch.equal(pfc.index.cpu(), T(g[pre + "index"]))           # same sampled class set, bit exact
        assert maxrel(x_grad, g[pre + "x_grad"]) < 1e-4
        assert abs(float(loss_v) - float(g[pre + "loss_v"])) < 1e-4 * max(1.0, abs(float(g[pre + "loss_v"])))
        swg = pfc.sub_weight.grad
        assert maxrel(swg[:: max(1, swg.shape[0] // 48)][:48], g[pre + "sub_weight_grad_rows"]) < 1e-4
        assert maxrel(swg.norm(dim=1), g[pre + "sub_weight_grad_rownorm"]) < 1e-4
        pfc.fused_sgd_update(0.1, 0.9, 5e-4)
        assert maxrel(pfc.weight[:: max(1, num_local // 64)][:64], g[pre + "weight_rows"]) < 1e-5
        assert maxrel(pfc.weight_mom[:: max(1, num_local // 64)][:64], g[pre + "mom_rows"]) < 1e-4
        assert abs(float(pfc.weight.double().sum()) - float(g[pre + "weight_sum"])) < 1e-3


def _pfc_w2_gpu_worker(rank, port, q):
    """one rank of the class-sharded PartialFC (uneven shards 501/500, sample_rate 0.2) — both ranks share cuda:0, collectives go
    through gloo (RCCL refuses two ranks on one device); kernels, sampling and the update are the product's HIP path."""
    import traceback
    import torch.distributed as dist
    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=2)
        g = load_golden("pfc_w2")
        B, C, rate = int(g["B"]), int(g["C"]), float(g["rate"])
        s, m, steps, mn = float(g["s"]), float(g["m"]), int(g["steps"]), str(g["margin"])
        pfc = PartialFC(rank=rank, local_rank=0, world_size=2, batch_size=B, resume=False, margin_softmax=getattr(losses, mn)(s=s, m=m),
                        num_classes=C, sample_rate=rate, embedding_size=512, prefix="/tmp")
        num_local, class_start = R.pfc_shard(C, 2, rank)
        assert (pfc.num_local, pfc.class_start) == (num_local, class_start)
        pfc.weight.copy_(R.closed_form((num_local, 512), 0.071 + 0.003 * rank, 1.1, 0.01).to(DEV))
        pfc.weight_mom.zero_()
        for st in range(steps):
            feats = F.normalize(R.closed_form((B, 512), 0.113 + 0.01 * rank + 0.001 * st, 0.2 + st, 1.0)).to(DEV)
            lab = ((R.closed_form_labels(B, C, tag=st + 3 * rank) * 31 + rank) % C).to(DEV)
            perm = R.closed_form((num_local,), 0.77 + 0.1 * st, 0.3 + rank, 0.5, 0.5).to(DEV)
            x_grad, loss_v = pfc.forward_backward(lab, feats, None, perm=perm)
            pre = "r%d_s%d_" % (rank, st)
            assert torch.equal(pfc.index.cpu(), T(g[pre + "index"])), "sampled class set differs"
            assert maxrel(x_grad, g[pre + "x_grad"]) < 1e-4
            assert abs(float(loss_v) - float(g[pre + "loss_v"])) < 1e-4 * max(1.0, abs(float(g[pre + "loss_v"])))
            swg = pfc.sub_weight.grad
            assert maxrel(swg[:: max(1, swg.shape[0] // 48)][:48], g[pre + "sub_weight_grad_rows"]) < 1e-4
            assert maxrel(swg.norm(dim=1), g[pre + "sub_weight_grad_rownorm"]) < 1e-4
            pfc.fused_sgd_update(0.1, 0.9, 5e-4)
            assert maxrel(pfc.weight[:: max(1, num_local // 64)][:64], g[pre + "weight_rows"]) < 1e-5
            assert maxrel(pfc.weight_mom[:: max(1, num_local // 64)][:64], g[pre + "mom_rows"]) < 1e-4
            assert abs(float(pfc.weight.double().sum()) - float(g[pre + "weight_sum"])) < 1e-3
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:                                            # noqa: BLE001 — reported to the parent
        q.put((rank, traceback.format_exc()))


def test_partial_fc_w2_vs_reference():
    """world_size 2 product path against the reference captured over gloo with mp.spawn: all six exchange points
    (label/feature all-gather, max + 2 sum all-reduces, reduce-scatter) with uneven class shards and negative sampling."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pfc_w2_gpu_worker, args=(r, 29655, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == {0: "ok", 1: "ok"}, res


def test_partial_fc_config3_scale_properties():
    """BASELINE config 3 head at full size (85 000 classes, sample_rate 0.1 -> 8 500 sampled rows, ArcFace) driven by
    the fused trainer on iresnet18: size-independent properties of the sampled-softmax step."""
    C, B = 85000, 64
    m, sd, _ = make_model("iresnet18", tag=1.0)
    pfc = PartialFC(rank=0, local_rank=0, world_size=1, batch_size=B, resume=False, margin_softmax=losses.ArcFace(s=30, m=0.4),
                    num_classes=C, sample_rate=0.1, embedding_size=512, prefix="/tmp")
    w0, m0 = pfc.weight.clone(), pfc.weight_mom.clone()
    tr = client.FusedTrainer(m, pfc, "ArcFace", 30.0, 0.4, lr=0.01)
    lab = ((R.closed_form_labels(B, C, tag=1) * 977) % C).to(DEV)
    loss = tr.step(R.closed_form_images(B).to(DEV), lab)
    torch.cuda.synchronize()
    idx = pfc.index
    assert idx.numel() == 8500 and bool((idx[1:] > idx[:-1]).all())                 # sorted, unique (partial_fc.py:100)
    assert bool(torch.isin(lab, idx).all())                                        # every positive class is sampled (:98)
    assert np.isfinite(float(loss)) and 5.0 < float(loss) < 40.0      # ~ ln(8500) + s*(1 - cos(theta + m)) for random weights
    changed = (pfc.weight != w0).any(dim=1)
    assert bool(changed[idx].all()) and int(changed.sum()) == 8500                 # only sampled rows were updated (:113-116)
    assert bool(((pfc.weight_mom != m0).any(dim=1) == changed).all())
    # further steps draw different negative sets and keep the invariants (also under back-to-back asynchronous enqueueing)
    for st in range(6):
        tr.step(R.closed_form_images(B, tag=1.0 + st).to(DEV), lab)
    torch.cuda.synchronize()
    assert pfc.index.numel() == 8500 and not torch.equal(pfc.index, idx)
    assert bool(torch.isfinite(pfc.weight).all()) and float(pfc.weight.abs().max()) < 1.0


def test_cpu_tensor_is_rejected_loudly():
    m = backbones.iresnet18()
    with pytest.raises(RuntimeError, match="MI355X"):
        m(torch.zeros(2, 3, 112, 112))


def test_dropout_matches_oracle_with_same_mask():
    """dropout > 0 (reference iresnet.py:96,169; 0.4 for the webface configuration, client.py:142): nn.Dropout between bn2 and fc.
    The mask is counter-based (seed, step, index) — torch's RNG stream is not part of the contract — so parity = the oracle run with
    the HIP path's own mask injected: embeddings and gradients as close as without dropout; plus the mask's statistics and its
    determinism, eval-mode identity and the backward's use of the same mask."""
    import ctypes as C
    p, B, NC = 0.4, 8, 20
    layers = R.IRESNET_LAYERS["iresnet18"]
    sd = R.closed_form_state_dict(layers, tag=1.0)

    def run(seed):
        m = backbones.iresnet18(False, dropout=p, fp16=True)
        m.dropout_seed = seed
        m.load_state_dict(sd)
        m = m.to(DEV).train()
        x = R.closed_form_images(B).to(DEV)
        f = m(x)
        plan = m._plan(B)
        mask = plan.act[plan.mask_off: plan.mask_off + B * 25088].clone().view(B, 25088)
        (f * R.closed_form((B, 512), 0.37, 0.9, 1.0).to(DEV)).sum().backward()
        grads = {k: v.grad.clone() for k, v in m.named_parameters() if v.grad is not None}
        f2 = m(x)                                                 # second training forward: step 1 -> another mask
        mask2 = plan.act[plan.mask_off: plan.mask_off + B * 25088].clone().view(B, 25088)
        m.eval()
        with torch.no_grad():
            fe = m(x)
        return m, f.detach(), mask, grads, mask2, fe, f2.detach()
    m, f, mask, grads, mask2, fe, f2 = run(100)
    keep = float(mask.float().mean())
    assert set(mask.unique().tolist()) <= {0, 1} and abs(keep - (1 - p)) < 4 * (p * (1 - p) / mask.numel()) ** 0.5 + 1e-3, keep
    assert not torch.equal(mask, mask2) and abs(float(mask2.float().mean()) - (1 - p)) < 5e-3
    _, f_b, mask_b, _, _, _, _ = run(100)
    assert torch.equal(mask, mask_b) and torch.equal(f, f_b)       # same seed, same step -> same mask, same result
    _, _, mask_c, _, _, _, _ = run(101)
    assert not torch.equal(mask, mask_c)
    # oracle with the same mask (fp32) — forward and backward
    sdo = {k: v.clone() for k, v in sd.items()}
    keys = R.trainable_keys(sdo)
    for k in keys:
        sdo[k].requires_grad_(True)
    fo = R.iresnet_forward(sdo, R.closed_form_images(B), layers, True, dropout_p=p, dropout_mask=mask.cpu())
    (fo * R.closed_form((B, 512), 0.37, 0.9, 1.0)).sum().backward()
    assert rel(f, fo) < 2e-2, rel(f, fo)                              # bf16 backbone: the no-dropout level on this net is ~1e-2
    for k in ("fc.weight", "layer4.1.conv2.weight", "bn2.weight", "conv1.weight"):
        assert rel(grads[k], sdo[k].grad) < 0.1, (k, rel(grads[k], sdo[k].grad))
    # fc.weight's gradient is exactly zero in the dropped input columns of every image where they are dropped in all images
    dead = (mask.sum(dim=0) == 0).nonzero().flatten()
    if dead.numel():
        assert float(grads["fc.weight"][:, dead].abs().max()) == 0.0
    # eval mode: dropout is the identity (embeddings equal a dropout-free model's)
    m0 = backbones.iresnet18(False, dropout=0, fp16=True)
    m0.load_state_dict({k: v for k, v in m.state_dict().items()})
    m0 = m0.to(DEV).eval()
    with torch.no_grad():
        assert torch.equal(fe, m0(R.closed_form_images(B).to(DEV)))


def test_client_train_ragged_and_single_image_batches_vs_oracle():
    """Edge cases of the reference hot loop (client.py:536-551): a ragged last batch (a DataLoader without drop_last) and a batch of ONE
    image, which the reference duplicates before the forward pass (client.py:538-540: BatchNorm needs two samples) — through Client.train
    (batches of 5, 3 and 1 images: three plan sizes, the 2-plan cache evicts one) against the oracle's loop on the same data."""
    class Args:
        network, loss, local_epoch, output_dir, BCE_local, aggr_alg = "iresnet18", "CosFace", 1, "/tmp", False, "FedAvg"

    class DS:
        ID_base = 0

    class Loader(list):
        dataset = DS()
    C = 10
    batches = [(R.closed_form_images(n, tag=float(i)), R.closed_form_labels(n, C, tag=i)) for i, n in enumerate((5, 3, 1))]

    class Data:
        train_class_sizes, train_dataset_sizes, train_loaders = [C], [9], [Loader(batches)]
    from fedfr_amd.config import config as cfg
    saved = cfg.lr
    cfg.lr = 0.01
    try:
        cl = client.Client(0, Args, Data, device=DEV)
        layers = R.IRESNET_LAYERS["iresnet18"]
        cl.backbone_state_dict = R.closed_form_state_dict(layers, tag=2.0)
        cl.fc_module.fc.data = R.head_fc(C, seed=5)
        cl.train(0)
        lr_eff = cfg.lr_func(0) * cfg.lr
    finally:
        cfg.lr = saved
    sd = R.closed_form_state_dict(layers, tag=2.0)
    fc = R.head_fc(C, seed=5)
    losses_ref, sd, fc = R.client_train(sd, fc, batches, layers, "CosFace", 30.0, 0.4, lr_eff, 0.9, 5e-4)
    assert abs(cl.get_train_loss() - float(np.mean(losses_ref))) < 1e-2 * abs(float(np.mean(losses_ref))), (cl.get_train_loss(), losses_ref)
    out = cl.get_model()
    assert int(out["bn1.num_batches_tracked"]) == int(sd["bn1.num_batches_tracked"])        # three forward passes
    errs = {k: rel(out[k], sd[k]) for k in ("conv1.weight", "bn1.weight", "prelu.weight", "layer2.0.downsample.0.weight", "fc.bias",
                                            "bn1.running_mean", "bn1.running_var")}
    errs["head"] = rel(cl.fc_module.fc.data, fc)
    print("MEASURED ragged client loop:", {k: "%.2e" % v for k, v in errs.items()})
    # batches of 5 / 3 / 2 images: BatchNorm statistics over so few samples are ill-conditioned, the bf16 gradient noise is several times that
    # of the batch-8 client fixture (test_fused_client_loop_vs_reference: 1e-2) — measured 1.6e-2 on conv1.weight
    # (fp16: head 5.2e-3, conv1.weight 3.3e-3: at the spec)
    for k, v in errs.items():
        assert v < T16(SPEC, 3.3e-2 if k == "head" else 2.1e-2), (k, v)          # bf16 subset: measured 2.6e-2 / 1.6e-2


@pytest.mark.parametrize("arch,batch", [("iresnet18", 72), ("iresnet50", 40)])
def test_forward_moment_pass_equals_measured_statistics(arch, batch):
    """option fwd_xmom (default on): in the training forward pass of the 14x14 / 28x28 blocks conv2's epilogue leaves the raw moments
    (sum c2, sum c2 x, sum c2^2) and ONE pass writes out = bn3(c2) + x and the next block's bn1(out), whose statistics are derived
    (mean = sc mean(c2) + sh + mean(x), var = sc^2 var(c2) + var(x) + 2 sc cov) instead of measured on `out`.  The only thing the derived
    moments do not see is the rounding of `out` to bf16, so everything agrees with the measured-statistics path to bf16 noise: embeddings,
    running statistics of every BatchNorm, gradients."""
    outs = []
    for opt in (0, 1):
        with _C.option_scope("fwd_xmom", opt):
            m, sd, _ = make_model(arch, tag=2.0)
            m.train()
            x = R.closed_form_images(batch).to(DEV)
            for p_ in m.parameters():
                p_.grad = None
            f = m(x)
            (f * R.closed_form((batch, 512), 0.37, 0.9, 1.0).to(DEV)).sum().backward()
            outs.append((f.detach().clone(), {k: p_.grad.clone() for k, p_ in m.named_parameters() if p_.grad is not None},
                         {k: v.clone() for k, v in m.state_dict().items() if "running" in k}))
    (f0, g0, s0), (f1, g1, s1) = outs
    assert float((f0 - f1).abs().max() / f0.abs().max()) < 8e-3            # bf16 activations downstream of statistics that differ by ~1e-6
    for k in s0:
        tol = 3e-4                                                       # momentum 0.1 x statistics of bf16 activations that moved by a rounding here and there
        assert float((s0[k] - s1[k]).abs().max()) <= tol * (1.0 + float(s0[k].abs().max())), k
    worst = 0.0
    for k in g0:
        # d(bias) of a BatchNorm whose output only feeds another BatchNorm is a cancelling sum (exactly 0 in exact arithmetic): bf16 noise in
        # both runs, uncorrelated between them — skipped, as the reference comparison does (test_block_gpu.py)
        if k.endswith("bn3.bias") or k.endswith("downsample.1.bias") or k in ("bn2.bias", "fc.bias"):
            continue
        a, b = g0[k].double().flatten(), g1[k].double().flatten()
        worst = max(worst, float((a - b).norm() / (a.norm() + 1e-12)))
    assert worst < 0.1, worst                                            # measured 0.07 (a bn bias), conv weights 0.03: two bf16 evaluations of one step


def test_sphnet_trains_through_the_fused_trainer_like_iresnet():
    """Round 3: sphnet is a C++ plan behind the iresnet plan's entry points (csrc/net_sph.inc), so everything built on them takes it:
    FusedTrainer (one fedfr_net_forward + head + fedfr_net_backward2_sgd + flat SGD per step), the flat state for FedAvg, and
    `args.network = "sphnet"` in Client.train (the reference's run.sh configuration).  Checked against the eager path of the same
    model — reference-style step (client.py:543-549) + torch.optim.SGD — over 3 steps: same losses and weights to fp32 noise."""
    B, C_, lr = 8, 20, 0.01
    sd = R.sphere_state_dict(20, tag=1.0)
    x = [R.closed_form_images(B, tag=4.0 + i).to(DEV) for i in range(3)]
    lab = [R.closed_form_labels(B, C_, tag=i).to(DEV) for i in range(3)]
    # eager: autograd bridge + stock optimizer
    m0 = backbones.sphnet(False, dropout=0, fp16=True, type=20).to(DEV)
    m0.load_state_dict(sd)
    m0.train()
    fcm = client.FC_module(512, C_, "/tmp").to(DEV)
    fcm.fc.data = R.head_fc(C_).to(DEV)
    model = client.Sequential_model(m0, fcm)
    opt = torch.optim.SGD([{"params": m0.parameters()}, {"params": [fcm.fc]}], lr=lr, momentum=0.9, weight_decay=5e-4)
    l0 = []
    for i in range(3):
        opt.zero_grad()
        loss = ops.cross_entropy(losses.CosFace(s=30, m=0.4)(model(x[i]), lab[i]), lab[i])
        loss.backward()
        opt.step()
        m0.mark_weights_dirty()
        l0.append(float(loss))
    # fused
    m1 = backbones.sphnet(False, dropout=0, fp16=True, type=20).to(DEV)
    m1.load_state_dict(sd)
    fc1 = R.head_fc(C_).to(DEV)
    tr = client.FusedTrainer(m1, fc1, "CosFace", 30.0, 0.4, lr=lr, momentum=0.9, weight_decay=5e-4)
    l1 = [float(tr.step(x[i], lab[i])) for i in range(3)]
    tr.finish()
    assert all(abs(a - b) < 2e-3 * abs(a) for a, b in zip(l0, l1)), (l0, l1)
    s0, s1 = m0.state_dict(), m1.state_dict()
    worst = max(rel(s1[k], s0[k]) for k in s0)
    assert worst < 2e-3, worst
    assert rel(fc1, fcm.fc.data) < 2e-3
    # flat state: what FedPavg / load_state_dict exchange
    fsd = client.flat_state_dict(m1)
    agg = server.FedPavg([fsd, client.flat_state_dict(m0)], [1.0, 1.0])
    m2 = backbones.sphnet(False, type=20).to(DEV)
    m2.load_state_dict(agg)
    assert rel(m2.state_dict()["layer3.2.conv1.weight"], 0.5 * (s0["layer3.2.conv1.weight"] + s1["layer3.2.conv1.weight"])) < 1e-6


def test_fp16_overflow_guard_keeps_the_weights_finite():
    """fp16 storage (the product library): an absurd loss scale overflows every fp16 gradient of the backbone.  The update kernels skip the
    non-finite elements (fedfr_sgd_step_scaled), so parameters / momentum / mirrors stay finite; finish() reports the overflow and halves the
    DEVICE's scale (_C.LossScaleState: it outlives the trainer, so the next round starts from the lowered value — ADVICE r4); with a sane scale
    the same trainer trains on, and `growth_interval` clean steps bring the scale back up to its initial value."""
    if _C.storage_dtype() != torch.float16:
        pytest.skip("loss scaling exists on fp16 storage only (this process loaded the bf16 build)")
    state = _C.loss_scale_state(DEV)
    saved = (state.scale, state.clean_steps, state.growth_interval)
    try:
        _overflow_guard_body(state)
    finally:
        state.scale, state.clean_steps, state.growth_interval = saved
        state.word.zero_()


def _overflow_guard_body(state):
    B, C = 8, 40
    m, sd, layers = make_model("iresnet18", tag=5.0)
    fc = R.head_fc(C).to(DEV)
    tr = client.FusedTrainer(m, fc, "CosFace", 30.0, 0.4, lr=0.05, momentum=0.9, weight_decay=5e-4)
    before = m._flat_params.clone()
    tr.loss_scale = 2.0 ** 60
    imgs, lab = R.closed_form_images(B, tag=0.0).to(DEV), R.closed_form_labels(B, C, tag=0).to(DEV)
    tr.step(imgs, lab)
    with pytest.warns(UserWarning, match="non-finite gradients"):
        tr.finish()
    assert tr.overflows == 1 and tr.loss_scale == 2.0 ** 59
    assert bool(torch.isfinite(m._flat_params).all()) and bool(torch.isfinite(tr.mom).all())
    nt = m.trainable_count()
    skipped = float((m._flat_params[:nt] == before[:nt]).float().mean())
    assert skipped > 0.5, skipped                                # (almost) every backbone gradient overflowed: those elements were not touched
    # the lowered scale belongs to the device, not to the trainer: a NEW trainer (= the next FL round) starts from it
    tr2 = client.FusedTrainer(m, fc, "CosFace", 30.0, 0.4, lr=0.05, momentum=0.9, weight_decay=5e-4)
    assert tr2.loss_scale == 2.0 ** 59 and tr2.overflows == 0 and state.overflows >= 1
    tr2.loss_scale = state.initial / 2                            # (as if the back-offs had ended one step below the initial scale)
    state.growth_interval = 2
    mid = m._flat_params.clone()
    ls = [float(tr2.step(imgs, lab)) for _ in range(3)]
    tr2.finish()
    assert tr2.overflows == 0 and all(np.isfinite(ls)) and bool(torch.isfinite(m._flat_params).all())
    assert float((m._flat_params[:nt] != mid[:nt]).float().mean()) > 0.9          # ... and now it trains
    assert tr2.loss_scale == state.initial                         # three clean steps >= growth_interval: doubled once, never beyond the initial scale
    # forward_backward() + optimizer_step() (the gradients-then-update contract; FEDFR_FUSE_SGD=0 takes the same route): guarded as well
    tr3 = client.FusedTrainer(m, fc, "CosFace", 30.0, 0.4, lr=0.05, momentum=0.9, weight_decay=5e-4)
    tr3.loss_scale = 2.0 ** 60
    before3, fc_before = m._flat_params.clone(), fc.clone()
    tr3.forward_backward(imgs, lab)
    assert not bool(torch.isfinite(m._flat_grads[:nt]).all())      # the unscaled gradients the caller sees do carry the overflow
    tr3.optimizer_step()
    with pytest.warns(UserWarning, match="non-finite gradients"):
        assert tr3.check_overflow() is True
    tr3.finish()
    assert bool(torch.isfinite(m._flat_params).all()) and bool(torch.isfinite(tr3.mom).all()) and bool(torch.isfinite(fc).all())
    assert float((m._flat_params[:nt] == before3[:nt]).float().mean()) > 0.5
    # PartialFC head: its sampled-row update is guarded by the same word
    from fedfr_amd.partial_fc import PartialFC as _PFC
    pfc = _PFC(rank=0, local_rank=0, world_size=1, batch_size=B, resume=False, margin_softmax=losses.CosFace(s=30, m=0.4), num_classes=C,
               sample_rate=1.0, embedding_size=512, prefix="/tmp")
    tr4 = client.FusedTrainer(m, pfc, "CosFace", 30.0, 0.4, lr=0.05, momentum=0.9, weight_decay=5e-4)
    tr4.loss_scale = state.initial
    tr4.step(imgs, lab)
    tr4.finish()
    pfc.sub_weight.grad.fill_(float("inf"))                        # a poisoned row gradient: the guarded update must leave the rows alone
    rows = pfc.sub_weight.data.clone()
    pfc.fused_sgd_update(0.05, 0.9, 5e-4, overflow=state.word)
    torch.cuda.synchronize()
    assert torch.equal(pfc.sub_weight.data, rows) and int(state.word.item()) != 0
    state.word.zero_()


def test_bf16_build_reference_parity_subset():
    """The SECOND build of the library, libfedfr_hip_bf16.so (`make bf16`; csrc/common.h FEDFR_FP16=0: the same kernels on bf16 storage, no loss
    scale, ~1 % faster): the reference-parity tests of this file and the block fixtures in a child process that loads it, at the bounds 7
    mantissa bits allow (T16's second argument).  Its MEASURED lines are read back: whole-network embeddings sit at 1.2-2.6e-2 — outside
    north_star's 1e-2, which is why this build is not the default."""
    import re
    import subprocess
    import sys
    if _C.storage_dtype() != torch.float16:
        pytest.skip("this process already runs on the bf16 build")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert os.path.exists(os.path.join(root, "fedfr_amd", "libfedfr_hip_bf16.so")), "libfedfr_hip_bf16.so is not built (make -C fedfr_amd/csrc bf16)"
    env = dict(os.environ, FEDFR_HIP_LIB_NAME="libfedfr_hip_bf16.so")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_e2e_gpu.py"), os.path.join(root, "tests", "test_block_gpu.py"),
                        "-x", "-q", "-s", "-p", "no:cacheprovider", "-k",
                        "backbone_forward_vs_reference or train_step_grads_vs_reference or fused_client_loop or sgd_inside_backward or block_vs_reference "
                        "or without_the_prelu_kink or sphnet_vs_reference or full_size_step_invariants"],
                       env=env, capture_output=True, text=True, timeout=1200, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    emb = re.findall(r"MEASURED (iresnet\d+) embeddings: eval ([\d.e+-]+) train ([\d.e+-]+)", r.stdout)
    cos = re.findall(r"MEASURED (iresnet\d+) cosine ([\d.e+-]+) loss ([\d.e+-]+)", r.stdout)
    assert len(emb) == 2 and len(cos) == 2, r.stdout[-3000:]
    print("bf16 build:", emb, cos)
    assert all(float(ev) > 5e-3 for _, ev, _ in emb)          # (it really was the bf16 build: fp16 storage measures 1.5-2e-3 here)


# every process-global switch of the library that selects another code path or tuning value, with the alternative setting(s) worth keeping alive
# ("each is a code path the parity suite must keep alive").  Switches with tests of their own elsewhere are listed too: this is
# the ONE place that walks the whole option table (test_abi checks that the header documents it).
SWITCH_ALTERNATIVES = {
    "tn_use_tr": [0], "fuse_bnred_next": [0], "dgrad_parity": [1, 0], "wgrad_pair_reduce": [0], "nt_glds": [0, 12],
    "tn_glds": [0, 1], "wgrad9": [0], "fuse_bnbwd": [0, 1], "conv_c64p": [0], "bn_sliced": [0],
    "conv28_tpw2": [0, 1], "wgrad9p_bg": [0], "wgrad9p": [0],
    "fuse_bnbwd28": [0], "fwd_xmom": [0], "stem_bnred": [0], "stem_fuse_wgrad": [0], "c64p_bnbwd": [0],
    # (round 6 removed eight tuning switches whose alternative lost every sweep and validated nothing: tn_target_blocks, wgrad_depth, wgrad9_wgs,
    # bn_sliced_bwd_passes, bn_sliced_pre, event_nofence, fc_wgrad_aux, nt_nbuf)
    # not exercised by an iresnet training step: eval_fuse (eval-mode forward: test_eval_forward_fused_epilogues_match_separate_passes), sph_*
    # (sphnet: test_sphnet_options below)
    "eval_fuse": [], "sph_fuse_prelu_bwd": [], "sph_fin_multi": [], "sph_pair_wgrad": [],
}


def test_option_table_is_covered():
    """Every switch the loaded library has appears in SWITCH_ALTERNATIVES (a new switch must come with the setting that keeps its other path tested)."""
    assert set(_C.options()) == set(SWITCH_ALTERNATIVES), set(_C.options()) ^ set(SWITCH_ALTERNATIVES)


_SWITCH_CACHE = {}


def _switch_step(arch, B, C, **opts):
    """one training step (forward, CosFace, CE, dual-stream backward) with the plan created under the given switches"""
    import contextlib
    key = (arch, B, C)
    if key not in _SWITCH_CACHE:                         # closed-form weights / images are generated on the CPU: once per session
        _SWITCH_CACHE[key] = (R.closed_form_state_dict(R.IRESNET_LAYERS[arch], tag=4.0), R.head_fc(C), R.closed_form_images(B), R.closed_form_labels(B, C))
    if not opts and ("ref",) + key in _SWITCH_CACHE:
        return _SWITCH_CACHE[("ref",) + key]
    sd, fc0, x0, lab0 = _SWITCH_CACHE[key]
    with contextlib.ExitStack() as es:
        for k, v in opts.items():
            es.enter_context(_C.option_scope(k, v))
        m = getattr(backbones, arch)(False, dropout=0, fp16=True)
        m.load_state_dict(sd)
        m = m.to(DEV)
        fc = fc0.clone().to(DEV)
        tr = client.FusedTrainer(m, fc, "CosFace", 30.0, 0.4, lr=0.0)
        x = x0.to(DEV)
        lab = lab0.to(DEV)
        loss = float(tr.forward_backward(x, lab))
        tr.finish()
        torch.cuda.synchronize()
        out = (loss, m._flat_grads[: m.trainable_count()].clone(), tr.fc_grad.clone())
        m._plans = {}
        if not opts:
            _SWITCH_CACHE[("ref",) + key] = out
        return out


@pytest.mark.parametrize("name", sorted(k for k, v in SWITCH_ALTERNATIVES.items() if v))
def test_every_switch_alternative_matches_the_default(name):
    """iresnet18 at batch 128 (every map size of the step: 112 / 56 / 28 / 14 / 7, the LDS-DMA conv kernels, the persistent 64-channel kernel, paired
    and single nine-tap weight gradients, sliced and row-slab BatchNorm passes): one training step under each alternative setting of the switch
    against the default selection — the same loss and the same gradients up to what a different summation order or 16-bit rounding point may do
    (scheduling-only switches: the same bits)."""
    B, C = 128, 64
    ref = _switch_step("iresnet18", B, C)
    gn = float(ref[1].norm())
    for val in SWITCH_ALTERNATIVES[name]:
        got = _switch_step("iresnet18", B, C, **{name: val})
        dl, dg, df = abs(got[0] - ref[0]) / abs(ref[0]), float((got[1] - ref[1]).norm()) / gn, float((got[2] - ref[2]).norm() / ref[2].norm())
        print("switch %s = %d: loss %.2e, backbone gradients %.2e, head gradient %.2e" % (name, val, dl, dg, df))
        # (the bf16 build's storage noise is 8x the product's: a switch that moves one fp32 bit of a BatchNorm coefficient flips bf16 roundings through
        # the net — sliced vs row-slab passes differ by 4.6e-3 in the head gradient there, 2e-5 in the loss)
        assert np.isfinite(got[0]) and dl < T16(1e-3, 4e-3) and df < T16(2e-3, 1.2e-2), (name, val, dl, df)
        # kernels that round at other points (fused epilogues, derived statistics) move a 16-bit network by its storage noise; pure reorderings stay at 1e-4
        loose = name in ("fwd_xmom", "fuse_bnbwd", "fuse_bnbwd28", "c64p_bnbwd", "conv_c64p", "bn_sliced", "fuse_bnred_next", "stem_bnred", "nt_glds", "conv28_tpw2")
        assert dg < (2e-2 if loose else 1e-3), (name, val, dg)
        if name in ("wgrad9p_bg",):
            assert got[0] == ref[0] and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]), name


def test_sphnet_options():
    """the three sphnet switches (fused PReLU backward in the dgrad epilogue, one finalize launch for all PReLU layers, paired weight gradients):
    each alternative gives the default's embeddings bit for bit and its gradients to fp32 summation order / one 16-bit rounding point."""
    B = 16
    x = R.closed_form_images(B, tag=4.0).to(DEV)
    w = R.closed_form((B, 512), 0.37, 0.9, 1.0).to(DEV)

    def run(**opts):
        import contextlib
        with contextlib.ExitStack() as es:
            for k, v in opts.items():
                es.enter_context(_C.option_scope(k, v))
            net = backbones.sphnet(False, dropout=0, fp16=True, type=20).to(DEV)
            net.load_state_dict(R.sphere_state_dict(20, tag=1.0))
            net.train()
            f = net(x)
            (f * w).sum().backward()
            torch.cuda.synchronize()
            return f.detach().clone(), net._flat_grads[: net.trainable_count()].clone()
    f0, g0 = run()
    for name in ("sph_fuse_prelu_bwd", "sph_fin_multi", "sph_pair_wgrad"):
        f1, g1 = run(**{name: 0})
        d = float((g1 - g0).norm() / g0.norm())
        print("sphnet switch %s = 0: gradients %.2e" % (name, d))
        assert torch.equal(f1, f0) and d < (2e-2 if name == "sph_fuse_prelu_bwd" else 1e-3), (name, d)
