"""Pin the CPU restatement (oracle/ref_cpu.py) to golden vectors captured from the imported
reference by tools/make_golden.py.  CPU only; no reference import at test time."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from oracle import ref_cpu as R

torch.set_num_threads(8)


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach() if isinstance(a, torch.Tensor) else T(a)
    b = T(b)
    torch.testing.assert_close(a.to(b.dtype), b, rtol=rtol, atol=atol)


def test_spec_counts():
    # r50 475 keys, r100 925 keys (771 float + 154 int64)
    s50 = R.iresnet_spec(R.IRESNET_LAYERS["iresnet50"])
    s100 = R.iresnet_spec(R.IRESNET_LAYERS["iresnet100"])
    assert len(s50) == 475 and len(s100) == 925
    assert sum(1 for _, _, k in s100 if k == "bn_nbt") == 154
    n_par = sum(int(np.prod(s)) for k, s, kind in s100 if kind in ("conv", "bn_w", "bn_b", "prelu", "fc_w", "fc_b"))
    assert n_par == 65156160
    n_par = sum(int(np.prod(s)) for k, s, kind in s50 if kind in ("conv", "bn_w", "bn_b", "prelu", "fc_w", "fc_b"))
    assert n_par == 43590848


@pytest.mark.parametrize("arch,batch,fname", [("iresnet50", 8, "r50_b8"), ("iresnet100", 6, "r100_b6")])
def test_backbone_matches_reference(arch, batch, fname):
    g = load_golden(fname)
    layers = R.IRESNET_LAYERS[arch]
    C = int(g["num_classes"])
    sd = R.closed_form_state_dict(layers)
    x = R.closed_form_images(batch)
    lab = R.closed_form_labels(batch, C)
    with torch.no_grad():
        fe = R.iresnet_forward({k: v.clone() for k, v in sd.items()}, x, layers, training=False)
    close(fe, g["feat_eval"], 1e-4, 1e-5)
    fc = R.head_fc(C)
    feats, cosine, loss, grads, fcg = R.train_step_grads(sd, fc, x, lab, layers, "CosFace", 30.0, 0.4)
    close(feats, g["feat_train"], 1e-4, 1e-5)
    close(cosine, g["cosine"], 1e-4, 1e-6)
    assert abs(loss - float(g["loss"])) < 1e-5
    names = [str(n) for n in g["grad_names"]]
    assert names == R.trainable_keys(sd)
    norms = np.array([float(grads[k].norm()) for k in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-3, atol=1e-7)
    for k in g.files:
        if k.startswith("g_") and k[2:] in grads:
            close(grads[k[2:]], g[k], 2e-3, 1e-6)
    close(grads["layer3.1.conv1.weight"][:4, :16], g["g_layer3.1.conv1.weight_slice"], 2e-3, 1e-6)
    close(grads["fc.weight"][:4, :2048], g["g_fc.weight_slice"], 2e-3, 1e-7)
    close(fcg[:8], g["g_fc_head_rows"], 1e-4, 1e-7)
    for k in ("bn1", "layer1.0.bn1", "layer2.0.downsample.1", "layer4.2.bn3", "bn2", "features"):
        close(sd[k + ".running_mean"], g["rm_" + k], 1e-4, 1e-6)
        close(sd[k + ".running_var"], g["rv_" + k], 1e-4, 1e-6)
        assert int(sd[k + ".num_batches_tracked"]) == int(g["nbt_" + k])


def test_freeze_bn_matches_reference():
    """IResNet.freeze_BN(test_mode=True) (iresnet.py:140-147): BatchNorms in eval mode inside a training net — the oracle's bn_frozen
    forward and its autograd against the imported reference; model.train() afterwards is an ordinary training forward again."""
    g = load_golden("freeze_bn_r18")
    layers = R.IRESNET_LAYERS["iresnet18"]
    B = int(g["B"])
    sd = R.closed_form_state_dict(layers, tag=7.0)
    x = R.closed_form_images(B, tag=3.0)
    w = R.closed_form((B, 512), 0.37, 0.9, 1.0)
    keys = R.trainable_keys(sd)
    work = {k: (v.clone().requires_grad_(True) if k in keys else v.clone()) for k, v in sd.items()}
    feats = R.iresnet_forward(work, x, layers, training=True, bn_frozen=True)
    close(feats, g["feats"], 1e-4, 1e-5)
    (feats * w).sum().backward()
    names = [str(n) for n in g["grad_names"]]
    assert names == keys
    np.testing.assert_allclose(np.array([float(work[k].grad.norm()) for k in names]), g["grad_norms"], rtol=2e-3, atol=1e-7)
    for k in g.files:
        if k.startswith("g_") and k[2:] in work:
            close(work[k[2:]].grad, g[k], 2e-3, 1e-6)
    close(work["layer3.1.conv1.weight"].grad[:4, :16], g["g_layer3.1.conv1.weight_slice"], 2e-3, 1e-6)
    for k in ("bn1", "layer2.0.downsample.1", "layer4.1.bn3", "bn2", "features"):      # nothing tracked
        assert torch.equal(work[k + ".running_mean"], sd[k + ".running_mean"]) and int(work[k + ".num_batches_tracked"]) == int(g["nbt_" + k])
        close(work[k + ".running_mean"], g["rm_" + k], 1e-6, 1e-7)
    with torch.no_grad():
        again = R.iresnet_forward({k: v.detach().clone() for k, v in work.items()}, x, layers, training=True)
    close(again, g["feats_after_train_call"], 1e-4, 1e-5)


def test_heads_match_reference():
    g = load_golden("heads")
    B, C = int(g["B"]), int(g["C"])
    x = R.closed_form((B, 512), 0.113, 0.2, 1.0)
    w = R.closed_form((C, 512), 0.071, 1.1, 0.01)
    lab, lab_m1 = T(g["labels"]), T(g["labels_m1"])
    for nm, fn, s, m in (("cos", R.cosface, 30.0, 0.4), ("arc", R.arcface, 30.0, 0.4),
                         ("cos64", R.cosface, 64.0, 0.4), ("arc64", R.arcface, 64.0, 0.5)):
        xx = x.clone().requires_grad_(True)
        ww = w.clone().requires_grad_(True)
        cosine = R.fc_module_forward(xx, ww)
        logits = fn(cosine.clone(), lab, s, m)
        loss = F.cross_entropy(logits, lab)
        loss.backward()
        close(cosine, g[nm + "_cosine"])
        close(logits, g[nm + "_logits"], 1e-5, 1e-5)
        close(loss, g[nm + "_loss"])
        close(xx.grad, g[nm + "_dx"], 1e-4, 1e-7)
        close(ww.grad, g[nm + "_dw"], 1e-4, 1e-6)
        close(fn(R.fc_module_forward(x, w), lab_m1, s, m), g[nm + "_logits_m1"], 1e-5, 1e-5)
    close(R.fc_module_forward(x, w, normalize_feat=False), g["nonorm_cosine"])


def test_bce_matches_reference():
    g = load_golden("bce")
    B, C = int(g["B"]), int(g["C"])
    x = R.closed_form((B, 512), 0.113, 0.2, 1.0).requires_grad_(True)
    weight = R.closed_form((C, 512), 0.071, 1.1, 0.05).requires_grad_(True)
    bias = R.closed_form((C,), 0.5, 0.1, 0.1).requires_grad_(True)
    cw = (torch.eye(512) + R.closed_form((512, 512), 0.013, 0.7, 0.01)).requires_grad_(True)
    cb = R.closed_form((512,), 0.3, 0.2, 0.01).requires_grad_(True)
    z, gt = R.bce_module_forward(x, T(g["labels"]), cw, cb, weight, bias)
    loss = R.bce_loss(z, gt)
    loss.backward()
    close(z, g["z"], 1e-4, 1e-5)
    assert bool((gt == T(g["gt"])).all())
    close(loss, g["loss"])
    close(x.grad, g["dx"], 1e-4, 1e-7)
    close(weight.grad, g["d_weight"], 1e-4, 1e-6)
    close(bias.grad, g["d_bias"], 1e-4, 1e-7)
    close(cw.grad[:8, :64], g["d_conv_w_slice"], 1e-4, 1e-7)
    close(cb.grad, g["d_conv_b"], 1e-4, 1e-7)


def test_sgd_matches_reference():
    g = load_golden("sgd")
    ps = [R.closed_form(s, 0.2 + 0.1 * i, 0.3 * i, 0.5) for i, s in enumerate([(7, 5), (33,), (4, 3, 3, 3)])]
    bufs = [None] * 3
    for step in range(3):
        grads = [R.closed_form(tuple(p.shape), 0.15 + 0.05 * i + 0.01 * step, 0.7 * step, 0.3)
                 for i, p in enumerate(ps)]
        R.sgd_step(ps, grads, bufs, 0.1, 0.9, 5e-4)
        for i in range(3):
            close(ps[i], g["p%d_s%d" % (i, step)], 1e-6, 1e-7)
            close(bufs[i], g["m%d_s%d" % (i, step)], 1e-6, 1e-7)


def test_fedavg_matches_reference():
    g = load_golden("fedavg")
    layers = R.IRESNET_LAYERS["iresnet18"]
    sizes = [int(v) for v in g["sizes"]]
    models = []
    for i in range(3):
        sd = R.closed_form_state_dict(layers, tag=float(i + 1))
        models.append({k: v for k, v in sd.items() if not k.startswith("fc.weight")})
    agg = R.fedpavg(models, sizes)
    for k in g.files:
        if k.startswith("agg_") and k[4:] in agg:
            a, b = agg[k[4:]], T(g[k])
            assert a.dtype == b.dtype, k          # F9: int64 nbt becomes float32
            assert torch.equal(a, b), k           # same op order => bit exact
    tot = sum(float(v.double().sum()) for v in agg.values())
    assert tot == float(g["agg_checksum"])
    fcs = [R.closed_form((60, 512), 0.1 + 0.01 * i, 0.2 * i, 0.02) for i in range(3)]
    pre = R.closed_form((60, 512), 0.31, 0.5, 0.02)
    assert torch.equal(R.fedavg_on_fc(pre, fcs, sizes, 1), T(g["fc_p1"]))
    assert torch.equal(R.fedavg_on_fc(pre, fcs, sizes, 0.5), T(g["fc_p05"]))


def _pfc_inputs(B, C, rank, st, num_local):
    feats = F.normalize(R.closed_form((B, 512), 0.113 + 0.01 * rank + 0.001 * st, 0.2 + st, 1.0))
    lab = (R.closed_form_labels(B, C, tag=st + 3 * rank) * 31 + rank) % C
    perm = R.closed_form((num_local,), 0.77 + 0.1 * st, 0.3 + rank, 0.5, 0.5)
    return feats, lab, perm


def _pfc_check(g, comm, rank, world):
    B, C, rate = int(g["B"]), int(g["C"]), float(g["rate"])
    s, m, steps, mn = float(g["s"]), float(g["m"]), int(g["steps"]), str(g["margin"])
    num_local, _ = R.pfc_shard(C, world, rank)
    weight = R.closed_form((num_local, 512), 0.071 + 0.003 * rank, 1.1, 0.01)
    mom = torch.zeros_like(weight)
    for st in range(steps):
        feats, lab, perm = _pfc_inputs(B, C, rank, st, num_local)
        r = R.pfc_forward_backward(lab, feats, weight, mom, comm, B, C, rate, mn, s, m, perm)
        pre = "r%d_s%d_" % (rank, st)
        close(r["x_grad"], g[pre + "x_grad"], 1e-4, 1e-6)
        close(r["loss_v"], g[pre + "loss_v"], 1e-5, 1e-6)
        if (pre + "index") in g.files:
            assert torch.equal(r["index"], T(g[pre + "index"]))
        swg = r["sub_weight_grad"]
        gr = int(g["grad_rows"]) if "grad_rows" in g.files else 48
        wr = int(g["w_rows"]) if "w_rows" in g.files else 64
        close(swg[:: max(1, swg.shape[0] // gr)][:gr], g[pre + "sub_weight_grad_rows"], 1e-4, 1e-6)
        close(swg.norm(dim=1), g[pre + "sub_weight_grad_rownorm"], 1e-4, 1e-6)
        R.pfc_sgd_update(weight, mom, r["index"], swg, 0.1, 0.9, 5e-4)
        close(weight[:: max(1, num_local // wr)][:wr], g[pre + "weight_rows"], 1e-5, 1e-7)
        close(mom[:: max(1, num_local // wr)][:wr], g[pre + "mom_rows"], 1e-4, 1e-6)
        assert abs(float(weight.double().sum()) - float(g[pre + "weight_sum"])) < 1e-3
        assert abs(float(mom.double().sum()) - float(g[pre + "mom_sum"])) < 1e-4


@pytest.mark.parametrize("name", ["pfc_w1_arc_r01", "pfc_w1_cos_r1", "pfc_w1_cos_r03"])
def test_partial_fc_w1_matches_reference(name):
    _pfc_check(load_golden(name), R.SingleRankComm(), 0, 1)


def _pfc_w2_worker(rank, port):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        _pfc_check(load_golden("pfc_w2"), R.DistComm(), rank, 2)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_partial_fc_w2_gloo_matches_reference():
    import torch.multiprocessing as mp
    mp.spawn(_pfc_w2_worker, args=(29633,), nprocs=2, join=True)


class _OracleThreadComm:
    """the oracle's four verbs on top of fedfr_amd.comm.ThreadComm (W simulated ranks = W threads of this process)."""

    def __init__(self, comm):
        self.c, self.world_size, self.rank = comm, comm.world_size, comm.rank

    def all_gather(self, t):
        return self.c.all_gather(t)

    def all_reduce_max(self, t):
        return self.c.all_reduce(t.clone(), "max")

    def all_reduce_sum(self, t):
        return self.c.all_reduce(t.clone(), "sum")

    def reduce_scatter_sum(self, t):
        return self.c.reduce_scatter(t)


@pytest.mark.parametrize("name,world", [("pfc_w4", 4), ("pfc_w8", 8)])
def test_partial_fc_w4_w8_match_reference(name, world):
    """world_size 4 (uneven shards 501/501/501/500) and world_size 8 at BASELINE config 5's shard geometry (85 003 classes: 10 626 /
    10 625 per rank, 1 062 sampled) against the reference captured over gloo with 4 / 8 processes; here the ranks are threads."""
    from fedfr_amd.comm import ThreadComm
    g = load_golden(name)
    torch.set_num_threads(1)
    try:
        ThreadComm.run(world, lambda c: _pfc_check(g, _OracleThreadComm(c), c.rank, world))
    finally:
        torch.set_num_threads(8)


def test_client_loop_matches_reference():
    g = load_golden("client_r18")
    layers = R.IRESNET_LAYERS["iresnet18"]
    B, C, steps = int(g["B"]), int(g["C"]), int(g["steps"])
    sd = R.closed_form_state_dict(layers, tag=2.0)
    fc = R.head_fc(C)
    batches = [(R.closed_form_images(B, tag=float(st)), R.closed_form_labels(B, C, tag=st)) for st in range(steps)]
    losses, sd, fc = R.client_train(sd, fc, batches, layers, "CosFace", 30.0, 0.4, float(g["lr"]), 0.9, 5e-4)
    np.testing.assert_allclose(np.array(losses), g["losses"], rtol=2e-4)
    for k in g.files:
        if k.startswith("sd_") and k[3:] in sd:
            close(sd[k[3:]], g[k], 5e-3, 2e-4)      # 3 SGD steps: fp32 thread-order noise compounds
    close(sd["fc.weight"][:4, :2048], g["sd_fc.weight_slice"], 5e-3, 2e-4)
    close(fc, g["head_fc"], 5e-3, 2e-4)


@pytest.mark.parametrize("variant", ["full", "seq", "bce_rw"])
def test_public_data_loop_matches_reference(variant):
    """train_with_public_data body (client.py:354-441): Branch_model + 10*BCE + mu*contrastive / plain Sequential /
    BCE + reweight_cosface (detached CosFace term), captured from the imported reference modules."""
    g = load_golden("client_public_" + variant)
    layers = R.IRESNET_LAYERS["iresnet18"]
    sd, fc, bce, batches, kw = R.public_fixture_state(g, variant)
    rows, sd, fc, bce = R.client_train_public(sd, fc, bce, batches, layers, **kw)
    got = np.array([[np.nan if v is None else v for v in r] for r in rows])
    np.testing.assert_allclose(got, g["rows"], rtol=3e-4, equal_nan=True)
    for k in g.files:
        if k.startswith("sd_") and k[3:] in sd:
            close(sd[k[3:]], g[k], 5e-3, 2e-4)
    close(fc, g["head_fc"], 5e-3, 2e-4)
    if bce is not None:
        close(bce["weight"], g["bce_weight"], 5e-3, 2e-4)
        close(bce["bias"], g["bce_bias"], 5e-3, 2e-4)
        close(bce["converter.0.weight"][:8, :64], g["bce_conv_w_slice"], 5e-3, 2e-4)
        close(bce["converter.0.bias"], g["bce_conv_b"], 5e-3, 2e-4)


def test_sweeps_and_hard_negative_mining_match_reference():
    """data_update_fc class centres (client.py:159-188), Generate_pretrain_feats / Initialize_pretrain_FC
    (server.py:182-263) and choose_hard_negative_2's index set (client.py:191-236), captured from the imported reference modules."""
    g = load_golden("mining_r18")
    layers = R.IRESNET_LAYERS["iresnet18"]
    sd, local, public = R.mining_fixture_state(g)
    for nba in (True, False):
        close(R.class_centers(sd, local, layers, int(g["n_local"]), nba), g["local_centers_nba%d" % int(nba)], 2e-4, 1e-5)
    pub = R.embed(sd, public, layers)
    close(pub, g["public_feats"], 2e-4, 1e-6)
    close(R.class_centers(sd, public, layers, int(g["n_public"]), True), g["public_centers"], 2e-4, 1e-6)
    loc = R.embed(sd, local, layers)
    close(loc @ pub.t(), g["similarity"], 2e-4, 1e-5)
    idx = R.hard_negative_indices(loc, pub, float(g["hn_threshold"]))
    assert torch.equal(idx, T(g["hn_index"]))
    assert 0 < len(idx) < pub.shape[0] and float(g["hn_gap"]) > 0.02        # a non-trivial subset, away from the threshold


def test_roc_histogram_matches_reference_kernel_body():
    """the oracle's pair histogram / TPR read-out against roc_cuda.py's own calc_ROC body and plot_ROC, executed on
    the CPU through a numba stub (tools/make_golden.py gen_roc), plus a known-answer case."""
    g = load_golden("roc")
    hist = R.roc_histogram(g["features"], g["labels"], int(g["target_size"]))
    assert np.array_equal(hist, g["hist"])
    T_, N_ = int(g["target_size"]), len(g["labels"])
    assert int(hist.sum()) == int(g["total_pairs"]) == T_ * (T_ - 1) // 2 + T_ * (N_ - T_)
    assert R.roc_tpr_at_fpr(hist) == [float(v) for v in g["tpr"]]
    # known answer: one-hot features -> dot is exactly 0 or 1 -> bins 1000 / 2000
    f = np.eye(4, dtype=np.float32)[[0, 0, 1, 2, 1]]
    h = R.roc_histogram(f, np.array([7, 7, 8, 9, 8]), 3)
    assert h[2000, 0] == 2 and h[1000, 1] == 7 and h.sum() == 9


@pytest.mark.parametrize("type_", [20, 64])
def test_sphnet_matches_reference(type_):
    """sphere20 / sphere64 (the reference's default, sphnet.py:72, and what run.sh trains) forward + parameter gradients of
    the functional restatement vs the imported reference module."""
    g = load_golden("sphnet%d" % type_)
    B = int(g["B"])
    sd = R.sphere_state_dict(type_, tag=1.0)
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    x = R.closed_form_images(B, tag=4.0)
    dfe = R.closed_form((B, 512), 0.37, 0.9, 1.0)
    feats, grads = R.sphere_step_grads(sd, x, dfe, type_)
    close(feats, g["feats"], 2e-4, 1e-5)
    for k in sd:
        assert abs(float(grads[k].norm()) - float(g["gnorm_" + k])) < 2e-4 * float(g["gnorm_" + k]) + 1e-6, k
        if ("g_" + k) in g.files:
            close(grads[k], g["g_" + k], 2e-3, 2e-4 * float(T(g["g_" + k]).abs().max()) + 1e-7)
    for key in [f for f in g.files if f.endswith("_slice")]:          # slices of large tensors: "g_<param>_slice" = grad[:a, :b]
        ref = T(g[key])
        close(grads[key[2:-6]][: ref.shape[0], : ref.shape[1]], ref, 2e-3, 2e-4 * float(ref.abs().max()) + 1e-7)


# ---- IBasicBlock fixtures (tests/golden/block.npz; reference backbones/iresnet.py:28-57) -------------------------------------------
def _block_rel(a, ref, ref_norm=None):
    a = R.fixture_sample(a).double()
    r = T(ref).double().reshape(-1)
    return float((a - r).norm() / (r.norm() + 1e-30))


# gradients that sit BEHIND the PReLU derivative in the block's backward pass (bn2's bias gradient sums the masked gradient itself)
POST_MASK = ("dx", "g_bn1.weight", "g_bn1.bias", "g_conv1.weight", "g_bn2.bias")


@pytest.mark.parametrize("name", sorted(R.BLOCK_FIXTURES))
def test_block_matches_reference(name):
    """the restated block (ibasic_block) reproduces the reference block's y, dx, every parameter gradient and the BN buffers."""
    g = load_golden("block")
    y, dx, grads, sd = R.block_fixture_run(name)
    assert _block_rel(y, g[name + "_y"]) < 1e-6 and _block_rel(dx, g[name + "_dx"]) < 1e-6
    assert abs(float(y.double().norm()) - float(g[name + "_y_norm"])) < 1e-6 * float(g[name + "_y_norm"])
    for k, v in grads.items():
        assert _block_rel(v, g[name + "_g_" + k]) < 1e-5, k
        assert abs(float(v.double().norm()) - float(g[name + "_gn_" + k])) <= 1e-5 * float(g[name + "_gn_" + k]) + 1e-12, k
    for k, v in sd.items():
        if "running" in k or "tracked" in k:
            close(v, g[name + "_b_" + k])


@pytest.mark.parametrize("name", sorted(R.BLOCK_FIXTURES))
def test_block_bf16_storage_floor(name):
    """What bf16 STORAGE alone (fp32 arithmetic, oracle/bf16_emul.py) does to one block against the fp32 reference — the floor any
    bf16 implementation sits on.  Without the PReLU kink ("_lin": slope 1) every output is inside north_star's 1e-2; with the real
    slopes the outputs in front of the kink are too, and the gradients behind it are NOT (2-6e-2): a rounding of the PReLU input flips
    the derivative of the elements next to zero.  tests/test_e2e_gpu.py holds the HIP path to the same split."""
    from oracle import bf16_emul as E
    g = load_golden("block")
    lin = R.BLOCK_FIXTURES[name][5]
    y, dx, grads, _ = R.block_fixture_run(name, lambda sd, p, x, s, t: E.block(sd, p, E.q(x), s, t))
    errs = {"y": _block_rel(y, g[name + "_y"]), "dx": _block_rel(dx, g[name + "_dx"])}
    for k, v in grads.items():
        if float(g[name + "_gn_" + k]) > 1e-6 * max(float(g[name + "_gn_" + kk]) for kk in grads):    # bn3 / downsample.1 bias: exactly zero
            errs["g_" + k] = _block_rel(v, g[name + "_g_" + k])
    front = {k: e for k, e in errs.items() if k not in POST_MASK}
    behind = {k: e for k, e in errs.items() if k in POST_MASK}
    assert max(front.values()) < 1e-2, front
    if lin:
        assert max(behind.values()) < 1e-2, behind
    else:
        assert 1e-2 < max(behind.values()) < 8e-2, behind


def test_prelu_hook_is_transparent_and_injects_a_sign_pattern():
    """oracle.ref_cpu's ``prelu_hook`` (round 6; what tests/test_e2e_gpu.py::test_prelu_kink_law_and_mask_injected_gradients injects the HIP path's
    PReLU sign pattern through): a hook that differentiates with the layer's OWN sign pattern reproduces the plain oracle's gradients; a hook with a
    pattern that differs in a fraction f of the elements moves the gradients behind that PReLU by about (1 - slope) sqrt(f) and leaves the forward
    values and everything in front of the first flipped layer untouched."""
    class _Masked(torch.autograd.Function):
        @staticmethod
        def forward(ctx, z, w, pos):
            ctx.save_for_backward(z, w, pos)
            return F.prelu(z, w)

        @staticmethod
        def backward(ctx, dy):
            z, w, pos = ctx.saved_tensors
            return torch.where(pos, dy, dy * w.view(1, -1, 1, 1)), torch.where(pos, torch.zeros_like(dy), dy * z).sum(dim=(0, 2, 3)), None
    layers = R.IRESNET_LAYERS["iresnet18"]
    sd = R.closed_form_state_dict(layers)
    fc0, x, lab = R.head_fc(16), R.closed_form_images(2), R.closed_form_labels(2, 16)
    run = lambda hook: R.train_step_grads({k: v.clone() for k, v in sd.items()}, fc0.clone(), x, lab, layers, prelu_hook=hook)   # noqa: E731
    _, c0, l0, g0, _ = run(None)
    _, c1, l1, g1, _ = run(lambda name, z, w: _Masked.apply(z, w, z.detach() > 0))
    assert torch.equal(c0, c1) and l0 == l1
    for k in g0:
        assert float((g0[k] - g1[k]).norm()) <= 1e-5 * float(g0[k].norm()) + 1e-12, k
    gen = torch.Generator().manual_seed(3)
    flips = {}

    def flipped(name, z, w):
        pos = z.detach() > 0
        if name == "layer3.1.prelu":                       # flip 1 % of this ONE layer's pattern
            f = torch.rand(pos.shape, generator=gen) < 0.01
            pos = pos ^ f
            flips[name] = (float(f.float().mean()), float(w.detach().mean()))
        return _Masked.apply(z, w, pos)
    _, c2, l2, g2, _ = run(flipped)
    assert torch.equal(c0, c2) and l0 == l2                # forward values do not depend on the injected pattern
    f, slope = flips["layer3.1.prelu"]
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))   # noqa: E731
    assert rel(g2["layer4.0.conv1.weight"], g0["layer4.0.conv1.weight"]) < 1e-6          # in front of the flipped layer (backward order): untouched
    e = rel(g2["layer3.1.conv1.weight"], g0["layer3.1.conv1.weight"])                    # right behind it
    pred = (1.0 - slope) * f ** 0.5
    assert 0.2 * pred < e < 2.0 * pred, (e, pred)
