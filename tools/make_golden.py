#!/usr/bin/env python3
"""Generate tests/golden/*.npz by driving the IMPORTED reference (/root/reference).

Runs only in the build container (the reference never travels to the GPU box).
Inputs/weights are closed-form (oracle.ref_cpu.closed_form*), so fixtures hold only
outputs + the few generator parameters.  Re-run:  python tools/make_golden.py

What is captured (reference symbol → fixture):
  backbones.iresnet.IBasicBlock           → block.npz
  backbones.iresnet50 / iresnet100        → r50_b8.npz / r100_b6.npz  (train fwd+bwd, eval fwd)
  client.FC_module, losses.CosFace/ArcFace, F.cross_entropy → heads.npz
  client.BCE_module, losses.BCE_loss      → bce.npz
  torch.optim.SGD (momentum .9, wd 5e-4)  → sgd.npz
  server.FedPavg / FedAvg_on_FC           → fedavg.npz
  partial_fc.PartialFC (W=1, and W=2 over gloo) → pfc_w1_*.npz / pfc_w2.npz
  a 3-step Client.train-equivalent loop on iresnet18 → client_r18.npz
  roc_cuda.calc_ROC / plot_ROC (kernel body executed through a numba stub) → roc.npz
  eval-mode embedding sweeps, class centres and feature-based hard-negative mining → mining_r18.npz
  the train_with_public_data loop body (Branch_model + BCE + contrastive; Sequential + reweight) → client_public_{full,seq}.npz
"""
import os
import sys
import types
import contextlib

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch
import torch.nn.functional as F

torch.set_num_threads(8)


# ---- stubs for absent third-party modules -------------------------------
class _ED(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


_m = types.ModuleType("easydict")
_m.EasyDict = _ED
sys.modules["easydict"] = _m
for _n in ("mxnet", "cv2", "prettytable", "torchvision", "torchvision.transforms"):
    sys.modules[_n] = types.ModuleType(_n)
sys.modules["mxnet"].ndarray = types.ModuleType("nd")
sys.modules["mxnet.ndarray"] = sys.modules["mxnet"].ndarray
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.modules["prettytable"].PrettyTable = object

import warnings
warnings.filterwarnings("ignore")
import backbones          # noqa: E402  (reference)
import losses             # noqa: E402
import client             # noqa: E402
import server             # noqa: E402

from oracle import ref_cpu as R   # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def save(name, **arrs):
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print("wrote %-22s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def load_closed_form(model, layers, tag=0.0):
    sd = R.closed_form_state_dict(layers, tag=tag)
    ref_keys = list(model.state_dict().keys())
    assert ref_keys == list(sd.keys()), "spec order differs from reference state_dict order"
    model.load_state_dict(sd)
    return sd


# ---- 1. IBasicBlock -------------------------------------------------------------------------
def gen_block():
    """reference IBasicBlock (backbones/iresnet.py:28-57) fwd + bwd on the RNG-free inputs of oracle.ref_cpu.block_fixture."""
    from backbones.iresnet import IBasicBlock, conv1x1
    out = {}
    for name in R.BLOCK_FIXTURES:
        cin, cout, stride, hw, batch, lin = R.BLOCK_FIXTURES[name]
        sd0, x, gy, _ = R.block_fixture(name)
        ds = None
        if stride != 1 or cin != cout:
            ds = torch.nn.Sequential(conv1x1(cin, cout, stride), torch.nn.BatchNorm2d(cout, eps=1e-5))
        blk = IBasicBlock(cin, cout, stride, ds)
        blk.train()
        assert list(blk.state_dict().keys()) == list(sd0.keys()), "block spec order differs from the reference's state_dict order"
        blk.load_state_dict(sd0)
        x = x.clone().requires_grad_(True)
        y = blk(x)
        y.backward(gy)
        for key, t in (("y", y), ("dx", x.grad)):
            out["%s_%s" % (name, key)] = R.fixture_sample(t)
            out["%s_%s_norm" % (name, key)] = t.detach().double().norm()
        for k, p in blk.named_parameters():
            out[name + "_g_" + k] = R.fixture_sample(p.grad)
            out[name + "_gn_" + k] = p.grad.double().norm()
        for k, b in blk.named_buffers():
            out[name + "_b_" + k] = b
    save("block", **out)


# ---- 2. full backbones ------------------------------------------------------------------------
def gen_backbone(arch, layers, batch, fname, num_classes=1000):
    model = getattr(backbones, arch)(False, dropout=0, fp16=False)
    load_closed_form(model, layers)
    fcm = client.FC_module(512, num_classes, "/tmp")
    fcm.fc.data = R.head_fc(num_classes)
    x = R.closed_form_images(batch)
    lab = R.closed_form_labels(batch, num_classes)
    out = {"batch": batch, "num_classes": num_classes}
    # eval-mode forward first (does not touch buffers)
    model.eval()
    with torch.no_grad():
        out["feat_eval"] = model(x)
    model.train()
    feats = model(x)
    cosine = fcm(feats)
    logits = losses.CosFace(s=30, m=0.4)(cosine.clone(), lab)
    loss = F.cross_entropy(logits, lab)
    loss.backward()
    out["feat_train"] = feats
    out["cosine"] = cosine
    out["loss"] = loss
    names, norms = [], []
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        names.append(k)
        norms.append(float(p.grad.norm()))
        if p.grad.numel() <= 512 and (k.startswith(("bn1", "prelu", "layer1.0", "layer4.2", "bn2", "fc.bias",
                                                     "features", "layer3.5")) or "downsample.1" in k):
            out["g_" + k] = p.grad
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms, dtype=np.float64)
    out["g_conv1.weight"] = model.conv1.weight.grad
    out["g_fc_head_norm"] = fcm.fc.grad.norm()
    out["g_fc_head_rows"] = fcm.fc.grad[:8]
    g = model.layer3[1].conv1.weight.grad
    out["g_layer3.1.conv1.weight_slice"] = g[:4, :16]
    g = model.fc.weight.grad
    out["g_fc.weight_slice"] = g[:4, :2048]
    for k in ("bn1", "layer1.0.bn1", "layer2.0.downsample.1", "layer4.2.bn3", "bn2", "features"):
        mod = dict(model.named_modules())[k]
        out["rm_" + k] = mod.running_mean
        out["rv_" + k] = mod.running_var
        out["nbt_" + k] = mod.num_batches_tracked
    save(fname, **out)


# ---- 3. heads ---------------------------------------------------------------------------------------
def gen_heads():
    B, C = 16, 40
    x = R.closed_form((B, 512), 0.113, 0.2, 1.0)
    w = R.closed_form((C, 512), 0.071, 1.1, 0.01)
    lab = R.closed_form_labels(B, C)
    lab_m1 = lab.clone()
    lab_m1[[1, 5, 11]] = -1      # PartialFC convention rows (losses.py:24,39)
    out = {"B": B, "C": C, "labels": lab, "labels_m1": lab_m1}
    fcm = client.FC_module(512, C, "/tmp")
    fcm.fc.data = w.clone()
    for nm, cls, s, m in (("cos", losses.CosFace, 30.0, 0.4), ("arc", losses.ArcFace, 30.0, 0.4),
                          ("cos64", losses.CosFace, 64.0, 0.4), ("arc64", losses.ArcFace, 64.0, 0.5)):
        xx = x.clone().requires_grad_(True)
        fcm.fc.grad = None
        cosine = fcm(xx)
        logits = cls(s=s, m=m)(cosine.clone(), lab)
        loss = F.cross_entropy(logits, lab)
        loss.backward()
        out[nm + "_cosine"] = cosine
        out[nm + "_logits"] = logits
        out[nm + "_loss"] = loss
        out[nm + "_dx"] = xx.grad
        out[nm + "_dw"] = fcm.fc.grad.clone()
        with torch.no_grad():
            out[nm + "_logits_m1"] = cls(s=s, m=m)(fcm(x).clone(), lab_m1)
    with torch.no_grad():
        out["nonorm_cosine"] = fcm(x, normalize_feat=False)
    save("heads", **out)


def gen_bce():
    B, C = 12, 10
    x = R.closed_form((B, 512), 0.113, 0.2, 1.0).requires_grad_(True)
    mod = client.BCE_module(512, C, 1)
    mod.weight.data = R.closed_form((C, 512), 0.071, 1.1, 0.05)
    mod.bias.data = R.closed_form((C,), 0.5, 0.1, 0.1)
    cw = torch.eye(512) + R.closed_form((512, 512), 0.013, 0.7, 0.01)
    mod.converter[0].weight.data = cw
    mod.converter[0].bias.data = R.closed_form((512,), 0.3, 0.2, 0.01)
    lab = torch.tensor([0, 3, 9, 12, 5, 5, 25, 1, 2, 7, 10, 4])   # >= C => all-negative rows
    z, gt = mod(x, lab)
    loss = losses.BCE_loss()(z.clone(), gt)
    loss.backward()
    save("bce", B=B, C=C, labels=lab, z=z, gt=gt, loss=loss, dx=x.grad,
         d_weight=mod.weight.grad, d_bias=mod.bias.grad,
         d_conv_w_slice=mod.converter[0].weight.grad[:8, :64], d_conv_b=mod.converter[0].bias.grad,
         conv_w_norm=mod.converter[0].weight.grad.norm())


# ---- 4. SGD ------------------------------------------------------------------------------------------
def gen_sgd():
    ps = [torch.nn.Parameter(R.closed_form(s, 0.2 + 0.1 * i, 0.3 * i, 0.5)) for i, s in
          enumerate([(7, 5), (33,), (4, 3, 3, 3)])]
    opt = torch.optim.SGD(ps, lr=0.1, momentum=0.9, weight_decay=5e-4)
    out = {}
    for step in range(3):
        for i, p in enumerate(ps):
            p.grad = R.closed_form(tuple(p.shape), 0.15 + 0.05 * i + 0.01 * step, 0.7 * step, 0.3)
        opt.step()
        for i, p in enumerate(ps):
            out["p%d_s%d" % (i, step)] = p.detach().clone()
            out["m%d_s%d" % (i, step)] = opt.state[p]["momentum_buffer"].clone()
    save("sgd", **out)


# ---- 5. FedAvg ------------------------------------------------------------------------------------------
def gen_fedavg():
    layers = R.IRESNET_LAYERS["iresnet18"]
    sizes = [1200, 800, 3100]
    models = []
    for i in range(3):
        sd = R.closed_form_state_dict(layers, tag=float(i + 1))
        small = {k: v for k, v in sd.items() if not k.startswith("fc.weight")}   # keep fixture small
        models.append(small)
    agg = server.FedPavg(models, sizes)
    out = {"sizes": np.array(sizes)}
    for k in ("conv1.weight", "bn1.running_var", "bn1.num_batches_tracked", "layer2.0.downsample.0.weight",
              "layer4.1.prelu.weight", "fc.bias", "features.running_mean", "layer3.1.bn2.num_batches_tracked"):
        out["agg_" + k] = agg[k]
    out["agg_dtype_nbt"] = np.array(str(agg["bn1.num_batches_tracked"].dtype))
    tot = 0.0
    for k, v in agg.items():
        tot += float(v.double().sum())
    out["agg_checksum"] = tot
    fcs = [R.closed_form((60, 512), 0.1 + 0.01 * i, 0.2 * i, 0.02) for i in range(3)]
    pre = R.closed_form((60, 512), 0.31, 0.5, 0.02)
    out["fc_p1"] = server.FedAvg_on_FC(pre, fcs, sizes, 1)
    out["fc_p05"] = server.FedAvg_on_FC(pre, fcs, sizes, 0.5)
    save("fedavg", **out)


# ---- 6. PartialFC ----------------------------------------------------------------------------------------
class _FakeStream:
    def __init__(self, *a, **k):
        pass

    def wait_stream(self, s):
        pass


@contextlib.contextmanager
def _cpu_cuda_shims():
    import torch._dynamo  # noqa: F401  (App. C: import before patching torch.device)
    real_device = torch.device
    saved = (torch.cuda.Stream, torch.cuda.current_stream, torch.cuda.stream, torch.Tensor.cuda)

    class _DevMeta(type):
        def __instancecheck__(cls, inst):
            return isinstance(inst, real_device)

    class _Dev(metaclass=_DevMeta):
        def __new__(cls, *a, **k):
            if a and isinstance(a[0], str) and a[0].startswith("cuda"):
                return real_device("cpu")
            return real_device(*a, **k)

    torch.device = _Dev
    torch.cuda.Stream = _FakeStream
    torch.cuda.current_stream = lambda *a, **k: _FakeStream()
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.device = real_device
        torch.cuda.Stream, torch.cuda.current_stream, torch.cuda.stream, torch.Tensor.cuda = saved


def _pfc_run(rank, world, B, C, rate, margin_name, s, m, steps, seed_base, q=None, port=29611, grad_rows=48, w_rows=64):
    import torch.distributed as dist
    import partial_fc
    torch.set_num_threads(max(1, 8 // world))
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)

        def _rs(out, lst, *a, **k):   # gloo lacks reduce_scatter
            t = torch.cat([x.detach() for x in lst], dim=0)
            dist.all_reduce(t)
            out.data.copy_(t.chunk(world, dim=0)[rank])
        dist.reduce_scatter = _rs
    else:
        # world_size 1: make the six dist calls identities
        dist.all_gather = lambda outs, t, *a, **k: [o.copy_(t) for o in outs][:0]
        dist.all_reduce = lambda t, *a, **k: None

        def _rs1(out, lst, *a, **k):
            out.data.copy_(lst[0].detach())
        dist.reduce_scatter = _rs1
    margin = getattr(losses, margin_name)(s=s, m=m)
    with _cpu_cuda_shims():
        pfc = partial_fc.PartialFC(rank=rank, local_rank=0, world_size=world, batch_size=B, resume=False,
                                   margin_softmax=margin, num_classes=C, sample_rate=rate,
                                   embedding_size=512, prefix="/tmp")
        num_local, class_start = R.pfc_shard(C, world, rank)
        pfc.weight = R.closed_form((num_local, 512), 0.071 + 0.003 * rank, 1.1, 0.01)
        pfc.weight_mom = torch.zeros_like(pfc.weight)
        if int(rate) == 1:
            pfc.sub_weight = torch.nn.Parameter(pfc.weight)
            pfc.sub_weight_mom = pfc.weight_mom
        dummy = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([{"params": [dummy]}, {"params": [pfc.sub_weight]}], lr=0.1, momentum=0.9,
                              weight_decay=5e-4)
        res = {}
        real_rand = torch.rand
        for st in range(steps):
            # upstream callers hand PartialFC L2-normalised embeddings (ArcFace's acos is unclamped)
            feats = F.normalize(R.closed_form((B, 512), 0.113 + 0.01 * rank + 0.001 * st, 0.2 + st, 1.0))
            lab = (R.closed_form_labels(B, C, tag=st + 3 * rank) * 31 + rank) % C
            perm = R.closed_form((num_local,), 0.77 + 0.1 * st, 0.3 + rank, 0.5, 0.5)
            opt.zero_grad()                                # caller protocol: fresh grads every step
            torch.rand = lambda *a, **k: perm.clone()      # inject the draw (partial_fc.py:97)
            x_grad, loss_v = pfc.forward_backward(lab, feats, opt)
            torch.rand = real_rand
            sw_grad = pfc.sub_weight.grad.clone()
            opt.step()
            pfc.update()
            pre = "r%d_s%d_" % (rank, st)
            res[pre + "x_grad"] = x_grad.detach().clone()
            res[pre + "loss_v"] = loss_v.detach().clone()
            res[pre + "sub_weight_grad_rows"] = sw_grad[:: max(1, sw_grad.shape[0] // grad_rows)][:grad_rows].clone()
            res[pre + "sub_weight_grad_rownorm"] = sw_grad.norm(dim=1)
            if pfc.index is not None:
                res[pre + "index"] = pfc.index.clone()
            res[pre + "weight_rows"] = pfc.weight[:: max(1, num_local // w_rows)][:w_rows].clone()
            res[pre + "mom_rows"] = pfc.weight_mom[:: max(1, num_local // w_rows)][:w_rows].clone()
            res[pre + "weight_sum"] = pfc.weight.double().sum()
            res[pre + "mom_sum"] = pfc.weight_mom.double().sum()
    if world > 1:
        q.put({k: v.numpy() for k, v in res.items()})
        dist.barrier()
        dist.destroy_process_group()
    return res


def gen_pfc():
    import torch.multiprocessing as mp
    # subprocess per config so the dist monkey-patches do not leak
    cfgs = [("pfc_w1_arc_r01", 1, 8, 2000, 0.1, "ArcFace", 30.0, 0.4, 2),
            ("pfc_w1_cos_r1", 1, 8, 300, 1.0, "CosFace", 30.0, 0.4, 2),
            ("pfc_w1_cos_r03", 1, 8, 1000, 0.3, "CosFace", 64.0, 0.4, 2)]
    ctx = mp.get_context("spawn")
    for name, W, B, C, rate, mn, s, m, steps in cfgs:
        q = ctx.Queue()
        p = ctx.Process(target=_pfc_single_entry, args=(q, B, C, rate, mn, s, m, steps))
        p.start()
        res = q.get()
        p.join()
        save(name, B=B, C=C, rate=rate, s=s, m=m, steps=steps, margin=np.array(mn), **res)
    # W = 2 over gloo, uneven shards (1001 classes → 501/500)
    q = ctx.Queue()
    B, C, rate, mn, s, m, steps = 8, 1001, 0.2, "CosFace", 30.0, 0.4, 2
    procs = [ctx.Process(target=_pfc_run, args=(r, 2, B, C, rate, mn, s, m, steps, 0, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        res.update(q.get())
    for p in procs:
        p.join()
    save("pfc_w2", B=B, C=C, rate=rate, s=s, m=m, steps=steps, margin=np.array(mn), **res)
    # W = 4 (2003 classes -> shards 501/501/501/500) and W = 8 at BASELINE config 5's shard geometry (85 003 classes -> 10 626 x 3 +
    # 10 625 x 5 per rank, sample_rate 0.1 -> 1 062 sampled rows, CosFace): fewer sampled rows are kept per rank to bound the fixture
    for name, W, B, C, rate, mn, s, m, steps, port, gr, wr in (("pfc_w4", 4, 4, 2003, 0.2, "ArcFace", 30.0, 0.4, 2, 29613, 16, 16),
                                                                ("pfc_w8", 8, 4, 85003, 0.1, "CosFace", 30.0, 0.4, 2, 29615, 6, 6)):
        q = ctx.Queue()
        procs = [ctx.Process(target=_pfc_run, args=(r, W, B, C, rate, mn, s, m, steps, 0, q, port, gr, wr)) for r in range(W)]
        for p in procs:
            p.start()
        res = {}
        for _ in procs:
            res.update(q.get())
        for p in procs:
            p.join()
        save(name, B=B, C=C, rate=rate, s=s, m=m, steps=steps, margin=np.array(mn), grad_rows=gr, w_rows=wr, **res)


def _pfc_single_entry(q, B, C, rate, mn, s, m, steps):
    res = _pfc_run(0, 1, B, C, rate, mn, s, m, steps, 0)
    q.put({k: v.numpy() for k, v in res.items()})


# ---- 7. client loop on iresnet18 ------------------------------------------------------------------------------
def gen_client():
    layers = R.IRESNET_LAYERS["iresnet18"]
    C, B, steps, lr = 20, 8, 3, 0.01
    model = backbones.iresnet18(False, dropout=0, fp16=False)
    load_closed_form(model, layers, tag=2.0)
    fcm = client.FC_module(512, C, "/tmp")
    fcm.fc.data = R.head_fc(C)
    model.train()
    fcm.train()
    seq = torch.nn.Sequential(model, fcm)
    opt = torch.optim.SGD(params=seq.parameters(), lr=lr, momentum=0.9, weight_decay=5e-4)
    margin = losses.CosFace(s=30, m=0.4)
    out = {"B": B, "C": C, "steps": steps, "lr": lr}
    ls = []
    for st in range(steps):
        imgs = R.closed_form_images(B, tag=float(st))
        lab = R.closed_form_labels(B, C, tag=st)
        opt.zero_grad()
        logits = seq(imgs)
        logits = margin(logits, lab)
        loss = F.cross_entropy(logits, lab)
        loss.backward()
        opt.step()
        ls.append(float(loss))
    out["losses"] = np.array(ls, dtype=np.float64)
    sd = model.state_dict()
    for k in ("conv1.weight", "bn1.weight", "bn1.bias", "bn1.running_mean", "bn1.running_var",
              "bn1.num_batches_tracked", "prelu.weight", "layer2.0.downsample.0.weight",
              "layer4.1.bn3.running_var", "fc.bias", "features.bias", "features.running_mean"):
        out["sd_" + k] = sd[k]
    out["sd_fc.weight_slice"] = sd["fc.weight"][:4, :2048]
    out["sd_layer3.1.conv2.weight_slice"] = sd["layer3.1.conv2.weight"][:4, :32]
    out["head_fc"] = fcm.fc.data
    tot = 0.0
    for k, v in sd.items():
        tot += float(v.double().abs().sum())
    out["sd_abs_checksum"] = tot
    save("client_r18", **out)


# ---- 8. train_with_public_data loop body (client.py:354-441) on iresnet18 ---------------------------------------
PUB_KEYS = ("conv1.weight", "bn1.weight", "bn1.running_mean", "bn1.num_batches_tracked", "prelu.weight",
            "layer2.0.downsample.0.weight", "layer4.1.bn3.running_var", "fc.bias", "features.running_mean")


def public_setup():
    """shared by the generator and the tests' oracle run: sizes + deterministic initial state (no RNG)."""
    return dict(n_local=6, n_public=14, B=8, steps=3, lr=0.01, mu=5.0, temperature=0.5, tag=2.0, last_tag=5.0)


def gen_public(variant):
    """variant 'full': Branch_model + BCE (weight 10) + model-contrastive (weight mu) — the FedFR objective;
    variant 'seq': Sequential_model, CosFace over [local | public] only;
    variant 'bce_rw': Branch_model + BCE + reweight_cosface (num_client 4), no contrastive.  NB the reference concatenates inside
    torch.no_grad() (client.py:272-276), so with reweighting the CosFace term reaches the reported loss but carries NO gradient —
    reproduced as is (quirks are kept)."""
    cfgp = public_setup()
    layers = R.IRESNET_LAYERS["iresnet18"]
    nl, npub, B, steps, lr = cfgp["n_local"], cfgp["n_public"], cfgp["B"], cfgp["steps"], cfgp["lr"]
    C = nl + npub
    backbone = backbones.iresnet18(False, dropout=0, fp16=False)
    load_closed_form(backbone, layers, tag=cfgp["tag"])
    backbone.train()
    fcm = client.FC_module(512, nl, "/tmp")
    fcm.fc.data = R.head_fc(nl, seed=11)
    fcm.update_with_pretrain(R.head_fc(npub, seed=12))              # client.py:312
    fcm.train()
    margin = losses.CosFace(s=30, m=0.4)
    out = dict(cfgp)
    out["variant"] = np.array(variant)
    has_bce, has_con, has_rw = variant in ("full", "bce_rw"), variant == "full", variant == "bce_rw"
    num_client, num_classes = 4, nl
    if has_bce:
        bcem = client.BCE_module(512, nl, 1)
        bcem.weight.data = R.head_fc(nl, seed=13)
        bcem.train()
        bce_crit = losses.BCE_loss()
        model = client.Branch_model(backbone, fcm, bcem)
    else:
        model = client.Sequential_model(backbone, fcm)
    if has_con:
        import copy
        global_model = copy.deepcopy(backbone).eval()
        last_model = backbones.iresnet18(False, dropout=0, fp16=False)
        load_closed_form(last_model, layers, tag=cfgp["last_tag"])
        last_model.eval()
        con_criterion = torch.nn.CosineSimilarity(dim=1)

    def reweight(logits, labels):                                     # client.py:269-285, verbatim semantics
        with torch.no_grad():
            idx_bool = torch.ones(logits.shape).bool()
            idx_bool[torch.arange(len(labels)), labels] = False
            tmp = logits.detach().clone()[idx_bool].reshape(len(labels), logits.shape[1] - 1)[:, :num_classes].repeat(1, num_client - 1)
            logits = torch.cat([logits, tmp], dim=1)
        return logits
    opt = torch.optim.SGD(params=model.parameters(), lr=lr, momentum=0.9, weight_decay=5e-4)
    rows = []
    for st in range(steps):
        imgs = R.closed_form_images(B, tag=float(st))
        labels = R.closed_form_labels(B, C, tag=st)
        opt.zero_grad()
        if variant == "full":
            with torch.no_grad():
                global_feats = global_model(imgs)
                last_feats = last_model(imgs)
            cos_logits, bce_logits, bce_gts, feats = model(imgs, labels, contrastive=True, detach=False)
            pos_sim = con_criterion(feats, global_feats) / cfgp["temperature"]
            neg_sim = con_criterion(feats, last_feats) / cfgp["temperature"]
            con_label = torch.zeros(len(labels)).long()
            con_loss = F.cross_entropy(torch.stack([pos_sim, neg_sim], dim=1), con_label)
            cos_logits = margin(cos_logits, labels)
            cos_loss = F.cross_entropy(cos_logits, labels)
            bce_loss = bce_crit(bce_logits, bce_gts)
            loss = cos_loss + 10 * bce_loss + cfgp["mu"] * con_loss
            rows.append([float(loss), float(cos_loss), float(con_loss), float(bce_loss)])
        elif variant == "bce_rw":
            cos_logits, bce_logits, bce_gts = model(imgs, labels, contrastive=False, detach=False)
            cos_logits = reweight(margin(cos_logits, labels), labels)
            cos_loss = F.cross_entropy(cos_logits, labels)
            bce_loss = bce_crit(bce_logits, bce_gts)
            loss = cos_loss + 10 * bce_loss
            rows.append([float(loss), float(cos_loss), float("nan"), float(bce_loss)])
        else:
            logits = model(imgs)
            logits = margin(logits, labels)
            loss = F.cross_entropy(logits, labels)
            rows.append([float(loss), float(loss), float("nan"), float("nan")])
        loss.backward()
        opt.step()
    out["rows"] = np.array(rows, dtype=np.float64)
    sd = backbone.state_dict()
    for k in PUB_KEYS:
        out["sd_" + k] = sd[k]
    out["sd_layer3.1.conv2.weight_slice"] = sd["layer3.1.conv2.weight"][:4, :32]
    out["head_fc"] = fcm.fc.data
    out["sd_abs_checksum"] = sum(float(v.double().abs().sum()) for v in sd.values())
    if has_bce:
        out["bce_weight"] = bcem.weight.data
        out["bce_bias"] = bcem.bias.data
        out["bce_conv_w_diag"] = torch.diagonal(bcem.converter[0].weight.data).clone()
        out["bce_conv_w_slice"] = bcem.converter[0].weight.data[:8, :64]
        out["bce_conv_b"] = bcem.converter[0].bias.data
    save("client_public_" + variant, **out)


# ---- 9. inference sweeps + hard-negative mining (client.py:159-236, server.py:182-263) on iresnet18 ------------------------------
def gen_mining():
    cfgm = dict(tag=2.0, B=4, nb=3, n_local=5, n_public=6)
    layers = R.IRESNET_LAYERS["iresnet18"]
    backbone = backbones.iresnet18(False, dropout=0, fp16=False)
    load_closed_form(backbone, layers, tag=cfgm["tag"])
    backbone.eval()
    _, local, public = R.mining_fixture_state(cfgm)
    out = dict(cfgm)
    with torch.no_grad():
        # client.py:171-178 (data_update_fc), both norm_before_avg settings
        for nba in (True, False):
            init_fc = torch.zeros(cfgm["n_local"], 512)
            num_samples = torch.zeros(cfgm["n_local"])
            for img, label in local:
                features = backbone(img)
                if nba:
                    features = F.normalize(features)
                u_label = torch.unique(label)
                for l in u_label:
                    init_fc[l:l + 1, :] += torch.sum(features[label == l, :], dim=0)
                    num_samples[l] += torch.sum(label == l)
            init_fc /= num_samples.unsqueeze(1)
            out["local_centers_nba%d" % int(nba)] = init_fc
        # server.py:250-259 (Generate_pretrain_feats) and :204-229 (Initialize_pretrain_FC, norm_before_avg = True)
        raw_feats, raw_labels, ID_feature = [], [], dict()
        for img, label in public:
            features = F.normalize(backbone(img))
            raw_feats.append(features)
            raw_labels.append(label)
            for i, ID in enumerate(label):
                ID = ID.item()
                if ID not in ID_feature:
                    ID_feature[ID] = [torch.zeros_like(features[0]), 0]
                ID_feature[ID][0] += features[i]
                ID_feature[ID][1] += 1
        raw_feats, raw_labels = torch.cat(raw_feats, dim=0), torch.cat(raw_labels)
        init_matrix = torch.zeros(len(ID_feature), 512)
        for i in range(len(init_matrix)):
            init_matrix[i] = ID_feature[i][0] / ID_feature[i][1]
        out["public_feats"], out["public_labels"], out["public_centers"] = raw_feats, raw_labels, init_matrix
        # client.py:197-226 (choose_hard_negative_2); the threshold sits in the widest gap of the per-column maxima so that the
        # selected set is stable under the bf16 backbone's 1e-2-class embedding noise
        local_feats = torch.cat([F.normalize(backbone(img)) for img, _ in local], dim=0)
        similarity = torch.matmul(local_feats, raw_feats.t())
        colmax = torch.sort(similarity.max(dim=0).values).values
        gaps = colmax[1:] - colmax[:-1]
        k = int(torch.argmax(gaps[2:-2])) + 2
        thr = float((colmax[k] + colmax[k + 1]) / 2)
        unique_idx = []
        for i in range(len(similarity)):
            unique_idx.append(torch.where(similarity[i:i + 1] > thr)[1].numpy())
        from functools import reduce
        unique_idx = sorted(reduce(np.union1d, unique_idx))
        out["local_feats"], out["similarity"], out["hn_threshold"], out["hn_gap"] = local_feats, similarity, thr, float(gaps[k])
        out["hn_index"] = np.array(unique_idx, dtype=np.int64)
        out["hn_num_id"] = len(torch.unique(raw_labels[unique_idx]))
    save("mining_r18", **out)


# ---- 10. pairwise ROC histogram (roc_cuda.py) ---------------------------------------------------------------------------
def gen_roc():
    """Runs the reference's own `calc_ROC` kernel BODY (roc_cuda.py:14-30) and its `plot_ROC` (:61-88): numba is absent, so a stub
    makes `@cuda.jit` the identity, `cuda.grid(2)` return the (i, j) of a Python double loop and `cuda.atomic.add` a plain add."""
    import tempfile
    nb = types.ModuleType("numba")
    cu = types.ModuleType("numba.cuda")
    state = {"ij": (0, 0)}
    cu.jit = lambda f: f
    cu.grid = lambda n: state["ij"]

    class _Atomic:
        @staticmethod
        def add(arr, idx, v):
            arr[idx] += v
    cu.atomic = _Atomic
    nb.cuda = cu
    sys.modules["numba"], sys.modules["numba.cuda"] = nb, cu
    import roc_cuda                                            # noqa: E402  (reference)
    N, D, T, nid = 72, 64, 30, 9
    g = torch.Generator().manual_seed(5)
    centers = F.normalize(torch.randn(nid, D, generator=g))
    label = (torch.arange(N) * 5 + 2) % nid
    feat = F.normalize(centers[label] + 0.35 * torch.randn(N, D, generator=g)).numpy().astype(np.float32)
    label = label.numpy().astype(np.int32)
    # main program of roc_cuda.py:129-136: target identities first
    target_label = [0, 1, 2]
    t_idx = np.isin(label, target_label)
    feature = np.concatenate([feat[t_idx], feat[~t_idx]], axis=0)
    lab = np.concatenate([label[t_idx], label[~t_idx]])
    target_size = int(t_idx.sum())
    out_sum = np.zeros(2001 * 2, dtype=np.int64)
    batch_size = 8                                             # several producer batches (roc_cuda.py:32-37)
    for bi in range((target_size + batch_size - 1) // batch_size):
        start = bi * batch_size
        index = np.arange(len(feature))[start:min(start + batch_size, target_size)]
        feature_c, label_c = feature[start:, :].astype(np.float32), lab[start:].astype(np.int32)
        subfeature, sublabel = feature[index, :].astype(np.float32), lab[index].astype(np.int32)
        out = np.zeros(2001 * 2, dtype=np.float64)
        for i in range(len(index)):
            for j in range(feature_c.shape[0]):
                state["ij"] = (i, j)
                roc_cuda.calc_ROC(feature_c, label_c, subfeature, sublabel, out)
        out_sum += out.astype(np.int64)
    hist = out_sum.reshape([-1, 2])
    with tempfile.TemporaryDirectory() as td, contextlib.redirect_stdout(open(os.devnull, "w")):
        roc_cuda.plot_ROC(hist, td, 0, target_label)
        line = [l for l in open(os.path.join(td, "local_log.txt")) if "TPR" in l][0]
    tpr = eval(line.split("=")[1])
    save("roc", features=feature, labels=lab.astype(np.int64), target_size=target_size, hist=hist, tpr=np.array(tpr), total_pairs=int(hist.sum()))


# ---- 11. sphnet (backbones/sphnet.py), sphere20 and sphere64 (the reference's default, sphnet.py:72, and what run.sh trains) fwd + bwd ----
def gen_sphnet(type_=20):
    from backbones.sphnet import sphere as ref_sphere
    B = 8 if type_ == 20 else 4
    net = ref_sphere(type_)
    sd = R.sphere_state_dict(type_, tag=1.0)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd)
    net.train()
    x = R.closed_form_images(B, tag=4.0)
    dfe = R.closed_form((B, 512), 0.37, 0.9, 1.0)
    feats = net(x)
    (feats * dfe).sum().backward()
    out = {"B": B, "feats": feats.detach()}
    for k, p in net.named_parameters():
        g = p.grad
        out["gnorm_" + k] = g.norm()
        if g.numel() <= 4096:
            out["g_" + k] = g
    out["g_layer2.2.conv1.weight_slice"] = net.layer2[2].conv1.weight.grad[:4, :16]
    out["g_layer1.0.weight"] = net.layer1[0].weight.grad
    out["g_fc.weight_slice"] = net.fc.weight.grad[:4, :2048]
    if type_ == 64:         # the 16-unit 14x14 stage (layer3) and its neighbours: slices of a deep, a first and a last unit
        out["g_layer3.17.conv2.weight_slice"] = net.layer3[17].conv2.weight.grad[:4, :16]      # (layerN = [conv, prelu, unit 0, unit 1, ...])
        out["g_layer3.2.conv1.weight_slice"] = net.layer3[2].conv1.weight.grad[:4, :16]
        out["g_layer4.4.conv1.weight_slice"] = net.layer4[4].conv1.weight.grad[:2, :16]
    out["keys"] = np.array(list(sd.keys()))
    save("sphnet%d" % type_, **out)


def gen_freeze_bn():
    """The reference's IResNet.freeze_BN(test_mode=True) (iresnet.py:140-147): every BatchNorm in eval mode inside a training net.
    iresnet18, closed-form weights with non-trivial running statistics, one forward + backward of sum(feats * w)."""
    layers = R.IRESNET_LAYERS["iresnet18"]
    model = backbones.iresnet18(False, dropout=0, fp16=False)
    load_closed_form(model, layers, tag=7.0)
    B = 4
    x = R.closed_form_images(B, tag=3.0)
    w = R.closed_form((B, 512), 0.37, 0.9, 1.0)
    model.train()
    model.freeze_BN()
    feats = model(x)
    (feats * w).sum().backward()
    out = {"B": B, "feats": feats}
    names, norms = [], []
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        names.append(k)
        norms.append(float(p.grad.norm()))
        if p.grad.numel() <= 512:
            out["g_" + k] = p.grad
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms, dtype=np.float64)
    out["g_conv1.weight"] = model.conv1.weight.grad
    out["g_layer3.1.conv1.weight_slice"] = model.layer3[1].conv1.weight.grad[:4, :16]
    out["g_fc.weight_slice"] = model.fc.weight.grad[:4, :2048]
    sd_after = {k: v.clone() for k, v in model.state_dict().items()}        # (state_dict() returns the live tensors)
    for k in ("bn1", "layer2.0.downsample.1", "layer4.1.bn3", "bn2", "features"):
        out["rm_" + k] = sd_after[k + ".running_mean"]
        out["rv_" + k] = sd_after[k + ".running_var"]
        out["nbt_" + k] = sd_after[k + ".num_batches_tracked"]
    # model.train() undoes it (nn.Module.train resets the BatchNorm submodules): the second forward is an ordinary training forward
    model.train()
    out["feats_after_train_call"] = model(x)
    save("freeze_bn_r18", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["block", "r50", "r100", "heads", "bce", "sgd", "fedavg", "pfc", "client", "public", "mining", "roc", "sphnet", "sphnet64", "freeze_bn"]
    if "freeze_bn" in which:
        gen_freeze_bn()
    if "block" in which:
        gen_block()
    if "r50" in which:
        gen_backbone("iresnet50", R.IRESNET_LAYERS["iresnet50"], 8, "r50_b8")
    if "r100" in which:
        gen_backbone("iresnet100", R.IRESNET_LAYERS["iresnet100"], 6, "r100_b6")
    if "heads" in which:
        gen_heads()
    if "bce" in which:
        gen_bce()
    if "sgd" in which:
        gen_sgd()
    if "fedavg" in which:
        gen_fedavg()
    if "pfc" in which:
        gen_pfc()
    if "client" in which:
        gen_client()
    if "sphnet" in which:
        gen_sphnet(20)
    if "sphnet64" in which:
        gen_sphnet(64)
    if "roc" in which:
        gen_roc()
    if "mining" in which:
        gen_mining()
    if "public" in which:
        for v in ("full", "seq", "bce_rw"):
            gen_public(v)
