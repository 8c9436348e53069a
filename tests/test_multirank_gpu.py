"""Multi-rank paths on ONE GPU: W simulated ranks as threads (fedfr_amd.comm.ThreadComm) or as processes sharing cuda:0 over gloo.

* class-sharded PartialFC at world_size 4 and 8 (BASELINE config 5's shard geometry) against the reference captured over gloo;
* the stock-optimizer contract of PartialFC (partial_fc.py:124-126);
* the FedAvg exchange (ONE all-reduce of the flat state, real HIP scale kernels) against FedPavg;
* BASELINE config 5 (per-client backbones + global sharded PartialFC + private BCE heads + FedAvg) against the oracle's composition;
* BASELINE config 3 at full size as a property test; client memory stays flat across many clients.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from conftest import load_golden  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402

from fedfr_amd import backbones, losses, client, server, ops, _C  # noqa: E402
from fedfr_amd.comm import ThreadComm, TorchDistComm  # noqa: E402
from fedfr_amd.partial_fc import PartialFC  # noqa: E402

DEV = torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.asarray(a))


def rel(a, b):
    a, b = a.detach().double().cpu(), (T(b) if not isinstance(b, torch.Tensor) else b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def maxrel(a, b):
    a, b = a.detach().double().cpu(), (T(b) if not isinstance(b, torch.Tensor) else b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _pfc_inputs(B, C, rank, st, num_local):
    feats = F.normalize(R.closed_form((B, 512), 0.113 + 0.01 * rank + 0.001 * st, 0.2 + st, 1.0))
    lab = (R.closed_form_labels(B, C, tag=st + 3 * rank) * 31 + rank) % C
    perm = R.closed_form((num_local,), 0.77 + 0.1 * st, 0.3 + rank, 0.5, 0.5)
    return feats, lab, perm


def _pfc_rank(g, comm, world, optimizer_factory=None):
    """one rank of the golden protocol of tools/make_golden.py:_pfc_run through the product PartialFC."""
    rank = comm.rank
    B, C, rate = int(g["B"]), int(g["C"]), float(g["rate"])
    s, m, steps, mn = float(g["s"]), float(g["m"]), int(g["steps"]), str(g["margin"])
    gr = int(g["grad_rows"]) if "grad_rows" in g.files else 48
    wr = int(g["w_rows"]) if "w_rows" in g.files else 64
    pfc = PartialFC(rank=rank, local_rank=0, world_size=world, batch_size=B, resume=False, margin_softmax=getattr(losses, mn)(s=s, m=m),
                    num_classes=C, sample_rate=rate, embedding_size=512, prefix="/tmp", comm=comm)
    num_local, class_start = R.pfc_shard(C, world, rank)
    assert (pfc.num_local, pfc.class_start) == (num_local, class_start)
    pfc.weight.copy_(R.closed_form((num_local, 512), 0.071 + 0.003 * rank, 1.1, 0.01).to(DEV))
    pfc.weight_mom.zero_()
    opt = optimizer_factory(pfc) if optimizer_factory else None
    for st in range(steps):
        feats, lab, perm = _pfc_inputs(B, C, rank, st, num_local)
        if opt is not None:
            opt.zero_grad()
        x_grad, loss_v = pfc.forward_backward(lab.to(DEV), feats.to(DEV), opt, perm=perm.to(DEV))
        pre = "r%d_s%d_" % (rank, st)
        if (pre + "index") in g.files:
            assert torch.equal(pfc.index.cpu(), T(g[pre + "index"])), "sampled class set differs"
        assert maxrel(x_grad, g[pre + "x_grad"]) < 1e-4
        assert abs(float(loss_v) - float(g[pre + "loss_v"])) < 1e-4 * max(1.0, abs(float(g[pre + "loss_v"])))
        swg = pfc.sub_weight.grad
        assert maxrel(swg[:: max(1, swg.shape[0] // gr)][:gr], g[pre + "sub_weight_grad_rows"]) < 1e-4
        assert maxrel(swg.norm(dim=1), g[pre + "sub_weight_grad_rownorm"]) < 1e-4
        if opt is not None:
            opt.step()                     # stock torch.optim.SGD on the aliased sampled rows + momentum rows ...
            pfc.update()                   # ... then scatter back (the caller protocol of partial_fc.py:113-116, :124-126)
        else:
            pfc.fused_sgd_update(0.1, 0.9, 5e-4)
        assert maxrel(pfc.weight[:: max(1, num_local // wr)][:wr], g[pre + "weight_rows"]) < 1e-5
        assert maxrel(pfc.weight_mom[:: max(1, num_local // wr)][:wr], g[pre + "mom_rows"]) < 1e-4
        assert abs(float(pfc.weight.double().sum()) - float(g[pre + "weight_sum"])) < 1e-3
    return "ok"


@pytest.mark.parametrize("name,world", [("pfc_w4", 4), ("pfc_w8", 8)])
def test_partial_fc_w4_w8_vs_reference(name, world):
    """world_size 4 (uneven shards 501/501/501/500, ArcFace) and world_size 8 at BASELINE config 5's shard geometry (85 003 classes:
    10 626 / 10 625 per rank, sample_rate 0.1 -> 1 062 sampled rows, CosFace) — the product's HIP path on every rank, the four packed
    collectives through the in-process communicator, against the reference captured with 4 / 8 gloo processes."""
    g = load_golden(name)
    assert ThreadComm.run(world, lambda c: _pfc_rank(g, c, world), device=DEV) == ["ok"] * world


@pytest.mark.parametrize("name", ["pfc_w1_arc_r01", "pfc_w1_cos_r1", "pfc_w1_cos_r03"])
def test_partial_fc_stock_optimizer_contract(name):
    """partial_fc.py:124-126: ``prepare`` aliases the sampled rows and their momentum rows into the LAST param group of a stock
    torch.optim.SGD; ``opt.step(); pfc.update()`` then equals the reference's update (same goldens as the fused route)."""
    g = load_golden(name)

    def make_opt(pfc):
        dummy = torch.nn.Parameter(torch.zeros(1, device=DEV))
        return torch.optim.SGD([{"params": [dummy]}, {"params": [pfc.sub_weight]}], lr=0.1, momentum=0.9, weight_decay=5e-4)
    from fedfr_amd.comm import SingleComm
    assert _pfc_rank(g, SingleComm(), 1, make_opt) == "ok"


def _closed_form_backbone(tag):
    m = backbones.iresnet18(False, dropout=0, fp16=True)
    m.load_state_dict(R.closed_form_state_dict(R.IRESNET_LAYERS["iresnet18"], tag=tag))
    return m.to(DEV)


def test_fedavg_all_reduce_real_kernels_four_ranks():
    """The round's exchange with the REAL HIP scale kernels under a 4-rank group: every rank ends with the same model, and it is
    FedPavg of the four local models — bit for bit here, because the in-process communicator adds in ascending rank order like
    server.py:27-33 (RCCL's ring order would differ in the last bit)."""
    sizes = [300.0, 100.0, 250.0, 50.0]
    models = [_closed_form_backbone(float(r + 1)) for r in range(4)]
    for r, m in enumerate(models):
        m._flat_nbt += 3 * r                                        # different counters per client (F9 path)
    expect = server.FedPavg([client.flat_state_dict(m) for m in models], sizes)

    def run(c):
        calls = []
        real = c.all_reduce
        c.all_reduce = lambda t, op="sum": (calls.append(t.numel()), real(t, op))[1]
        w = server.fedavg_all_reduce(models[c.rank], sizes[c.rank], sum(sizes), c)
        assert calls == [models[c.rank]._flat_state.numel()]        # ONE collective, the whole state in place
        return w
    ws = ThreadComm.run(4, run, device=DEV)
    assert ws == [s / sum(sizes) for s in sizes]
    torch.cuda.synchronize()
    for m in models:
        out = m.state_dict()
        for k, v in expect.items():
            if v.is_floating_point() and out[k].is_floating_point():
                assert torch.equal(out[k], v), k
            else:
                assert int(out[k]) == int(float(v)), k                # float average truncated to int64 (F9)


def _fedavg_gloo_worker(rank, port, tmp):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        m = _closed_form_backbone(float(rank + 1))
        sizes = [300.0, 100.0]
        comm = TorchDistComm()
        total = server.exchange_data_sizes(sizes[rank], comm)
        server.fedavg_all_reduce(m, sizes[rank], total, comm)
        torch.cuda.synchronize()
        torch.save({k: v.cpu() for k, v in m.state_dict().items()}, os.path.join(tmp, "r%d.pt" % rank))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_fedavg_all_reduce_two_processes_shared_gpu(tmp_path):
    """Two PROCESSES sharing cuda:0, torch.distributed over gloo (RCCL refuses two ranks on one device), real HIP kernels."""
    import torch.multiprocessing as mp
    mp.spawn(_fedavg_gloo_worker, args=(29671, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    expect = server.FedPavg([client.flat_state_dict(_closed_form_backbone(1.0)), client.flat_state_dict(_closed_form_backbone(2.0))], [300, 100])
    for k in a:
        assert torch.equal(a[k], b[k]), k
        if a[k].is_floating_point():
            assert torch.equal(a[k], expect[k].cpu()), k              # two addends: a + b == b + a in fp32
        else:
            assert int(a[k]) == int(float(expect[k]))


# ---- BASELINE config 5 ------------------------------------------------------------------------------------------------------------
def _config5_inputs(rank, W, B, n_ids, steps):
    """rank's data shard: identities [rank * n_ids, (rank + 1) * n_ids)."""
    return [(R.closed_form_images(B, tag=float(rank * 3 + s)), R.closed_form_labels(B, n_ids, tag=s + rank) + rank * n_ids) for s in range(steps)]


def test_config5_hybrid_vs_oracle():
    """4 clients (threads) on one GPU: per-client iresnet18 + ONE PartialFC sharded over the 4 ranks (uneven shards, sample_rate 0.5,
    injected draws) + a private BCE head each, 2 local steps, then the FedAvg exchange.  Against the oracle's composition of the
    restated reference pieces (oracle/ref_cpu.py:config5_client_steps) run on the CPU with the same communicator pattern."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle_golden import _OracleThreadComm
    W, B, n_ids, steps, lr, rate = 4, 4, 10, 2, 0.01, 0.5
    C = W * n_ids + 3                                           # 43 classes -> shards 11/11/11/10: PartialFC shards != client id ranges
    layers = R.IRESNET_LAYERS["iresnet18"]
    sizes = [40.0, 30.0, 20.0, 10.0]

    def state(rank):
        nl, _ = R.pfc_shard(C, W, rank)
        return (R.closed_form_state_dict(layers, tag=float(rank + 1)), R.closed_form((nl, 512), 0.071 + 0.003 * rank, 1.1, 0.01),
                {"conv_w": torch.eye(512), "conv_b": torch.zeros(512), "weight": R.head_fc(n_ids, seed=20 + rank), "bias": torch.zeros(n_ids)},
                [R.closed_form((nl,), 0.77 + 0.1 * st, 0.3 + rank, 0.5, 0.5) for st in range(steps)])

    # ---- oracle on the CPU
    def oracle_rank(c):
        torch.set_num_threads(2)
        sd, w, bce, perms = state(c.rank)
        mom = torch.zeros_like(w)
        out = R.config5_client_steps(sd, w, mom, bce, _config5_inputs(c.rank, W, B, n_ids, steps), layers, _OracleThreadComm(c), B, C, rate,
                                     c.rank * n_ids, lr, perms)
        return out, sd, w, bce
    ref = ThreadComm.run(W, oracle_rank)
    ref_avg = R.fedpavg([r[1] for r in ref], sizes)

    # ---- product path, 4 ranks on cuda:0
    def hip_rank(c):
        rank = c.rank
        sd, w, bce, perms = state(rank)
        bb = backbones.iresnet18(False, dropout=0, fp16=True)
        bb.load_state_dict(sd)
        bb = bb.to(DEV)
        pfc = PartialFC(rank=rank, local_rank=0, world_size=W, batch_size=B, resume=False, margin_softmax=losses.CosFace(s=30, m=0.4),
                        num_classes=C, sample_rate=rate, embedding_size=512, prefix="/tmp", comm=c)
        pfc.weight.copy_(w.to(DEV))
        pfc.weight_mom.zero_()
        bm = client.BCE_module(512, n_ids, 1).to(DEV)
        bm.weight.data = bce["weight"].to(DEV)
        tr = client.ShardedHeadTrainer(bb, pfc, bm, id_base=rank * n_ids, lr=lr)
        ls = []
        for st, (imgs, lab) in enumerate(_config5_inputs(rank, W, B, n_ids, steps)):
            loss, cos, b = tr.step(imgs.to(DEV), lab.to(DEV), perm=perms[st].to(DEV))
            ls.append((float(loss), float(cos), float(b)))
        local = {k: v.clone() for k, v in bb.state_dict().items()}
        tr.end_round(sizes[rank], sum(sizes))
        torch.cuda.synchronize()
        return ls, local, bb.state_dict(), pfc.weight.clone(), bm
    out = ThreadComm.run(W, hip_rank, device=DEV)
    for rank in range(W):
        ls, local, avg, pw, bm = out[rank]
        rls, rsd, rw, rbce = ref[rank]
        for (l, c_, b), (rl, rc, rb) in zip(ls, rls):
            assert abs(c_ - rc) < 1e-2 * abs(rc), (rank, c_, rc)              # bf16 backbone: 1e-2-class embeddings -> loss
            assert abs(b - rb) < 1e-2 * abs(rb), (rank, b, rb)
            assert abs(l - rl) < 1e-2 * abs(rl)
        assert ls[0][1] == out[0][0][0][1]                                       # the sharded-head loss is the same number on every rank
        for k in ("conv1.weight", "layer2.0.downsample.0.weight", "bn1.weight", "prelu.weight", "fc.bias"):
            assert rel(local[k], rsd[k]) < 1e-2, (rank, k, rel(local[k], rsd[k]))
        for k in ("bn1.running_mean", "layer4.1.bn3.running_var"):
            assert rel(local[k], rsd[k]) < 3e-2, (rank, k)
        # this rank's PartialFC shard / BCE centres after 2 updates: the rows are O(0.01) and each update (lr x gradient of a
        # 1e-2-class bf16 embedding) is as large as the row itself, so the rows carry the embeddings' error class
        assert rel(pw, rw) < 3e-2, (rank, rel(pw, rw))
        assert rel(bm.weight.data, rbce["weight"]) < 3e-2 and rel(bm.converter[0].weight.data, rbce["conv_w"]) < 1e-3
        # round end: every rank holds the same averaged backbone == FedPavg of the four local models
        for k in ("conv1.weight", "layer3.1.conv2.weight", "bn2.running_var", "fc.bias"):
            assert torch.equal(avg[k], out[0][2][k]), k
            assert rel(avg[k], ref_avg[k]) < 1e-2, (k, rel(avg[k], ref_avg[k]))
        assert int(avg["bn1.num_batches_tracked"]) == int(float(ref_avg["bn1.num_batches_tracked"]))
    exp = server.FedPavg([_fsd(out[r][1]) for r in range(W)], sizes)
    for k in ("conv1.weight", "layer3.1.conv2.weight", "bn2.running_var"):
        assert torch.equal(out[0][2][k], exp[k]), k                                # exact vs FedPavg of the HIP-side local models


def _fsd(sd):
    return {k: v.to(DEV) for k, v in sd.items()}


def test_config5_scale_properties():
    """Config 5's real head geometry on one GPU: 8 ranks (threads), 85 000 classes sharded 10 625 per rank, sample_rate 0.1 -> 1 062
    sampled rows, global batch 8 x 16 (iresnet18 backbones keep it light): size-independent properties of the hybrid step."""
    W, B, C = 8, 16, 85000
    n_ids = C // W

    def run(c):
        rank = c.rank
        bb = _closed_form_backbone(float(rank + 1))
        pfc = PartialFC(rank=rank, local_rank=0, world_size=W, batch_size=B, resume=False, margin_softmax=losses.CosFace(s=30, m=0.4),
                        num_classes=C, sample_rate=0.1, embedding_size=512, prefix="/tmp", comm=c)
        assert (pfc.num_local, pfc.num_sample) == (10625, 1062)
        w0 = pfc.weight.clone()
        bm = client.BCE_module(512, 64, 1).to(DEV)
        tr = client.ShardedHeadTrainer(bb, pfc, bm, id_base=rank * n_ids, lr=0.01)
        lab = (R.closed_form_labels(B, 64, tag=rank) + rank * n_ids).to(DEV)
        out = [tr.step(R.closed_form_images(B, tag=float(rank + st)).to(DEV), lab) for st in range(2)]
        torch.cuda.synchronize()
        idx = pfc.index
        assert idx.numel() == 1062 and bool((idx[1:] > idx[:-1]).all())
        mine = lab[(lab >= pfc.class_start) & (lab < pfc.class_start + pfc.num_local)] - pfc.class_start
        assert bool(torch.isin(mine, idx).all())                                  # this shard's positives are always sampled
        changed = (pfc.weight != w0).any(dim=1)
        assert 1062 <= int(changed.sum()) <= 2 * 1062                            # two steps, two sampled sets, nothing else touched
        tr.end_round(100.0 + rank, sum(100.0 + r for r in range(W)))
        torch.cuda.synchronize()
        return [float(o[1]) for o in out], bb._flat_state.clone()
    res = ThreadComm.run(W, run, device=DEV)
    for r in range(1, W):
        assert res[r][0] == res[0][0]                                             # one global softmax: same loss everywhere
        assert torch.equal(res[r][1], res[0][1])                                  # one averaged model after the exchange
    assert all(np.isfinite(v) and 5.0 < v < 40.0 for v in res[0][0])             # ~ ln(8 * 1062) + s * m for random weights


# ---- BASELINE configs 4 and 5 at their FULL size (iresnet100, batch 128 per client) on one GPU ---------------------------------------
_R100_SD = {}


def _r100_state(rank):
    """iresnet100 closed-form state, generated ONCE (8 s of hashing on the CPU) and made client-specific by a rank-dependent scale of the stem
    and bn weights (different local models, so 'one averaged model' means something)."""
    if "sd" not in _R100_SD:
        _R100_SD["sd"] = R.closed_form_state_dict(R.IRESNET_LAYERS["iresnet100"])
    sd = {k: v.clone() for k, v in _R100_SD["sd"].items()}
    sd["conv1.weight"] = sd["conv1.weight"] * (1.0 + 0.05 * rank)
    for k in sd:
        if k.endswith("bn2.weight") or k.endswith("bn1.bias"):
            sd[k] = sd[k] + 0.01 * rank
    return sd


def test_config4_full_size():
    """BASELINE config 4 at full size: 4 clients (thread-ranks) x iresnet100 + CosFace over a dense 1000-class head, batch 128, ONE local
    train step each (client.py:537-550), then the round's exchange (server.fedavg_all_reduce, ONE all-reduce of the 261 MB flat state,
    replacing server.py:25-34): every rank ends with the same model and it is FedPavg of the four local models BIT FOR BIT (the in-process
    communicator sums in ascending rank order, like the reference loop)."""
    W, B, C = 4, 128, 1000
    sizes = [300.0, 100.0, 250.0, 50.0]
    imgs = [R.closed_form_images(B, tag=float(r)).to(DEV) for r in range(W)]
    labs = [R.closed_form_labels(B, C, tag=r).to(DEV) for r in range(W)]
    for r in range(W):
        _r100_state(r)            # (the shared base state, built outside the threads)

    def run(c):
        rank = c.rank
        bb = backbones.iresnet100(False, dropout=0, fp16=True)
        bb.load_state_dict(_r100_state(rank))
        bb = bb.to(DEV)
        fc = R.head_fc(C).to(DEV)
        tr = client.FusedTrainer(bb, fc, "CosFace", 30.0, 0.4, lr=0.01, aux_slot=rank)
        loss = float(tr.step(imgs[rank], labs[rank]))
        tr.finish()
        torch.cuda.synchronize()
        local = client.flat_state_dict(bb, clone=True)
        calls = []
        real = c.all_reduce
        c.all_reduce = lambda t, op="sum": (calls.append(t.numel()), real(t, op))[1]
        server.fedavg_all_reduce(bb, sizes[rank], sum(sizes), c)
        torch.cuda.synchronize()
        assert calls == [bb._flat_state.numel()]                     # ONE collective: the whole state, in place
        return loss, local, bb._flat_state.clone(), {k: bb.state_dict()[k].clone() for k in ("bn1.num_batches_tracked", "layer3.9.bn2.num_batches_tracked")}
    out = ThreadComm.run(W, run, device=DEV)
    assert all(np.isfinite(o[0]) and 5.0 < o[0] < 40.0 for o in out), [o[0] for o in out]
    assert len({o[0] for o in out}) == W                              # four different clients
    expect = server.FedPavg([o[1] for o in out], sizes)
    for r in range(W):
        assert torch.equal(out[r][2], out[0][2]), r                   # one averaged model on every rank
    m = backbones.iresnet100(False, dropout=0, fp16=True).to(DEV)
    m.load_state_dict(expect)
    n_float = m._flat_params.numel() + m._flat_bufs.numel()
    assert torch.equal(out[0][2][:n_float], m._flat_state[:n_float])   # == FedPavg of the four local models, bit for bit
    for k, v in out[0][3].items():
        assert int(v) == int(float(expect[k])), k                     # counters: float average truncated to int64 (F9)


def test_config5_full_size_properties():
    """BASELINE config 5 at full size on one GPU: 8 clients (thread-ranks) x iresnet100, batch 128 each (global batch 1024), ONE CosFace PartialFC
    over 85 000 identities class-sharded 10 625 per rank (sample_rate 0.1 -> 1 062 sampled rows per shard, partial_fc.py:118-176) + a private BCE
    head per client; 2 steps, then the FedAvg exchange (server.py:25-34).  Size-independent properties: one global softmax (the same loss on every
    rank), sorted sampled index sets that contain the shard's positives, only sampled rows touched, one averaged model == FedPavg of the local
    models, everything finite.  ~10 GB of arenas per client: 80 GB of the 288."""
    W, B, C = 8, 128, 85000
    n_ids = C // W
    sizes = [100.0 + r for r in range(W)]
    for r in range(W):
        _r100_state(r)
    imgs = [[R.closed_form_images(B, tag=float(r + st)).to(DEV) for st in range(2)] for r in range(W)]

    def run(c):
        rank = c.rank
        bb = backbones.iresnet100(False, dropout=0, fp16=True)
        bb.load_state_dict(_r100_state(rank))
        bb = bb.to(DEV)
        pfc = PartialFC(rank=rank, local_rank=0, world_size=W, batch_size=B, resume=False, margin_softmax=losses.CosFace(s=30, m=0.4),
                        num_classes=C, sample_rate=0.1, embedding_size=512, prefix="/tmp", comm=c)
        assert (pfc.num_local, pfc.num_sample) == (10625, 1062)
        w0 = pfc.weight.clone()
        bm = client.BCE_module(512, 256, 1).to(DEV)
        tr = client.ShardedHeadTrainer(bb, pfc, bm, id_base=rank * n_ids, lr=0.01, aux_slot=rank)       # (own weight-gradient stream per client)
        lab = (R.closed_form_labels(B, 256, tag=rank) + rank * n_ids).to(DEV)
        outs = [tr.step(imgs[rank][st], lab) for st in range(2)]
        torch.cuda.synchronize()
        idx = pfc.index
        assert idx.numel() == 1062 and bool((idx[1:] > idx[:-1]).all())
        mine = lab[(lab >= pfc.class_start) & (lab < pfc.class_start + pfc.num_local)] - pfc.class_start
        assert bool(torch.isin(mine, idx).all())
        changed = (pfc.weight != w0).any(dim=1)
        assert 1062 <= int(changed.sum()) <= 2 * 1062
        assert bool(torch.isfinite(bb._flat_params).all()) and bool(torch.isfinite(pfc.weight).all())
        local = client.flat_state_dict(bb, clone=True)
        tr.end_round(sizes[rank], sum(sizes))
        torch.cuda.synchronize()
        return [(float(o[0]), float(o[1]), float(o[2])) for o in outs], local, bb._flat_state.clone()
    res = ThreadComm.run(W, run, device=DEV)
    for r in range(1, W):
        assert [o[1] for o in res[r][0]] == [o[1] for o in res[0][0]]           # one global softmax: the same sharded-head loss on every rank
        assert torch.equal(res[r][2], res[0][2])                                # one averaged model after the exchange
    assert all(np.isfinite(v) for o in res for t in o[0] for v in t)
    assert all(5.0 < o[1] < 40.0 for o in res[0][0]), res[0][0]                  # ~ ln(8 * 1062) + s * m for random class weights
    assert len({res[r][0][0][2] for r in range(W)}) == W                        # the private BCE losses differ per client
    expect = server.FedPavg([res[r][1] for r in range(W)], sizes)
    m = backbones.iresnet100(False, dropout=0, fp16=True).to(DEV)
    m.load_state_dict(expect)
    n_float = m._flat_params.numel() + m._flat_bufs.numel()
    assert torch.equal(res[0][2][:n_float], m._flat_state[:n_float])             # == FedPavg of the eight local models, bit for bit


def test_config3_full_size_properties():
    """BASELINE config 3 once at full size: iresnet100 + ArcFace + PartialFC sample_rate 0.1 over 85 000 identities, B = 128."""
    C, B = 85000, 128
    m = backbones.iresnet100(False, dropout=0, fp16=True)
    m.load_state_dict(R.closed_form_state_dict(R.IRESNET_LAYERS["iresnet100"]))
    m = m.to(DEV)
    pfc = PartialFC(rank=0, local_rank=0, world_size=1, batch_size=B, resume=False, margin_softmax=losses.ArcFace(s=30, m=0.4),
                    num_classes=C, sample_rate=0.1, embedding_size=512, prefix="/tmp")
    w0 = pfc.weight.clone()
    p0 = m._flat_params.clone()
    tr = client.FusedTrainer(m, pfc, "ArcFace", 30.0, 0.4, lr=0.01)
    lab = ((R.closed_form_labels(B, C, tag=1) * 977) % C).to(DEV)
    ls = [float(tr.step(R.closed_form_images(B, tag=float(st)).to(DEV), lab)) for st in range(3)]
    tr.finish()
    torch.cuda.synchronize()
    assert pfc.index.numel() == 8500 and bool((pfc.index[1:] > pfc.index[:-1]).all()) and bool(torch.isin(lab, pfc.index).all())
    assert all(np.isfinite(v) and 5.0 < v < 45.0 for v in ls), ls
    changed = (pfc.weight != w0).any(dim=1)
    assert 8500 <= int(changed.sum()) <= 3 * 8500
    assert bool(torch.isfinite(m._flat_params).all()) and not torch.equal(m._flat_params, p0)
    assert int(m.state_dict()["bn1.num_batches_tracked"]) == 3 + 3            # closed-form state starts at 3 + (idx % 5) = 3 for bn1 ... +3 steps


def test_many_clients_share_one_backbone_memory_flat():
    """Server.train trains clients one after another in one process (server.py:283); every Client must not keep its own
    backbone + activation arenas resident.  8 clients: allocated memory after client 2..8 equals that after client 1."""
    class Args:
        network, loss, local_epoch, output_dir, BCE_local, aggr_alg = "iresnet18", "CosFace", 1, "/tmp", False, "FedAvg"

    class DS:
        ID_base = 0

    class Loader(list):
        dataset = DS()
    n = 8

    class Data:
        train_class_sizes = [10] * n
        train_dataset_sizes = [8] * n
        train_loaders = [Loader([(R.closed_form_images(4, tag=float(c)), R.closed_form_labels(4, 10, tag=c))]) for c in range(n)]
    from fedfr_amd.config import config as cfg
    cfg.lr = 0.01
    clients = [client.Client(c, Args, Data, device=DEV) for c in range(n)]
    sd = R.closed_form_state_dict(R.IRESNET_LAYERS["iresnet18"], tag=2.0)
    mem = []
    streams_before = set(client._AUX_STREAMS)          # (other tests of the session may have opened slots of their own: concurrent clients, thread-ranks)
    for c in clients:
        c.backbone_state_dict = sd
        c.train(0)
        c.backbone_state_dict = {k: v.cpu() for k, v in c.backbone_state_dict.items()}      # what a driver keeps per client
        torch.cuda.synchronize()
        mem.append(torch.cuda.memory_allocated(DEV))
    assert max(mem[1:]) - mem[0] < 8 << 20, mem                 # < 8 MB drift over 7 more clients (one backbone's arenas are ~GBs)
    assert len({id(c._get_backbone()) for c in clients}) == 1
    assert len(set(client._AUX_STREAMS) - streams_before) <= 1   # one auxiliary stream per device (slot 0), not one per trainer


def test_parallel_clients_round_equals_sequential():
    """``args.parallel_clients = 2``: two clients of a round train CONCURRENTLY on the GPU (own stream pair + resident backbone each).
    Clients are independent and every kernel is deterministic, so the round's aggregate is bit-identical to the sequential round run with
    the same kernel selection (Server.train switches the paired weight-gradient kernel on for concurrent clients: the sequential run
    is made with option wgrad9p = 1 too), and equal to the default sequential round up to fp32 summation order in the weight gradients."""
    class DS:
        ID_base = 0

    class Loader(list):
        dataset = DS()
    n = 4

    class Data:
        train_class_sizes = [10] * n
        train_dataset_sizes = [300, 100, 200, 50]
        train_loaders = [Loader([(R.closed_form_images(4, tag=float(c * 2 + s_)), R.closed_form_labels(4, 10, tag=c + s_)) for s_ in range(2)])
                         for c in range(n)]
    from fedfr_amd.config import config as cfg
    cfg.lr = 0.01
    outs = []
    for par, w9p in ((1, 1), (2, None), (1, 0)):
        class Args:
            network, loss, local_epoch, output_dir, BCE_local, aggr_alg = "iresnet18", "CosFace", 1, "/tmp", False, "FedAvg"
            parallel_clients = par
        clients = [client.Client(c, Args, Data, device=DEV) for c in range(n)]
        for c in clients:
            c.fc_module.fc.data = R.head_fc(10, seed=30 + c.cid)
        srv = server.Server(clients, Data, Args, device=DEV)
        srv.federated_model.load_state_dict(R.closed_form_state_dict(R.IRESNET_LAYERS["iresnet18"], tag=2.0))
        prev_w9p = _C.get_option("wgrad9p")
        if w9p is not None:
            _C.call("fedfr_set_option", b"wgrad9p", w9p)
        try:
            loss = srv.train()
        finally:
            _C.call("fedfr_set_option", b"wgrad9p", prev_w9p)
        torch.cuda.synchronize()
        outs.append((loss, {k: v.clone() for k, v in srv.federated_model.state_dict().items()}, [c.get_train_loss() for c in clients]))
    assert outs[0][2] == outs[1][2]                                   # per-client mean losses
    assert outs[0][0] == outs[1][0]
    for k, v in outs[0][1].items():
        assert torch.equal(v, outs[1][1][k]), k
    # the library default for a sequential round (single-layer weight-gradient kernel): same round up to summation order
    worst = max(float((outs[2][1][k].float() - v.float()).norm() / (v.float().norm() + 1e-12)) for k, v in outs[0][1].items() if v.dtype.is_floating_point)
    # (fp16 storage: the same perturbation crosses 8x more rounding boundaries over the round's steps — measured 1.7e-4 state, 7e-6 loss)
    tol = 1e-5 if _C.storage_dtype() == torch.bfloat16 else 5e-4
    assert worst < tol and abs(outs[2][0] - outs[0][0]) < tol * abs(outs[0][0]), (worst, outs[2][0], outs[0][0])


def _rccl_world1_worker(rank, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=DEV)
    try:
        c = TorchDistComm()
        assert c.bitwise_gather and (c.rank, c.world_size) == (0, 1)
        t = torch.arange(12, dtype=torch.float32, device=DEV).view(4, 3)
        assert torch.equal(c.all_gather(t), t)                                   # all_gather_into_tensor
        assert torch.equal(c.all_reduce(t.clone(), "sum"), t) and torch.equal(c.all_reduce(t.clone(), "max"), t)
        assert torch.equal(c.reduce_scatter(t.clone()), t)                       # reduce_scatter_tensor
        lab = torch.tensor([2 ** 40 + 3, 7, -1], dtype=torch.int64, device=DEV)
        packed = torch.cat([torch.ones(3, 2, device=DEV), lab.view(torch.int32).view(3, 2).view(torch.float32)], dim=1)
        assert torch.equal(c.all_gather(packed)[:, 2:].contiguous().view(torch.int32).view(-1).view(torch.int64), lab)
        m = _closed_form_backbone(1.0)
        before = m._flat_state.clone()
        nbt = m._flat_nbt.clone()
        assert server.fedavg_all_reduce(m, 5.0, server.exchange_data_sizes(5.0, c), c) == 1.0
        torch.cuda.synchronize()
        P = m._flat_params.numel() + m._flat_bufs.numel()
        assert torch.equal(m._flat_state[:P], before[:P]) and torch.equal(m._flat_nbt, nbt)      # weight 1.0: the exchange is the identity
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_torch_dist_comm_over_rccl_world1():
    """The RCCL code paths of the exchange layer (all_gather_into_tensor, reduce_scatter_tensor, in-place all_reduce of the model state) at
    world size 1 — what one GPU can run of them; multi-rank semantics are covered by the gloo / thread tests above."""
    import torch.multiprocessing as mp
    mp.spawn(_rccl_world1_worker, args=(29683,), nprocs=1, join=True)


def _config5_gloo_worker(rank, port):
    """rank of a 2-PROCESS config-5 run (torch.distributed over gloo, both ranks on cuda:0): product path, then the oracle's composition
    over the same process group on the CPU."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        W, B, n_ids, steps, lr, rate = 2, 4, 10, 2, 0.01, 0.5
        C = W * n_ids + 1                                              # 21 classes -> shards 11 / 10
        layers = R.IRESNET_LAYERS["iresnet18"]
        sizes = [30.0, 10.0]
        nl, _ = R.pfc_shard(C, W, rank)
        sd0 = R.closed_form_state_dict(layers, tag=float(rank + 1))
        w0 = R.closed_form((nl, 512), 0.071 + 0.003 * rank, 1.1, 0.01)
        bce0 = R.head_fc(n_ids, seed=20 + rank)
        perms = [R.closed_form((nl,), 0.77 + 0.1 * st, 0.3 + rank, 0.5, 0.5) for st in range(steps)]
        batches = _config5_inputs(rank, W, B, n_ids, steps)
        # ---- product path
        comm = TorchDistComm()
        bb = backbones.iresnet18(False, dropout=0, fp16=True)
        bb.load_state_dict(sd0)
        bb = bb.to(DEV)
        pfc = PartialFC(rank=rank, local_rank=0, world_size=W, batch_size=B, resume=False, margin_softmax=losses.CosFace(s=30, m=0.4),
                        num_classes=C, sample_rate=rate, embedding_size=512, prefix="/tmp")          # default comm = the process group
        assert isinstance(pfc.comm, TorchDistComm)
        pfc.weight.copy_(w0.to(DEV))
        pfc.weight_mom.zero_()
        bm = client.BCE_module(512, n_ids, 1).to(DEV)
        bm.weight.data = bce0.clone().to(DEV)
        tr = client.ShardedHeadTrainer(bb, pfc, bm, id_base=rank * n_ids, lr=lr)
        ls = []
        for st, (imgs, lab) in enumerate(batches):
            loss, cos, b = tr.step(imgs.to(DEV), lab.to(DEV), perm=perms[st].to(DEV))
            ls.append((float(loss), float(cos), float(b)))
        local = {k: v.clone() for k, v in bb.state_dict().items()}
        tr.end_round(sizes[rank], server.exchange_data_sizes(sizes[rank], comm))
        torch.cuda.synchronize()
        avg = bb.state_dict()
        # ---- oracle over the same group (CPU tensors)
        sd = {k: v.clone() for k, v in sd0.items()}
        w, mom = w0.clone(), torch.zeros_like(w0)
        bce = {"conv_w": torch.eye(512), "conv_b": torch.zeros(512), "weight": bce0.clone(), "bias": torch.zeros(n_ids)}
        ref = R.config5_client_steps(sd, w, mom, bce, batches, layers, R.DistComm(), B, C, rate, rank * n_ids, lr, perms)
        for (l, c_, b), (rl, rc, rb) in zip(ls, ref):
            assert abs(c_ - rc) < 1e-2 * abs(rc) and abs(b - rb) < 1e-2 * abs(rb) and abs(l - rl) < 1e-2 * abs(rl), (ls, ref)
        for k in ("conv1.weight", "bn1.weight", "layer2.0.downsample.0.weight", "fc.bias"):
            assert rel(local[k], sd[k]) < 1e-2, (k, rel(local[k], sd[k]))
        assert rel(pfc.weight, w) < 3e-2
        # FedAvg of the two local models, gathered through the group for the check
        mine = local["conv1.weight"].contiguous().cpu()
        both = [torch.zeros_like(mine) for _ in range(W)]
        dist.all_gather(both, mine)
        exp = np.float32(sizes[0] / 40.0) * both[0] + np.float32(sizes[1] / 40.0) * both[1]
        assert torch.equal(avg["conv1.weight"].cpu(), exp)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_config5_two_processes_gloo_vs_oracle():
    """Config 5 as two PROCESSES (ranks share cuda:0; exchange through torch.distributed / gloo, i.e. TorchDistComm inside PartialFC
    and fedavg_all_reduce) against the oracle's composition run over the same process group."""
    import torch.multiprocessing as mp
    mp.spawn(_config5_gloo_worker, args=(29691,), nprocs=2, join=True)


def test_bench_self_launch_two_ranks_rehearsal():
    """`python bench.py --gpus 2` with NO launcher (what a driver that reuses its N = 1 command line runs): bench.py starts the two ranks
    itself before touching the GPU, the ranks find each other, run local steps + the FedAvg exchange, and rank 0's single JSON line comes
    back through the parent.  On this one-GPU box the ranks share cuda:0 and use gloo (RCCL refuses two ranks per device) — the rehearsal
    switches of bench.py; the launcher, the rendezvous, `rccl_ranks` and the exchange are the code the 8-GPU run uses."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(FEDFR_BENCH_SHARE_GPU="1", FEDFR_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--arch", "iresnet50",
                        "--batch", "8", "--no-cpu-baseline", "--no-profile"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["config"]["clients"] == 2 and d["steps"] == 2
    assert d["value"] > 0 and d["fedavg_round_ms"] >= d["fedavg_exchange_ms"] > 0
    assert "REHEARSAL" in d["data"] and d["collective_backend"] == "gloo"
