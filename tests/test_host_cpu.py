"""Host-side logic that needs no GPU: reference-compatible module tree / state_dict, flat-storage aliasing,
loud failure without a device, FedAvg collective plumbing over 2 gloo ranks."""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from fedfr_amd import backbones, client, server


def test_state_dict_keys_shapes_and_aliasing():
    layers = R.IRESNET_LAYERS["iresnet18"]
    sd = R.closed_form_state_dict(layers)
    m = backbones.iresnet18(False, dropout=0, fp16=True)
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd)
    out = m.state_dict()
    for k, v in sd.items():
        assert out[k].dtype == v.dtype and out[k].shape == v.shape and torch.equal(out[k], v), k
    # conv weights are channels_last views (KRSC storage) of one flat tensor; SGD/FedAvg see one buffer
    w = m.layer2[0].conv1.weight
    assert w.shape == (128, 64, 3, 3) and w.stride() == (576, 1, 192, 64)
    flat, bufs, nbt = m.flat_state()
    assert w.data_ptr() >= flat.data_ptr() and w.data_ptr() < flat.data_ptr() + flat.numel() * 4
    assert not m.features.weight.requires_grad and m.fc.weight.requires_grad
    assert sum(p.numel() for p in m.parameters()) == 24025600
    # in-place updates through a stock optimizer hit the flat storage
    opt = torch.optim.SGD(m.parameters(), lr=1.0)
    m.bn1.weight.grad = torch.ones_like(m.bn1.weight)
    before = float(flat.sum())
    opt.step()
    assert abs(float(flat.sum()) - (before - 64.0)) < 1e-2
    m2 = copy.deepcopy(m)
    m2.conv1.weight.data.zero_()
    assert float(m.conv1.weight.abs().sum()) > 0 and float(m2._flat_params[:1728].abs().sum()) == 0.0


def test_no_cpu_fallback():
    m = backbones.iresnet18()
    with pytest.raises(RuntimeError, match="MI355X"):
        m(torch.zeros(2, 3, 112, 112))
    with pytest.raises(RuntimeError, match="GPU"):
        server.FedPavg([{"a": torch.ones(3)}, {"a": torch.ones(3)}], [1, 1])
    with pytest.raises(ValueError):
        backbones.iresnet50(pretrained=True)


def _fedavg_state(arch, tag):
    if arch == "sphnet":
        return R.sphere_state_dict(20, tag=tag)
    return R.closed_form_state_dict(R.IRESNET_LAYERS[arch], tag=tag)


def _fedavg_worker(rank, port, tmp, arch="iresnet18"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        m = backbones.sphnet(type=20) if arch == "sphnet" else backbones.iresnet18()
        m.load_state_dict(_fedavg_state(arch, float(rank + 1)))
        sizes = [300.0, 100.0]

        def cpu_scale(dst, src, w, accumulate):      # test double for the HIP axpy: the collective plumbing is under test
            assert not accumulate
            dst.copy_(src * np.float32(w))
        from fedfr_amd.comm import TorchDistComm
        comm = TorchDistComm()
        total = server.exchange_data_sizes(sizes[rank], comm)          # round start: Σ n_j becomes known to every rank
        assert total == 400.0
        calls = []
        real_all_reduce = comm.all_reduce
        comm.all_reduce = lambda t, op="sum": (calls.append(t.numel()), real_all_reduce(t, op))[1]
        w = server.fedavg_all_reduce(m, sizes[rank], total, comm, _axpy=cpu_scale,
                                     _i64=lambda acc, src, w_: acc.copy_(src.float() * np.float32(w_)),
                                     _trunc=lambda acc, dst: dst.copy_(acc.to(torch.int64)))
        assert abs(w - sizes[rank] / 400.0) < 1e-12
        assert calls == [m._flat_state.numel()]                        # ONE collective: the whole flat state, in place
        torch.save({k: v.clone() for k, v in m.state_dict().items()}, os.path.join(tmp, "r%d.pt" % rank))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("arch,port", [("iresnet18", 29655), ("sphnet", 29656)])
def test_fedavg_all_reduce_two_ranks_gloo(tmp_path, arch, port):
    """sphnet has no BatchNorm counters: the float image of the counters is an EMPTY slice of the state (data_ptr() of
    an empty tensor is 0, so the split point must not come from pointer differences)."""
    import torch.multiprocessing as mp
    mp.spawn(_fedavg_worker, args=(port, str(tmp_path), arch), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    ref = R.fedpavg([_fedavg_state(arch, 1.0), _fedavg_state(arch, 2.0)], [300, 100])
    for k in a:
        assert torch.equal(a[k], b[k]), k                   # every client ends the round with the same model
        if a[k].is_floating_point():
            torch.testing.assert_close(a[k], ref[k], rtol=1e-6, atol=1e-7)
        else:
            assert int(a[k]) == int(ref[k])                 # F9: float average truncated back to int64


def test_checkpoint_interchange_with_reference():
    """checkpoints in the reference's `torch.save(backbone.state_dict())` format (client.py:484-495, server.py:148)
    load into this package's backbone and vice versa, key for key and bit for bit (runs where /root/reference is present)."""
    import os
    import sys
    import tempfile
    if not os.path.isdir("/root/reference/backbones"):
        pytest.skip("reference checkout not present on this machine")
    sys.path.insert(0, "/root/reference")
    try:
        import importlib
        ref_backbones = importlib.import_module("backbones")
        if not hasattr(ref_backbones, "iresnet18") or ref_backbones.__file__.startswith(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))):
            pytest.skip("reference backbones shadowed")
    finally:
        sys.path.remove("/root/reference")
    from fedfr_amd import backbones as my_backbones
    torch.manual_seed(1)
    ref = ref_backbones.iresnet18(False, dropout=0, fp16=False)
    for m in ref.modules():                                     # non-trivial BN buffers
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            m.running_mean.normal_()
            m.running_var.uniform_(0.5, 2.0)
            m.num_batches_tracked.fill_(7)
    with tempfile.TemporaryDirectory() as td:
        p1, p2 = os.path.join(td, "ref.pth"), os.path.join(td, "mine.pth")
        torch.save(ref.state_dict(), p1)
        mine = my_backbones.iresnet18(False, dropout=0, fp16=True)
        missing = mine.load_state_dict(torch.load(p1))
        assert not missing.missing_keys and not missing.unexpected_keys
        sd_ref, sd_mine = ref.state_dict(), mine.state_dict()
        assert list(sd_ref.keys()) == list(sd_mine.keys())
        for k in sd_ref:
            assert sd_ref[k].shape == sd_mine[k].shape and sd_ref[k].dtype == sd_mine[k].dtype, k
            assert torch.equal(sd_ref[k], sd_mine[k]), k
        torch.save(mine.state_dict(), p2)
        ref2 = ref_backbones.iresnet18(False, dropout=0, fp16=False)
        ref2.load_state_dict(torch.load(p2))                    # strict
        for k, v in ref2.state_dict().items():
            assert torch.equal(v, sd_ref[k]), k


def test_asm_read_hazard_checker(tmp_path):
    """tools/check_asm_reads.py (static check of the hand-scheduled `ds_read_b64_tr_b16` reads in wgrad9.hip / gemm_tn_glds.hip):
    a register delivered by an in-flight read may not be touched before the next `s_waitcnt lgkmcnt(0)`."""
    import subprocess, sys, os
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_asm_reads.py")
    good = tmp_path / "good.s"
    good.write_text("kern:\n\tds_read_b64_tr_b16 v[10:11], v4 offset:0\n\tv_mfma_f32_16x16x32_bf16 v[20:23], v[0:3], v[4:7], v[20:23]\n"
                    "\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v30, v10\n")
    bad = tmp_path / "bad.s"
    bad.write_text("kern:\n\tds_read_b64_tr_b16 v[10:11], v4 offset:0\n\tv_mov_b32_e32 v30, v10\n\ts_waitcnt lgkmcnt(0)\n")
    assert subprocess.run([sys.executable, tool, str(good)], capture_output=True).returncode == 0
    r = subprocess.run([sys.executable, tool, str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "pending" in r.stdout


def test_oracle_dropout_is_torch_dropout_with_injected_mask():
    """the oracle's masked dropout (iresnet_forward(dropout_mask=...)) == nn.Dropout(p) as the reference applies it (iresnet.py:169) when
    torch draws that same mask."""
    p = 0.4
    x = R.closed_form((4, 25088), 0.013, 0.2, 1.0)
    torch.manual_seed(5)
    y = torch.nn.functional.dropout(x.clone(), p=p, training=True)
    mask = (y != 0).float()
    assert torch.allclose(x * mask / (1 - p), y)


# ---- exchange layer (fedfr_amd/comm.py) on CPU tensors ----------------------------------------------------------------------------
def _comm_script(c):
    """the four verbs against their definitions; returns per-rank results for cross-rank checks."""
    W, r = c.world_size, c.rank
    g = c.all_gather(torch.full((2, 3), float(r)))
    assert g.shape == (2 * W, 3) and all(float(g[2 * i, 0]) == i for i in range(W))
    s = c.all_reduce(torch.arange(4, dtype=torch.float32) * (r + 1), "sum")
    assert torch.equal(s, torch.arange(4, dtype=torch.float32) * (W * (W + 1) / 2))
    m = c.all_reduce(torch.tensor([float(r), -float(r)]), "max")
    assert torch.equal(m, torch.tensor([float(W - 1), 0.0]))
    full = torch.arange(W * 2 * 3, dtype=torch.float32).view(W * 2, 3) + r
    rs = c.reduce_scatter(full)
    exp = (torch.arange(W * 2 * 3, dtype=torch.float32).view(W * 2, 3) * W + W * (W - 1) / 2)[2 * r: 2 * r + 2]
    assert torch.equal(rs, exp)
    # int64 labels bit-cast into float lanes survive an all_gather (PartialFC's packed gather)
    lab = torch.tensor([2 ** 40 + r, 7 * r, -1], dtype=torch.int64)
    packed = torch.cat([torch.ones(3, 2), lab.view(torch.int32).view(3, 2).view(torch.float32)], dim=1)
    back = c.all_gather(packed)[:, 2:].contiguous().view(torch.int32).view(-1).view(torch.int64).view(W, 3)
    assert torch.equal(back[r], lab) and int(back[(r + 1) % W][0]) == 2 ** 40 + (r + 1) % W
    c.barrier()
    return r


def test_thread_comm_verbs():
    from fedfr_amd.comm import ThreadComm, SingleComm
    assert ThreadComm.run(3, _comm_script) == [0, 1, 2]
    assert ThreadComm.run(8, _comm_script) == list(range(8))
    assert _comm_script(SingleComm()) == 0
    with pytest.raises(ZeroDivisionError):                       # a failing rank surfaces, the others are released (no hang)
        ThreadComm.run(4, lambda c: 1 / 0 if c.rank == 2 else c.barrier())


def _dist_comm_worker(rank, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        from fedfr_amd.comm import TorchDistComm, default_comm
        c = default_comm(2)
        assert isinstance(c, TorchDistComm) and (c.rank, c.world_size) == (rank, 2)
        _comm_script(c)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_torch_dist_comm_verbs_gloo():
    import torch.multiprocessing as mp
    mp.spawn(_dist_comm_worker, args=(29681,), nprocs=2, join=True)
    from fedfr_amd.comm import default_comm, SingleComm
    assert isinstance(default_comm(1), SingleComm)
    with pytest.raises(RuntimeError):
        default_comm(2)                                          # world_size 2 without a process group: loud, not silent


def test_bench_self_launch_starts_the_ranks_before_any_gpu_call(monkeypatch, capsys):
    """`python bench.py --gpus N` without a launcher (VERDICT r2 missing #1): ONE child `python -m torch.distributed.run --nproc-per-node N`
    with HSA_ENABLE_IPC_MODE_LEGACY=0, rank 0's JSON line relayed to stdout, the child's failure returned; no launch for N = 1 or when a
    launcher's environment is already there."""
    import importlib.util
    import io
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert bench.self_launch(["--steps", "3"]) is None and bench.self_launch(["--gpus", "1"]) is None
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert bench.self_launch(["--gpus", "4"]) is None                  # already under torch.distributed.run
    monkeypatch.delenv("WORLD_SIZE")
    seen = {}

    class FakeProc:
        pid = 2 ** 22 + 12345              # no such process group: a deadline's killpg must find nothing to end

        def __init__(self, cmd, env=None, stdout=None, stderr=None, text=None, cwd=None, start_new_session=False):
            seen["cmd"], seen["env"], seen["session"] = cmd, env, start_new_session
            self.stdout = io.StringIO(seen.get("out", 'NCCL version banner\n{"metric": "images/sec", "value": 1.0, "n_gpus": 4}\n'))
            self.rc = seen.get("rc", 0)

        def wait(self, timeout=None):
            if seen.get("hang"):
                raise subprocess.TimeoutExpired("bench", timeout)
            return self.rc
    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    assert bench.self_launch(["--gpus", "4", "--steps", "2"]) == 0
    cmd, env = seen["cmd"], seen["env"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["session"] is True      # own process group: a deadline ends exactly these processes
    out = capsys.readouterr()
    assert out.out.strip() == '{"metric": "images/sec", "value": 1.0, "n_gpus": 4}' and "NCCL version banner" in out.err
    seen["rc"] = 3
    assert bench.self_launch(["--gpus=2"]) == 3                        # a failing rank fails the bench
    seen["rc"], seen["out"] = 0, "no result line\n"
    assert bench.self_launch(["--gpus=2"]) == 1                        # ranks that exit 0 without a JSON line fail it too
    seen["hang"] = True                                                # no result by the deadline -> non-zero, nothing on stdout
    monkeypatch.setenv("FEDFR_BENCH_DEADLINE_S", "0.1")
    capsys.readouterr()
    assert bench.self_launch(["--gpus=2"]) == 124
    assert capsys.readouterr().out.strip() == ""


def test_flat_state_dict_lazy_views_behave_like_a_dict():
    """the lazily built FlatStateDict must answer EVERY dict method as the built dictionary would — popitem / setdefault /
    move_to_end on an unbuilt instance, and clear() must leave it empty for good (no lazy rebuild, no flat tensors left for
    load_state_dict's fast path)."""
    m = backbones.iresnet18()
    keys = list(m.state_dict().keys())
    sd = client.flat_state_dict(m)
    k, v = sd.popitem()
    assert k == keys[-1] and len(sd) == len(keys) - 1
    sd = client.flat_state_dict(m)
    assert sd.setdefault(keys[0], None) is not None and len(sd) == len(keys)
    sd = client.flat_state_dict(m)
    sd.move_to_end(keys[0])
    assert list(sd.keys())[-1] == keys[0]
    sd = client.flat_state_dict(m)
    sd.clear()
    assert len(sd) == 0 and list(sd.items()) == [] and sd.flat is None
    sd = client.flat_state_dict(m)
    sd.update({"extra": torch.zeros(1)})
    assert len(sd) == len(keys) + 1


def test_head_split_k_count_never_leaves_an_empty_range():
    """FusedTrainer's split-K degree for the head GEMMs must satisfy head_sgemm_splitk's own rule (csrc/head.hip: chunk =
    ceil32(ceil(K / splits)), chunk * (splits - 1) < K) for EVERY class count — a client's class count is arbitrary (800 used to give
    6 splits of 160 = an empty sixth range)."""
    for k in range(1, 8200):
        s = client._split_for(k)
        chunk = -(-(-(-k // s)) // 32) * 32
        assert 1 <= s <= 8 and chunk * (s - 1) < k, (k, s, chunk)
    assert client._split_for(512) == 4 and client._split_for(1000) == 7 and client._split_for(800) == 5


def test_bench_self_launch_returns_nonzero_without_a_result(tmp_path):
    """`python bench.py --gpus 2` starts its ranks as a child launcher; ranks that die (here: no GPU) or print no JSON
    line must make the parent exit non-zero and print nothing on stdout."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FEDFR_BENCH_DEADLINE_S="240")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-profile"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert '"metric"' not in r.stdout


def test_loss_scale_guard_backs_off_and_warns():
    """client._LossScaleGuard over _C.LossScaleState (the fp16 library's GradScaler stand-in, client.py:301,394-396): a disabled state (bf16 library)
    never reads the word; a set overflow word halves the DEVICE's scale, warns, is cleared and counted; the periodic check fires every
    `overflow_check_every` steps; the lowered scale outlives the trainer; `growth_interval` clean steps double it again, never beyond the
    initial scale."""
    import warnings
    import torch
    from fedfr_amd import client, _C

    class T(client._LossScaleGuard):
        pass
    dev = torch.device("cpu")
    _C._LOSS_SCALE_STATES.pop(str(dev), None)
    t = T()
    t._init_loss_scale(dev)
    st = t._ls
    assert st is _C.loss_scale_state(dev)
    assert t.loss_scale == _C.loss_scale() == 256.0 and t.guarded          # the product library (fp16 storage) is the one loaded by default
    assert t.check_overflow() is False
    st.enabled = False                                                      # what the bf16 library's state looks like: nothing is ever read
    st.word[0] = 1
    assert t.check_overflow() is False and int(st.word[0]) == 1
    st.enabled = True
    with pytest.warns(UserWarning, match="loss scale lowered to 128"):
        assert t.check_overflow() is True
    assert t.loss_scale == 128.0 and t.overflows == 1 and st.overflows == 1 and int(st.word[0]) == 0
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert t.check_overflow() is False
    t.overflow_check_every = 3
    st.word[0] = 1
    t._count_step(); t._count_step()
    assert t.loss_scale == 128.0
    with pytest.warns(UserWarning):
        t._count_step()
    assert t.loss_scale == 64.0 and t.overflows == 2 and t._steps_since_check == 0
    # the next round's trainer (a new object) starts from the lowered scale ...
    t2 = T()
    t2._init_loss_scale(dev)
    assert t2.loss_scale == 64.0 and t2.overflows == 0
    # ... and clean steps grow it back: x2 per growth_interval polled clean steps, capped at the initial value
    st.growth_interval = 5
    for _ in range(4):
        t2._count_step()
    assert t2.check_overflow() is False and t2.loss_scale == 64.0            # 4 clean steps: not yet
    t2._count_step()
    assert t2.check_overflow() is False and t2.loss_scale == 128.0           # 5
    for _ in range(20):
        t2._count_step()
        t2.check_overflow()
    assert t2.loss_scale == 256.0 == st.initial
    _C._LOSS_SCALE_STATES.pop(str(dev), None)
