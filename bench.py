#!/usr/bin/env python3
"""bench.py — headline benchmark of the FedFR per-client training hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either under a launcher — python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py
   --gpus N ... — or bare: `python bench.py --gpus N` starts that launcher itself as a child process before touching the GPU)

One "step" = one full local training step of one client on one batch of synthetic 112x112 faces:
iresnet forward + CosFace margin + softmax-CE + backward + momentum-SGD (reference hot loop client.py:537-550).
N > 1: one simulated client per GPU (weak scaling, no data-path collective inside a step); the K timed steps are one
local epoch and are followed, inside the timed region, by the round's FedAvg exchange — one RCCL all-reduce of the
flat model state over xGMI (reference server.py:25-34) — so ms_total is the FedAvg round time at N clients.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant MFMA kernel, timed live
with HIP events on its stream) and `cpu_baseline` (the CPU oracle timed on this host's cores on a bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

FWD_GFLOP_PER_IMG = {"iresnet100": 24.18, "iresnet50": 12.62,       # SURVEY.md §8(d), measured on the reference
                     "sphnet": 13.79}                               # sphere64 (run.sh): 6.897 G MACs per 112x112 image (conv 6.884 + fc 0.013)
BF16_DENSE_PEAK_TFLOPS = 2500.0                                       # MI355X_MICROARCH.md: ~2.5 PF dense bf16
HBM_PEAK_GBPS = 8000.0                                                # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
SLOT_NAMES = ["gemm_nt_kernel<128,128>", "gemm_nt_kernel<128,64>", "gemm_nt_kernel<64,128>", "gemm_nt_kernel<64,64>",
              "gemm_tn_kernel<128,128>", "gemm_tn_kernel<128,64>", "gemm_tn_kernel<64,128>", "gemm_tn_kernel<64,64>",
              "(retired slot 8)", "(retired slot 9)", "(retired slot 10)", "(retired slot 11)",      # the register-staged halo conv kernels of rounds 1-3
              "conv3x3_glds_kernel<14,14>", "conv3x3_glds_kernel<28,7>", "gemm_tn_glds_kernel<128,128>", "conv3x3 64-channel layers (c64p / glds<56,4>)",
              "wgrad9p_kernel<64x64x9, two layers per launch> (+ wgrad9_kernel<32x64x9>)", "gemm_nt_glds_kernel (7x7 and stride-2 3x3 convs)"]
HBM_SLOTS = {20: "bn_apply", 21: "bn_bwd_reduce", 22: "bn_bwd_apply", 23: "bn_finalize", 24: "bn_bwd_finalize", 25: "reduce_slabs", 26: "sgd"}
# ALGORITHMIC HBM bytes per launch on the 256->256 @14x14 layer (58 of iresnet100's 103 convs; B = 128): conv fwd/dgrad = input 12.85 MB
# (the image, once) + weights 1.18 + output 12.85; wgrad = both operands once 25.7 + the fp32 weight gradient 2.36 — its split-K slabs and
# their reduction pass are overhead, not algorithmic
# (the paired kernel, default since round 3, takes the two 3x3 layers of a block per launch: 2 x 28.1 MB)
W9P = "wgrad9p_kernel<64x64x9, two layers per launch> (+ wgrad9_kernel<32x64x9>)"
ALGORITHMIC_MB = {"conv3x3_glds_kernel<14,14>": 26.9, W9P: 56.2, "gemm_tn_glds_kernel<128,128>": 28.1}
PMC_KEYS = {"conv3x3_glds_kernel<14,14>": ("conv3x3_glds_kernel<14, 14",), W9P: ("wgrad9p_kernel", "reduce_slabs"),
            "gemm_tn_glds_kernel<128,128>": ("gemm_tn_glds_kernel", "reduce_slabs")}


# translation units (fedfr_amd/csrc) a replayed counter belongs to: a summary under profiles/ carries the git blob hashes of the sources it was
# collected on (tools/source_stamp.py); a difference to the working tree marks the field stale
_COMMON_TU = ["common.h", "gemm_dev.h"]
KERNEL_TUS = {"conv14": ["conv_glds_impl.h", "epi_mfma.h", "conv_glds8_w14.hip", "conv_glds8_fused_w14.hip", "nt_epilogue.h"] + _COMMON_TU,
              "conv28": ["conv_glds_impl.h", "epi_mfma.h", "conv_glds8_w28.hip", "conv_glds8_w28s.hip", "conv_glds8_fused_w28.hip", "conv_glds8_fused_w28s.hip",
                         "nt_epilogue.h"] + _COMMON_TU,
              "w9p": ["wgrad9p.hip", "gemm_tn_dev.h"] + _COMMON_TU,
              "tn_glds": ["gemm_tn_glds.hip", "gemm_tn_dev.h"] + _COMMON_TU}


def staleness(summary_path, tus):
    """-> {"stale": True / False, ...}: do the sources the summary was collected on still match the working tree?  A summary that predates source
    stamps (rounds 1-4) counts as stale."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("source_stamp", os.path.join(ROOT, "tools", "source_stamp.py"))
    ss = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ss)
    then = ss.parse_stamp(summary_path)
    if then is None:
        return {"stale": True, "stale_note": "summary carries no source stamp (collected before round 5): archived, not this build's"}
    now = ss.source_hashes()
    changed = sorted(t for t in tus if then.get(t) != now.get(t))
    return {"stale": bool(changed), "stale_sources": changed} if changed else {"stale": False}


def pmc_traffic():
    """HBM bytes per launch from the NEWEST committed PMC summary under profiles/ (tools/pmc_traffic.sh: separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes, FETCH x2 gfx950 correction; bench.py cannot run rocprofv3 on itself).  For the weight-gradient
    kernels the split-K slab reduction pass that follows each launch is part of the traffic.  -> {kernel: (bytes, file, detail)}"""
    import glob
    import re
    files = []
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic_*.txt")):
        m = re.search(r"r(\d+)_pmc_hbm_traffic_.*?(?:_v(\d+))?\.txt$", os.path.basename(f))
        if m:
            files.append(((int(m.group(1)), int(m.group(2) or 0)), f))
    out = {}
    for _, f in sorted(files, reverse=True):
        rows = []
        for line in open(f):
            m = re.search(r"^(.*?)\s+launches\s+\d+\s+FETCH_SIZE.*?x2 corrected:\s*([\d.]+) MB\)\s+WRITE_SIZE.*?\(\s*([\d.]+) MB\)", line)
            if m:
                rows.append((m.group(1), float(m.group(2)), float(m.group(3))))
        for kern, keys in PMC_KEYS.items():
            if kern in out:
                continue
            parts = [(n, fe, wr) for n, fe, wr in rows if any(k in n for k in keys)]
            if parts and any(keys[0] in n for n, _, _ in parts):
                if kern == W9P:        # one paired launch is followed by TWO slab reductions (one per layer)
                    parts = [(n + (" x 2" if "reduce_slabs" in n else ""), fe * (2 if "reduce_slabs" in n else 1), wr * (2 if "reduce_slabs" in n else 1))
                             for n, fe, wr in parts]
                tot = sum(fe + wr for _, fe, wr in parts)
                tus = KERNEL_TUS["conv14" if kern.startswith("conv3x3") else "w9p" if kern == W9P else "tn_glds"] + (["ew.hip"] if kern != "conv3x3_glds_kernel<14,14>" else [])
                out[kern] = (int(round(tot * 1e6)), os.path.relpath(f, ROOT), staleness(f, tus),
                             " + ".join("%s: fetch %.1f MB, write %.1f MB" % (re.sub(r"^void\s+(\(anonymous namespace\)::)?", "", n).split("<")[0].split("(")[0], fe, wr)
                                        for n, fe, wr in parts))
    return out


def pmc_mfma_busy():
    """MFMA-pipe utilisation of the dominant kernels from the NEWEST committed SQ-counter summaries under profiles/ (tools/pmc_sq.sh: three
    rocprofv3 --pmc passes over tools/conv_bench.py; bench.py cannot run rocprofv3 on itself): MfmaBusy = SQ_VALU_MFMA_BUSY_CYCLES per SIMD
    / wave lifetime in cycles (SQ_WAVE_CYCLES counts quad-cycles per wave), i.e. the share of the time a wave is resident during which its
    SIMD's matrix pipe is busy; `lds_wait` = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES.  -> {kernel: {...}}"""
    import glob
    import re
    want = {"conv3x3_glds_kernel<14,14>": ("conv14_fwd", "conv3x3_glds_kernel<14, 14, 32, 4, false"),
            "conv3x3_glds_kernel<14,14> (dgrad + BatchNorm-backward reduction epilogue)": ("conv14_fdgrad", "conv3x3_glds_kernel<14, 14, 32, 4, true"),
            "conv3x3_glds_kernel<28,7>": ("conv28_fwd", "conv3x3_glds_kernel<28, 7, 40, 4, false"),
            "conv3x3_glds_kernel<28,7> (dgrad + BatchNorm-backward reduction epilogue)": ("conv28_fdgrad", "conv3x3_glds_kernel<28, 7, 40, 4, true"),
            W9P: ("wpair14", "wgrad9p_kernel")}
    out = {}
    for name, (tag, key) in want.items():
        files = []
        for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq_%s_v*.txt" % tag)):
            m = re.search(r"r(\d+)_pmc_sq_.*_v(\d+)\.txt$", os.path.basename(f))
            if m:
                files.append(((int(m.group(1)), int(m.group(2))), f))
        if not files:
            continue
        f = sorted(files)[-1][1]
        cur, vals = None, {}
        for line in open(f):
            if line.startswith("=="):
                cur = line
                continue
            m = re.match(r"\s+(SQ_\w+)\s+per-dispatch\s+([\d.]+)", line)
            if m and cur and key in cur:
                vals.setdefault(m.group(1), float(m.group(2)))       # (the first dispatch block of the kernel in the file)
        if {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAVES"} <= set(vals):
            simds = 256 * 4
            waves_per_simd = vals["SQ_WAVES"] / simds
            life = 4.0 * vals["SQ_WAVE_CYCLES"] / vals["SQ_WAVES"]               # cycles a wave is resident
            busy = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / simds                      # MFMA-busy cycles per SIMD
            out[name] = {"mfma_busy": round(busy / life, 3), "waves_per_simd": round(waves_per_simd, 2),
                         "lds_wait": round(vals.get("SQ_WAIT_INST_LDS", 0.0) / vals["SQ_WAVE_CYCLES"], 3),
                         "parked_wait": round(vals.get("SQ_WAIT_ANY", 0.0) / vals["SQ_WAVE_CYCLES"], 3),
                         "lds_bank_conflict_per_active": round(vals.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(vals.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0), 3),
                         "source": os.path.relpath(f, ROOT)}
            out[name].update(staleness(f, KERNEL_TUS["conv14" if tag.startswith("conv14") else "conv28" if tag.startswith("conv28") else "w9p"]))
    return out


def _cpu_steps(arch, batch, steps, threads):
    from oracle import ref_cpu as R
    torch.set_num_threads(threads)
    layers = R.IRESNET_LAYERS[arch]
    sd = R.closed_form_state_dict(layers)
    g = torch.Generator().manual_seed(100)
    fc = torch.randn(1000, 512, generator=g) * 0.01
    batches = [(torch.rand(batch, 3, 112, 112, generator=g) * 2 - 1, torch.randint(0, 1000, (batch,), generator=g))
               for _ in range(steps + 1)]
    R.client_train(sd, fc, batches[:1], layers, "CosFace", 30.0, 0.4, 1e-3)          # warm-up step
    t0 = time.perf_counter()
    R.client_train(sd, fc, batches[1:], layers, "CosFace", 30.0, 0.4, 1e-3)
    return batch * steps / (time.perf_counter() - t0)


def host_cores():
    """CPUs this process may actually use: logical CPUs, capped by the affinity mask and by the cgroup CPU quota (the GPU box exposes
    every host CPU to os.cpu_count() but grants a share of them: 128 threads on a 16-CPU share ran the oracle at half the speed of 8)."""
    n = os.cpu_count() or 8
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(arch):
    """Reference CPU path (SURVEY §8d / BASELINE.md §4) = the oracle restatement, pinned to the imported reference by tests/golden:
    full fp32 train steps (fwd + margin + CE + bwd + momentum-SGD) at batch 32 — BASELINE config 1 (iresnet50) and the headline net.
    Thread count: one iresnet50 step is timed at {1/2, all} of the CPUs this process may use (host_cores(): affinity mask and cgroup
    quota, not os.cpu_count()) and the faster setting is kept; then 1 warm-up + 3 timed steps per network."""
    ncpu = host_cores()
    cands = sorted({max(1, ncpu // 2), ncpu})
    sweep = {}
    for t in cands:
        sweep[t] = round(_cpu_steps("iresnet50", 32, 1, t), 2)
        print("cpu_baseline: iresnet50 b32 on %d threads: %.2f img/s" % (t, sweep[t]), file=sys.stderr, flush=True)
    best = max(sweep, key=sweep.get)
    r50 = _cpu_steps("iresnet50", 32, 3, best)
    print("cpu_baseline: iresnet50 b32, 3 steps: %.2f img/s" % r50, file=sys.stderr, flush=True)
    if arch == "sphnet":
        # the oracle's sphere64 restatement: forward + every parameter gradient (no head, no optimiser: both are < 1 % of the step)
        from oracle import ref_cpu as R
        torch.set_num_threads(best)
        sd = R.sphere_state_dict(64, tag=1.0)
        xs, de = torch.rand(32, 3, 112, 112) * 2 - 1, torch.randn(32, 512) * 0.01
        R.sphere_step_grads(sd, xs, de, 64)
        t0 = time.perf_counter()
        for _ in range(3):
            R.sphere_step_grads(sd, xs, de, 64)
        head = 32 * 3 / (time.perf_counter() - t0)
    else:
        head = _cpu_steps(arch, 32, 3, best) if arch != "iresnet50" else r50
    print("cpu_baseline: %s b32, 3 steps: %.2f img/s" % (arch, head), file=sys.stderr, flush=True)
    # FedPavg restatement over 1 / 2 / 4 / 8 iresnet100 state_dicts (anchor for the round time; 925 tensors, 260.9 MB each)
    fedpavg_s = {}
    try:
        from oracle import ref_cpu as R
        torch.set_num_threads(best)
        sd0 = R.closed_form_state_dict(R.IRESNET_LAYERS["iresnet100"])
        sds = [{k: v.clone() for k, v in sd0.items()} for _ in range(8)]
        for n in (1, 2, 4, 8):
            R.fedpavg(sds[:n], [1000.0 + r for r in range(n)])
            t0 = time.perf_counter()
            for _ in range(3):
                R.fedpavg(sds[:n], [1000.0 + r for r in range(n)])
            fedpavg_s[str(n)] = round((time.perf_counter() - t0) / 3, 4)
        del sds
        print("cpu_baseline: FedPavg over 1/2/4/8 iresnet100 state_dicts: %s s" % fedpavg_s, file=sys.stderr, flush=True)
    except Exception as e:      # noqa: BLE001
        fedpavg_s = {"error": "%s: %s" % (type(e).__name__, e)}
    torch.set_num_threads(ncpu)
    return {"value": round(head, 3), "fedpavg_s": fedpavg_s, "unit": "images/sec", "cores": best, "kind": "port",
            "sample": "%s+CosFace fp32 batch 32, 3 full train steps after 1 warm-up, torch CPU on %d threads (this process may use %d of the "
                      "host's %d logical CPUs)" % (arch, best, ncpu, os.cpu_count() or 0),
            "config1_iresnet50_b32": {"value": round(r50, 3), "unit": "images/sec", "cores": best, "steps": 3},
            "thread_sweep_iresnet50_b32_img_per_s": {str(k): v for k, v in sweep.items()}}


def self_launch(argv):
    """`python bench.py --gpus N` (N > 1) called WITHOUT a launcher: start the N ranks ourselves — before this process has made a single
    GPU call — as ONE child `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` (fresh processes:
    nothing that has touched the GPU is ever exec'ed), with HSA_ENABLE_IPC_MODE_LEGACY=0 in their environment (RCCL's dmabuf IPC needs it on
    this image), relay rank 0's single JSON line to stdout and everything else to stderr, and return the child's exit code."""
    import socket
    import subprocess
    n = None
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if not n or n <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return None
    with socket.socket() as sk:                          # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("bench.py: launching %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    # the child runs in a session of its own, so that a deadline can end exactly the processes started here (its process group), never
    # anything matched by name.  FEDFR_BENCH_DEADLINE_S (default 1500 s): no JSON line from rank 0 by then = a hung rank / collective
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, cwd=ROOT, start_new_session=True)
    deadline = float(os.environ.get("FEDFR_BENCH_DEADLINE_S", "1500"))
    import signal
    import threading
    found = {"json": None}

    def pump():
        for line in proc.stdout:
            t = line.strip()
            if t.startswith("{") and t.endswith("}") and '"metric"' in t:
                found["json"] = t
            else:
                sys.stderr.write(line)
    th = threading.Thread(target=pump, daemon=True)
    th.start()
    timed_out = False
    try:
        rc = proc.wait(timeout=deadline)
    except subprocess.TimeoutExpired:
        timed_out = True
        print("bench.py: no result from the %d ranks within %.0f s: ending their process group" % (n, deadline), file=sys.stderr, flush=True)
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)              # pgid == the child's pid (start_new_session)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        rc = 124
    th.join(10)
    json_line = found["json"]
    if json_line is not None and not timed_out:
        print(json_line, flush=True)
    elif rc == 0:
        print("bench.py: the ranks exited 0 but rank 0 printed no JSON line", file=sys.stderr, flush=True)
        rc = 1
    return rc


def select_library(argv):
    """--lib fp16 (default): libfedfr_hip.so, the product library — IEEE fp16 storage, the reference's own AMP type (backbones/iresnet.py:159),
    inside north_star's 1e-2 on whole-network outputs.  --lib bf16: the same kernels on bf16 storage (libfedfr_hip_bf16.so, `make bf16`).
    Must act before fedfr_amd loads the library; the ranks of a self-launched run get the same argv."""
    lib = "fp16"
    for i, a in enumerate(argv):
        if a == "--lib" and i + 1 < len(argv):
            lib = argv[i + 1]
        elif a.startswith("--lib="):
            lib = a.split("=", 1)[1]
    if lib == "bf16":
        os.environ["FEDFR_HIP_LIB_NAME"] = "libfedfr_hip_bf16.so"
    elif lib != "fp16":
        raise SystemExit("bench.py: --lib must be fp16 or bf16")
    return lib


def bf16_build_leg(args):
    """SECONDARY line: the same workload on the bf16-storage build, measured by a FRESH child process (`python bench.py --lib bf16 --secondary`:
    timed region + parity leg only) started after this process's own timed region and legs are over — never a re-exec of a process that has
    touched the GPU.  -> {"ms_per_step", "value", "parity", "library"} or {"error": ...}"""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "fedfr_amd", "libfedfr_hip_bf16.so")):
        return {"error": "fedfr_amd/libfedfr_hip_bf16.so is not built (make -C fedfr_amd/csrc bf16)"}
    cmd = [sys.executable, os.path.abspath(__file__), "--lib", "bf16", "--secondary", "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--arch", args.arch, "--batch", str(args.batch), "--classes", str(args.classes), "--head", args.head]
    env = {k: v for k, v in os.environ.items() if k not in ("FEDFR_HIP_LIB_NAME", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    line = next((t for t in reversed(r.stdout.splitlines()) if t.startswith("{") and '"metric"' in t), None)
    if r.returncode != 0 or line is None:
        return {"error": "child exited %d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:])}
    d = json.loads(line)
    return {"library": d["library"], "storage": d["storage"], "ms_per_step": d["ms_per_step"], "value": d["value"], "unit": d["unit"], "steps": d["steps"],
            "warmup": d["warmup"], "parity": d["parity"], "measured_by": "fresh child process `bench.py --lib bf16 --secondary`, after this process's timed region"}


def parity_leg(dev):
    """CHECKER leg (about a second; not part of any timed region): the loaded library's whole-network error against the reference's own
    outputs on the committed fixture tests/golden/r100_b6.npz (iresnet100, 6 closed-form images; captured by tools/make_golden.py from the
    imported reference) — relative L2 error of the eval-mode and train-mode embeddings and of the train-mode cosine logits.  The fixture's
    closed-form weights / images come from the oracle's generators (oracle/ref_cpu.py: the checker's half, nothing here is timed)."""
    import numpy as np
    from oracle import ref_cpu as R
    from fedfr_amd import backbones, client
    g = np.load(os.path.join(ROOT, "tests", "golden", "r100_b6.npz"))
    batch, ncls = int(g["batch"]), int(g["num_classes"])
    m = backbones.iresnet100(False, dropout=0, fp16=True)
    m.load_state_dict(R.closed_form_state_dict(R.IRESNET_LAYERS["iresnet100"]))
    m = m.to(dev)
    x = R.closed_form_images(batch).to(dev)

    def rel(a, b):
        a, b = a.detach().double().cpu(), torch.from_numpy(np.asarray(b)).double()
        return float((a - b).norm() / (b.norm() + 1e-30))
    m.eval()
    with torch.no_grad():
        fe = m(x)
    m.train()
    fcm = client.FC_module(512, ncls, "/tmp").to(dev)
    fcm.fc.data = R.head_fc(ncls).to(dev)
    with torch.no_grad():
        ft = m(x)
        cos = fcm(ft)
    out = {"fixture": "tests/golden/r100_b6.npz (iresnet100, batch %d, reference-generated)" % batch, "metric": "relative L2 error vs the fp32 reference",
           "embeddings_eval": round(rel(fe, g["feat_eval"]), 5), "embeddings_train": round(rel(ft, g["feat_train"]), 5),
           "cosine_logits_train": round(rel(cos, g["cosine"]), 5), "north_star_tolerance_16bit": 1e-2}
    del m, fcm
    torch.cuda.empty_cache()
    return out


ROUND_LOCAL_STEPS = 8     # local steps per client in the `fedavg_round` leg (the reference: local_epoch x len(loader), client.py:537)


def round_leg(args, dev, imgs, labs, B, NC):
    """The second half of BASELINE's metric on ONE GPU: a whole FedAvg round of 1 / 2 / 4 / 8 clients through the reference-shaped objects
    (fedfr_amd.server.Server.train == reference server.py:265-338): every client receives the global model, trains ROUND_LOCAL_STEPS local steps of
    the headline workload (Client.train == client.py:511-571, loss.item() every step as the reference does), then FedPavg over the client states
    (server.py:25-34) + load_state_dict into the global model.  Clients run one after another, the way the reference runs them (server.py:283), and
    with args.parallel_clients = 2 (two clients of the round concurrently on this GPU).  -> dict for the JSON line"""
    from fedfr_amd import client, server
    from fedfr_amd.config import config as cfg
    S = ROUND_LOCAL_STEPS
    nbuf = len(imgs)

    class Args:
        network, loss, local_epoch, output_dir, BCE_local, aggr_alg, parallel_clients = args.arch, "CosFace", 1, "/tmp", False, "FedAvg", 1

    class DS:
        ID_base = 0

    class Loader(list):
        dataset = DS()

    class Data:
        train_class_sizes = [NC] * 8
        train_dataset_sizes = [1000.0 + r for r in range(8)]
        train_loaders = [Loader([(imgs[(s_ + c) % nbuf], labs[(s_ + c) % nbuf]) for s_ in range(S)]) for c in range(8)]      # resident in HBM
    lr0 = cfg.lr
    cfg.lr = 1e-3
    try:
        clients = [client.Client(c, Args, Data, device=dev) for c in range(8)]
        srv = server.Server(clients, Data, Args, device=dev)
        out = {"unit": "ms", "local_steps_per_client": S, "batch": B,
               "what": "Server.train(): per client load_state_dict(global) + %d local train steps (loss.item() per step) + state snapshot, then FedPavg + "
                       "load_state_dict; all on one MI355X" % S, "sequential": {}, "parallel_clients_2": {}}
        for par, key in ((1, "sequential"), (2, "parallel_clients_2")):
            Args.parallel_clients = par
            srv.current_client_list = list(range(min(par, 2)))
            srv.train()                                    # warm-up: arenas of the resident backbone(s), streams
            torch.cuda.synchronize()
            for n in (1, 2, 4, 8):
                srv.current_client_list = list(range(n))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                srv.train()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) * 1e3
                # the aggregation share, timed on its own over the same client states
                models = [clients[i].get_model() for i in range(n)]
                sizes = [clients[i].get_data_size() for i in range(n)]
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                srv.federated_model.load_state_dict(server.FedPavg(models, sizes))
                torch.cuda.synchronize()
                agg = (time.perf_counter() - t1) * 1e3
                out[key][str(n)] = {"round_ms": round(dt, 2), "aggregation_ms": round(agg, 3), "ms_per_client_step": round((dt - agg) / (n * S), 3)}
        return out
    finally:
        cfg.lr = lr0


def rccl_world1_round_leg(args):
    """The same round through fedavg_all_reduce over RCCL at world size 1 (the N > 1 code path: local steps, then ONE in-place all-reduce of the
    flat model state), in a FRESH child process started after this process's legs (`FEDFR_FORCE_DIST=1 bench.py --secondary`): never a re-exec of a
    process that has touched the GPU."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--secondary", "--gpus", "1", "--steps", str(ROUND_LOCAL_STEPS), "--warmup", "3",
           "--arch", args.arch, "--batch", str(args.batch), "--classes", str(args.classes), "--head", "dense"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update({"FEDFR_FORCE_DIST": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "MASTER_ADDR": "127.0.0.1"})
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        env["MASTER_PORT"] = str(sk.getsockname()[1])
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    line = next((t for t in reversed(r.stdout.splitlines()) if t.startswith("{") and '"metric"' in t), None)
    if r.returncode != 0 or line is None:
        return {"error": "child exited %d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:])}
    d = json.loads(line)
    return {"round_ms": d.get("fedavg_round_ms"), "exchange_ms": d.get("fedavg_exchange_ms"), "local_steps": d["steps"], "rccl_ranks": d.get("rccl_ranks"),
            "collective_backend": d.get("collective_backend"),
            "measured_by": "fresh child process `FEDFR_FORCE_DIST=1 bench.py --secondary --steps %d` (RCCL communicator of ONE rank: the collective's "
                           "launch + in-place pass over 261 MB, no xGMI traffic)" % ROUND_LOCAL_STEPS}


def main():
    select_library(sys.argv[1:])
    rc = self_launch(sys.argv[1:])
    if rc is not None:
        raise SystemExit(rc)
    # stdout carries exactly ONE line, the JSON result: everything else that writes to file descriptor 1 during the run (the RCCL
    # version banner librccl prints at communicator creation, warnings of native libraries) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--arch", default="iresnet100", choices=["iresnet100", "iresnet50", "sphnet"],
                    help="sphnet = sphere64, the backbone the reference's run.sh trains (backbones/sphnet.py)")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--classes", type=int, default=1000)
    ap.add_argument("--head", default="dense", choices=["dense", "pfc", "pfc-sharded"],
                    help="dense: CosFace + dense cosine head (headline); pfc: ArcFace + PartialFC sample_rate 0.1 (BASELINE config 3; use --classes 85000); "
                         "pfc-sharded: BASELINE config 5 = per-client backbone + ONE CosFace PartialFC class-sharded over all ranks (sample_rate 0.1) + "
                         "a private BCE head per client (use --classes 85000 under torch.distributed.run)")
    ap.add_argument("--lib", default="fp16", choices=["fp16", "bf16"],
                    help="storage type of activations / gradients / MFMA operands: fp16 = libfedfr_hip.so (product default), bf16 = libfedfr_hip_bf16.so (`make bf16`)")
    ap.add_argument("--secondary", action="store_true",
                    help="timed region + parity leg only (what the default run's `bf16_build` child process executes)")
    ap.add_argument("--no-bf16-build", action="store_true", help="skip the secondary bf16-build line (a child process after the legs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    args = ap.parse_args()
    secondary = args.secondary
    if secondary:
        args.no_profile = args.no_cpu_baseline = args.no_bf16_build = True

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    # rehearsal switches (NOT for measurements): FEDFR_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and FEDFR_BENCH_BACKEND=gloo replaces RCCL
    # (which refuses two ranks on one device), so that the N > 1 code path can be run end to end on a one-GPU box
    share = os.environ.get("FEDFR_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("FEDFR_BENCH_BACKEND", "nccl")
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or os.environ.get("FEDFR_FORCE_DIST") == "1"      # FORCE_DIST: exercise the RCCL path on 1 GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    rccl_ranks = None
    if use_dist:
        # what the communicator itself saw: a sum all-reduce of ones over the backend that carries the FedAvg exchange
        ones = torch.ones(1, dtype=torch.float32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        rccl_ranks = int(round(float(ones.item())))
        if rccl_ranks != dist.get_world_size() or rccl_ranks != world:
            raise SystemExit("bench.py: the process group reports %d ranks (all-reduce of ones: %d), expected %d" % (dist.get_world_size(), rccl_ranks, world))

    from fedfr_amd import _C, backbones, client, server
    torch.manual_seed(100 + rank)                                  # reference seed 100 (train.py:35)
    B, NC = args.batch, args.classes
    model = getattr(backbones, args.arch)(False, dropout=0, fp16=True).to(dev)     # reference init (random weights)
    if args.head == "pfc":
        from fedfr_amd import losses
        from fedfr_amd.partial_fc import PartialFC
        fc = PartialFC(rank=0, local_rank=dev_index, world_size=1, batch_size=B, resume=False,
                       margin_softmax=losses.ArcFace(s=30, m=0.4), num_classes=NC, sample_rate=0.1, embedding_size=512, prefix="/tmp")
        tr = client.FusedTrainer(model, fc, "ArcFace", 30.0, 0.4, lr=1e-3, momentum=0.9, weight_decay=5e-4)
    elif args.head == "pfc-sharded":
        from fedfr_amd import losses
        from fedfr_amd.comm import SingleComm, TorchDistComm
        from fedfr_amd.partial_fc import PartialFC
        comm = TorchDistComm() if (use_dist and world > 1) else SingleComm()
        pfc = PartialFC(rank=rank if world > 1 else 0, local_rank=dev_index, world_size=world, batch_size=B, resume=False,
                        margin_softmax=losses.CosFace(s=30, m=0.4), num_classes=NC, sample_rate=0.1, embedding_size=512, prefix="/tmp", comm=comm)
        n_ids = NC // world                                                       # this client's identities: [rank * n_ids, (rank + 1) * n_ids)
        bce = client.BCE_module(512, n_ids, 1).to(dev)
        tr = client.ShardedHeadTrainer(model, pfc, bce, id_base=rank * n_ids, lr=1e-3, momentum=0.9, weight_decay=5e-4)
    else:
        fc = (torch.randn(NC, 512) * 0.01).to(dev)                                # client.py:66
        tr = client.FusedTrainer(model, fc, "CosFace", 30.0, 0.4, lr=1e-3, momentum=0.9, weight_decay=5e-4)
    g = torch.Generator().manual_seed(100 + rank)
    nbuf = 4
    imgs = [(torch.rand(B, 3, 112, 112, generator=g) * 2 - 1).to(dev) for _ in range(nbuf)]   # already resident in HBM
    if args.head == "pfc-sharded":
        labs = [(torch.randint(0, NC // world, (B,), generator=g) + rank * (NC // world)).to(dev) for _ in range(nbuf)]
    else:
        labs = [torch.randint(0, NC, (B,), generator=g).to(dev) for _ in range(nbuf)]

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # the step's critical path (forward, dgrad -> BN-backward chain) runs on a HIGH-priority stream, the trainer's auxiliary
    # weight-gradient stream keeps the default priority: workgroups of critical-path kernels are dispatched first when both
    # streams have work (FEDFR_MAIN_PRIORITY=0 turns this off)
    hi = None
    if os.environ.get("FEDFR_MAIN_PRIORITY", "1") != "0":
        lo_p, hi_p = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
        hi = torch.cuda.Stream(device=dev, priority=hi_p)
        hi.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(hi)
    for i in range(args.warmup):
        tr.step(imgs[i % nbuf], labs[i % nbuf])
    total_size = sum(1000.0 + r for r in range(world))            # client sizes are pre-agreed (the server knows them): no size exchange
    if use_dist:
        server.fedavg_all_reduce(model, 1000.0 + rank, total_size)    # warm the RCCL communicator
        model.refresh_shadows(True)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = tr.step(imgs[i % nbuf], labs[i % nbuf])
    t_enqueue = time.perf_counter() - t0         # host time to ENQUEUE the K steps (no sync inside): well below dt = the GPU is never waiting for the host
    if isinstance(loss, tuple):
        loss = loss[0]
    t_local = None
    if use_dist:
        tr.finish()
        torch.cuda.synchronize()
        t_local = time.perf_counter() - t0
        server.fedavg_all_reduce(model, 1000.0 + rank, total_size)
        model.refresh_shadows(True)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt, t_local], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, dist.ReduceOp.MAX)
        dt, t_local = float(tt[0]), float(tt[1])
    final_loss = float(loss)
    # host cost of one step: enqueue time of a step issued into an EMPTY queue (the timed loop's own enqueue time includes the back-pressure of
    # a full queue).  Far below ms_per_step = the GPU never waits for the host.
    host_ms = []
    if not use_dist:
        for i in range(5):
            torch.cuda.synchronize()
            th = time.perf_counter()
            tr.step(imgs[i % nbuf], labs[i % nbuf])
            host_ms.append((time.perf_counter() - th) * 1e3)
        torch.cuda.synchronize()

    # ---- roofline leg: HIP-event timing of every MFMA GEMM launch and of the HBM-bound BatchNorm / SGD kernels, in a separate short
    # pass over the same workload
    leg_errors = {}
    roofline = None
    # rank-0-only legs re-run tr.step(): with a step that contains collectives (the sharded head at world > 1) the other ranks would have to
    # join in, so those configurations report the timed region only
    collective_step = args.head == "pfc-sharded" and world > 1
    if rank == 0 and not args.no_profile and not collective_step:
        try:
            psteps = 3
            # kernels are timed one at a time: the weight-gradient GEMMs normally share the GPU with the dgrad/BN chain on a
            # second stream, which stretches every kernel's wall time; for a per-kernel roofline the pass runs single-stream
            # (rocprof summary of the matching command: FEDFR_DUAL_STREAM=0 python bench.py ..., profiles/*_single_stream*)
            saved_aux = tr.aux_stream
            tr.finish()
            torch.cuda.synchronize()
            tr.aux_stream = None
            _C.call("fedfr_profile_enable", 1)
            for i in range(psteps):
                tr.step(imgs[i % nbuf], labs[i % nbuf])
            torch.cuda.synchronize()
            tr.aux_stream = saved_aux

            def read(slot):
                ms, n, fl = C.c_double(), C.c_longlong(), C.c_double()
                _C.call("fedfr_profile_read", slot, C.byref(ms), C.byref(n), C.byref(fl))
                return ms.value, n.value, fl.value

            def read_bytes(slot):
                b = C.c_double()
                _C.call("fedfr_profile_read_bytes", slot, C.byref(b))
                return b.value

            def family(m_, k, f, s_):
                """one MFMA kernel family of the step, with the roofline that bounds its launches: time at the dense MFMA peak vs time to move its
                ALGORITHMIC bytes (operands read once, output written once) at the HBM peak"""
                by = alg_bytes[s_]
                t_mfma, t_hbm = f / (BF16_DENSE_PEAK_TFLOPS * 1e12), by / (HBM_PEAK_GBPS * 1e9)
                e = {"kernel": SLOT_NAMES[s_], "ms_per_step": round(m_ / psteps, 3), "launches_per_step": k // psteps, "tflops": round(f / (m_ * 1e-3) / 1e12, 1),
                     "algorithmic_gb_per_step": round(by / psteps / 1e9, 3), "gbps": round(by / (m_ * 1e-3) / 1e9, 1),
                     "bound": "hbm" if t_hbm > t_mfma else "mfma"}
                e["frac_of_bound"] = round((t_hbm if t_hbm > t_mfma else t_mfma) / (m_ * 1e-3), 4)
                return e
            rows = []
            for slot in range(len(SLOT_NAMES)):
                ms, n, fl = read(slot)
                if n:
                    rows.append((ms, n, fl, slot))
            hbm_rows = {name: read(slot) for slot, name in HBM_SLOTS.items()}
            alg_bytes = {slot: read_bytes(slot) for _, _, _, slot in rows}      # (before the counters are reset)
            _C.call("fedfr_profile_enable", 0)
            rows.sort(reverse=True)
            traffic = pmc_traffic()
            sq = pmc_mfma_busy()

            def entry(ms, n, fl, slot):
                name = SLOT_NAMES[slot]
                ach = fl / (ms * 1e-3) / 1e12
                tr_bytes, tr_file, tr_stale, tr_detail = traffic.get(name, (None, None, None, None))
                e = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(ach / BF16_DENSE_PEAK_TFLOPS, 4), "traffic": tr_bytes,
                     "launches_per_step": n // psteps, "avg_launch_us": round(ms * 1e3 / n, 2), "gflop_per_launch": round(fl / n / 1e9, 3)}
                if name in sq:
                    e["mfma_busy"] = sq[name]["mfma_busy"]
                    e["mfma_busy_stale"] = sq[name].get("stale")
                    e["mfma_busy_note"] = ("SQ_VALU_MFMA_BUSY_CYCLES per SIMD / cycles a wave is resident (rocprofv3 --pmc, %s): the share of a wave's "
                                           "lifetime during which its SIMD's matrix pipe is busy; launch overhead outside the waves' lifetime is not in it" % sq[name]["source"])
                if tr_bytes is not None:
                    e["traffic_note"] = ("HBM bytes per launch on the 256->256 @14x14 layer, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, "
                                         "FETCH x2 gfx950 correction), %s: %s; algorithmic %.1f MB" % (tr_file, tr_detail, ALGORITHMIC_MB.get(name, float("nan"))))
                    e["algorithmic_bytes"] = int(ALGORITHMIC_MB[name] * 1e6) if name in ALGORITHMIC_MB else None
                    e["traffic_stale"] = tr_stale
                return e
            if rows:
                roofline = entry(*rows[0])
                roofline["timing"] = "HIP events around each launch on its stream, single-stream pass of %d steps" % psteps
                if len(rows) > 1:                      # the two 3x3 kernels (forward/dgrad and weight gradient) tie for the largest time share
                    roofline["second"] = entry(*rows[1])
                roofline["sq_counters"] = sq
                roofline["all_gemm_kernels"] = [family(m_, k, f, s_) for m_, k, f, s_ in rows]
                # the HBM-bound third of the step: BatchNorm forward / backward streaming passes (algorithmic bytes = tensors read + written once)
                fam = [hbm_rows[k] for k in ("bn_apply", "bn_bwd_reduce", "bn_bwd_apply") if hbm_rows[k][1]]
                if fam:
                    fms, fb = sum(r[0] for r in fam), sum(r[2] for r in fam)
                    roofline["hbm"] = {"bound": "hbm", "kernel": "bn_apply + bn_bwd_reduce + bn_bwd_apply", "achieved": round(fb / (fms * 1e-3) / 1e9, 1),
                                       "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(fb / (fms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                       "traffic": None, "ms_per_step": round(fms / psteps, 3), "launches_per_step": sum(r[1] for r in fam) // psteps,
                                       "algorithmic_gb_per_step": round(fb / psteps / 1e9, 2)}
                roofline["hbm_kernels"] = [{"kernel": k, "ms_per_step": round(v[0] / psteps, 3), "launches_per_step": v[1] // psteps,
                                            "gbps": round(v[2] / (v[0] * 1e-3) / 1e9, 1) if v[0] else None} for k, v in hbm_rows.items() if v[1]]
            # the same launches timed in the trainer's DEFAULT execution (weight gradients on the second stream): a kernel's events are
            # recorded on its own stream, so this is its wall time beside whatever the other stream runs (VERDICT r2 weak #7b)
            if roofline is not None and saved_aux is not None:
                _C.call("fedfr_profile_enable", 1)
                for i in range(psteps):
                    tr.step(imgs[i % nbuf], labs[i % nbuf])
                tr.finish()
                torch.cuda.synchronize()
                for ent in (roofline, roofline.get("second")):
                    if ent:
                        ms2, n2, _ = read(SLOT_NAMES.index(ent["kernel"]))
                        ent["dual_stream_avg_launch_us"] = round(ms2 * 1e3 / n2, 2) if n2 else None
                _C.call("fedfr_profile_enable", 0)
        except Exception as e:      # an auxiliary leg must never cost the headline number
            leg_errors['roofline'] = "%s: %s" % (type(e).__name__, e)
            print("bench.py: the roofline leg failed: %r" % (e,), file=sys.stderr, flush=True)

    # ---- end to end with the input path in the loop: the host hands over uint8 HWC batches (1 byte per pixel-channel, pinned memory);
    # upload on a copy stream + on-device ToTensor/Normalize/flip (fedfr_preprocess_u8, dataset.py:81-92) overlap the previous step
    end_to_end = None
    if rank == 0 and world == 1 and not args.no_profile:
        try:
            from fedfr_amd import ops
            tr.finish()
            gh = torch.Generator().manual_seed(7)
            host = [torch.randint(0, 256, (B, 112, 112, 3), dtype=torch.uint8, generator=gh).pin_memory() for _ in range(2)]
            flips = [(torch.rand(B, generator=gh) < 0.5).to(torch.uint8).pin_memory() for _ in range(2)]
            copy_stream = torch.cuda.Stream(device=dev)
            main = torch.cuda.current_stream()
            staged = [None, None]

            def stage(i):
                with torch.cuda.stream(copy_stream):
                    u8 = host[i % 2].to(dev, non_blocking=True)
                    fl = flips[i % 2].to(dev, non_blocking=True)
                    staged[i % 2] = (ops.preprocess_u8(u8, fl), u8, fl)
            esteps = max(5, min(args.steps, 20))
            stage(0)
            for i in range(2):                                             # warm-up of the staging path
                main.wait_stream(copy_stream)
                x = staged[i % 2][0]
                stage(i + 1)
                tr.step(x, labs[i % nbuf])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(2, 2 + esteps):
                main.wait_stream(copy_stream)
                x = staged[i % 2][0]
                x.record_stream(main)
                stage(i + 1)
                tr.step(x, labs[i % nbuf])
            tr.finish()
            torch.cuda.synchronize()
            dte = time.perf_counter() - t1
            end_to_end = {"value": round(B * esteps / dte, 1), "unit": "images/sec", "ms_per_step": round(dte * 1e3 / esteps, 3), "steps": esteps,
                          "note": "pinned uint8 HWC host batches (4.8 MB/step) -> H2D on a copy stream -> fedfr_preprocess_u8 -> train step; "
                                  "`value` above excludes this input path (inputs resident in HBM)"}
        except Exception as e:      # an auxiliary leg must never cost the headline number
            leg_errors['end_to_end'] = "%s: %s" % (type(e).__name__, e)
            print("bench.py: the end_to_end leg failed: %r" % (e,), file=sys.stderr, flush=True)

    # ---- two independent clients training concurrently on this GPU (Server.train with args.parallel_clients = 2): the clients of an FL
    # round are independent, and a second client's kernel chain fills the CUs one chain leaves idle between its ~1250 dependent launches
    concurrent = None
    if rank == 0 and world == 1 and not args.no_profile and args.head == "dense":
        try:
            import threading
            tr.finish()
            torch.cuda.synchronize()
            torch.manual_seed(101)
            model2 = getattr(backbones, args.arch)(False, dropout=0, fp16=True).to(dev)
            fc2 = (torch.randn(NC, 512) * 0.01).to(dev)
            lo_p, hi_p = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
            st2 = torch.cuda.Stream(device=dev, priority=hi_p)
            with torch.cuda.stream(st2):
                tr2 = client.FusedTrainer(model2, fc2, "CosFace", 30.0, 0.4, lr=1e-3, momentum=0.9, weight_decay=5e-4, aux_slot=1)
            pairs = [(tr, torch.cuda.current_stream()), (tr2, st2)]
            csteps = max(5, min(args.steps, 20))
            bar = threading.Barrier(3)

            werr = []

            def worker(k):
                t_, s_ = pairs[k]
                try:
                    torch.cuda.set_device(dev)
                    with torch.cuda.stream(s_):
                        for i in range(3):
                            t_.step(imgs[(i + k) % nbuf], labs[(i + k) % nbuf])
                        s_.synchronize()
                        bar.wait(timeout=300)
                        for i in range(csteps):
                            t_.step(imgs[(i + k) % nbuf], labs[(i + k) % nbuf])
                        t_.finish()
                        s_.synchronize()
                        bar.wait(timeout=300)
                except BaseException as e:      # noqa: BLE001 — a failing client must release the others, never hang the bench
                    werr.append(e)
                    bar.abort()
            ths = [threading.Thread(target=worker, args=(k,), daemon=True) for k in range(2)]
            for t_ in ths:
                t_.start()
            bar.wait(timeout=300)
            tc = time.perf_counter()
            bar.wait(timeout=300)
            dtc = time.perf_counter() - tc
            for t_ in ths:
                t_.join(60)
            if werr:
                raise werr[0]
            concurrent = {"clients_on_this_gpu": 2, "value": round(2 * B * csteps / dtc, 1), "unit": "images/sec",
                          "ms_per_step_per_client": round(dtc * 1e3 / csteps, 3), "steps": csteps,
                          "note": "two independent clients (own backbone, optimiser, HIP stream pair), each running the same bs=%d train step; "
                                  "aggregate over both (the lone client's kernel selection).  "
                                  "`value` above is ONE client alone" % B}
            del tr2, model2
        except Exception as e:      # an auxiliary leg must never cost the headline number
            leg_errors['concurrent'] = "%s: %s" % (type(e).__name__, e)
            print("bench.py: the concurrent leg failed: %r" % (e,), file=sys.stderr, flush=True)

    # ---- the second half of BASELINE's metric at N = 1: server-side FedAvg of 1 / 2 / 4 / 8 client states (server.py:25-34) on this GPU
    fedavg = None
    if rank == 0 and world == 1 and not args.no_profile:
        try:
            tr.finish()
            torch.cuda.synchronize()
            sds = [client.flat_state_dict(model, clone=True) for _ in range(8)]
            for k, sd in enumerate(sds):
                sd.flat[0].mul_(1.0 + 0.01 * k)             # distinct client states
            nbytes = sum(t.numel() * t.element_size() for t in sds[0].flat)
            fedavg = {"state_bytes": nbytes, "unit": "ms", "kernel": "fedfr_fedavg_multi (one pass over <= 8 client states) + fedfr_fedavg_i64",
                      "peak_gbps": HBM_PEAK_GBPS, "clients": {}}
            for n in (1, 2, 4, 8):
                ws_ = [1000.0 + r for r in range(n)]
                server.FedPavg(sds[:n], ws_)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 10
                e0.record()
                for _ in range(reps):
                    agg = server.FedPavg(sds[:n], ws_)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / reps
                alg = (n + 1) * nbytes                        # every client state read once, the aggregate written once
                fedavg["clients"][str(n)] = {"ms": round(ms, 4), "algorithmic_bytes": alg, "achieved_gbps": round(alg / (ms * 1e-3) / 1e9, 1),
                                             "frac_of_hbm_peak": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
            del sds, agg
            torch.cuda.empty_cache()
        except Exception as e:      # an auxiliary leg must never cost the headline number
            leg_errors['fedavg'] = "%s: %s" % (type(e).__name__, e)
            print("bench.py: the fedavg leg failed: %r" % (e,), file=sys.stderr, flush=True)

    # ---- ... and the ROUND itself at N = 1: 1 / 2 / 4 / 8 clients trained one after another on this GPU (the reference's way, server.py:283) and
    # two at a time, + FedPavg + load_state_dict; then the same round through fedavg_all_reduce over RCCL at world 1 (fresh child process)
    fedavg_round = None
    if rank == 0 and world == 1 and not args.no_profile and not use_dist and args.head == "dense":
        try:
            tr.finish()
            torch.cuda.synchronize()
            fedavg_round = round_leg(args, dev, imgs, labs, B, NC)
        except Exception as e:      # an auxiliary leg must never cost the headline number
            leg_errors['fedavg_round'] = "%s: %s" % (type(e).__name__, e)
            print("bench.py: the fedavg_round leg failed: %r" % (e,), file=sys.stderr, flush=True)
        if fedavg_round is not None and not args.no_bf16_build:      # (child processes: after this process's own GPU legs)
            try:
                torch.cuda.synchronize()
                fedavg_round["rccl_world1"] = rccl_world1_round_leg(args)
            except Exception as e:      # noqa: BLE001
                fedavg_round["rccl_world1"] = {"error": "%s: %s" % (type(e).__name__, e)}

    parity = None
    if rank == 0 and (secondary or not args.no_profile) and not collective_step:
        try:
            tr.finish()
            torch.cuda.synchronize()
            parity = parity_leg(dev)
        except Exception as e:      # an auxiliary leg must never cost the headline number
            leg_errors['parity'] = "%s: %s" % (type(e).__name__, e)
            print("bench.py: the parity leg failed: %r" % (e,), file=sys.stderr, flush=True)

    bf16_build = None
    if rank == 0 and world == 1 and not args.no_bf16_build and not args.no_profile and _C.storage_dtype() == torch.float16:
        try:
            tr.finish()
            torch.cuda.synchronize()
            bf16_build = bf16_build_leg(args)
        except Exception as e:      # an auxiliary leg must never cost the headline number
            leg_errors['bf16_build'] = "%s: %s" % (type(e).__name__, e)
            print("bench.py: the bf16_build leg failed: %r" % (e,), file=sys.stderr, flush=True)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # reported on rank 0 at N = 1 only
        try:
            cpu = cpu_baseline(args.arch)
        except Exception as e:      # an auxiliary leg must never cost the headline number
            leg_errors['cpu'] = "%s: %s" % (type(e).__name__, e)
            print("bench.py: the cpu leg failed: %r" % (e,), file=sys.stderr, flush=True)

    if rank == 0:
        ms_step = dt * 1e3 / args.steps
        value = world * B * args.steps / dt
        step_tflop = 3 * FWD_GFLOP_PER_IMG[args.arch] * B / 1e3
        out = {
            "metric": "images/sec (%s+%s train step, bs=%d/GPU, 112x112)" % (args.arch, {"dense": "CosFace", "pfc": "ArcFace+PartialFC",
                                                                                 "pfc-sharded": "CosFace+sharded PartialFC+BCE head"}[args.head], B),
            "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp16" if _C.storage_dtype() == torch.float16 else "bf16", "data": "synthetic" if backend == "nccl" and not share else "synthetic (REHEARSAL: ranks share one GPU / gloo — not a measurement)",
            "config": {"workload": "%s + %s, full train step "
                                   "(fwd+bwd+momentum-SGD), batch %d/GPU, 112x112 synthetic faces, random-init weights, "
                                   "%s activations/weights with fp32 accumulate + fp32 master weights, fp32 head"
                                   % (args.arch, ("CosFace(s=30,m=0.4) + dense %d-class cosine head" % NC) if args.head == "dense"
                                      else ("ArcFace(s=30,m=0.4) + PartialFC sample_rate 0.1 over %d classes" % NC) if args.head == "pfc"
                                      else ("CosFace(s=30,m=0.4) PartialFC sample_rate 0.1 over %d classes class-sharded over %d rank(s) + private "
                                            "BCE head over %d identities per client (BASELINE config 5)" % (NC, world, NC // world)), B,
                                      "fp16" if _C.storage_dtype() == torch.float16 else "bf16"),
                       "global_batch": world * B, "parallelism": "1 client per GPU (FedAvg), dp%d" % world,
                       "clients": world},
            "images_per_sec_per_gpu": round(value / world, 1),
            "host_enqueue_ms_per_step": {"empty_queue_median": round(sorted(host_ms)[len(host_ms) // 2], 3) if host_ms else None,
                                         "timed_loop": round(t_enqueue * 1e3 / args.steps, 3)},      # timed loop: includes queue back-pressure
            "step_mfma_frac": round(step_tflop / (ms_step * 1e-3) / BF16_DENSE_PEAK_TFLOPS, 4),
            "step_algorithmic_tflop": round(step_tflop, 3),
            "final_loss": round(final_loss, 4),
            "storage": "fp16" if _C.storage_dtype() == torch.float16 else "bf16",
            "library": os.path.basename(_C.LIB_PATH),
            "options_non_default": _C.options_non_default(),
            "parity": parity,
            "bf16_build": bf16_build,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "end_to_end": end_to_end,
            "concurrent_clients": concurrent,
            "fedavg": fedavg,
        }
        if fedavg_round is not None:
            # BASELINE's "FedAvg round time 1/2/4/8 clients" measured on ONE GPU (clients one after another, as the reference runs them)
            out["fedavg_round_ms"] = {k: v["round_ms"] for k, v in fedavg_round["sequential"].items()}
            out["fedavg_round"] = fedavg_round
            if cpu and isinstance(cpu.get("fedpavg_s"), dict):
                fedavg_round["cpu_oracle_fedpavg_s"] = cpu["fedpavg_s"]       # the aggregation alone on the host cores (cpu_baseline), beside it
        if rccl_ranks is not None:
            out["rccl_ranks"] = rccl_ranks
            out["collective_backend"] = "rccl" if backend == "nccl" else backend
        if leg_errors:
            out["leg_errors"] = leg_errors
        if use_dist:
            out["fedavg_round_ms"] = round(dt * 1e3, 3)
            out["fedavg_exchange_ms"] = round((dt - t_local) * 1e3, 3)
        # the line's LAST key: the numbers a reader of a truncated log tail needs (everything above has the detail)
        out["summary"] = {"ms_per_step": out["ms_per_step"], "images_per_sec": out["value"], "dtype": out["dtype"], "step_mfma_frac": out["step_mfma_frac"],
                          "roofline_frac": roofline and roofline.get("frac"), "roofline_kernel_us": roofline and roofline.get("avg_launch_us"),
                          "parity_vs_reference": parity and {k: parity[k] for k in ("embeddings_eval", "embeddings_train", "cosine_logits_train")},
                          "fedavg_round_ms": out.get("fedavg_round_ms"),
                          "fedavg_round_ms_parallel_clients_2": fedavg_round and {k: v["round_ms"] for k, v in fedavg_round["parallel_clients_2"].items()},
                          "fedavg_round_rccl_world1": fedavg_round and fedavg_round.get("rccl_world1") and
                          {k: fedavg_round["rccl_world1"].get(k) for k in ("round_ms", "exchange_ms", "local_steps", "error") if k in fedavg_round["rccl_world1"]},
                          "cpu_baseline_images_per_sec": cpu and cpu.get("value"), "cpu_cores": cpu and cpu.get("cores"),
                          "options_non_default": out["options_non_default"], "leg_errors": sorted(leg_errors) or None}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
