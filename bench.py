#!/usr/bin/env python3
"""bench.py — headline benchmark of the FedFR per-client training hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

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

FWD_GFLOP_PER_IMG = {"iresnet100": 24.18, "iresnet50": 12.62}       # SURVEY.md §8(d), measured on the reference
BF16_DENSE_PEAK_TFLOPS = 2500.0                                       # MI355X_MICROARCH.md: ~2.5 PF dense bf16
# HBM bytes per launch from the PMC counters (separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH x2 gfx950 correction, collected with
# tools/pmc_traffic.sh on the 256->256 @14x14 layer = 58 of iresnet100's 103 convs; committed under profiles/, file named per entry).
# bench.py cannot run rocprofv3 on itself, so `roofline.traffic` quotes that measurement for the kernel it names.
PMC_TRAFFIC_MB = {"gemm_tn_glds_kernel<128,128>": {"fetch": 44.9, "write": 16.5, "algorithmic": 25.7 + 16.5, "file": "profiles/r01_pmc_hbm_traffic_256x256_14_v8.txt"},
                  "conv3x3_glds_kernel<14,14>": {"fetch": 22.5, "write": 13.2, "algorithmic": 14.0 + 12.9, "file": "profiles/r01_pmc_hbm_traffic_256x256_14_v13.txt"},
                  "wgrad9_kernel<32x64x9>": {"fetch": 25.8, "write": 18.9, "algorithmic": 25.7 + 18.9, "file": "profiles/r01_pmc_hbm_traffic_256x256_14_v13.txt"}}
SLOT_NAMES = ["gemm_nt_kernel<128,128>", "gemm_nt_kernel<128,64>", "gemm_nt_kernel<64,128>", "gemm_nt_kernel<64,64>",
              "gemm_tn_kernel<128,128>", "gemm_tn_kernel<128,64>", "gemm_tn_kernel<64,128>", "gemm_tn_kernel<64,64>",
              "conv3x3_halo2_kernel<128,14>", "conv3x3_halo2_kernel<128,28>", "conv3x3_halo2_kernel<64,*>", "conv3x3_halo_kernel<*>",
              "conv3x3_glds_kernel<14,14>", "conv3x3_glds_kernel<28,7>", "gemm_tn_glds_kernel<128,128>", "conv3x3_glds_kernel<56,4>",
              "wgrad9_kernel<32x64x9>"]


def cpu_baseline(arch, batch=16, steps=2):
    """Reference CPU path = the oracle restatement (pinned to the imported reference by tests/golden), one warm-up +
    `steps` timed full train steps on all host threads."""
    from oracle import ref_cpu as R
    layers = R.IRESNET_LAYERS[arch]
    torch.manual_seed(100)
    sd = R.closed_form_state_dict(layers)
    fc = torch.randn(1000, 512) * 0.01
    cores = torch.get_num_threads()
    g = torch.Generator().manual_seed(100)
    batches = [(torch.rand(batch, 3, 112, 112, generator=g) * 2 - 1, torch.randint(0, 1000, (batch,), generator=g))
               for _ in range(steps + 1)]
    R.client_train(sd, fc, batches[:1], layers, "CosFace", 30.0, 0.4, 1e-3)
    t0 = time.perf_counter()
    R.client_train(sd, fc, batches[1:], layers, "CosFace", 30.0, 0.4, 1e-3)
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%s+CosFace fp32, batch %d, %d full train steps (fwd+bwd+SGD) after 1 warm-up, torch CPU" % (arch, batch, steps)}


def main():
    # stdout carries exactly ONE line, the JSON result: everything else that writes to file descriptor 1 during the run (the RCCL
    # version banner librccl prints at communicator creation, warnings of native libraries) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--arch", default="iresnet100", choices=["iresnet100", "iresnet50"])
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--classes", type=int, default=1000)
    ap.add_argument("--head", default="dense", choices=["dense", "pfc"],
                    help="dense: CosFace + dense cosine head (headline); pfc: ArcFace + PartialFC sample_rate 0.1 (BASELINE config 3; use --classes 85000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("FEDFR_FORCE_DIST") == "1"      # FORCE_DIST: exercise the RCCL path on 1 GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from fedfr_amd import _C, backbones, client, server
    torch.manual_seed(100 + rank)                                  # reference seed 100 (train.py:35)
    B, NC = args.batch, args.classes
    model = getattr(backbones, args.arch)(False, dropout=0, fp16=True).to(dev)     # reference init (random weights)
    if args.head == "pfc":
        from fedfr_amd import losses
        from fedfr_amd.partial_fc import PartialFC
        fc = PartialFC(rank=0, local_rank=local_rank, world_size=1, batch_size=B, resume=False,
                       margin_softmax=losses.ArcFace(s=30, m=0.4), num_classes=NC, sample_rate=0.1, embedding_size=512, prefix="/tmp")
        tr = client.FusedTrainer(model, fc, "ArcFace", 30.0, 0.4, lr=1e-3, momentum=0.9, weight_decay=5e-4)
    else:
        fc = (torch.randn(NC, 512) * 0.01).to(dev)                                # client.py:66
        tr = client.FusedTrainer(model, fc, "CosFace", 30.0, 0.4, lr=1e-3, momentum=0.9, weight_decay=5e-4)
    g = torch.Generator().manual_seed(100 + rank)
    nbuf = 4
    imgs = [(torch.rand(B, 3, 112, 112, generator=g) * 2 - 1).to(dev) for _ in range(nbuf)]   # already resident in HBM
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
    t_local = None
    if use_dist:
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

    # ---- roofline leg: HIP-event timing of every MFMA GEMM launch, in a separate short pass (same workload) ----
    roofline = None
    if rank == 0 and not args.no_profile:
        psteps = 3
        # kernels are timed one at a time: the weight-gradient GEMMs normally share the GPU with the dgrad/BN chain on a
        # second stream, which stretches every kernel's wall time; for a per-kernel roofline the pass runs single-stream
        # (rocprof summary of the matching command: FEDFR_DUAL_STREAM=0 python bench.py ..., profiles/*_single_stream*)
        saved_aux = tr.aux_stream
        torch.cuda.synchronize()
        tr.aux_stream = None
        _C.call("fedfr_profile_enable", 1)
        for i in range(psteps):
            tr.step(imgs[i % nbuf], labs[i % nbuf])
        torch.cuda.synchronize()
        tr.aux_stream = saved_aux
        rows = []
        for slot in range(len(SLOT_NAMES)):
            ms, n, fl = C.c_double(), C.c_longlong(), C.c_double()
            _C.call("fedfr_profile_read", slot, C.byref(ms), C.byref(n), C.byref(fl))
            if n.value:
                rows.append((ms.value, n.value, fl.value, slot))
        _C.call("fedfr_profile_enable", 0)
        rows.sort(reverse=True)
        if rows:
            ms, n, fl, slot = rows[0]
            ach = fl / (ms * 1e-3) / 1e12
            roofline = {"bound": "mfma", "kernel": SLOT_NAMES[slot], "achieved": round(ach, 2), "peak": BF16_DENSE_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / BF16_DENSE_PEAK_TFLOPS, 4),
                        "traffic": (round((PMC_TRAFFIC_MB[SLOT_NAMES[slot]]["fetch"] + PMC_TRAFFIC_MB[SLOT_NAMES[slot]]["write"]) * 1e6)
                                    if SLOT_NAMES[slot] in PMC_TRAFFIC_MB else None),
                        "traffic_note": ("HBM bytes per launch on the 256->256 @14x14 layer (fetch %(fetch).1f MB + write %(write).1f MB, algorithmic %(algorithmic).1f MB): "
                                         "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, %(file)s" % PMC_TRAFFIC_MB[SLOT_NAMES[slot]]
                                         if SLOT_NAMES[slot] in PMC_TRAFFIC_MB else None),
                        "timing": "HIP events around each launch, single-stream pass of %d steps" % psteps,
                        "launches_per_step": n // psteps, "avg_launch_us": round(ms * 1e3 / n, 2),
                        "gflop_per_launch": round(fl / n / 1e9, 3),
                        "all_gemm_kernels": [{"kernel": SLOT_NAMES[s], "ms_per_step": round(m_ / psteps, 3), "launches_per_step": k // psteps,
                                              "tflops": round(f / (m_ * 1e-3) / 1e12, 1)} for m_, k, f, s in rows]}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # reported on rank 0 at N = 1 only
        cpu = cpu_baseline(args.arch)

    if rank == 0:
        ms_step = dt * 1e3 / args.steps
        value = world * B * args.steps / dt
        step_tflop = 3 * FWD_GFLOP_PER_IMG[args.arch] * B / 1e3
        out = {
            "metric": "images/sec (iresnet100+CosFace train step, bs=128/GPU, 112x112)" if args.arch == "iresnet100"
                      else "images/sec (%s+CosFace train step, bs=%d/GPU, 112x112)" % (args.arch, B),
            "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "%s + %s, full train step "
                                   "(fwd+bwd+momentum-SGD), batch %d/GPU, 112x112 synthetic faces, random-init weights, "
                                   "bf16 activations/weights with fp32 accumulate + fp32 master weights, fp32 head"
                                   % (args.arch, ("CosFace(s=30,m=0.4) + dense %d-class cosine head" % NC) if args.head == "dense"
                                      else ("ArcFace(s=30,m=0.4) + PartialFC sample_rate 0.1 over %d classes" % NC), B),
                       "global_batch": world * B, "parallelism": "1 client per GPU (FedAvg), dp%d" % world,
                       "clients": world},
            "images_per_sec_per_gpu": round(value / world, 1),
            "step_mfma_frac": round(step_tflop / (ms_step * 1e-3) / BF16_DENSE_PEAK_TFLOPS, 4),
            "step_algorithmic_tflop": round(step_tflop, 3),
            "final_loss": round(final_loss, 4),
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if use_dist:
            out["fedavg_round_ms"] = round(dt * 1e3, 3)
            out["fedavg_exchange_ms"] = round((dt - t_local) * 1e3, 3)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
