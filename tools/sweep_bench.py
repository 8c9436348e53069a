#!/usr/bin/env python3
"""Measurement for the SURVEY §8f rows N1 / N2 / N3 on one MI355X:
  N1  eval-mode embedding sweep (server.py:242-263 Generate_pretrain_feats): images/s of iresnet100 forward at the reference's
      public batch size 512 (config.py:27), normalised embeddings kept on the GPU;
  N2  hard-negative mining (client.py:208-226): N_local x 420 671 x 512 similarity GEMM with the threshold / column-OR epilogue
      (420 671 = size of the reference's public set), fp32 MFMA, similarity matrix never materialised.
Prints one JSON object; the CPU column times the oracle restatement of the same step on the host (bounded sample)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from fedfr_amd import backbones, ops, client
dev = torch.device("cuda:0")
out = {}
# ---- N1
B = 512
m = backbones.iresnet100(False, dropout=0, fp16=True).to(dev).eval()
x = [(torch.rand(B, 3, 112, 112) * 2 - 1).to(dev) for _ in range(2)]
with torch.no_grad():
    for i in range(3):
        f, _ = ops.normalize_rows(m(x[i % 2]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for i in range(n):
        f, _ = ops.normalize_rows(m(x[i % 2]))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
out["N1_embedding_sweep"] = {"images_per_sec": round(B * n / dt, 1), "batch": B, "ms_per_batch": round(dt / n * 1e3, 2),
                            "fwd_tflops": round(24.18e9 * B * n / dt / 1e12, 1), "frac_of_bf16_peak": round(24.18e9 * B * n / dt / 2.5e15, 4),
                            "public_set_420671_images_s": round(420671 / (B * n / dt), 2)}
# ---- N2
NL, NP, D, thr = 2048, 420671, 512, 0.4
g = torch.Generator().manual_seed(1)
a = F.normalize(torch.randn(NL, D, generator=g)).to(dev)
b = F.normalize(torch.randn(NP, D, generator=g)).to(dev)
b[::7] = a[torch.arange(0, (NP + 6) // 7) % NL] * 0.9 + b[::7] * 0.1          # plant hard negatives
b = F.normalize(b)
for _ in range(2):
    flags = ops.similarity_column_flags(a, b, thr)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    flags = ops.similarity_column_flags(a, b, thr)
idx = torch.nonzero(flags).flatten()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
fl = 2.0 * NL * NP * D
out["N2_hard_negative_mining"] = {"ms": round(dt * 1e3, 2), "tflops_fp32": round(fl / dt / 1e12, 1), "frac_of_fp32_mfma_peak_157": round(fl / dt / 157.3e12, 3),
                                 "n_local": NL, "n_public": NP, "selected": int(idx.numel())}
# CPU reference of the same mining step (the reference's own way: full matmul in 100 row chunks + where + union), bounded sample
torch.set_num_threads(min(64, torch.get_num_threads()))
ac, bc = a[:256].cpu(), b[:100000].cpu()
t0 = time.perf_counter()
sim = ac @ bc.t()
u = torch.unique(torch.where(sim > thr)[1])
dtc = time.perf_counter() - t0
out["N2_hard_negative_mining"]["cpu_reference"] = {"sample": "256 x 100000 x 512 fp32 matmul + where + unique on %d threads" % torch.get_num_threads(),
                                                  "ms": round(dtc * 1e3, 1), "gflops": round(2.0 * 256 * 100000 * 512 / dtc / 1e9, 1)}
assert torch.equal(torch.nonzero(ops.similarity_column_flags(a[:256], b[:100000], thr)).flatten().cpu(), u)
# ---- N3: pairwise ROC histogram (roc_cuda.py), 100 000 embeddings, 2 500 target rows
from fedfr_amd import eval_roc
N, T = 100000, 2500
lab = torch.randint(0, 4000, (N,), generator=g)
f3 = F.normalize(torch.randn(N, D, generator=g)).to(dev)
lab = lab.to(dev)
h = eval_roc.roc_histogram(f3, lab, T)
torch.cuda.synchronize()
t0 = time.perf_counter()
h = eval_roc.roc_histogram(f3, lab, T)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
pairs = int(h.sum())
assert pairs == T * (T - 1) // 2 + T * (N - T)
out["N3_roc_histogram"] = {"N": N, "T": T, "pairs": pairs, "ms": round(dt * 1e3, 2), "gpairs_per_s": round(pairs / dt / 1e9, 2),
                           "tflops_fp64": round(2.0 * pairs * D / dt / 1e12, 1), "frac_of_fp64_mfma_peak_78.6": round(2.0 * pairs * D / dt / 78.6e12, 3)}
from oracle import ref_cpu as R
t0 = time.perf_counter()
R.roc_histogram(f3[:20000].cpu().numpy(), lab[:20000].cpu().numpy(), 100)
dtc = time.perf_counter() - t0
out["N3_roc_histogram"]["cpu_port"] = {"sample": "oracle (numpy float64), 100 target rows x 20000", "gpairs_per_s": round((100 * 99 // 2 + 100 * 19900) / dtc / 1e9, 4)}
print(json.dumps(out))
