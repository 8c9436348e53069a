#!/usr/bin/env python3
"""Step time of the reference's MAIN training loop, Client.train_with_public_data (client.py:287-508), at its own sizes: combined batch 256
(config.com_batch_size), a cosine head over [local | public] identities (100 + 6 000), the personalised BCE branch over the 100 local identities
(weight 10) and the model-contrastive term against two frozen eval-mode backbones (weight config.mu) — through client.FusedHeadTrainer exactly as
Client.train_with_public_data drives it.  usage: python tools/public_data_bench.py [arch] [steps] [variant: full|bce|seq] [lr]
Prints one JSON line (ms per step, images/s) — a diagnostic, not the headline bench."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fedfr_amd import _C, backbones, client, losses, ops
from fedfr_amd.config import config as cfg

arch = sys.argv[1] if len(sys.argv) > 1 else "iresnet100"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
variant = sys.argv[3] if len(sys.argv) > 3 else "full"
lr = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-3       # (random-init weights and two fixed random batches: the reference's 0.05 diverges)
dev = torch.device("cuda:0")
B, NL, NP = cfg.com_batch_size, 100, 6000
torch.manual_seed(100)
bb = getattr(backbones, arch)(False, dropout=0, fp16=True).to(dev)
fcm = client.FC_module(512, NL, "/tmp").to(dev)
fcm.update_with_pretrain((torch.randn(NP, 512) * 0.01).to(dev))
use_bce, use_con = variant in ("full", "bce"), variant == "full"
head_params = list(fcm.parameters())
if use_bce:
    bce_module = client.BCE_module(512, NL, cfg.converter_layer).to(dev)
    bce_loss = losses.BCE_loss()
    head_params += list(bce_module.parameters())
if use_con:
    # the frozen global / last-round models run in eval mode: their BatchNorm running statistics must be real (a random-init net normalised with
    # mean 0 / var 1 overflows after a few blocks) — settle them with train-mode forward passes of the same random weights first
    bb.train()
    xcal = (torch.rand(B, 3, 112, 112) * 2 - 1).to(dev)
    with torch.no_grad():
        for _ in range(40):
            bb(xcal)
    g_model = getattr(backbones, arch)(False, dropout=0, fp16=True).to(dev)
    l_model = getattr(backbones, arch)(False, dropout=0, fp16=True).to(dev)
    g_model.load_state_dict(bb.state_dict()); l_model.load_state_dict(bb.state_dict())
    g_model.eval(); l_model.eval()
margin = losses.CosFace(s=30, m=0.4)
tr = client.FusedHeadTrainer(bb, head_params, lr=lr, momentum=cfg.momentum, weight_decay=cfg.weight_decay)
state = {}


def head_loss(feats, labels):
    cos_loss = ops.cross_entropy(margin(fcm(feats), labels), labels)
    loss = cos_loss
    if use_bce:
        z, gt = bce_module(feats, labels)
        loss = loss + 10 * bce_loss(z, gt)
    if use_con:
        loss = loss + cfg.mu * ops.contrastive_loss(feats, state["g"], state["l"], 0.5)
    return loss, cos_loss


imgs = [(torch.rand(B, 3, 112, 112) * 2 - 1).to(dev) for _ in range(2)]
labs = [torch.randint(0, NL + NP, (B,)).to(dev) for _ in range(2)]


def one(i):
    x, y = imgs[i % 2], labs[i % 2]
    if use_con:
        with torch.no_grad():
            state["g"], state["l"] = g_model(x), l_model(x)
    return tr.step(x, y, head_loss)


for i in range(3):
    one(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    out = one(i)
tr.finish()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"what": "train_with_public_data step (%s)" % variant, "arch": arch, "batch": B, "classes": NL + NP, "ms_per_step": round(dt * 1e3, 3),
                  "images_per_sec": round(B / dt, 1), "loss": float(out[0]), "storage": str(_C.storage_dtype()).split(".")[-1]}))
