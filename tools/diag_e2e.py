#!/usr/bin/env python3
"""Diagnostic (GPU box): how far is the HIP path from (a) the fp32 reference goldens and (b) the bf16-storage
emulating oracle?  Prints relative L2 errors; used to set the tolerances asserted in tests/test_e2e_gpu.py."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from oracle import ref_cpu as R, bf16_emul as E
from fedfr_amd import backbones, client, losses, ops

DEV = torch.device("cuda:0")
def rel(a, b):
    a = a.detach().double().cpu(); b = (b if isinstance(b, torch.Tensor) else torch.from_numpy(np.asarray(b))).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))

for arch, batch, fname in (("iresnet50", 8, "r50_b8"), ("iresnet100", 6, "r100_b6")):
    g = load_golden(fname); layers = R.IRESNET_LAYERS[arch]; C = int(g["num_classes"])
    sd = R.closed_form_state_dict(layers)
    m = getattr(backbones, arch)().to(DEV); m.load_state_dict(sd)
    x = R.closed_form_images(batch); lab = R.closed_form_labels(batch, C)
    m.eval()
    with torch.no_grad():
        fe = m(x.to(DEV))
        fe_em = E.iresnet_forward({k: v.clone() for k, v in sd.items()}, x, layers, training=False)
    print("%s eval : hip-vs-fp32 %.4f  hip-vs-emul %.4f  emul-vs-fp32 %.4f" % (arch, rel(fe, g["feat_eval"]), rel(fe, fe_em), rel(fe_em, g["feat_eval"])))
    # train step
    m.train()
    fcm = client.FC_module(512, C, "/tmp").to(DEV); fcm.fc.data = R.head_fc(C).to(DEV)
    cosine = client.Sequential_model(m, fcm)(x.to(DEV))
    logits = losses.CosFace(s=30, m=0.4)(cosine, lab.to(DEV)); loss = ops.cross_entropy(logits, lab.to(DEV)); loss.backward()
    sd2 = {k: v.clone() for k, v in sd.items()}
    t0 = time.time()
    f_em, c_em, l_em, g_em, fcg_em = E.train_step_grads(sd2, R.head_fc(C), x, lab, layers)
    print("   (emul step %.1fs)" % (time.time() - t0))
    print("%s train: cosine hip-vs-fp32 %.4f hip-vs-emul %.4f emul-vs-fp32 %.4f | loss hip %.5f emul %.5f fp32 %.5f" % (
        arch, rel(cosine, g["cosine"]), rel(cosine, c_em), rel(c_em, g["cosine"]), float(loss), l_em, float(g["loss"])))
    names = [str(n) for n in g["grad_names"]]; params = dict(m.named_parameters())
    hn = np.array([float(params[k].grad.norm()) for k in names]); en = np.array([float(g_em[k].norm()) for k in names]); rn = g["grad_norms"]
    dirs = np.array([rel(params[k].grad, g_em[k]) for k in names])
    big = rn > 1e-6 * rn.max()
    def st(a, b): e = np.abs(a[big] - b[big]) / b[big]; return "med %.4f max %.4f" % (np.median(e), e.max())
    print("   grad norms: hip-vs-fp32 %s | hip-vs-emul %s | emul-vs-fp32 %s" % (st(hn, rn), st(hn, en), st(en, rn)))
    worst = np.argsort(-dirs * big)[:6]
    print("   grad direction hip-vs-emul: med %.4f max %.4f ; worst: %s" % (np.median(dirs[big]), dirs[big].max(), [(names[i], round(float(dirs[i]), 3)) for i in worst]))
    print("   head fc grad hip-vs-emul %.4f" % rel(fcm.fc.grad, fcg_em))
    out = m.state_dict()
    for k in ("bn1", "layer2.0.downsample.1", "layer4.2.bn3", "features"):
        print("   %s running_var hip-vs-fp32 %.4f hip-vs-emul %.4f" % (k, rel(out[k + ".running_var"], g["rv_" + k]), rel(out[k + ".running_var"], sd2[k + ".running_var"])))
    del m; torch.cuda.empty_cache()

g = load_golden("client_r18"); B, C, steps, lr = int(g["B"]), int(g["C"]), int(g["steps"]), float(g["lr"])
layers = R.IRESNET_LAYERS["iresnet18"]; sd = R.closed_form_state_dict(layers, tag=2.0)
m = backbones.iresnet18().to(DEV); m.load_state_dict(sd); fc = R.head_fc(C).to(DEV)
tr = client.FusedTrainer(m, fc, "CosFace", 30.0, 0.4, lr=lr)
ls = [float(tr.step(R.closed_form_images(B, tag=float(s)).to(DEV), R.closed_form_labels(B, C, tag=s).to(DEV))) for s in range(steps)]
print("client r18 losses hip", ls, "fp32", list(g["losses"]))
out = m.state_dict()
for k in ("conv1.weight", "bn1.running_var", "layer2.0.downsample.0.weight", "fc.bias", "features.bias"):
    print("   sd %s hip-vs-fp32 %.4f" % (k, rel(out[k], g["sd_" + k])))
print("   head fc hip-vs-fp32 %.4f" % rel(fc, g["head_fc"]))
