import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from fedfr_amd import _C
from oracle import ref_cpu as R
from test_e2e_gpu import make_model, DEV
arch, batch = sys.argv[1], int(sys.argv[2])
outs = []
for opt in (0, 1):
    with _C.option_scope("fwd_xmom", opt):
        m, sd, _ = make_model(arch, tag=2.0)
        m.train()
        x = R.closed_form_images(batch).to(DEV)
        f = m(x)
        (f * R.closed_form((batch, 512), 0.37, 0.9, 1.0).to(DEV)).sum().backward()
        outs.append((f.detach().clone(), {k: p_.grad.clone() for k, p_ in m.named_parameters() if p_.grad is not None},
                     {k: v.clone() for k, v in m.state_dict().items() if "running" in k}))
(f0, g0, s0), (f1, g1, s1) = outs
print("feats", float((f0 - f1).abs().max() / f0.abs().max()))
rows = []
for k in g0:
    a, b = g0[k].double().flatten(), g1[k].double().flatten()
    rows.append((float((a - b).norm() / (a.norm() + 1e-12)), k, float(a.norm()), float(b.norm())))
for r in sorted(rows, reverse=True)[:25]: print("%.4f %-40s %.4e %.4e" % r)
rows = []
for k in s0:
    rows.append((float((s0[k] - s1[k]).abs().max() / (1 + s0[k].abs().max())), k))
for r in sorted(rows, reverse=True)[:8]: print("%.2e %s" % r)
