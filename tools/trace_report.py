#!/usr/bin/env python3
"""Report on a one-step kernel trace cut by tools/trace_last_step.py: forward / backward wall time, per-kernel totals per queue in the
backward pass (main = the queue with most kernels), main-queue idle gaps and what surrounds the largest of them, tail of the step."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows: r['s'] = float(r['start_us']); r['e'] = float(r['end_us'])
qs = collections.Counter(r['queue'] for r in rows); mainq = max(qs, key=qs.get)
bw0 = [r for r in rows if r['name'].startswith('bn1d_bwd')] or [r for r in rows if 'sph_dfeats_kernel' in r['name']]     # first backward kernel (iresnet / sphnet)
tb = bw0[0]['s']
tf = min(r['s'] for r in rows)
te = max(r['e'] for r in rows)
print("fwd %.2f ms; bwd+sgd %.2f ms" % ((tb - tf) / 1e3, (te - tb) / 1e3))
bw = [r for r in rows if r['s'] >= tb]
agg = collections.defaultdict(lambda: [0, 0])
for r in bw:
    k = (r['queue'] == mainq and 'main' or 'aux', r['name'][:48]); agg[k][0] += r['e'] - r['s']; agg[k][1] += 1
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print("   %-60s %4d %7.3f ms  avg %.1f us" % (n, c, t / 1e3, t / c))
m = sorted([r for r in rows if r['queue'] == mainq], key=lambda r: r['s'])
gaps = [(b['s'] - a['e'], a['name'][:36], b['name'][:36], b['s']) for a, b in zip(m, m[1:]) if b['s'] >= tb]
print("main queue idle in bwd: %.2f ms, of it gaps > 5 us: %.2f ms (%d)" % (sum(g[0] for g in gaps) / 1e3, sum(g[0] for g in gaps if g[0] > 5) / 1e3, sum(1 for g in gaps if g[0] > 5)))
for g in sorted(gaps, reverse=True)[:8]: print("  %6.1f us at %.2f ms: after %-36s before %s" % (g[0], g[3] / 1e3, g[1], g[2]))
print("last 14 kernels:")
for r in sorted(rows, key=lambda r: r['s'])[-14:]: print("  %8.1f -> %8.1f  %s %s" % (r['s'], r['e'], r['queue'] == mainq and 'main' or 'aux ', r['name'][:60]))

# ---- round 3: where the MAIN queue's time goes, by phase and by network stage.  A kernel's stage = the map size of the nearest MFMA conv
# kernel on the main queue (forward: the last one before it; backward: the next one after it, whose input it produces), read off the kernel name.
def stage_of(name):
    if 'stem' in name: return 'stem'
    if 'c64p_kernel<112' in name or 'glds_kernel<112' in name: return 'L1.0 (112x112)'
    if 'c64p_kernel<56' in name or 'glds_kernel<56' in name: return 'L1 (56x56)'
    if 'glds_kernel<28' in name: return 'L2 (28x28)'
    if 'glds_kernel<14' in name: return 'L3 (14x14)'
    if 'gemm_nt_glds' in name: return 'L4 / stride-2'
    if 'gemm_nt_kernel' in name: return '1x1 / stride-2 dgrad / fc'
    return None
def by_stage(ks, forward):
    out = collections.OrderedDict()
    cur = 'stem' if forward else 'head'
    seq = ks if forward else ks[::-1]
    lab = []
    for r in seq:
        s_ = stage_of(r['name'])
        if s_: cur = s_
        lab.append(cur)
    if not forward: lab = lab[::-1]
    prev_end = None
    for r, l in zip(ks, lab):
        d = out.setdefault(l, [0.0, 0.0, 0.0, 0])     # conv/GEMM time, other kernel time, gap before, launches
        mf = stage_of(r['name']) is not None
        d[0 if mf else 1] += r['e'] - r['s']
        if prev_end is not None: d[2] += max(0.0, r['s'] - prev_end)
        d[3] += 1
        prev_end = max(prev_end or 0.0, r['e'])
    return out
for title, ks, fw in (("forward", [r for r in m if r['s'] < tb], True), ("backward", [r for r in m if r['s'] >= tb], False)):
    print("main queue, %s: %d launches, kernels %.2f ms, gaps %.2f ms" % (title, len(ks), sum(r['e'] - r['s'] for r in ks) / 1e3,
          sum(max(0.0, b['s'] - a['e']) for a, b in zip(ks, ks[1:])) / 1e3))
    for l, (a, b, g, c) in by_stage(ks, fw).items():
        print("   %-28s %4d launches  MFMA kernels %6.3f ms  other (BatchNorm ...) %6.3f ms  gaps %6.3f ms" % (l, c, a / 1e3, b / 1e3, g / 1e3))
    agg = collections.defaultdict(lambda: [0, 0])
    for r in ks:
        agg[r['name'][:56]][0] += r['e'] - r['s']; agg[r['name'][:56]][1] += 1
    for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:16]:
        print("      %-58s %4d %7.3f ms  avg %.1f us" % (n, c, t / 1e3, t / c))
