#!/usr/bin/env python3
"""Report on a one-step kernel trace cut by tools/trace_last_step.py: forward / backward wall time, per-kernel totals per queue in the
backward pass (main = the queue with most kernels), main-queue idle gaps and what surrounds the largest of them, tail of the step."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows: r['s'] = float(r['start_us']); r['e'] = float(r['end_us'])
qs = collections.Counter(r['queue'] for r in rows); mainq = max(qs, key=qs.get)
tb = [r for r in rows if r['name'].startswith('bn1d_bwd')][0]['s']
tf = [r for r in rows if r['name'].startswith('stem_fwd')][0]['s']
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
