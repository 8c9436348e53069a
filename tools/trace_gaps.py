#!/usr/bin/env python3
"""Reads a rocprofv3 kernel-trace CSV of bench.py and reports, for the last full training steps: wall time per step, the union of
kernel-busy intervals (any stream), idle time, and the busiest kernels — i.e. how much of a step is dependency/launch gaps."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# step boundaries: the flat SGD kernel over the backbone runs once per step (largest sgd launch)
sgd = [e for e in ev if e[2].startswith("sgd_kernel")]
big = sorted(sgd, key=lambda e: e[1] - e[0], reverse=True)[: len(sgd) // 2]
marks = sorted(e[1] for e in big)
steps = list(zip(marks[-6:-1], marks[-5:]))
for a, b in steps[-3:]:
    ks = [e for e in ev if e[0] >= a and e[1] <= b + 1]
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in ks:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = sum(e - s for s, e, _ in ks)
    print("step wall %.2f ms | kernels %d | busy (union) %.2f ms | idle %.2f ms (%.1f%%) | sum of kernel durations %.2f ms (overlap %.2f ms)"
          % ((b - a) / 1e6, len(ks), busy / 1e6, (b - a - busy) / 1e6, 100.0 * (b - a - busy) / (b - a), tot / 1e6, (tot - busy) / 1e6))
agg = collections.defaultdict(lambda: [0, 0])
a, b = steps[-1]
for s, e, n in ev:
    if s >= a and e <= b + 1:
        agg[n.split("(")[0][:60]][0] += e - s; agg[n.split("(")[0][:60]][1] += 1
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:16]:
    print("  %-62s %4d launches %7.3f ms" % (n, c, t / 1e6))

# ---- largest idle gaps of the last step: which kernel ended before / started after
a, b = steps[-1]
ks = sorted([e for e in ev if e[0] >= a and e[1] <= b + 1])
gaps, cur_e, last = [], None, None
for s_, e_, n_ in ks:
    if cur_e is not None and s_ > cur_e:
        gaps.append((s_ - cur_e, last, n_.split("(")[0][:40], (cur_e - a) / 1e6))
    if cur_e is None or e_ > cur_e:
        cur_e, last = e_, n_.split("(")[0][:40]
gaps.sort(reverse=True)
print("idle gaps > 5 us: %d, total %.3f ms; gaps <= 5 us: %d, total %.3f ms" % (sum(1 for g in gaps if g[0] > 5000), sum(g[0] for g in gaps if g[0] > 5000) / 1e6,
                                                                              sum(1 for g in gaps if g[0] <= 5000), sum(g[0] for g in gaps if g[0] <= 5000) / 1e6))
for g in gaps[:12]:
    print("  %6.1f us at t=%6.2f ms  after %-40s before %s" % (g[0] / 1e3, g[3], g[1], g[2]))
