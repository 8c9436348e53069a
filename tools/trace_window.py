#!/usr/bin/env python3
"""Kernel-by-kernel listing of a window of a one-step trace cut by tools/trace_last_step.py: every launch of both queues between two times
(ms from the step start), in start order, with the gap to the previous launch of the same queue.
usage: trace_window.py <last_step.csv> <from_ms> <to_ms>"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
t0, t1 = float(sys.argv[2]) * 1e3, float(sys.argv[3]) * 1e3
qs = collections.Counter(r["queue"] for r in rows)
mainq = max(qs, key=qs.get)
last = {}
for r in sorted(rows, key=lambda r: float(r["start_us"])):
    s, e = float(r["start_us"]), float(r["end_us"])
    q = "main" if r["queue"] == mainq else "aux "
    gap = s - last[q] if q in last else 0.0
    last[q] = e
    if t0 <= s <= t1:
        print("%9.1f -> %9.1f  %6.1f us  gap %6.1f  %s  %s" % (s, e, e - s, gap, q, r["name"][:70]))
