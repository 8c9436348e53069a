#!/usr/bin/env python3
"""Cuts the last full training step out of a rocprofv3 kernel-trace CSV of bench.py: one line per kernel (start us, end us, queue id,
short name, grid) relative to the step start.  usage: trace_last_step.py <kernel_trace.csv> <out.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Grid_Size", r.get("Grid_Size_X", "0")),
             r.get("Workgroup_Size", r.get("Workgroup_Size_X", "0"))) for r in rows)
stems = sorted(e[0] for e in ev if e[2].startswith("stem_fwd"))        # one per step (the fused optimizer made sgd launches a poor marker)
if len(stems) < 2:                                                      # sphnet plan: its first kernel pads the input image
    stems = sorted(e[0] for e in ev if "pad_input_nhwc_kernel" in e[2])
a, b = stems[-2], stems[-1]
with open(sys.argv[2], "w") as f:
    f.write("start_us,end_us,queue,name,grid,wg\n")
    for s, e, n, q, g, w in ev:
        if s >= a and s < b:
            f.write("%.3f,%.3f,%s,%s,%s,%s\n" % ((s - a) / 1e3, (e - a) / 1e3, q, n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace(",", ";")[:80], g, w))
print("step wall %.3f ms" % ((b - a) / 1e6))
