#!/bin/bash
# A/B per-kernel comparison on ONE box: rocprofv3 kernel stats of bench.py, variant "new" = defaults, variant "ab" = libfedfr_hip_ab.so
# or, when AB_OPTIONS is set (e.g. AB_OPTIONS=tn_pair=0), the default library with FEDFR_OPTIONS=$AB_OPTIONS
# usage (on the GPU box): bash tools/ab_profile.sh [single|dual]   -> gpurun_out/ab_prof_{new,ab}.csv
set -e
mode=${1:-single}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ "$mode" = single ] && export FEDFR_DUAL_STREAM=0
for v in new ab; do
  unset FEDFR_HIP_LIB_NAME FEDFR_OPTIONS
  if [ $v = ab ]; then if [ -n "$AB_OPTIONS" ]; then export FEDFR_OPTIONS=$AB_OPTIONS; else export FEDFR_HIP_LIB_NAME=libfedfr_hip_ab.so; fi; fi
  rm -rf /tmp/prof_$v
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile > /tmp/prof_$v.log 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/ab_prof_${v}_$mode.csv
done
python3 - <<'P'
import csv, os
R = os.environ["GRAFT_REPO_ROOT"]; mode = os.environ.get("FEDFR_DUAL_STREAM") == "0" and "single" or "dual"
def load(v):
    return {r["Name"]: r for r in csv.DictReader(open("%s/gpurun_out/ab_prof_%s_%s.csv" % (R, v, mode)))}
a, b = load("new"), load("ab")
ta = sum(float(r["TotalDurationNs"]) for r in a.values()) / 13e6; tb = sum(float(r["TotalDurationNs"]) for r in b.values()) / 13e6
print("kernel-sum ms/step: new %.3f  ab %.3f" % (ta, tb))
names = sorted(set(a) | set(b), key=lambda n: -max(float(a.get(n, {"TotalDurationNs": 0})["TotalDurationNs"]), float(b.get(n, {"TotalDurationNs": 0})["TotalDurationNs"])))
for n in names[:28]:
    ra, rb = a.get(n), b.get(n)
    f = lambda r: (float(r["TotalDurationNs"]) / 13e6, float(r["AverageNs"]) / 1e3, r["Calls"]) if r else (0, 0, 0)
    print("%-70s new %6.3f ms (%6.1f us x%5s) | ab %6.3f ms (%6.1f us x%5s)" % ((n[:70],) + f(ra) + f(rb)))
P
