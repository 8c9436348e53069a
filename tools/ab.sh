#!/bin/bash
# ONE same-box A/B driver (one gpurun call = one box; numbers from different boxes differ by up to 5 % and are never compared).
#   bash tools/ab.sh [--env] [--lib] [--kernels "<regex>"] "<variant A>" "<variant B>" ...
# A variant is a FEDFR_OPTIONS string ("" = defaults; "fwd_bnfuse=0,wgrad9p_bg=0"), with --env a list of environment assignments
# ("GPU_MAX_HW_QUEUES=8 FEDFR_DUAL_STREAM=0"), with --lib a library file name under fedfr_amd/ (libfedfr_hip_ab3.so, built HERE beforehand with
# tools/build_ablate.sh: built .so files travel with the snapshot).  Every variant runs REPS (2) times, interleaved.
#   --kernels "<regex>": instead of the step time, rocprofv3 per-kernel averages of the matching kernels (single-stream pass; DUAL=1: default step)
# environment: REPS, STEPS (30), WARMUP (10), BENCH_ARGS (extra bench.py arguments, e.g. "--arch sphnet")
set -o pipefail
MODE=opt; KRX=""
while [[ "$1" == --* ]]; do
  case "$1" in --env) MODE=env;; --lib) MODE=lib;; --kernels) KRX="$2"; shift;; *) echo "unknown flag $1"; exit 2;; esac
  shift
done
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
run_one() {     # $1 = variant; prints one line
  local v="$1"
  ( case $MODE in
      opt) export FEDFR_OPTIONS="$v";;
      env) for kv in $v; do export "$kv"; done;;
      lib) export FEDFR_HIP_LIB_NAME="$v";;
    esac
    if [ -n "$KRX" ]; then
      cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ab_ks
      [ "${DUAL:-0}" = "1" ] || export FEDFR_DUAL_STREAM=0
      rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_ks -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile $BENCH_ARGS > /tmp/ab_ks.json 2> /tmp/ab_ks.err || { tail -5 /tmp/ab_ks.err; exit 1; }
      python3 - "$(find /tmp/ab_ks -name '*kernel_stats.csv' | head -1)" "$KRX" "$v" <<'PY'
import csv, json, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("[%s] step %s ms; kernel sum per step %.3f ms" % (sys.argv[3], json.loads(open("/tmp/ab_ks.json").read().strip().splitlines()[-1])["ms_per_step"],
                                                        sum(float(r["TotalDurationNs"]) for r in rows) / 13e6))
for r in rows:
    if re.search(sys.argv[2], r["Name"]):
        print("   %-100s calls/step %6.1f avg %7.2f us  %7.3f ms/step" % (r["Name"][:100], int(r["Calls"]) / 13.0, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 13e6))
PY
    else
      cd $R && python bench.py --steps ${STEPS:-30} --warmup ${WARMUP:-10} --no-cpu-baseline --no-profile $BENCH_ARGS > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { tail -20 gpurun_out/ab_tmp.err; exit 1; }
      python3 - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
print("[%s]" % sys.argv[1], d["ms_per_step"], d["value"], d.get("final_loss"), d.get("options_non_default"), d.get("library"))
PY
    fi ) || exit 1
}
for rep in $(seq 1 ${REPS:-2}); do
  for v in "$@"; do run_one "$v" || exit 1; done
done
