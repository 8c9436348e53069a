#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "libfedfr_hip.so|" "libfedfr_hip_ab8.so|wgrad_depth=4" "libfedfr_hip_ab8.so|wgrad_depth=6" "libfedfr_hip_ab8.so|wgrad_depth=8"; do
  lib=${cfg%%|*}; opts=${cfg##*|}
  FEDFR_OPTIONS="$opts" FEDFR_HIP_LIB_NAME=$lib python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-profile > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err || { tail -20 gpurun_out/ab_tmp.err; exit 1; }
  python - "$cfg" <<'P'
import json, sys
d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
print("[%s] %.3f ms/step  %.0f img/s" % (sys.argv[1], d["ms_per_step"], d["value"]))
P
done; done
