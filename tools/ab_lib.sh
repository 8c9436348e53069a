#!/bin/bash
# same-box A/B of alternative library builds (fedfr_amd/libfedfr_hip_ab*.so, built HERE beforehand; FEDFR_HIP_LIB_NAME selects one):
#   bash tools/ab_lib.sh libfedfr_hip.so libfedfr_hip_ab1.so ...      (each run twice, interleaved)
mkdir -p gpurun_out
for rep in 1 2; do
  for lib in "$@"; do
    FEDFR_HIP_LIB_NAME=$lib python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-profile > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err || { tail -20 gpurun_out/ab_tmp.err; exit 1; }
    python - <<P
import json
d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
print("[$lib]", d["ms_per_step"], d["value"])
P
  done
done
