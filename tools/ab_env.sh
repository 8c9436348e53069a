#!/bin/bash
# same-box A/B of environment settings: bash tools/ab_env.sh "" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" ...   (each twice, interleaved)
mkdir -p gpurun_out
for rep in 1 2; do
  for e in "$@"; do
    env $e python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-profile > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err || { tail -20 gpurun_out/ab_tmp.err; exit 1; }
    python - "$e" <<'P'
import json, sys
d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
print("[%s] %.3f ms/step  %.0f img/s" % (sys.argv[1] or "default", d["ms_per_step"], d["value"]))
P
  done
done
