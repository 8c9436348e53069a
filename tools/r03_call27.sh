#!/bin/bash
# round 3, call 27: the other bench configurations on the closing build (sphnet, iresnet50, PartialFC, config 5 at world 1, --gpus 1 self-launch path)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
for args in "--arch sphnet" "--arch iresnet50" "--head pfc --classes 85000" "--head pfc-sharded --classes 85000"; do
  timeout -k 10 400 python bench.py $args --steps 20 --warmup 5 --no-cpu-baseline --no-profile > gpurun_out/r03_c27_tmp.json 2> gpurun_out/r03_c27_tmp.err || { echo "FAILED: $args"; tail -20 gpurun_out/r03_c27_tmp.err; exit 1; }
  python - "$args" <<'P'
import json, sys
d=json.loads(open("gpurun_out/r03_c27_tmp.json").read().strip().splitlines()[-1])
print("[%s] %.3f ms/step  %.0f img/s  rccl_ranks %s" % (sys.argv[1], d["ms_per_step"], d["value"], d.get("rccl_ranks")))
P
done
