#!/bin/bash
# same-box A/B of library option sets (FEDFR_OPTIONS strings), each run twice, interleaved:
#   bash tools/ab_opts.sh "" "ew_reduce_nt=1" "ew_reduce_nt=2,ew_bwd_apply_blocks=1024" ...
# FEDFR_AB_LIB=libfedfr_hip_abN.so selects another library build for every run; FEDFR_AB_STEPS (default 30)
mkdir -p gpurun_out
for rep in 1 2; do
  for opts in "$@"; do
    FEDFR_OPTIONS="$opts" FEDFR_HIP_LIB_NAME=${FEDFR_AB_LIB:-libfedfr_hip.so} python bench.py --steps ${FEDFR_AB_STEPS:-30} --warmup 10 --no-cpu-baseline --no-profile > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err || { tail -20 gpurun_out/ab_tmp.err; exit 1; }
    python - "$opts" <<'P'
import json, sys
d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
print("[%s] %.3f ms/step  %.0f img/s  loss %.4f" % (sys.argv[1] or "default", d["ms_per_step"], d["value"], d["final_loss"]))
P
  done
done
