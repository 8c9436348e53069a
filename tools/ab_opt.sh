#!/bin/bash
# same-box A/B of FEDFR_OPTIONS settings: bash tools/ab_opt.sh "opt_a" "opt_b" ...   (each run REPS times, interleaved; "" = defaults)
# environment: REPS (2), STEPS (30), WARMUP (10), BENCH_ARGS (extra bench.py arguments, e.g. "--arch sphnet"), FEDFR_HIP_LIB_NAME (another library)
set -e -o pipefail
mkdir -p gpurun_out
for rep in $(seq 1 ${REPS:-2}); do
  for opt in "$@"; do
    FEDFR_OPTIONS="$opt" python bench.py --steps ${STEPS:-30} --warmup ${WARMUP:-10} --no-cpu-baseline --no-profile $BENCH_ARGS > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err || { tail -20 gpurun_out/ab_tmp.err; exit 1; }
    python - <<P
import json
d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
print("[$opt]", d["ms_per_step"], d["value"], d.get("final_loss"), d.get("options_non_default"))
P
  done
done
