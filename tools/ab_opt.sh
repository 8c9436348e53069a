#!/bin/bash
# same-box A/B of FEDFR_OPTIONS settings: bash tools/ab_opt.sh "opt_a" "opt_b" ...   (each run twice, interleaved)
set -e -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
  for opt in "$@"; do
    FEDFR_OPTIONS="$opt" python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-profile > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err || { tail -20 gpurun_out/ab_tmp.err; exit 1; }
    python - <<P
import json
d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
print("[$opt]", d["ms_per_step"], d["value"], d.get("final_loss"))
P
  done
done
