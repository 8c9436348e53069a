#!/bin/bash
# same-box A/B of the prefetch profiles of the sliced BatchNorm-backward apply pass
set -e -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "bn_" > gpurun_out/ab_bn_sliced_tests.log 2>&1 || { tail -30 gpurun_out/ab_bn_sliced_tests.log; exit 1; }
tail -1 gpurun_out/ab_bn_sliced_tests.log
for opt in "bn_sliced=0" "bn_sliced_pre=0" "bn_sliced_pre=1" "bn_sliced_pre=2" "bn_sliced_pre=3" "bn_sliced=0" "bn_sliced_pre=0" "bn_sliced_pre=2" "bn_sliced_pre=3"; do
  FEDFR_OPTIONS="$opt" python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-profile > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err || { tail -20 gpurun_out/ab_tmp.err; exit 1; }
  python - <<P
import json
d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
print("$opt", d["ms_per_step"], d["value"], d.get("final_loss"))
P
done
