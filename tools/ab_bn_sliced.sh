#!/bin/bash
# same-box A/B of the channel-sliced BatchNorm passes (option bn_sliced): unit tests, launch-floor chains, bench with the option on / off
set -e -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "bn_" > gpurun_out/ab_bn_sliced_tests.log 2>&1 || { tail -30 gpurun_out/ab_bn_sliced_tests.log; exit 1; }
tail -2 gpurun_out/ab_bn_sliced_tests.log
python tools/launch_floor.py > gpurun_out/ab_bn_sliced_floor.json
cat gpurun_out/ab_bn_sliced_floor.json
for v in 1 0 1 0; do
  FEDFR_OPTIONS="bn_sliced=$v" python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-profile > gpurun_out/ab_bn_sliced_$v.json 2>gpurun_out/ab_bn_sliced_$v.err || { tail -20 gpurun_out/ab_bn_sliced_$v.err; exit 1; }
  python - <<P
import json
d=json.loads(open("gpurun_out/ab_bn_sliced_$v.json").read().strip().splitlines()[-1])
print("bn_sliced=$v", d["ms_per_step"], d["value"], d.get("final_loss"))
P
done
