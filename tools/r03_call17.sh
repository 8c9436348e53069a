#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_e2e_gpu.py -x -q -s -k "fp16_storage" > gpurun_out/r03_c17.txt 2>&1 || { tail -30 gpurun_out/r03_c17.txt; exit 1; }
grep -E "fp16 build|passed|failed" gpurun_out/r03_c17.txt
for lib in libfedfr_hip.so libfedfr_hip_fp16.so libfedfr_hip.so libfedfr_hip_fp16.so; do
  FEDFR_HIP_LIB_NAME=$lib python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$lib]', d['ms_per_step'], d['value'], d['dtype'], d['final_loss'])"
done
