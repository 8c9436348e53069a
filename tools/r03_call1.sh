#!/bin/bash
# round-3 GPU call 1: new tests, MFMA shape probe, wgrad9 timing ablations, baseline bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "fedavg or dropout or freeze_bn or self_launch or parallel_clients or fedpavg" > gpurun_out/r03_c1_tests.txt 2>&1 || { tail -30 gpurun_out/r03_c1_tests.txt; exit 1; }
tail -3 gpurun_out/r03_c1_tests.txt
timeout -k 10 120 ./tools/probe/mfma_shape_probe > gpurun_out/r03_c1_mfma_probe.txt 2>&1 || { tail gpurun_out/r03_c1_mfma_probe.txt; exit 1; }
cat gpurun_out/r03_c1_mfma_probe.txt
for lib in libfedfr_hip.so libfedfr_hip_ab1.so libfedfr_hip_ab2.so libfedfr_hip_ab3.so libfedfr_hip_ab4.so libfedfr_hip_ab8.so libfedfr_hip_ab11.so; do
  echo "== $lib" >> gpurun_out/r03_c1_w9_ablate.txt
  FEDFR_HIP_LIB_NAME=$lib timeout -k 10 120 python tools/conv_bench.py 50 "s3_256x256@14" wgrad 2>/dev/null | grep -E "wgrad" >> gpurun_out/r03_c1_w9_ablate.txt || exit 1
done
cat gpurun_out/r03_c1_w9_ablate.txt
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/r03_c1_bench.json 2> gpurun_out/r03_c1_bench.err || { tail -20 gpurun_out/r03_c1_bench.err; exit 1; }
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03_c1_bench.json").read().strip().splitlines()[-1])
print("bench", d["ms_per_step"], d["value"], "fedavg", d.get("fedavg"), "dual", d["roofline"].get("dual_stream_avg_launch_us"), d.get("leg_errors"))
P
