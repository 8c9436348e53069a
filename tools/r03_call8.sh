#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -k "sphnet" > gpurun_out/r03_c8_tests.txt 2>&1 || { tail -40 gpurun_out/r03_c8_tests.txt; exit 1; }
tail -3 gpurun_out/r03_c8_tests.txt
timeout -k 10 600 python bench.py --arch sphnet --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r03_c8_bench_sph.json 2> gpurun_out/r03_c8_bench_sph.err || { tail -20 gpurun_out/r03_c8_bench_sph.err; exit 1; }
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03_c8_bench_sph.json").read().strip().splitlines()[-1])
print("sphnet bench", d["ms_per_step"], d["value"], d["step_mfma_frac"], d.get("leg_errors"))
for k in d["roofline"]["all_gemm_kernels"]: print("  ", k)
print(d["roofline"].get("hbm_kernels"))
P
