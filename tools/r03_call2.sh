#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 ./tools/probe/hbm_stream_probe 56 > gpurun_out/r03_c2_hbm_probe_56.txt 2>&1 || { tail gpurun_out/r03_c2_hbm_probe_56.txt; exit 1; }
timeout -k 10 300 ./tools/probe/hbm_stream_probe 112 > gpurun_out/r03_c2_hbm_probe_112.txt 2>&1 || { tail gpurun_out/r03_c2_hbm_probe_112.txt; exit 1; }
cat gpurun_out/r03_c2_hbm_probe_56.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "fedpavg or fedavg or client or server or round or checkpoint" > gpurun_out/r03_c2_tests.txt 2>&1 || { tail -30 gpurun_out/r03_c2_tests.txt; exit 1; }
tail -3 gpurun_out/r03_c2_tests.txt
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r03_c2_bench.json 2> gpurun_out/r03_c2_bench.err || { tail -20 gpurun_out/r03_c2_bench.err; exit 1; }
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03_c2_bench.json").read().strip().splitlines()[-1])
print("bench", d["ms_per_step"], d["value"], "fedavg", {k:(v["ms"],v["achieved_gbps"]) for k,v in d["fedavg"]["clients"].items()}, d.get("leg_errors"))
P
