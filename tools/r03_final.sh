#!/bin/bash
# round-3 closing artefacts: rocprofv3 kernel stats (dual / single stream), PMC traffic of the dominant layer, trace report, default bench line
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
bash tools/profile_round.sh r03 > $R/gpurun_out/r03_profile_round.log 2>&1 || { tail -20 $R/gpurun_out/r03_profile_round.log; exit 1; }
cd $R
bash tools/trace_step.sh r03_final > /dev/null 2>&1 || exit 1
timeout -k 10 900 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err || { tail -20 gpurun_out/r03_bench_default.err; exit 1; }
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03_bench_default.json").read().strip().splitlines()[-1])
print("bench", d["ms_per_step"], d["value"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("dual_stream_avg_launch_us"), "second", d["roofline"]["second"]["kernel"], d["roofline"]["second"]["frac"])
print("hbm", d["roofline"]["hbm"]); print("cpu", d["cpu_baseline"]); print("fedavg", d["fedavg"]); print("conc", d["concurrent_clients"]); print("e2e", d["end_to_end"]); print(d.get("leg_errors"))
P
