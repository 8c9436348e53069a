#!/bin/bash
# One gpurun call that produces the artefacts of a round on ONE box (so that they agree with each other): rocprofv3 kernel stats of the bench step
# (dual-stream = default, and the single-stream pass the roofline leg times), HBM traffic of the dominant layer and of the whole step (PMC), the SQ
# counters of the dominant kernels, a one-step kernel trace with its report, the full default bench line, the bf16-storage line and the other
# BASELINE configurations.  usage: bash tools/round_artifacts.sh <tag>   -> gpurun_out/<tag>_*   (copy what is to be judged into profiles/)
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
bash tools/profile_round.sh $TAG > /dev/null 2>&1; echo "profile_round done"
cp gpurun_out/profile_round/${TAG}_bench_r100_b128_kernel_stats_dual_stream.csv gpurun_out/${TAG}_kernel_stats_dual_stream.csv
cp gpurun_out/profile_round/${TAG}_bench_r100_b128_kernel_stats_single_stream.csv gpurun_out/${TAG}_kernel_stats_single_stream.csv
cp gpurun_out/profile_round/${TAG}_pmc_hbm_traffic_256x256_14.txt gpurun_out/${TAG}_pmc_hbm_traffic_256x256_14.txt
bash tools/trace_step.sh $TAG > /dev/null 2>&1; echo "trace done"
bash tools/pmc_step_traffic.sh $TAG > gpurun_out/${TAG}_pmc_hbm_traffic_whole_step.txt 2>&1; echo "step traffic done"
bash tools/pmc_sq_all.sh > /dev/null 2>&1; echo "sq counters done"
bash tools/pmc_sq_step.sh $TAG > /dev/null 2>&1; echo "whole-step sq counters done"
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_r100_b128.json 2> gpurun_out/${TAG}_bench.err; echo "bench done"
python3 bench.py --lib bf16 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_r100_b128_bf16_storage.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --arch iresnet50 --steps 20 --warmup 5 --no-cpu-baseline --no-profile > gpurun_out/${TAG}_bench_r50_b128_config2.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --head pfc --classes 85000 --steps 20 --warmup 5 --no-cpu-baseline --no-profile > gpurun_out/${TAG}_bench_r100_b128_pfc85k_config3.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --arch sphnet --steps 20 --warmup 5 --no-cpu-baseline --no-profile > gpurun_out/${TAG}_bench_sphnet64_b128.json 2>> gpurun_out/${TAG}_bench.err
for f in r100_b128 r100_b128_bf16_storage r50_b128_config2 r100_b128_pfc85k_config3 sphnet64_b128; do python3 -c "
import json,sys
j=json.loads(open('gpurun_out/${TAG}_bench_$f.json').read().strip().splitlines()[-1]); print('$f', j['ms_per_step'], j['value'], j.get('storage'), j.get('parity',{}) and j['parity'].get('embeddings_train'))"; done
