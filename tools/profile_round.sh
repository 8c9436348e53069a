#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel stats of the bench command (dual-stream = the default execution, and the single-stream
# pass the roofline leg times), plus the HBM-traffic PMC passes of the dominant layer.  Outputs under gpurun_out/profile_round/.
# usage: bash tools/profile_round.sh <tag>          (copy the summaries you want judged into profiles/ afterwards)
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/profile_round; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dual -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/${TAG}_bench_dual.json 2> $OUT/dual.err
cp $(find /tmp/prof_dual -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_r100_b128_kernel_stats_dual_stream.csv
FEDFR_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_single -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/${TAG}_bench_single.json 2> $OUT/single.err
cp $(find /tmp/prof_single -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_r100_b128_kernel_stats_single_stream.csv
cd $R && bash tools/pmc_traffic.sh "s3_256x256@14" fwd,dgrad,wgrad,wpair > $OUT/${TAG}_pmc_hbm_traffic_256x256_14.txt 2> $OUT/pmc.err
ls -la $OUT
