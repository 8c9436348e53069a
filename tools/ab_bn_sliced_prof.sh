#!/bin/bash
# kernel stats (rocprofv3) of the bench with the channel-sliced BatchNorm passes on / off, dual- and single-stream, on one box
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/ab_bn_sliced_prof; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  for ds in 1 0; do
    export FEDFR_OPTIONS="bn_sliced=$v" FEDFR_DUAL_STREAM=$ds
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${v}_${ds} -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/bench_sliced${v}_dual${ds}.json 2> $OUT/err_${v}_${ds}.txt || exit 1
    cp $(find /tmp/prof_${v}_${ds} -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_sliced${v}_dual${ds}.csv
  done
done
ls -la $OUT
