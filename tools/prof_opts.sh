#!/bin/bash
# rocprofv3 kernel stats of the bench under given FEDFR_OPTIONS values (dual-stream): bash tools/prof_opts.sh tag1 "opts1" tag2 "opts2" ...
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_opts; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
while [ $# -ge 2 ]; do
  tag=$1; export FEDFR_OPTIONS="$2"; shift 2
  rm -rf /tmp/prof_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/bench_$tag.json 2> $OUT/err_$tag.txt || exit 1
  cp $(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$tag.csv
done
ls $OUT
