#!/bin/bash
# one-step kernel trace of the default (dual-stream) bench step: rocprofv3 --kernel-trace, cut + report.  usage: [BENCH_ARGS="--arch sphnet"] bash tools/trace_step.sh <tag> [FEDFR_OPTIONS]
TAG=${1:-r03}; OPTS=${2:-}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_trace
FEDFR_OPTIONS="$OPTS" rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_trace -- python3 $R/bench.py $BENCH_ARGS --steps 6 --warmup 3 --no-cpu-baseline --no-profile > $R/gpurun_out/${TAG}_trace_bench.json 2> $R/gpurun_out/${TAG}_trace.err || { tail -20 $R/gpurun_out/${TAG}_trace.err; exit 1; }
python3 $R/tools/trace_last_step.py $(find /tmp/prof_trace -name "*kernel_trace.csv" | head -1) $R/gpurun_out/${TAG}_last_step.csv > $R/gpurun_out/${TAG}_trace_report.txt
python3 $R/tools/trace_report.py $R/gpurun_out/${TAG}_last_step.csv 20 >> $R/gpurun_out/${TAG}_trace_report.txt
cat $R/gpurun_out/${TAG}_trace_report.txt
