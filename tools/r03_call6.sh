#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
FEDFR_OPTIONS="fork_mode=1,event_nofence=1" timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -k "train_step_grads or client_r18 or sgd" > gpurun_out/r03_c6_tests.txt 2>&1 || { tail -30 gpurun_out/r03_c6_tests.txt; exit 1; }
tail -2 gpurun_out/r03_c6_tests.txt
bash tools/ab_opts.sh "" "fork_mode=1" "event_nofence=1" "fork_mode=1,event_nofence=1" > gpurun_out/r03_c6_ab.txt 2>&1 || { cat gpurun_out/r03_c6_ab.txt; exit 1; }
cat gpurun_out/r03_c6_ab.txt
bash tools/trace_step.sh r03_fork1 "fork_mode=1,event_nofence=1" | grep -E "step wall|fwd |main queue idle|gaps|us at" | head -14
