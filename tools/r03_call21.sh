#!/bin/bash
# round 3, call 21: late join, fc wgrad on aux, head off-path kernels on aux: e2e tests, A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_e2e_gpu.py tests/test_block_gpu.py tests/test_multirank_gpu.py -x -q > gpurun_out/r03_c21_e2e.txt 2>&1 || { tail -30 gpurun_out/r03_c21_e2e.txt; exit 1; }
tail -2 gpurun_out/r03_c21_e2e.txt
bash tools/ab_opts.sh "" "late_join=0" "fc_wgrad_aux=0" > gpurun_out/r03_c21.txt 2>&1 || { cat gpurun_out/r03_c21.txt; exit 1; }
cat gpurun_out/r03_c21.txt
bash tools/ab_env.sh "" "FEDFR_HEAD_OFF_PATH=0" > gpurun_out/r03_c21b.txt 2>&1 || { cat gpurun_out/r03_c21b.txt; exit 1; }
cat gpurun_out/r03_c21b.txt
