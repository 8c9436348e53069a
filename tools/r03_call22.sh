#!/bin/bash
# round 3, call 22: split-K head GEMMs + fused softmax: kernel tests, head/e2e tests, A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "sgemm or softmax or normalize" > gpurun_out/r03_c22_tests.txt 2>&1 || { tail -30 gpurun_out/r03_c22_tests.txt; exit 1; }
tail -2 gpurun_out/r03_c22_tests.txt
bash tools/ab_env.sh "" "FEDFR_HEAD_SPLITK=0" > gpurun_out/r03_c22.txt 2>&1 || { cat gpurun_out/r03_c22.txt; exit 1; }
cat gpurun_out/r03_c22.txt
timeout -k 10 900 python -m pytest tests/test_e2e_gpu.py tests/test_multirank_gpu.py -x -q > gpurun_out/r03_c22_e2e.txt 2>&1 || { tail -30 gpurun_out/r03_c22_e2e.txt; exit 1; }
tail -2 gpurun_out/r03_c22_e2e.txt
