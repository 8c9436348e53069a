#!/bin/bash
# round 3, call 23: forward moment pass (fwd_xmom): e2e tests, A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_e2e_gpu.py -x -q -k "moment_pass" > gpurun_out/r03_c23_t.txt 2>&1 || { tail -40 gpurun_out/r03_c23_t.txt; exit 1; }
tail -2 gpurun_out/r03_c23_t.txt
bash tools/ab_opts.sh "" "fwd_xmom=0" > gpurun_out/r03_c23.txt 2>&1 || { cat gpurun_out/r03_c23.txt; exit 1; }
cat gpurun_out/r03_c23.txt
timeout -k 10 900 python -m pytest tests/test_e2e_gpu.py tests/test_block_gpu.py tests/test_multirank_gpu.py -x -q > gpurun_out/r03_c23_e2e.txt 2>&1 || { tail -40 gpurun_out/r03_c23_e2e.txt; exit 1; }
tail -2 gpurun_out/r03_c23_e2e.txt
