#!/bin/bash
# round 3, call 24: stem BN-backward reduction in the first block's apply pass + live rows of the persistent conv's statistics
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_e2e_gpu.py tests/test_block_gpu.py -x -q > gpurun_out/r03_c24_e2e.txt 2>&1 || { tail -40 gpurun_out/r03_c24_e2e.txt; exit 1; }
tail -2 gpurun_out/r03_c24_e2e.txt
bash tools/ab_opts.sh "" "stem_bnred=0" > gpurun_out/r03_c24.txt 2>&1 || { cat gpurun_out/r03_c24.txt; exit 1; }
cat gpurun_out/r03_c24.txt
