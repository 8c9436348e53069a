#!/bin/bash
# round 3, call 20: BN-backward reduction in the two-tiles 28x28 dgrad (fuse_bnbwd28): kernel test, A/B, e2e tests
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "fused_bn_bwd_reduction or conv_dgrad or bn_sliced or sliced" > gpurun_out/r03_c20_tests.txt 2>&1 || { tail -30 gpurun_out/r03_c20_tests.txt; exit 1; }
tail -2 gpurun_out/r03_c20_tests.txt
bash tools/ab_opts.sh "" "fuse_bnbwd28=0" > gpurun_out/r03_c20.txt 2>&1 || { cat gpurun_out/r03_c20.txt; exit 1; }
cat gpurun_out/r03_c20.txt
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py tests/test_block_gpu.py -x -q > gpurun_out/r03_c20_e2e.txt 2>&1 || { tail -30 gpurun_out/r03_c20_e2e.txt; exit 1; }
tail -2 gpurun_out/r03_c20_e2e.txt
