#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -k "fused_bn_backward or train_step_grads or freeze_bn or block" > gpurun_out/r03_c5_tests.txt 2>&1 || { tail -30 gpurun_out/r03_c5_tests.txt; exit 1; }
tail -3 gpurun_out/r03_c5_tests.txt
timeout -k 10 600 python -m pytest tests/test_block_gpu.py -x -q > gpurun_out/r03_c5_tests2.txt 2>&1 || { tail -30 gpurun_out/r03_c5_tests2.txt; exit 1; }
tail -2 gpurun_out/r03_c5_tests2.txt
bash tools/ab_opts.sh "" "bn_fuse_bwd=0" > gpurun_out/r03_c5_ab.txt 2>&1 || { cat gpurun_out/r03_c5_ab.txt; exit 1; }
cat gpurun_out/r03_c5_ab.txt
FEDFR_DUAL_STREAM=0 bash tools/ab_opts.sh "" "bn_fuse_bwd=0" > gpurun_out/r03_c5_ab_single.txt 2>&1; cat gpurun_out/r03_c5_ab_single.txt
