#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export FEDFR_HIP_LIB_NAME=libfedfr_hip_fp16.so
timeout -k 10 900 python -m pytest tests/test_e2e_gpu.py -x -q -s -k "backbone_forward_vs_reference or train_step_grads_vs_reference or fused_client_loop or sphnet_vs_reference or freeze_bn_vs" > gpurun_out/r03_c16_fp16.txt 2>&1
echo "rc=$?"
grep -E "MEASURED|passed|failed|Error|assert" gpurun_out/r03_c16_fp16.txt | head -40
