#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for lib in libfedfr_hip.so libfedfr_hip_ab1.so; do
  echo "== $lib"
  FEDFR_HIP_LIB_NAME=$lib timeout -k 10 120 python tools/conv_bench.py 50 "s3_256x256@14" wgrad 2>/dev/null | grep -E "wgrad" | head -1
  FEDFR_HIP_LIB_NAME=$lib timeout -k 10 120 python tools/conv_bench.py 50 "s2_128x128@28" wgrad 2>/dev/null | grep -E "wgrad" | head -1
done
bash tools/ab_lib.sh libfedfr_hip.so libfedfr_hip_ab1.so
