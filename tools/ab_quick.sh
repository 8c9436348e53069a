#!/bin/bash
# kernel tests of the conv / BatchNorm families + same-box A/B of option sets + the network-level suites, one call:
#   bash tools/ab_quick.sh <tag> "<pytest -k expression for tests/test_kernels_gpu.py>" "<opts A>" "<opts B>" ...
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$PWD}"
TAG=$1; KEXPR=$2; shift 2
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -k "$KEXPR" > gpurun_out/${TAG}_tests.txt 2>&1 || { tail -30 gpurun_out/${TAG}_tests.txt; exit 1; }
tail -2 gpurun_out/${TAG}_tests.txt
bash tools/ab_opts.sh "$@" > gpurun_out/${TAG}.txt 2>&1 || { cat gpurun_out/${TAG}.txt; exit 1; }
cat gpurun_out/${TAG}.txt
timeout -k 10 900 python -m pytest tests/test_e2e_gpu.py tests/test_block_gpu.py -x -q > gpurun_out/${TAG}_e2e.txt 2>&1 || { tail -40 gpurun_out/${TAG}_e2e.txt; exit 1; }
tail -2 gpurun_out/${TAG}_e2e.txt
