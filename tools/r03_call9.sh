#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -k "sphnet or eval or forward_vs_reference" > gpurun_out/r03_c9_tests.txt 2>&1 || { tail -40 gpurun_out/r03_c9_tests.txt; exit 1; }
tail -3 gpurun_out/r03_c9_tests.txt
for o in "sph_fuse_act=1" "sph_fuse_act=0" "sph_fuse_act=1" "sph_fuse_act=0"; do
FEDFR_OPTIONS=$o timeout -k 10 600 python bench.py --arch sphnet --no-cpu-baseline --no-profile --steps 30 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[sphnet $o]', d['ms_per_step'], d['value'])"
done
bash tools/ab_opts.sh "" ""
