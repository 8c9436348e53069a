#!/bin/bash
# round 4, one gpurun call: SQ counters on the kernels the step runs + a same-box baseline step
set -o pipefail
mkdir -p gpurun_out/pmc_sq
bash tools/pmc_sq.sh conv14_fwd     "s3_256x256@14" fwd    > /dev/null && echo done conv14_fwd
bash tools/pmc_sq.sh conv14_fdgrad  "s3_256x256@14" fdgrad > /dev/null && echo done conv14_fdgrad
bash tools/pmc_sq.sh conv28_fwd     "s2_128x128@28" fwd    > /dev/null && echo done conv28_fwd
bash tools/pmc_sq.sh conv28_fdgrad  "s2_128x128@28" fdgrad > /dev/null && echo done conv28_fdgrad
bash tools/pmc_sq.sh wpair14        "s3_256x256@14" wpair  > /dev/null && echo done wpair14
bash tools/pmc_sq.sh wpair28        "s2_128x128@28" wpair  > /dev/null && echo done wpair28


