#!/bin/bash
# same-box A/B of option nt_glds on the layer shapes the register-staged NT kernel serves (tools/conv_bench.py) and on the train step
mkdir -p gpurun_out
for v in 0 1 2; do
  echo "== nt_glds=$v"
  for f in s2 @7 ds_; do python tools/conv_bench.py 30 $f fwd,dgrad nt_glds=$v | grep -E "s2 |@7|ds_"; done
done
bash tools/ab_opt_env.sh "FEDFR_OPTIONS=nt_glds=0" "FEDFR_OPTIONS=nt_glds=1" "FEDFR_OPTIONS=nt_glds=2"
