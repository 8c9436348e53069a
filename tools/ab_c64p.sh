python -m pytest tests/test_kernels_gpu.py -q -x -k "conv" > gpurun_out/r2_c64p_tests.log 2>&1; tail -4 gpurun_out/r2_c64p_tests.log
for opt in "conv_c64p=0" "conv_c64p=1"; do
  python tools/conv_bench.py 20 "s1_64x64@" fwd,dgrad $opt 2>/dev/null | grep -E "s1_64x64@112 |s1_64x64@56 "
  FEDFR_OPTIONS="$opt" python bench.py --no-cpu-baseline --no-profile --steps 30 --warmup 8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('OPT [%s]' % '$opt', d['ms_per_step'], d['value'])"
done
