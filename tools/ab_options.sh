for opt in "" "fuse_bnapply=1" "fuse_bnbwd=1" "tn_pair=1" "wgrad_depth=3" "fuse_bnapply=1,fuse_bnbwd=1"; do
  FEDFR_OPTIONS="$opt" python bench.py --no-cpu-baseline --no-profile --steps 30 --warmup 8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('OPT [%s]' % '$opt', d['ms_per_step'], d['value'])"
done
