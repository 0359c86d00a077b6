# environment-selected variants of the plan against the default, alternated on one box (timed loop of bench.py)
for rep in 1 2 3 4; do
for v in default span_auto lists_inv; do
  unset CM2_TILE_SPAN CM2_OS_LISTS
  [ $v = span_auto ] && export CM2_TILE_SPAN=auto
  [ $v = lists_inv ] && export CM2_OS_LISTS=inv
  python bench.py --no-cpu --no-filters --no-parity --no-pcg --no-raster --deflation 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'plan': '$v', 'rep': $rep, 'ms_per_step': round(d['ms_per_step'], 4), 'stages': {k[:5]: round(v['ms'], 4) for k, v in d['stages'].items()}}))"
done
done
