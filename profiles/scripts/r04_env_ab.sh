# environment-selected variants of the plan against the default, alternated on one box (timed loop of bench.py)
# usage: bash r04_env_ab.sh "name:VAR=value ..."   (default is always included)
VARIANTS=${1:-"span_auto:CM2_TILE_SPAN=auto lists_inv:CM2_OS_LISTS=inv"}
for rep in 1 2 3 4 5; do
for v in default $VARIANTS; do
  name=${v%%:*}
  if [ "$v" = default ]; then assign=""; else assign=${v#*:}; fi
  env $assign python bench.py --no-cpu --no-filters --no-parity --no-pcg --no-raster --deflation 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'plan': '$name', 'rep': $rep, 'ms_per_step': round(d['ms_per_step'], 4), 'stages': {k[:5]: round(v['ms'], 4) for k, v in d['stages'].items()}}))"
done
done
