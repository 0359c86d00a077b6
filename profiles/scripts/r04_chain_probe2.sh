for rep in 1 2 3 4 5; do
for name in base AB_GROW; do
  CM2_LIB_PATH=$PWD/profiles/scripts/_variants/lib_$name.so python bench.py --no-cpu --no-filters --no-parity --no-pcg --no-raster --deflation 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'lib': '$name', 'rep': $rep, 'ms_per_step': round(d['ms_per_step'], 4), 'stages': {k[:5]: round(v['ms'], 4) for k, v in d['stages'].items()}}))"
done
done
