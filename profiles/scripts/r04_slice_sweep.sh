# slice length of the fixed-order P^T lists (CM2_PT_SLICE; default chosen for ~0.92 x 512 groups per slice = 1856 at C4)
for rep in 1 2; do
for S in default 1536 1664 1792 1920 2048; do
  if [ $S = default ]; then unset CM2_PT_SLICE; else export CM2_PT_SLICE=$S; fi
  python bench.py --no-cpu --no-filters --no-parity --no-pcg --no-raster --deflation 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'slice': '$S', 'rep': $rep, 'ms_per_step': round(d['ms_per_step'], 4), 'stages': {k[:5]: round(v['ms'], 4) for k, v in d['stages'].items()}}))"
done
done
