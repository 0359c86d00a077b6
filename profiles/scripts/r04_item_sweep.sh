# work-item size of k_P_tiles (CM2_TILE_SLICE; default nt / 8192 = 12207 samples at C4) and tile width (CM2_TILE_PIXELS)
for rep in 1 2; do
for S in default 6104 8138 16276 24414 48828; do
  if [ $S = default ]; then unset CM2_TILE_SLICE; else export CM2_TILE_SLICE=$S; fi
  python bench.py --no-cpu --no-filters --no-parity --no-pcg --no-raster --deflation 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'item': '$S', 'rep': $rep, 'ms_per_step': round(d['ms_per_step'], 4), 'stages': {k[:5]: round(v['ms'], 4) for k, v in d['stages'].items()}}))"
done
done
unset CM2_TILE_SLICE
for rep in 1 2; do
for TP in default 1024 1280 2048; do
  if [ $TP = default ]; then unset CM2_TILE_PIXELS; else export CM2_TILE_PIXELS=$TP; fi
  python bench.py --no-cpu --no-filters --no-parity --no-pcg --no-raster --deflation 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'tile_pixels': '$TP', 'rep': $rep, 'ms_per_step': round(d['ms_per_step'], 4), 'stages': {k[:5]: round(v['ms'], 4) for k, v in d['stages'].items()}}))"
done
done
