for sl in 8192 12207 15258 24000 32000; do echo "# CM2_TILE_SLICE=$sl"; CM2_TILE_SLICE=$sl timeout -k 10 300 python bench.py --config c5 --no-filters --no-cpu --no-raster --no-pcg 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read()); print(round(r['ms_per_step'],4), {k[:8]: v['ms'] for k,v in r['stages'].items()})"; done
