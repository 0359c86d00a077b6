# Both latency chains of the overlap-save kernel shortened, separately and together (timing only: PROBE_AB
# replaces the (alpha, beta) table loads by constants -- wrong results).  Libraries built locally by
#   CM2_EXTRA_HIPCC_FLAGS="-DCM2_OS_PROBE_AB" / "-DCM2_OS_BOTH_HALVES" python -m cosmomap2_amd.build
# and copied to profiles/scripts/_variants/lib_<name>.so; two alternations.
for rep in 1 2; do
for name in base PROBE_AB BOTH_HALVES PROBE_AB_BOTH_HALVES; do
  CM2_LIB_PATH=$PWD/profiles/scripts/_variants/lib_$name.so python bench.py --no-cpu --no-filters --no-parity --no-pcg --no-raster --deflation 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'lib': '$name', 'rep': $rep, 'ms_per_step': round(d['ms_per_step'], 4), 'stages': {k[:5]: round(v['ms'], 4) for k, v in d['stages'].items()}}))"
done
done
