#!/bin/bash
# Same-box alternation of variants over the timed loop of bench.py.
# usage: bash profiles/scripts/r05_ab.sh OUTFILE REPS "BENCH ARGS" name[:VAR=value[,VAR=value...]] ...
#   a variant named lib_X loads profiles/scripts/_variants/lib_X.so (build_variant.py) unless it sets CM2_LIB_PATH
out=$1; reps=$2; bargs=$3; shift 3
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $(dirname $R/gpurun_out/$out)
: > $R/gpurun_out/$out
for rep in $(seq 1 $reps); do
for v in "$@"; do
  name=${v%%:*}
  assign=""
  [ "$v" != "$name" ] && assign=$(echo "${v#*:}" | tr ',' ' ')
  case $name in lib_*) assign="$assign CM2_LIB_PATH=$R/profiles/scripts/_variants/$name.so";; esac
  env CM2_AB=1 $assign python3 $R/bench.py $bargs --no-cpu --no-filters --no-parity --no-pcg --no-raster --deflation 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'variant': '$name', 'rep': $rep, 'ms_per_step': round(d['ms_per_step'], 4), 'stages': {k[:5]: round(v['ms'], 4) for k, v in d['stages'].items()}, 'os': d['config']['tile_plan']['overlap_save'].get('os_lists')}))" | tee -a $R/gpurun_out/$out
done
done
python3 - <<PY
import json, collections
rows = [json.loads(l) for l in open("$R/gpurun_out/$out")]
by = collections.defaultdict(list)
for r in rows: by[r["variant"]].append(r)
for k, v in by.items():
    ms = sorted(r["ms_per_step"] for r in v)
    st = {s: round(sum(r["stages"][s] for r in v) / len(v), 4) for s in v[0]["stages"]}
    print("%-24s step median %.4f  min %.4f  max %.4f   stage means %s" % (k, ms[len(ms) // 2], ms[0], ms[-1], st))
PY
