#!/bin/bash
# Uneven hit map / hot pixel legs of bench.py, alternating the fused form of the fixed-order P^T (one launch) with
# the separate kernels of round 4 (CM2_PT_FUSE=0): step over the uniform step measured the same way in the same
# process, and the stage times.  usage: bash profiles/scripts/r05_uneven_ab.sh OUTFILE REPS
out=${1:-r05_uneven_ab.jsonl}; reps=${2:-3}
R=${GRAFT_REPO_ROOT:-$PWD}
: > $R/gpurun_out/$out
for rep in $(seq 1 $reps); do
for v in fused separate:CM2_PT_FUSE=0; do
  name=${v%%:*}; assign=""
  [ "$v" != "$name" ] && assign=${v#*:}
  env CM2_AB=1 $assign python3 $R/bench.py --no-cpu --no-filters --no-parity --no-pcg --deflation 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
u = d['uneven_hit_map']; h = u.get('hot_pixel', {})
print(json.dumps({'variant': '$name', 'rep': $rep, 'ms_per_step': round(d['ms_per_step'], 4),
                  'uniform_same_method': u.get('uniform_ms_same_method'),
                  'uneven_ms': u.get('ms_per_step'), 'uneven_over_uniform': u.get('over_uniform'), 'uneven_stages': u.get('stages_ms_in_sequence'),
                  'hot_ms': h.get('ms_per_step'), 'hot_over_uniform': h.get('over_uniform'), 'hot_PT_ms': h.get('PT_ms'),
                  'error': u.get('error')}))" | tee -a $R/gpurun_out/$out
done
done
