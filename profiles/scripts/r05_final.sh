#!/bin/bash
# The round's evidence, in two GPU calls (a call is limited to ~20 minutes and must keep writing output):
#   part a: profile passes of the default bench (kernel stats, FETCH_SIZE, WRITE_SIZE) + the other configurations
#   part b: C5 whole with the oracle's PCG, the oracle's own Arnoldi at C4
# usage: bash profiles/scripts/r05_final.sh a|b [tag]      (outputs under gpurun_out/prof_<tag>/)
set -o pipefail
part=${1:-a}
tag=${2:-r05}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
if [ "$part" = a ]; then
  bash $R/profiles/scripts/profile_bench.sh $tag || exit 1
  cd /tmp && export TMPDIR=/tmp
  for c in c2 c3 c5; do
    python3 $R/bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err || { tail -3 $O/bench_$c.err; exit 1; }
    echo "$c done"
  done
else
  cd /tmp && export TMPDIR=/tmp
  python3 $R/bench.py --config c5 --scaling strong --gpus 1 --no-raster --no-filters --parity-host-seconds 700 > $O/bench_c5_whole.json 2> $O/bench_c5_whole.err || { tail -3 $O/bench_c5_whole.err; exit 1; }
  echo "c5 whole done"
  python3 $R/bench.py --no-raster --no-filters --no-traffic --parity-arnoldi --parity-host-seconds 600 > $O/bench_c4_parity_arnoldi.json 2> $O/bench_c4_parity_arnoldi.err || { tail -3 $O/bench_c4_parity_arnoldi.err; exit 1; }
  echo "c4 parity-arnoldi done"
fi
