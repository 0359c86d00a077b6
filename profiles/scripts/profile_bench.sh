#!/bin/bash
# rocprofv3 passes over bench.py on the MI355X box: kernel stats, FETCH_SIZE, WRITE_SIZE (separate
# passes), plus the un-profiled bench line.  Usage: bash profiles/scripts/profile_bench.sh <tag>
set -o pipefail
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
echo bench done
rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-filters --no-raster > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
echo kt done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-filters --no-raster --arnoldi-steps 40 > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-filters --no-raster --arnoldi-steps 40 > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
echo write done
grep -l "Memory access fault" $O/*.log && exit 1
# keep the merged-back payload small: only the csv summaries
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*_agent_info.csv" -delete
ls -la $O/*/* | head
