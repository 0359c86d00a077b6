#!/bin/bash
# k_os_real at C5 whole: duration inside bench.py's sequence by kernel trace (no counters), beside the probe's
# event timings (alone / in sequence / behind an idle gap)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c5w_trace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/profiles/scripts/r05_sustained_probe.py c5w 3 > $O/probe_c5w.json 2> $O/probe_c5w.err || { tail -5 $O/probe_c5w.err; exit 1; }
cat $O/probe_c5w.json
rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 $R/bench.py --config c5 --scaling strong --gpus 1 --no-cpu --no-filters --no-parity --no-pcg --no-raster --deflation 0 --steps 5 --warmup 1 > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*_agent_info.csv" -delete
head -8 $O/kt/*/*_kernel_stats.csv | cut -c1-200
