#!/bin/bash
# wall-clock split of the one-off setup and the kernels behind it
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_setup
rm -rf $O; mkdir -p $O
python3 $R/profiles/scripts/setup_probe.py > $O/setup_probe.jsonl 2> $O/setup_probe.err
cat $O/setup_probe.jsonl
rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 $R/profiles/scripts/setup_probe.py > $O/kt.log 2>&1
f=$(find $O/kt -name "*_kernel_stats.csv" | head -1)
cp $f $O/setup_kernel_stats.csv
head -30 $O/setup_kernel_stats.csv | cut -c1-160
find $O/kt -name "*_kernel_trace.csv" -delete
