#!/bin/bash
# HIP runtime calls longer than 5 ms during two consecutive setups (which call stalls, and where)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_setup_slow
rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --hip-runtime-trace -d $O/kt --output-format csv -- python3 $R/profiles/scripts/setup_calls.py > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
grep '"rep"' $O/kt.log | cut -c1-400
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_setup_slow"
api = sorted(csv.DictReader(open(glob.glob(O + "/kt/**/*_hip_api_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
ker = sorted(csv.DictReader(open(glob.glob(O + "/kt/**/*_kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(api[0]["Start_Timestamp"])
ev = []
for r in api:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d > 5: ev.append((int(r["Start_Timestamp"]), "API  %-28s %8.2f ms" % (r["Function"], d)))
for r in ker:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d > 3: ev.append((int(r["Start_Timestamp"]), "KERN %-60s %8.2f ms" % (r["Kernel_Name"][:60], d)))
for ts, s in sorted(ev):
    print("%9.2f  %s" % ((ts - t0) / 1e6, s))
PY
find $O/kt -name "*_trace.csv" -delete
