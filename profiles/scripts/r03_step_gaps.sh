#!/bin/bash
# gaps between the three kernels of a step in the timed loop of bench.py (kernel trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_gaps
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/kt --output-format csv -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu --no-filters --no-raster --no-pcg > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
python3 - <<'PY'
import csv, glob, os, json
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_gaps"
rows = sorted(csv.DictReader(open(glob.glob(O + "/kt/**/*_kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# steps = consecutive (k_P_tiles, k_os_real, k_Pt_tiles_fixed)
steps = []
for i in range(len(seq) - 2):
    if "k_P_tiles" in seq[i][0] and "k_os_real" in seq[i + 1][0] and "k_Pt_tiles_fixed" in seq[i + 2][0]:
        steps.append(i)
# the longest run of back-to-back steps (the timed loop)
best, cur = [], []
for a in steps:
    if cur and a == cur[-1] + 3: cur.append(a)
    else:
        if len(cur) > len(best): best = cur
        cur = [a]
if len(cur) > len(best): best = cur
best = best[5:]            # skip warm-up
import statistics as st
d = lambda j: [ (seq[i + j][2] - seq[i + j][1]) / 1e3 for i in best ]
g = lambda j: [ (seq[i + j + 1][1] - seq[i + j][2]) / 1e3 for i in best[:-1] ]
out = {"steps": len(best), "P_us": round(st.median(d(0)), 1), "N_us": round(st.median(d(1)), 1), "Pt_us": round(st.median(d(2)), 1),
       "gap_P_to_N_us": round(st.median(g(0)), 1), "gap_N_to_Pt_us": round(st.median(g(1)), 1), "gap_Pt_to_next_P_us": round(st.median(g(2)), 1),
       "step_us_start_to_start": round(st.median([(seq[b][1] - seq[a][1]) / 1e3 for a, b in zip(best[:-1], best[1:])]), 1)}
print(json.dumps(out))
open(O + "/step_gaps.json", "w").write(json.dumps(out))
PY
find $O/kt -name "*_trace.csv" -delete
