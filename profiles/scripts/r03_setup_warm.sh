#!/bin/bash
# kernels of the one-off setup at C4 size, cold (first build in the process) and warm (second build)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_setup_warm
rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --hip-runtime-trace --stats -d $O/kt --output-format csv -- python3 $R/profiles/scripts/setup_calls.py > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
grep '"rep"' $O/kt.log > $O/setup_calls_under_profiler.jsonl
python3 - <<'PY'
import csv, glob, os, collections, json
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_setup_warm"
f = glob.glob(O + "/kt/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the second build starts at the second k_real_spectrum / first kernel after the first matvec
names = [r["Kernel_Name"] for r in rows]
marks = [i for i, n in enumerate(names) if "k_real_spectrum" in n]
cut = marks[1] if len(marks) > 1 else len(rows)
# walk back to the input generation of rep 1 (torch kernels before the spectrum)
out = []
for label, part in (("cold", rows[:cut]), ("warm", rows[cut:])):
    agg = collections.OrderedDict()
    for r in part:
        n = r["Kernel_Name"]
        if "rocprim" in n:
            import re
            m = re.search(r"detail::(\w+)<", n.split("trampoline_kernel<")[-1])
            n = "rocprim " + (m.group(1) if m else "?") + (" u64" if "unsigned long" in n.split("trampoline_kernel<")[-1][:400] else "") + " #%d" % (int(r["Grid_Size"]) if "Grid_Size" in r else 0)
        n = n[:70]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += d
    top = sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]
    out.append({"build": label, "kernel_ms_total": round(sum(v[1] for v in agg.values()), 2),
                "top": [[k, v[0], round(v[1], 3)] for k, v in top]})
with open(O + "/setup_kernels_cold_warm.json", "w") as fh:
    json.dump(out, fh, indent=1)
for o in out:
    print(o["build"], o["kernel_ms_total"])
    for k, c, ms in o["top"]:
        print("   %-70s %3d %8.3f" % (k, c, ms))
PY
f=$(find $O/kt -name "*_hip_api_stats.csv" | head -1); [ -n "$f" ] && head -25 $f | cut -c1-150 && cp $f $O/hip_api_stats.csv
find $O/kt -name "*_kernel_trace.csv" -delete
find $O/kt -name "*_hip_api_trace.csv" -delete
