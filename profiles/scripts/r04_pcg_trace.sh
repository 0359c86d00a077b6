# kernel timeline of the PCG loop of bench.py (C4): where an iteration's 1.67 ms go beside the 1.45 ms matvec
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_pcg
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/kt --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-filters --no-raster --no-parity --deflation 0 > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
python3 - <<PY
import csv, glob, json, collections
f = glob.glob("$O/kt/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].split("::")[-1].split("<")[0].strip()
names = [short(r["Kernel_Name"]) for r in rows]
idx = [i for i, n in enumerate(names) if n == "k_pcg_update_xr"]
# iterations between consecutive update_xr kernels (the two timed PCG runs: take the last 7 intervals)
out = []
for a, b in zip(idx[-8:-1], idx[-7:]):
    seg = rows[a + 1:b + 1]
    t0 = int(rows[a]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    span = int(seg[-1]["End_Timestamp"]) - t0
    per = collections.OrderedDict()
    for r in seg:
        per[short(r["Kernel_Name"])] = per.get(short(r["Kernel_Name"]), 0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out.append({"span_us": span / 1e3, "busy_us": busy / 1e3, "idle_us": (span - busy) / 1e3, "kernels": len(seg),
                "per_kernel_us": {k: round(v, 1) for k, v in per.items()}})
json.dump(out, open("$R/gpurun_out/r04_pcg_iteration_trace.json", "w"), indent=1)
print(json.dumps(out[-1], indent=1))
print([round(o["span_us"]) for o in out], [round(o["idle_us"]) for o in out])
PY
rm -rf $O/kt
