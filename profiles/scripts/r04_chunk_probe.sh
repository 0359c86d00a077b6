#!/bin/bash
# r04: chunk-local tile order go / no-go: times, then FETCH_SIZE / WRITE_SIZE passes
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_chunk
rm -rf $O; mkdir -p $O
timeout -k 10 300 python3 $R/profiles/scripts/r04_chunk_probe.py > $O/times.jsonl 2> $O/times.err || { tail -20 $O/times.err; exit 1; }
cat $O/times.jsonl
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "k_os_real" -d $O/$c --output-format csv -- python3 $R/profiles/scripts/r04_chunk_probe.py > $O/$c.log 2>&1 || { tail -5 $O/$c.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04_chunk"
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(O + "/" + c + "/**/*_counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_os_real" in r["Kernel_Name"] and "ELi0E" not in r["Kernel_Name"]]
    # launches in program order: time-order apply is MODE 0 (excluded); 11 launches per (order, lists) leg
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    vals = [float(r["Counter_Value"]) for r in rows]
    legs = [vals[i:i + 11] for i in range(0, len(vals), 11)]
    for i, leg in enumerate(legs):
        med = sorted(leg)[len(leg) // 2]
        print(c, "leg", i, "launches", len(leg), "median KB %.0f = %.3f GB" % (med, med * 1024 / 1e9))
PY
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*_agent_info.csv" -delete
