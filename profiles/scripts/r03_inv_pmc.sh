#!/bin/bash
# HBM traffic of k_os_real with run-coded (cut by time) and inverse (cut by address) lists
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_inv
rm -rf $O; mkdir -p $O
for lists in rc inv; do
  for c in FETCH_SIZE WRITE_SIZE; do
    CM2_OS_LISTS=$lists timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "k_os_real" -d $O/${lists}_$c --output-format csv -- python3 $R/profiles/scripts/uneven_probe.py ${MAPS:-uniform} > $O/${lists}_$c.log 2>&1 || { tail -5 $O/${lists}_$c.log; exit 1; }
  done
done
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_inv"
for d in sorted(glob.glob(O + "/*_SIZE")):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_os_real" in r["Kernel_Name"] and "ELi0E" not in r["Kernel_Name"]]
    v = sorted(v); med = v[len(v)//2]
    name = os.path.basename(d)
    # FETCH_SIZE / WRITE_SIZE are in kilobytes (guide: FETCH_SIZE x 2 for wide reads on gfx950)
    print(name, len(v), "median per launch: %.3f GB%s" % (med * 1024 / 1e9, "  (x2 = %.3f GB)" % (2 * med * 1024 / 1e9) if "FETCH" in name else ""))
PY
find $O -name "*_kernel_trace.csv" -delete
