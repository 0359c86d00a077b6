#!/bin/bash
# HBM-side counters of the overlap-save kernels on the tile order (os_probe.py at C4 size), one
# rocprofv3 --pmc pass per counter group.  Usage: bash profiles/scripts/prof_os_pmc.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_os_pmc
mkdir -p $O
export PROBE_VARIANTS=${PROBE_VARIANTS:-pair,real32:rc,real32:plain,real16:rc,real16:plain}
pass() {   # name, counters...
    local name=$1; shift
    timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "k_os_real|overlap_save_reg" -d $O/$name --output-format csv -- python3 $R/profiles/scripts/os_probe.py > $O/$name.log 2>&1 || { tail -5 $O/$name.log; return 1; }
    grep -q "Memory access fault" $O/$name.log && return 1
    echo "$name done"
}
pass fetch FETCH_SIZE || exit 1
pass write WRITE_SIZE || exit 1
pass tcc TCC_HIT_sum TCC_MISS_sum || exit 1
pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum || echo "ea pass failed (counter names?)"
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*_agent_info.csv" -delete
python3 - <<'PY'
import csv, glob, os, collections, json
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_os_pmc"
res = collections.defaultdict(dict)
for f in glob.glob(O + "/*/**/*_counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        res[k][c] = sorted(v)[len(v) // 2]
for k, d in sorted(res.items()):
    print(json.dumps({"kernel": k, **d}))
PY
