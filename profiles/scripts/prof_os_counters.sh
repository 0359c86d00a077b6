#!/bin/bash
# Unit-utilisation counters of k_overlap_save_reg on the tile order (os_probe.py: 11 launches at
# C4 size), one rocprofv3 --pmc pass per group.  Usage: bash profiles/scripts/prof_os_counters.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_os
mkdir -p $O
B="python3 $R/profiles/scripts/os_probe.py"
pass() {   # name, counters...
    local name=$1; shift
    timeout -k 10 150 rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "overlap_save_reg|k_os_real" -d $O/$name --output-format csv -- $B > $O/$name.log 2>&1 || { tail -5 $O/$name.log; return 1; }
    grep -q "Memory access fault" $O/$name.log && return 1
    echo "$name done"
}
[ -n "$SKIP_SQ" ] || pass sq1 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM || exit 1
[ -n "$SKIP_SQ" ] || pass sq2 SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR || exit 1
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT || exit 1
# Texture-addresser / vector-L1 counters.  Round 2 asked for eight of them in ONE --pmc group:
# rocprofiler refused it at start-up ("Could not construct profile cfg ... error code 38: Request
# exceeds the capabilities of the hardware to collect", gpurun_out/prof_os/ta.log) and the process
# then hung until the time limit -- the TA and TCP blocks have two counter slots each.  Here: at
# most two counters of one block per pass, only names `rocprofv3 -L` lists on this GPU.
if [ -z "$SKIP_TA" ]; then
  rocprofv3 -L > $O/counters_available.txt 2>&1 || true
  have() { grep -qw "$1" $O/counters_available.txt; }
  n=0
  for pair in "TA_BUSY_avr TA_TA_BUSY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
              "TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum" \
              "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
              "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
    set -- $pair
    use=""
    for c in "$@"; do have $c && use="$use $c"; done
    [ -n "$use" ] || continue
    n=$((n + 1))
    pass ta$n $use || echo "pass ta$n ($use) failed: see $O/ta$n.log"
  done
fi
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*_agent_info.csv" -delete
ls $O/*/*/ | head -20
python3 - <<'PY'
import csv, glob, os, collections, json
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_os"
res = collections.defaultdict(dict)
for f in glob.glob(O + "/*/**/*_counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:50], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        res[k][c] = sorted(v)[len(v) // 2]
for k, d in sorted(res.items()):
    print(json.dumps({"kernel": k, **d}))
PY
