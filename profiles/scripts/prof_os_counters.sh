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
    timeout -k 10 150 rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex overlap_save_reg -d $O/$name --output-format csv -- $B > $O/$name.log 2>&1 || { tail -5 $O/$name.log; return 1; }
    grep -q "Memory access fault" $O/$name.log && return 1
    echo "$name done"
}
[ -n "$SKIP_SQ" ] || pass sq1 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM || exit 1
[ -n "$SKIP_SQ" ] || pass sq2 SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR || exit 1
# (a pass with the TA_* / TCP_* stall counters never finished on this pool and was dropped)
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT || exit 1
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*_agent_info.csv" -delete
ls $O/*/*/ | head -20
