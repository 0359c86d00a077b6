#!/bin/bash
# Why does N^-1 cost 43 % more per sample at C5 WHOLE (1e9 samples, TOD buffers of 8 GB) than on one GPU's share
# (1.25e8, 1 GB)?    Same nside, tiles, lists and run lengths; what differs is the
# address range a window's ~1500 runs are spread over.
#   part 1: timed loop of bench.py at both sizes, global tile order against the span order (CM2_TILE_SPAN: a
#           window's addresses then live inside one span), alternated
#   part 2: counters of k_os_real at both sizes, <= 2 counters of one block per pass (prof_os_counters.sh says why)
# Usage: bash profiles/scripts/r05_c5_whole_probe.sh [timing|counters|both]
set -o pipefail
what=${1:-both}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c5w
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu --no-filters --no-parity --no-pcg --no-raster --deflation 0 --steps 10 --warmup 2"
line() {   # name size-args env-assignments...
    local name=$1 size=$2; shift 2
    env "$@" python3 $R/bench.py --config c5 $size $COMMON 2> $O/$name.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'run': '$name', 'nt': d['config']['nt_per_gpu'], 'ms_per_step': round(d['ms_per_step'], 4),
                  'stages_ms': {k[:5]: round(v['ms'], 4) for k, v in d['stages'].items()},
                  'per_1e8': {k[:5]: round(v['ms'] * 1e8 / d['config']['nt_per_gpu'], 4) for k, v in d['stages'].items()},
                  'tile_plan': d['config']['tile_plan']}))" | tee -a $O/timing.jsonl
}
if [ "$what" != counters ]; then
  rm -f $O/timing.jsonl
  for rep in 1 2; do
    line share_default_$rep "" CM2_X=0 || exit 1
    line whole_default_$rep "--scaling strong --gpus 1" CM2_X=0 || exit 1
    line whole_span_auto_$rep "--scaling strong --gpus 1" CM2_TILE_SPAN=auto || exit 1
    line whole_span_8M_$rep "--scaling strong --gpus 1" CM2_TILE_SPAN=8388608 || exit 1
    line whole_span_64M_$rep "--scaling strong --gpus 1" CM2_TILE_SPAN=67108864 || exit 1
  done
fi
if [ "$what" != timing ]; then
  rocprofv3 -L > $O/counters_available.txt 2>&1 || true
  grep -i -E "utcl|tlb|translat|xnack" $O/counters_available.txt | head -60 > $O/translation_counters.txt
  have() { grep -qw "$1" $O/counters_available.txt; }
  pass() {   # name size-args counters...
      local name=$1 size=$2; shift 2
      timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "k_os_real" -d $O/$name --output-format csv -- python3 $R/bench.py --config c5 $size --no-cpu --no-filters --no-parity --no-pcg --no-raster --deflation 0 --steps 3 --warmup 1 > $O/$name.log 2>&1 || { tail -5 $O/$name.log; return 1; }
      grep -q "Memory access fault" $O/$name.log && return 1
      echo "$name done"
  }
  n=0
  for group in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
               "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" \
               "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
               "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum" \
               "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    use=""
    for c in $group; do have $c && use="$use $c"; done
    [ -n "$use" ] || { echo "no counter of ($group) on this GPU"; continue; }
    n=$((n + 1))
    pass share_g$n "" $use || echo "share pass $n ($use) failed"
    pass whole_g$n "--scaling strong --gpus 1" $use || echo "whole pass $n ($use) failed"
  done
  find $O -name "*_kernel_trace.csv" -delete
  find $O -name "*_agent_info.csv" -delete
  python3 - <<'PY'
import csv, glob, os, collections, json
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05_c5w"
res = collections.defaultdict(dict)
for f in glob.glob(O + "/*/**/*_counter_collection.csv", recursive=True):
    run = os.path.relpath(f, O).split(os.sep)[0].split("_")[0]          # share | whole
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        res[(run, k)][c] = sorted(v)[len(v) // 2]
with open(O + "/counters.jsonl", "w") as out:
    for (run, k), d in sorted(res.items()):
        s = json.dumps({"run": run, "kernel": k, **d})
        print(s)
        out.write(s + "\n")
PY
fi
