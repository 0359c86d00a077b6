#!/bin/bash
# closing measurements, part A: smoke, GPU suite, the other configurations, the 2-rank self-launch,
# the setup split (plain and under the profiler)
set -o pipefail
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest_close.log 2>&1; rc=$?; tail -2 gpurun_out/r3_gputest_close.log
[ $rc -eq 0 ] || exit $rc
for c in c2 c3 c5; do
  timeout -k 10 400 python bench.py --config $c --no-filters > gpurun_out/r3_bench_$c.json 2> gpurun_out/r3_bench_$c.err || { tail -3 gpurun_out/r3_bench_$c.err; exit 1; }
done
CM2_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 5 --no-cpu > gpurun_out/r3_bench_2rank.json 2> gpurun_out/r3_bench_2rank.err || { tail -5 gpurun_out/r3_bench_2rank.err; exit 1; }
python profiles/scripts/setup_calls.py > gpurun_out/r3_setup_calls.jsonl 2> gpurun_out/r3_setup_calls.err || exit 1
bash profiles/scripts/r03_setup_warm.sh > gpurun_out/r3_setup_warm.log 2>&1 || { tail -5 gpurun_out/r3_setup_warm.log; exit 1; }
python - <<'PY'
import json
for c in ("c2", "c3", "c5"):
    r = json.load(open("gpurun_out/r3_bench_%s.json" % c))
    p = r.get("pcg") or {}
    print(c, round(r["ms_per_step"], 4), "%.3g" % r["value"], r["step_frac_of_hbm_peak"], p.get("iters"), (p.get("two_level") or {}).get("iters"), r.get("setup_split_seconds"))
r = json.load(open("gpurun_out/r3_bench_2rank.json")); print(r["n_gpus"], r["distributed"], r["pcg"]["iters"])
for l in open("gpurun_out/r3_setup_calls.jsonl"): print(l[:400])
PY
