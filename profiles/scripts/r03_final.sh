#!/bin/bash
# the round's closing measurements: GPU suite, bench at every configuration, profile set
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest_final.log 2>&1; tail -2 gpurun_out/r3_gputest_final.log
for c in c2 c3 c5; do
  python bench.py --config $c --no-filters > gpurun_out/r3_bench_$c.json 2> gpurun_out/r3_bench_$c.err || tail -3 gpurun_out/r3_bench_$c.err
done
bash profiles/scripts/profile_bench.sh r03 2>&1 | tail -3
python - <<'PY'
import json
for c in ("c2", "c3", "c5"):
    try:
        r = json.load(open("gpurun_out/r3_bench_%s.json" % c))
        p = r.get("pcg") or {}
        print(c, round(r["ms_per_step"], 4), "%.3g" % r["value"], r["step_frac_of_hbm_peak"], p.get("iters"), (p.get("two_level") or {}).get("iters"))
    except Exception as e:
        print(c, "failed", e)
r = json.load(open("gpurun_out/prof_r03/bench_default.json"))
print("c4", r["ms_per_step"], r["value"], r["roofline"]["frac"], {k[:12]: v["ms"] for k, v in r["stages"].items()}, r["pcg"]["seconds"], r["pcg"]["two_level"]["seconds"])
PY
