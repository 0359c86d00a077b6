#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -2
python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest_final2.log 2>&1; tail -2 gpurun_out/r3_gputest_final2.log
bash profiles/scripts/profile_bench.sh r03 2>&1 | tail -2
python - <<'PY'
import json
r = json.load(open("gpurun_out/prof_r03/bench_default.json"))
print("c4", r["ms_per_step"], r["value"], r["roofline"]["frac"], {k[:12]: v["ms"] for k, v in r["stages"].items()}, r["pcg"]["seconds"], r["pcg"]["two_level"]["seconds"], r["setup_split_seconds"])
print(r["uneven_hit_map"]["ms_per_step"], r["uneven_hit_map"]["hot_pixel"]["ms_per_step"], r["cpu_baseline"]["value"], r["cpu_baseline_all_cores"]["value"])
PY
