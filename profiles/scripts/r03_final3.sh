#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest_final3.log 2>&1; tail -2 gpurun_out/r3_gputest_final3.log
CM2_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 5 --no-cpu > gpurun_out/r3_bench_2rank_b.json 2> gpurun_out/r3_bench_2rank_b.err; echo "2-rank rc=$?"
python -c "
import json
r=json.load(open('gpurun_out/r3_bench_2rank_b.json')); print(r['n_gpus'], r['distributed'], r['other_scaling_point']['scaling'], r['pcg']['iters'])"
bash profiles/scripts/profile_bench.sh r03 2>&1 | tail -2
python - <<'PY'
import json
r = json.load(open("gpurun_out/prof_r03/bench_default.json"))
print("c4", r["ms_per_step"], r["value"], r["roofline"]["frac"], {k[:12]: v["ms"] for k, v in r["stages"].items()}, r["pcg"]["seconds"], r["pcg"]["two_level"]["seconds"])
print(r["uneven_hit_map"]["ms_per_step"], r["uneven_hit_map"]["hot_pixel"]["ms_per_step"])
PY
