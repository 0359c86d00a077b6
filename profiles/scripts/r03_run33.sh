#!/bin/bash
# full GPU suite, then two default bench runs (fixed-order P^T with one barrier per slice)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest_33.log 2>&1; rc=$?; tail -3 gpurun_out/r3_gputest_33.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
timeout -k 10 300 python bench.py > gpurun_out/r3_bench_33_$i.json 2> gpurun_out/r3_bench_33_$i.err || exit 1
python - <<PY
import json
r = json.load(open("gpurun_out/r3_bench_33_$i.json"))
print("c4", r["ms_per_step"], r["roofline"]["frac"], {k[:14]: v["ms"] for k, v in r["stages"].items()}, r["pcg"]["seconds"], r["pcg"]["two_level"]["seconds"])
u = r["uneven_hit_map"]; print(u["ms_per_step"], u.get("stages_ms"), u["hot_pixel"])
PY
done
