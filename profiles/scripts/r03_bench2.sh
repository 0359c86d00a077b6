#!/bin/bash
# two default bench runs, key numbers of each
set -o pipefail
mkdir -p gpurun_out
for i in 1 2; do
timeout -k 10 300 python bench.py > gpurun_out/r3_bench_b$i.json 2> gpurun_out/r3_bench_b$i.err || { tail -5 gpurun_out/r3_bench_b$i.err; exit 1; }
python - <<PY
import json
r = json.load(open("gpurun_out/r3_bench_b$i.json"))
print("c4", round(r["ms_per_step"], 4), r["roofline"]["frac"], {k[:14]: v["ms"] for k, v in r["stages"].items()}, r["pcg"]["seconds"], r["pcg"]["two_level"]["seconds"])
print("   setup", r["setup_split_seconds"])
u = r["uneven_hit_map"]; print("   uneven", u["ms_per_step"], u.get("stages_ms"), u["hot_pixel"]["ms_per_step"], u["hot_pixel"]["PT_ms"])
PY
done
