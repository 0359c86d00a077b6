python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest5.log 2>&1; tail -3 gpurun_out/r3_gputest5.log
python profiles/scripts/setup_calls.py 2>/dev/null | tee gpurun_out/r3_setup_calls2.jsonl | cut -c1-700
python bench.py > gpurun_out/r3_bench_c4_b.json 2> gpurun_out/r3_bench_c4_b.err; python - <<'PY'
import json
r=json.load(open("gpurun_out/r3_bench_c4_b.json"))
print(r["ms_per_step"], r["roofline"]["frac"], {k[:20]:v["ms"] for k,v in r["stages"].items()}, r["pcg"]["iters"], r["pcg"]["seconds"], r["pcg"]["two_level"]["iters"], r["pcg"]["two_level"]["seconds"], r["pcg"]["two_level"]["build_seconds"])
print(r["uneven_hit_map"]); print(r["setup_seconds"], r["setup_split_seconds"])
PY
