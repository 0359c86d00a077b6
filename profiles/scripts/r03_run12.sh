python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest2.log 2>&1; tail -3 gpurun_out/r3_gputest2.log
for tp in 1536 2048 3072; do PROBE_TP=$tp PROBE_VARIANTS=real32:rc python profiles/scripts/os_probe.py 2>/dev/null | cut -c1-110; done
python bench.py --no-cpu > gpurun_out/r3_bench_c4_real32.json 2> gpurun_out/r3_bench_c4_real32.err; python - <<'PY'
import json
r=json.load(open("gpurun_out/r3_bench_c4_real32.json"))
print(r["ms_per_step"], r["roofline"]["frac"], {k:v["ms"] for k,v in r["stages"].items()}, r["pcg"]["iters"], r["pcg"]["seconds"], r["pcg"]["two_level"]["iters"], r["pcg"]["two_level"]["seconds"], r["uneven_hit_map"]["ms_per_step"], r["raster_pointing"]["ms_per_step"])
PY
