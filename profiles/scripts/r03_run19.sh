python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_sharded.py -m gpu -x -q -k "hot_pixel or tiled or uneven or two_ranks or c_abi or plain_c or full_size_properties_c4" > gpurun_out/r3_gputest6.log 2>&1; tail -4 gpurun_out/r3_gputest6.log
python bench.py --no-cpu --no-filters > gpurun_out/r3_bench_c4_c.json 2> gpurun_out/r3_bench_c4_c.err; python - <<'PY'
import json
r=json.load(open("gpurun_out/r3_bench_c4_c.json"))
print(r["ms_per_step"], {k[:20]:v["ms"] for k,v in r["stages"].items()})
print(r["uneven_hit_map"])
PY
