#!/bin/bash
# run-coded vs inverse lists at the bench configurations (N^-1 stage and step)
set -o pipefail
mkdir -p gpurun_out
for c in c5 c3 c4; do
  for lists in rc inv; do
    CM2_OS_LISTS=$lists timeout -k 10 400 python bench.py --config $c --no-filters --no-cpu --no-raster > gpurun_out/r3_cfg_${c}_$lists.json 2> gpurun_out/r3_cfg_${c}_$lists.err || { tail -3 gpurun_out/r3_cfg_${c}_$lists.err; exit 1; }
    python - <<PY
import json
r = json.load(open("gpurun_out/r3_cfg_${c}_$lists.json"))
print("$c", "$lists", round(r["ms_per_step"], 4), {k[:10]: v["ms"] for k, v in r["stages"].items()}, r["pcg"]["seconds"], r["pcg"]["iters"])
PY
  done
done
