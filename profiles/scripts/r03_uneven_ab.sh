#!/bin/bash
# plan builders / slice lengths on the uniform, uneven and hot-pixel hit maps
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r3_uneven_ab.jsonl; : > $O
run() { echo "# $*" >> $O; env "$@" python profiles/scripts/uneven_probe.py 2>> gpurun_out/r3_uneven_ab.err | cut -c1-1200 >> $O || exit 1; }
run CM2_X=default
run CM2_PT_SLICE=1536
run CM2_PT_SLICE=1280
run CM2_FX_BUILD=serial
python - <<'PY'
import json
for l in open("gpurun_out/r3_uneven_ab.jsonl"):
    if l.startswith("#"): print(l.strip()); continue
    r = json.loads(l); print("   ", r["hit_map"], r["tiles"], r["P"], r["N^-1"], r["P^T"], r["plan"])
PY
