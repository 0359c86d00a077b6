#!/bin/bash
# run-coded lists cut by time (rc) against inverse lists cut by address (inv): P, N^-1, P^T on the
# three hit maps, twice each
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r3_inv_ab.jsonl; : > $O
run() { echo "# $*" >> $O; env "$@" python profiles/scripts/uneven_probe.py 2>> gpurun_out/r3_inv_ab.err | cut -c1-1200 >> $O || exit 1; }
run CM2_OS_LISTS=rc
run CM2_OS_LISTS=inv
run CM2_OS_LISTS=rc
run CM2_OS_LISTS=inv
python - <<'PY'
import json
for l in open("gpurun_out/r3_inv_ab.jsonl"):
    if l.startswith("#"): print(l.strip()); continue
    r = json.loads(l); print("   ", r["hit_map"], r["tiles"], r["P"], r["N^-1"], r["P^T"], r["os"]["os_lists"], r["os"]["tile_bytes_per_sample"])
PY
