#!/bin/bash
# parity of every overlap-save kernel / list-format variant, then their timings at C4 size
set -o pipefail
mkdir -p gpurun_out
K='toeplitz or overlap_save or tiled or forced or default_method or noise_and_filter'
for v in real16:rc real16:plain real32:rc real32:plain; do
  k=${v%%:*}; l=${v##*:}
  CM2_OS_KERNEL=$k CM2_OS_LISTS=$l timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$K" > gpurun_out/r3_os_parity_${k}_${l}.log 2>&1
  echo "$v rc=$? $(tail -1 gpurun_out/r3_os_parity_${k}_${l}.log)"
done
timeout -k 10 400 python profiles/scripts/os_probe.py > gpurun_out/r3_os_probe.jsonl 2> gpurun_out/r3_os_probe.err
cat gpurun_out/r3_os_probe.jsonl
