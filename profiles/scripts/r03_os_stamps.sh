#!/bin/bash
# diagnostic build with phase stamps, phase timing of the real-window kernels, then restore nothing
# (the box is discarded; the shipped library is built without the flag)
set -o pipefail
mkdir -p gpurun_out
touch cosmomap2_amd/csrc/cm2_fft_real.hip; CM2_EXTRA_HIPCC_FLAGS=-DCM2_OS_STAMPS python -m cosmomap2_amd.build > gpurun_out/r3_stamps_build.log 2>&1 || { tail -20 gpurun_out/r3_stamps_build.log; exit 1; }
timeout -k 10 400 python profiles/scripts/os_stamps.py > gpurun_out/r3_os_stamps.jsonl 2> gpurun_out/r3_os_stamps.err
cat gpurun_out/r3_os_stamps.jsonl; tail -3 gpurun_out/r3_os_stamps.err
