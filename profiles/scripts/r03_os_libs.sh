#!/bin/bash
# time the real32:rc (or $PROBE_VARIANTS) overlap-save kernel of every library build under
# cosmomap2_amd/csrc/build/variants (profiles/scripts/build_os_variants.sh)
mkdir -p gpurun_out
: > gpurun_out/r3_os_libs.jsonl
for lib in cosmomap2_amd/csrc/build/variants/lib_*.so; do
  n=$(basename $lib .so)
  CM2_LIB_PATH=$PWD/$lib PROBE_VARIANTS=${PROBE_VARIANTS:-real32:rc} timeout -k 10 300 python profiles/scripts/os_probe.py 2>> gpurun_out/r3_os_libs.err | sed "s/^{/{\"lib\": \"$n\", /" | tee -a gpurun_out/r3_os_libs.jsonl | cut -c1-150
done
