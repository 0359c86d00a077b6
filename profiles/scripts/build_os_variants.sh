#!/bin/bash
# Several builds of libcosmomap2_hip.so that differ only in the -D switches of cm2_fft_real.hip
# (local cross-compilation; the probe loads each through CM2_LIB_PATH).  usage: name:"flags" ...
set -e
cd "$(dirname "$0")/../.."
mkdir -p cosmomap2_amd/csrc/build/variants
FL="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=fast -fno-fast-math -Wall -Wno-unused-result -munsafe-fp-atomics"
OBJS=$(ls cosmomap2_amd/csrc/build/*.o | grep -v cm2_fft_real.o)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc $FL $flags -Rpass-analysis=kernel-resource-usage -c cosmomap2_amd/csrc/cm2_fft_real.hip -o /tmp/emu/real_$name.o 2> /tmp/emu/real_$name.log
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o cosmomap2_amd/csrc/build/variants/lib_$name.so $OBJS /tmp/emu/real_$name.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
    echo "$name: $(grep -A8 'k_os_realILi32ELi2ELi2' /tmp/emu/real_$name.log | grep 'VGPRs Spill' | head -1 | sed 's/.*remark: *//')" ) &
done
wait
