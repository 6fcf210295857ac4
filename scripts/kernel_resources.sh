#!/bin/bash
# Registers, spills, LDS and occupancy of every kernel as the compiler reports them (device code only, nothing is linked),
# for all translation units of the library.
# usage: scripts/kernel_resources.sh [name filter] > report
cd "$(dirname "$0")/.."
for tu in engine mfma_kernels chain_kernels cached_kernels_a cached_kernels_b cached_kernels_c; do
    echo "== muzero_amd/csrc/$tu.hip"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 --offload-device-only -c \
        -Rpass-analysis=kernel-resource-usage muzero_amd/csrc/$tu.hip -o /tmp/mz_${tu}_dev.o 2>&1 |
        awk -v pat="${1:-.}" '/Function Name:/ {keep = ($0 ~ pat)} keep && /remark:/ {sub(/^.*remark: [^ ]* /, ""); print}'
done
