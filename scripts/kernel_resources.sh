#!/bin/bash
# Registers, spills, LDS and occupancy of every kernel as the compiler reports them (device code only, nothing is linked),
# for all translation units of the library; one line per kernel.
# usage: scripts/kernel_resources.sh [name filter] > report
cd "$(dirname "$0")/.."
echo "kernel | SGPRs | VGPRs | AGPRs | scratch bytes/lane | waves/SIMD | SGPR spills | VGPR spills | static LDS bytes/block"
for tu in engine mfma_kernels chain_kernels cached_kernels_a cached_kernels_b cached_kernels_c; do
    opt=$(python3 -c "import sys; sys.path.insert(0, '.'); from muzero_amd.build import OPT, OPT_DEFAULT; print(OPT.get('$tu.hip', OPT_DEFAULT))")
    echo "== muzero_amd/csrc/$tu.hip ($opt)"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 $opt -ffp-contract=off -std=c++17 --offload-device-only -c \
        -Rpass-analysis=kernel-resource-usage muzero_amd/csrc/$tu.hip -o /tmp/mz_${tu}_dev.o 2>&1 |
        awk -v pat="${1:-.}" '
            /Function Name:/ { name = $0; sub(/^.*Function Name: /, "", name); sub(/ \[-Rpass.*$/, "", name); keep = (name ~ pat); n = 0; next }
            keep && /remark:/ { v = $0; sub(/ \[-Rpass.*$/, "", v); sub(/^.*: /, "", v); f[n++] = v
                                if (n == 9) { cmd = "c++filt \"" name "\""; cmd | getline pretty; close(cmd)
                                              printf "%s | %s | %s | %s | %s | %s | %s | %s | %s\n", pretty, f[0], f[1], f[2], f[3], f[5], f[6], f[7], f[8] } }'
done
