#!/bin/bash
# Variant libraries with another instruction-scheduling strategy for the matrix-core translation units:
#   scripts/exp_sched_flags.sh build   (here, on the CPU)   then   scripts/exp_sched_flags.sh run   (on the GPU)
cd "$(dirname "$0")/.."
variants="max-ilp max-memory-clause iterative-ilp"
if [ "$1" = build ]; then
  for v in $variants; do
    mkdir -p /tmp/sched_$v
    for tu in cached_kernels_b mfma_kernels; do
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Os -mllvm --amdgpu-sched-strategy=$v \
        -c muzero_amd/csrc/$tu.hip -o /tmp/sched_$v/$tu.o || exit 1 &
    done
    wait
    objs=""
    for tu in engine chain_kernels cached_kernels_a cached_kernels_c; do objs="$objs muzero_amd/build/$tu.o"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/sched_$v/cached_kernels_b.o /tmp/sched_$v/mfma_kernels.o \
      -o muzero_amd/libmuzero_amd_sched_$v.so || exit 1
    echo built $v
  done
else
  for v in $variants; do
    for w in cfg3 cfg5; do
      bash scripts/ab_bench.sh $w 30 muzero_amd/libmuzero_amd.so muzero_amd/libmuzero_amd_sched_$v.so 1 || exit 1
    done
  done
fi
