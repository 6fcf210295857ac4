#!/bin/bash
# A/B of two builds of the library on one box: scripts/ab_bench.sh <workload> <steps> <lib A> <lib B> [rounds]
# (bench.py once per library and round, alternating; prints ms per step and the dominant kernel's ms)
cd "$(dirname "$0")/.."
wl=${1:-cfg3}; steps=${2:-50}; a=${3:-muzero_amd/libmuzero_amd_base.so}; b=${4:-muzero_amd/libmuzero_amd.so}; rounds=${5:-2}
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
    for lib in $a $b; do
        MUZERO_AMD_LIB=$PWD/$lib MZ_BENCH_MIN_S=2 timeout -k 10 180 python bench.py --no-cpu-baseline --workload $wl --steps $steps --warmup 5 > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "bench failed for $lib"; tail -5 gpurun_out/ab.err; exit 1; }
        python - "$lib" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])
print('%-40s %s ms/step %.4f kernel ms %.4f' % (sys.argv[1], d['config']['name'], d['ms_per_step'], d['roofline'].get('kernel_ms', float('nan'))))
PY
    done
done
