#!/bin/bash
# cfg3 with tiles of 16 trees (two blocks per compute unit) against tiles of 32 (one block): scripts/exp_tile16.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for r in 1 2; do
  for tm in 32 16; do
    MZ_SEARCH_TM=$tm MZ_BENCH_MIN_S=2 timeout -k 10 180 python bench.py --no-cpu-baseline --workload cfg3 --steps 50 --warmup 5 > gpurun_out/t16.json 2> gpurun_out/t16.err || { echo "bench failed"; tail -5 gpurun_out/t16.err; exit 1; }
    python - $tm <<'PY'
import json, sys
d = json.loads(open('gpurun_out/t16.json').read().strip().splitlines()[-1])
print('tile %s: ms/step %.4f kernel ms %.4f' % (sys.argv[1], d['ms_per_step'], d['roofline'].get('kernel_ms', float('nan'))))
PY
  done
done
