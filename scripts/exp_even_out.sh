#!/bin/bash
# The evened-out cut of the level blocks against the plain one (MZ_LVL_PLAIN=1) at several batch sizes: scripts/exp_even_out.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for trees in ${TREES:-512 1024 2048 4096}; do
  for plain in 1 0; do
    if [ $plain = 1 ]; then export MZ_LVL_PLAIN=1; else unset MZ_LVL_PLAIN; fi
    MZ_BENCH_TREES=$trees MZ_BENCH_MIN_S=2 timeout -k 10 180 python bench.py --no-cpu-baseline --workload cfg5 --steps 30 --warmup 5 > gpurun_out/eo.json 2> gpurun_out/eo.err || { echo "bench failed"; tail -5 gpurun_out/eo.err; exit 1; }
    python - $trees $plain <<'PY'
import json, sys
d = json.loads(open('gpurun_out/eo.json').read().strip().splitlines()[-1])
print('%5s trees, %s cut: ms/step %.4f  %.2f M simulations/s' % (sys.argv[1], 'plain' if sys.argv[2] == '1' else 'evened-out', d['ms_per_step'], d['value'] / 1e6))
PY
  done
done
