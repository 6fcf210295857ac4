#!/bin/bash
# GPU box: the Atari-shaped search with the second network level's blocks forced to 4 / 6 / 8 / 12 / 16 column tiles
cd ${GRAFT_REPO_ROOT:-.}
for w in 8 4 6 12 16 8; do
  MZ_LVL_FORCE_W=$w MZ_BENCH_MIN_S=2 timeout -k 10 180 python bench.py --no-cpu-baseline --workload cfg5 --steps 30 --warmup 5 > gpurun_out/w.json 2> gpurun_out/w.err || { echo fail; tail -3 gpurun_out/w.err; exit 1; }
  python -c "
import json;d=json.loads(open('gpurun_out/w.json').read().strip().splitlines()[-1]);print('W=$w ms/step %.4f' % d['ms_per_step'])"
done
