#!/bin/bash
# GPU box: the Atari-shaped searches with their launches issued one by one or replayed from a captured graph
cd ${GRAFT_REPO_ROOT:-.}
for wl in cfg5 cfg5d; do
for g in 0 1 0 1; do
  MZ_LEVEL_GRAPH=$g MZ_BENCH_MIN_S=2 timeout -k 10 180 python bench.py --no-cpu-baseline --workload $wl --steps 20 --warmup 5 > gpurun_out/g.json 2> gpurun_out/g.err || { echo fail; tail -5 gpurun_out/g.err; exit 1; }
  python -c "
import json;d=json.loads(open('gpurun_out/g.json').read().strip().splitlines()[-1]);print('$wl graph=$g ms/step %.4f mean depth %.2f' % (d['ms_per_step'], d['mean_leaf_depth']))"
done
done
