#!/bin/bash
# GPU box: the Atari-shaped searches with the batch as one range or as two halves on streams of their own
cd ${GRAFT_REPO_ROOT:-.}
for wl in cfg5 cfg5d; do
for r in 1 2 1 2; do
  MZ_LEVEL_RANGES=$r MZ_BENCH_MIN_S=2 timeout -k 10 180 python bench.py --no-cpu-baseline --workload $wl --steps 20 --warmup 5 > gpurun_out/r.json 2> gpurun_out/r.err || { echo fail; tail -3 gpurun_out/r.err; exit 1; }
  python -c "
import json;d=json.loads(open('gpurun_out/r.json').read().strip().splitlines()[-1]);print('$wl ranges=$r ms/step %.4f' % d['ms_per_step'])"
done
done
MZ_LEVEL_RANGES=2 timeout -k 10 300 python -u -m pytest tests/test_gpu_cfg5_parity.py tests/test_gpu_puct_cache.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2
