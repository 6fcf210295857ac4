#!/bin/bash
# GPU box: every benchmark workload once, the default one with its CPU baseline; lines under gpurun_out/<tag>_*.json
set -e
tag=${1:-full}
timeout -k 10 300 python bench.py > gpurun_out/${tag}_cfg2.json 2> gpurun_out/${tag}_cfg2.err
for w in cfg4s cfg4 cfg3 cfg3h cfg5h; do
    timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_$w.json 2> gpurun_out/${tag}_$w.err
    echo "done $w"
done
cat gpurun_out/${tag}_*.json
