#!/bin/bash
# GPU box: one bench line per workload (the default one with its CPU baseline) into gpurun_out/<tag>_bench_lines.jsonl
set -e
tag=${1:-lines}
out=gpurun_out/${tag}_bench_lines.jsonl
: > $out
timeout -k 10 300 python bench.py 2> gpurun_out/${tag}_default.err | tail -1 >> $out
echo "done the default command"
export MZ_BENCH_MIN_S=2   # comparison lines: two seconds of timed work each
timeout -k 10 300 python bench.py --workload cfg1 --steps 50 --warmup 5 --no-cpu-baseline 2> gpurun_out/${tag}_cfg1.err | tail -1 >> $out
for w in cfg2 cfg4s cfg4 cfg3h cfg3s cfg5 cfg5d cfg5s cfg5h; do
    timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline 2> gpurun_out/${tag}_$w.err | tail -1 >> $out
    echo "done $w"
done
