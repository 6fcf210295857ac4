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
for w in cfg2 cfg4s cfg4 cfg3b cfg3h cfg3s cfg5 cfg5d cfg5s cfg5h; do
    timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline 2> gpurun_out/${tag}_$w.err | tail -1 >> $out
    echo "done $w"
done
# experiments (their config.workload says so): the Atari-shaped search at 8192 trees, without the cached PUCT decisions,
# and as one chained launch
extra=gpurun_out/${tag}_experiment_lines.jsonl
: > $extra
MZ_BENCH_TREES=8192 timeout -k 10 300 python bench.py --workload cfg5 --no-cpu-baseline 2> gpurun_out/${tag}_cfg5_8192.err | tail -1 >> $extra
MZ_BENCH_TREES=8192 timeout -k 10 300 python bench.py --workload cfg5d --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/${tag}_cfg5d_8192.err | tail -1 >> $extra
MZ_PUCT_CACHE=0 timeout -k 10 300 python bench.py --workload cfg5 --no-cpu-baseline 2> gpurun_out/${tag}_cfg5_nocache.err | tail -1 >> $extra
MZ_PUCT_CACHE=0 timeout -k 10 300 python bench.py --workload cfg5d --no-cpu-baseline 2> gpurun_out/${tag}_cfg5d_nocache.err | tail -1 >> $extra
MZ_SEARCH_CHAIN=1 timeout -k 10 300 python bench.py --workload cfg5 --no-cpu-baseline 2> gpurun_out/${tag}_cfg5_chain.err | tail -1 >> $extra
echo "done the experiments"
