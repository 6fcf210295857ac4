# GPU box: cached PUCT decisions on / off for the Atari-shaped level-by-level searches
export MZ_BENCH_MIN_S=1
for w in cfg5 cfg5d; do for pc in 0 1; do
MZ_PUCT_CACHE=$pc timeout -k 10 200 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$w cache=$pc', round(d['ms_per_step'],3), 'ms/step, mean leaf depth', round(d['mean_leaf_depth'],2))" || exit 1
done; done
