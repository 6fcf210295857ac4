# GPU box: timing of the level-by-level searches and the L2-miss read traffic of one of them
export MZ_BENCH_MIN_S=1
for w in cfg5 cfg5d cfg5s; do
timeout -k 10 200 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$w', round(d['ms_per_step'],3), 'ms/step')" || exit 1
done
MZ_BENCH_TREES=8192 timeout -k 10 200 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg5 8192 trees', round(d['ms_per_step'],3), 'ms/step')" || exit 1
root=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $root/gpurun_out/exp_lv_fetch -o r --output-format csv -- python3 $root/bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $root/gpurun_out/exp_lv_fetch.err
cd $root && python - <<PY
import csv, glob
f = glob.glob('gpurun_out/exp_lv_fetch/**/*counter_collection.csv', recursive=True)[0]
tot = {}
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == 'FETCH_SIZE' and 'mz::' in r['Kernel_Name']:
        k = r['Kernel_Name'].split('(')[0]
        a = tot.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
for k, (n, v) in tot.items():
    if n > 100: print(k, n, 'launches, FETCH_SIZE mean', round(v / n, 1), 'KiB')
PY
