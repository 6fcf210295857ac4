#!/bin/bash
# GPU box: kernel durations of one benchmark workload grouped by (kernel, grid size) - tells the launches of one kernel
# apart that serve different network levels.  Usage: bash scripts/prof_by_grid.sh <workload> [steps] [env assignments...]
w=${1:-cfg5}; steps=${2:-3}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/bygrid_dir
MZ_BENCH_MIN_S=0.05 timeout -k 10 300 rocprofv3 --kernel-trace -d $out/bygrid_dir -o r --output-format csv -- \
    python3 $root/bench.py --workload $w --steps $steps --warmup 2 --no-cpu-baseline > $out/bygrid_bench.json 2> $out/bygrid.err || { tail -5 $out/bygrid.err; exit 1; }
f=$(find $out/bygrid_dir -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name'].split('(')[0][-48:]
    key = (name, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Grid_Size_Y', ''), r.get('Workgroup_Size_X', ''))
    agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
    v.sort()
    print('%-50s grid %s x %s wg %s  calls %6d  avg %7.2f us  median %7.2f  p10 %7.2f  p90 %7.2f' % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v), v[len(v) // 2], v[len(v) // 10], v[len(v) * 9 // 10]))
PY
