#!/bin/bash
# GPU box: one pass of hardware counters over a benchmark workload; prints per-kernel sums.
# Usage: bash scripts/prof_pmc.sh <workload> <tag> <counter> [<counter> ...]   (at most 8 SQ_ counters per pass)
w=$1; tag=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace -d $out/${tag}_dir -o r --output-format csv -- \
    python3 $root/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/${tag}.err || { tail -5 $out/${tag}.err; exit 1; }
python3 - $out/${tag}_dir <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    n[(k, r['Counter_Name'])] += 1
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:3]:
    print(k)
    for c, v in d.items():
        print('   %-32s %16.0f per launch' % (c, v / max(n[(k, c)], 1)))
PY
