#!/bin/bash
# GPU box: per-kernel time statistics of one benchmark workload (rocprofv3 --kernel-trace --stats, CSV).
# Usage: bash scripts/prof_kernels.sh <workload> <tag> [steps]; output gpurun_out/<tag>_kernel_stats.csv (+ the bench line).
w=${1:-cfg3}; tag=${2:-prof}; steps=${3:-5}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/${tag}_dir -o r --output-format csv -- \
    python3 $root/bench.py --workload $w --steps $steps --warmup 2 --no-cpu-baseline > $out/${tag}_bench.json 2> $out/${tag}.err || { tail -5 $out/${tag}.err; exit 1; }
f=$(find $out/${tag}_dir -name "*kernel_stats.csv" | head -1)
cp "$f" $out/${tag}_kernel_stats.csv
python3 - "$out/${tag}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:6]:
    print('%-70s calls %6s  avg %10.1f us  total %8.2f ms  %5s%%' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, r['Percentage']))
PY
