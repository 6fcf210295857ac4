#!/bin/bash
# GPU box: kernel-time statistics and HBM-traffic counters of the benchmark workloads.
# Usage: bash scripts/profile_round.sh <tag>; outputs under gpurun_out/<tag>_*.
# Counter passes run on their own (never together with API tracing), one counter per pass.
set -e
tag=${1:-prof}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
# the default command (the headline workload, cfg3), exactly as the driver runs it
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/${tag}_stats_default -o r --output-format csv -- \
    python3 $root/bench.py > $out/${tag}_stats_default.json 2> $out/${tag}_stats_default.err
echo "profiled the default command"
export MZ_BENCH_MIN_S=0.5   # the other passes time one short block each (set here: nothing may stand between rocprofv3's "--" and the program)
for w in ${WORKLOADS:-cfg3 cfg2 cfg4s cfg3h cfg5}; do
    timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $out/${tag}_stats_$w -o r --output-format csv -- \
        python3 $root/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $out/${tag}_stats_$w.json 2> $out/${tag}_stats_$w.err
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -s KILL 240 rocprofv3 --pmc $c --kernel-trace -d $out/${tag}_pmc_${w}_$c -o r --output-format csv -- \
            python3 $root/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/${tag}_pmc_${w}_$c.err
    done
    echo "profiled $w"
done
