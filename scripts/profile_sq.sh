#!/bin/bash
# GPU box: shader-sequencer counters of the dominant kernels (one pass of 8 SQ counters per workload).
set -e
tag=${1:-sq}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in ${WORKLOADS:-cfg3 cfg2 cfg4s cfg3h}; do
    timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
        --kernel-trace -d $out/${tag}_${w} -o r --output-format csv -- \
        python3 $root/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/${tag}_${w}.err
    echo "sq $w"
done
