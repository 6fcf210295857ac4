#!/bin/bash
# GPU box: matrix-core counters of the workloads whose leaves are evaluated with MFMA (one pass, 5 SQ counters).
set -e
tag=${1:-mfma}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in cfg3 cfg5; do
    timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES \
        --kernel-trace -d $out/${tag}_${w} -o r --output-format csv -- \
        python3 $root/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_${w}.json 2> $out/${tag}_${w}.err
    echo "mfma $w"
done
