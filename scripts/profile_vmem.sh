#!/bin/bash
# GPU box: vector-memory and LDS counters of the headline kernel (three passes, each within the per-block limits).
set -e
tag=${1:-vmem}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
pass() {
    timeout -s KILL 240 rocprofv3 --pmc $2 --kernel-trace -d $out/${tag}_$1 -o r --output-format csv -- \
        python3 $root/bench.py --workload cfg3 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/${tag}_$1.err
    echo "vmem pass $1"
}
pass sq1 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
pass sq2 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES TA_BUSY_avr"
pass tcp "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
