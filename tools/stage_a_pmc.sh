#!/bin/bash
# PMC passes over stage A alone (cfg3 shape): two sets of SQ counters, non-temporal (default) against plain stores.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/stage_a_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A SETS
SETS[sq1]="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
SETS[sq2]="SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM"
# (a TA / TCP set aborted rocprofv3 on this image and hung its finalisation: SQ sets only, every pass under its own timeout)
for v in 0 16; do
  for s in sq1 sq2; do
    SC_MTFFT_DEBUG=$v timeout 150 rocprofv3 --kernel-trace --pmc ${SETS[$s]} -d $OUT/${s}_$v -- python $ROOT/tools/stage_a_only.py > /dev/null 2> $OUT/${s}_$v.err
    db=$(find $OUT/${s}_$v -name "*.db" | head -1)
    echo "== counters $s, SC_MTFFT_DEBUG=$v"
    [ -n "$db" ] && python $ROOT/tools/rocpd_summary.py $db 2>&1 | grep "mtfft16" | grep -v "^_Z14mtfft16.*  *[0-9][0-9]* *[0-9.]* *[0-9.]*" 
  done
done
