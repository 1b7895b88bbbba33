#!/bin/bash
# PMC passes over stage A alone at the lengths given (default 250 256 1000 1024): two sets of SQ counters per length.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/stage_a_mixed_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A SETS
SETS[sq1]="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
SETS[sq2]="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_WAVES"
for n in ${@:-250 256 1000 1024}; do
  for s in sq1 sq2; do
    STAGE_A_N=$n timeout 150 rocprofv3 --kernel-trace --pmc ${SETS[$s]} -d $OUT/${s}_$n -- python $ROOT/tools/stage_a_len.py > /dev/null 2> $OUT/${s}_$n.err
    db=$(find $OUT/${s}_$n -name "*.db" | head -1)
    echo "== N=$n counters $s"
    [ -n "$db" ] && python $ROOT/tools/rocpd_summary.py $db 2>&1 | grep -i "mtfft" | cut -c1-400
  done
done
