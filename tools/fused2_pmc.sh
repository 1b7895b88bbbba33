#!/bin/bash
# SQ counters of the planes-format stage-B kernel (and the complex64 one beside it) at the cfg3 volume: two rocprofv3 --pmc
# passes of tools/fused2_time.py, each under its own timeout.  Output: gpurun_out/$ROUND/fused2_pmc_{a,b}.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=${ROUND:-r04}
OUT=$ROOT/gpurun_out/$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/fused2_time.py 0"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/f2a -- $CMD > /dev/null 2> $OUT/f2a.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_LDS -d $OUT/f2b -- $CMD > /dev/null 2> $OUT/f2b.err
cd $ROOT
for d in f2a f2b; do
    db=$(find $OUT/$d -name "*.db" | head -1)
    [ -n "$db" ] && python tools/rocpd_summary.py $db | grep -v "at::native\|rocclr\|planes_from" > $OUT/fused2_pmc_${d#f2}.txt 2>&1
done
cat $OUT/fused2_pmc_a.txt $OUT/fused2_pmc_b.txt
