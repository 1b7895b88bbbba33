#!/bin/bash
# SQ / LDS counters of the stage-A kernels (planes-format and complex64 output) at the cfg3 volume: one rocprofv3 --pmc pass of
# tools/stage_a_planes_ab.py.  Output: gpurun_out/$ROUND/stage_a_planes_pmc.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=${ROUND:-r04}
OUT=$ROOT/gpurun_out/$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/stage_a_planes_ab.py quick"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $OUT/sapmc -- $CMD > /dev/null 2> $OUT/sapmc.err
cd $ROOT
db=$(find $OUT/sapmc -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py $db | grep -v "at::native\|rocclr" > $OUT/stage_a_planes_pmc.txt 2>&1
cat $OUT/stage_a_planes_pmc.txt
