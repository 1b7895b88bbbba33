#!/bin/bash
# rocprofv3 --kernel-trace --stats over the full Wilson factorisation + DTF (tools/mvar_time.py C T window): per-kernel times of sc_mvar.hip
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/mvar_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in 128 130 64; do
  rocprofv3 --kernel-trace --stats -d $OUT/kt_$c -- python $ROOT/tools/mvar_time.py $c 1792 256 > $OUT/run_$c.txt 2> $OUT/kt_$c.err
  db=$(find $OUT/kt_$c -name "*.db" | head -1)
  echo "== C=$c"; grep "C=" $OUT/run_$c.txt
  [ -n "$db" ] && python $ROOT/tools/rocpd_summary.py $db 2>&1 | head -24
  rm -rf $OUT/kt_$c
done
