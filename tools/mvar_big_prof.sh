ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/mvp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in 160 256; do
  rocprofv3 --kernel-trace --stats -d $OUT/kt_$c -- python $ROOT/tools/mvar_time.py $c 1792 256 > $OUT/run_$c.txt 2> $OUT/kt_$c.err
  db=$(find $OUT/kt_$c -name "*.db" | head -1)
  echo "== C=$c"; grep "C=\|factor\|timers" $OUT/run_$c.txt
  [ -n "$db" ] && python $ROOT/tools/rocpd_summary.py $db 2>&1 | grep -v "^#" | head -9
  rm -rf $OUT/kt_$c
done
