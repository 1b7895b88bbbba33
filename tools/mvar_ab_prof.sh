ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/mvp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for sw in none products; do
  if [ $sw = none ]; then unset SC_MVAR_INVERSE; else export SC_MVAR_INVERSE=$sw; fi
  rocprofv3 --kernel-trace --stats -d $OUT/kt_$sw -- python $ROOT/tools/mvar_time.py 128 1792 256 > $OUT/run_$sw.txt 2> $OUT/kt_$sw.err
  db=$(find $OUT/kt_$sw -name "*.db" | head -1)
  echo "== $sw"; grep "C=" $OUT/run_$sw.txt
  [ -n "$db" ] && python $ROOT/tools/rocpd_summary.py $db 2>&1 | head -12
  rm -rf $OUT/kt_$sw
done
