ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for L in 4000 4096; do
  rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/grp_$L -- python $ROOT/tools/granger_window_time.py $L 64 200 > $ROOT/gpurun_out/gr_$L.txt 2> $ROOT/gpurun_out/gr_$L.err
  db=$(find $ROOT/gpurun_out/grp_$L -name "*.db" | head -1)
  echo "== L=$L"; grep "L=" $ROOT/gpurun_out/gr_$L.txt
  [ -n "$db" ] && python $ROOT/tools/rocpd_summary.py $db 2>&1 | grep -v "^#" | head -12
  rm -rf $ROOT/gpurun_out/grp_$L
done
