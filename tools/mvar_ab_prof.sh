#!/bin/bash
# rocprofv3 --kernel-trace --stats over tools/mvar_time.py 128 1792 256 for a list of SC_MVAR_INVERSE values ("none" = unset):
# registers = the round-3 inverse; 1 / 2 / 3 = timing ablations of m_inverse_mfma (no matrix-core updates / no pivot steps /
# neither: the results are garbage, only the kernel's time means something).   usage: mvar_ab_prof.sh [C] values...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/mvp
C=${1:-128}; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for sw in "${@:-none}"; do
  if [ $sw = none ]; then unset SC_MVAR_INVERSE; else export SC_MVAR_INVERSE=$sw; fi
  rocprofv3 --kernel-trace --stats -d $OUT/kt_$sw -- python $ROOT/tools/mvar_time.py $C 1792 256 > $OUT/run_$sw.txt 2> $OUT/kt_$sw.err
  db=$(find $OUT/kt_$sw -name "*.db" | head -1)
  echo "== SC_MVAR_INVERSE=$sw"; grep "C=\|factor\|timers" $OUT/run_$sw.txt
  [ -n "$db" ] && python $ROOT/tools/rocpd_summary.py $db 2>&1 | grep -v "^#" | head -9
  rm -rf $OUT/kt_$sw
done
