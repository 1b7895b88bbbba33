#!/bin/bash
# Profile collection of a round (ROUND=r03 by default; run on the GPU box from the repo root; raw outputs under gpurun_out/$ROUND/, summaries are
# written into profiles/ by tools/profile_round.py afterwards):
#   1. the bench line itself                                 python bench.py
#   2. per-kernel durations of the same command              rocprofv3 --kernel-trace --stats
#   3. HBM traffic, two separate PMC passes                  rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE
#   4. SQ counters of the dominant kernel                    rocprofv3 --kernel-trace --pmc SQ_...
# (counters in their own runs with --kernel-trace only: MI355X_MICROARCH.md, rocprofv3 PMC slots)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=${ROUND:-r06}
OUT=$ROOT/gpurun_out/$ROUND
mkdir -p $OUT
cd $ROOT
python bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_line.err
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 40 --warmup 2 --timed-only"
rocprofv3 --kernel-trace --stats -d $OUT/kt -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
SHORT="python $ROOT/bench.py --steps 2 --warmup 1 --timed-only"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -- $SHORT > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -- $SHORT > /dev/null 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/sq -- $SHORT > /dev/null 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_LDS -d $OUT/sq2 -- $SHORT > /dev/null 2> $OUT/sq2.err
# BASELINE configs[3] (pairwise Granger): HBM traffic of the resident Wilson kernel, same two passes
CFG4="python $ROOT/bench.py --config cfg4 --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/kt4 -- python $ROOT/bench.py --config cfg4 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg4_under_rocprof.json 2> $OUT/kt4.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch4 -- $CFG4 > /dev/null 2> $OUT/fetch4.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write4 -- $CFG4 > /dev/null 2> $OUT/write4.err
cd $ROOT
for d in kt fetch write sq sq2 kt4 fetch4 write4; do
    db=$(find $OUT/$d -name "*.db" | head -1)
    [ -n "$db" ] && python tools/rocpd_summary.py $db > $OUT/$d.txt 2>&1
    rm -rf $OUT/$d          # (the raw databases are hundreds of MB: gpurun copies back at most 64 MB)
done
if [ "${PARTIAL:-0}" = "1" ]; then      # only the bench line, its kernel stats and the PMC passes (after a change to a kernel in bench.py's source hash)
  ls -la $OUT; exit 0
fi
# Round 6: only what changed or what the bench line needs is measured again (the review of round 5: GPU minutes belong to kernels, not to
# re-measuring unchanged ones); FULL=1 adds the whole round-5 set (MVAR sizes, stage-B ablations, issue rates, ...), whose r05 files stand.
python tools/shape_sweep.py > $OUT/shape_sweep.txt 2>&1
MIX_GEOS=0,1 python tools/stage_a_mixed.py time > $OUT/stage_a_mixed.txt 2>&1
python tools/e2e_lengths.py 256 250 200 500 1000 1024 > $OUT/e2e_lengths.txt 2>&1
for c in cfg2 cfg4 cfg5; do python bench.py --config $c --steps 10 --warmup 3 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; done
python tools/api_timeline.py > $OUT/api_timeline.txt 2>&1
bash tools/sharded_one_rank.sh > $OUT/sharded_one_rank.txt 2>&1
if [ "${FULL:-0}" = "1" ]; then
  python tools/mvar_time.py 64 1792 256 > $OUT/mvar_64ch.txt 2>&1
  python tools/mvar_time.py 128 1792 256 > $OUT/mvar_128ch.txt 2>&1
  python tools/mvar_size_time.py > $OUT/mvar_size_time.txt 2>&1
  python tools/stage_a_ab.py 0 16 2 4 6 > $OUT/stage_a_ab.txt 2>&1
  SC_AB_LONG=1 python tools/stage_a_ab.py 0 16 2 4 6 >> $OUT/stage_a_ab.txt 2>&1
  python tools/engine_time.py > $OUT/engine_time.txt 2>&1
  python tools/stage_a_breakdown.py > $OUT/stage_a.txt 2>&1
  python tools/plane_pass_time.py > $OUT/plane_pass.txt 2>&1
  python tools/fused_ablation.py > $OUT/fused_ablation.txt 2>&1
  python tools/fused2_time.py 0 1 2 3 8 9 10 11 64 > $OUT/fused2_ablation.txt 2>&1
  python tools/stage_a_planes_ab.py > $OUT/stage_a_planes_ab.txt 2>&1
  python tools/stage_a_planes_check.py > $OUT/stage_a_planes_check.txt 2>&1
  python tools/api_wall.py > $OUT/api_wall.txt 2>&1
  python tools/numpy_host_time.py > $OUT/numpy_host.txt 2>&1
  python tools/stage_a_long.py > $OUT/stage_a_wide.txt 2>&1
  python tools/stage_a_antiphase_ab.py > $OUT/stage_a_antiphase_ab.txt 2>&1
  python tools/global_time.py > $OUT/global_canonical.txt 2>&1
  python tools/measure_table.py > $OUT/measure_table.txt 2>&1
  python tools/fused2_fold_ab.py > $OUT/fused2_fold_ab.txt 2>&1
  [ -x tools/issue_rates_f64 ] && ./tools/issue_rates_f64 > $OUT/issue_rates_f64.txt 2>&1
fi
ls -la $OUT; du -sh $OUT
