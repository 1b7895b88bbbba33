set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/final
python bench.py --steps 20 --warmup 3 > gpurun_out/final/bench_line.json 2> gpurun_out/final/bench_err.txt
rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats -d /tmp/prof -o res -- python bench.py --steps 5 --warmup 2 > gpurun_out/final/bench_under_rocprof.json 2>/dev/null
python tools/rocpd_summary.py $(find /tmp/prof -name "*.db" | head -1) > gpurun_out/final/kernel_stats.txt 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE"; do rm -rf /tmp/pmc; rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc -o res -- python bench.py --steps 2 --warmup 1 > /dev/null 2>&1; python tools/rocpd_summary.py $(find /tmp/pmc -name "*.db" | head -1) 2>&1 | grep -E "n=" >> gpurun_out/final/hbm_traffic.txt; done
for pass in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM"; do rm -rf /tmp/pmc; rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc -o res -- python bench.py --steps 3 --warmup 1 > /dev/null 2>&1; python tools/rocpd_summary.py $(find /tmp/pmc -name "*.db" | head -1) 2>&1 | grep -E "n=" | grep -E "fused_csm|mtfft16|combine" >> gpurun_out/final/pmc_sq.txt; done
python tools/fused_ablation.py > gpurun_out/final/fused_ablation.txt 2>&1
python tools/mtfft_ablation.py > gpurun_out/final/mtfft_ablation.txt 2>&1
./tools/hbm_write_bench > gpurun_out/final/hbm_write.txt 2>&1
./tools/abs_loop_bench > gpurun_out/final/abs_loop.txt 2>&1
tail -c 600 gpurun_out/final/bench_line.json; cat gpurun_out/final/kernel_stats.txt | head -12
