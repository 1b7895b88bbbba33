mkdir -p gpurun_out/r04b
python -m pytest tests -x -q -m gpu > gpurun_out/r04b/gputests_all.txt 2>&1
tail -5 gpurun_out/r04b/gputests_all.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r04b/bench_full.json 2> gpurun_out/r04b/bench_full.err
cat gpurun_out/r04b/bench_full.json | cut -c1-1500
