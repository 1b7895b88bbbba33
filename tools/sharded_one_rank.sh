export MASTER_ADDR=127.0.0.1
for tr in 1000 500 250 125; do
  for g in 4 2 1; do
    echo "trials=$tr groups=$g"
    SC_BENCH_GROUPS=$g SC_BENCH_FORCE_SHARDED=1 SC_FORCE_EXCHANGE=1 SC_BENCH_BACKEND=nccl python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 1 --steps 20 --warmup 3 --trials $tr --no-cpu-baseline --no-f64 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step', round(d['ms_per_step'],3), d['roofline']['stage_ms'])"
  done
done
echo plain; for tr in 1000 125; do python bench.py --steps 20 --warmup 3 --trials $tr --no-cpu-baseline --no-f64 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step', round(d['ms_per_step'],3), d['roofline']['stage_ms'])"; done
