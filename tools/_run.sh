python bench.py > gpurun_out/r2/bench_c3.json 2> gpurun_out/r2/bench_c3.err; tail -c 3000 gpurun_out/r2/bench_c3.json
for w in c2 c5; do python bench.py --workload $w --steps 20 > gpurun_out/r2/bench_$w.json 2>/dev/null; cut -c1-700 gpurun_out/r2/bench_$w.json; done
python -m pytest tests/test_gpu_distributed.py -x -q -m gpu 2>&1 | tail -3
EKS_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 5 --warmup 2 --scaling strong --no-cpu-baseline 2>/dev/null | cut -c1-900
