for m in side after_const_r; do
EKS_ADAM_PREPARE=$m python bench.py --workload c3adam --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3adam $m', d['ms_per_step'], d['ms_per_step_min'], d['roofline']['frac'], d['roofline']['kernel_avg_ms'])"
done
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "adam" -x 2>&1 | tail -3
python -m pytest tests/test_gpu_configs.py tests/test_gpu_drivers.py tests/test_gpu_nan.py -q -m gpu -x 2>&1 | tail -3
