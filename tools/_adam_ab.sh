one() { timeout 300 python bench.py --workload c3adam --no-cpu-baseline 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), d['config']['adam_iterations'], {k:round(v,3) for k,v in d['roofline']['other_stage_ms'].items()}, d['roofline']['search_kernel_ms'])" || tail -5 /tmp/err.txt; }
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "nll_grad or adam" 2>&1 | tail -4
one default; EKS_ADAM_PER_ITERATION=1 one periter; one default; EKS_ADAM_PER_ITERATION=1 one periter
