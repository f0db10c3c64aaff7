#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): tests, bench lines, rocprofv3 kernel stats and PMC passes for the
# C3 workload.  Everything lands under gpurun_out/evidence/; tools/make_profiles.py (run afterwards in
# the build container) turns it into profiles/<tag>_*.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/evidence
rm -rf $O; mkdir -p $O
cd $R
python -c "import bench; print(bench.kernel_sources_sha16())" > $O/kernel_sources_sha16.txt
python -m pytest tests -q -m gpu --durations=15 2>&1 | tail -24 > $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
# the driver's command (headline + the short legs of the other configurations under `extras`), then the headline alone
python bench.py > $O/bench_c3_default_run.json 2> $O/bench_c3_default_run.err
python bench.py --no-extras > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --workload c3adam > $O/bench_c3adam.json 2> $O/bench_c3adam.err
# the same search by the kernels that read y every iteration (one launch per iteration: round 5's cooperative loop is gone),
# and with every chain streamed by its own wave inside the lag kernel (the exact fallback for poles outside the lags' range)
EKS_ADAM_STREAM=1 python bench.py --workload c3adam --no-cpu-baseline > $O/bench_c3adam_streaming.json 2>/dev/null
EKS_ADAM_LAG_RHO_PPM=0 python bench.py --workload c3adam --steps 3 --warmup 1 --regions 2 --no-cpu-baseline > $O/bench_c3adam_all_chains_streamed.json 2>/dev/null
python tools/lag_adam_check.py > $O/lag_adam_check.txt 2>&1
python tools/lag_prepass_time.py 2>&1 | grep -v amdgpu > $O/lag_prepass_time.txt
python tools/lag_adam_shapes.py 2>&1 | grep -v amdgpu > $O/lag_adam_shapes.txt
for m in side after_const_r first; do EKS_ADAM_PREPARE=$m python bench.py --workload c3adam --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('EKS_ADAM_PREPARE=$m', d['ms_per_step'])"; done > $O/c3adam_prepare_order.txt
for c in 8 16 32 64 128; do echo "== EKS_DENSE_CHUNK=$c"; EKS_DENSE_CHUNK=$c python tools/ekf_time.py 2>&1 | grep "cold start\|run_kalman"; done > $O/ekf_chunk_trade.txt
EKS_NLL_LEGACY=1 python bench.py --no-cpu-baseline > $O/bench_c3_legacy_nll.json 2>/dev/null
EKS_NLL_NOLAG=1 python bench.py --no-cpu-baseline > $O/bench_c3_nolag.json 2>/dev/null
python bench.py --workload c4 --no-cpu-baseline > $O/bench_c4.json 2>/dev/null
python bench.py --workload c4w --no-cpu-baseline > $O/bench_c4w.json 2>/dev/null
python bench.py --workload c4adam --no-cpu-baseline > $O/bench_c4adam.json 2>/dev/null
# the N > 1 launch path as the round-end driver invokes it (no outer torchrun); two ranks share this box's GPU
EKS_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload c3 --steps 10 --no-cpu-baseline > $O/bench_c3_2ranks_gloo.json 2>/dev/null
EKS_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload c3 --steps 10 --no-cpu-baseline --scaling strong > $O/bench_c3_2ranks_gloo_strong.json 2>/dev/null
python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2>/dev/null
python bench.py --workload c2 --no-cpu-baseline > $O/bench_c2.json 2>/dev/null
python bench.py --workload pupil > $O/bench_pupil.json 2>/dev/null
python bench.py --workload ekf > $O/bench_ekf.json 2>/dev/null
python tools/ekf_time.py > $O/ekf_time.txt 2>&1
python tools/driver_time.py 2>&1 | grep -E " ms" > $O/driver_time.txt
python tools/host_path_time.py 2>&1 | grep -v amdgpu > $O/host_path_time.txt
for m in diag dense; do python tools/first_call.py $m 2>&1 | grep -v amdgpu; done > $O/first_call.txt
python tools/fit_time.py 2>&1 | grep -v amdgpu > $O/fit_time.txt
python tools/host_boundary_ab.py 2>&1 | grep -v amdgpu > $O/host_boundary_ab.txt
EKS_HIP_LIB=build_alt/gridstamps/libeks_hip.so python tools/grid_stamps.py 2>&1 | grep -v amdgpu > $O/grid_stamps.txt
python tools/adam_time.py > $O/adam_time.txt 2>&1
python tools/dense_adam_time.py > $O/dense_adam_time.txt 2>&1
python tools/dense_adam_time_d.py 2>&1 | grep adam > $O/dense_adam_time_d.txt
python tools/pupil_time.py 2>&1 | grep -v amdgpu > $O/pupil_time.txt
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-events --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3adam -- python3 $R/bench.py --workload c3adam --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
B3="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-extras"
# device timeline of one c3adam step (idle gaps in front of every kernel)
rocprofv3 --kernel-trace --output-format csv -d $O/trace_c3adam -- python3 $R/bench.py --workload c3adam --steps 3 --warmup 2 --regions 1 --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
python3 $R/tools/trace_gaps.py $(find $O/trace_c3adam -name "*_kernel_trace.csv" | head -1) 22 > $O/c3adam_timeline.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- $B3 > /dev/null 2>&1
# keep only the summaries (the traces are large)
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*_agent_info.csv" -delete
ls -R $O | head -40
