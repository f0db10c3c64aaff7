cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/fuzz_median.py 511 200 2>&1 | tail -1
python $R/tools/med_rows_sweep.py 2>&1 | grep MED
EKS_MED_SAMPLE_LEGACY=1 python $R/tools/med_rows_sweep.py 2>&1 | grep MED | sed 's/^/legacy sample: /'
for a in 0; do
EKS_MED_SAMPLE_LEGACY=$a rm -rf $R/gpurun_out/sbx_$a; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sbx_$a -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/sb_$a.json 2>/dev/null
done
find $R/gpurun_out -name "*_kernel_trace.csv" -size +5M -delete
