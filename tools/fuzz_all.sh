O=${1:-gpurun_out/r04_fuzz.txt}; : > $O
echo "## fuzz_parity 120 cases seed 411" >> $O; python tools/fuzz_parity.py 120 411 2>&1 | tail -2 >> $O
echo "## fuzz_parity2 40 cases seed 402" >> $O; python tools/fuzz_parity2.py 40 402 2>&1 | tail -2 >> $O
echo "## fuzz_drivers 30 cases seed 403" >> $O; python tools/fuzz_drivers.py 30 403 2>&1 | tail -2 >> $O
echo "## fuzz_adam 20 cases seed 404" >> $O; python tools/fuzz_adam.py 20 404 2>&1 | tail -2 >> $O
echo "## fuzz_pupil 12 cases seed 405" >> $O; python tools/fuzz_pupil.py 12 405 2>&1 | tail -2 >> $O
echo "## fuzz_median seed 406, 120 cases" >> $O; python tools/fuzz_median.py 406 120 2>&1 | tail -1 >> $O
echo "## fuzz_ekf 30 rigs seed 407" >> $O; python tools/fuzz_ekf.py 30 407 2>&1 | tail -2 >> $O
cat $O
