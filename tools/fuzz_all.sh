O=gpurun_out/r03_fuzz.txt; : > $O
echo "## fuzz_parity 60 cases seed 301" >> $O; python tools/fuzz_parity.py 60 301 2>&1 | tail -2 >> $O
echo "## fuzz_parity2 40 cases seed 302" >> $O; python tools/fuzz_parity2.py 40 302 2>&1 | tail -2 >> $O
echo "## fuzz_drivers 30 cases seed 303" >> $O; python tools/fuzz_drivers.py 30 303 2>&1 | tail -2 >> $O
echo "## fuzz_adam 20 cases seed 304" >> $O; python tools/fuzz_adam.py 20 304 2>&1 | tail -2 >> $O
echo "## fuzz_pupil 12 cases seed 305" >> $O; python tools/fuzz_pupil.py 12 305 2>&1 | tail -2 >> $O
echo "## fuzz_median seed 306, 120 cases" >> $O; python tools/fuzz_median.py 306 120 2>&1 | tail -1 >> $O
echo "## fuzz_ekf 30 rigs seed 307" >> $O; python tools/fuzz_ekf.py 30 307 2>&1 | tail -2 >> $O
cat $O
