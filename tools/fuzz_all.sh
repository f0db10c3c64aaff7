# Runs ON THE GPU BOX: the seven parity fuzzers (oracles as the checker).  usage: tools/fuzz_all.sh [output file] [seed base]
O=${1:-gpurun_out/r04_fuzz.txt}; B=${2:-400}; : > $O
echo "## fuzz_parity 120 cases seed $((B+11))" >> $O; FUZZ_DETAIL=1 python tools/fuzz_parity.py 120 $((B+11)) 2>&1 | grep -E "detail|^worst" | cut -c1-420 >> $O
echo "## fuzz_parity2 40 cases seed $((B+2))" >> $O; python tools/fuzz_parity2.py 40 $((B+2)) 2>&1 | tail -1 >> $O
echo "## fuzz_drivers 30 cases seed $((B+3))" >> $O; python tools/fuzz_drivers.py 30 $((B+3)) 2>&1 | tail -1 >> $O
echo "## fuzz_adam 20 cases seed $((B+4))" >> $O; python tools/fuzz_adam.py 20 $((B+4)) 2>&1 | tail -1 >> $O
echo "## fuzz_pupil 12 cases seed $((B+5))" >> $O; python tools/fuzz_pupil.py 12 $((B+5)) 2>&1 | tail -1 >> $O
echo "## fuzz_median seed $((B+6)), 200 cases" >> $O; python tools/fuzz_median.py $((B+6)) 200 2>&1 | tail -1 >> $O
echo "## fuzz_ekf 30 rigs seed $((B+7))" >> $O; python tools/fuzz_ekf.py 30 $((B+7)) 2>&1 | tail -1 >> $O
cat $O
