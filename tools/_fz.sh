for seed in 1234 777 31337 2025; do echo "## fuzz_parity 160 cases seed $seed"; FUZZ_DETAIL=1 python tools/fuzz_parity.py 160 $seed 2>&1 | grep -E "detail|^worst" | cut -c1-300; done
for seed in 41 42 43; do echo "## fuzz_adam 25 cases seed $seed"; python tools/fuzz_adam.py 25 $seed 2>&1 | tail -1; done
echo "## fuzz_parity2 60 cases seed 99"; python tools/fuzz_parity2.py 60 99 2>&1 | tail -1
