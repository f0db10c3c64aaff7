python tools/fuzz_parity.py 150 2222 2>&1 | grep -E "above|nll [2-9]\.[0-9]e-05|nll 1\.[0-9]e-05" | head
echo "--- sequential assemble"
EKS_NLL_ASSEMBLE_SEQ=1 python tools/fuzz_parity.py 150 2222 2>&1 | grep -E "above|nll [2-9]\.[0-9]e-05|nll 1\.[0-9]e-05|worst" | head
