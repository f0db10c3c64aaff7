for v in 0 1; do echo "== EKS_ADAM_LAG_VALU=$v"; EKS_ADAM_LAG_VALU=$v python tools/lag_prepass_time.py 2>&1 | grep "whole\|lag_sums"; done
for l in mff4 mff64; do echo "== $l"; EKS_HIP_LIB=build_alt/$l/libeks_hip.so python tools/lag_prepass_time.py 2>&1 | grep "lag_sums"; done
python tools/lag_adam_check.py quick 2>&1 | grep -v amdgpu
python bench.py --workload c3adam --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['cpu_baseline']['parity'])"
