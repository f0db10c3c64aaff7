#!/bin/bash
# Diagnostic / A-B builds of libeks_hip.so beside the shipped one: recompiles the listed sources with
# extra flags and links them with the shipped objects.   tools/build_alt.sh NAME "FLAGS" file1.hip [file2.hip ...]
# -> build_alt/NAME/libeks_hip.so (load it with EKS_HIP_LIB=...)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2; shift 2
OUT=$R/build_alt/$NAME; mkdir -p $OUT
cp $R/eks_amd/lib/*.o $OUT/
for f in "$@"; do
  extra=""; case $f in eks_diag_nll.hip|eks_diag.hip|eks_lag_adam.hip) extra="-fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -Wno-unused-result -I $R/eks_amd/csrc $extra $FLAGS -c $R/eks_amd/csrc/$f -o $OUT/${f%.hip}.o
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OUT/*.o -o $OUT/libeks_hip.so
echo $OUT/libeks_hip.so
