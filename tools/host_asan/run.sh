#!/bin/bash
# Host code of libeks_hip.so under AddressSanitizer + UBSan (CPU only): tools/host_asan/run.sh [work dir]
set -e
H=$(cd "$(dirname "$0")" && pwd); W=${1:-/tmp/eks_host_asan}; mkdir -p $W
# (exit 77: no compiler with the sanitizer runtimes on this machine)
command -v g++ > /dev/null || exit 77
echo 'int main(){return 0;}' | g++ -fsanitize=address,undefined -x c++ - -o $W/probe 2> /dev/null || exit 77
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -x c++ $H/../../eks_amd/csrc/eks_host.hip $H/harness.cpp -o $W/harness -lpthread
python3 $H/make_corpus.py $W/corpus
ASAN_OPTIONS=detect_leaks=1 $W/harness $W/corpus
