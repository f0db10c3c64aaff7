#!/bin/bash
# Runs ON THE GPU BOX: HBM-side traffic (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate passes) and kernel times of
# every kernel of the given bench workloads -> gpurun_out/pmc_wl/<workload>.txt.  Used to look for launches that
# move more than their algorithmic bytes (e.g. neighbouring blocks on different XCDs fetching pieces of the same lines).
# usage: tools/pmc_by_workload.sh c5 c4w c3adam ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_wl
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for wl in "$@"; do
  B="python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events"
  rm -rf /tmp/pw_*
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw_t -- $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pw_f -- $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw_w -- $B > /dev/null 2>&1
  python3 $R/tools/pmc_by_workload.py $wl /tmp/pw_t /tmp/pw_f /tmp/pw_w > $O/$wl.txt 2>&1
done
