#!/bin/bash
# wave life times of k_dfs on the headline targets (5 M-key table): speculation on / off.  tools/lifetimes_ab.sh <tag> [tests-k-expr]
set -o pipefail
tag=${1:-life}
out=gpurun_out/$tag
mkdir -p $out
python3 -c "import __graft_entry__ as g; g.build()" || exit 1
if [ -n "$2" ]; then
  timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$2" > $out/tests.log 2>&1
  rc=$?; tail -5 $out/tests.log; [ $rc -ne 0 ] && exit $rc
fi
echo "== KM_SPECULATE=1"; timeout -k 10 300 python3 tools/dfs_lifetimes.py > $out/life1.txt 2>&1 || { tail $out/life1.txt; exit 1; }
grep -v amdgpu.ids $out/life1.txt
if [ -n "$DFS_COUNTERS" ]; then exit 0; fi
echo "== KM_SPECULATE=0"; KM_SPECULATE=0 timeout -k 10 300 python3 tools/dfs_lifetimes.py > $out/life0.txt 2>&1 || { tail $out/life0.txt; exit 1; }
grep -v amdgpu.ids $out/life0.txt
