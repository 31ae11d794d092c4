#!/bin/bash
# parity subset + the hard workload alone.  tools/r4_hard.sh <tag> [tests-k-expr]
set -o pipefail
tag=${1:-hard}
out=gpurun_out/$tag
mkdir -p $out
python3 -c "import __graft_entry__ as g; g.build()" || exit 1
if [ -n "$2" ]; then
  timeout -k 10 1000 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$2" > $out/tests.log 2>&1
  rc=$?; tail -15 $out/tests.log; [ $rc -ne 0 ] && exit $rc
fi
timeout -k 10 300 python3 tools/hard_only.py > $out/hard.txt 2>&1 || { tail $out/hard.txt; exit 1; }
grep -v amdgpu.ids $out/hard.txt
