#!/bin/bash
# wave life times of k_dfs for several builds of the library (km_amd/variants/*.so).  tools/variants_lifetimes.sh <tag> [tests-k-expr]
set -o pipefail
tag=${1:-var}
out=gpurun_out/$tag
mkdir -p $out
python3 -c "import __graft_entry__ as g; g.build()" || exit 1
if [ -n "$2" ]; then
  timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$2" > $out/tests.log 2>&1
  rc=$?; tail -5 $out/tests.log; [ $rc -ne 0 ] && exit $rc
fi
for so in km_amd/variants/*.so; do
  n=$(basename $so .so)
  echo "== $n"
  KM_LIBRARY=$PWD/$so timeout -k 10 300 python3 tools/dfs_lifetimes.py > $out/life_$n.txt 2>&1 || { tail $out/life_$n.txt; exit 1; }
  grep -v "amdgpu.ids\|^  target" $out/life_$n.txt
done
