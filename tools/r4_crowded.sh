#!/bin/bash
set -o pipefail
out=gpurun_out/${1:-cr}
mkdir -p $out
python3 -c "import __graft_entry__ as g; g.build()" || exit 1
for v in 0 1; do
  echo "== KM_TABLE_LEAN_CROWDED=$v"
  KM_TABLE_LEAN_CROWDED=$v KM_BUILD_VERBOSE=1 DFS_COUNTERS=1 timeout -k 10 300 python3 tools/dfs_lifetimes.py > $out/life_$v.txt 2>&1 || { tail $out/life_$v.txt; exit 1; }
  grep "k_dfs\|lifetime\|non-resident\|table built\|45- 69\|20- 31" $out/life_$v.txt
done
