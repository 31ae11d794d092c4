#!/bin/bash
# bench.py --only-step for several builds of the library (km_amd/variants/*.so)
set -o pipefail
tag=${1:-bv}
out=gpurun_out/$tag
mkdir -p $out
python3 -c "import __graft_entry__ as g; g.build()" || exit 1
for so in km_amd/variants/*.so; do
  n=$(basename $so .so)
  KM_LIBRARY=$PWD/$so timeout -k 10 500 python3 bench.py --no-cpu --only-step --steps 20 --repeats 3 --cache /tmp/kmc > $out/bench_$n.json 2> $out/bench_$n.err || { tail -5 $out/bench_$n.err; exit 1; }
  python3 - $out/bench_$n.json $n <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h = j.get("config4_hard") or {}
print(sys.argv[2], "value %.1f M" % (j["value"] / 1e6), "kernel_only %.4f" % j["kernel_only"]["ms_per_step"],
      "kernel_ms", {k: round(v, 4) for k, v in j["kernel_ms"].items() if isinstance(v, float)}, "frac %.3f" % j["roofline"]["frac"], "ok", j["oracle_check"]["ok"])
PY
  KM_LIBRARY=$PWD/$so timeout -k 10 300 python3 tools/hard_only.py 2>&1 | grep -v amdgpu.ids | tail -2
done
