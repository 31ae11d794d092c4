#!/bin/bash
# usage: tools/sweep.sh "<VAR=val ...>" ...   (one bench run per argument; results in gpurun_out/sweep/)
mkdir -p gpurun_out/sweep
n=0
for cfg in "$@"; do
  n=$((n+1))
  echo "== $cfg" >> gpurun_out/sweep/log.txt
  env $cfg timeout -k 10 240 python bench.py --no-cpu --e2e 0 --steps 10 --cache /tmp/kmc $SWEEP_ARGS > gpurun_out/sweep/r$n.json 2>> gpurun_out/sweep/err.txt || exit 1
  python - "$cfg" gpurun_out/sweep/r$n.json >> gpurun_out/sweep/log.txt <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "| value %.2fM  step %.4f  unpip %.4f  kernels %s  build %.2fs" % (
    d["value"] / 1e6, d["ms_per_step"], d["ms_per_step_unpipelined"],
    {k: round(v, 4) for k, v in d["kernel_ms"].items()}, d["setup_s"]["table_build"]))
PY
done
cat gpurun_out/sweep/log.txt
