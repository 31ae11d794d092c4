#!/bin/bash
# bench.py --only-step for several directory sizes (KM_DIR_LOG2): does a directory that fits the Infinity Cache pay?
out=gpurun_out/${1:-dirsweep}; mkdir -p $out
python3 -c "import __graft_entry__ as g; g.build()" || exit 1
for lg in ${DIR_LOGS:-27 26 25 24}; do
  KM_DIR_LOG2=$lg timeout -k 10 500 python3 bench.py --no-cpu --only-step --steps 40 --repeats 3 --no-hard --cache /tmp/kmc > $out/bench_$lg.json 2> $out/bench_$lg.err || { tail -3 $out/bench_$lg.err; continue; }
  python3 - $out/bench_$lg.json $lg <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("dir 2^%s" % sys.argv[2], "value %.1f M" % (j["value"] / 1e6), "B/kmer %.1f" % j["config"]["table_bytes_per_kmer"], "max_probe", j["config"]["table_max_probe"],
      "kernel_ms", {k: round(v, 4) for k, v in j["kernel_ms"].items() if isinstance(v, float)}, "frac %.3f" % j["roofline"]["frac"], "ok", j["oracle_check"]["ok"])
PY
done
