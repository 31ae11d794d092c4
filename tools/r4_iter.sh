#!/bin/bash
# Round-4 iteration on the GPU box: selected parity tests, then the per-kernel times of the default
# workload with and without the speculation of k_dfs.  Usage: tools/r4_iter.sh <tag> [pytest -k expression]
set -o pipefail
tag=${1:-it}
kexpr=${2:-"chain_runs or randomized or epilogue or synthetic_slices or full_batch or single_bubble or large_tier or long_targets"}
out=gpurun_out/$tag
mkdir -p $out
python3 -c "import __graft_entry__ as g; g.build()" || exit 1
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$kexpr" > $out/tests.log 2>&1
rc=$?
tail -5 $out/tests.log
[ $rc -ne 0 ] && exit $rc
B="python3 bench.py --steps 20 --warmup 4 --no-cpu --only-step --check 200 --repeats 3 --cache /tmp/kmc"
timeout -k 10 600 $B > $out/bench_spec1.json 2> $out/bench_spec1.err || { tail -20 $out/bench_spec1.err; exit 1; }
KM_SPECULATE=0 timeout -k 10 600 $B > $out/bench_spec0.json 2> $out/bench_spec0.err || { tail -20 $out/bench_spec0.err; exit 1; }
python3 - $out <<'PY'
import json, sys
for f in ("bench_spec1.json", "bench_spec0.json"):
    j = json.loads(open(sys.argv[1] + "/" + f).read().strip().splitlines()[-1])
    print(f, "value %.1f M" % (j["value"] / 1e6), "ms/step %.4f" % j["ms_per_step"], "kernel_only %.4f" % j["kernel_only"]["ms_per_step"],
          "kernel_ms", {k: round(v, 4) for k, v in j["kernel_ms"].items() if isinstance(v, float)}, "frac %.3f" % j["roofline"]["frac"],
          "check", j["oracle_check"])
PY
