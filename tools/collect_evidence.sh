#!/bin/bash
# Round evidence on the GPU box: kernel trace + stats (every kernel alone), PMC passes (separate runs, as the
# guide prescribes), a kernel + memory-copy timeline of the PIPELINED step (what `value` measures), the default
# bench line.  Results under gpurun_out/$1/ ; copy what is to be judged into profiles/.
set -o pipefail
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -c "import __graft_entry__ as g; g.build()" || exit 1
# --serial: each kernel alone on the GPU (KM_RUN_SERIAL), so that a kernel's duration in the trace is
# its own; with k_graph_pure beside k_dfs the profiler stretches both.  The PMC passes serialise the kernels anyway.
B="python3 bench.py --steps 20 --warmup 4 --no-cpu --only-step --check 0 --inflight 1 --serial --repeats 1 --cache /tmp/kmc"
# the pipelined step as an interpreter-free C++ consumer of the C-ABI (tools/kmclient.cpp): rocprofv3's
# memory-copy tracing takes the Python / torch process down in __cxa_finalize at exit, before it has written anything
python3 bench.py --dump-case /tmp/kmc_client --cache /tmp/kmc || exit 1
P="km_amd/kmclient pump /tmp/kmc_client 40 8 4 3"
echo "[1/7] kernel trace, every kernel alone"; rocprofv3 --kernel-trace --stats -d $out/trace --output-format csv -- $B > $out/trace.json 2> $out/trace.err || exit 1
echo "[2/7] kernel + memory-copy timeline of the pipelined step (4 in flight)"; rocprofv3 --kernel-trace --memory-copy-trace -d $out/timeline --output-format csv -- $P > $out/timeline.json 2> $out/timeline.err || exit 1
python3 tools/timeline_summary.py $out/timeline $out/timeline_summary.json || exit 1
rm -rf $out/timeline                               # (hundreds of MB of rows; the summary is what is kept)
echo "[3/7] pmc reads"; rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_32B_sum -d $out/pmc1 --output-format csv -- $B > $out/pmc1.json 2> $out/pmc1.err || exit 1
echo "[4/7] pmc writes / L2"; rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $out/pmc2 --output-format csv -- $B > $out/pmc2.json 2> $out/pmc2.err || exit 1
echo "[5/7] pmc fetch size"; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc3 --output-format csv -- $B > $out/pmc3.json 2> $out/pmc3.err || exit 1
echo "[6/7] pmc SQ"; rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY -d $out/pmc4 --output-format csv -- $B > $out/pmc4.json 2> $out/pmc4.err || exit 1
python3 tools/pmc_summary.py $out $out/pmc_summary.json $tag
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/pmc1 $out/pmc2 $out/pmc3 $out/pmc4 $out/trace
echo "[7/7] default bench"; python3 bench.py --cache /tmp/kmc > $out/bench_N1.json 2> $out/bench_N1.err || exit 1
tail -c 400 $out/bench_N1.json
