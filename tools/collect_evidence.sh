#!/bin/bash
# Round evidence on the GPU box: kernel trace + stats, PMC passes (separate runs, as the
# guide prescribes), the default bench line.  Results under gpurun_out/$1/ ; copy what is to
# be judged into profiles/.
set -o pipefail
out=gpurun_out/${1:-r02}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -c "import __graft_entry__ as g; g.build()" || exit 1
# --serial: each kernel alone on the GPU (KM_RUN_SERIAL), so that a kernel's duration in the trace is
# its own; with k_graph_pure beside k_dfs the profiler stretches both (147 / 97 us against 97 / - by
# HIP events without it).  The PMC passes serialise the kernels anyway.
B="python3 bench.py --steps 20 --warmup 4 --no-cpu --only-step --check 0 --inflight 1 --serial --cache /tmp/kmc"
echo "[1/6] kernel trace"; rocprofv3 --kernel-trace --stats -d $out/trace --output-format csv -- $B > $out/trace.json 2> $out/trace.err || exit 1
echo "[2/6] pmc reads"; rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_32B_sum -d $out/pmc1 --output-format csv -- $B > $out/pmc1.json 2> $out/pmc1.err || exit 1
echo "[3/6] pmc writes / L2"; rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $out/pmc2 --output-format csv -- $B > $out/pmc2.json 2> $out/pmc2.err || exit 1
echo "[4/6] pmc fetch size"; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc3 --output-format csv -- $B > $out/pmc3.json 2> $out/pmc3.err || exit 1
echo "[5/6] pmc SQ"; rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY -d $out/pmc4 --output-format csv -- $B > $out/pmc4.json 2> $out/pmc4.err || exit 1
python3 tools/pmc_summary.py $out $out/pmc_summary.json
cp $out/pmc_summary.json profiles/pmc_traffic.json
echo "[6/6] default bench"; python3 bench.py --cache /tmp/kmc > $out/bench_N1.json 2> $out/bench_N1.err || exit 1
find $out -name "*kernel_stats.csv" | head -3
tail -c 600 $out/bench_N1.json
