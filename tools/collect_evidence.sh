#!/bin/bash
# Round evidence on the GPU box: kernel trace + stats (every kernel alone, ROTATING over four distinct target sets:
# no launch replays the set of the launch before it out of the 256 MiB Infinity Cache), PMC passes (separate runs, as
# the guide prescribes), the same three for the second workload (config4_hard), a kernel + memory-copy timeline of the
# PIPELINED step (what `value` measures), the default bench line.  Results under gpurun_out/$1/ ; copy what is to be
# judged into profiles/.
set -o pipefail
tag=${1:-r04}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -c "import __graft_entry__ as g; g.build()" || exit 1
# --serial: each kernel alone on the GPU (KM_RUN_SERIAL); --inflight 4 --one-at-a-time: four batches with their own
# target sets, run strictly one after the other — what bench.py's own HIP events (roofline.frac) time
B="python3 bench.py --steps 20 --warmup 4 --no-cpu --only-step --check 0 --inflight 4 --one-at-a-time --serial --repeats 1 --cache /tmp/kmc"
H="python3 bench.py --steps 16 --warmup 4 --no-cpu --profile-hard --inflight 2 --cache /tmp/kmc"
python3 bench.py --dump-case /tmp/kmc_client --cache /tmp/kmc || exit 1
P="km_amd/kmclient pump /tmp/kmc_client 40 8 4 3"
echo "[1/9] kernel trace, every kernel alone, rotating sets"; rocprofv3 --kernel-trace --stats -d $out/trace --output-format csv -- $B > $out/trace.json 2> $out/trace.err || exit 1
find $out/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
echo "[2/9] kernel + memory-copy timeline of the pipelined step (4 in flight), C++ client"; rocprofv3 --kernel-trace --memory-copy-trace -d $out/timeline --output-format csv -- $P > $out/timeline.json 2> $out/timeline.err || exit 1
python3 tools/timeline_summary.py $out/timeline $out/timeline_summary.json || exit 1
rm -rf $out/timeline
echo "[2b] the same through the Python process (does it still end in __cxa_finalize?)"
if rocprofv3 --kernel-trace --memory-copy-trace -d $out/timeline_py --output-format csv -- python3 bench.py --steps 20 --warmup 4 --no-cpu --timeline --repeats 1 --cache /tmp/kmc > $out/timeline_py.json 2> $out/timeline_py.err; then
  echo "python timeline: completed, $(find $out/timeline_py -name '*kernel_trace.csv' | wc -l) kernel trace file(s)" | tee $out/timeline_py_outcome.txt
else
  echo "python timeline: FAILED rc=$? ; tail of stderr:" | tee $out/timeline_py_outcome.txt; tail -5 $out/timeline_py.err | tee -a $out/timeline_py_outcome.txt
fi
rm -rf $out/timeline_py
echo "[3/9] pmc reads"; rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_32B_sum -d $out/pmc1 --output-format csv -- $B > $out/pmc1.json 2> $out/pmc1.err || exit 1
echo "[4/9] pmc writes / L2"; rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $out/pmc2 --output-format csv -- $B > $out/pmc2.json 2> $out/pmc2.err || exit 1
echo "[5/9] pmc fetch size"; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc3 --output-format csv -- $B > $out/pmc3.json 2> $out/pmc3.err || exit 1
echo "[6/9] pmc SQ"; rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY -d $out/pmc4 --output-format csv -- $B > $out/pmc4.json 2> $out/pmc4.err || exit 1
python3 tools/pmc_summary.py $out $out/pmc_summary.json $tag
rm -rf $out/pmc1 $out/pmc2 $out/pmc3 $out/pmc4 $out/trace
echo "[7/9] config4_hard: kernel trace, every kernel alone, rotating over its two sets"; rocprofv3 --kernel-trace --stats -d $out/htrace --output-format csv -- $H > $out/hard_trace.json 2> $out/hard_trace.err || exit 1
find $out/htrace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/hard_kernel_stats.csv
rm -rf $out/htrace
echo "[8/9] config4_hard: pmc reads + SQ"; rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_32B_sum -d $out/pmc1 --output-format csv -- $H > $out/hpmc1.json 2> $out/hpmc1.err || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $out/pmc2 --output-format csv -- $H > $out/hpmc2.json 2> $out/hpmc2.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY -d $out/pmc4 --output-format csv -- $H > $out/hpmc4.json 2> $out/hpmc4.err || exit 1
mkdir -p $out/pmc3
python3 tools/pmc_summary.py $out $out/hard_pmc_summary.json ${tag}_hard
rm -rf $out/pmc1 $out/pmc2 $out/pmc3 $out/pmc4
echo "[9/9] default bench"; python3 bench.py --cache /tmp/kmc > $out/bench_N1.json 2> $out/bench_N1.err || exit 1
tail -c 400 $out/bench_N1.json
