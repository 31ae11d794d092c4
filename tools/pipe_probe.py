#!/usr/bin/env python3
"""Diagnostics: the delivered pipeline without torch — per-batch HIP-event timings as they
come out INSIDE the pipelined region (several workspaces in flight), for different numbers of
workspaces.  usage: pipe_probe.py [n_keys] [inflight,...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from km_amd import kmer as km, lib as kmlib, synth  # noqa: E402

n_keys = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
fls = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4").split(",")]
T, L, K = 10000, 500, 31
nset = max(fls)
case = synth.make_case(n_targets=T * nset, length=L, k=K, n_keys=n_keys, seed=5, exact_pad=False)
db = kmlib.Database.from_records(case["keys"], case["counts"], K).upload(0)
blob = np.frombuffer(b"ACGT", dtype=np.uint8)[case["targets"]].copy()
offs = np.arange(T + 1, dtype=np.uint64) * np.uint64(L)
both = kmlib.KM_STAGE_WALK | kmlib.KM_STAGE_GRAPH | (kmlib.KM_RUN_TIMED if os.environ.get("PROBE_TIMED") else 0)
for n_fl in fls:
    batches, streams = [], []
    for q in range(n_fl):
        b = kmlib.Batch(db, max_targets=T, max_total_bases=T * L)
        b.set_targets_packed(blob[q * T:(q + 1) * T].reshape(-1), offs)
        batches.append(b)
        streams.append(kmlib.stream_create(0))
    lean = both | kmlib.KM_RUN_DELIVER | kmlib.KM_DELIVER_LEAN
    for flags, name in ((both | kmlib.KM_RUN_DELIVER, "deliver"), (lean, "lean"), (lean | kmlib.KM_RUN_HIPGRAPH, "lean+graph"),
                        (both, "kernels only"), (both | kmlib.KM_RUN_HIPGRAPH, "kernels+graph"),
                        (kmlib.KM_STAGE_GRAPH, "graph stage only"), (kmlib.KM_STAGE_WALK, "walk stage only")):
        wait = bool(flags & kmlib.KM_RUN_DELIVER)
        for rep in range(2):
            steps = 40
            t0 = time.perf_counter()
            host = 0.0
            for i in range(steps):
                q = i % n_fl
                if wait and i >= n_fl:
                    batches[q].wait_result()
                h0 = time.perf_counter()
                batches[q].run(flags, streams[q])
                host += time.perf_counter() - h0
            for b in batches:
                if wait:
                    b.wait_result()
                else:
                    b.timings()
            dt = (time.perf_counter() - t0) / steps * 1e3
        tm = np.array([b.timings() for b in batches])
        print("inflight %d  %-13s  %.3f ms/step (host time in run(): %.3f ms/step)   last-run events per batch (ms): walk %s graph %s outk %s d2h %s"
              % (n_fl, name, dt, host / steps * 1e3, np.round(tm[:, 0], 3), np.round(tm[:, 1], 3), np.round(tm[:, 6], 3),
                 np.round(tm[:, 7], 3)), flush=True)
    for b in batches:
        b.close()
    for s in streams:
        kmlib.stream_destroy(s)
db.close()
