#!/usr/bin/env python3
"""Diagnostics: the walk + graph kernels of one 10 000-target batch in a loop (a target for
`rocprofv3 --pc-sampling-*` / `--pmc`).  usage: dfs_loop.py [rounds] [n_keys] [n_targets]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from km_amd import lib as kmlib, synth  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 50
NK = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
T = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
L, K = 500, 31
case = synth.make_case(n_targets=T, length=L, k=K, n_keys=NK, seed=synth.HEADLINE_SEED, exact_pad=False)
db = kmlib.Database.from_records(case["keys"], case["counts"], K).upload(0)
b = kmlib.Batch(db, max_targets=T, max_total_bases=T * L)
blob = np.frombuffer(b"ACGT", dtype=np.uint8)[case["targets"]].copy().reshape(-1)
b.set_targets_packed(blob, np.arange(T + 1, dtype=np.uint64) * np.uint64(L))
st = kmlib.stream_create(0)
flags = kmlib.KM_STAGE_WALK | kmlib.KM_STAGE_GRAPH | kmlib.KM_RUN_TIMED
for name, fl in (("walk + graph", flags), ("walk only", kmlib.KM_STAGE_WALK | kmlib.KM_RUN_TIMED)):
    tm = []
    for _ in range(R):
        b.run(fl, st)
        b.sync()
        tm.append(b.timings())
    tm = np.array(tm)[min(5, R - 1):].mean(axis=0)
    print("%-12s: k_seed %.1f us, k_dfs %.1f us, graph %.1f us" % (name, tm[3] * 1e3, tm[5] * 1e3, tm[1] * 1e3),
          flush=True)
# wall-clock per run (one batch at a time), with and without the event records of KM_RUN_TIMED
import time  # noqa: E402

b2 = kmlib.Batch(db2 := kmlib.Database.from_records(case["keys"], case["counts"], K).upload(0), max_targets=T, max_total_bases=T * L)
b2.set_targets_packed(blob, np.arange(T + 1, dtype=np.uint64) * np.uint64(L))
for name, fl in (("untimed", kmlib.KM_STAGE_WALK | kmlib.KM_STAGE_GRAPH),
                 ("timed  ", kmlib.KM_STAGE_WALK | kmlib.KM_STAGE_GRAPH | kmlib.KM_RUN_TIMED),
                 ("untimed", kmlib.KM_STAGE_WALK | kmlib.KM_STAGE_GRAPH)):
    for _ in range(5):
        b2.run(fl, st)
        b2.sync()
    t0 = time.perf_counter()
    for _ in range(R):
        b2.run(fl, st)
        b2.sync()
    print("%s: %.1f us per run (host clock, sync after every run)" % (name, (time.perf_counter() - t0) / R * 1e6), flush=True)
b2.close()
db2.close()
b.close()
db.close()
