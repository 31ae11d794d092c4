#!/usr/bin/env python3
"""Diagnostics: N steps of the headline-shaped batch on one stream, each kernel alone (KM_RUN_SERIAL) —
the program to put behind `rocprofv3 ... --` (kernel trace, PC sampling).  usage: step_loop.py [n_keys] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from km_amd import lib as kmlib, synth  # noqa: E402

T, L, K = 10000, 500, 31
nk = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
case = synth.make_case(n_targets=T, length=L, k=K, n_keys=nk, seed=synth.HEADLINE_SEED, exact_pad=False)
db = kmlib.Database.from_records(case["keys"], case["counts"], K).upload(0)
b = kmlib.Batch(db, max_targets=T, max_total_bases=T * L)
blob = np.frombuffer(b"ACGT", dtype=np.uint8)[case["targets"]].copy().reshape(-1)
b.set_targets_packed(blob, np.arange(T + 1, dtype=np.uint64) * np.uint64(L))
st = kmlib.stream_create(0)
flags = kmlib.KM_STAGE_WALK | kmlib.KM_STAGE_GRAPH | kmlib.KM_RUN_TIMED | kmlib.KM_RUN_SERIAL
tm = []
for _ in range(steps):
    b.run(flags, st)
    b.sync()
    tm.append(b.timings())
tm = np.array(tm)[min(5, steps - 1):].mean(axis=0)
print("k_pack %.1f k_seed %.1f k_dfs %.1f graph %.1f us" % (tm[4] * 1e3, tm[3] * 1e3, tm[5] * 1e3, tm[1] * 1e3), flush=True)
b.close()
db.close()
kmlib.stream_destroy(st)
