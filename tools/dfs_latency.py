#!/usr/bin/env python3
"""Diagnostics: is k_dfs bound by the latency of its table lookups?  The same 500 targets
against a table that fits the Infinity Cache (their own k-mers only) and against one that does
not (padded to n_keys).  usage: dfs_latency.py [n_keys_big]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from km_amd import lib as kmlib, synth  # noqa: E402

T, L, K = 500, 500, 31
big = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
flags = kmlib.KM_STAGE_WALK | kmlib.KM_STAGE_GRAPH | kmlib.KM_RUN_TIMED
for n_keys in (1, big):
    case = synth.make_case(n_targets=T, length=L, k=K, n_keys=n_keys, seed=7, exact_pad=False)
    db = kmlib.Database.from_records(case["keys"], case["counts"], K).upload(0)
    b = kmlib.Batch(db, max_targets=T, max_total_bases=T * L)
    blob = np.frombuffer(b"ACGT", dtype=np.uint8)[case["targets"]].copy().reshape(-1)
    b.set_targets_packed(blob, np.arange(T + 1, dtype=np.uint64) * np.uint64(L))
    st = kmlib.stream_create(0)
    tm = []
    for _ in range(30):
        b.run(flags, st)
        tm.append(b.timings())
    tm = np.array(tm)[5:].mean(axis=0)
    print("table of %9d keys (%6.1f MB): k_seed %.1f us, k_dfs %.1f us, graph %.1f us" %
          (len(case["keys"]), db.info.table_bytes / 1e6, tm[3] * 1e3, tm[5] * 1e3, tm[1] * 1e3), flush=True)
    b.close()
    db.close()
