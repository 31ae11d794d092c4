#!/usr/bin/env python3
"""Diagnostics: duration of k_graph when it stops after stage N (KM_DEBUG_FLAGS ablation; results
of such runs are invalid, only the timing matters).  usage: graph_stages.py [n_keys]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag  # noqa: E402

_diag.build()                       # KM_DEBUG_FLAGS exists in the diagnostics build only
from km_amd import lib as kmlib, synth  # noqa: E402

n_keys = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
T, L, K = 10000, 500, 31
case = synth.make_case(n_targets=T, length=L, k=K, n_keys=n_keys, seed=5, exact_pad=False)
db = kmlib.Database.from_records(case["keys"], case["counts"], K).upload(0)
blob = np.frombuffer(b"ACGT", dtype=np.uint8)[case["targets"]].copy().reshape(-1)
offs = np.arange(T + 1, dtype=np.uint64) * np.uint64(L)
b = kmlib.Batch(db, max_targets=T, max_total_bases=T * L)
b.set_targets_packed(blob, offs)
st = kmlib.stream_create(0)
flags = kmlib.KM_STAGE_WALK | kmlib.KM_STAGE_GRAPH | kmlib.KM_RUN_TIMED
names = {0: "full", 8: "0 launch + headers", 9: "1a LDS fill", 1: "1 prefix table + dup check", 2: "2 adjacency", 3: "2b links + dist init", 4: "3 dijkstra",
         5: "4 prev arrays", 6: "5 strip ref edges", 7: "6 candidates"}
base = int(os.environ.get("KM_BASE_FLAGS", "0"), 0)      # walk-kernel ablation bits (low byte)
stages = (0, 8, 9, 1, 2, 3, 4, 5, 6, 7) if not os.environ.get("ONLY_FULL") else (0,)
for n in stages:
    os.environ["KM_DEBUG_FLAGS"] = hex((((0x80 | n) << 8) if n else 0) | base)
    tm = []
    for _ in range(12):
        b.run(flags, st)
        tm.append(b.timings())
    tm = np.array(tm)[2:].mean(axis=0)
    print("stop after %-28s graph stage %.1f us   (walk %.1f, k_dfs %.1f)" % (names[n], tm[1] * 1e3, tm[0] * 1e3, tm[5] * 1e3), flush=True)
