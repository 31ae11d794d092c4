#!/usr/bin/env python3
"""Diagnostics: is k_dfs bound by cold instruction fetches?  Times k_dfs as launched (cold), as the
second of two back-to-back launches (instruction cache and data warm), and as the second launch after
a 1 GiB memset (instruction cache warm, table lines cold again).  usage: dfs_replay.py [n_keys]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[2] == "child":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import _diag
    _diag.build()                   # KM_DFS_REPLAY exists in the diagnostics build only
    import numpy as np
    from km_amd import lib as kmlib, synth
    T, L, K = 10000, 500, 31
    nk = int(sys.argv[1])
    case = synth.make_case(n_targets=T, length=L, k=K, n_keys=nk, seed=synth.HEADLINE_SEED, exact_pad=False)
    db = kmlib.Database.from_records(case["keys"], case["counts"], K).upload(0)
    b = kmlib.Batch(db, max_targets=T, max_total_bases=T * L)
    blob = np.frombuffer(b"ACGT", dtype=np.uint8)[case["targets"]].copy().reshape(-1)
    b.set_targets_packed(blob, np.arange(T + 1, dtype=np.uint64) * np.uint64(L))
    st = kmlib.stream_create(0)
    flags = kmlib.KM_STAGE_WALK | kmlib.KM_STAGE_GRAPH | kmlib.KM_RUN_TIMED | kmlib.KM_RUN_SERIAL
    tm = []
    for _ in range(30):
        b.run(flags, st)
        b.sync()
        tm.append(b.timings())
    tm = np.array(tm)[5:].mean(axis=0)
    print("KM_DFS_REPLAY=%s: k_dfs %.1f us (k_seed %.1f, graph %.1f)" %
          (os.environ.get("KM_DFS_REPLAY", "0"), tm[5] * 1e3, tm[3] * 1e3, tm[1] * 1e3), flush=True)
    b.close()
    db.close()
    sys.exit(0)
nk = sys.argv[1] if len(sys.argv) > 1 else "20000000"
for mode in ("0", "1", "2"):
    subprocess.call([sys.executable, os.path.abspath(__file__), nk, "child"], env=dict(os.environ, KM_DFS_REPLAY=mode))
