#!/usr/bin/env python3
"""Diagnostics: how many flagged targets does the epilogue of k_dfs answer?  Headline-shaped batch and a
multi-variant one.  usage: epi_stats.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from km_amd import lib as kmlib, synth  # noqa: E402

for label, kw in (("headline shape", dict(seed=synth.HEADLINE_SEED)),
                  ("1-3 variants, branch noise", dict(seed=77, variant_frac=0.7, variants_per_target=(1, 3), hom_frac=0.2,
                                                     branch_noise_frac=0.03, noise_frac=0.03))):
    T, L, K = 10000, 500, 31
    case = synth.make_case(n_targets=T, length=L, k=K, n_keys=5_000_000, exact_pad=False, **kw)
    db = kmlib.Database.from_records(case["keys"], case["counts"], K).upload(0)
    b = kmlib.Batch(db, max_targets=T, max_total_bases=T * L)
    blob = np.frombuffer(b"ACGT", dtype=np.uint8)[case["targets"]].copy().reshape(-1)
    b.set_targets_packed(blob, np.arange(T + 1, dtype=np.uint64) * np.uint64(L))
    st = kmlib.stream_create(0)
    flags = kmlib.KM_STAGE_WALK | kmlib.KM_STAGE_GRAPH | kmlib.KM_RUN_TIMED | kmlib.KM_RUN_SERIAL
    tm = []
    for _ in range(20):
        b.run(flags, st)
        b.sync()
        tm.append(b.timings())
    tm = np.array(tm)[5:].mean(axis=0)
    print("%s: flagged %d, pure hand-overs %d, left to k_graph by the epilogue %d; k_seed %.1f k_dfs %.1f graph %.1f us" %
          ((label,) + b.debug_counts() + (tm[3] * 1e3, tm[5] * 1e3, tm[1] * 1e3)), flush=True)
    b.close()
    db.close()
