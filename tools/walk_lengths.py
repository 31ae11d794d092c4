#!/usr/bin/env python3
"""Diagnostics: how long are the walks of a headline-shaped batch?  Extra logical probes per
target (beyond its trivial seeds) and walk-discovered nodes, as a histogram."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from km_amd import lib as kmlib, synth  # noqa: E402

T, L, K = 10000, 500, 31
case = synth.make_case(n_targets=T, length=L, k=K, n_keys=5_000_000, seed=synth.HEADLINE_SEED, exact_pad=False)
db = kmlib.Database.from_records(case["keys"], case["counts"], K).upload(0)
b = kmlib.Batch(db, max_targets=T, max_total_bases=T * L)
blob = np.frombuffer(b"ACGT", dtype=np.uint8)[case["targets"]].copy().reshape(-1)
b.set_targets_packed(blob, np.arange(T + 1, dtype=np.uint64) * np.uint64(L))
b.run()
r = b.fetch()
extra = np.diff(r["node_off"].astype(np.int64)) - r["n_ref"].astype(np.int64)
probes = r["probes"].astype(np.int64)
base = np.median(probes)
print("targets with walk-discovered nodes:", int((extra > 0).sum()), "max extra nodes", int(extra.max()))
print("extra nodes histogram:", np.histogram(extra[extra > 0], bins=[1, 10, 20, 31, 40, 60, 80, 100, 130, 161])[0].tolist())
print("logical probes per target: median %d, max %d; extra probes of the 10 longest: %s" %
      (base, probes.max(), (np.sort(probes)[-10:] - int(base)).tolist()))
