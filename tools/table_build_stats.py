#!/usr/bin/env python3
"""Diagnostics: table builds with KM_BUILD_VERBOSE (rounds, buckets that grow, buckets holding a key outside its
home pair, the settle pass) — each set of records three times: as given, again, shuffled.  max_probe must agree.
usage: table_build_stats.py [n_keys]   (0: the bundled .jf fixtures only)"""
import os
import sys
import time

os.environ["KM_BUILD_VERBOSE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from km_amd import lib as kmlib, synth  # noqa: E402


def three_builds(label, keys, counts, k):
    seen = []
    for how in ("as given", "as given, again", "shuffled"):
        if how == "shuffled":
            perm = np.random.default_rng(5).permutation(len(keys))
            keys, counts = keys[perm], counts[perm]
        t0 = time.time()
        db = kmlib.Database.from_records(keys, counts, k).upload(0)
        info = db.info
        seen.append(info.max_probe)
        print("%s, %s: max_probe %d, %d slots, %.2f s" % (label, how, info.max_probe, info.n_slots, time.time() - t0), flush=True)
        db.close()
    print("%s: max_probe %s -> %s" % (label, seen, "SAME" if len(set(seen)) == 1 else "DIFFERENT"), flush=True)


nk = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
jfdir = os.path.join(ROOT, "tests", "data", "jf")
for name in sorted(os.listdir(jfdir)):
    db = kmlib.Database.open(os.path.join(jfdir, name))
    keys, counts = db.records()
    k = db.info.k
    db.close()
    three_builds(name, keys, counts, k)
if nk:
    case = synth.make_case(n_targets=10000, length=500, k=31, n_keys=nk, seed=synth.HEADLINE_SEED, exact_pad=False)
    three_builds("headline table", case["keys"], case["counts"], 31)
