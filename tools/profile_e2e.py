"""cProfile of the end-to-end drop-in path (strings in -> TSV rows out) on the GPU box."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from km_amd import kmer as km, lib as kmlib, synth  # noqa: E402
from km_amd.finder import BatchFinder  # noqa: E402
from km_amd.jellyfish import Jellyfish  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
case = synth.make_case(n_targets=n, length=500, k=31, n_keys=2_000_000, seed=synth.HEADLINE_SEED, exact_pad=False)
db = kmlib.Database.from_records(case["keys"], case["counts"], 31).upload(0)
jf = Jellyfish("synthetic.jf", cutoff=0.05, n_cutoff=5, db=db)
finder = BatchFinder(jf)
tg = [(case["names"][i], km.decode(case["targets"][i])) for i in range(n)]
finder.rows(tg[:64])
t = time.perf_counter()
rows = finder.rows(tg)
print("e2e %.1f ms for %d targets, %d rows" % ((time.perf_counter() - t) * 1e3, n, sum(len(r) for r in rows)))
import io
for _ in range(3):
    sink = io.StringIO()
    t = time.perf_counter()
    finder.write_rows(tg, sink)
    dt = time.perf_counter() - t
print("write_rows %.1f ms for %d targets, %d rows (%.0f targets/s)"
      % (dt * 1e3, n, sink.getvalue().count("\n"), n / dt))
pr = cProfile.Profile()
pr.enable()
finder.write_rows(tg, io.StringIO())
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
pr = cProfile.Profile()
pr.enable()
finder.rows(tg)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
