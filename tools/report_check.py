#!/usr/bin/env python3
"""Parity of the native reporting (km_report_rows) on a large sample: every target of the headline-shaped
batch through BatchFinder (HIP walk + path search, native rows) against km_amd/report.py (numpy: the
reference's own arithmetic) on the same delivered arrays.  Counts rows that differ, split by whether the
native path had FLAGGED the target (err 100: a printed value within 1e-6 of a rounding tie).
usage: report_check.py [n_targets] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KM_HIP_RUNTIME", "system")
import numpy as np  # noqa: E402

from km_amd import kmer as km, lib as kmlib, report, synth  # noqa: E402
from km_amd.finder import BatchFinder  # noqa: E402
from km_amd.jellyfish import Jellyfish  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else synth.HEADLINE_SEED
kw = {} if len(sys.argv) <= 3 else dict(variant_frac=0.85, variants_per_target=(1, 3), hom_frac=0.15, branch_noise_frac=0.03)
case = synth.make_case(n_targets=T, length=500, k=31, n_keys=2_000_000, seed=seed, exact_pad=False, **kw)
db = kmlib.Database.from_records(case["keys"], case["counts"], 31).upload(0)
jf = Jellyfish("synthetic.jf", cutoff=0.05, n_cutoff=5, db=db)
finder = BatchFinder(jf)
targets = [(nm, km.decode(r)) for nm, r in zip(case["names"], case["targets"])]
names, seqs = [t[0] for t in targets], [t[1] for t in targets]
packed = kmlib.pack_sequences(seqs)
b = finder._ensure(T, int(packed[1][-1]))
b.set_targets_packed(*packed)
b.run(kmlib.KM_STAGE_WALK | kmlib.KM_STAGE_GRAPH | kmlib.KM_RUN_DELIVER)
raw = b.result()
text, row_off, special = kmlib.report_text(raw, names, seqs, 31, "synthetic.jf", packed)
py = [report.target_rows(res, "synthetic.jf") for res in finder.analyse(targets)]
n_var = n_diff_flagged = n_diff_unflagged = n_flagged = 0
first = None
for t in range(T):
    native = text[int(row_off[t]):int(row_off[t + 1])].splitlines()
    want = py[t]
    if isinstance(want, BaseException):
        continue
    n_var += len(want) > 1
    flagged = t in special
    n_flagged += flagged
    if native != want:
        if flagged:
            n_diff_flagged += 1
        else:
            n_diff_unflagged += 1
            if first is None:
                first = (t, [a for a, c in zip(native, want) if a != c][:1], [c for a, c in zip(native, want) if a != c][:1])
print("%d targets, %d with variant rows, %d flagged by the native path (err 100); native text differs from numpy's: "
      "%d flagged targets, %d UNFLAGGED targets" % (T, n_var, n_flagged, n_diff_flagged, n_diff_unflagged))
if first:
    print("first unflagged difference: target %d\n  native %s\n  numpy  %s" % first)
sys.exit(1 if n_diff_unflagged else 0)
