import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from km_amd import lib as kmlib, synth
n = int(sys.argv[1])
case = synth.make_case(n_targets=10000, length=500, k=31, n_keys=n, seed=synth.HEADLINE_SEED, exact_pad=False)
db = kmlib.Database.from_records(case["keys"], case["counts"], 31).upload(0)
i = db.info
print("slots", i.n_slots, "groups", i.n_groups, "max_probe", i.max_probe, "bytes/kmer", i.table_bytes / len(case["keys"]))
