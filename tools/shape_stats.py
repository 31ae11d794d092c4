import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from km_amd import lib as kmlib, synth
T, L, K = 10000, 500, 31
case = synth.make_case(n_targets=T, length=L, k=K, n_keys=20_000_000, seed=synth.HEADLINE_SEED, exact_pad=False)
db = kmlib.Database.from_records(case["keys"], case["counts"], K).upload(0)
b = kmlib.Batch(db, max_targets=T, max_total_bases=T * L)
blob = np.frombuffer(b"ACGT", dtype=np.uint8)[case["targets"]].copy().reshape(-1)
b.set_targets_packed(blob, np.arange(T + 1, dtype=np.uint64) * np.uint64(L))
b.run(kmlib.KM_STAGE_WALK | kmlib.KM_STAGE_GRAPH | kmlib.KM_RUN_DELIVER)
r = b.result()
npaths = np.diff(r["path_off"].astype(np.int64))
extra = np.diff(r["extra_off"].astype(np.int64))
print("targets with extra nodes:", int((extra > 0).sum()))
print("paths per target histogram (targets with extra nodes):", np.bincount(npaths[extra > 0]).tolist())
print("paths per target histogram (all):", np.bincount(npaths).tolist())
# runs per path for 2-path targets
roff = r["run_off"].astype(np.int64); poff = r["path_off"].astype(np.int64)
two = np.flatnonzero((npaths == 2) & (extra > 0))
nruns = np.array([int(roff[poff[t] + 2] - roff[poff[t]]) for t in two])
print("2-path targets by total runs:", np.bincount(nruns).tolist())
