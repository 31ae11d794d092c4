#!/usr/bin/env python3
"""Diagnostics: bench.py's config4_hard workload alone (1 M-key table: quick), pipelined and per kernel.
usage: hard_only.py [inflight]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KM_HIP_RUNTIME", "system")
import numpy as np  # noqa: E402

from km_amd import lib as kmlib, synth  # noqa: E402

n_fl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T, L, K = 10000, 500, 31
hc = synth.make_case(n_targets=2 * T, length=L, k=K, n_keys=20_000_000, seed=synth.HEADLINE_SEED + 7, variant_frac=0.85,
                     variants_per_target=(1, 3), hom_frac=0.15, branch_noise_frac=0.03, noise_frac=0.03, heavy_frac=float(os.environ.get("HEAVY", "0.04")),
                     exact_pad=False)
db = kmlib.Database.from_records(hc["keys"], hc["counts"], K).upload(0)
offs = np.arange(T + 1, dtype=np.uint64) * np.uint64(L)
blob = np.frombuffer(b"ACGT", dtype=np.uint8)[hc["targets"]].copy()
streams = [kmlib.stream_create(0) for _ in range(n_fl)]
batches = []
for q in range(n_fl):
    b = kmlib.Batch(db, max_targets=T, max_total_bases=T * L)
    b.set_targets_packed(blob[(q % 2) * T:(q % 2 + 1) * T].reshape(-1), offs)
    batches.append(b)
deliver = kmlib.KM_STAGE_WALK | kmlib.KM_STAGE_GRAPH | kmlib.KM_RUN_DELIVER | kmlib.KM_DELIVER_LEAN
kmlib.pump(batches, streams, 8, deliver)
for rep in range(3):
    t0 = time.perf_counter()
    kmlib.pump(batches, streams, 40, deliver)
    print("pipelined, %d in flight: %.3f ms/step" % (n_fl, (time.perf_counter() - t0) / 40 * 1e3), flush=True)
tm = []
for i in range(12):
    b = batches[i % n_fl]
    t0 = time.perf_counter()
    b.run(deliver | kmlib.KM_RUN_TIMED, streams[i % n_fl])
    s = b.wait_result()
    tm.append(b.timings() + ((time.perf_counter() - t0) * 1e3,))
tm = np.mean(np.array(tm)[2:], axis=0)
print("one batch at a time: walk %.3f (pack %.3f seed %.3f dfs %.3f) graph %.3f deliver %.3f d2h %.3f; whole step on the host %.3f ms; "
      "large tier %d, flagged/left %s" % (tm[0], tm[4], tm[3], tm[5], tm[1], tm[6], tm[7], tm[8], int(s.n_big_tier), batches[0].debug_counts()))
