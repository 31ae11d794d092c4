#!/usr/bin/env python3
"""Diagnostics: the pipelined step from a minimal Python process (what km_amd/kmclient pump does, through
ctypes): is bench.py's step slower than the C++ client's because of the interpreter, or of what bench.py did
before?  usage: pump_min.py <dir of bench.py --dump-case> [full_first]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KM_HIP_RUNTIME", "system")
import numpy as np  # noqa: E402

from km_amd import lib as kmlib  # noqa: E402

d = sys.argv[1]
keys, counts, tg = (np.load(os.path.join(d, f + ".npy")) for f in ("keys", "counts", "targets"))
n_fl, K = 4, 31
T, L = tg.shape[0] // n_fl, tg.shape[1]
db = kmlib.Database.from_records(keys, counts, K).upload(0)
ascii_ = np.frombuffer(b"ACGT", dtype=np.uint8)[tg].copy()
offs = np.arange(T + 1, dtype=np.uint64) * np.uint64(L)
streams = [kmlib.stream_create(0) for _ in range(n_fl)]
batches = []
for q in range(n_fl):
    b = kmlib.Batch(db, max_targets=T, max_total_bases=T * L)
    b.set_targets_packed(ascii_[q * T:(q + 1) * T].reshape(-1), offs)
    batches.append(b)
lean = kmlib.KM_STAGE_WALK | kmlib.KM_STAGE_GRAPH | kmlib.KM_RUN_DELIVER | kmlib.KM_DELIVER_LEAN
if os.environ.get("PUMP_SERIAL"):
    lean |= kmlib.KM_RUN_SERIAL
full = lean & ~kmlib.KM_DELIVER_LEAN


def timed(flags, label):
    kmlib.pump(batches, streams, 8, flags)
    ms = []
    for _ in range(3):
        t0 = time.perf_counter()
        kmlib.pump(batches, streams, 40, flags)
        ms.append((time.perf_counter() - t0) / 40 * 1e3)
    print("%s: %.4f .. %.4f ms/step" % (label, min(ms), max(ms)), flush=True)


if len(sys.argv) > 2:
    timed(full, "full delivery first")
timed(lean, "lean delivery")
if len(sys.argv) > 2:
    [b.result() for b in batches]
    timed(lean, "lean delivery after result() views were taken")
for b in batches:
    b.close()
db.close()
