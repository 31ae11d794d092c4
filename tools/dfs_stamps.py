#!/usr/bin/env python3
"""Diagnostics: where does a DFS step of k_dfs spend its time?  Builds the library with
-DKM_DFS_STAMPS (shader-clock stamps between the sections of a step, summed per wave), runs the
bench workload's first targets and prints the section shares of the longest-running waves and of
all waves.  usage: dfs_stamps.py [n_targets] [n_keys]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "km_amd", "csrc")
so = os.path.join(tempfile.gettempdir(), "libkmgpu_dfs_stamps.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                       "-ffp-contract=off", "-pthread", "-DKM_DFS_STAMPS"] + (["-DKM_DFS_STAMPS_CALIBRATE"] if os.environ.get("CALIBRATE") else []) + ["-o", so,
                       os.path.join(CSRC, "kmgpu.hip"), os.path.join(CSRC, "jf_reader.cpp"),
                       os.path.join(CSRC, "report.cpp")], cwd=ROOT)
os.environ["KM_LIBRARY"] = so
os.environ["KM_SEED_STAMPS"] = "1"          # allocates the stamp buffer

import numpy as np  # noqa: E402

from km_amd import lib as kmlib, synth  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
NK = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
L, K = 500, 31
case = synth.make_case(n_targets=T, length=L, k=K, n_keys=NK, seed=synth.HEADLINE_SEED, exact_pad=False)
db = kmlib.Database.from_records(case["keys"], case["counts"], K).upload(0)
b = kmlib.Batch(db, max_targets=T, max_total_bases=T * L)
blob = np.frombuffer(b"ACGT", dtype=np.uint8)[case["targets"]].copy().reshape(-1)
b.set_targets_packed(blob, np.arange(T + 1, dtype=np.uint64) * np.uint64(L))
st = kmlib.stream_create(0)
flags = kmlib.KM_STAGE_WALK | kmlib.KM_STAGE_GRAPH | kmlib.KM_RUN_TIMED | kmlib.KM_RUN_SERIAL
tm = []
for _ in range(8):
    b.run(flags, st)
    b.sync()
    tm.append(b.timings()[5])
print("k_dfs of this (stamped) build: %.1f us by HIP events" % (np.mean(tm[3:]) * 1e3))
rec = b.debug_stamps().reshape(-1, 32)
rec = rec[rec[:, 31] == 0x6466735F7374616D]
names = ["other", "late request", "wait+resolve", "next request", "unwind", "set probe", "rejoin", "push/booking",
         "thresholds", "next key", "chain: find", "chain: bucket change", "chain: non-resident lookup",
         "chain: end expansion", "epilogue", "-"]
tot = rec[:, 17].astype(np.float64)
setup = rec[:, 25].astype(np.float64)
order = np.argsort(-tot)
print("%d k_dfs waves; shader-clock ticks; s_memtime ticks per microsecond of s_memrealtime (100 MHz): median %.0f; longest eight:" % (len(rec), np.median(rec[:, 17] / (rec[:, 28] / 100.0))))
for i in order[:8]:
    r = rec[i]
    sec = r[:16].astype(np.float64)
    print("  target %5d: %7d ticks (setup %5d, unaccounted %5d), %3d steps of which %2d general, %d runs, %2d bucket "
          "loads (largest %4d slots), %3d non-resident lookups, %3d new nodes, %d stamps:\n      " %
          (r[18], r[17], r[25], r[17] - r[25] - sec.sum(), r[16], r[23], r[24], r[20], r[22], r[21], r[26], r[27]) +
          ", ".join("%s %d" % (nm, v) for nm, v in zip(names, sec) if v))
mid = order[len(order) // 2 - 2:len(order) // 2 + 2]
print("four waves around the median:")
for i in mid:
    r = rec[i]
    sec = r[:16].astype(np.float64)
    print("  target %5d: %7d ticks (setup %5d, unaccounted %5d), %3d steps of which %2d general, %d runs, %2d bucket "
          "loads (largest %4d slots), %3d non-resident lookups, %3d new nodes, %d stamps:\n      " %
          (r[18], r[17], r[25], r[17] - r[25] - sec.sum(), r[16], r[23], r[24], r[20], r[22], r[21], r[26], r[27]) +
          ", ".join("%s %d" % (nm, v) for nm, v in zip(names, sec) if v))
print("bucket loads %d, chain lookups in a bucket too large for the lanes %d, largest bucket %d slots; "
      "loads per wave p50 %d max %d" % (rec[:, 20].sum(), rec[:, 21].sum(), rec[:, 22].max(), np.median(rec[:, 20]), rec[:, 20].max()))
print("largest-bucket histogram (per wave, bins of 32 slots):", np.bincount(np.minimum(rec[:, 22] // 32, 16).astype(np.int64)).tolist())
sec = rec[:, :16].astype(np.float64).sum(axis=0)
print("all waves: wave time p50 %d p90 %d p99 %d max %d; setup p50 %d; per-section share: " %
      (np.percentile(tot, 50), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(), np.median(setup)) +
      ", ".join("%s %.0f%%" % (nm, 100 * v / sec.sum()) for nm, v in zip(names, sec) if v))
per = {"chain: find": (10, rec[:, 16] - rec[:, 23]), "chain: bucket change": (11, rec[:, 20]), "chain: non-resident lookup": (12, rec[:, 21]),
       "general step (1+2+3+8+9)": (None, rec[:, 23])}
for nm, (ix, cnt) in per.items():
    v = rec[:, ix].sum() if ix is not None else rec[:, [1, 2, 3, 8, 9]].sum()
    print("  %s: %.0f ticks each (%d events)" % (nm, v / max(1, cnt.sum()), cnt.sum()))
b.close()
db.close()
