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
flags = kmlib.KM_STAGE_WALK | kmlib.KM_STAGE_GRAPH
for _ in range(3):
    b.run(flags, st)
    b.sync()
rec = b.debug_stamps()
rec = rec[rec[:, 15] == 0x6466735F7374616D]
names = ["other", "late request", "wait+resolve", "dir+pair request", "unwind", "set probe", "rejoin", "push",
         "thresholds", "next key"]
rec = np.concatenate([rec[:, :12], rec[:, 12:14], rec[:, 14:]], axis=1)
SEC = [0, 1, 2, 3, 4, 5, 6, 7, 12, 13]
tot = rec[:, 9].astype(np.float64)
order = np.argsort(-tot)
print("%d k_dfs waves; shader-clock ticks; longest five:" % len(rec))
for i in order[:5]:
    r = rec[i]
    sec = r[SEC].astype(np.float64)
    print("  target %5d: %7d ticks, %3d expansions (%5.0f ticks each), %4d probes, setup+tail %4.1f %%: " %
          (r[10], r[9], r[8], sec.sum() / max(1, r[8]), r[11], 100 * (1 - sec.sum() / r[9])) +
          ", ".join("%s %.0f%%" % (nm, 100 * v / sec.sum()) for nm, v in zip(names, sec) if v))
extra = rec[:, 14]
loads, nonres, maxS = extra & 0xFFFFF, (extra >> 20) & 0xFFFFF, extra >> 40
print("bucket loads %d, chain lookups in a bucket too large for the lanes %d, largest bucket %d slots; "
      "loads per wave p50 %d max %d" % (loads.sum(), nonres.sum(), maxS.max(), np.median(loads), loads.max()))
print("largest-bucket histogram (per wave):", np.bincount(np.minimum(maxS // 32, 16)).tolist())
sec = rec[:, SEC].astype(np.float64).sum(axis=0)
print("all waves: %.0f ticks per expansion; setup+tail %.1f %% of wave time; " %
      (sec.sum() / rec[:, 8].sum(), 100 * (1 - sec.sum() / tot.sum())) +
      ", ".join("%s %.0f%%" % (nm, 100 * v / sec.sum()) for nm, v in zip(names, sec)))
b.close()
db.close()
