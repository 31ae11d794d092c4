"""Per-stage latency of k_seed from its in-kernel time stamps (KM_SEED_STAMPS diagnostics).
Usage on the GPU box:  KM_SEED_STAMPS=1 python tools/seed_stamps.py [--cache DIR] [--keys N]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KM_SEED_STAMPS", "1")
import __graft_entry__ as ge  # noqa: E402

ge.build()
from km_amd import lib as kmlib, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--targets", type=int, default=10000)
ap.add_argument("--length", type=int, default=500)
ap.add_argument("--keys", type=int, default=100_000_000)
ap.add_argument("--cache", default="")
args = ap.parse_args()
tag = "%s/case_%d_%d_%d" % (args.cache, args.targets, args.length, args.keys)
if args.cache and os.path.exists(tag + "_keys.npy"):
    case = {f: np.load("%s_%s.npy" % (tag, f)) for f in ("keys", "counts", "targets")}
else:
    case = synth.make_case(n_targets=args.targets, length=args.length, k=31, n_keys=args.keys,
                           seed=synth.HEADLINE_SEED, exact_pad=False)
db = kmlib.Database.from_records(case["keys"], case["counts"], 31, True)
db.upload(0)
bases = np.frombuffer(b"ACGT", dtype=np.uint8)[case["targets"]].copy()
offs = np.arange(args.targets + 1, dtype=np.uint64) * np.uint64(args.length)
b = kmlib.Batch(db, ratio=0.05, count=5, max_stack=500, max_break=10, max_node=10000,
                max_targets=args.targets, max_total_bases=args.targets * args.length)
b.set_targets_packed(bases.reshape(-1), offs)
for _ in range(3):
    b.run(kmlib.KM_STAGE_WALK | kmlib.KM_RUN_TIMED)
    b.sync()
st = b.debug_stamps().astype(np.int64)
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/seed_stamps.npy", st)
st = st[st[:, 0] != 0]
names = ["header", "bases", "scan", "dir", "slot", "resolve", "tail"]
d = np.diff(st[:, :8], axis=1)
print("waves", len(st), "k_seed ms (events)", b.timings()[3])
life = st[:, 7] - st[:, 0]
print("lifetime cycles: mean %.0f p50 %.0f p90 %.0f p99 %.0f" % (
    life.mean(), np.percentile(life, 50), np.percentile(life, 90), np.percentile(life, 99)))
for j, nme in enumerate(names):
    c = d[:, j]
    print("%-8s mean %7.0f  p50 %7.0f  p90 %7.0f  p99 %7.0f" % (
        nme, c.mean(), np.percentile(c, 50), np.percentile(c, 90), np.percentile(c, 99)))
real = st[:, 8:10]
t0 = real[:, 0].min()
span = (real[:, 1].max() - t0) / 100.0       # s_memrealtime ticks at 100 MHz -> us
print("span us %.1f" % span)
pm = st[:, 12]
print("max probes per wave: mean %.2f p50 %d p90 %d p99 %d max %d; per-lane mean %.3f" % (
    pm.mean(), np.percentile(pm, 50), np.percentile(pm, 90), np.percentile(pm, 99), pm.max(), st[:, 13].sum() / (64.0 * len(st))))
clk = life.sum() / max(1, (real[:, 1] - real[:, 0]).sum()) * 100.0
print("shader clock MHz ~ %.0f" % clk)
# concurrency over time
ev = np.concatenate([np.stack([real[:, 0], np.ones(len(real), np.int64)], 1),
                     np.stack([real[:, 1], -np.ones(len(real), np.int64)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
conc = np.cumsum(ev[:, 1])
dt = np.diff(ev[:, 0])
print("mean waves resident %.0f (of %d slots)" % ((conc[:-1] * dt).sum() / max(1, dt.sum()), 256 * 4 * 8))
