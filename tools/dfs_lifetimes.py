#!/usr/bin/env python3
"""Diagnostics: when do the waves of k_dfs start and end?  (normal build, KM_SEED_STAMPS set: every working
wave records s_memrealtime at entry and exit.)  usage: dfs_lifetimes.py [n_keys]"""
import os
import sys

os.environ["KM_SEED_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
COUNTERS = bool(os.environ.get("DFS_COUNTERS"))
if COUNTERS:                 # a build that also counts, per wave, what the walk did (walk_kernel.h: KM_DFS_COUNTERS)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _diag
    _diag.build(["-DKM_DFS_COUNTERS"], tag="dfs_counters")
import numpy as np  # noqa: E402

from km_amd import lib as kmlib, synth  # noqa: E402

T, L, K = 10000, 500, 31
nk = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
case = synth.make_case(n_targets=T, length=L, k=K, n_keys=nk, seed=synth.HEADLINE_SEED, exact_pad=False)
db = kmlib.Database.from_records(case["keys"], case["counts"], K).upload(0)
b = kmlib.Batch(db, max_targets=T, max_total_bases=T * L)
blob = np.frombuffer(b"ACGT", dtype=np.uint8)[case["targets"]].copy().reshape(-1)
b.set_targets_packed(blob, np.arange(T + 1, dtype=np.uint64) * np.uint64(L))
st = kmlib.stream_create(0)
flags = kmlib.KM_STAGE_WALK | kmlib.KM_STAGE_GRAPH | kmlib.KM_RUN_TIMED | kmlib.KM_RUN_SERIAL
tm = []
for _ in range(6):
    b.run(flags, st)
    b.sync()
    tm.append(b.timings()[5])
rec = b.debug_stamps().reshape(-1, 32)
rec = rec[rec[:, 31] == 0x6C6966655F646673]
t0, t1 = rec[:, 0].astype(np.float64) / 100.0, rec[:, 1].astype(np.float64) / 100.0      # us
base = t0.min()
life = t1 - t0
print("k_dfs %.1f us by HIP events (k_seed runs its stamped variant in this mode); %d working waves" % (np.mean(tm[2:]) * 1e3, len(rec)))
print("wave start after the first one: p50 %.1f p90 %.1f p99 %.1f max %.1f us" % tuple(np.percentile(t0 - base, [50, 90, 99, 100])))
print("wave lifetime: p50 %.1f p90 %.1f p99 %.1f max %.1f us" % tuple(np.percentile(life, [50, 90, 99, 100])))
print("last wave ends %.1f us after the first one starts" % (t1.max() - base))
ts, tw = rec[:, 4].astype(np.float64) / 100.0, rec[:, 5].astype(np.float64) / 100.0
print("phases, all waves: setup p50 %.1f p99 %.1f, walk p50 %.1f p99 %.1f, epilogue p50 %.1f p99 %.1f us" %
      (np.percentile(ts - t0, 50), np.percentile(ts - t0, 99), np.percentile(tw - ts, 50), np.percentile(tw - ts, 99),
       np.percentile(t1 - tw, 50), np.percentile(t1 - tw, 99)))
tA, tB, tC = (rec[:, q].astype(np.float64) / 100.0 for q in (6, 7, 8))
print("setup, p50: header + packed words %.1f, clears %.1f, node set (+ the purity of the reference) %.1f us" %
      (np.median(tB - t0), np.median(tC - tB), np.median(ts - tC)))
order = np.argsort(-(t1 - base))[:8]
for i in order:
    print("  target %5d: start %.1f, setup %.1f, walk %.1f, epilogue %.1f, end %.1f us, %d new nodes" %
          (rec[i, 2], t0[i] - base, ts[i] - t0[i], tw[i] - ts[i], t1[i] - tw[i], t1[i] - base, rec[i, 3]))
# what a wave's walk time goes with: the nodes it discovers (chain steps) — a straight-line fit and its residue
nodes = rec[:, 3].astype(np.float64)
walk = tw - ts
A = np.stack([np.ones_like(nodes), nodes], axis=1)
coef, *_ = np.linalg.lstsq(A, walk, rcond=None)
res = walk - A @ coef
print("walk time ~ %.1f us + %.3f us per new node (r = %.2f); residue p50 %.1f p90 %.1f p99 %.1f max %.1f us" %
      (coef[0], coef[1], np.corrcoef(nodes, walk)[0, 1], *np.percentile(res, [50, 90, 99, 100])))
for lo, hi in ((0, 1), (1, 20), (20, 32), (32, 45), (45, 70), (70, 100), (100, 1000)):
    m = (nodes >= lo) & (nodes < hi)
    if m.any():
        print("  %3d-%3d new nodes: %5d waves, walk p50 %.1f p90 %.1f max %.1f us, life p50 %.1f max %.1f" %
              (lo, hi - 1, m.sum(), *np.percentile(walk[m], [50, 90, 100]), np.median(life[m]), life[m].max()))
if COUNTERS:
    short = ["slow", "spec", "rec", "bload", "gen", "runs", "full", "v0", "steps", "noalign"]
    names = ["slow steps", "speculation rounds", "children recorded by them", "bucket loads", "general expansions", "runs",
             "rounds that reached the target", "rounds with no step standing", "chain steps", "slow steps without an alignment"]
    c = rec[:, 9:19].astype(np.float64)
    print("per wave (mean / p50 / max):")
    for j, nm in enumerate(names):
        print("  %-36s %6.2f %5.0f %5.0f" % (nm, c[:, j].mean(), np.median(c[:, j]), c[:, j].max()))
    for lo, hi in ((20, 32), (32, 45), (45, 70)):
        m = (nodes >= lo) & (nodes < hi)
        if m.any():
            print("  %d-%d new nodes:" % (lo, hi - 1), ", ".join("%s %.2f" % (short[j], c[m, j].mean()) for j in range(len(short))))
    tn = ["speculation rounds", "booking", "general expansions", "bucket loads", "general child step (rejoin / push)", "unwinding", "alignment search"]
    tt = rec[:, 19:26].astype(np.float64) / 100.0
    print("time per wave in (us, mean / p50 / max):")
    for j, nm in enumerate(tn):
        print("  %-36s %6.2f %6.2f %6.2f" % (nm, tt[:, j].mean(), np.median(tt[:, j]), tt[:, j].max()))
    more = {"after an expansion (thresholds, prefetch)": rec[:, 29] >> 32, "run preamble": rec[:, 29] & 0xFFFFFFFF,
            "chain steps in the lanes": rec[:, 30] >> 32, "tail of a slow step": rec[:, 30] & 0xFFFFFFFF,
            "pops at the end of a seed": rec[:, 6] >> 32, "seed start": rec[:, 6] & 0xFFFFFFFF}
    extra = 0.0
    for nm, v in more.items():
        v = v.astype(np.float64) / 100.0
        extra += v.mean()
        print("  %-36s %6.2f %6.2f %6.2f" % (nm, v.mean(), np.median(v), v.max()))
    print("  accounted for: %.1f of a mean walk of %.1f us" % (tt.sum(axis=1).mean() + extra, walk.mean()))
    print("  non-resident lookups per wave: mean %.2f max %d, %.2f us each; largest bucket met by one: %d slots" %
          (rec[:, 26].mean(), rec[:, 26].max(), rec[:, 27].sum() / 100.0 / max(1, rec[:, 26].sum()), rec[:, 28].max()))
    for i in order[:6]:
        print("    (non-resident lookups %d, %.1f us, largest bucket %d)" % (rec[i, 26], rec[i, 27] / 100.0, rec[i, 28]))
        print("  target %d: walk %.1f us:" % (rec[i, 2], walk[i]), dict(zip(short, rec[i, 9:19].tolist())),
              {nm.split()[0]: round(float(v), 1) for nm, v in zip(tn, tt[i])})
b.close()
db.close()
