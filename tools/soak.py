"""Soak: many seeded synthetic batches (different seeds, k, coverages, walk budgets) through the
HIP path, every target checked against the plain-C oracle (test infrastructure) and the native
rows against the Python reporting.  Usage on the GPU box: python tools/soak.py [rounds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from km_amd import kmer as km, lib as kmlib, report, synth  # noqa: E402
from km_amd.finder import BatchFinder  # noqa: E402
from km_amd.jellyfish import Jellyfish  # noqa: E402
from oracle import c_oracle  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "20261004")))
bad = 0
only = os.environ.get("SOAK_ONLY")
for rd in range(rounds):
    seed = int(rng.integers(1, 1 << 30))
    k = int(rng.choice([21, 25, 31, 31, 31, 32]))
    n = int(rng.choice([600, 1500]))
    length = int(rng.choice([200, 350, 500]))
    if os.environ.get("SOAK_LONG"):                 # targets beyond the LDS tier: every one of them through the large tier
        n, length = int(rng.choice([24, 70])), int(rng.choice([2200, 3100]))
    cov = (20, 300) if rng.random() < 0.3 else (50, 2000)
    multi = rng.random() < 0.4                      # several variants per target, homozygous ones, dead-end branches
    heavy = rng.random() < 0.15 or bool(os.environ.get("SOAK_HEAVY"))   # 3-5 tandem duplications: the large tier
    n_keys = int(rng.choice([200_000, 1_500_000]))
    vfrac = float(rng.choice([0.3, 0.7, 1.0]))
    ratio, count = float(rng.choice([0.05, 0.05, 0.2, 0.01])), int(rng.choice([5, 5, 2, 30]))
    steps, branchs = (int(rng.choice([500, 60])), int(rng.choice([10, 3])))
    if only is not None and int(only) != rd:
        continue
    case = synth.make_case(n_targets=n, length=length, k=k, n_keys=n_keys, seed=seed,
                           variant_frac=vfrac, cov=cov, exact_pad=False,
                           variants_per_target=(3, 5) if heavy else ((1, 3) if multi else (1, 1)),
                           kinds=("dup",) if heavy else ("snv", "ins", "del", "dup"), hom_frac=0.25 if multi else 0.0,
                           branch_noise_frac=0.03 if multi else 0.0, noise_frac=0.03 if multi else 0.01)
    t0 = time.perf_counter()
    db = kmlib.Database.from_records(case["keys"], case["counts"], k).upload(0)
    b = kmlib.Batch(db, ratio=ratio, count=count, max_stack=steps, max_break=branchs, max_targets=n,
                    max_total_bases=n * length)
    b.set_targets([km.decode(r) for r in case["targets"]])
    b.run()
    r = b.fetch()
    # every key, pad included: at k = 21 a random pad k-mer now and then IS a neighbour of a target k-mer
    co = c_oracle.COracle(case["keys"], case["counts"], k)
    noff, poff = r["node_off"].astype(np.int64), r["path_off"].astype(np.int64)
    mism = 0
    for t in range(n):
        want = co.analyse(case["targets"][t], ratio=ratio, count=count, max_stack=steps, max_break=branchs)
        ok = (want["status"] == int(r["status"][t]) == 0
              and (r["node_kmer"][noff[t]:noff[t + 1]] == want["kmers"]).all()
              and (r["node_count"][noff[t]:noff[t + 1]] == want["counts"]).all()
              and int(r["probes"][t]) == want["probes"]
              and [kmlib.expand_path(r, p).tolist() for p in range(poff[t], poff[t + 1])] == want["paths"]
              and r["path_min_cov"][poff[t]:poff[t + 1]].tolist() == want["min_cov"])
        mism += not ok
        if not ok and mism <= 3:
            got_paths = [kmlib.expand_path(r, p).tolist() for p in range(poff[t], poff[t + 1])]
            print("  MISMATCH target", t, case["names"][t], "status", want["status"], int(r["status"][t]),
                  "nodes", len(want["kmers"]), int(noff[t + 1] - noff[t]), "probes", want["probes"], int(r["probes"][t]),
                  "paths", len(want["paths"]), len(got_paths), "min_cov", want["min_cov"],
                  r["path_min_cov"][poff[t]:poff[t + 1]].tolist(), flush=True)
            nk = min(len(want["kmers"]), int(noff[t + 1] - noff[t]))
            dk = np.nonzero(r["node_kmer"][noff[t]:noff[t] + nk] != want["kmers"][:nk])[0]
            dc = np.nonzero(r["node_count"][noff[t]:noff[t] + nk] != want["counts"][:nk])[0]
            print("   first kmer diff", dk[:3], "first count diff", dc[:3], "paths equal", got_paths == want["paths"], flush=True)
            np.save("gpurun_out/soak_fail_target.npy", case["targets"][t])
    # native vs python reporting on the same fetched arrays
    jf = Jellyfish("soak.jf", cutoff=ratio, n_cutoff=count, db=db)
    finder = BatchFinder(jf, steps, branchs, 10000)
    targets = [(nm, km.decode(rw)) for nm, rw in zip(case["names"], case["targets"])]
    native = finder.rows(targets[:400])
    python = [report.target_rows(res, jf.filename) for res in finder.analyse(targets[:400])]
    rep_mism = sum(a != b_ for a, b_ in zip(native, python))
    bad += mism + rep_mism
    print("round %d seed %d k %d n %d len %d -p %g -c %d multi %d steps %d/%d: walk/path mismatches %d, report mismatches %d, "
          "large tier %d, max_probe %d, %.1f s" % (rd, seed, k, n, length, ratio, count, multi, steps, branchs, mism, rep_mism, int(r["n_big_tier"]), db.info.max_probe,
                                    time.perf_counter() - t0), flush=True)
print("SOAK", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
