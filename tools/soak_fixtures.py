"""Soak on the REAL k-mer tables of the bundled fixtures (crowded minimizer buckets, max_probe > 2):
random windows of the catalog targets, some with random point mutations, walked on the GPU and
checked target by target against the plain-C oracle.  Usage: python tools/soak_fixtures.py [n]"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from km_amd import kmer as km, lib as kmlib  # noqa: E402
from oracle import c_oracle, jf_reader as jr, km_oracle as ko  # noqa: E402

n_per_db = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "99")))
catalog = [ko.read_fasta_concat(f) for f in sorted(glob.glob(os.path.join(ROOT, "tests/data/catalog/GRCh38/*.fa")))]
bad = 0
for path in sorted(glob.glob(os.path.join(ROOT, "tests/data/jf/*.jf"))):
    d = jr.read_jf(path)
    k = d["k"]
    db = kmlib.Database.load(path, 0)
    co = c_oracle.COracle(d["keys"], d["counts"], k, canonical=d["canonical"])
    seqs = []
    while len(seqs) < n_per_db:
        src = catalog[int(rng.integers(0, len(catalog)))]
        L = int(rng.integers(k + 5, min(len(src), 450)))
        a = int(rng.integers(0, len(src) - L + 1))
        s = list(src[a:a + L].upper())
        for _ in range(int(rng.integers(0, 3))):            # 0-2 point mutations
            p = int(rng.integers(0, L))
            s[p] = "ACGT"[int(rng.integers(0, 4))]
        s = "".join(s)
        kms = [s[i:i + k] for i in range(L - k + 1)]
        if len(set(kms)) == len(kms) and set(s) <= set("ACGT"):
            seqs.append(s)
    for ratio, count, steps, branchs in ((0.05, 5, 500, 10), (0.30, 500, 500, 10), (0.01, 2, 120, 4)):
        b = kmlib.Batch(db, ratio=ratio, count=count, max_stack=steps, max_break=branchs, max_node=10000,
                        max_targets=len(seqs), max_total_bases=sum(len(s) for s in seqs))
        b.set_targets(seqs)
        b.run()
        r = b.fetch()
        noff, poff = r["node_off"].astype(np.int64), r["path_off"].astype(np.int64)
        mism = 0
        for t, s in enumerate(seqs):
            want = co.analyse(km.encode(s) if hasattr(km, "encode") else np.array(["ACGT".index(c) for c in s], np.uint8),
                              ratio=ratio, count=count, max_stack=steps, max_break=branchs)
            st = int(r["status"][t])
            if want["status"] != st:
                mism += 1
                continue
            if st != 0:
                continue
            ok = ((r["node_kmer"][noff[t]:noff[t + 1]] == want["kmers"]).all()
                  and (r["node_count"][noff[t]:noff[t + 1]] == want["counts"]).all()
                  and int(r["probes"][t]) == want["probes"]
                  and [kmlib.expand_path(r, p).tolist() for p in range(poff[t], poff[t + 1])] == want["paths"]
                  and r["path_min_cov"][poff[t]:poff[t + 1]].tolist() == want["min_cov"])
            mism += not ok
        bad += mism
        print("%s -p %g -c %d -s %d -b %d: %d targets, mismatches %d, max_probe %d, large tier %d, multi-path %d" % (
            os.path.basename(path), ratio, count, steps, branchs, len(seqs), mism, db.info.max_probe,
            int(r["n_big_tier"]), int((np.diff(poff) > 1).sum())), flush=True)
        b.close()
print("SOAK", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
