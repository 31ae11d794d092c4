"""Which flagged targets of the headline-shaped batch does the epilogue of k_dfs NOT answer, and why?  (CPU only:
the oracle's walk + the shape conditions of tests/test_bubble_theory.py with a reason attached.)
    python tests/host/left_shapes.py [n_targets]"""
import collections
import os
import sys
from multiprocessing import Pool

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from km_amd import kmer as km, synth          # noqa: E402
from oracle import km_oracle as ko            # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
case = synth.make_case(n_targets=N, length=500, k=31, n_keys=3_000_000, seed=synth.HEADLINE_SEED, exact_pad=False)
DB = ko.KmerDB(None, cutoff=0.05, n_cutoff=5, records={"k": 31, "canonical": True, "keys": case["keys"], "counts": case["counts"]})


def reason(kmers, n_ref):
    m = len(kmers)
    if m == n_ref:
        return "no walk nodes"
    by_prefix = {}
    for i, mer in enumerate(kmers):
        by_prefix.setdefault(mer[:-1], []).append(i)
    if by_prefix.get(kmers[n_ref - 1][1:]):
        return "something behind the last reference suffix"
    head_of = {}
    for j, mer in enumerate(kmers):
        share = by_prefix[mer[:-1]]
        if len(share) > 2:
            return "three nodes share a prefix"
        if len(share) == 2:
            other = share[0] if share[1] == j else share[1]
            if j >= n_ref:
                if not (1 <= other <= n_ref - 1):
                    return "two walk nodes share a prefix" if other >= n_ref else "walk node shares prefix with reference node 0"
                head_of[j] = other - 1
            elif other < n_ref:
                return "two reference nodes share a prefix"
    nb, e = 0, n_ref
    while e < m:
        if e not in head_of:
            return "a walk chain that does not start at a head (dead end / nested)"
        a, s = head_of[e], e
        while True:
            behind = by_prefix.get(kmers[e][1:], [])
            if len(behind) == 0:
                return "dead end (a walk node with no successor)"
            if len(behind) > 1:
                return "a walk node with several successors"
            nxt = behind[0]
            if nxt < n_ref:
                b = nxt
                break
            if nxt != e + 1 or nxt in head_of:
                return "walk chain out of order / joins another head"
            e = nxt
        if a < b and (b - a) + 10 > 100 * (e - s + 2):
            return "bubble cheaper than the reference route"
        nb += 1
        e += 1
    return "ok (%d bubble%s)" % (nb, "" if nb == 1 else "s") if nb <= 8 else "more than 8 bubbles"


def one(t):
    seq = km.decode(case["targets"][t])
    r = ko.analyse_target(seq, "t%d" % t, DB)
    n_ref = len(seq) - 31 + 1
    return t, reason(r["kmers"], n_ref), len(r["kmers"]) - n_ref, len(r["paths"])


if __name__ == "__main__":
    with Pool(8) as pool:
        res = pool.map(one, range(N), chunksize=50)
    cnt = collections.Counter(r[1] for r in res)
    for k_, v in cnt.most_common():
        print("%6d  %s" % (v, k_))
    for t, why, nw, npaths in res:
        if not why.startswith("ok") and why != "no walk nodes":
            print("  target %d: %s; %d walk nodes, %d paths" % (t, why, nw, npaths))
