"""The closed form k_graph uses for "reference chain + one forward bubble" (graph_kernel.h, 2c'
and 2c), checked on the CPU against the oracle's restatement of the reference algorithm
(oracle/km_oracle.py: graph_paths = Graph.py:63-240): whenever the shape conditions hold on a node
list — stated here exactly as the kernel reads them off its prefix table — the paths must be the
reference path and the one path through the bubble, nothing else.  Random node lists come from the
oracle's own walk over seeded synthetic databases (several variants per target, sibling noise,
small k for chance overlaps), so that most shapes do NOT qualify; those are only counted."""
import numpy as np
import pytest

from km_amd import kmer as km
from km_amd import synth
from oracle import km_oracle as ko


def bubble_shape(kmers, n_ref):
    """(a, b) if the node list is the reference chain plus one forward bubble by the kernel's
    conditions (R1)-(R4) and the cost margin, else None."""
    m = len(kmers)
    if not (m > n_ref >= 2):
        return None
    by_prefix = {}
    for i, mer in enumerate(kmers):
        by_prefix.setdefault(mer[:-1], []).append(i)
    a = None
    for j, mer in enumerate(kmers):
        share = by_prefix[mer[:-1]]
        if len(share) > 2:
            return None
        if len(share) == 2:
            other = share[0] if share[1] == j else share[1]
            if j == n_ref:
                if not (1 <= other <= n_ref - 1):
                    return None
                a = other - 1
            elif j < n_ref:
                if other != n_ref:
                    return None
            else:
                return None
        elif j == n_ref:
            return None
    if by_prefix.get(kmers[n_ref - 1][1:]):            # (R1) nothing behind the last reference suffix
        return None
    b = None
    for e in range(n_ref, m):                           # (R3) / (R4)
        behind = by_prefix.get(kmers[e][1:], [])
        if len(behind) != 1:
            return None
        if e + 1 < m:
            if behind[0] != e + 1:
                return None
        elif behind[0] < n_ref:
            b = behind[0]
        else:
            return None
    if a is None or b is None or not (a < b <= n_ref - 1):
        return None
    if (b - a) + 10 > 100 * (m - n_ref + 1):
        return None
    return a, b


@pytest.mark.parametrize("k", [9, 13, 21, 31])
def test_closed_form_equals_the_reference_algorithm(k):
    rng = np.random.default_rng(9000 + k)
    n_shape = n_other = 0
    for trial in range(24):
        spec = dict(n_targets=14, length=int(rng.integers(3 * k + 8, 260)), k=k, n_keys=3000,
                    seed=int(rng.integers(1, 1 << 30)), variant_frac=0.9,
                    variants_per_target=(1, int(rng.integers(1, 3))), vaf=(0.1, 0.9),
                    hom_frac=float(rng.choice([0.0, 0.3])), branch_noise_frac=float(rng.choice([0.0, 0.05])),
                    noise_frac=float(rng.choice([0.0, 0.05])), cov=(30, 400))
        case = synth.make_case(**spec)
        db = ko.KmerDB(None, cutoff=0.05, n_cutoff=5,
                       records={"k": k, "canonical": True, "keys": case["keys"], "counts": case["counts"]})
        for row, name in zip(case["targets"], case["names"]):
            seq = km.decode(row)
            try:
                mers = ko.ref_kmers(seq, name, k)
                nodes = ko.walk(mers, db)
            except (ValueError, ko.NodeLimit):
                continue
            kmers = list(nodes.keys())
            n_ref = len(mers)
            shape = bubble_shape(kmers, n_ref)
            if shape is None:
                n_other += 1
                continue
            n_shape += 1
            a, b = shape
            m = len(kmers)
            want = sorted([tuple(range(n_ref)),
                           tuple(range(a + 1)) + tuple(range(n_ref, m)) + tuple(range(b, n_ref))])
            got = [tuple(p) for p in ko.graph_paths(kmers, n_ref)]
            assert got == want, (k, trial, name, a, b, n_ref, m)
    assert n_shape >= 10 and n_other >= 10, (n_shape, n_other)       # both sides of the conditions were met


def bubbles_shape(kmers, n_ref, allow_back=False, allow_loop=False):
    """The shape the epilogue of k_dfs answers (walk_kernel.h): the reference chain plus ANY number of
    bubbles, each a run of consecutive walk nodes s..e hanging off reference node a and leading back to
    reference node b — forward (b > a, with the cost margin) or, with allow_back, backward (b <= a: a
    tandem duplication).  Returns [(a, s, e, b)] or None.  With allow_loop a bubble may also lead back to
    ITS OWN head beside b (a tandem duplication a little shorter than k: the head shares its prefix with b =
    a + 1, so whatever points at b points at the head too): tuples are then (a, s, e, b, looped)."""
    m = len(kmers)
    if not (m > n_ref >= 2):
        return None
    by_prefix = {}
    for i, mer in enumerate(kmers):
        by_prefix.setdefault(mer[:-1], []).append(i)
    if by_prefix.get(kmers[n_ref - 1][1:]):
        return None
    head_of = {}                                        # walk node -> a
    for j, mer in enumerate(kmers):
        share = by_prefix[mer[:-1]]
        if len(share) > 2:
            return None
        if len(share) == 2:
            other = share[0] if share[1] == j else share[1]
            if j >= n_ref:
                if not (1 <= other <= n_ref - 1):
                    return None                         # two walk nodes share a prefix
                head_of[j] = other - 1
            elif other < n_ref:
                return None                             # two reference nodes share a prefix
    bubbles, e = [], n_ref
    while e < m:
        if e not in head_of:
            return None                                 # a chain must start at a head
        a, s = head_of[e], e
        looped = False
        while True:
            behind = by_prefix.get(kmers[e][1:], [])
            if allow_loop and len(behind) == 2 and s in behind and min(behind) == a + 1:
                b, looped = a + 1, True                 # the end of the bubble points at b and at its own head
                break
            if len(behind) != 1:
                return None
            nxt = behind[0]
            if nxt < n_ref:
                b = nxt
                break
            if nxt != e + 1 or nxt in head_of:
                return None
            e = nxt
        if b > n_ref - 1:
            return None
        if a < b:
            if (b - a) + 10 > 100 * (e - s + 2):
                return None
        elif not allow_back:
            return None
        bubbles.append((a, s, e, b, looped) if allow_loop else (a, s, e, b))
        e += 1
    return bubbles


@pytest.mark.parametrize("k", [13, 21, 31])
def test_closed_form_for_several_bubbles(k):
    """The same argument carries to several forward bubbles (each one more path through exactly that
    bubble): checked here against the oracle so that the next kernel step starts from a proven form."""
    rng = np.random.default_rng(9500 + k)
    n_multi = 0
    for trial in range(22):
        spec = dict(n_targets=12, length=int(rng.integers(5 * k + 8, 320)), k=k, n_keys=3000,
                    seed=int(rng.integers(1, 1 << 30)), variant_frac=1.0, variants_per_target=(2, 3),
                    kinds=("snv", "ins", "del"), vaf=(0.2, 0.8), noise_frac=0.0, cov=(60, 400))
        case = synth.make_case(**spec)
        db = ko.KmerDB(None, cutoff=0.05, n_cutoff=5,
                       records={"k": k, "canonical": True, "keys": case["keys"], "counts": case["counts"]})
        for row, name in zip(case["targets"], case["names"]):
            seq = km.decode(row)
            try:
                mers = ko.ref_kmers(seq, name, k)
                nodes = ko.walk(mers, db)
            except (ValueError, ko.NodeLimit):
                continue
            kmers = list(nodes.keys())
            n_ref = len(mers)
            shape = bubbles_shape(kmers, n_ref)
            if shape is None:
                continue
            n_multi += len(shape) > 1
            want = sorted([tuple(range(n_ref))] +
                          [tuple(range(a + 1)) + tuple(range(s, e + 1)) + tuple(range(b, n_ref)) for a, s, e, b in shape])
            got = [tuple(p) for p in ko.graph_paths(kmers, n_ref)]
            assert got == want, (k, trial, name, shape, n_ref, len(kmers))
    assert n_multi >= 10, n_multi


@pytest.mark.parametrize("k", [13, 21, 31])
def test_closed_form_with_backward_bubbles(k):
    """Tandem duplications: the bubble leads BACK to a reference node b <= a.  The path through it is
    still 0..a, the bubble, b..n_ref-1 (b..a twice), next to the reference path and one path per other
    bubble — what the epilogue of k_dfs writes for them, checked against the oracle."""
    rng = np.random.default_rng(4242 + k)
    n_back = n_mixed = 0
    for trial in range(24):
        spec = dict(n_targets=12, length=int(rng.integers(5 * k + 8, 400)), k=k, n_keys=3000,
                    seed=int(rng.integers(1, 1 << 30)), variant_frac=1.0,
                    variants_per_target=(1, int(rng.integers(1, 4))), kinds=("snv", "ins", "del", "dup"),
                    vaf=(0.2, 0.8), noise_frac=float(rng.choice([0.0, 0.03])), cov=(60, 400))
        case = synth.make_case(**spec)
        db = ko.KmerDB(None, cutoff=0.05, n_cutoff=5,
                       records={"k": k, "canonical": True, "keys": case["keys"], "counts": case["counts"]})
        for row, name in zip(case["targets"], case["names"]):
            seq = km.decode(row)
            try:
                mers = ko.ref_kmers(seq, name, k)
                nodes = ko.walk(mers, db)
            except (ValueError, ko.NodeLimit):
                continue
            kmers = list(nodes.keys())
            n_ref = len(mers)
            shape = bubbles_shape(kmers, n_ref, allow_back=True)
            if shape is None:
                continue
            back = sum(1 for a, s, e, b in shape if b <= a)
            n_back += back > 0 and back == len(shape)
            n_mixed += 0 < back < len(shape)
            want = sorted([tuple(range(n_ref))] +
                          [tuple(range(a + 1)) + tuple(range(s, e + 1)) + tuple(range(b, n_ref)) for a, s, e, b in shape])
            got = [tuple(p) for p in ko.graph_paths(kmers, n_ref)]
            assert got == want, (k, trial, name, shape, n_ref, len(kmers))
    assert n_back >= 10 and n_mixed >= 10, (n_back, n_mixed)


def _tandem_case(rng, k):
    """A random target with a tandem duplication of k-4 .. k-1 bases (and sometimes an SNV elsewhere) in a
    database of its own: (sequence, KmerDB)."""
    while True:
        L = int(rng.integers(6 * k, 12 * k))
        row = rng.integers(0, 4, size=L, dtype=np.uint8)
        refk = km.sliding_kmers(row[None, :], k)[0]
        if len(set(refk.tolist())) != len(refk):
            continue
        n = int(rng.integers(k - 4, k))
        p = int(rng.integers(k, L - k - n))
        muts = [np.concatenate([row[:p + n], row[p:p + n], row[p + n:]]).astype(np.uint8)]
        if rng.random() < 0.5:                                    # a second, ordinary variant well away from it
            q = int(rng.integers(k, L - k))
            if abs(q - p) > 3 * k:
                snv = row.copy()
                snv[q] = (snv[q] + 1 + rng.integers(0, 3)) % 4
                muts.append(snv)
        counts = {}
        refset = set(refk.tolist())
        for x in refk.tolist():
            counts[km.canonical(x, k)] = 100
        for mut in muts:
            for x in km.sliding_kmers(mut[None, :], k)[0].tolist():
                if x not in refset:
                    counts[km.canonical(x, k)] = 60
        keys = np.array(sorted(counts), dtype=np.uint64)
        vals = np.array([counts[x] for x in sorted(counts)], dtype=np.uint32)
        return km.decode(row), ko.KmerDB(None, cutoff=0.05, n_cutoff=5,
                                         records={"k": k, "canonical": True, "keys": keys, "counts": vals})


@pytest.mark.parametrize("k", [13, 21, 31])
def test_closed_form_with_looped_bubbles(k):
    """A tandem duplication a little shorter than k: the bubble's head shares its prefix with b = a + 1, so the
    bubble's last node points at b AND at the head — a loop.  The reference's algorithm then finds one more
    path, through the loop edge: 0..a, the bubble TWICE, b..n_ref-1 (shortest source -> end of the bubble, the
    loop edge, shortest head -> sink); everything else is as for an ordinary forward bubble with b - a = 1.
    All 13 targets of the bench batch that round 3's epilogue still left to k_graph are of this kind."""
    rng = np.random.default_rng(77000 + k)
    n_loop = n_loop_mixed = 0
    for trial in range(260):
        seq, db = _tandem_case(rng, k)
        try:
            mers = ko.ref_kmers(seq, "t", k)
            nodes = ko.walk(mers, db)
        except (ValueError, ko.NodeLimit):
            continue
        kmers = list(nodes.keys())
        n_ref = len(mers)
        shape = bubbles_shape(kmers, n_ref, allow_back=True, allow_loop=True)
        if shape is None:
            continue
        loops = sum(1 for x in shape if x[4])
        n_loop += loops > 0
        n_loop_mixed += loops > 0 and len(shape) > loops
        want = [tuple(range(n_ref))]
        for a, s, e, b, looped in shape:
            once = tuple(range(a + 1)) + tuple(range(s, e + 1)) + tuple(range(b, n_ref))
            want.append(once)
            if looped:
                want.append(tuple(range(a + 1)) + 2 * tuple(range(s, e + 1)) + tuple(range(b, n_ref)))
        got = [tuple(p) for p in ko.graph_paths(kmers, n_ref)]
        assert got == sorted(want), (k, trial, shape, n_ref, len(kmers))
    assert n_loop >= 15 and n_loop_mixed >= 3, (n_loop, n_loop_mixed)
