"""Host-side variant naming, path quantification and TSV rows.

Consumes what the GPU path search returns (node k-mers / counts and the
alternative paths as node-index arrays) and produces the rows
``km find_mutation`` prints.  SURVEY.md §8f-1 keeps this part on the host; it is
needed for TSV parity.  Behaviour follows the reference:

    split_paths / name_variant  <- km/utils/MutationFinder.py:190-373, 405-488
    fit_paths                   <- km/utils/PathQuant.py:93-154 (same numpy calls,
                                   same dtypes, so the printed %.3f / %.1f agree)
    cluster_groups              <- km/utils/MutationFinder.py:651-723
    target_rows                 <- km/utils/MutationFinder.py:575-648, 726-833
    HEADER / format_row         <- km/utils/PathQuant.py:37-49, 72-90

Node numbering is the kernels' canonical order: node i (< n_ref) is the i-th
k-mer of the target, the two capping nodes follow the last real node.
"""

import re

import numpy as np

from . import kmer as km

HEADER = "\t".join(["Database", "Query", "Type", "Variant_name", "rVAF", "Expression",
                    "Min_coverage", "Start_offset", "Sequence", "Reference_expression",
                    "Reference_sequence", "Info"])

_LAST = np.frombuffer(b"ACGT", dtype=np.uint8)


class TargetResult:
    """Everything the kernels produced for one target."""

    __slots__ = ("name", "seq", "k", "n_ref", "kmers", "counts", "paths", "min_cov", "probes")

    def __init__(self, name, seq, k, n_ref, kmers, counts, paths, min_cov, probes=0):
        self.name, self.seq, self.k, self.n_ref = name, seq, k, n_ref
        self.kmers, self.counts, self.paths, self.min_cov = kmers, counts, paths, min_cov
        self.probes = probes

    def last_bases(self):
        return _LAST[(self.kmers & np.uint64(3)).astype(np.intp)]

    def spell(self, path, tails, whole_first=True):
        """Sequence of a node path: first k-mer (or only its last base) + last bases."""
        path = np.asarray(path, dtype=np.intp)
        if path.size == 0:
            return ""
        tail = tails[path[1:]].tobytes().decode()
        head = km.unpack(self.kmers[path[0]], self.k) if whole_first else chr(tails[path[0]])
        return head + tail


def split_paths(ref, alt, k):
    """Positions where `alt` leaves and rejoins `ref` (both index arrays).
    Returns (start, end_ref, end_var, end_ref_overlap)."""
    nr, na = len(ref), len(alt)
    m = min(nr, na)
    neq = np.flatnonzero(ref[:m] != alt[:m])
    start = int(neq[0]) if neq.size else m
    # common suffix, limited so that the two ends stay k apart from `start`
    room = min(nr, na) - (start + k) + 1
    if room > 0:
        tail = np.flatnonzero(ref[nr - room:][::-1] != alt[na - room:][::-1])
        same = int(tail[0]) if tail.size else room
    else:
        same = 0
    end_ref, end_var = nr - same, na - same
    # keep matching leftwards, overlap allowed, but never past `start` on the reference.
    # The reference indexes the variant with k_seq - 1, which may go negative and then
    # wraps like any Python index; numpy fancy indexing wraps the same way.
    room2 = end_ref - start
    if room2 > 0:
        reach = min(room2, end_var + na)          # beyond that the reference's index raises
        steps = np.arange(reach)
        tail = np.flatnonzero(ref[end_ref - 1 - steps] != alt[end_var - 1 - steps])
        if tail.size:
            more = int(tail[0])
        elif reach < room2:
            raise IndexError("list index out of range")
        else:
            more = room2
    else:
        more = 0
    return start, end_ref, end_var, end_ref - more


def name_variant(res, tails, ref, alt, offset=0):
    k = res.k
    start, end_ref, end_var, end_ovl = split_paths(ref, alt, k)
    only_ref, only_var = ref[start:end_ref], alt[start:end_var]
    if len(ref) - len(only_ref) + len(only_var) != len(alt):
        raise Exception("mutation identification could be incorrect")
    gone = res.spell(only_ref, tails, whole_first=False)
    new = res.spell(only_var, tails, whole_first=False)
    if gone:
        assert gone != new
        cut = 0
        while gone[-(cut + 1):] == new[-(cut + 1):]:
            cut += 1
        if cut:
            gone, new = gone[:-cut], new[:-cut]
    if end_ref == end_var:
        kind = "Reference" if start == end_ref else "Substitution"
    elif start == end_ovl:
        kind = "ITD"
    elif end_ref < end_var:
        kind = "Insertion" if not gone else "Indel"
    else:
        kind = "Deletion" if not new else "Indel"
    if kind == "Reference":
        return "Reference\t"
    return "%s\t%d:%s/%s:%d" % (kind, start + k + offset, gone.lower(), new, end_ref + 1 + offset)


def fit_paths(paths, counts_f32, n_total):
    """Least squares + projected gradient refinement of per-path expression.
    Returns (coef, rvaf); rvaf aliases coef when every coefficient is zero, as in
    the reference."""
    # occurrences of every node on every path (a tandem-duplication path visits nodes twice)
    contrib = np.empty((n_total, len(paths)), dtype=np.int32)
    for col, p in enumerate(paths):
        contrib[:, col] = np.bincount(p, minlength=n_total)
    coef = np.linalg.lstsq(contrib, counts_f32, rcond=None)[0]
    coef[coef < 0] = 0
    step = np.inf
    while step > 0.01:
        est = np.dot(contrib, coef)
        grad = 2 * (counts_f32 - est) * contrib.T
        grad = grad.sum(axis=1) / n_total
        coef += 0.1 * grad
        grad[coef < 0] = 0
        coef[coef < 0] = 0
        step = np.max(np.abs(grad))
    if max(coef) == 0:
        rvaf = coef
    else:
        rvaf = coef / np.sum(coef)
    return coef, rvaf


def format_row(db_name, query, name, rvaf, expr, min_cov, off, seq, ref_expr, ref_seq, note):
    return "%s\t%s\t%s\t%.3f\t%.1f\t%d\t%d\t%s\t%.1f\t%s\t%s" % (
        db_name, query, name, rvaf, expr, min_cov, off, seq, ref_expr, ref_seq, note)


def cluster_groups(diffs):
    """Group variants whose [start, end_ref] spans overlap; yields (lo, hi, members)."""
    todo = list(range(len(diffs)))

    def first_overlap(lo, hi):
        for v in todo:
            s, e = diffs[v][0], diffs[v][1]
            if e < lo or s > hi:
                continue
            if lo == hi == s == e:
                continue                    # terminal ITD, ignored in cluster mode
            if hi == e and (lo == hi or s == e):
                continue                    # quasi-terminal ITD
            return v
        return -1

    while todo:
        seed = todo.pop(0)
        lo, hi = diffs[seed][0], diffs[seed][1]
        members = [seed]
        v = first_overlap(lo, hi)
        while v != -1:
            todo.remove(v)
            members.append(v)
            lo, hi = min(lo, diffs[v][0]), max(hi, diffs[v][1])
            v = first_overlap(lo, hi)
        yield lo, hi, members


class _Desc:
    """Inverts the order of one sort-key component."""

    def __init__(self, v):
        self.v = v

    def __eq__(self, o):
        return self.v == o.v

    def __lt__(self, o):
        return self.v > o.v


def _natural(text):
    return [int(t) if t.isdigit() else t.lower() for t in re.split("([0-9]+)", text)]


def row_key(row):
    f = row.split("\t")
    comps = [_natural(x) for x in f[11].split(" ") + [f[1], f[3], f[2], f[6]]]
    return tuple([_Desc(comps[0])] + comps[1:])


def target_rows(res, db_name):
    """Sorted TSV rows of one target."""
    k, n_ref = res.k, res.n_ref
    n_total = len(res.kmers) + 2
    counts_f32 = np.empty(n_total, dtype=np.float32)
    counts_f32[:-2] = res.counts
    counts_f32[-2:] = -1
    tails = res.last_bases()
    ref = np.arange(n_ref, dtype=np.int64)
    ref_seq = res.seq[: n_ref + k - 1]
    rows = []
    paths = [np.asarray(p, dtype=np.int64) for p in res.paths]
    if len(paths) == 1 and paths[0].size == n_ref and (paths[0] == ref).all():
        # Only the reference path: PathQuant's answer is known in closed form.  Both
        # coefficients are positive as soon as one k-mer of the path has coverage, and
        # adjust_for_reference then overwrites them with min(counts incl. the -1 caps) = -1;
        # with zero coverage everywhere rVAF aliases coef and both print as nan
        # (km/utils/PathQuant.py:144-154).  No cluster is reported for the reference alone.
        expr = float("nan") if int(res.counts[:n_ref].max()) == 0 else -1.0
        return [format_row(db_name, res.name, "Reference\t", float("nan"), expr, res.min_cov[0], 0,
                           ref_seq, expr, ref_seq, "vs_ref")]
    for p, mc in zip(paths, res.min_cov):
        if p.size == n_ref and (p == ref).all():
            # the reference against itself: same closed form as above
            expr = float("nan") if int(res.counts[:n_ref].max()) == 0 else -1.0
            rows.append(format_row(db_name, res.name, "Reference\t", float("nan"), expr, mc, 0,
                                   ref_seq, expr, ref_seq, "vs_ref"))
            continue
        coef, rvaf = fit_paths([p, ref], counts_f32, n_total)
        rows.append(format_row(db_name, res.name, name_variant(res, tails, ref, p), rvaf[0], coef[0],
                               mc, 0, res.spell(p, tails), coef[1], ref_seq, "vs_ref"))
    if paths:
        diffs = [split_paths(ref, p, k) for p in paths]
        num = 0
        for lo, hi, members in cluster_groups(diffs):
            if len(members) == 1:
                p = paths[members[0]]
                if p.size == n_ref and (p == ref).all():
                    continue
            num += 1
            size = max(abs(diffs[v][2] - diffs[v][1] + 1) for v in members)
            off = max(0, lo - size)
            cref = ref[off:hi]
            clipped = [paths[v][off:diffs[v][2] + hi - diffs[v][1]] for v in members]
            coef, rvaf = fit_paths([cref] + clipped, counts_f32, n_total)
            cref_seq = res.spell(cref, tails)
            for p, rv, ce in zip(clipped, rvaf[1:], coef[1:]):
                mc = int(res.counts[p].min())
                rows.append(format_row(db_name, res.name, name_variant(res, tails, cref, p, off), rv, ce,
                                       mc, off, res.spell(p, tails), coef[0], cref_seq,
                                       "cluster %d n=%d" % (num, len(clipped))))
    rows.sort(key=row_key)
    return rows
