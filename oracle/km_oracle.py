"""TEST INFRASTRUCTURE — CPU oracle, not product code.

Structure-faithful CPU restatement of the reference's ``find_mutation`` path
(string k-mers, one table lookup per ``query``, recursive extension, dense
float32 Dijkstra in numpy), written from the reference's *behaviour*:

    KmerDB.query / get_child      <- km/utils/Jellyfish.py:47-53, 55-72
    walk (register + extend)      <- km/utils/MutationFinder.py:100-124, 137-165
    ref_kmers                     <- km/utils/common.py:48-63
    build_graph / dijkstra_prev   <- km/utils/MutationFinder.py:508-557,
                                     km/utils/Graph.py:41-61, 63-119
    strip_ref_edges               <- km/utils/Graph.py:121-198
    enumerate_paths               <- km/utils/Graph.py:200-240
    path_diff / variant_name      <- km/utils/MutationFinder.py:190-373, 405-488
    PathFit (lstsq + projected GD)<- km/utils/PathQuant.py:93-154
    clusters / rows / sort        <- km/utils/MutationFinder.py:575-648, 651-833
    run_find_mutation             <- km/tools/find_mutation.py:17-60

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product (``km_amd``) never does.

CANONICAL ORDER.  The reference iterates a Python ``set`` of k-mer *strings*
(MutationFinder.py:100,111,115) and a ``set`` of index tuples (Graph.py:233-240),
so node numbering, the order of alternative paths and therefore the
``cluster N`` numbering of its TSV change with PYTHONHASHSEED.  This oracle
fixes one admissible order: reference k-mers are registered and extended in
target order (node i == i-th k-mer of the target), walk-discovered nodes follow
in registration order, and alternative paths are sorted by their node-index
tuple.  Everything else follows the reference statement by statement.

PARITY PIN.  Checked in tests/test_oracle_golden.py against golden vectors
produced by the unmodified reference (tests/golden/make_golden.py): exact TSV
on every seed-stable case (all bundled NPM1/FLT3/DNMT3A fixtures, catalog x 5
DBs, synthetic slices), node sets / path sequences / min coverages on all
cases, and TSV equality modulo cluster renumbering where the reference itself
is seed-dependent.
"""

import os
import re
import sys

import numpy as np

from . import jf_reader as jr

sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))

SOURCE = "BigBang"      # capping nodes (MutationFinder.py:97-98)
SINK = "BigCrunch"


class NodeLimit(Exception):
    """len(node_data) > max_node at an extension call (MutationFinder.py:143-148)."""


class RepeatedKmer(ValueError):
    pass


# --------------------------------------------------------------------------- DB
class KmerDB:
    """Dict-backed k-mer count table with the reference adapter's interface."""

    def __init__(self, filename=None, cutoff=0.30, n_cutoff=500, records=None):
        if records is None:
            records = jr.read_jf(filename)
        self.filename = filename
        self.k = int(records["k"])
        self.canonical = bool(records["canonical"])
        self.cutoff = cutoff
        self.n_cutoff = n_cutoff
        self.table = dict(zip(np.asarray(records["keys"]).tolist(),
                              np.asarray(records["counts"]).tolist()))
        self.probes = 0          # logical probes: one per query() call

    def query(self, seq):
        self.probes += 1
        v = jr.pack(seq)
        if self.canonical:
            v = jr.canonical(v, len(seq))
        return self.table.get(v, 0)

    def get_child(self, seq, forward=True):
        scored = []
        total = 0
        for base in "ACGT":
            nxt = (seq[1:] + base) if forward else (base + seq[:-1])
            c = self.query(nxt)
            scored.append((nxt, c))
            total += c
        floor = max(total * self.cutoff, self.n_cutoff)
        return [s for s, c in scored if c >= floor]


# ------------------------------------------------------------------------- walk
def ref_kmers(seq, name, k):
    seen = set()
    out = []
    for pos in range(len(seq) - k + 1):
        mer = seq[pos:pos + k]
        if mer in seen:
            raise RepeatedKmer("%s found multiple times in reference %s, at pos. %d"
                               % (mer, name, pos))
        seen.add(mer)
        out.append(mer)
    return out


def walk(ref_mers, db, max_stack=500, max_break=10, max_node=10000, loops=None):
    """Register the target's k-mers, then depth-first extend from each of them.
    Returns the ordered {kmer: count} node dict.  `loops` (a list): the k-mers at which the
    reference logs 'Broke loop at kmer' (MutationFinder.py:160-161), in walk order."""
    nodes = {}
    for mer in ref_mers:                       # canonical order: target order
        nodes[mer] = db.query(mer)

    def extend(stack, breaks):
        if len(stack) > max_stack:
            return
        if len(nodes) > max_node:
            raise NodeLimit(max_node)
        kids = db.get_child(stack[-1], forward=True)
        if len(kids) > 1:
            breaks += 1
            if breaks > max_break:
                return
        for kid in kids:
            if kid in nodes or kid in stack:
                if loops is not None and kid in stack and kid not in nodes:
                    loops.append(kid)
                for mer in stack:
                    nodes[mer] = db.query(mer)
            else:
                extend(stack + [kid], breaks)

    for mer in ref_mers:
        extend([mer], 0)
    return nodes


# ------------------------------------------------------------------------ graph
def dijkstra_prev(w, start):
    """Predecessor array of the reference's dense float32 Dijkstra."""
    n = w.shape[0]
    prev = np.full(n, -1, dtype=np.int32)
    dist = np.full(n, np.inf, dtype=np.float32)
    todo = np.ones(n, dtype=bool)
    dist[start] = 0
    for _ in range(n):
        cand = np.flatnonzero(todo)
        i = cand[np.argmin(dist[cand])]        # first index of the minimum
        nd = w[i, :] + dist[i]
        better = nd < dist
        dist[better] = nd[better]
        prev[better] = i
        todo[i] = False
    return prev


def build_graph(kmers, ref_index, start_ix, end_ix):
    """Weight matrix + edge set over `kmers` (last two entries = caps)."""
    n = len(kmers)
    w = np.full((n, n), np.inf, dtype=np.float32)
    edges = set()

    def put(i, j, val):
        w[i, j] = val
        edges.add((i, j))

    by_prefix = {}
    for i, mer in enumerate(kmers):
        by_prefix.setdefault(mer[:-1], []).append(i)
    for i, mer in enumerate(kmers):
        for j in by_prefix.get(mer[1:], ()):
            if i != j:
                put(i, j, 1)
    for a, b in zip(ref_index[:-1], ref_index[1:]):
        put(a, b, 0.01)
    put(n - 2, start_ix, 0.01)
    put(end_ix, n - 1, 0.01)
    return w, edges


def strip_ref_edges(edges, before, after, source):
    """Drop from `edges` every edge but the first along the sink-tree chain of
    each node whose predecessor is the source (Graph.py:184-197, including the
    ``if last_cur`` truthiness test that skips index 0)."""
    removed = 0
    for cur in sorted(set(np.flatnonzero(before == source).tolist())):
        last = None
        while after[cur] != -1:
            cur = int(after[cur])
            if last and (last, cur) in edges:
                edges.remove((last, cur))
                removed += 1
            last = cur
    return removed


def enumerate_paths(edges, before, after, source, sink):
    found = set()
    for a, b in edges:
        left = [a]
        while before[left[-1]] != -1:
            left.append(int(before[left[-1]]))
        right = [b]
        while after[right[-1]] != -1:
            right.append(int(after[right[-1]]))
        if left[-1] != source or right[-1] != sink:
            continue
        found.add(tuple(reversed(left)) + tuple(right))
    return sorted(found)                       # canonical order (see header)


def graph_paths(kmers, n_ref, stats=None):
    """All source->sink paths through a non-reference edge, caps stripped.
    `kmers`: node list (ref k-mers first, in target order) WITHOUT caps.  `stats` (a dict): what the
    reference logs on the way — 'Removed %d ref edges.' (Graph.py:198) and '%d edges in non-ref edge
    set.' (Graph.py:231) — in OUR node order (the reference's own numbers move by one with its hash
    seed: `if last_cur` skips whichever node happens to have index 0)."""
    names = list(kmers) + [SOURCE, SINK]
    n = len(names)
    ref_index = list(range(n_ref))
    w, edges = build_graph(names, ref_index, 0, n_ref - 1)
    before = dijkstra_prev(w, n - 2)
    after = dijkstra_prev(w.transpose(), n - 1)
    removed = strip_ref_edges(edges, before, after, n - 2)
    if stats is not None:
        stats["removed_ref_edges"] = removed
        stats["nonref_edges"] = len(edges)
    return [p[1:-1] for p in enumerate_paths(edges, before, after, n - 2, n - 1)]


# ----------------------------------------------------------------------- naming
def path_diff(ref, alt, k):
    """(start, end_ref, end_var, kmers_ref, kmers_var, end_ref_overlap)."""
    i = 0
    while i < len(ref) and i < len(alt) and ref[i] == alt[i]:
        i += 1
    jr_, ja = len(ref), len(alt)
    while jr_ >= i + k and ja >= i + k and ref[jr_ - 1] == alt[ja - 1]:
        jr_ -= 1
        ja -= 1
    kr, ka = jr_, ja
    while kr > i and ref[kr - 1] == alt[ka - 1]:
        kr -= 1
        ka -= 1
    return i, jr_, ja, ref[i:jr_], alt[i:ja], kr


def spell(kmers, path, whole_first):
    if not path:
        return ""
    head = kmers[path[0]] if whole_first else kmers[path[0]][-1]
    return head + "".join(kmers[i][-1] for i in path[1:])


def variant_name(kmers, ref, alt, k, offset=0):
    start, end_ref, end_var, only_ref, only_var, end_ovl = path_diff(ref, alt, k)
    if len(ref) - len(only_ref) + len(only_var) != len(alt):
        raise Exception("mutation identification could be incorrect")
    gone = spell(kmers, only_ref, False)
    new = spell(kmers, only_var, False)
    cut = 1
    if gone:
        assert gone != new
        while gone[-cut:] == new[-cut:]:
            cut += 1
    cut -= 1
    if cut:
        gone, new = gone[:-cut], new[:-cut]
    if end_ref == end_var:
        kind = "Reference" if start == end_ref else "Substitution"
    elif start == end_ovl:
        kind = "ITD"
    else:
        kind = "Indel"
        if end_ref < end_var:
            if not gone:
                kind = "Insertion"
        elif not new:
            kind = "Deletion"
    if kind == "Reference":
        return kind + "\t"
    return "%s\t%d:%s:%d" % (kind, start + k + offset, gone.lower() + "/" + new,
                             end_ref + 1 + offset)


# --------------------------------------------------------------- quantification
class PathFit:
    def __init__(self, paths, counts):
        self.n = len(counts)
        self.counts = np.array(counts, dtype=np.float32)
        self.contrib = np.zeros((self.n, len(paths)), dtype=np.int32)
        for col, p in enumerate(paths):
            for i in p:
                self.contrib[i, col] += 1
        self.coef = np.linalg.lstsq(self.contrib, self.counts, rcond=None)[0]
        self.coef[self.coef < 0] = 0
        step = np.inf
        while step > 0.01:
            est = np.dot(self.contrib, self.coef)
            g = 2 * (self.counts - est) * self.contrib.T
            g = g.sum(axis=1) / self.n
            self.coef += 0.1 * g
            g[self.coef < 0] = 0
            self.coef[self.coef < 0] = 0
            step = np.max(np.abs(g))
        if max(self.coef) == 0:
            self.rvaf = self.coef               # aliasing is intentional (PathQuant.py:145-146)
        else:
            self.rvaf = self.coef / np.sum(self.coef)

    def as_reference(self):
        self.rvaf[0] = np.nan
        self.rvaf[1] = np.nan
        self.coef[self.coef >= 0] = min(self.counts)


def row_text(db_name, query, name, rvaf, expr, min_cov, off, seq, ref_expr, ref_seq, note):
    return "%s\t%s\t%s\t%.3f\t%.1f\t%d\t%d\t%s\t%.1f\t%s\t%s" % (
        db_name, query, name, rvaf, expr, min_cov, off, seq, ref_expr, ref_seq, note)


class _Rev:
    def __init__(self, v):
        self.v = v

    def __eq__(self, o):
        return o.v == self.v

    def __lt__(self, o):
        return self.v > o.v


def _nat(s):
    return [int(t) if t.isdigit() else t.lower() for t in re.split("([0-9]+)", s)]


def row_sort_key(text):
    f = text.split("\t")
    parts = f[11].split(" ") + [f[1], f[3], f[2], f[6]]
    keyed = [_nat(p) for p in parts]
    return tuple([_Rev(keyed[0])] + keyed[1:])


def find_clusters(paths, ref, k):
    diffs = [path_diff(ref, p, k) for p in paths]
    todo = list(range(len(paths)))             # ascending == CPython small-int set order

    def overlapping(lo, hi):
        for v in todo:
            s, e = diffs[v][0], diffs[v][1]
            if e >= lo and s <= hi:
                if lo == hi == s == e:
                    continue
                if hi == e and (lo == hi or s == e):
                    continue
                return v
        return -1

    groups = []
    while todo:
        seed = todo.pop(0)
        grp = [seed]
        lo, hi = diffs[seed][0], diffs[seed][1]
        v = overlapping(lo, hi)
        while v != -1:
            todo.remove(v)
            grp.append(v)
            lo = min(lo, diffs[v][0])
            hi = max(hi, diffs[v][1])
            v = overlapping(lo, hi)
        groups.append((lo, hi, grp))
    for lo, hi, grp in groups:
        if len(grp) == 1 and tuple(paths[grp[0]]) == tuple(ref):
            continue
        size = max(abs(diffs[v][2] - diffs[v][1] + 1) for v in grp)
        off = max(0, lo - size)
        clipped = [tuple(paths[v][off:diffs[v][2] + hi - diffs[v][1]]) for v in grp]
        yield tuple(ref[off:hi]), clipped, off


# ---------------------------------------------------------------------- drivers
def analyse_target(seq, name, db, max_stack=500, max_break=10, max_node=10000):
    """One target through walk + graph.  Returns a dict of the hot path's
    outputs (what the GPU kernels must reproduce bit for bit)."""
    k = db.k
    mers = ref_kmers(seq, name, k)
    assert len(mers)
    p0 = db.probes
    loops = []
    nodes = walk(mers, db, max_stack, max_break, max_node, loops)
    probes = db.probes - p0
    kmers = list(nodes.keys())
    counts = list(nodes.values())
    stats = {}
    paths = graph_paths(kmers, len(mers), stats)
    return {"name": name, "k": k, "n_ref": len(mers), "kmers": kmers, "counts": counts,
            "paths": paths, "probes": probes, "loop_kmers": loops,
            "removed_ref_edges": stats["removed_ref_edges"], "nonref_edges": stats["nonref_edges"],
            "min_cov": [min(counts[i] for i in p) for p in paths]}


def target_rows(res, db_name):
    """TSV rows of one analysed target, sorted as the reference prints them."""
    k = res["k"]
    kmers = res["kmers"] + [SOURCE, SINK]
    counts = res["counts"] + [-1, -1]
    ref = tuple(range(res["n_ref"]))
    ref_seq = spell(kmers, ref, True)
    rows = []
    for p in res["paths"]:
        fit = PathFit([p, ref], counts)
        if tuple(p) == ref:
            fit.as_reference()
        rows.append(row_text(db_name, res["name"], variant_name(kmers, ref, p, k),
                             fit.rvaf[0], fit.coef[0], min(counts[i] for i in p), 0,
                             spell(kmers, p, True), fit.coef[1], ref_seq, "vs_ref"))
    if res["paths"]:
        for num, (cref, clipped, off) in enumerate(find_clusters(res["paths"], ref, k), 1):
            fit = PathFit([cref] + clipped, counts)
            for p, rv, ce in zip(clipped, fit.rvaf[1:], fit.coef[1:]):
                assert p != cref
                rows.append(row_text(db_name, res["name"], variant_name(kmers, cref, p, k, off),
                                     rv, ce, min(counts[i] for i in p), off,
                                     spell(kmers, p, True), fit.coef[0], spell(kmers, cref, True),
                                     "cluster %d n=%d" % (num, len(clipped))))
    return sorted(rows, key=row_sort_key)


HEADER = "\t".join(["Database", "Query", "Type", "Variant_name", "rVAF", "Expression",
                    "Min_coverage", "Start_offset", "Sequence", "Reference_expression",
                    "Reference_sequence", "Info"])


def read_fasta_concat(path):
    """All records of a FASTA file concatenated and upper-cased
    (km/utils/common.py:25-45, km/tools/find_mutation.py:39-43)."""
    chunks = []
    with open(path) as fh:
        for line in fh:
            if not line.startswith(">"):
                chunks.append(line.strip())
    return "".join(chunks).upper()


def run_find_mutation(targets, db_path, count=5, ratio=0.05, steps=500, branchs=10,
                      nodes=10000):
    """Lines the reference driver prints (minus the ``#Elapsed time`` trailer),
    plus the sys.exit message if the node limit stops the run."""
    lines = ["#count:%s" % count, "#ratio:%s" % ratio, "#steps:%s" % steps,
             "#branchs:%s" % branchs, "#nodes:%s" % nodes, "#graphical:False",
             "#verbose:False", "#debug:False", "#target_fn:%s" % (list(targets),),
             "#jellyfish_fn:%s" % db_path]
    db = KmerDB(db_path, cutoff=ratio, n_cutoff=count)
    lines.append(HEADER)
    seqs = []
    for t in targets:
        name = os.path.splitext(os.path.basename(t))[0]
        seq = read_fasta_concat(t)
        ref_kmers(seq, name, db.k)              # the driver builds every RefSeq first
        seqs.append((seq, name))
    for seq, name in seqs:
        try:
            res = analyse_target(seq, name, db, steps, branchs, nodes)
        except NodeLimit:
            return lines, "ERROR: Node query count limit exceeded: max=%d" % nodes
        lines.extend(target_rows(res, db_path))
    return lines, None


def coverage(db_path, seq):
    """(sum, len(seq), min, max, mean, n, n_zero) <- km/utils/common.py:73-92."""
    db = KmerDB(db_path)
    c = [int(db.query(seq[i:i + db.k])) for i in range(len(seq) - db.k + 1)]
    mean = float(sum(c)) / len(c) if c else 0
    return sum(c), len(seq), min(c), max(c), mean, len(c), sum(1 for x in c if x == 0)
