"""Batched counterpart of the reference's per-target loop.

km/tools/find_mutation.py:47-58 builds one ``MutationFinder`` per target and
runs ``graph_analysis`` on it, target after target.  :class:`BatchFinder` hands
ALL targets of a run to the GPU in one ``km_batch_run`` (walk + path search) and
then rebuilds, per target, the objects the host-side reporting needs.
"""

import numpy as np

from . import lib as _lib
from . import report


class NodeLimitExceeded(Exception):
    """MutationFinder.py:143-148 — the reference calls sys.exit with this text."""

    def __init__(self, max_node):
        self.max_node = max_node
        super().__init__("ERROR: Node query count limit exceeded: max=%d" % max_node)

    def __reduce__(self):
        # pickle rebuilds an exception from its args (the formatted message) by default; the ranks of
        # km_amd.dist gather row lists that may hold this one
        return (NodeLimitExceeded, (self.max_node,))


def repeated_kmer_message(seq, name, k):
    """The ValueError text of km/utils/common.py:55-59 for the first repeated k-mer."""
    seen = set()
    for pos in range(len(seq) - k + 1):
        mer = seq[pos:pos + k]
        if mer in seen:
            return "%s found multiple times in reference %s, at pos. %d" % (mer, name, pos)
        seen.add(mer)
    return None


class BatchFinder:
    def __init__(self, jf, max_stack=500, max_break=10, max_node=10000):
        self.jf = jf
        self.max_stack, self.max_break, self.max_node = max_stack, max_break, max_node
        self._batch = None
        self._cap = (0, 0)

    def _ensure(self, n_targets, n_bases):
        if self._batch is None or n_targets > self._cap[0] or n_bases > self._cap[1]:
            if self._batch is not None:
                self._batch.close()
            cap = (max(n_targets, 64), max(n_bases, 1 << 16))
            self._batch = _lib.Batch(self.jf.db, self.jf.cutoff, self.jf.n_cutoff, self.max_stack,
                                     self.max_break, self.max_node, cap[0], cap[1])
            self._cap = cap
        return self._batch

    def graph_log(self, n_targets):
        """What the reference logs with -v from inside the walk and the graph, for the last batch
        (km_batch_graph_log): (removed_ref_edges, nonref_edges, {target: [loop node, ...]}, n_loop_breaks)."""
        return self._batch.graph_log(n_targets)

    def run_raw(self, seqs, stream=None):
        """Walk + path search for a list of sequences; returns the raw result dict."""
        b = self._ensure(len(seqs), sum(len(s) for s in seqs))
        b.set_targets(seqs)
        b.run(stream=stream)
        return b.fetch()

    def _raise_input_errors(self, raw, names, seqs):
        # RefSeq construction comes first for every target in the reference
        # (km/tools/find_mutation.py:37-45): its errors pre-empt all output
        k = self.jf.k
        bad = np.nonzero(np.isin(raw["status"], (_lib.T_EMPTY, _lib.T_BAD_BASE, _lib.T_REPEAT_KMER, _lib.T_INTERNAL)))[0]
        if bad.size == 0:
            return
        t = int(bad[0])
        st = int(raw["status"][t])
        if st == _lib.T_EMPTY:
            exc = AssertionError("target %s is shorter than k=%d" % (names[t], k))
        elif st == _lib.T_BAD_BASE:
            exc = ValueError("target %s contains characters other than ACGT" % names[t])
        elif st == _lib.T_REPEAT_KMER:
            exc = ValueError(repeated_kmer_message(seqs[t], names[t], k))
        else:
            exc = RuntimeError("libkmgpu: internal workspace overflow on target %s" % names[t])
        exc.km_input_error = True            # raised before any row of its batch (km_amd.cli holds rows back for it)
        raise exc

    def rows(self, targets, db_name=None):
        """targets: list of (name, seq).  The TSV rows of every target through the native
        reporting path (km_report_rows): a list with, per target, the list of its row strings
        or the exception the reference would have raised at that target."""
        names = [t[0] for t in targets]
        seqs = [t[1] for t in targets]
        packed = _lib.pack_sequences(seqs)               # encoded once, for the GPU and for the report
        b = self._ensure(len(seqs), int(packed[1][-1]))
        b.set_targets_packed(*packed)
        # kernels + device-side compaction + one asynchronous D2H into the batch's pinned buffer;
        # `raw` are views into that buffer (no host reorganisation, no copy)
        b.run(_lib.KM_STAGE_WALK | _lib.KM_STAGE_GRAPH | _lib.KM_RUN_DELIVER | _lib.KM_DELIVER_LEAN)
        raw = b.result()
        self.last_raw = raw                          # views into the batch's pinned buffer (CLI -v)
        self._raise_input_errors(raw, names, seqs)
        out = _lib.report_rows(raw, names, seqs, self.jf.k, self.jf.filename if db_name is None else db_name,
                               packed=packed)
        for t, st in enumerate(raw["status"].tolist()):
            if st == _lib.T_NODE_LIMIT:
                out[t] = NodeLimitExceeded(self.max_node)
        return out

    def write_rows(self, targets, out, db_name=None):
        """`rows`, written straight to the stream `out`: when no target of the batch needs special
        handling the native text IS the TSV body and goes out in one write.  Stops like the
        reference at a node-limit exit / naming exception (after the rows of the earlier targets)."""
        names = [t[0] for t in targets]
        seqs = [t[1] for t in targets]
        packed = _lib.pack_sequences(seqs)
        b = self._ensure(len(seqs), int(packed[1][-1]))
        b.set_targets_packed(*packed)
        b.run(_lib.KM_STAGE_WALK | _lib.KM_STAGE_GRAPH | _lib.KM_RUN_DELIVER | _lib.KM_DELIVER_LEAN)
        raw = b.result()
        self.last_raw = raw
        self._raise_input_errors(raw, names, seqs)
        text, row_off, special = _lib.report_text(raw, names, seqs, self.jf.k,
                                                  self.jf.filename if db_name is None else db_name, packed=packed)
        limit = np.nonzero(raw["status"] == _lib.T_NODE_LIMIT)[0]
        if not special and limit.size == 0:
            out.write(text)
            return
        stops = sorted(set(special) | set(limit.tolist()))
        pos = 0
        for t in stops:
            out.write(text[int(row_off[pos]):int(row_off[t])])
            pos = t + 1
            if raw["status"][t] == _lib.T_NODE_LIMIT:
                out.flush()
                raise NodeLimitExceeded(self.max_node)
            blk = special[t]
            if isinstance(blk, BaseException):
                out.flush()
                raise blk
            for row in blk:
                out.write(row + "\n")
        out.write(text[int(row_off[pos]):])

    def analyse(self, targets):
        """targets: list of (name, seq).  Returns a list with one TargetResult per
        target, or the exception the reference would have raised at that target."""
        names = [t[0] for t in targets]
        seqs = [t[1] for t in targets]
        k = self.jf.k
        raw = self.run_raw(seqs)
        self._raise_input_errors(raw, names, seqs)
        out = []
        noff = raw["node_off"]
        poff = raw["path_off"]
        for t in range(len(targets)):
            if raw["status"][t] == _lib.T_NODE_LIMIT:
                out.append(NodeLimitExceeded(self.max_node))
                continue
            a, e = int(noff[t]), int(noff[t + 1])
            paths = [_lib.expand_path(raw, p) for p in range(int(poff[t]), int(poff[t + 1]))]
            mc = raw["path_min_cov"][int(poff[t]):int(poff[t + 1])].tolist()
            out.append(report.TargetResult(names[t], seqs[t], k, int(raw["n_ref"][t]),
                                           raw["node_kmer"][a:e], raw["node_count"][a:e], paths, mc,
                                           int(raw["probes"][t])))
        return out
