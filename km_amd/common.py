"""Drop-ins for the two helpers of km/utils/common.py that touch the k-mer database.

``get_cov(db, ref_seq)`` is what ``km min_cov`` (km/tools/min_cov.py:10-25) and
``km find_report -e <exclusion.jf>`` (km/tools/find_report.py:137-139) call: the coverage
statistics of every k-mer of a sequence.  The reference opens the database and issues one
``Jellyfish.query`` per k-mer on every call (km/utils/common.py:73-92); here a database is
loaded into HBM once per path and all k-mers of a call — or of MANY sequences, ``get_cov_many``
— go through one ``kmjf_query_batch`` launch.
"""

import numpy as np

from . import kmer as km
from .jellyfish import Jellyfish

_OPEN = {}


def _handle(db):
    if isinstance(db, Jellyfish):
        return db
    jf = _OPEN.get(db)
    if jf is None:
        jf = _OPEN[db] = Jellyfish(db)
    return jf


def close(db):
    """Free the HBM table of one database opened through get_cov (a path), if it is open."""
    jf = _OPEN.pop(db, None)
    if jf is not None:
        jf.db.close()


def close_all():
    for jf in _OPEN.values():
        jf.db.close()
    _OPEN.clear()


def _stats(seq, c):
    c = c.astype(np.int64)
    n = int(c.size)
    if n == 0:
        # the reference takes min() of an empty list here
        raise ValueError("min() arg is an empty sequence")
    total = int(c.sum())
    return (total, len(seq), int(c.min()), int(c.max()), float(total) / n, n, int((c == 0).sum()))


def get_cov(db, ref_seq):
    """(count, len(ref_seq), min, max, mean, kmer_nb, kmer_nb_0) — km/utils/common.py:73-92."""
    jf = _handle(db)
    return _stats(ref_seq, jf.query_seq(ref_seq))


def get_cov_many(db, seqs):
    """get_cov of every sequence of `seqs` with ONE lookup launch."""
    jf = _handle(db)
    k = jf.k
    parts, sizes = [], []
    for s in seqs:
        codes = km.encode(s)
        if (codes > 3).any():
            raise ValueError("non-ACGT character in sequence")
        kms = km.sliding_kmers(codes, k) if len(s) >= k else np.zeros(0, np.uint64)
        parts.append(kms)
        sizes.append(kms.size)
    flat = np.concatenate(parts) if parts else np.zeros(0, np.uint64)
    counts = jf.query_many(flat) if flat.size else np.zeros(0, np.uint32)
    out, pos = [], 0
    for s, n in zip(seqs, sizes):
        out.append(_stats(s, counts[pos:pos + n]))
        pos += n
    return out
