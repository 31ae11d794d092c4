"""GPU-backed drop-in for the reference's k-mer DB adapter.

Mirrors the duck-type the rest of km touches:
``Jellyfish(filename, cutoff, n_cutoff)`` with attributes ``k filename canonical
cutoff n_cutoff`` and methods ``query(seq) -> int`` and
``get_child(seq, forward=True) -> [str]`` — km/utils/Jellyfish.py:14-72.

The scalar methods exist for drop-in use and tests; the hot path uses the
batched entry points (:meth:`query_many`, :meth:`children_many`, and
:class:`km_amd.finder.BatchFinder`) so that one launch serves a whole run.
"""

import os

import numpy as np

from . import kmer as km
from . import lib as _lib


def default_device():
    """Device ordinal: KM_DEVICE, else LOCAL_RANK (one process per GPU), else 0."""
    for var in ("KM_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(var)
        if v not in (None, ""):
            return int(v)
    return 0


class Jellyfish:
    def __init__(self, filename, cutoff=0.30, n_cutoff=500, device=None, db=None):
        self.filename = filename
        self.cutoff = cutoff
        self.n_cutoff = n_cutoff
        self.device = default_device() if device is None else int(device)
        # file -> HBM directly (no host copy of the records); a caller-provided db is used as is
        self.db = db if db is not None else _lib.Database.load(filename, self.device)
        info = self.db.info
        if info.n_slots == 0:
            self.db.upload(self.device)
            info = self.db.info
        self.k = int(info.k)
        self.canonical = bool(info.canonical)

    # ---- reference-shaped scalar API ------------------------------------------------
    def query(self, seq):
        """km/utils/Jellyfish.py:47-53."""
        return int(self.db.query(np.array([km.pack_str(seq)], dtype=np.uint64))[0])

    def get_child(self, seq, forward=True):
        """km/utils/Jellyfish.py:55-72."""
        mask, _ = self.db.children(np.array([km.pack_str(seq)], dtype=np.uint64),
                                   self.cutoff, self.n_cutoff, forward)
        out = []
        for c, base in enumerate(km.BASES):
            if (int(mask[0]) >> c) & 1:
                out.append((seq[1:] + base) if forward else (base + seq[:-1]))
        return out

    # ---- batched API ------------------------------------------------------------------
    def query_many(self, packed):
        return self.db.query(packed)

    def query_seq(self, seq):
        """Counts of every k-mer of `seq` (the loop of common.get_cov,
        km/utils/common.py:73-92) in one launch."""
        codes = km.encode(seq)
        if (codes > 3).any():
            raise ValueError("non-ACGT character in sequence")
        return self.db.query(km.sliding_kmers(codes, self.k))

    def children_many(self, packed, forward=True):
        return self.db.children(packed, self.cutoff, self.n_cutoff, forward)
