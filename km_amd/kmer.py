"""2-bit packed k-mer helpers (host side, numpy).

Encoding follows the Jellyfish key layout the reference's DB files use
(A=0, C=1, G=2, T=3; first base in the most significant used bits), so packed
values can be compared directly with `.jf` record keys.
"""

import numpy as np

BASES = "ACGT"
_LUT = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate(BASES):
    _LUT[ord(_c)] = _i
    _LUT[ord(_c.lower())] = _i


def encode(seq):
    """ASCII DNA -> uint8 base codes (0..3; 4 = not ACGT)."""
    if isinstance(seq, str):
        seq = seq.encode("ascii")
    return _LUT[np.frombuffer(seq, dtype=np.uint8)]


def decode(codes):
    return bytes(np.frombuffer(b"ACGTN", dtype=np.uint8)[np.asarray(codes, dtype=np.uint8)]).decode()


def pack_str(seq):
    v = 0
    for c in encode(seq).tolist():
        v = (v << 2) | c
    return v


def unpack(v, k):
    v = int(v)
    return "".join(BASES[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))


def sliding_kmers(codes, k):
    """All k-mers of a code row / matrix of rows: (..., L) uint8 -> (..., L-k+1) uint64."""
    codes = np.asarray(codes, dtype=np.uint8)
    n = codes.shape[-1] - k + 1
    if n <= 0:
        return np.zeros(codes.shape[:-1] + (0,), dtype=np.uint64)
    out = np.zeros(codes.shape[:-1] + (n,), dtype=np.uint64)
    for j in range(k):
        out = (out << np.uint64(2)) | codes[..., j:j + n].astype(np.uint64)
    return out


def revcomp(keys, k):
    x = ~np.asarray(keys, dtype=np.uint64)
    m2 = np.uint64(0x3333333333333333)
    m4 = np.uint64(0x0F0F0F0F0F0F0F0F)
    x = ((x >> np.uint64(2)) & m2) | ((x & m2) << np.uint64(2))
    x = ((x >> np.uint64(4)) & m4) | ((x & m4) << np.uint64(4))
    x = x.byteswap()
    return x >> np.uint64(64 - 2 * k)


def canonical(keys, k):
    keys = np.asarray(keys, dtype=np.uint64)
    return np.minimum(keys, revcomp(keys, k))
