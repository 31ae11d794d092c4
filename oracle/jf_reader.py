"""TEST INFRASTRUCTURE — CPU oracle, not product code.

Pure-numpy reader/writer for Jellyfish ``binary/sorted`` k-mer count files.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path (``km_amd``) has its own native
reader in ``km_amd/csrc/jf_reader.cpp`` and never touches this file.

Third-party dependency restated here: **Jellyfish** (gmarcais/Jellyfish; the
reference pins it loosely as ``pyjellyfish>=1.3.0`` in pyproject.toml:10, CI
builds v2.2.6 in .travis.yml:20-22; the bundled fixtures were written by
v2.2.3).  Its source is NOT under /root/reference.  The on-disk layout
restated below is the published ``binary/sorted`` dumper format:

    bytes 0..8      ASCII decimal N, zero padded to 9 digits
    bytes 9..9+N    JSON header, padded so that 9+N is a multiple of
                    header["alignment"]
    then            fixed-size records  [key_bytes LE key][counter_len LE count]
                    key_bytes = ceil(key_len / 8), key_len = 2*k bits

Key encoding: A=0, C=1, G=2, T=3, two bits per base, first base in the most
significant used bits.  With ``canonical: true`` a k-mer is stored as
min(key, revcomp(key)).

Parity pin: this decoding reproduces every number of the reference's own
pure-lookup known-answer test (km/tests/test_main.py:581-652, ``test_min_cov``)
— see tests/test_oracle_golden.py.

Reference call sites this replaces: km/utils/Jellyfish.py:24-25 (open file,
global k) and :29-45 (header scan for ``canonical``).
"""

import json

import numpy as np

_CODE = {"A": 0, "C": 1, "G": 2, "T": 3}
_BASES = "ACGT"


def parse_header(raw):
    """Return (header_dict, payload_offset) for the bytes of a .jf file."""
    n = int(raw[:9].decode("ascii"))
    text = raw[9:9 + n].decode("ascii", errors="ignore")
    # the JSON object is followed by NUL / space padding up to `alignment`
    end = text.rindex("}") + 1
    return json.loads(text[:end]), 9 + n


def read_jf(path):
    """Read a ``binary/sorted`` file.

    Returns dict(k, canonical, keys=uint64[n], counts=uint32[n], header).
    """
    with open(path, "rb") as fh:
        raw = fh.read()
    header, off = parse_header(raw)
    if header.get("format") != "binary/sorted":
        raise ValueError("unsupported Jellyfish format %r" % header.get("format"))
    key_len = int(header["key_len"])
    if key_len % 2 or key_len > 64:
        raise ValueError("unsupported key_len %d" % key_len)
    kb = (key_len + 7) // 8
    cb = int(header["counter_len"])
    if cb > 4:
        raise ValueError("unsupported counter_len %d" % cb)
    rec = kb + cb
    body = np.frombuffer(raw, dtype=np.uint8, offset=off)
    n = body.size // rec
    body = body[: n * rec].reshape(n, rec)
    keys = np.zeros(n, dtype=np.uint64)
    for b in range(kb):
        keys |= body[:, b].astype(np.uint64) << np.uint64(8 * b)
    counts = np.zeros(n, dtype=np.uint32)
    for b in range(cb):
        counts |= body[:, kb + b].astype(np.uint32) << np.uint32(8 * b)
    return {
        "k": key_len // 2,
        "canonical": bool(header["canonical"]),
        "keys": keys,
        "counts": counts,
        "header": header,
    }


def write_jf(path, keys, counts, k, canonical=True):
    """Write a ``binary/sorted``-shaped file that :func:`read_jf` (and the
    product's native reader) can load.  Record order is by key, not by
    Jellyfish's matrix hash position, so real Jellyfish could not binary-search
    it — loaders that read *all* records (ours) do not care."""
    keys = np.asarray(keys, dtype=np.uint64)
    counts = np.asarray(counts, dtype=np.uint32)
    header = {
        "alignment": 8,
        "canonical": bool(canonical),
        "counter_len": 4,
        "format": "binary/sorted",
        "key_len": 2 * k,
        "val_len": 12,
        "size": int(max(16, 1 << int(np.ceil(np.log2(max(2, 2 * len(keys))))))),
        "cmdline": ["km_amd-synthetic"],
    }
    text = json.dumps(header, separators=(",", ":")).encode("ascii")
    pad = (-(9 + len(text))) % 8
    text += b"\0" * pad
    kb = (2 * k + 7) // 8
    rec = np.zeros((len(keys), kb + 4), dtype=np.uint8)
    for b in range(kb):
        rec[:, b] = ((keys >> np.uint64(8 * b)) & np.uint64(0xFF)).astype(np.uint8)
    for b in range(4):
        rec[:, kb + b] = ((counts >> np.uint32(8 * b)) & np.uint32(0xFF)).astype(np.uint8)
    with open(path, "wb") as fh:
        fh.write(b"%09d" % len(text))
        fh.write(text)
        fh.write(rec.tobytes())


def pack(seq):
    """2-bit pack a DNA string (first base most significant)."""
    v = 0
    for ch in seq:
        v = (v << 2) | _CODE[ch]
    return v


def unpack(v, k):
    v = int(v)
    return "".join(_BASES[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))


def revcomp(v, k):
    """Reverse complement of a packed k-mer (plain Python ints)."""
    v = int(v)
    r = 0
    for _ in range(k):
        r = (r << 2) | (3 - (v & 3))
        v >>= 2
    return r


def canonical(v, k):
    r = revcomp(v, k)
    return v if v < r else r


def revcomp_np(keys, k):
    """Vectorised reverse complement of uint64 packed k-mers."""
    x = ~np.asarray(keys, dtype=np.uint64)
    m2 = np.uint64(0x3333333333333333)
    m4 = np.uint64(0x0F0F0F0F0F0F0F0F)
    x = ((x >> np.uint64(2)) & m2) | ((x & m2) << np.uint64(2))
    x = ((x >> np.uint64(4)) & m4) | ((x & m4) << np.uint64(4))
    x = x.byteswap()
    return x >> np.uint64(64 - 2 * k)


def canonical_np(keys, k):
    keys = np.asarray(keys, dtype=np.uint64)
    return np.minimum(keys, revcomp_np(keys, k))
