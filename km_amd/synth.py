"""Seeded synthetic workloads: random targets, a
canonical k-mer count table with injected variants / noise / padding.

The generator is the build's own; nothing here comes from the reference.  It is
used by bench.py (full size), by the GPU parity tests and by
tests/golden/make_golden.py (small slices that are also run through the
reference to produce committed golden TSVs).
"""

import hashlib
import json
import os

import numpy as np

from . import kmer as km

HEADLINE_SEED = 20260101


def _unique_kmer_rows(rng, n, length, k):
    """n random ACGT rows of `length` with no repeated k-mer inside a row
    (the reference raises ValueError on those: km/utils/common.py:55-59)."""
    rows = rng.integers(0, 4, size=(n, length), dtype=np.uint8)
    while True:
        kms = np.sort(km.sliding_kmers(rows, k), axis=1)
        bad = np.nonzero((kms[:, 1:] == kms[:, :-1]).any(axis=1))[0]
        if bad.size == 0:
            return rows
        rows[bad] = rng.integers(0, 4, size=(bad.size, length), dtype=np.uint8)


def _mutate(rng, row, kind, k):
    """Return (mutated_row, site_lo, site_hi): ref bases [site_lo, site_hi) are
    replaced.  Sites keep k-1 flanking bases on both sides so the variant path
    rejoins the target."""
    L = row.size
    lo_ok, hi_ok = k, L - k
    if kind == "snv":
        p = int(rng.integers(lo_ok, hi_ok))
        b = (int(row[p]) + int(rng.integers(1, 4))) % 4
        return np.concatenate([row[:p], [b], row[p + 1:]]).astype(np.uint8), p, p + 1
    if kind == "ins":
        n = int(rng.integers(1, 31))
        p = int(rng.integers(lo_ok, hi_ok))
        ins = rng.integers(0, 4, size=n, dtype=np.uint8)
        return np.concatenate([row[:p], ins, row[p:]]).astype(np.uint8), p, p
    if kind == "del":
        n = int(rng.integers(1, 31))
        p = int(rng.integers(lo_ok, max(lo_ok + 1, hi_ok - n)))
        return np.concatenate([row[:p], row[p + n:]]).astype(np.uint8), p, p + n
    if kind == "dup":  # tandem duplication of row[p:p+n] inserted right after itself
        n = int(rng.integers(20, 101))
        n = min(n, L - 2 * k - 1)
        p = int(rng.integers(lo_ok, max(lo_ok + 1, hi_ok - n)))
        return np.concatenate([row[:p + n], row[p:p + n], row[p + n:]]).astype(np.uint8), p + n, p + n
    raise ValueError(kind)


def make_case(n_targets=100, length=500, k=31, n_keys=200_000, seed=HEADLINE_SEED,
              variant_frac=0.30, variants_per_target=(1, 1), vaf=(0.1, 0.6),
              kinds=("snv", "ins", "del", "dup"), noise_frac=0.01, noise_counts=(2, 5),
              cov=(50, 2000), hom_frac=0.0, branch_noise_frac=0.0, name=None,
              exact_pad=True, canonical=True, heavy_frac=0.0, **_):
    """Build targets + (keys, counts).  Returns dict(targets=uint8[n,L] codes,
    names, keys uint64 (canonical, distinct), counts uint32, k)."""
    rng = np.random.default_rng(seed)
    rows = _unique_kmer_rows(rng, n_targets, length, k)
    base_cov = rng.integers(cov[0], cov[1], size=n_targets)
    ref_km = km.sliding_kmers(rows, k)                      # (T, n_ref)
    n_ref = ref_km.shape[1]
    jitter = rng.uniform(0.9, 1.1, size=ref_km.shape)
    ref_cnt = np.maximum(1, np.rint(base_cov[:, None] * jitter)).astype(np.int64)

    add_keys, add_cnts, add_tids = [], [], []
    has_var = rng.random(n_targets) < variant_frac
    for t in np.nonzero(has_var)[0]:
        nv = int(rng.integers(variants_per_target[0], variants_per_target[1] + 1))
        # heavy_frac: that share of the variant targets carries 3-5 tandem duplications instead (long
        # walks: the large tier of k_dfs / k_graph).  Nothing is drawn for it when it is 0, so that the
        # cases the goldens were made from stay what they are.
        heavy = bool(heavy_frac) and rng.random() < heavy_frac
        if heavy:
            nv = int(rng.integers(3, 6))
        for _v in range(nv):
            kind = "dup" if heavy else kinds[int(rng.integers(0, len(kinds)))]
            mut, lo, hi = _mutate(rng, rows[t], kind, k)
            f = float(rng.uniform(vaf[0], vaf[1]))
            if hom_frac and rng.random() < hom_frac:
                f = 1.0
            mk = km.sliding_kmers(mut, k)
            alt = mk[~np.isin(mk, ref_km[t])]
            if alt.size == 0:
                continue
            c = np.maximum(0, np.rint(base_cov[t] * f * rng.uniform(0.9, 1.1, size=alt.size)))
            add_keys.append(alt)
            add_cnts.append(c.astype(np.int64))
            add_tids.append(np.full(alt.size, t, dtype=np.int32))
            # ref k-mers spanning the replaced site lose the variant's share
            s0, s1 = max(0, lo - k + 1), min(n_ref, max(hi, lo + 1))
            span = np.arange(s0, s1)
            span = span[~np.isin(ref_km[t, span], mk)]
            ref_cnt[t, span] = np.rint(ref_cnt[t, span] * (1.0 - f)).astype(np.int64)

    # sibling noise: same k-1 prefix, different last base
    def siblings(frac, lo_c, hi_c, scale_by_cov):
        sel = rng.random(ref_km.shape) < frac
        tt, pp = np.nonzero(sel)
        if tt.size == 0:
            return
        sib = (ref_km[tt, pp] & ~np.uint64(3)) | (
            (ref_km[tt, pp] + rng.integers(1, 4, size=tt.size).astype(np.uint64)) & np.uint64(3))
        if scale_by_cov:
            c = np.maximum(1, np.rint(base_cov[tt] * rng.uniform(lo_c, hi_c, size=tt.size)))
        else:
            c = rng.integers(lo_c, hi_c, size=tt.size)
        add_keys.append(sib)
        add_cnts.append(c.astype(np.int64))
        add_tids.append(tt.astype(np.int32))

    siblings(noise_frac, noise_counts[0], noise_counts[1], False)
    if branch_noise_frac:
        siblings(branch_noise_frac, 0.06, 0.5, True)   # above the 5 % ratio: real dead-end branches

    keys = np.concatenate([ref_km.ravel()] + add_keys) if add_keys else ref_km.ravel()
    cnts = np.concatenate([ref_cnt.ravel()] + add_cnts) if add_cnts else ref_cnt.ravel()
    ref_tid = np.repeat(np.arange(n_targets, dtype=np.int32), n_ref)
    tids = np.concatenate([ref_tid] + add_tids) if add_tids else ref_tid
    if canonical:
        keys = km.canonical(keys, k)
    keep = cnts > 0
    keys, cnts, tids = keys[keep], cnts[keep], tids[keep]
    uk, first = np.unique(keys, return_index=True)          # first occurrence wins
    keys, cnts, tids = uk, cnts[first], tids[first]

    n_real = int(keys.size)
    n_pad = max(0, n_keys - keys.size)
    if n_pad and not exact_pad:
        # headline-size tables: skip the 100M-key sort; random 62-bit pads are distinct
        # with overwhelming probability, the rare one equal to a real key is dropped
        chunks_k, chunks_c = [keys], [cnts]
        left = n_pad
        while left > 0:
            m = min(left, 1 << 24)
            hi = (1 << (2 * k)) if 2 * k < 64 else int(np.iinfo(np.uint64).max)
            pad = rng.integers(0, hi, size=m, dtype=np.uint64)
            if canonical:
                pad = km.canonical(pad, k)
            pos = np.searchsorted(keys, pad)
            pos[pos >= keys.size] = keys.size - 1
            ok = keys[pos] != pad
            chunks_k.append(pad[ok])
            chunks_c.append(rng.integers(2, 51, size=m)[ok])
            left -= m
        keys = np.concatenate(chunks_k)
        cnts = np.concatenate(chunks_c)
    elif n_pad:
        pad = (rng.integers(0, 1 << (2 * k), size=n_pad, dtype=np.uint64)
               if 2 * k < 64 else rng.integers(0, np.iinfo(np.uint64).max, size=n_pad, dtype=np.uint64))
        if canonical:
            pad = km.canonical(pad, k)
        padc = rng.integers(2, 51, size=n_pad)
        keys = np.concatenate([keys, pad])
        cnts = np.concatenate([cnts, padc])
        uk, first = np.unique(keys, return_index=True)
        keys, cnts = uk, cnts[first]
    names = ["%s%05d" % ((name or "syn") + "_t", i) for i in range(n_targets)]
    return {"targets": rows, "names": names, "keys": keys.astype(np.uint64),
            "counts": np.minimum(cnts, 0xFFFFFFFF).astype(np.uint32), "k": k, "n_real": n_real,
            # target each of the first n_real (non-pad) keys was generated for
            "key_target": tids}


def fast_canonical_keys(n, seed=0, k=31):
    """`n` DISTINCT canonical k-mers without a sort (real-size samples: example/run_leucegene.sh:24 counts with
    -s 799063683): the numbers seed_offset .. seed_offset + n - 1 pushed through a bijection of the 2k - 4 middle
    bits, between a first base A and a last base in {A, C, G} — the reverse complement of such a k-mer starts with
    T, G or C, so the k-mer itself is the canonical one.  Unsorted."""
    bits = 2 * k - 4
    assert 8 <= bits <= 58 and 0 < n < (1 << bits) - (1 << 20)
    M = np.uint64((1 << bits) - 1)
    idx = np.arange(n, dtype=np.uint64)
    x = (idx + np.uint64((int(seed) * 0x9E3779B1) % (1 << 20))) & M
    for mul, sh in ((0xFF51AFD7ED558CCD, 29), (0xC4CEB9FE1A85EC53, 31), (0x9E3779B97F4A7C15, 27)):
        x ^= x >> np.uint64(sh)                       # (each step is a bijection on `bits` bits)
        x *= np.uint64(mul)
        x &= M
    x ^= x >> np.uint64(bits // 2)
    x <<= np.uint64(2)
    x |= idx % np.uint64(3)
    return x


def make_sample(target_seqs, seed, k=31, n_keys=2_000_000, cov=(50, 2000), variant_frac=0.5,
                vaf=(0.1, 0.6), kinds=("snv", "ins", "del", "dup")):
    """One synthetic SAMPLE for BASELINE config 5: a k-mer count table for a GIVEN
    catalog of target sequences (strings, any lengths), seed = sample index.  Every target gets a
    coverage, half of them one variant at a random VAF, and the table is padded with random
    canonical k-mers to `n_keys`.  Returns (keys uint64, counts uint32), distinct canonical keys."""
    rng = np.random.default_rng(1000003 * (int(seed) + 1))
    all_k, all_c = [], []
    for seq in target_seqs:
        row = km.encode(seq).astype(np.uint8)
        if row.size < k or (row > 3).any():
            continue
        ref = km.sliding_kmers(row, k)
        base = int(rng.integers(cov[0], cov[1]))
        cnt = np.maximum(1, np.rint(base * rng.uniform(0.9, 1.1, size=ref.size))).astype(np.int64)
        if rng.random() < variant_frac and row.size >= 2 * k + 4:     # room for a variant with k-1 flanks
            kind = kinds[int(rng.integers(0, len(kinds)))]
            mut, lo, hi = _mutate(rng, row, kind, k)
            f = float(rng.uniform(vaf[0], vaf[1]))
            mk = km.sliding_kmers(mut, k)
            alt = mk[~np.isin(mk, ref)]
            if alt.size:
                all_k.append(alt)
                all_c.append(np.maximum(1, np.rint(base * f * rng.uniform(0.9, 1.1, size=alt.size))).astype(np.int64))
                s0, s1 = max(0, lo - k + 1), min(ref.size, max(hi, lo + 1))
                span = np.arange(s0, s1)
                span = span[~np.isin(ref[span], mk)]
                cnt[span] = np.maximum(1, np.rint(cnt[span] * (1.0 - f))).astype(np.int64)
        all_k.append(ref)
        all_c.append(cnt)
    keys = km.canonical(np.concatenate(all_k), k) if all_k else np.zeros(0, np.uint64)
    cnts = np.concatenate(all_c) if all_c else np.zeros(0, np.int64)
    n_pad = max(0, n_keys - keys.size)
    if n_pad > 20_000_000 and 12 <= k <= 31:
        # a real-size sample: distinct pads by construction (no sort of 10^8 keys), minus the few that are a real key
        uk, first = np.unique(keys, return_index=True)
        uc = np.minimum(cnts[first], 0xFFFFFFFF).astype(np.uint32)
        pads = fast_canonical_keys(n_pad, seed, k)
        at = np.searchsorted(uk, pads)
        at[at >= uk.size] = max(0, uk.size - 1)
        pads = pads[uk[at] != pads] if uk.size else pads
        pc = (np.arange(pads.size, dtype=np.uint32) * np.uint32(2654435761) >> np.uint32(16)) % np.uint32(49) + np.uint32(2)
        return np.concatenate([uk.astype(np.uint64), pads]), np.concatenate([uc, pc.astype(np.uint32)])
    if n_pad:
        hi = (1 << (2 * k)) if 2 * k < 64 else int(np.iinfo(np.uint64).max)
        keys = np.concatenate([keys, km.canonical(rng.integers(0, hi, size=n_pad, dtype=np.uint64), k)])
        cnts = np.concatenate([cnts, rng.integers(2, 51, size=n_pad)])
    uk, first = np.unique(keys, return_index=True)          # first occurrence wins
    return uk.astype(np.uint64), np.minimum(cnts[first], 0xFFFFFFFF).astype(np.uint32)


def write_jf(path, keys, counts, k, canonical=True):
    """Write keys/counts in the `binary/sorted` record layout our loaders read
    (9-digit length, JSON header, fixed key+count records)."""
    keys = np.asarray(keys, dtype=np.uint64)
    counts = np.asarray(counts, dtype=np.uint32)
    header = {"alignment": 8, "canonical": bool(canonical), "counter_len": 4,
              "format": "binary/sorted", "key_len": 2 * k, "val_len": 12,
              "size": int(1 << int(np.ceil(np.log2(max(16, 2 * keys.size))))),
              "cmdline": ["km_amd-synthetic"]}
    text = json.dumps(header, separators=(",", ":")).encode("ascii")
    text += b"\0" * ((-(9 + len(text))) % 8)
    kb = (2 * k + 7) // 8
    rec = np.zeros((keys.size, kb + 4), dtype=np.uint8)
    for b in range(kb):
        rec[:, b] = ((keys >> np.uint64(8 * b)) & np.uint64(0xFF)).astype(np.uint8)
    for b in range(4):
        rec[:, kb + b] = ((counts >> np.uint32(8 * b)) & np.uint32(0xFF)).astype(np.uint8)
    with open(path, "wb") as fh:
        fh.write(b"%09d" % len(text))
        fh.write(text)
        fh.write(rec.tobytes())


def write_case(outdir, **spec):
    """Materialise a case as FASTA files + one .jf.  Returns (fasta_paths, jf_path, meta)."""
    case = make_case(**spec)
    fas = []
    tdir = os.path.join(outdir, "targets")
    os.makedirs(tdir, exist_ok=True)
    h = hashlib.md5()
    for name, row in zip(case["names"], case["targets"]):
        p = os.path.join(tdir, name + ".fa")
        s = km.decode(row)
        with open(p, "w") as fh:
            fh.write(">synthetic:1-%d | name=%s\n%s\n" % (len(s), name, s))
        fas.append(p)
        h.update(s.encode())
    dbp = os.path.join(outdir, (spec.get("name") or "syn") + ".jf")
    write_jf(dbp, case["keys"], case["counts"], case["k"])
    h.update(case["keys"].tobytes())
    h.update(case["counts"].tobytes())
    return fas, dbp, {"md5": h.hexdigest(), "n_keys": int(case["keys"].size)}


# Small slices that tests/golden/make_golden.py runs through the reference.
GOLDEN_SPECS = [
    # config-4 shaped slice
    dict(name="cfg4_small", n_targets=100, length=500, n_keys=150_000, seed=HEADLINE_SEED),
    # every target mutated, 1-3 variants each, some homozygous, real dead-end branches
    dict(name="stress", n_targets=60, length=300, n_keys=60_000, seed=11, variant_frac=1.0,
         variants_per_target=(1, 3), hom_frac=0.25, branch_noise_frac=0.03, noise_frac=0.03),
    # low coverage around the -c 5 threshold
    dict(name="lowcov", n_targets=40, length=200, n_keys=20_000, seed=12, variant_frac=0.6,
         cov=(2, 30), vaf=(0.2, 0.8)),
    # tight walk budgets
    dict(name="tight", n_targets=40, length=300, n_keys=30_000, seed=13, variant_frac=1.0,
         variants_per_target=(1, 2), branch_noise_frac=0.05,
         params=dict(steps=40, branchs=2)),
    # node limit -> sys.exit after some targets were printed
    dict(name="nodelimit", n_targets=12, length=300, n_keys=10_000, seed=14, variant_frac=1.0,
         kinds=("dup",), params=dict(nodes=300)),
    # short k
    dict(name="k21", n_targets=30, length=150, k=21, n_keys=20_000, seed=15, variant_frac=0.8),
]
