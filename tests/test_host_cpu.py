"""CPU-only tests of the product's host logic and of the C-ABI library surface.
The oracle is used only as the checker (it feeds the host reporting code with the
hot path's outputs, which on a GPU box come from the HIP kernels)."""
import json
import os
import re

import numpy as np
import pytest

from km_amd import kmer as km
from km_amd import lib as kmlib
from km_amd import report, synth
from oracle import km_oracle as ko

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")


def _load(name):
    with open(os.path.join(GOLD, name)) as fh:
        return json.load(fh)


@pytest.fixture(autouse=True)
def _cwd(monkeypatch):
    monkeypatch.chdir(HERE)


def _result_from_oracle(res, seq):
    kmers = np.array([km.pack_str(s) for s in res["kmers"]], dtype=np.uint64)
    return report.TargetResult(res["name"], seq, res["k"], res["n_ref"], kmers,
                               np.array(res["counts"], dtype=np.uint32),
                               [np.array(p, dtype=np.int64) for p in res["paths"]], res["min_cov"])


@pytest.mark.parametrize("idx", range(10))
def test_report_rows_match_golden(idx):
    case = _load("fixtures_tsv.json")["cases"][idx]
    db = ko.KmerDB(case["db"], cutoff=0.05, n_cutoff=5)
    rows = []
    for fa in case["targets"]:
        seq = ko.read_fasta_concat(fa)
        name = os.path.splitext(os.path.basename(fa))[0]
        res = ko.analyse_target(seq, name, db)
        rows += report.target_rows(_result_from_oracle(res, seq), case["db"])
    assert [report.HEADER] + rows == case["lines"][10:]


@pytest.mark.parametrize("name", ["stress", "lowcov", "k21"])
def test_report_rows_synthetic(name, tmp_path):
    spec = [s for s in synth.GOLDEN_SPECS if s["name"] == name][0]
    fas, dbp, _ = synth.write_case(str(tmp_path), **spec)
    db = ko.KmerDB(dbp, cutoff=0.05, n_cutoff=5)
    for fa in fas[:25]:
        seq = ko.read_fasta_concat(fa)
        nm = os.path.splitext(os.path.basename(fa))[0]
        res = ko.analyse_target(seq, nm, db)
        assert report.target_rows(_result_from_oracle(res, seq), "x.jf") == ko.target_rows(res, "x.jf")


def test_split_paths_against_oracle_random():
    rng = np.random.default_rng(3)
    for _ in range(300):
        n = int(rng.integers(5, 60))
        ref = list(range(n))
        a, b = sorted(rng.integers(0, n + 1, size=2).tolist())
        mid = list(range(100, 100 + int(rng.integers(0, 40))))
        alt = ref[:a] + mid + ref[b:]
        if rng.random() < 0.3:
            alt = ref[:b] + ref[a:]            # duplication-like
        k = int(rng.integers(2, 8))
        want = ko.path_diff(ref, alt, k)
        got = report.split_paths(np.array(ref), np.array(alt), k)
        assert got == (want[0], want[1], want[2], want[5])


def test_kmer_helpers():
    s = "ACGTTGCAAGGCTTAACCGGTTACGATCGAT"
    v = km.pack_str(s)
    assert km.unpack(v, len(s)) == s
    rc = int(km.revcomp(np.array([v], dtype=np.uint64), len(s))[0])
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    assert km.unpack(rc, len(s)) == "".join(comp[c] for c in reversed(s))
    codes = km.encode(s + "ACGT")
    sl = km.sliding_kmers(codes, 31)
    assert sl.size == 5 and int(sl[0]) == v


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads and exports every entry point of include/kmgpu.h."""
    lib = kmlib.load()
    hdr = open(os.path.join(ROOT, "include", "kmgpu.h")).read()
    declared = set(re.findall(r"^(?:int|void|const char\*)\s+(km\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(kmlib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.km_version()
    assert lib.km_strerror(2).decode().startswith("not a Jellyfish")


def test_kmjf_broadcast_refuses_bad_arguments_without_touching_a_gpu():
    """The single-process multi-GPU entry of the C-ABI exists and fails cleanly: no device list (KM_E_ARG) before any
    HIP call, and — on a box without a GPU, like the one the CPU suite runs on — one device gives an error code,
    not a crash."""
    import ctypes as C
    lib = kmlib.load()
    keys = np.arange(10, dtype=np.uint64)
    db = kmlib.Database.from_records(keys, np.ones(10, np.uint32), 31)
    out = (C.c_void_p * 1)()
    assert lib.kmjf_broadcast(db._h, None, 0, out) == 4                       # KM_E_ARG
    dev = (C.c_int * 2)(0, 0)
    assert lib.kmjf_broadcast(db._h, dev, 2, (C.c_void_p * 2)()) == 4        # a device named twice
    n = C.c_int(0)
    if lib.km_device_count(C.byref(n)) != 0 or n.value == 0:
        rc = lib.kmjf_broadcast(db._h, (C.c_int * 1)(0), 1, out)
        assert rc != 0 and lib.km_last_error()
    db.close()


def test_native_reader_matches_oracle_reader(tmp_path):
    """Host-only ABI calls: kmjf_open / kmjf_info / kmjf_records need no GPU."""
    from oracle import jf_reader as jr
    for name in sorted(os.listdir(os.path.join(HERE, "data", "jf"))):
        p = os.path.join(HERE, "data", "jf", name)
        db = kmlib.Database.open(p)
        want = jr.read_jf(p)
        info = db.info
        assert (info.k, bool(info.canonical), info.n_records) == (want["k"], want["canonical"],
                                                                 len(want["keys"]))
        keys, counts = db.records()
        assert (keys == want["keys"]).all() and (counts == want["counts"]).all()
        db.close()
    # error behaviour
    bad = tmp_path / "bad.jf"
    bad.write_bytes(b"not a jellyfish file at all")
    with pytest.raises(kmlib.KmError) as e:
        kmlib.Database.open(str(bad))
    assert e.value.code == 2
    with pytest.raises(kmlib.KmError) as e:
        kmlib.Database.open(str(tmp_path / "missing.jf"))
    assert e.value.code == 1
    # synthetic writer -> native reader round trip
    case = synth.make_case(n_targets=3, length=100, n_keys=500, seed=5)
    p = tmp_path / "syn.jf"
    synth.write_jf(str(p), case["keys"], case["counts"], 31)
    db = kmlib.Database.open(str(p))
    keys, counts = db.records()
    assert (keys == case["keys"]).all() and (counts == case["counts"]).all()
    # calls that need the device table fail loudly, never fall back
    with pytest.raises(kmlib.KmError) as e:
        db.query(np.array([1], dtype=np.uint64))
    assert e.value.code == 7


def test_threshold_shortcut_is_exact(tmp_path):
    """The walk skips the float64 threshold for sums below a precomputed bound
    (device_common.h: threshold_shortcut).  A host program built from the same header checks, over
    fixed and random (ratio, cutoff) pairs, that every such sum has the threshold the float64
    expression gives and that the bound is tight.  hipcc compiles it; nothing runs on a GPU."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "threshold_check")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I" + os.path.join(root, "include"),
                           "-o", exe, os.path.join(root, "tests", "host", "threshold_check.hip")],
                          stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout[-500:] + out.stderr[-500:]
    assert int(out.stdout.split()[1]) > 3000
