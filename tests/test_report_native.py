"""CPU-only parity tests of the native reporting path (km_report_rows, csrc/report.cpp)
against the reference's golden TSVs and against the Python restatement km_amd/report.py.
The oracle only produces the hot path's outputs here (on a GPU box they come from the HIP
kernels); it is the checker's input, not the thing under test."""
import json
import os

import numpy as np
import pytest

from km_amd import kmer as km
from km_amd import lib as kmlib
from km_amd import report, synth
from oracle import km_oracle as ko

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


@pytest.fixture(autouse=True)
def _cwd(monkeypatch):
    monkeypatch.chdir(HERE)


def _load(name):
    with open(os.path.join(GOLD, name)) as fh:
        return json.load(fh)


def _raw_from_oracle(results):
    """The dict Batch.fetch() returns, rebuilt from oracle outputs (paths run-length encoded
    like the kernels deliver them)."""
    status, n_ref, probes, node_off = [], [], [], [0]
    node_kmer, node_count = [], []
    path_off, run_off, run_start, run_len, path_len, path_min_cov = [0], [0], [], [], [], []
    for res in results:
        status.append(0)
        n_ref.append(res["n_ref"])
        probes.append(res["probes"])
        node_kmer += [km.pack_str(s) for s in res["kmers"]]
        node_count += list(res["counts"])
        node_off.append(len(node_kmer))
        for p, mc in zip(res["paths"], res["min_cov"]):
            p = list(p)
            i = 0
            while i < len(p):
                j = i
                while j + 1 < len(p) and p[j + 1] == p[j] + 1:
                    j += 1
                run_start.append(p[i])
                run_len.append(j - i + 1)
                i = j + 1
            run_off.append(len(run_start))
            path_len.append(len(p))
            path_min_cov.append(mc)
        path_off.append(len(path_len))
    u32, u64 = np.uint32, np.uint64
    return {"status": np.array(status, u32), "n_ref": np.array(n_ref, u32), "probes": np.array(probes, u64),
            "node_off": np.array(node_off, u64), "node_kmer": np.array(node_kmer, u64),
            "node_count": np.array(node_count, u32), "path_off": np.array(path_off, u32),
            "run_off": np.array(run_off, u64), "run_start": np.array(run_start, u32),
            "run_len": np.array(run_len, u32), "path_len": np.array(path_len, u32),
            "path_min_cov": np.array(path_min_cov, u32)}


def _python_rows(res, seq, db_name):
    kmers = np.array([km.pack_str(s) for s in res["kmers"]], dtype=np.uint64)
    tr = report.TargetResult(res["name"], seq, res["k"], res["n_ref"], kmers,
                             np.array(res["counts"], dtype=np.uint32),
                             [np.array(p, dtype=np.int64) for p in res["paths"]], res["min_cov"])
    return report.target_rows(tr, db_name)


@pytest.mark.parametrize("idx", range(10))
def test_native_rows_match_reference_golden_tsv(idx):
    case = _load("fixtures_tsv.json")["cases"][idx]
    db = ko.KmerDB(case["db"], cutoff=0.05, n_cutoff=5)
    results, names, seqs = [], [], []
    for fa in case["targets"]:
        seq = ko.read_fasta_concat(fa)
        name = os.path.splitext(os.path.basename(fa))[0]
        results.append(ko.analyse_target(seq, name, db))
        names.append(name)
        seqs.append(seq)
    blocks = kmlib.report_rows(_raw_from_oracle(results), names, seqs, 31, case["db"])
    rows = [r for b in blocks for r in b]
    assert [report.HEADER] + rows == case["lines"][10:]


@pytest.mark.parametrize("name", [s["name"] for s in synth.GOLDEN_SPECS])
def test_native_rows_match_python_restatement(name, tmp_path):
    spec = [s for s in synth.GOLDEN_SPECS if s["name"] == name]
    fas, dbp, _ = synth.write_case(str(tmp_path), **spec[0])
    db = ko.KmerDB(dbp, cutoff=0.05, n_cutoff=5)
    results, names, seqs = [], [], []
    for fa in fas[:40]:
        seq = ko.read_fasta_concat(fa)
        nm = os.path.splitext(os.path.basename(fa))[0]
        results.append(ko.analyse_target(seq, nm, db))
        names.append(nm)
        seqs.append(seq)
    blocks = kmlib.report_rows(_raw_from_oracle(results), names, seqs, results[0]["k"], "x.jf")
    n_var = 0
    for res, seq, got in zip(results, seqs, blocks):
        assert got == _python_rows(res, seq, "x.jf")
        n_var += len(got) > 1
    assert n_var >= 3 or name in ("budget", "nodelimit")


def test_native_rows_skip_non_ok_targets_and_empty_batch():
    assert kmlib.report_rows({"status": np.zeros(0, np.uint32), "n_ref": np.zeros(0, np.uint32),
                              "probes": np.zeros(0, np.uint64), "node_off": np.zeros(1, np.uint64),
                              "node_kmer": np.zeros(0, np.uint64), "node_count": np.zeros(0, np.uint32),
                              "path_off": np.zeros(1, np.uint32), "run_off": np.zeros(1, np.uint64),
                              "run_start": np.zeros(0, np.uint32), "run_len": np.zeros(0, np.uint32),
                              "path_len": np.zeros(0, np.uint32), "path_min_cov": np.zeros(0, np.uint32)},
                             [], [], 31, "x.jf") == []
    case = _load("fixtures_tsv.json")["cases"][0]
    db = ko.KmerDB(case["db"], cutoff=0.05, n_cutoff=5)
    fa = case["targets"][0]
    seq = ko.read_fasta_concat(fa)
    res = ko.analyse_target(seq, "t", db)
    raw = _raw_from_oracle([res, res])
    raw["status"][0] = kmlib.T_NODE_LIMIT
    blocks = kmlib.report_rows(raw, ["a", "b"], [seq, seq], 31, "d.jf")
    assert blocks[0] == [] and blocks[1] == _python_rows(dict(res, name="b"), seq, "d.jf")


def test_native_rows_many_random_variants():
    """A few hundred synthetic variant targets (all variant kinds, clusters, tandem duplications):
    every row text of the native path equals the Python restatement's."""
    case = synth.make_case(n_targets=260, length=300, k=31, n_keys=150_000, seed=4242, variant_frac=0.9,
                           exact_pad=False)
    db = ko.KmerDB(None, cutoff=0.05, n_cutoff=5,
                   records={"k": 31, "canonical": True, "keys": case["keys"], "counts": case["counts"]})
    results, names, seqs = [], [], []
    for i in range(260):
        seq = km.decode(case["targets"][i])
        results.append(ko.analyse_target(seq, case["names"][i], db))
        names.append(case["names"][i])
        seqs.append(seq)
    blocks = kmlib.report_rows(_raw_from_oracle(results), names, seqs, 31, "rnd.jf")
    kinds = set()
    for res, seq, got in zip(results, seqs, blocks):
        assert got == _python_rows(res, seq, "rnd.jf")
        kinds.update(r.split("\t")[2] for r in got)
    assert {"Reference", "Substitution", "Insertion", "Deletion", "ITD"} <= kinds


def test_native_rows_rank_deficient_cluster_fit():
    """A tandem-duplication path next to the path that walks the duplicated stretch twice makes the
    cluster's least-squares matrix exactly rank deficient (sigma_3 ~ 1e-15): the minimum-norm
    answer must be numpy's (found by tools/soak.py; solving through A^T A kept the null direction)."""
    from oracle import c_oracle
    k = 32
    case = synth.make_case(n_targets=1500, length=200, k=k, n_keys=200_000, seed=1028014213,
                           variant_frac=0.7, cov=(50, 2000), exact_pad=False)
    co = c_oracle.COracle(case["keys"][:case["n_real"]], case["counts"][:case["n_real"]], k)
    results, names, seqs = [], [], []
    for t in range(150, 260):
        w = co.analyse(case["targets"][t], max_stack=60, max_break=3)
        results.append({"name": case["names"][t], "k": k, "n_ref": w["n_ref"],
                        "kmers": [km.unpack(int(x), k) for x in w["kmers"]], "counts": w["counts"].tolist(),
                        "paths": w["paths"], "min_cov": w["min_cov"], "probes": w["probes"]})
        names.append(case["names"][t])
        seqs.append(km.decode(case["targets"][t]))
    assert "syn_t00194" in names
    blocks = kmlib.report_rows(_raw_from_oracle(results), names, seqs, k, "soak.jf")
    for res, seq, got in zip(results, seqs, blocks):
        assert got == _python_rows(res, seq, "soak.jf"), res["name"]


def _as_delivery(raw, lean):
    """The same results in the form km_batch_result delivers them: no node_kmer (the target's own
    k-mers come from its sequence), walk-discovered k-mers in extra_kmer, ref_max_cov, and — lean —
    no node_count rows for bare-reference targets."""
    n = len(raw["status"])
    noff, poff = raw["node_off"].astype(np.int64), raw["path_off"].astype(np.int64)
    counts, extra, node_off, extra_off, ref_max = [], [], [0], [0], []
    for t in range(n):
        nr = int(raw["n_ref"][t])
        c = raw["node_count"][noff[t]:noff[t + 1]]
        kms = raw["node_kmer"][noff[t]:noff[t + 1]]
        paths = [kmlib.expand_path(raw, p).tolist() for p in range(poff[t], poff[t + 1])]
        bare = len(c) == nr and paths == [list(range(nr))]
        ref_max.append(int(c[:nr].max()) if bare else 0xFFFFFFFF)
        if not (lean and bare):
            counts += c.tolist()
        extra += kms[nr:].tolist()
        node_off.append(len(counts))
        extra_off.append(len(extra))
    out = {k_: v for k_, v in raw.items() if k_ not in ("node_kmer", "node_count", "node_off")}
    out.update(node_off=np.array(node_off, np.uint64), node_count=np.array(counts, np.uint32),
               extra_off=np.array(extra_off, np.uint64), extra_kmer=np.array(extra, np.uint64),
               ref_max_cov=np.array(ref_max, np.uint32))
    return out


@pytest.mark.parametrize("name", ["cfg4_small", "stress", "lowcov", "k21"])
def test_native_rows_from_delivery_views(name, tmp_path):
    """km_report_rows over the delivered forms (full and lean) == over the fetched arrays."""
    spec = next(s_ for s_ in synth.GOLDEN_SPECS if s_["name"] == name)
    fas, dbp, _meta = synth.write_case(str(tmp_path), **spec)
    db = ko.KmerDB(dbp, cutoff=0.05, n_cutoff=5)
    names, seqs, results = [], [], []
    for f in fas:
        nm = os.path.splitext(os.path.basename(f))[0]
        seq = ko.read_fasta_concat(f)
        names.append(nm); seqs.append(seq)
        results.append(ko.analyse_target(seq, nm, db))
    raw = _raw_from_oracle(results)
    want = kmlib.report_rows(raw, names, seqs, db.k, dbp)
    full = _as_delivery(raw, lean=False)
    lean = _as_delivery(raw, lean=True)
    assert len(lean["node_count"]) <= len(full["node_count"]) and (name != "cfg4_small" or len(lean["node_count"]) < 0.6 * len(full["node_count"]))
    assert kmlib.report_rows(full, names, seqs, db.k, dbp) == want
    assert kmlib.report_rows(lean, names, seqs, db.k, dbp) == want


def test_native_rows_lean_zero_coverage_reference():
    """A bare-reference target whose k-mers all have count 0 prints nan expression (rVAF aliases
    coef in km/utils/PathQuant.py:144-154); ref_max_cov carries that through a lean delivery."""
    k = 31
    seq = km.decode(np.random.default_rng(5).integers(0, 4, size=80, dtype=np.uint8))
    db = ko.KmerDB(None, cutoff=0.05, n_cutoff=5, records={"k": k, "canonical": True,
                                                            "keys": np.zeros(0, np.uint64), "counts": np.zeros(0, np.uint32)})
    res = ko.analyse_target(seq, "zero", db)
    raw = _raw_from_oracle([res])
    want = kmlib.report_rows(raw, ["zero"], [seq], k, "mem.jf")
    assert "nan" in want[0][0].split("\t")[5]
    assert kmlib.report_rows(_as_delivery(raw, True), ["zero"], [seq], k, "mem.jf") == want


def _dump_view(path, raw, seqs, k):
    """The arrays of a delivery view (extra_kmer form, ref_max_cov set for bare-reference targets) in
    the plain binary layout tests/host/report_views.cpp reads."""
    n = len(seqs)
    n_ref = raw["n_ref"].astype(np.int64)
    noff = raw["node_off"].astype(np.int64)
    extra, xoff, ref_max = [], [0], []
    for t in range(n):
        extra += raw["node_kmer"][noff[t] + n_ref[t]:noff[t + 1]].tolist()
        xoff.append(len(extra))
        npaths = int(raw["path_off"][t + 1] - raw["path_off"][t])
        bare = npaths == 1 and noff[t + 1] - noff[t] == n_ref[t] and int(raw["path_len"][raw["path_off"][t]]) == n_ref[t]
        ref_max.append(int(raw["node_count"][noff[t]:noff[t] + n_ref[t]].max()) if bare else 0xFFFFFFFF)
    blob = "".join(seqs).encode()
    boff = np.cumsum([0] + [len(s_) for s_ in seqs]).astype(np.uint64)
    with open(path, "wb") as fh:
        fh.write(np.array([n, k], dtype=np.uint32).tobytes())
        for arr, dt in ((np.frombuffer(blob, dtype=np.uint8), np.uint8), (boff, np.uint64), (raw["status"], np.uint32),
                        (raw["n_ref"], np.uint32), (raw["node_off"], np.uint64), (raw["node_count"], np.uint32),
                        (np.array(xoff), np.uint64), (np.array(extra), np.uint64), (raw["path_off"], np.uint32),
                        (raw["run_off"], np.uint64), (raw["run_start"], np.uint32), (raw["run_len"], np.uint32),
                        (raw["path_min_cov"], np.uint32), (np.array(ref_max), np.uint32)):
            a = np.ascontiguousarray(arr, dtype=dt)
            fh.write(np.uint64(a.size).tobytes())
            fh.write(a.tobytes())


def _synthetic_view(n_targets=60):
    case = synth.make_case(n_targets=n_targets, length=220, k=21, n_keys=4000, seed=9123, variant_frac=0.6,
                           variants_per_target=(1, 2), cov=(60, 500))
    db = ko.KmerDB(None, cutoff=0.05, n_cutoff=5,
                   records={"k": 21, "canonical": True, "keys": case["keys"], "counts": case["counts"]})
    results, seqs = [], []
    for row, name in zip(case["targets"], case["names"]):
        seq = km.decode(row)
        results.append(ko.analyse_target(seq, name, db))
        seqs.append(seq)
    return _raw_from_oracle(results), seqs, results


def test_inconsistent_views_are_error_codes_under_the_sanitizers(tmp_path):
    """csrc/report.cpp built for the CPU with AddressSanitizer + UBSan (tests/host/report_views.cpp): the full
    and the lean view give the same rows; views whose offsets, node indices or counts do not hang together
    — among them the one that overwrote the heap of a test process in round 2 (a target with variant paths
    delivered without counts) — come back as KM_E_ARG / err 5, with no out-of-bounds access on the way."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    root = os.path.dirname(HERE)
    exe = str(tmp_path / "report_views")
    subprocess.check_call([gxx, "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-o", exe,
                           os.path.join(HERE, "host", "report_views.cpp"),
                           os.path.join(root, "km_amd", "csrc", "report.cpp")])
    raw, seqs, results = _synthetic_view()
    assert any(len(r["paths"]) > 1 for r in results) and any(len(r["paths"]) == 1 for r in results)
    view = str(tmp_path / "view.bin")
    _dump_view(view, raw, seqs, 21)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    proc = subprocess.run([exe, view], capture_output=True, text=True, env=env, timeout=600)
    assert proc.returncode == 0 and "VIEWS OK" in proc.stdout, proc.stdout[-3000:] + proc.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in proc.stderr and "runtime error" not in proc.stderr, proc.stderr[-3000:]


def test_inconsistent_view_through_the_binding():
    """The same through ctypes and the product library: err 5 turns into an exception for that target only,
    lengths that do not match the offsets into KM_E_ARG for the call."""
    raw, seqs, results = _synthetic_view(24)
    names = [r["name"] for r in results]
    good = kmlib.report_rows(raw, names, seqs, 21, "view.jf")
    tv = next(t for t, r in enumerate(results) if len(r["paths"]) > 1)
    bad = dict(raw)
    bad["run_start"] = raw["run_start"].copy()
    bad["run_start"][int(raw["run_off"][int(raw["path_off"][tv]) + 1])] += 50000
    rows = kmlib.report_rows(bad, names, seqs, 21, "view.jf")
    assert isinstance(rows[tv], RuntimeError)
    assert all(rows[t] == good[t] for t in range(len(names)) if t != tv)
    bad = dict(raw)
    bad["node_off"] = raw["node_off"].copy()
    bad["node_off"][-1] += 3
    with pytest.raises(kmlib.KmError):
        kmlib.report_rows(bad, names, seqs, 21, "view.jf")


def _build_host(tmp_path, name):
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / name)
    subprocess.check_call([gxx, "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(HERE, "host", name + ".cpp")])
    return exe


def test_fixed_point_printing_is_printf(tmp_path):
    """The rows print rVAF and expression with an integer-arithmetic "%.3f" / "%.1f" (csrc/report.cpp:
    fixed_digits): 40 M values — random bit patterns, exact rounding ties, ratios of small integers — against
    snprintf, digit for digit."""
    import subprocess
    exe = _build_host(tmp_path, "fixed_digits_check")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.strip().endswith(" 0 mismatches")


def test_natural_order_of_rows_is_pythons(tmp_path):
    """The row order compares natural-sort keys (km_amd/report.py: _natural — re.split("([0-9]+)"), digit runs as
    ints, the rest lower-cased).  csrc/report.cpp compares the two strings directly, without building the lists:
    same answer as Python's list comparison on random strings of digits, letters of both cases and punctuation."""
    import random
    import re
    import subprocess
    exe = _build_host(tmp_path, "natural_check")

    def natural(text):
        return [int(x) if x.isdigit() else x.lower() for x in re.split("([0-9]+)", text)]

    rng = random.Random(77)
    alphabet = "0012789abAB:/._-= "
    pairs = []
    for _ in range(20000):
        a = "".join(rng.choice(alphabet) for _ in range(rng.randint(0, 9))).replace("\n", "")
        if rng.random() < 0.5:                      # mostly related strings: a prefix, one edit, more digits
            b = list(a)
            for _ in range(rng.randint(0, 2)):
                if b and rng.random() < 0.5:
                    b[rng.randrange(len(b))] = rng.choice(alphabet)
                else:
                    b.insert(rng.randint(0, len(b)), rng.choice(alphabet))
            b = "".join(b)[:rng.randint(0, 12)]
        else:
            b = "".join(rng.choice(alphabet) for _ in range(rng.randint(0, 9)))
        pairs.append((a, b))
    pairs += [("", ""), ("7", "007"), ("a7", "a007b"), ("45:/TCTG:45", "45:/TCTG:46"), ("cluster", "vs_ref"),
              ("n=10", "n=9"), ("0.312", "nan"), ("123456789012345678901234567890", "123456789012345678901234567891")]
    p = subprocess.run([exe], input="".join(a + "\n" + b + "\n" for a, b in pairs), capture_output=True, text=True,
                       timeout=120)
    assert p.returncode == 0
    got = [int(x) for x in p.stdout.split()]
    assert len(got) == len(pairs)
    for (a, b), g in zip(pairs, got):
        ka, kb = natural(a), natural(b)
        want = -1 if ka < kb else (1 if ka > kb else 0)
        assert g == want, (a, b, g, want)


def test_worker_threads_spread_once_and_keep_the_process_mask(monkeypatch):
    """The team of km_report_rows places each worker on a CPU of its own when the worker starts and then gives
    it the process's mask back (csrc/report.cpp, Team::spread): afterwards no thread of the process is pinned,
    the calling thread's mask is untouched, and the rows are those of a single-threaded call."""
    if not hasattr(os, "sched_getaffinity") or len(os.sched_getaffinity(0)) < 2:
        pytest.skip("needs Linux and two CPUs")
    raw, seqs, results = _synthetic_view(n_targets=256)      # >= 64 targets per thread, or the call stays single-threaded
    names = [r["name"] for r in results]
    before = os.sched_getaffinity(0)
    monkeypatch.setenv("KM_REPORT_THREADS", "1")
    want = kmlib.report_rows(raw, names, seqs, 21, "view.jf")
    monkeypatch.setenv("KM_REPORT_THREADS", "4")
    for _ in range(3):
        assert kmlib.report_rows(raw, names, seqs, 21, "view.jf") == want
    assert os.sched_getaffinity(0) == before
    masks = set()
    for tid in os.listdir("/proc/self/task"):
        try:
            masks.add(frozenset(os.sched_getaffinity(int(tid))))
        except OSError:
            pass                                             # a thread that ended meanwhile
    assert masks == {frozenset(before)}
