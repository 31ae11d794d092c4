"""Pins the CPU oracle (oracle/) against golden vectors produced by the
unmodified reference (tests/golden/make_golden.py).  CPU only."""
import hashlib
import json
import os
import re

import numpy as np
import pytest

from oracle import jf_reader as jr
from oracle import km_oracle as ko
from km_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _load(name):
    with open(os.path.join(GOLD, name)) as fh:
        return json.load(fh)


@pytest.fixture(autouse=True)
def _cwd(monkeypatch):
    monkeypatch.chdir(HERE)


def test_min_cov_known_answers():
    """km/tests/test_main.py:581-652 (numbers produced by real Jellyfish)."""
    seq = ko.read_fasta_concat("./data/catalog/GRCh38/FLT3-ITD_exons_13-15.fa")
    assert ko.coverage("./data/jf/02H025_NPM1.jf", seq) == (0, 345, 0, 0, 0.0, 315, 315)
    s, ln, mn, mx, mean, n, n0 = ko.coverage("./data/jf/03H112_IandI.jf", seq)
    assert (s, ln, mn, mx, n, n0) == (275596, 345, 618, 1368, 315, 0)
    assert "%.2f" % mean == "874.91"
    for g in _load("fixtures_children.json")["min_cov"]:
        assert list(ko.coverage(g["db"], ko.read_fasta_concat(g["target"]))) == g["cov"]


def test_jf_record_counts():
    want = {"02H025_NPM1": 1938, "02H033_DNMT3A_sub": 209, "03H112_IandI": 1604,
            "03H116_ITD": 2560, "05H094_FLT3-TKD_del": 274}
    for name, n in want.items():
        d = jr.read_jf("./data/jf/%s.jf" % name)
        assert d["k"] == 31 and d["canonical"] and len(d["keys"]) == n
        assert (jr.canonical_np(d["keys"], 31) == d["keys"]).all()


def test_get_child_vectors():
    for case in _load("fixtures_children.json")["cases"]:
        db = ko.KmerDB(case["db"], cutoff=case["ratio"], n_cutoff=case["count"])
        for seq, cnt, kids in case["children"]:
            assert db.query(seq) == cnt
            assert db.get_child(seq) == kids


@pytest.mark.parametrize("idx", range(10))
def test_fixture_tsv_exact(idx):
    case = _load("fixtures_tsv.json")["cases"][idx]
    assert case["stable"]
    lines, err = ko.run_find_mutation(case["targets"], case["db"])
    assert err == case["exit"]
    assert lines == case["lines"]
    assert hashlib.md5(("\n".join(lines) + "\n").encode()).hexdigest() == case["md5"]


def test_reference_known_answers():
    """Rows the reference's own tests index (km/tests/test_main.py:56-66 etc.)."""
    lines, _ = ko.run_find_mutation(["./data/catalog/GRCh38/NPM1_4ins_exons_10-11utr.fa"],
                                    "./data/jf/02H025_NPM1.jf")
    f = lines[13].split("\t")
    assert f[2] == "Insertion" and f[3] == "45:/TCTG:45" and f[11] == "cluster 1 n=1"
    lines, _ = ko.run_find_mutation(["./data/catalog/GRCh38/FLT3-ITD_exons_13-15.fa"],
                                    "./data/jf/03H116_ITD.jf")
    f = lines[13].split("\t")
    assert f[2] == "ITD" and f[3].startswith("204:/AACTCC") and f[3].endswith("CACC:204")
    lines, _ = ko.run_find_mutation(["./data/catalog/GRCh38/FLT3-TKD_exon_20.fa"],
                                    "./data/jf/05H094_FLT3-TKD_del.jf")
    assert lines[13].split("\t")[2:4] == ["Deletion", "32:gat/:35"]
    lines, _ = ko.run_find_mutation(["./data/catalog/GRCh38/DNMT3A_R882_exon_23.fa"],
                                    "./data/jf/02H033_DNMT3A_sub.jf")
    assert lines[13].split("\t")[2:4] == ["Substitution", "33:c/T:34"]


def test_walk_vectors():
    """Node sets, logical probe counts, path sequences and min coverages."""
    for case in _load("fixtures_walk.json")["cases"]:
        db = ko.KmerDB(case["db"], cutoff=0.05, n_cutoff=5)
        for fa, g in zip(case["targets_fa"], case["targets"]):
            res = ko.analyse_target(ko.read_fasta_concat(fa), g["name"], db)
            assert len(res["kmers"]) + 2 == g["num_k"]
            assert sorted([k, c] for k, c in zip(res["kmers"], res["counts"])) == g["nodes"]
            assert res["probes"] in g["probes_seen"]
            allk = res["kmers"] + ["", ""]
            seqs = sorted(ko.spell(allk, p, True) for p in res["paths"])
            assert seqs == g["path_seqs"]
            by_seq = {ko.spell(allk, p, True): m for p, m in zip(res["paths"], res["min_cov"])}
            assert [by_seq[s] for s in seqs] == g["path_min_cov"]


def test_repeated_kmer_raises():
    with pytest.raises(ValueError):
        ko.ref_kmers("A" * 32, "polyA", 31)      # km/tests/test_main.py:555-561


_norm = lambda l: re.sub(r"cluster \d+ n=", "cluster * n=", l)


def _blocks(lines):
    out = {}
    for l in lines:
        if l.startswith("#") or l.startswith("Database"):
            continue
        out.setdefault(l.split("\t")[1], []).append(l)
    return out


@pytest.mark.parametrize("idx", range(len(synth.GOLDEN_SPECS)))
def test_synth_tsv(idx, tmp_path, monkeypatch):
    case = _load("synth_tsv.json")["cases"][idx]
    spec = case["spec"]
    fas, dbp, meta = synth.write_case(str(tmp_path), **spec)
    assert meta["md5"] == case["input_md5"], "synthetic generator drifted; regenerate goldens"
    monkeypatch.chdir(tmp_path)
    rel = [os.path.relpath(f, str(tmp_path)) for f in fas]
    lines, err = ko.run_find_mutation(rel, os.path.relpath(dbp, str(tmp_path)),
                                      **spec.get("params", {}))
    assert err == case["exit"]
    # walk-level outputs are seed-stable in the reference for every case
    db = ko.KmerDB(dbp, cutoff=0.05, n_cutoff=5)
    if case["stable"]:
        assert lines == case["lines"]
    else:
        # the reference itself renumbers clusters with PYTHONHASHSEED: compare modulo that
        assert sorted(map(_norm, lines)) == sorted(map(_norm, case["lines"]))
        mine = _blocks(lines)
        seen = [_blocks(x) for x in case["lines_by_seed"]]
        exact = sum(any(mine[t] == s.get(t) for s in seen) for t in mine)
        # every target block must at least be a renumbering of a reference block
        for t in mine:
            assert sorted(map(_norm, mine[t])) == sorted(map(_norm, seen[0][t]))
        assert exact >= 0.5 * len(mine)
