"""Pins the plain-C oracle (oracle/km_oracle.c) against the Python oracle, which is itself
pinned against the reference's golden vectors (tests/test_oracle_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from km_amd import kmer as km
from km_amd import synth
from oracle import c_oracle
from oracle import jf_reader as jr
from oracle import km_oracle as ko

HERE = os.path.dirname(os.path.abspath(__file__))


def _same(c, p, k):
    assert c["status"] == 0
    assert [km.unpack(x, k) for x in c["kmers"]] == p["kmers"]
    assert c["counts"].tolist() == p["counts"]
    assert c["probes"] == p["probes"]
    assert c["paths"] == [list(x) for x in p["paths"]]
    assert c["min_cov"] == p["min_cov"]


@pytest.mark.parametrize("dbname", ["02H025_NPM1.jf", "03H116_ITD.jf", "03H112_IandI.jf",
                                    "05H094_FLT3-TKD_del.jf", "02H033_DNMT3A_sub.jf"])
def test_c_oracle_on_fixture_catalog(dbname, monkeypatch):
    monkeypatch.chdir(HERE)
    d = jr.read_jf("./data/jf/" + dbname)
    co = c_oracle.COracle(d["keys"], d["counts"], d["k"], d["canonical"])
    py = ko.KmerDB("./data/jf/" + dbname, cutoff=0.05, n_cutoff=5)
    for f in sorted(os.listdir("./data/catalog/GRCh38")):
        seq = ko.read_fasta_concat("./data/catalog/GRCh38/" + f)
        _same(co.analyse(km.encode(seq)), ko.analyse_target(seq, f, py), 31)


@pytest.mark.parametrize("name", ["cfg4_small", "stress", "lowcov", "tight", "k21"])
def test_c_oracle_on_synthetic(name):
    spec = [s for s in synth.GOLDEN_SPECS if s["name"] == name][0]
    case = synth.make_case(**spec)
    k = case["k"]
    prm = spec.get("params", {})
    co = c_oracle.COracle(case["keys"], case["counts"], k)
    py = ko.KmerDB(None, 0.05, 5, records={"k": k, "canonical": True, "keys": case["keys"],
                                           "counts": case["counts"]})
    for row, nm in list(zip(case["targets"], case["names"]))[:30]:
        c = co.analyse(row, max_stack=prm.get("steps", 500), max_break=prm.get("branchs", 10))
        p = ko.analyse_target(km.decode(row), nm, py, prm.get("steps", 500), prm.get("branchs", 10))
        _same(c, p, k)


def test_c_oracle_statuses():
    case = synth.make_case(n_targets=12, length=300, n_keys=10000, seed=14, variant_frac=1.0, kinds=("dup",))
    co = c_oracle.COracle(case["keys"], case["counts"], 31)
    py = ko.KmerDB(None, 0.05, 5, records={"k": 31, "canonical": True, "keys": case["keys"],
                                           "counts": case["counts"]})
    hit = 0
    for row in case["targets"]:
        c = co.analyse(row, max_node=272)
        try:
            ko.analyse_target(km.decode(row), "t", py, 500, 10, 272)
            assert c["status"] == 0
        except ko.NodeLimit:
            assert c["status"] == 1
            hit += 1
    assert hit > 0
    assert co.analyse(np.zeros(32, np.uint8))["status"] == 2        # poly-A: repeated k-mer
    assert co.analyse(np.zeros(10, np.uint8))["status"] == 3        # shorter than k
