"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): one broadcast of the
database records, target sharding, ordered gather.  The per-shard compute is done
by the oracle here (no GPU in this container); on a GPU box the same plumbing
drives the HIP path (bench.py, km_amd/dist.py)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np, torch
import torch.distributed as dist
from km_amd import dist as kd
from oracle import jf_reader as jr, km_oracle as ko

os.chdir(%(here)r)
rank, local_rank, world = kd.init(backend="gloo")
assert world == 2 and dist.get_backend() == "gloo"
cat = sorted(os.listdir("./data/catalog/GRCh38"))
targets = [(os.path.splitext(f)[0], ko.read_fasta_concat("./data/catalog/GRCh38/" + f)) for f in cat]
DB = "./data/jf/03H116_ITD.jf"

def load(path):
    d = jr.read_jf(path)
    return d["keys"], d["counts"], d["k"], d["canonical"]

seen = {}
def analyse(d_keys, d_cnts, n, k, canonical, mine):
    keys = d_keys.numpy().view(np.uint64)
    cnts = d_cnts.numpy().view(np.uint32)
    seen["n"] = n
    db = ko.KmerDB(DB, cutoff=0.05, n_cutoff=5,
                   records={"k": k, "canonical": canonical, "keys": keys, "counts": cnts})
    return [ko.target_rows(ko.analyse_target(seq, name, db), DB) for name, seq in mine]

rows = kd.find_mutation_sharded(targets, DB, analyse, load)
lo, hi = kd.shard_range(len(targets), rank, world)
out = {"rank": rank, "shard": [lo, hi], "n_records": seen["n"]}
if rank == 0:
    out["rows"] = [r for per_target in rows for r in per_target]

# sample-sharded: each rank opens its own databases, no broadcast
SAMPLES = ["./data/jf/" + f for f in sorted(os.listdir("./data/jf"))]
opened = []
def run_sample(path):
    opened.append(path)
    db = ko.KmerDB(path, cutoff=0.05, n_cutoff=5)
    return [r for name, seq in targets for r in ko.target_rows(ko.analyse_target(seq, name, db), path)]
per_sample = kd.find_mutation_samples(SAMPLES, run_sample)
out["opened"] = opened
if rank == 0:
    out["per_sample"] = per_sample

# the sample-matrix driver (one TSV stream per target) through the same orchestration
def run_sample_blocks(path):
    db = ko.KmerDB(path, cutoff=0.05, n_cutoff=5)
    return [ko.target_rows(ko.analyse_target(seq, name, db), path) for name, seq in targets]
files = kd.sample_matrix(SAMPLES, ["./data/catalog/GRCh38/" + f for f in cat], %(outdir)r,
                         run_sample=run_sample_blocks, read_target=ko.read_fasta_concat)
if rank == 0:
    out["matrix_files"] = files
print("RESULT " + json.dumps(out), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_range_covers_everything():
    from km_amd import dist as kd
    for n in (0, 1, 7, 9, 10000):
        for world in (1, 2, 3, 8):
            spans = [kd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_matches_single_process(tmp_path):
    script = tmp_path / "worker.py"
    outdir = str(tmp_path / "matrix")
    script.write_text(WORKER % {"root": ROOT, "here": HERE, "outdir": outdir})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29500 + (os.getpid() % 2000)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                        "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    res = [json.loads(l[len("RESULT "):]) for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    assert len(res) == 2
    by_rank = {r["rank"]: r for r in res}
    assert by_rank[0]["shard"] == [0, 5] and by_rank[1]["shard"] == [5, 9]
    assert by_rank[0]["n_records"] == by_rank[1]["n_records"] == 2560      # the broadcast arrived
    gold = json.load(open(os.path.join(HERE, "golden", "fixtures_tsv.json")))
    case = [c for c in gold["cases"] if len(c["targets"]) == 9 and c["db"].endswith("03H116_ITD.jf")][0]
    assert by_rank[0]["rows"] == case["lines"][11:]      # after the 10 '#' lines + header
    # sample-sharded run: 5 databases over 2 ranks, rows back in sample order
    dbs = sorted(os.listdir(os.path.join(HERE, "data", "jf")))
    assert by_rank[0]["opened"] == ["./data/jf/" + d for d in dbs[0::2]]
    assert by_rank[1]["opened"] == ["./data/jf/" + d for d in dbs[1::2]]
    for d, rows in zip(dbs, by_rank[0]["per_sample"]):
        want = [c for c in gold["cases"] if len(c["targets"]) == 9 and c["db"].endswith(d)][0]
        assert rows == want["lines"][11:]
    # sample matrix: per-target streams equal the reference's concatenated find_mutation outputs
    mat = json.load(open(os.path.join(HERE, "golden", "sample_matrix.json")))
    assert len(by_rank[0]["matrix_files"]) == len(mat["targets"]) == 9
    for f, want in zip(by_rank[0]["matrix_files"], mat["targets"]):
        got = [l for l in open(f).read().splitlines() if not l.startswith("#Elapsed time")]
        assert got == want["stream"], f


WORKER2 = r'''
import io, os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np, torch
import torch.distributed as dist
from km_amd import dist as kd
from km_amd.cli import _print_rows
from km_amd.finder import NodeLimitExceeded
from oracle import jf_reader as jr, km_oracle as ko

os.chdir(%(here)r)
rank, local_rank, world = kd.init(backend="gloo")
cat = sorted(os.listdir("./data/catalog/GRCh38"))
targets = [(os.path.splitext(f)[0], ko.read_fasta_concat("./data/catalog/GRCh38/" + f)) for f in cat]
DB = "./data/jf/03H116_ITD.jf"
MODE = %(mode)r

def load(path):
    d = jr.read_jf(path)
    return d["keys"], d["counts"], d["k"], d["canonical"]

calls = []
def analyse(d_keys, d_cnts, n, k, canonical, mine):
    calls.append([name for name, _ in mine])
    db = ko.KmerDB(DB, cutoff=0.05, n_cutoff=5, records={"k": k, "canonical": canonical,
                   "keys": d_keys.numpy().view(np.uint64), "counts": d_cnts.numpy().view(np.uint32)})
    rows = []
    for name, seq in mine:
        if MODE == "node_limit" and name == targets[6][0]:
            rows.append(NodeLimitExceeded(123))
        elif MODE == "input_error" and name == targets[7][0]:
            raise ValueError("%%s found multiple times in reference %%s, at pos. %%d" %% ("ACGT", name, 5))
        else:
            rows.append(ko.target_rows(ko.analyse_target(seq, name, db), DB))
    return rows

out = {"rank": rank, "world": world}
try:
    rows = kd.find_mutation_sharded(targets, DB, analyse, load, chunk=2)      # 5 pieces dealt round-robin
    out["calls"] = calls
    if rank == 0:
        if MODE == "node_limit":
            buf = io.StringIO()
            try:
                _print_rows(rows, buf)
                out["exit"] = None
            except SystemExit as e:
                out["exit"] = str(e)
            out["printed"] = buf.getvalue().splitlines()
        else:
            out["rows"] = [r for per_target in rows for r in per_target]
except ValueError as e:
    out["raised"] = str(e)
json.dump(out, open(os.path.join(%(outdir)r, "result_%%d.json" %% rank), "w"))
dist.barrier()
dist.destroy_process_group()
'''


def _run_ranks(tmp_path, world, mode):
    script = tmp_path / ("worker_%s_%d.py" % (mode, world))
    outdir = tmp_path / ("out_%s_%d" % (mode, world))
    outdir.mkdir()
    script.write_text(WORKER2 % {"root": ROOT, "here": HERE, "mode": mode, "outdir": str(outdir)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 31500 + (os.getpid() % 2000) + world
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    res = [json.load(open(str(outdir / ("result_%d.json" % r)))) for r in range(world)]     # (one file per rank:
    return {r["rank"]: r for r in res}                                                          # stdout lines interleave)


def _golden_rows():
    gold = json.load(open(os.path.join(HERE, "golden", "fixtures_tsv.json")))
    case = [c for c in gold["cases"] if len(c["targets"]) == 9 and c["db"].endswith("03H116_ITD.jf")][0]
    return case["lines"][11:]


@pytest.mark.parametrize("world", [2, 3])
def test_round_robin_chunks_match_single_process(tmp_path, world):
    """A catalog larger than one chunk per rank: successive chunks dealt round-robin (km_amd.dist.shard_plan),
    every rank ONE call of the compute over its pieces, rows back in target order on rank 0."""
    from km_amd import dist as kd
    by_rank = _run_ranks(tmp_path, world, "plain")
    names = sorted(os.path.splitext(f)[0] for f in os.listdir(os.path.join(HERE, "data", "catalog", "GRCh38")))
    plan = kd.shard_plan(9, world, 2)
    assert [r for _, _, r in plan] == [q % world for q in range(5)]
    for r in range(world):
        assert by_rank[r]["calls"] == [[n for lo, hi, rr in plan if rr == r for n in names[lo:hi]]]
    assert by_rank[0]["rows"] == _golden_rows()


def test_node_limit_and_input_errors_cross_the_ranks(tmp_path):
    """A node-limit result gathered from another rank is still the reference's exit message after the
    rows of the earlier targets (MutationFinder.py:143-148; the exception survives pickling), and an input
    error raised on another rank reaches rank 0 through the gather instead of leaving it waiting."""
    by_rank = _run_ranks(tmp_path, 2, "node_limit")
    assert by_rank[0]["exit"] == "ERROR: Node query count limit exceeded: max=123"
    gold = _golden_rows()
    printed = by_rank[0]["printed"]
    assert printed and printed == gold[:len(printed)] and len(printed) < len(gold)
    by_rank = _run_ranks(tmp_path, 2, "input_error")
    assert "found multiple times in reference" in by_rank[0]["raised"]
    assert "raised" not in by_rank[1] or by_rank[1]["raised"] == by_rank[0]["raised"]
