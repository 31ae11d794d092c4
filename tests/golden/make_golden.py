#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the UNMODIFIED reference.

Runs only in the build container (needs /root/reference); nothing here travels
to the GPU box except the JSON/TSV fixtures it writes.

The reference's lookup layer is the third-party Jellyfish SWIG binding
(km/utils/Jellyfish.py:9-12,24-25,50-53), which is absent from this image.  A
stand-in module named ``jellyfish`` (our own code, written to a temp dir at run
time, below) provides exactly the surface the reference consumes --
``QueryMerFile(path)``, ``qmf[MerDNA]``, ``MerDNA(str)``, ``MerDNA.k()``,
``MerDNA.canonicalize()`` -- on top of oracle/jf_reader.py.  The stand-in is
pinned by the reference's own known-answer tests (km/tests/test_main.py), which
this script runs first (``--selftest``): numbers in those tests were produced by
real Jellyfish.

Every case is run in a fresh interpreter under several PYTHONHASHSEEDs, because
the reference iterates a ``set`` of k-mer strings (MutationFinder.py:100,115);
only seed-stable outputs are stored, unstable cases are recorded as such.

usage:  python tests/golden/make_golden.py [--selftest] [--out tests/golden]
"""

import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
SEEDS = [0, 1, 2, 3, 4, 5]

STANDIN = r'''
"""Stand-in for the Jellyfish python binding (own code; see make_golden.py)."""
import sys
sys.path.insert(0, %(repo)r)
from oracle import jf_reader as _jr

PROBES = [0]          # logical probe counter (one per QueryMerFile lookup)


class MerDNA:
    _k = 0

    def __init__(self, seq):
        self.v = _jr.pack(seq)
        self.n = len(seq)

    @staticmethod
    def k():
        return MerDNA._k

    def canonicalize(self):
        self.v = _jr.canonical(self.v, self.n)


class QueryMerFile:
    def __init__(self, path):
        db = _jr.read_jf(path)
        MerDNA._k = db["k"]
        self.table = dict(zip(db["keys"].tolist(), db["counts"].tolist()))

    def __getitem__(self, mer):
        PROBES[0] += 1
        return self.table.get(mer.v, 0)
'''

# Worker: executed in a child interpreter with the stand-in first on sys.path.
WORKER = r'''
import io, json, sys, os, contextlib, argparse
sys.path.insert(0, %(ref)r)
sys.path.insert(0, %(standin_dir)r)
import jellyfish
from km.tools import find_mutation as fm
from km.utils import MutationFinder as umf, common as uc, Sequence as us
from km.utils.Jellyfish import Jellyfish

job = json.loads(sys.argv[1])
out = {}
if job["kind"] == "tsv":
    ns = argparse.Namespace(count=job.get("count", 5), ratio=job.get("ratio", 0.05),
                            steps=job.get("steps", 500), branchs=job.get("branchs", 10),
                            nodes=job.get("nodes", 10000), graphical=False, verbose=False,
                            debug=False, target_fn=job["targets"], jellyfish_fn=job["db"])
    buf = io.StringIO()
    code = None
    with contextlib.redirect_stdout(buf):
        try:
            fm.main_find_mut(ns, None)
        except SystemExit as e:
            code = str(e)
    lines = [l for l in buf.getvalue().splitlines() if not l.startswith("#Elapsed time")]
    out = {"lines": lines, "exit": code}
elif job["kind"] == "walk":
    jf = Jellyfish(job["db"], cutoff=job.get("ratio", 0.05), n_cutoff=job.get("count", 5))
    res = []
    for t in job["targets"]:
        name = os.path.splitext(os.path.basename(t))[0]
        seqs, _ = uc.file_2_seq(t)
        ref = us.RefSeq("".join(seqs), name, jf.k)
        jellyfish.PROBES[0] = 0
        f = umf.MutationFinder(ref, jf, job.get("steps", 500), job.get("branchs", 10),
                               job.get("nodes", 10000))
        probes = jellyfish.PROBES[0]
        f.graph_analysis()
        paths = sorted("".join(f.kmer[i][-1] if j else f.kmer[i] for j, i in enumerate(p.seq_index))
                       for p in f.alt_paths)
        mincov = {}
        for p in f.alt_paths:
            mincov[p.seq] = min(f.get_counts(p.seq_index))
        res.append({"name": name, "n_ref": len(ref.ref_mer), "num_k": f.num_k,
                    "probes": probes,
                    "nodes": sorted([k, int(v)] for k, v in f.node_data.items()),
                    "path_seqs": paths,
                    "path_min_cov": [mincov[s] for s in paths]})
    out = {"targets": res}
elif job["kind"] == "graphlog":
    # the INFO records the reference logs from inside the walk and the graph (km/utils/MutationFinder.py:161,
    # km/utils/Graph.py:198,231): how many reference edges it strips, how many edges stay, where it breaks a loop
    import logging, re

    class Grab(logging.Handler):
        def __init__(self):
            super().__init__(logging.INFO)
            self.msgs = []

        def emit(self, rec):
            self.msgs.append(rec.getMessage())

    grab = Grab()
    root = logging.getLogger()                 # the reference logs through the root logger (`import logging as log`)
    root.setLevel(logging.INFO)
    root.addHandler(grab)
    jf = Jellyfish(job["db"], cutoff=job.get("ratio", 0.05), n_cutoff=job.get("count", 5))
    res = []
    for t in job["targets"]:
        name = os.path.splitext(os.path.basename(t))[0]
        seqs, _ = uc.file_2_seq(t)
        ref = us.RefSeq("".join(seqs), name, jf.k)
        grab.msgs = []
        f = umf.MutationFinder(ref, jf, job.get("steps", 500), job.get("branchs", 10), job.get("nodes", 10000))
        f.graph_analysis()
        removed = [int(re.match(r"Removed (\d+) ref edges", m).group(1)) for m in grab.msgs if m.startswith("Removed ")]
        nonref = [int(re.match(r"(\d+) edges in non-ref edge set", m).group(1)) for m in grab.msgs if "edges in non-ref edge set" in m]
        loops = [m.split(": ", 1)[1] for m in grab.msgs if m.startswith("Broke loop at kmer")]
        res.append({"name": name, "removed_ref_edges": removed, "nonref_edges": nonref, "loop_kmers": loops,
                    "num_k": f.num_k})
    out = {"targets": res}
elif job["kind"] == "children":
    jf = Jellyfish(job["db"], cutoff=job.get("ratio", 0.05), n_cutoff=job.get("count", 5))
    res = []
    for s in job["kmers"]:
        res.append([s, jf.query(s), jf.get_child(s, forward=True)])
    out = {"children": res}
elif job["kind"] == "report":
    from km.tools import find_report as fr
    ns = argparse.Namespace(target=job["target"], infile=io.StringIO("\n".join(job["lines"]) + "\n"),
                            info=job.get("info", "vs_ref"), min_cov=job.get("min_cov", 1),
                            exclu=job.get("exclu", ""), format=job.get("format"))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        fr.create_report(ns)
    out = {"lines": buf.getvalue().splitlines()}
elif job["kind"] == "min_cov":
    seqs, _ = uc.file_2_seq(job["target"])
    out = {"cov": list(uc.get_cov(job["db"], "".join(seqs)))}
print(json.dumps(out))
'''


def _env(seed):
    env = dict(os.environ)
    env["PYTHONHASHSEED"] = str(seed)
    return env


class Runner:
    def __init__(self):
        self.tmp = tempfile.mkdtemp(prefix="km_standin_")
        with open(os.path.join(self.tmp, "jellyfish.py"), "w") as fh:
            fh.write(STANDIN % {"repo": REPO})
        self.worker = os.path.join(self.tmp, "worker.py")
        with open(self.worker, "w") as fh:
            fh.write(WORKER % {"ref": REF, "standin_dir": self.tmp})

    def run(self, job, seed, cwd=REF):
        p = subprocess.run([sys.executable, self.worker, json.dumps(job)], cwd=cwd,
                           env=_env(seed), capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError("worker failed: %s\n%s" % (job, p.stderr[-2000:]))
        return json.loads(p.stdout.strip().splitlines()[-1])

    def stable(self, job, seeds=SEEDS, cwd=REF):
        """Run under every seed; return (result_of_first, is_stable)."""
        outs = [self.run(job, s, cwd) for s in seeds]
        blobs = [json.dumps(o, sort_keys=True) for o in outs]
        return outs[0], all(b == blobs[0] for b in blobs), outs

    def selftest(self):
        """Run the reference's own test-suite against the stand-in."""
        env = _env(0)
        env["PYTHONPATH"] = self.tmp + os.pathsep + REF
        p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider",
                            "km/tests/test_main.py"], cwd=REF, env=env,
                           capture_output=True, text=True)
        print(p.stdout[-1500:])
        return p.returncode == 0


FIXTURE_PAIRS = [
    ("NPM1_4ins_exons_10-11utr.fa", "02H025_NPM1.jf"),
    ("FLT3-ITD_exons_13-15.fa", "03H116_ITD.jf"),
    ("FLT3-ITD_exons_13-15.fa", "03H112_IandI.jf"),
    ("FLT3-TKD_exon_20.fa", "05H094_FLT3-TKD_del.jf"),
    ("DNMT3A_R882_exon_23.fa", "02H033_DNMT3A_sub.jf"),
]
CATALOG = sorted(os.listdir(os.path.join(REF, "data/catalog/GRCh38"))) if os.path.isdir(REF) else []
DBS = ["02H025_NPM1.jf", "02H033_DNMT3A_sub.jf", "03H112_IandI.jf", "03H116_ITD.jf",
       "05H094_FLT3-TKD_del.jf"]


def md5_lines(lines):
    return hashlib.md5(("\n".join(lines) + "\n").encode()).hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--selftest", action="store_true")
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    r = Runner()
    if args.selftest:
        ok = r.selftest()
        print("reference test-suite vs stand-in:", "PASS" if ok else "FAIL")
        sys.exit(0 if ok else 1)

    want = set(args.only.split(",")) if args.only else None

    def on(name):
        return want is None or name in want

    # ---- 1. bundled fixtures: TSV of the reference CLI driver ------------------
    if on("fixtures"):
        gold = {"cases": []}
        for fa, db in FIXTURE_PAIRS:
            job = {"kind": "tsv", "targets": ["./data/catalog/GRCh38/" + fa],
                   "db": "./data/jf/" + db}
            o, st, _ = r.stable(job)
            gold["cases"].append({"targets": job["targets"], "db": job["db"], "stable": st,
                                  "md5": md5_lines(o["lines"]), "lines": o["lines"],
                                  "exit": o["exit"]})
            print("tsv", fa, db, "stable" if st else "UNSTABLE", md5_lines(o["lines"]))
        # whole catalog (explicit, sorted file list -> independent of os.listdir order)
        for db in DBS:
            job = {"kind": "tsv", "targets": ["./data/catalog/GRCh38/" + f for f in CATALOG],
                   "db": "./data/jf/" + db}
            o, st, _ = r.stable(job)
            gold["cases"].append({"targets": job["targets"], "db": job["db"], "stable": st,
                                  "md5": md5_lines(o["lines"]), "lines": o["lines"],
                                  "exit": o["exit"]})
            print("tsv catalog", db, "stable" if st else "UNSTABLE", len(o["lines"]))
        with open(os.path.join(args.out, "fixtures_tsv.json"), "w") as fh:
            json.dump(gold, fh, indent=1)

    # ---- 2. walk-level vectors: node sets, probe counts, path sequences ---------
    if on("walk"):
        gold = {"cases": []}
        for db in DBS:
            job = {"kind": "walk", "targets": ["./data/catalog/GRCh38/" + f for f in CATALOG],
                   "db": "./data/jf/" + db}
            o, st, outs = r.stable(job)
            # probe counts may legitimately vary with seed; keep the set seen
            for ti, t in enumerate(o["targets"]):
                t["probes_seen"] = sorted({x["targets"][ti]["probes"] for x in outs})
            gold["cases"].append({"db": job["db"], "targets_fa": job["targets"],
                                  "stable": st, "targets": o["targets"]})
            print("walk", db, "stable" if st else "UNSTABLE",
                  [(t["name"][:6], t["num_k"], t["probes"]) for t in o["targets"]])
        with open(os.path.join(args.out, "fixtures_walk.json"), "w") as fh:
            json.dump(gold, fh)

    # ---- 3. get_child / query vectors + min_cov ----------------------------------
    if on("children"):
        sys.path.insert(0, REPO)
        from oracle import jf_reader as jr
        import numpy as np
        gold = {"cases": []}
        rng = np.random.default_rng(7)
        for db in DBS:
            d = jr.read_jf(os.path.join(REF, "data/jf", db))
            pick = rng.choice(len(d["keys"]), size=min(60, len(d["keys"])), replace=False)
            kmers = []
            for i in pick:
                s = jr.unpack(d["keys"][i], d["k"])
                kmers.append(s)
                kmers.append(jr.unpack(jr.revcomp(d["keys"][i], d["k"]), d["k"]))
            kmers.append("A" * d["k"])
            for cnt, ratio in ((5, 0.05), (500, 0.30), (2, 0.0)):
                job = {"kind": "children", "db": "./data/jf/" + db, "kmers": kmers,
                       "count": cnt, "ratio": ratio}
                o, st, _ = r.stable(job, seeds=[0, 1])
                gold["cases"].append({"db": job["db"], "count": cnt, "ratio": ratio,
                                      "children": o["children"]})
        mc = []
        for db in DBS:
            job = {"kind": "min_cov", "db": "./data/jf/" + db,
                   "target": "./data/catalog/GRCh38/FLT3-ITD_exons_13-15.fa"}
            o = r.run(job, 0)
            mc.append({"db": job["db"], "target": job["target"], "cov": o["cov"]})
        gold["min_cov"] = mc
        with open(os.path.join(args.out, "fixtures_children.json"), "w") as fh:
            json.dump(gold, fh)
        print("children/min_cov vectors written")

    # ---- 5. sample matrix: every catalog target against the 5 samples, the
    #         per-target TSV stream `find_report -f table` consumes, and what it prints
    if on("matrix"):
        gold = {"samples": ["./data/jf/" + d for d in DBS], "targets": []}
        for fa in CATALOG:
            tfa = "./data/catalog/GRCh38/" + fa
            stream = []
            for db in DBS:
                o, st, _ = r.stable({"kind": "tsv", "targets": [tfa], "db": "./data/jf/" + db}, seeds=[0, 1, 2])
                assert st and o["exit"] is None, (fa, db)
                stream += o["lines"]
            table = r.run({"kind": "report", "target": tfa, "lines": stream, "format": "table"}, 0)["lines"]
            plain = r.run({"kind": "report", "target": tfa, "lines": stream, "format": None}, 0)["lines"]
            gold["targets"].append({"target": tfa, "stream": stream, "find_report_table": table,
                                    "find_report": plain})
            print("matrix", fa, len(stream), "lines ->", len(table), "table lines")
        # find_report -e <exclusion db>: Exclu_min_cov = common.get_cov(exclu, variant sequence)[2]
        gold["exclu"] = []
        for fa, excl in (("FLT3-ITD_exons_13-15.fa", "03H112_IandI.jf"), ("NPM1_4ins_exons_10-11utr.fa", "03H116_ITD.jf"),
                         ("DNMT3A_R882_exon_23.fa", "02H033_DNMT3A_sub.jf")):
            tfa = "./data/catalog/GRCh38/" + fa
            stream = [t for t in gold["targets"] if t["target"] == tfa][0]["stream"]
            rep = r.run({"kind": "report", "target": tfa, "lines": stream, "format": None,
                         "exclu": "./data/jf/" + excl}, 0)["lines"]
            gold["exclu"].append({"target": tfa, "exclu": "./data/jf/" + excl, "find_report": rep})
        with open(os.path.join(args.out, "sample_matrix.json"), "w") as fh:
            json.dump(gold, fh, indent=1)

    # ---- 6. the -v lines of the walk and the graph: stripped reference edges, edges left, loop breaks
    if on("graphlog"):
        sys.path.insert(0, REPO)
        from km_amd import synth
        gold = {"cases": []}
        for db in DBS:
            job = {"kind": "graphlog", "targets": ["./data/catalog/GRCh38/" + f for f in CATALOG], "db": "./data/jf/" + db}
            o, st, outs = r.stable(job, seeds=[0, 1, 2])
            for ti, t in enumerate(o["targets"]):
                t["loop_kmers_by_seed"] = [sorted(x["targets"][ti]["loop_kmers"]) for x in outs]
            gold["cases"].append({"db": job["db"], "targets_fa": job["targets"], "stable": st, "targets": o["targets"]})
            print("graphlog", db, "stable" if st else "UNSTABLE",
                  [(t["name"][:6], t["removed_ref_edges"], t["nonref_edges"], len(t["loop_kmers"])) for t in o["targets"]])
        for spec in synth.GOLDEN_SPECS:
            with tempfile.TemporaryDirectory() as td:
                fas, dbp, meta = synth.write_case(td, **spec)
                job = {"kind": "graphlog", "targets": fas, "db": dbp}
                job.update(spec.get("params", {}))
                if spec.get("params", {}).get("nodes", 10000) < 1000:
                    continue                                   # (the node-limit case exits before the graph)
                o, st, outs = r.stable(job, seeds=[0, 1, 2], cwd=td)
                for ti, t in enumerate(o["targets"]):
                    t["loop_kmers_by_seed"] = [sorted(x["targets"][ti]["loop_kmers"]) for x in outs]
                gold["cases"].append({"spec": spec, "input_md5": meta["md5"], "stable": st, "targets": o["targets"]})
                print("graphlog synth", spec.get("name"), "stable" if st else "UNSTABLE",
                      sum(len(t["loop_kmers"]) for t in o["targets"]), "loop breaks")
        with open(os.path.join(args.out, "graph_log.json"), "w") as fh:
            json.dump(gold, fh)

    # ---- 4. synthetic slices (generator: km_amd/synth.py) -------------------------
    if on("synth"):
        sys.path.insert(0, REPO)
        from km_amd import synth
        gold = {"cases": []}
        for spec in synth.GOLDEN_SPECS:
            with tempfile.TemporaryDirectory() as td:
                fas, dbp, meta = synth.write_case(td, **spec)
                job = {"kind": "tsv", "targets": fas, "db": dbp}
                job.update(spec.get("params", {}))
                o, st, outs = r.stable(job, cwd=td)
                wjob = dict(job, kind="walk")
                w, wst, wouts = r.stable(wjob, cwd=td)
                # strip the temp dir from the Database column / header echo
                def scrub(lines):
                    return [l.replace(td + "/", "") for l in lines]
                case = {"spec": spec, "input_md5": meta["md5"], "stable": st,
                        "walk_stable": wst, "lines": scrub(o["lines"]), "exit": o["exit"],
                        "lines_by_seed": None if st else [scrub(x["lines"]) for x in outs],
                        "walk": [{"name": t["name"], "num_k": t["num_k"], "probes": t["probes"],
                                  "nodes_md5": hashlib.md5(json.dumps(t["nodes"]).encode()).hexdigest(),
                                  "path_seqs": t["path_seqs"], "path_min_cov": t["path_min_cov"]}
                                 for t in w["targets"]]}
                gold["cases"].append(case)
                print("synth", spec.get("name"), "tsv", "stable" if st else "UNSTABLE",
                      "walk", "stable" if wst else "UNSTABLE", len(o["lines"]), "lines")
        with open(os.path.join(args.out, "synth_tsv.json"), "w") as fh:
            json.dump(gold, fh)


if __name__ == "__main__":
    main()
