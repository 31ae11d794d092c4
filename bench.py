#!/usr/bin/env python3
"""Benchmark of the find_mutation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[3]): 10 000 random 500-nt targets per step,
k=31, against ONE table of 100 M distinct canonical 31-mers resident in HBM.

A step = one pass of the hot path over one 10 000-target batch: k_pack + k_seed + k_dfs (walk),
k_graph_pure + k_graph (path search), device-side compaction of the results and their D2H copy
into pinned host memory.  `value` = targets/s of the whole pipeline INCLUDING result delivery
(targets/s = n_targets / (walk + graph kernels + D2H)); the kernel-only rate is
reported beside it.  Targets and table are in HBM before the timed region starts.  The
`--inflight` workspaces each hold a DIFFERENT 10 000-target set (working set >> the 256 MiB
Infinity Cache), and the run checks 200 targets against the plain-C oracle given the full key
set, exiting non-zero on a mismatch.

With --gpus N > 1 and no torchrun environment the script starts N ranks itself
(python -m torch.distributed.run, before any GPU call) and relays rank 0's JSON line.  The table
records cross the links once (RCCL broadcast, every rank builds its own table); targets are
sharded by rank, no collective on the data path.  `value` is weak scaling (10 000 targets per
GPU per step); BASELINE config 4 proper (10 000 targets sharded over the N GPUs) is reported
under "config4_strong".

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_PROBE = 12           # 8-byte key + 4-byte count
K = 31


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------- CPU baseline
_CPU = {}


def _cpu_init(case, n_need):
    from oracle import km_oracle as ko
    nr = case["n_real"]
    # the non-pad keys of the sampled targets (the oracle's dict cannot hold 100 M keys in every
    # worker; a walk that reached a key outside this subset would differ from the GPU's, which the
    # in-run check — full key set — would catch)
    sel = case["key_target"] < n_need
    rec = {"k": K, "canonical": True, "keys": case["keys"][:nr][sel], "counts": case["counts"][:nr][sel]}
    _CPU["db"] = ko.KmerDB(None, cutoff=0.05, n_cutoff=5, records=rec)
    _CPU["case"] = case
    _CPU["n_keys"] = int(sel.sum())


def _cpu_work(span, phases=None):
    from km_amd import kmer as km
    from oracle import km_oracle as ko
    case, db = _CPU["case"], _CPU["db"]
    probes = 0
    t0 = time.perf_counter()
    for i in range(*span):
        seq, name = km.decode(case["targets"][i]), case["names"][i]
        if phases is None:
            res = ko.analyse_target(seq, name, db)
        else:                                  # the same calls, timed per phase
            ta = time.perf_counter()
            mers = ko.ref_kmers(seq, name, db.k)
            p0 = db.probes
            nodes = ko.walk(mers, db)
            tb = time.perf_counter()
            kmers, counts = list(nodes.keys()), list(nodes.values())
            paths = ko.graph_paths(kmers, len(mers))
            tc = time.perf_counter()
            res = {"name": name, "k": db.k, "n_ref": len(mers), "kmers": kmers, "counts": counts, "paths": paths,
                   "probes": db.probes - p0, "min_cov": [min(counts[j] for j in p) for p in paths]}
            ko.target_rows(res, "synthetic.jf")
            td = time.perf_counter()
            phases[0] += tb - ta; phases[1] += tc - tb; phases[2] += td - tc
        probes += res["probes"]
    return probes, time.perf_counter() - t0


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def usable_cores():
    """Host cores this process may really use (affinity mask and cgroup quota, not the machine's)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(txt[0]) // int(txt[1])))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
        except Exception:
            pass
    return max(1, min(n, 64))


def cpu_baseline(case, n_sample):
    """The oracle (structure-faithful Python restatement of the reference path: str k-mers, one
    lookup per query, recursive extend, dense numpy Dijkstra) timed on the host: one core (the
    reference is single-threaded) and all cores (one process per core, targets sharded).  Must
    run BEFORE this process touches the GPU (it forks)."""
    import multiprocessing as mp
    cores = usable_cores()
    per = max(8, n_sample // 4)
    n_all = per * cores
    n_need = min(len(case["targets"]), max(n_sample, n_all))
    t_load = time.perf_counter()
    _cpu_init(case, n_need)
    t_load = time.perf_counter() - t_load
    n_sample = min(n_sample, n_need)
    probes, dt = _cpu_work((0, n_sample))
    ph = [0.0, 0.0, 0.0]
    n_ph = min(60, n_sample)
    _cpu_work((0, n_ph), ph)
    out = {"value": n_sample / dt, "unit": "targets/s", "cores": 1, "kind": "port",
           "sample": "first %d of the targets of set 0 (walk + path search, oracle/km_oracle.py, dict-backed "
                     "table of the %d keys the first %d targets touch)" % (n_sample, _CPU["n_keys"], n_need),
           "probes_per_s": probes / dt, "seconds": dt, "cpu_model": cpu_model(),
           "table_load_s": t_load,
           "phase_split": {"targets": n_ph, "walk_s": ph[0], "graph_s": ph[1], "naming_quantification_rows_s": ph[2],
                           "walk_frac": ph[0] / max(1e-12, sum(ph)), "graph_frac": ph[1] / max(1e-12, sum(ph)),
                           "report_frac": ph[2] / max(1e-12, sum(ph))}}
    try:
        ctx = mp.get_context("fork")
        spans = [(c * per, min(n_need, (c + 1) * per)) for c in range(cores)]
        spans = [s for s in spans if s[1] > s[0]]
        t0 = time.perf_counter()
        with ctx.Pool(len(spans)) as pool:
            parts = pool.map(_cpu_work, spans)
        wall = time.perf_counter() - t0
        n_done = sum(s[1] - s[0] for s in spans)
        out["all_cores"] = {"cores": len(spans), "value": n_done / wall, "unit": "targets/s",
                            "sample": "%d targets, %d per process" % (n_done, per), "seconds": wall,
                            "probes_per_s": sum(p for p, _ in parts) / wall}
    except Exception as e:                      # pragma: no cover
        out["all_cores"] = {"error": repr(e)}
    # the same work through the plain-C oracle (oracle/km_oracle.c), for scale
    try:
        from oracle import c_oracle
        db = _CPU["db"]
        co = c_oracle.COracle(np.fromiter(db.table.keys(), dtype=np.uint64, count=len(db.table)),
                              np.fromiter(db.table.values(), dtype=np.uint32, count=len(db.table)), K) \
            if hasattr(db, "table") else None
        if co is not None:
            t1 = time.perf_counter()
            for i in range(n_need):
                co.analyse(case["targets"][i])
            out["c_oracle_targets_per_s"] = n_need / (time.perf_counter() - t1)
    except Exception as e:
        log("C oracle not timed:", e)
    _CPU.clear()
    return out


# ---------------------------------------------------------------------------- workload
def load_case(args, n_targets):
    from km_amd import synth
    tag = "%s/case_%d_%d_%d" % (args.cache, n_targets, args.length, args.keys)
    fields = ("keys", "counts", "targets", "key_target")
    if args.cache and os.path.exists(tag + "_keys.npy"):
        case = {f: np.load("%s_%s.npy" % (tag, f)) for f in fields}
        meta = json.load(open(tag + "_meta.json"))
        case.update(k=K, n_real=meta["n_real"], names=meta["names"])
        return case
    case = synth.make_case(n_targets=n_targets, length=args.length, k=K, n_keys=args.keys,
                           seed=synth.HEADLINE_SEED, exact_pad=False)
    if args.cache:
        os.makedirs(args.cache, exist_ok=True)
        for f in fields:
            np.save("%s_%s.npy" % (tag, f), case[f])
        json.dump({"n_real": int(case["n_real"]), "names": list(case["names"])}, open(tag + "_meta.json", "w"))
    return case


def oracle_check(case, views, set_ids, T, n_check):
    """`n_check` targets spread over the delivered batches against the plain-C oracle holding
    EVERY key of the table (pads included: a random pad 31-mer can neighbour a walk)."""
    from km_amd import lib as kmlib
    from oracle import c_oracle
    t0 = time.perf_counter()
    co = c_oracle.COracle(case["keys"], case["counts"], K)
    t_build = time.perf_counter() - t0
    ref = None
    n_done = n_multi = 0
    per = max(1, n_check // len(views))
    for v, sid in zip(views, set_ids):
        noff, xoff, poff = (v[x].astype(np.int64) for x in ("node_off", "extra_off", "path_off"))
        for t in range(0, T, max(1, T // per)):
            g = sid * T + t
            want = co.analyse(case["targets"][g])
            ok = int(v["status"][t]) == want["status"] == 0
            nr = int(v["n_ref"][t])
            if noff[t + 1] == noff[t]:        # lean delivery of a bare-reference target
                ok = ok and len(want["counts"]) == nr and int(v["ref_max_cov"][t]) == int(want["counts"].max())
                ok = ok and want["paths"] == [list(range(nr))]
            else:
                ok = ok and (v["node_count"][noff[t]:noff[t + 1]] == want["counts"]).all()
            ok = ok and (v["extra_kmer"][xoff[t]:xoff[t + 1]] == want["kmers"][nr:]).all()
            ok = ok and int(v["probes"][t]) == want["probes"]
            got = [kmlib.expand_path(v, p).tolist() for p in range(poff[t], poff[t + 1])]
            ok = ok and got == want["paths"]
            ok = ok and v["path_min_cov"][poff[t]:poff[t + 1]].tolist() == want["min_cov"]
            if not ok:
                return {"ok": False, "first_mismatch": {"set": sid, "target": t}, "checked": n_done}
            n_done += 1
            n_multi += len(got) > 1
    return {"ok": True, "checked": n_done, "with_variant_paths": n_multi, "oracle": "oracle/km_oracle.c",
            "oracle_keys": int(len(case["keys"])), "oracle_build_s": t_build}


def hard_workload(args, case, dev_index, stream, T, n_fl, profile_only=False):
    """BASELINE config 4's shape made hard (example/run_leucegene.sh:13-35 is what real catalogs
    look like): 85 % of the targets carry 1-3 variants of every kind (15 % of them homozygous), 3 % of the
    k-mers an above-threshold dead-end branch, 3 % sub-threshold noise, and 4 % of the variant targets 3-5
    tandem duplications (walks that outgrow the LDS tier).  Its table: the hard targets' own k-mers plus the
    headline case's random pads, 100 M keys.  Returns the `config4_hard` object; every number is this
    workload's own, including an oracle check against the plain-C oracle over the same table."""
    from km_amd import lib as kmlib, synth
    from oracle import c_oracle
    t0 = time.perf_counter()
    n_sets = 2
    hc = synth.make_case(n_targets=n_sets * T, length=args.length, k=K, n_keys=1, seed=synth.HEADLINE_SEED + 7,
                         variant_frac=0.85, variants_per_target=(1, 3), hom_frac=0.15, branch_noise_frac=0.03,
                         noise_frac=0.03, heavy_frac=0.04, exact_pad=False)
    real_k, real_c = hc["keys"][:hc["n_real"]], hc["counts"][:hc["n_real"]]
    pads_k, pads_c = case["keys"][case["n_real"]:], case["counts"][case["n_real"]:]
    n_pad = max(0, min(len(pads_k), args.keys - len(real_k)))
    pk, pc = pads_k[:n_pad], pads_c[:n_pad]
    pos = np.searchsorted(real_k, pk)
    pos[pos >= len(real_k)] = len(real_k) - 1
    ok = real_k[pos] != pk
    keys = np.concatenate([real_k, pk[ok]])
    cnts = np.concatenate([real_c, pc[ok]])
    t_gen = time.perf_counter() - t0
    db = kmlib.Database.from_records(keys, cnts, K).upload(dev_index)
    offsets = (np.arange(T + 1, dtype=np.uint64) * np.uint64(args.length))
    blob = np.frombuffer(b"ACGT", dtype=np.uint8)[hc["targets"]].copy()
    streams = [kmlib.stream_create(dev_index) for _ in range(n_fl)]
    batches = []
    for q in range(n_fl):
        bq = kmlib.Batch(db, ratio=0.05, count=5, max_stack=500, max_break=10, max_node=10000,
                         max_targets=T, max_total_bases=T * args.length)
        sid = q % n_sets
        bq.set_targets_packed(blob[sid * T:(sid + 1) * T].reshape(-1), offsets)
        batches.append(bq)
    stages = kmlib.KM_STAGE_WALK | kmlib.KM_STAGE_GRAPH
    deliver = stages | kmlib.KM_RUN_DELIVER | kmlib.KM_DELIVER_LEAN | kmlib.KM_DELIVER_COUNT16
    if profile_only:
        # every launch: one kernel alone (KM_RUN_SERIAL) over one of the two target sets, never the one before
        tm = []
        for i in range(args.warmup + args.steps):
            bq, sq = batches[i % n_fl], streams[i % n_fl]
            bq.run(deliver | kmlib.KM_RUN_TIMED | kmlib.KM_RUN_SERIAL, sq)
            sz = bq.wait_result()
            if i >= args.warmup:
                tm.append(bq.timings())
        tm = np.mean(np.array(tm), axis=0)
        probes = float(sz.logical_probes)
        out = {"steps": args.steps, "table_keys": int(len(keys)), "logical_probes_per_step": probes,
               "kernel_ms": {"walk": float(tm[0]), "k_pack": float(tm[4]), "k_seed": float(tm[3]), "k_dfs": float(tm[5]),
                             "graph": float(tm[1]), "deliver_kernels": float(tm[6]), "d2h_copy": float(tm[7])},
               "walk_stage_frac_of_hbm_peak": probes * BYTES_PER_PROBE / (float(tm[0]) * 1e-3) / 1e9 / HBM_PEAK_GBS,
               "targets_flagged": int(sz.n_flagged), "targets_in_large_tier": int(sz.n_big_tier)}
        for bq in batches:
            bq.close()
        db.close()
        return out
    kmlib.pump(batches, streams, max(n_fl, args.warmup), deliver)
    dts = []
    for _ in range(max(1, args.repeats)):
        t1 = time.perf_counter()
        kmlib.pump(batches, streams, args.steps, deliver)
        dts.append(time.perf_counter() - t1)
    dt = float(np.median(dts))
    sizes = [bq.wait_result() for bq in batches]
    views = [bq.result() for bq in batches]
    tm = []
    for i in range(min(args.steps, 40)):
        bq, sq = batches[i % n_fl], streams[i % n_fl]
        bq.run(deliver | kmlib.KM_RUN_TIMED | kmlib.KM_RUN_SERIAL, sq)
        bq.wait_result()
        tm.append(bq.timings())
    tm = np.mean(np.array(tm), axis=0)
    counts = [bq.debug_counts() for bq in batches[:n_sets]]
    # oracle check: targets spread over both sets, plus every large-tier target among the first 2 000 of set 0
    co = c_oracle.COracle(keys, cnts, K)
    check = {"ok": True, "checked": 0, "with_variant_paths": 0, "max_paths": 0}
    for sid in range(n_sets):
        v = views[sid]
        noff, xoff, poff = (v[x].astype(np.int64) for x in ("node_off", "extra_off", "path_off"))
        for t in range(0, T, max(1, T // max(1, args.check // n_sets))):
            want = co.analyse(hc["targets"][sid * T + t])
            nr = int(v["n_ref"][t])
            ok = int(v["status"][t]) == want["status"] == 0
            if noff[t + 1] == noff[t]:
                ok = ok and len(want["counts"]) == nr and int(v["ref_max_cov"][t]) == int(want["counts"].max())
                ok = ok and want["paths"] == [list(range(nr))]
            else:
                ok = ok and (v["node_count"][noff[t]:noff[t + 1]] == want["counts"]).all()
            ok = ok and (v["extra_kmer"][xoff[t]:xoff[t + 1]] == want["kmers"][nr:]).all()
            ok = ok and int(v["probes"][t]) == want["probes"]
            got = [kmlib.expand_path(v, p).tolist() for p in range(poff[t], poff[t + 1])]
            ok = ok and got == want["paths"] and v["path_min_cov"][poff[t]:poff[t + 1]].tolist() == want["min_cov"]
            if not ok:
                check = {"ok": False, "first_mismatch": {"set": sid, "target": t}, "checked": check["checked"]}
                break
            check["checked"] += 1
            check["with_variant_paths"] += len(got) > 1
            check["max_paths"] = max(check["max_paths"], len(got))
        if not check["ok"]:
            break
    probes = float(np.mean([int(s.logical_probes) for s in sizes]))
    out = {"workload": "synthetic_hard_%dx%dnt_targets_%dM_kmer_table_k31: 85%% of the targets with 1-3 variants (snv/ins/del/dup, "
                       "15%% homozygous), 3%% above-threshold dead-end branches, 3%% sub-threshold noise, 4%% of the variant targets "
                       "with 3-5 tandem duplications" % (T, args.length, round(len(keys) / 1e6)),
           "value": T / (dt / args.steps), "unit": "targets/s", "ms_per_step": dt / args.steps * 1e3,
           "ms_per_step_min": min(dts) / args.steps * 1e3, "ms_per_step_max": max(dts) / args.steps * 1e3,
           "distinct_target_sets": n_sets, "batches_in_flight": n_fl, "delivery": "lean",
           "logical_probes_per_step": probes, "gprobes_per_s": probes / (dt / args.steps) / 1e9,
           "targets_flagged": [c[0] for c in counts], "targets_left_to_k_graph": [c[1] + c[2] for c in counts],
           "targets_in_large_tier": [int(s.n_big_tier) for s in sizes[:n_sets]],
           "paths_per_step": float(np.mean([int(s.n_paths) for s in sizes])),
           "kernel_ms": {"walk": float(tm[0]), "k_pack": float(tm[4]), "k_seed": float(tm[3]), "k_dfs": float(tm[5]),
                         "graph": float(tm[1]), "deliver_kernels": float(tm[6]), "d2h_copy": float(tm[7]),
                         "note": "one batch at a time, every kernel alone (KM_RUN_SERIAL); the large tier runs on the host's "
                                 "request after these kernels and is part of ms_per_step, not of kernel_ms"},
           "walk_stage_frac_of_hbm_peak": probes * BYTES_PER_PROBE / (float(tm[0]) * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "oracle_check": check, "setup_s": {"generate": t_gen}}
    for bq in batches:
        bq.close()
    for st_ in streams:
        kmlib.stream_destroy(st_)
    db.close()
    return out


def dump_case(case, path):
    """The workload as km_amd/kmclient (tools/kmclient.cpp) reads it."""
    os.makedirs(path, exist_ok=True)
    np.save(os.path.join(path, "keys.npy"), np.ascontiguousarray(case["keys"], dtype=np.uint64))
    np.save(os.path.join(path, "counts.npy"), np.ascontiguousarray(case["counts"], dtype=np.uint32))
    np.save(os.path.join(path, "targets.npy"), np.ascontiguousarray(case["targets"], dtype=np.uint8))


def c_abi_end_to_end(case, args, n_fl):
    """`end_to_end_c_abi`: host strings in -> TSV text out with no interpreter between the calls — the C++
    consumer of include/kmgpu.h (tools/kmclient.cpp, mode e2e) run as a child process on the same workload:
    km_batch_set_targets of fresh strings every step, walk + path search, lean delivery, km_report_rows on a
    second thread while the next batches run."""
    import tempfile
    client = os.path.join(ROOT, "km_amd", "kmclient")
    if not os.path.exists(client):
        return None
    with tempfile.TemporaryDirectory() as td:
        dump_case(case, td)
        p = subprocess.run([client, "e2e", td, str(args.steps), str(max(args.warmup, n_fl)), str(n_fl), "3"],
                           capture_output=True, text=True, timeout=900)
    if p.returncode != 0:
        return {"error": p.stderr[-500:]}
    res = json.loads(p.stdout.strip().splitlines()[-1])
    res["value"] = res["targets_per_s_best"]
    res["unit"] = "targets/s"
    return res


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200,
                    help="steps of the timed region (x --repeats; 200 x 0.15 ms = 30 ms per repeat: round 3's 40 steps were "
                         "3.5 ms, shorter than the run-to-run effects they were compared against)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--targets", type=int, default=10000, help="targets per GPU per step")
    ap.add_argument("--length", type=int, default=500)
    ap.add_argument("--keys", type=int, default=100_000_000)
    ap.add_argument("--cpu-sample", type=int, default=200)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--check", type=int, default=200, help="targets checked against the C oracle (0 = off)")
    ap.add_argument("--e2e", type=int, default=10000,
                    help="targets for the end-to-end (strings in -> TSV rows out) measurement")
    ap.add_argument("--hipgraph", action="store_true",
                    help="replay each step as one captured hipGraph (measured: no gain, GPU-bound)")
    ap.add_argument("--walk-only", action="store_true", help="time the walk stage only (ablations)")
    ap.add_argument("--only-step", action="store_true",
                    help="skip the side measurements (end-to-end host path, probe kernels, one-target "
                         "latency, ingestion): every launch in a profile of this run is a 10 000-target launch")
    ap.add_argument("--no-ingest", dest="ingest", action="store_false",
                    help="skip the .jf ingestion measurement (writes a 1.2 GB file to the temp dir)")
    ap.add_argument("--cache", default="", help="directory to keep the generated workload in")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo: rehearsal of the "
                    "multi-rank flow, records broadcast through host memory)")
    ap.add_argument("--one-gpu", action="store_true", help="rehearsal: every rank uses device 0")
    ap.add_argument("--py-loop", action="store_true",
                    help="run the pipelined steps from a Python loop (wait_result + run per step) instead of "
                         "km_batch_pump")
    ap.add_argument("--serial", action="store_true",
                    help="KM_RUN_SERIAL on every run: each kernel alone on the GPU, one stream (the command "
                         "behind profiles/*kernel_stats.csv: rocprofv3 then times the kernels as the roofline "
                         "section does)")
    ap.add_argument("--dump-case", default="",
                    help="write the workload (keys.npy, counts.npy, targets.npy) for km_amd/kmclient into this directory and exit")
    ap.add_argument("--timeline", action="store_true",
                    help="run the warm-up and the pipelined lean timed region only, then exit (the program behind the "
                         "kernel + memory-copy timeline of tools/collect_evidence.sh)")
    ap.add_argument("--no-hard", dest="hard", action="store_false",
                    help="skip the second workload (config4_hard: several variants per target, branch noise, large-tier walks)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="how often the timed region (exactly --steps steps) is repeated; value = the median")
    ap.add_argument("--one-at-a-time", action="store_true",
                    help="run the batches strictly one after the other (run, wait for the delivery, next batch), still rotating "
                         "over the --inflight distinct target sets: with --serial every launch of a profile is then a kernel "
                         "alone on the GPU over a target set that was not the previous launch's (nothing replayed out of the "
                         "256 MiB Infinity Cache) — the command behind profiles/*kernel_stats.csv and the PMC passes")
    ap.add_argument("--profile-hard", action="store_true",
                    help="only the second workload (config4_hard), one batch at a time, every kernel alone, rotating over its "
                         "two target sets; prints its object and exits (the program behind profiles/*hard*)")
    ap.add_argument("--inflight", type=int, default=4,
                    help="batch workspaces in flight on separate HIP streams (software pipelining)")
    args = ap.parse_args()

    # ---- N > 1 without a torchrun environment: start the ranks ourselves, before any GPU call
    if args.gpus > 1 and "RANK" not in os.environ:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))

    T = args.targets
    n_fl = max(1, args.inflight)
    both_names = "walk" if args.walk_only else "walk+graph"

    # ---- rank 0: workload + CPU baseline, BEFORE this process touches the GPU ----------------
    t_gen = time.perf_counter()
    case = load_case(args, T * n_fl) if rank == 0 else None
    t_gen = time.perf_counter() - t_gen
    if args.dump_case and rank == 0:
        dump_case(case, args.dump_case)
        return
    cpu = None
    # N = 1: no PyTorch in this process.  libkmgpu.so is then served by the ROCm installation's HIP runtime, as
    # it is for any C consumer of include/kmgpu.h (km_amd/kmclient), not by the older one bundled in the torch
    # wheel — measured: 0.25 against 0.29 ms per pipelined step.  N > 1 needs torch.distributed (RCCL) and with
    # it torch's runtime; KM_BENCH_TORCH=1 forces that path at N = 1 too.
    use_torch = world > 1 or os.environ.get("KM_BENCH_TORCH") == "1"
    torch = dist = None
    if use_torch:
        import torch                      # before libkmgpu.so: one HIP runtime per process (torch's);
        import torch.distributed as dist  # importing torch does not touch the GPU
    else:
        os.environ.setdefault("KM_HIP_RUNTIME", "system")
    if rank == 0:
        import __graft_entry__ as ge
        ge.build()
        if not args.no_cpu and world == 1:
            cpu = cpu_baseline(case, min(args.cpu_sample, T))

    from km_amd import lib as kmlib
    from km_amd import synth
    if args.one_gpu:
        local_rank = 0
    if use_torch:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        bdev = dev if args.backend == "nccl" else torch.device("cpu")     # where collectives run
        device_sync = torch.cuda.synchronize
    else:
        n_dev = C.c_int(0)
        if kmlib.load().km_device_count(C.byref(n_dev)) != 0 or n_dev.value < 1:
            raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
        dev = bdev = None

        def device_sync():
            kmlib.device_sync(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        dist.barrier()
    # which ranks and devices take part (so that a driver can confirm N ranks on N devices)
    ranks_seen = {"world_size": world, "backend": args.backend if world > 1 else None, "devices": [local_rank],
                  "hip_runtime": None}
    if world > 1:
        mine = torch.tensor([rank, int(torch.cuda.current_device())], dtype=torch.int64, device=bdev)
        seen = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(seen, mine)
        ranks_seen["devices"] = [int(x[1].item()) for x in sorted(seen, key=lambda v: int(v[0].item()))]
        ranks_seen["world_size"] = int(dist.get_world_size())
    ranks_seen["hip_runtime"] = kmlib.HIP_RUNTIME_BOUND

    if args.profile_hard:
        hard = hard_workload(args, case, local_rank, None, T, n_fl, profile_only=True)
        print(json.dumps({"config4_hard_profile": hard}), flush=True)
        sys.exit(0)

    # ---- table: records to HBM (N > 1: ONE broadcast over RCCL), local build on every GPU --------------
    t_up = time.perf_counter()
    if use_torch:
        from km_amd import dist as kd
        d_keys, d_cnts, n_rec, _k, _canon = kd.broadcast_records(
            case["keys"] if rank == 0 else None, case["counts"] if rank == 0 else None, K, True, bdev)
        if rank == 0:
            bases_all = torch.from_numpy(np.frombuffer(b"ACGT", dtype=np.uint8)[case["targets"]].copy()).to(bdev)
        else:
            bases_all = torch.empty((T * n_fl, args.length), dtype=torch.uint8, device=bdev)
        if world > 1:
            dist.broadcast(bases_all, 0)
        d_keys, d_cnts, bases_all = d_keys.to(dev).contiguous(), d_cnts.to(dev).contiguous(), bases_all.to(dev)
        device_sync()
        t_bcast = time.perf_counter() - t_up
        t_build = time.perf_counter()
        db = kmlib.Database.empty(K, True)
        stream = torch.cuda.current_stream().cuda_stream
        db.upload_from_device(local_rank, d_keys.data_ptr(), d_cnts.data_ptr(), n_rec, stream)
        device_sync()
        t_build = time.perf_counter() - t_build
        del d_keys, d_cnts
        torch.cuda.empty_cache()
    else:
        stream = None
        n_rec = int(len(case["keys"]))
        bases_host = np.frombuffer(b"ACGT", dtype=np.uint8)[case["targets"]].copy()
        db = kmlib.Database.from_records(case["keys"], case["counts"], K)
        t_bcast = 0.0
        t_build = time.perf_counter()
        db.upload(local_rank)                              # records H2D + the device-side table build
        t_build = time.perf_counter() - t_build
    info = db.info
    n_probe = int(min(n_rec, T * (args.length - K + 1)))
    probe_keys = None
    if rank == 0:
        sel = np.random.default_rng(12345).choice(n_rec, size=n_probe, replace=False)
        probe_keys = np.ascontiguousarray(case["keys"][sel])

    # ---- the box's large-copy bandwidth (device-to-device), beside the 8 TB/s spec -------------
    d2d = kmlib.device_copy_GBs(local_rank, 1 << 30, 10) if rank == 0 else None      # read + write

    # ---- workspaces: `inflight` of them, each with its own HIP stream and its OWN target set ---
    # (weak scaling: every rank steps through the same n_fl sets, rotated by its rank)
    offsets = (np.arange(T + 1, dtype=np.uint64) * np.uint64(args.length))
    tstreams = [kmlib.stream_create(local_rank) for _ in range(n_fl)]     # non-blocking HIP streams
    set_ids = [(q + rank) % n_fl for q in range(n_fl)]
    batches = []
    for q in range(n_fl):
        bq = kmlib.Batch(db, ratio=0.05, count=5, max_stack=500, max_break=10, max_node=10000,
                         max_targets=T, max_total_bases=T * args.length)
        if use_torch:
            bq.set_targets_dev(bases_all[set_ids[q] * T:(set_ids[q] + 1) * T].data_ptr(), offsets, stream)
        else:
            bq.set_targets_packed(bases_host[set_ids[q] * T:(set_ids[q] + 1) * T].reshape(-1), offsets)
        batches.append(bq)
    device_sync()
    stages = kmlib.KM_STAGE_WALK | (0 if args.walk_only else kmlib.KM_STAGE_GRAPH) | (kmlib.KM_RUN_SERIAL if args.serial else 0)
    if args.hipgraph:
        stages |= kmlib.KM_RUN_HIPGRAPH
    deliver_full = stages | kmlib.KM_RUN_DELIVER
    # lean delivery: what `km find_mutation` needs to print its TSV (km_amd.finder.BatchFinder.rows):
    # bare-reference targets are delivered as path + min coverage + ref_max_cov, without their counts
    # ... and node counts as 16-bit values + the list of the exact counts >= 65535 (KM_DELIVER_COUNT16)
    deliver = deliver_full | kmlib.KM_DELIVER_LEAN | kmlib.KM_DELIVER_COUNT16

    def pipeline(n_steps, flags, wait):
        """n_steps steps round-robin over the workspaces; before a workspace is reused (and at
        the end) its previous delivery is awaited, i.e. its results are in pinned host memory.
        The loop itself runs inside the library (km_batch_pump) unless --py-loop: with the
        interpreter between a wait and the next launch a launch costs ~200 us of host time
        instead of ~50 (tools/launch_cost.py) and the GPU runs dry."""
        if args.one_at_a_time:
            for i in range(n_steps):
                q = i % n_fl
                batches[q].run(flags, tstreams[q])
                if wait:
                    batches[q].wait_result()
                else:
                    batches[q].sync()
            device_sync()
            return
        if wait and not args.py_loop:
            kmlib.pump(batches, tstreams, n_steps, flags)
            device_sync()
            return
        for i in range(n_steps):
            q = i % n_fl
            if wait and i >= n_fl:
                batches[q].wait_result()
            batches[q].run(flags, tstreams[q])
        if wait:
            for q in range(min(n_fl, n_steps)):
                batches[q].wait_result()
        device_sync()

    # ---- warm-up ---------------------------------------------------------------------------
    # (the table slots read are counted on request only — KM_RUN_COUNT_FETCHES, 2 % of a step: the warm-up asks)
    pipeline(max(n_fl, args.warmup), deliver | kmlib.KM_RUN_COUNT_FETCHES, True)
    sizes = [bq.wait_result() for bq in batches]
    probes_per_step = float(np.mean([int(s.logical_probes) for s in sizes]))
    seed_probes = float(np.mean([int(s.seed_probes) for s in sizes]))
    fetches_per_step = float(np.mean([int(s.table_fetches) for s in sizes]))
    pipeline(n_fl, deliver, True)                   # the kernels of the timed region, once, untimed

    # ---- timed region: exactly K steps, results delivered to pinned host memory ---------------
    def timed(flags, wait):
        if world > 1:
            dist.barrier()
        device_sync()
        t0 = time.perf_counter()
        pipeline(args.steps, flags, wait)
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=bdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    def delivered_bytes():
        vs = [bq.result() for bq in batches]
        # (with 16-bit counts the view also carries a 32-bit copy made on the host: not delivered)
        return vs, float(np.mean([sum(v[x].nbytes for x in v if isinstance(v[x], np.ndarray)
                                      and not (x == "node_count" and "node_count16" in v)) for v in vs]))

    if args.timeline:
        dts = [timed(deliver, True) for _ in range(max(1, args.repeats))]
        if rank == 0:
            print(json.dumps({"timeline_only": True, "ms_per_step": float(np.median(dts)) / args.steps * 1e3,
                              "steps": args.steps, "repeats": len(dts), "batches_in_flight": n_fl}), flush=True)
        for bq in batches:
            bq.close()
        for st_ in tstreams:
            kmlib.stream_destroy(st_)
        db.close()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(0)
    # full delivery first (every node count crosses PCIe), then the lean one that `value` reports
    pipeline(n_fl, deliver_full, True)
    dt_full = timed(deliver_full, True)
    views_full, out_bytes_full = delivered_bytes()
    check_full = None
    if rank == 0 and args.check > 0 and not args.walk_only:
        check_full = oracle_check(case, views_full, set_ids, T, max(n_fl, args.check // 4))
    del views_full
    pipeline(n_fl, deliver, True)
    dts = [timed(deliver, True) for _ in range(max(1, args.repeats))]      # each: exactly K steps
    dt = float(np.median(dts))
    views, out_bytes = delivered_bytes()
    # the same timed region driven from the interpreter (wait_result + run per step), for the record
    dt_py = None
    if not args.py_loop:
        args.py_loop = True
        pipeline(n_fl, deliver, True)
        dt_py = timed(deliver, True)
        args.py_loop = False
    check = None
    if rank == 0 and args.check > 0 and not args.walk_only:
        check = oracle_check(case, views, set_ids, T, args.check)
        log("oracle check:", check)
    del views
    # kernel-only rate (results stay in HBM): what round 1 reported as `value`
    dt_kernel = float(np.median([timed(stages, False) for _ in range(max(1, args.repeats))]))

    # ---- one step at a time (no pipelining), with and without delivery -------------------------
    batch = batches[0]
    st0 = tstreams[0]
    device_sync()
    t1 = time.perf_counter()
    n_serial = min(args.steps, 40)
    for i in range(n_serial):
        batch.run(deliver, st0)
        batch.wait_result()
    serial_ms = (time.perf_counter() - t1) / n_serial * 1e3
    # ---- per-kernel durations (HIP events on the launch stream), averaged over K launches
    # one batch at a time, rotating over the n_fl distinct target sets (each has its own table lines:
    # a set replayed back to back would find part of them in the 256 MiB Infinity Cache)
    def event_times(flags):
        tm = []
        for i in range(min(args.steps, 40)):
            bq, sq = batches[i % n_fl], tstreams[i % n_fl]
            bq.run((flags & ~kmlib.KM_RUN_HIPGRAPH) | kmlib.KM_RUN_TIMED, sq)
            bq.wait_result()
            tm.append(bq.timings())
        return [float(x) for x in np.mean(np.array(tm), axis=0)]

    # every kernel alone on the GPU (KM_RUN_SERIAL: the pass over the unflagged targets follows k_dfs
    # instead of running beside it) — a kernel's own duration, what rocprofv3 reports for
    # `bench.py --serial`; and the same step as the pipeline launches it (k_graph_pure beside k_dfs)
    walk_avg, graph_avg, _tot, seed_avg, pack_avg, dfs_avg, outk_avg, d2h_avg = event_times(deliver | kmlib.KM_RUN_SERIAL)
    # the walk stage as it is launched in production: ONE pair of events around its three kernels (the per-kernel
    # numbers above put two more event records between them, each a barrier packet the stage does not have otherwise)
    walk_stage_avg = event_times(deliver | kmlib.KM_RUN_SERIAL | kmlib.KM_RUN_TIMED_STAGES)[0]
    ovl = event_times(deliver)
    outk_full, d2h_full = event_times(deliver_full)[6:8]

    # ---- result fetch through the copying API (numpy arrays, node k-mers rebuilt), for scale
    t_f = time.perf_counter()
    res = batch.fetch()
    fetch_s = time.perf_counter() - t_f
    del res

    # ---- BASELINE config 4 proper: 10 000 targets sharded over the N GPUs ----------------------
    strong = None
    if world > 1:
        Ts = T // world
        offs_s = (np.arange(Ts + 1, dtype=np.uint64) * np.uint64(args.length))
        for q in range(n_fl):
            lo = set_ids[q] * T + rank * Ts
            batches[q].set_targets_dev(bases_all[lo:lo + Ts].data_ptr(), offs_s, stream)
        device_sync()
        pipeline(n_fl, deliver, True)
        dt_strong = timed(deliver, True)
        strong = {"targets_total": Ts * world, "targets_per_gpu": Ts, "ms_per_step": dt_strong / args.steps * 1e3,
                  "value": Ts * world / (dt_strong / args.steps), "unit": "targets/s", "scaling": "strong"}
        for q in range(n_fl):
            batches[q].set_targets_dev(bases_all[set_ids[q] * T:(set_ids[q] + 1) * T].data_ptr(), offsets, stream)

    # ---- what one GPU does with 1/2, 1/4, 1/8 of config 4's targets per step: the ceiling of config4_strong
    # (the same 10 000 targets sharded over N GPUs) — measured here because 8-GPU nodes are the driver's to use
    strong_ceiling = None
    if rank == 0 and world == 1 and not args.only_step and not use_torch:
        strong_ceiling = {"targets_total": T, "ms_per_step_all_targets_one_gpu": dt / args.steps * 1e3, "per_gpu_share": []}
        for parts in (2, 4, 8):
            Ts = T // parts
            offs_s = (np.arange(Ts + 1, dtype=np.uint64) * np.uint64(args.length))
            for q in range(n_fl):
                batches[q].set_targets_packed(bases_host[set_ids[q] * T:set_ids[q] * T + Ts].reshape(-1), offs_s)
            pipeline(2 * n_fl, deliver, True)
            dts_s = [timed(deliver, True) for _ in range(3)]
            ms_s = float(np.median(dts_s)) / args.steps * 1e3
            # one batch alone (what a rank that has nothing else in flight would see)
            lat = []
            for i in range(8):
                t_l = time.perf_counter()
                batches[i % n_fl].run(deliver, tstreams[i % n_fl])
                batches[i % n_fl].wait_result()
                lat.append((time.perf_counter() - t_l) * 1e3)
            # ... and with twice as many (smaller) batches in flight: a rank whose share is small is bound by the latency
            # of a batch's chain of kernels, not by the GPU — more batches in flight is what it has to answer with
            extra_b, extra_s = [], []
            for q in range(n_fl):
                bq = kmlib.Batch(db, ratio=0.05, count=5, max_stack=500, max_break=10, max_node=10000,
                                 max_targets=Ts, max_total_bases=Ts * args.length)
                lo = set_ids[q] * T + Ts * (1 if parts > 1 else 0)
                bq.set_targets_packed(bases_host[lo:lo + Ts].reshape(-1), offs_s)
                extra_b.append(bq)
                extra_s.append(kmlib.stream_create(local_rank))
            both_b, both_s = batches + extra_b, tstreams + extra_s
            kmlib.pump(both_b, both_s, 4 * n_fl, deliver)
            device_sync()
            dts_8 = []
            for _ in range(3):
                device_sync()
                t_8 = time.perf_counter()
                kmlib.pump(both_b, both_s, 2 * args.steps, deliver)
                device_sync()
                dts_8.append((time.perf_counter() - t_8) / (2 * args.steps) * 1e3)
            ms_8 = float(np.median(dts_8))
            for bq in extra_b:
                bq.close()
            for sq in extra_s:
                kmlib.stream_destroy(sq)
            strong_ceiling["per_gpu_share"].append(
                {"n_gpus": parts, "targets_per_gpu": Ts, "ms_per_step_pipelined": ms_s, "ms_one_batch_alone": float(np.median(lat[2:])),
                 "implied_value_at_n_gpus": T / (ms_s * 1e-3),
                 "implied_speedup_vs_1_gpu": (dt / args.steps * 1e3) / ms_s,
                 "with_%d_batches_in_flight" % (2 * n_fl): {"ms_per_step_pipelined": ms_8, "implied_value_at_n_gpus": T / (ms_8 * 1e-3),
                                                             "implied_speedup_vs_1_gpu": (dt / args.steps * 1e3) / ms_8}})
        for q in range(n_fl):
            batches[q].set_targets_packed(bases_host[set_ids[q] * T:(set_ids[q] + 1) * T].reshape(-1), offsets)
        pipeline(n_fl, deliver, True)
        strong_ceiling["note"] = ("one GPU, 4 batches in flight of T/N targets each, results delivered: what every rank of config4_strong "
                                  "would do per step if the N GPUs scaled perfectly; implied_speedup is the CEILING of the N-GPU "
                                  "curve for BASELINE config 4 proper (10 000 targets per step in all), set by the per-step latency "
                                  "floor of the kernels, not by any collective")

    # ---- a hard workload beside the headline one ------------------------------------------------------
    hard = None
    if rank == 0 and world == 1 and args.hard and not args.only_step:
        hard = hard_workload(args, case, local_rank, stream, T, n_fl)
        log("config4_hard:", {k_: hard[k_] for k_ in ("value", "ms_per_step", "targets_flagged", "targets_left_to_k_graph",
                                                       "targets_in_large_tier", "oracle_check")})

    # ---- end to end through the drop-in host path: strings -> GPU -> TSV rows -----------------
    e2e = None
    if rank == 0 and args.e2e > 0 and not args.only_step:
        from km_amd import kmer as km, report
        from km_amd.finder import BatchFinder
        from km_amd.jellyfish import Jellyfish
        n_e = min(args.e2e, T)
        tg = [(case["names"][i], km.decode(case["targets"][i])) for i in range(n_e)]
        jf = Jellyfish("synthetic.jf", cutoff=0.05, n_cutoff=5, device=local_rank, db=db)
        finder = BatchFinder(jf)
        import io
        finder.write_rows(tg[:64], io.StringIO())                # workspace allocation, first launch
        finder.write_rows(tg, io.StringIO())
        sink = io.StringIO()
        t_e = time.perf_counter()
        finder.write_rows(tg, sink)                              # what the CLI does per batch (km_report_rows)
        e2e = {"targets": n_e, "seconds": time.perf_counter() - t_e}
        e2e["rows"] = sink.getvalue().count("\n")
        e2e["targets_per_s"] = n_e / e2e["seconds"]

    # ---- the same end to end through the C-ABI alone (a C++ consumer, no Python between the calls)
    e2e_c = None
    if rank == 0 and world == 1 and args.e2e > 0 and not args.only_step:
        e2e_c = c_abi_end_to_end(case, args, n_fl)
        log("end_to_end_c_abi:", e2e_c)

    # ---- `.jf` ingestion -------------------------------------------------------
    ingest = None
    if rank == 0 and not args.only_step and args.ingest:
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "bench.jf")
            synth.write_jf(path, case["keys"], case["counts"], K)
            size = os.path.getsize(path)
            kmlib.Database.load(path, local_rank).close()          # page cache + first-touch warm-up
            t_parse = t_host = float("inf")
            for _ in range(3):
                t_i = time.perf_counter()
                d1 = kmlib.Database.open(path)
                t_p = time.perf_counter() - t_i
                d1.upload(local_rank)
                t_h = time.perf_counter() - t_i
                d1.close()
                if t_h < t_host:
                    t_parse, t_host = t_p, t_h
            t_direct = float("inf")
            for _ in range(3):                                      # best of 3: page-cache state varies
                t_i = time.perf_counter()
                d2 = kmlib.Database.load(path, local_rank)
                t_direct = min(t_direct, time.perf_counter() - t_i)
                d2.close()
        ingest = {"file_bytes": size, "records": int(len(case["keys"])),
                  "host_reader_parse_s": t_parse, "host_reader_plus_upload_s": t_host,
                  "direct_file_to_table_s": t_direct, "direct_GBs": size / t_direct / 1e9}

    # ---- BASELINE config 5: the 9-target catalog x N synthetic per-sample .jf (seed = sample index),
    #      sample-sharded; every sample = kmjf_load + one batch + native rows (km_amd.dist.sample_matrix)
    cfg5 = None
    cat_dir = os.path.join(ROOT, "tests", "data", "catalog", "GRCh38")
    if rank == 0 and not args.only_step and args.ingest and os.path.isdir(cat_dir) and world == 1:
        import tempfile
        from km_amd.cli import read_target
        files = [os.path.join(cat_dir, f) for f in sorted(os.listdir(cat_dir))]
        seqs = [read_target(f) for f in files]
        n_samples, n_sk = 6, 4_000_000
        with tempfile.TemporaryDirectory() as td:
            paths = []
            for si in range(n_samples):
                kk, cc = synth.make_sample(seqs, si, K, n_sk)
                pth = os.path.join(td, "sample_%02d.jf" % si)
                synth.write_jf(pth, kk, cc, K)
                paths.append(pth)
            from km_amd import dist as kd
            kd.sample_matrix(paths[:1], files, os.path.join(td, "warm"))
            t_s = time.perf_counter()
            outs = kd.sample_matrix(paths, files, os.path.join(td, "out"))
            t_s = time.perf_counter() - t_s
            n_rows = sum(sum(1 for l in open(o) if not l.startswith("#")) for o in outs)
        cfg5 = {"samples": n_samples, "keys_per_sample": n_sk, "targets": len(files), "seconds": t_s,
                "samples_per_s": n_samples / t_s, "tsv_lines": n_rows,
                "includes": ".jf -> HBM table per sample, one batch over the catalog, native rows, files written"}
        # the same at the size of a real sample (example/run_leucegene.sh:24): 100 M distinct k-mers per sample
        n_big, n_bk = 2, 100_000_000
        with tempfile.TemporaryDirectory() as td:
            paths = []
            t_g = time.perf_counter()
            for si in range(n_big):
                kk, cc = synth.make_sample(seqs, 100 + si, K, n_bk)
                pth = os.path.join(td, "big_%02d.jf" % si)
                synth.write_jf(pth, kk, cc, K)
                paths.append(pth)
                del kk, cc
            t_g = time.perf_counter() - t_g
            kd.sample_matrix(paths[:1], files, os.path.join(td, "warm"))
            t_s = time.perf_counter()
            outs = kd.sample_matrix(paths, files, os.path.join(td, "out"))
            t_s = time.perf_counter() - t_s
            n_rows = sum(sum(1 for l in open(o) if not l.startswith("#")) for o in outs)
        cfg5["real_size"] = {"samples": n_big, "keys_per_sample": n_bk, "seconds": t_s, "samples_per_s": n_big / t_s,
                             "seconds_per_sample": t_s / n_big, "tsv_lines": n_rows, "generate_and_write_s": t_g,
                             "note": "1.2 GB .jf per sample from page cache -> HBM -> 10.3 GB table, the catalog in one batch, rows"}

    # ---- probe kernels alone (rows A2 / A3) ----------------------------------------------------
    probe = None
    if rank == 0 and probe_keys is not None and not args.only_step:
        q_ms, c_ms, n_zero = kmlib.probe_bench(db, probe_keys, reps=10, ratio=0.05, n_cutoff=5)
        probe = {"n_kmers": n_probe,
                 "query": {"ms": q_ms, "G_probes_per_s": n_probe / q_ms / 1e6,
                           "achieved_GBs": n_probe * 12 / q_ms / 1e6, "frac": n_probe * 12 / q_ms / 1e6 / HBM_PEAK_GBS},
                 "get_child": {"ms": c_ms, "G_probes_per_s": 4 * n_probe / c_ms / 1e6,
                               "achieved_GBs": n_probe * 48 / c_ms / 1e6, "frac": n_probe * 48 / c_ms / 1e6 / HBM_PEAK_GBS}}
        assert n_zero == 0                                # every stored k-mer is found

    # ---- BASELINE config 2: latency of ONE target (FLT3-ITD, 75-nt ITD, walk depth 65) --------
    single = None
    fix_fa = os.path.join(ROOT, "tests", "data", "catalog", "GRCh38", "FLT3-ITD_exons_13-15.fa")
    fix_db = os.path.join(ROOT, "tests", "data", "jf", "03H116_ITD.jf")
    if rank == 0 and os.path.exists(fix_fa) and os.path.exists(fix_db) and not args.only_step:
        from km_amd.cli import read_target
        from km_amd.finder import BatchFinder as _BF
        from km_amd.jellyfish import Jellyfish as _JF
        jf1 = _JF(fix_db, cutoff=0.05, n_cutoff=5, device=local_rank)
        f1 = _BF(jf1)
        seq1 = read_target(fix_fa)
        f1.run_raw([seq1])
        lat = []
        for _ in range(20):
            t_s = time.perf_counter()
            r1 = f1.run_raw([seq1])
            lat.append((time.perf_counter() - t_s) * 1e3)
        single = {"target": "FLT3-ITD_exons_13-15 x 03H116_ITD.jf", "median_ms": float(np.median(lat)),
                  "includes": "H2D of the target, kernels, delivery, D2H of nodes and paths",
                  "nodes": int(r1["node_off"][1]), "paths": int(r1["path_off"][1]),
                  "logical_probes": int(r1["probes"][0])}

    rc = 0
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * T / (dt / args.steps)
        ms_kernel = dt_kernel / args.steps * 1e3
        gb = lambda nbytes, ms: nbytes / (ms * 1e-3) / 1e9 if ms > 0 else None
        alg_walk = probes_per_step * BYTES_PER_PROBE
        alg_seed = seed_probes * BYTES_PER_PROBE
        alg_dfs = (probes_per_step - seed_probes) * BYTES_PER_PROBE
        walk_achieved = gb(alg_walk, walk_stage_avg)
        # HBM traffic of the walk stage: PMC counters need their own rocprofv3 passes (the guide: separate
        # --pmc runs), so this number is NOT measured by this process — it is read from the committed
        # summary of tools/collect_evidence.sh, and the line says so
        traffic = traffic_source = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                traffic = pj.get("walk_stage_hbm_bytes_per_step")
                traffic_source = {"file": "profiles/pmc_traffic.json", "measured_in_this_run": False,
                                  "collected_by": pj.get("collected_by", "tools/collect_evidence.sh (rocprofv3 --pmc passes over "
                                                         "bench.py --only-step --inflight 1 --serial)"),
                                  "commit": pj.get("commit"), "round": pj.get("round")}
            except Exception:
                traffic = traffic_source = None
        out = {
            "metric": "find_mutation_targets_per_sec",
            "value": value,
            "unit": "targets/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "timed_region": {"repeats": len(dts), "steps_each": args.steps, "value_is": "median",
                             "ms_per_step_median": ms_per_step, "ms_per_step_min": min(dts) / args.steps * 1e3,
                             "ms_per_step_max": max(dts) / args.steps * 1e3},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "synthetic_%dx%dnt_targets_per_gpu_%dM_kmer_table_k31"
                                   % (T, args.length, round(n_rec / 1e6)),
                       "targets_per_gpu": T, "target_len": args.length, "k": K,
                       "table_keys": n_rec, "table_bytes": int(info.table_bytes),
                       "table_bytes_per_kmer": float(info.table_bytes) / max(1, n_rec),
                       "table_max_probe": int(info.max_probe),
                       "params": "-c 5 -p 0.05 -s 500 -b 10 -n 10000",
                       "stages": both_names + " + result delivery (device compaction, D2H to pinned host memory)",
                       "distinct_target_sets": n_fl,
                       "value_is": ("BASELINE config 4 on one GPU (10 000 targets x 500 nt, 100 M-key table)" if world == 1 else
                                    "weak scaling of config 4's shape: %d targets per GPU per step; config 4 PROPER (10 000 targets "
                                    "sharded over the %d GPUs) is the side key config4_strong" % (T, world)),
                       "parallelism": "target-sharded x%d, table replicated (records broadcast once over RCCL)" % world},
            "value_includes": "k_pack, k_seed, k_dfs, k_graph_pure, k_graph, k_out_scan, k_out_pack and the D2H of the "
                              "lean delivery into pinned host memory",
            "gprobes_per_s": world * probes_per_step / (dt / args.steps) / 1e9,
            "kernel_only": {"value": world * T / (dt_kernel / args.steps), "ms_per_step": ms_kernel,
                            "note": "same steps without result delivery (results left in HBM)"},
            "delivered_bytes_per_step": out_bytes,
            "d2h_GBs_inside_pipeline": gb(out_bytes, ms_per_step),
            "delivery": "lean (KM_DELIVER_LEAN | KM_DELIVER_COUNT16): statuses, probes, paths, min coverages, "
                        "walk-discovered k-mers and the counts of every target that has them or more than one path "
                        "(16-bit values + the exact counts >= 65535 in a list); a bare-reference target "
                        "(one Reference row in the TSV) is delivered as path + min coverage + max count",
            "full_delivery": {"value": world * T / (dt_full / args.steps), "ms_per_step": dt_full / args.steps * 1e3,
                              "delivered_bytes_per_step": out_bytes_full,
                              "deliver_kernels_ms": outk_full, "d2h_copy_ms": d2h_full, "oracle_check": check_full,
                              "note": "same pipeline with the counts of EVERY node crossing PCIe"},
            "logical_probes_per_step": probes_per_step,
            "table_fetches_per_step": fetches_per_step,
            "targets_in_large_tier": int(sizes[0].n_big_tier), "targets_flagged": int(sizes[0].n_flagged),
            "kernel_ms": {"walk": walk_avg, "walk_stage_events_only": walk_stage_avg, "k_pack": pack_avg, "k_seed": seed_avg, "k_dfs": dfs_avg,
                          "graph": graph_avg, "deliver_kernels": outk_avg, "d2h_copy": d2h_avg,
                          "note": "one batch at a time, every kernel alone on the GPU (KM_RUN_SERIAL), HIP events "
                                  "on the launch stream; graph = k_graph_pure + k_graph",
                          "as_pipelined": {"walk": ovl[0], "k_seed": ovl[3], "k_dfs": ovl[5], "graph": ovl[1],
                                           "note": "the same with k_graph_pure beside k_dfs on the side stream, "
                                                   "as every other number of this line runs it"}},
            "host_loop": ("python (wait_result + run per step)" if args.py_loop else
                          "km_batch_pump: the round-robin loop over the batches in flight runs inside the library"),
            "ms_per_step_python_loop": (dt_py / args.steps * 1e3) if dt_py else None,
            "batches_in_flight": n_fl,
            "hipgraph_replay": bool(args.hipgraph),
            "ms_per_step_unpipelined": serial_ms,
            "result_fetch_ms": outk_avg + d2h_avg,
            "result_fetch_copying_api_ms": fetch_s * 1e3,
            "oracle_check": check,
            "ranks_seen": ranks_seen,
            "config4_strong": strong,
            "config4_strong_ceiling_measured_on_one_gpu": strong_ceiling,
            "config4_hard": hard,
            "config5_samples": cfg5,
            "end_to_end_host_path": e2e,
            "end_to_end_c_abi": e2e_c,
            "single_target_latency": single,
            "probe_kernels": probe,
            "d2d_copy_GBs": d2d,
            "setup_s": {"generate": t_gen, "h2d_broadcast": t_bcast, "table_build": t_build},
            "jf_ingestion": ingest,
            # achieved = logical probes x 12 B / walk-stage time
            "roofline": {"bound": "hbm", "kernel": "walk stage (k_pack + k_seed + k_dfs)",
                         "achieved": walk_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": walk_achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": alg_walk,
                         "logical_probes_per_launch": probes_per_step,
                         "avg_launch_ms": walk_stage_avg,
                         "avg_launch_ms_with_per_kernel_events": walk_avg,
                         "timed_with": "HIP events on the launch stream before k_pack and after k_dfs (KM_RUN_TIMED | "
                                       "KM_RUN_TIMED_STAGES), one batch at a time, rotating over the distinct target sets",
                         "measured_copy_peak_GBs": d2d,
                         "per_kernel": {
                             "k_seed": {"achieved": gb(alg_seed, seed_avg), "frac": gb(alg_seed, seed_avg) / HBM_PEAK_GBS,
                                        "avg_launch_ms": seed_avg, "algorithmic_bytes": alg_seed},
                             "k_dfs": {"achieved": gb(alg_dfs, dfs_avg), "frac": gb(alg_dfs, dfs_avg) / HBM_PEAK_GBS,
                                       "avg_launch_ms": dfs_avg, "algorithmic_bytes": alg_dfs},
                             "whole_step_pipelined": {"achieved": gb(alg_walk, ms_per_step),
                                                      "frac": gb(alg_walk, ms_per_step) / HBM_PEAK_GBS}}},
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
        if check is not None and not check["ok"]:
            log("ORACLE CHECK FAILED:", check)
            rc = 3
        if hard is not None and not hard["oracle_check"]["ok"]:
            log("ORACLE CHECK FAILED (config4_hard):", hard["oracle_check"])
            rc = 3
    # release the library's device / pinned memory while the HIP runtime is still up
    for bq in batches:
        bq.close()
    del batches, batch
    for st_ in tstreams:
        kmlib.stream_destroy(st_)
    db.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
