#!/usr/bin/env python3
"""Benchmark of the find_mutation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[3]): per GPU 10 000 random
500-nt targets, k=31, against ONE table of 100 M distinct canonical 31-mers
resident in HBM.  A step = one pass of the hot path (walk kernel + path-search
kernel) over the GPU's 10 000 targets; targets and table are in HBM before the
timed region starts.  With N > 1 the table records are broadcast once over RCCL
(torch.distributed) and every rank builds its own table; targets are sharded by
rank, no further collectives (weak scaling: per-GPU work fixed).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_PROBE = 12           # 8-byte key + 4-byte count


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(case, n_sample, k):
    """The oracle (structure-faithful Python restatement of the reference path)
    timed on one host core over the first `n_sample` targets of this workload."""
    from km_amd import kmer as km
    from oracle import km_oracle as ko
    nr = case["n_real"]
    # The first n_real records are the non-pad keys, each tagged with the target it was
    # generated for; random pad 31-mers never touch a walk, and the k-mers of other targets
    # do not either (independent random sequences), so the oracle only needs the keys of
    # the targets it is timed on.
    n_need = min(len(case["targets"]), 4 * n_sample)
    sel = case["key_target"] < n_need
    rec = {"k": k, "canonical": True, "keys": case["keys"][:nr][sel], "counts": case["counts"][:nr][sel]}
    db = ko.KmerDB(None, cutoff=0.05, n_cutoff=5, records=rec)
    t0 = time.perf_counter()
    probes = 0
    rows = 0
    for i in range(n_sample):
        seq = km.decode(case["targets"][i])
        res = ko.analyse_target(seq, case["names"][i], db)
        probes += res["probes"]
        rows += len(res["paths"])
    dt = time.perf_counter() - t0
    # the same work through the plain-C oracle (oracle/km_oracle.c), for scale
    c_rate = None
    try:
        from oracle import c_oracle
        co = c_oracle.COracle(rec["keys"], rec["counts"], k)
        n_c = n_need
        t1 = time.perf_counter()
        for i in range(n_c):
            co.analyse(case["targets"][i])
        c_rate = n_c / (time.perf_counter() - t1)
    except Exception as e:          # the C oracle is optional test infrastructure
        log("C oracle not timed:", e)
    return {"value": n_sample / dt, "unit": "targets/s", "cores": 1, "kind": "port",
            "c_oracle_targets_per_s": c_rate,
            "sample": "first %d of the 10000 targets (walk + path search, oracle/km_oracle.py, "
                      "dict-backed table of the %d keys those targets touch)" % (n_sample, int(sel.sum())),
            "probes_per_s": probes / dt, "seconds": dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--targets", type=int, default=10000, help="targets per GPU")
    ap.add_argument("--length", type=int, default=500)
    ap.add_argument("--keys", type=int, default=100_000_000)
    ap.add_argument("--cpu-sample", type=int, default=300)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--check", action="store_true", help="verify a sample against the oracle")
    ap.add_argument("--e2e", type=int, default=10000,
                    help="targets for the end-to-end (strings in -> TSV rows out) measurement")
    ap.add_argument("--hipgraph", action="store_true",
                    help="replay each step as one captured hipGraph (measured: no gain, GPU-bound)")
    ap.add_argument("--walk-only", action="store_true", help="time the walk stage only (ablations)")
    ap.add_argument("--only-step", action="store_true",
                    help="skip the side measurements (end-to-end host path, probe kernels, one-target "
                         "latency): every launch in a profile of this run is a 10 000-target launch")
    ap.add_argument("--no-ingest", dest="ingest", action="store_false",
                    help="skip the .jf ingestion measurement (writes a 1.2 GB file to the temp dir)")
    ap.add_argument("--cache", default="", help="directory to keep the generated workload in "
                    "(re-used by later invocations with the same sizes)")
    ap.add_argument("--inflight", type=int, default=4,
                    help="batch workspaces in flight on separate HIP streams (software pipelining)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log("warning: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from km_amd import lib as kmlib
    from km_amd import synth

    K = 31
    T = args.targets
    # ---- rank 0 generates the whole job: world*T targets, one table --------------------
    t_gen = time.perf_counter()
    case = None
    if rank == 0:
        tag = "%s/case_%d_%d_%d" % (args.cache, T * world, args.length, args.keys)
        fields = ("keys", "counts", "targets", "key_target")
        if args.cache and os.path.exists(tag + "_keys.npy"):
            case = {f: np.load("%s_%s.npy" % (tag, f)) for f in fields}
            meta = json.load(open(tag + "_meta.json"))
            case.update(k=K, n_real=meta["n_real"], names=meta["names"])
        else:
            case = synth.make_case(n_targets=T * world, length=args.length, k=K, n_keys=args.keys,
                                   seed=synth.HEADLINE_SEED, exact_pad=False)
            if args.cache:
                os.makedirs(args.cache, exist_ok=True)
                for f in fields:
                    np.save("%s_%s.npy" % (tag, f), case[f])
                json.dump({"n_real": int(case["n_real"]), "names": list(case["names"])},
                          open(tag + "_meta.json", "w"))
    t_gen = time.perf_counter() - t_gen

    # ---- table: records to HBM, ONE broadcast over RCCL, local build on every GPU ------
    from km_amd import dist as kd
    t_up = time.perf_counter()
    d_keys, d_cnts, n_rec, _k, _canon = kd.broadcast_records(
        case["keys"] if rank == 0 else None, case["counts"] if rank == 0 else None, K, True, dev)
    if rank == 0:
        bases_all = torch.from_numpy(np.frombuffer(b"ACGT", dtype=np.uint8)[case["targets"]].copy()).to(dev)
    else:
        bases_all = torch.empty((T * world, args.length), dtype=torch.uint8, device=dev)
    if world > 1:
        dist.broadcast(bases_all, 0)
    torch.cuda.synchronize()
    t_bcast = time.perf_counter() - t_up
    t_build = time.perf_counter()
    db = kmlib.Database.empty(K, True)
    stream = torch.cuda.current_stream().cuda_stream
    db.upload_from_device(local_rank, d_keys.data_ptr(), d_cnts.data_ptr(), n_rec, stream)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    info = db.info
    # a resident sample of stored k-mers for the probe-only microbenchmark below
    n_probe = int(min(n_rec, T * (args.length - K + 1)))
    d_probe = d_keys[torch.randperm(n_rec, device=dev)[:n_probe]].contiguous() if rank == 0 else None
    del d_keys, d_cnts
    torch.cuda.empty_cache()

    # ---- this rank's shard of targets, resident in HBM -----------------------------------
    mine = bases_all[rank * T:(rank + 1) * T].contiguous()
    offsets = (np.arange(T + 1, dtype=np.uint64) * np.uint64(args.length))
    # `inflight` independent workspaces, each with its own HIP stream: while one batch is in
    # its latency-bound kernels (k_dfs, k_graph) the next one runs its HBM-bound k_seed
    n_fl = max(1, args.inflight)
    tstreams = [torch.cuda.Stream(device=dev) for _ in range(n_fl)]
    batches = []
    for q in range(n_fl):
        bq = kmlib.Batch(db, ratio=0.05, count=5, max_stack=500, max_break=10, max_node=10000,
                         max_targets=T, max_total_bases=T * args.length)
        bq.set_targets_dev(mine.data_ptr(), offsets, stream)
        batches.append(bq)
    torch.cuda.synchronize()
    batch = batches[0]
    both = kmlib.KM_STAGE_WALK | (0 if args.walk_only else kmlib.KM_STAGE_GRAPH)
    replay = (both | kmlib.KM_RUN_HIPGRAPH) if args.hipgraph else both

    # ---- warm-up ---------------------------------------------------------------------------
    for i in range(max(1, args.warmup)):
        for q in range(n_fl):
            batches[q].run(replay, tstreams[q].cuda_stream)
    for bq in batches:
        bq.sync()
    sizes = batch.sizes()
    probes_per_step = int(sizes.logical_probes)
    fetches_per_step = int(sizes.table_fetches)

    # ---- timed region: exactly K steps (one step = one pass over one batch) ---------------
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        q = i % n_fl
        batches[q].run(replay, tstreams[q].cuda_stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    for bq in batches:
        bq.sync()
    # k_seed as it ran inside the timed region (the last launch of every workspace; with
    # several batches in flight it shares the GPU with the other batches' kernels)
    seed_pipelined_ms = float(np.mean([bq.timings()[3] for bq in batches])) if not args.hipgraph else None

    # ---- latency of one isolated step (no pipelining) --------------------------------------
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(args.steps):
        batch.run(both, stream)
    torch.cuda.synchronize()
    serial_ms = (time.perf_counter() - t1) / args.steps * 1e3
    batch.sync()

    # ---- per-kernel durations (HIP events on the launch stream), averaged over K launches
    walk_ms, graph_ms, seed_ms = [], [], []
    for _ in range(args.steps):
        batch.run(both, stream)
        w, g, _tot, sd = batch.timings()
        walk_ms.append(w)
        graph_ms.append(g)
        seed_ms.append(sd)
    walk_avg = float(np.mean(walk_ms))
    graph_avg = float(np.mean(graph_ms))
    seed_avg = float(np.mean(seed_ms))
    seed_probes = int(batch.sizes().seed_probes)

    # ---- result fetch (D2H + host reorganisation), reported beside the kernel rate ----------
    t_f = time.perf_counter()
    res = batch.fetch()
    fetch_s = time.perf_counter() - t_f

    # ---- end to end through the drop-in host path: strings -> GPU -> TSV rows -----------
    e2e = None
    if rank == 0 and args.e2e > 0 and not args.only_step:
        from km_amd import kmer as km, report
        from km_amd.finder import BatchFinder
        from km_amd.jellyfish import Jellyfish
        n_e = min(args.e2e, T)
        tg = [(case["names"][i], km.decode(case["targets"][i])) for i in range(n_e)]
        jf = Jellyfish("synthetic.jf", cutoff=0.05, n_cutoff=5, device=local_rank, db=db)
        finder = BatchFinder(jf)
        finder.rows(tg[:64])                                     # workspace allocation, first launch
        t_e = time.perf_counter()
        n_rows = sum(len(r) for r in finder.rows(tg))            # native reporting (km_report_rows)
        e2e = {"targets": n_e, "rows": n_rows, "seconds": time.perf_counter() - t_e}
        e2e["targets_per_s"] = n_e / e2e["seconds"]
        t_e = time.perf_counter()
        n_py = 0
        for res_t in finder.analyse(tg[:min(n_e, 1000)]):        # the Python restatement, for scale
            n_py += len(report.target_rows(res_t, jf.filename))
        e2e["python_report_targets_per_s"] = min(n_e, 1000) / (time.perf_counter() - t_e)

    # ---- `.jf` ingestion: the same records as a real binary/sorted file, host
    #      reader + upload against the direct file -> HBM path
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

    # ---- probe kernels alone (rows A2 / A3): Jellyfish.query and get_child for a resident
    #      array of stored k-mers in random order; 12 resp. 48 algorithmic bytes per element
    probe = None
    if rank == 0 and d_probe is not None and not args.only_step:
        d_out = torch.empty(n_probe, dtype=torch.int32, device=dev)
        d_mask = torch.empty(n_probe, dtype=torch.uint8, device=dev)
        d_c4 = torch.empty((n_probe, 4), dtype=torch.int32, device=dev)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        for _ in range(2):
            db.query_dev(d_probe.data_ptr(), n_probe, d_out.data_ptr(), stream)
            db.children_dev(d_probe.data_ptr(), n_probe, 0.05, 5, d_mask.data_ptr(), d_c4.data_ptr(), True, stream)
        reps = 10
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            db.query_dev(d_probe.data_ptr(), n_probe, d_out.data_ptr(), stream)
        e1.record()
        for _ in range(reps):
            db.children_dev(d_probe.data_ptr(), n_probe, 0.05, 5, d_mask.data_ptr(), d_c4.data_ptr(), True, stream)
        e2.record()
        torch.cuda.synchronize()
        q_ms, c_ms = e0.elapsed_time(e1) / reps, e1.elapsed_time(e2) / reps
        probe = {"n_kmers": n_probe,
                 "query": {"ms": q_ms, "G_probes_per_s": n_probe / q_ms / 1e6,
                           "achieved_GBs": n_probe * 12 / q_ms / 1e6, "frac": n_probe * 12 / q_ms / 1e6 / HBM_PEAK_GBS},
                 "get_child": {"ms": c_ms, "G_probes_per_s": 4 * n_probe / c_ms / 1e6,
                               "achieved_GBs": n_probe * 48 / c_ms / 1e6, "frac": n_probe * 48 / c_ms / 1e6 / HBM_PEAK_GBS}}
        assert int((d_out == 0).sum().item()) == 0        # every stored k-mer is found
        del d_out, d_mask, d_c4

    # ---- BASELINE config 2: latency of ONE target (FLT3-ITD, 75-nt ITD, walk depth 65) ----
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
                  "includes": "H2D of the target, 5 kernels, D2H of nodes and paths",
                  "nodes": int(r1["node_off"][1]), "paths": int(r1["path_off"][1]),
                  "logical_probes": int(r1["probes"][0])}

    if args.check and rank == 0:
        from km_amd import kmer as km
        from oracle import km_oracle as ko
        nr = case["n_real"]
        cpu = ko.KmerDB(None, 0.05, 5, records={"k": K, "canonical": True,
                                                "keys": case["keys"][:nr], "counts": case["counts"][:nr]})
        for t in range(0, T, max(1, T // 50)):
            want = ko.analyse_target(km.decode(case["targets"][t]), "t", cpu)
            a, e = int(res["node_off"][t]), int(res["node_off"][t + 1])
            assert [km.unpack(x, K) for x in res["node_kmer"][a:e]] == want["kmers"], t
            assert res["node_count"][a:e].tolist() == want["counts"], t
            pa, pe = int(res["path_off"][t]), int(res["path_off"][t + 1])
            assert [kmlib.expand_path(res, p).tolist() for p in range(pa, pe)] == \
                [list(p) for p in want["paths"]], t
        log("check ok")

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * T / (dt / args.steps)
        # roofline of the dominant kernel (k_seed: ~89 % of all logical probes), timed by its
        # own HIP events on the launch stream
        achieved = seed_probes * BYTES_PER_PROBE / (seed_avg * 1e-3) / 1e9
        walk_achieved = probes_per_step * BYTES_PER_PROBE / (walk_avg * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("k_seed_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "find_mutation_targets_per_sec",
            "value": value,
            "unit": "targets/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "synthetic_%dx%dnt_targets_per_gpu_%dM_kmer_table_k31"
                                   % (T, args.length, round(n_rec / 1e6)),
                       "targets_per_gpu": T, "target_len": args.length, "k": K,
                       "table_keys": n_rec, "table_bytes": int(info.table_bytes),
                       "params": "-c 5 -p 0.05 -s 500 -b 10 -n 10000",
                       "parallelism": "target-sharded x%d, table replicated (1 RCCL broadcast)" % world},
            "gprobes_per_s": world * probes_per_step / (dt / args.steps) / 1e9,
            "logical_probes_per_step": probes_per_step,
            "table_fetches_per_step": fetches_per_step,
            "targets_in_large_tier": int(sizes.n_big_tier), "targets_flagged": int(sizes.n_flagged),
            "kernel_ms": {"walk": walk_avg, "k_seed": seed_avg, "graph": graph_avg},
            "batches_in_flight": n_fl,
            "hipgraph_replay": bool(args.hipgraph),
            "ms_per_step_unpipelined": serial_ms,
            "result_fetch_ms": fetch_s * 1e3,
            "end_to_end_host_path": e2e,
            "single_target_latency": single,
            "probe_kernels": probe,
            "setup_s": {"generate": t_gen, "h2d_broadcast": t_bcast, "table_build": t_build},
            "jf_ingestion": ingest,
            "roofline": {"bound": "hbm", "kernel": "k_seed", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "algorithmic_bytes_per_launch": seed_probes * BYTES_PER_PROBE,
                         "logical_probes_per_launch": seed_probes,
                         "avg_launch_ms": seed_avg,
                         "avg_launch_ms_inside_pipelined_region": seed_pipelined_ms,
                         "walk_stage_achieved_GBs": walk_achieved},
        }
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(case, min(args.cpu_sample, T), K)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
