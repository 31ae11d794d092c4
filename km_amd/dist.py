"""Multi-GPU find_mutation: one process per GPU over torch.distributed (backend "nccl" is
RCCL on ROCm; "gloo" for CPU rehearsals).

The reference runs the loop of km/tools/find_mutation.py:47-58 serially, and its
Leucegene-scale use is a shell loop over samples (example/run_leucegene.sh:29-35).  Every
(target x sample) unit is independent, so there are two shardings:

* :func:`find_mutation_sharded` — ONE database, targets sharded (:func:`shard_plan`: contiguous
  blocks for a small input, 8 192-target chunks dealt round-robin for a large catalog;
  BASELINE config 4).  The database crosses the links once, as its compact RECORDS
  (12 B per k-mer, one packed buffer), not as the sparse 100 B-per-k-mer table: every rank
  builds its own HBM table from the received records with the device insert kernels
  (``kmjf_upload_from_device``, 0.05 s per 100 M k-mers), which is 8x less link traffic than
  broadcasting the table the north-star sketch names.  Results return to rank 0 in input
  order.  No collective sits on the data path.
* :func:`find_mutation_samples` / :func:`sample_matrix` — MANY databases, sample-sharded
  (BASELINE config 5): rank r loads samples r, r + world, ... with ``kmjf_load`` (file -> HBM
  directly) and runs the whole catalog against each in one batch; no collective at all, only
  the final gather of the printed rows.

Both take the compute as a callable so that the orchestration can be rehearsed on CPU with
gloo (tests/test_dist_cpu.py supplies the oracle there); the DEFAULT callables are the HIP
path (:func:`hip_analyse`, :func:`hip_run_sample`) and are what the CLI and bench.py use.
"""

import os
import sys

import numpy as np

DEFAULT_PARAMS = {"ratio": 0.05, "count": 5, "steps": 500, "branchs": 10, "nodes": 10000}


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), \
        int(os.environ.get("WORLD_SIZE", "1"))


def devices_from_env():
    """KM_DEVICES=0,1,2,3 -> [0, 1, 2, 3] (the GPUs a self-launched run may use), else None."""
    v = os.environ.get("KM_DEVICES", "").strip()
    if not v:
        return None
    return [int(x) for x in v.split(",") if x.strip() != ""]


def local_device():
    """HIP device ordinal of this rank: KM_DEVICES[LOCAL_RANK] if given, else LOCAL_RANK."""
    _rank, local_rank, _world = env_world()
    devs = devices_from_env()
    return devs[local_rank % len(devs)] if devs else local_rank


def init(backend=None, device=None):
    """Initialise the default process group from the torchrun environment."""
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend or "nccl", **kw)
    return rank, local_rank, world


def launch_ranks(n, argv, module="km_amd"):
    """Start `n` ranks of `python -m <module> <argv>` under torch.distributed.run as a CHILD
    process (this process must not have touched the GPU) and return its exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", module] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def shard_range(n, rank, world):
    """Contiguous block [lo, hi) of `n` units for `rank` (sizes differ by at most 1)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


CHUNK = 8192          # targets per GPU batch (km_amd.cli.CHUNK is the same number)


def shard_plan(n, world, chunk=CHUNK):
    """How `n` targets are dealt to `world` ranks: a list of (lo, hi, rank) pieces covering [0, n) in
    order.  A small input (at most one chunk per rank) is cut into `world` contiguous blocks
    (:func:`shard_range`); a large catalog into successive `chunk`-target pieces dealt round-robin,
    so that every rank works through the same number of same-sized batches whatever clusters in the
    input (long walks cluster by locus) and the ranks finish together."""
    if n <= chunk * world:
        return [shard_range(n, r, world) + (r,) for r in range(world) if shard_range(n, r, world)[1] > shard_range(n, r, world)[0]]
    return [(lo, min(n, lo + chunk), (lo // chunk) % world) for lo in range(0, n, chunk)]


def broadcast_records(keys, counts, k, canonical, device, src=0):
    """Rank `src` passes numpy (keys uint64, counts uint32); every rank gets torch tensors
    (int64 / int32 views of ONE packed buffer) on `device` plus (n, k, canonical).  The records
    are the database's only trip over the links: one data broadcast of 12 B per k-mer, preceded
    by a 24-byte size message."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if rank == src:
        meta = torch.tensor([int(keys.size), int(k), int(bool(canonical))], dtype=torch.int64)
    else:
        meta = torch.zeros(3, dtype=torch.int64)
    meta = meta.to(device)
    # (a process group of ONE rank still issues the collectives: the RCCL path then runs on a one-GPU box as well)
    collective = dist.is_initialized()
    if collective:
        dist.broadcast(meta, src)
    n, k, canonical = (int(x) for x in meta.tolist())
    n_words = n + (n + 1) // 2                      # n keys, then n counts two to a 64-bit word
    if rank == src:
        host = np.zeros(n_words, dtype=np.int64)
        host[:n] = np.ascontiguousarray(keys, dtype=np.uint64).view(np.int64)
        host[n:].view(np.int32)[:n] = np.ascontiguousarray(counts, dtype=np.uint32).view(np.int32)
        buf = torch.from_numpy(host).to(device)
    else:
        buf = torch.empty(n_words, dtype=torch.int64, device=device)
    if collective:
        dist.broadcast(buf, src)
    d_keys = buf[:n]
    d_cnts = buf[n:].view(torch.int32)[:n]
    return d_keys, d_cnts, n, k, bool(canonical)


def gather_in_order(local_items, dst=0):
    """Concatenate per-rank lists on `dst` in rank order (== input order for
    contiguous shards).  Other ranks get None."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local_items)
    world = dist.get_world_size()
    bucket = [None] * world if dist.get_rank() == dst else None
    dist.gather_object(list(local_items), bucket, dst=dst)
    if dist.get_rank() != dst:
        return None
    out = []
    for part in bucket:
        out.extend(part)
    return out


# ------------------------------------------------------------------------- the HIP compute
def host_records(path):
    """`load_records` default: the native host reader (kmjf_open)."""
    from . import lib as kmlib
    db = kmlib.Database.open(path)
    keys, counts = db.records()
    info = db.info
    db.close()
    return keys, counts, int(info.k), bool(info.canonical)


def hip_analyse(db_name, params=None, device=None):
    """`analyse` default for :func:`find_mutation_sharded`: build this rank's table from the
    broadcast device records (kmjf_upload_from_device) and run its shard through
    BatchFinder.rows (km_batch_run + lean delivery + km_report_rows)."""
    prm = dict(DEFAULT_PARAMS, **(params or {}))

    def analyse(d_keys, d_cnts, n, k, canonical, my_targets):
        import torch
        from . import lib as kmlib
        from .finder import BatchFinder
        from .jellyfish import Jellyfish
        dev = local_device() if device is None else device
        torch.cuda.set_device(dev)
        if not d_keys.is_cuda:             # a gloo rehearsal broadcasts through host memory
            d_keys, d_cnts = d_keys.to("cuda:%d" % dev), d_cnts.to("cuda:%d" % dev)
        d_keys, d_cnts = d_keys.contiguous(), d_cnts.contiguous()
        db = kmlib.Database.empty(k, canonical)
        stream = torch.cuda.current_stream().cuda_stream
        db.upload_from_device(dev, d_keys.data_ptr(), d_cnts.data_ptr(), n, stream)
        torch.cuda.synchronize()
        jf = Jellyfish(db_name, cutoff=prm["ratio"], n_cutoff=prm["count"], device=dev, db=db)
        finder = BatchFinder(jf, prm["steps"], prm["branchs"], prm["nodes"])
        my_targets = list(my_targets)
        rows = []
        for lo in range(0, len(my_targets), CHUNK):          # one GPU batch per chunk, the workspace is reused
            rows.extend(finder.rows(my_targets[lo:lo + CHUNK]))
        return rows

    return analyse


def hip_run_sample(targets, params=None, device=None):
    """`run_sample` default for :func:`find_mutation_samples`: kmjf_load(path) on this rank's
    GPU, the whole catalog in one batch; returns the rows per target."""
    prm = dict(DEFAULT_PARAMS, **(params or {}))

    def run_sample(path):
        from .finder import BatchFinder
        from .jellyfish import Jellyfish
        dev = local_device() if device is None else device
        jf = Jellyfish(path, cutoff=prm["ratio"], n_cutoff=prm["count"], device=dev)
        rows = BatchFinder(jf, prm["steps"], prm["branchs"], prm["nodes"]).rows(list(targets))
        jf.db.close()
        return rows

    return run_sample


# ------------------------------------------------------------------------- orchestration
def find_mutation_sharded(targets, db_path, analyse=None, load_records=None, params=None, chunk=None):
    """Target-sharded find_mutation.  `targets`: list of (name, seq) known to every rank; rank 0
    reads the database (`load_records(db_path) -> keys, counts, k, canonical`; default: the
    native host reader), broadcasts its records once, every rank calls
    `analyse(d_keys, d_cnts, n, k, canonical, my_targets) -> list (rows per target)` on its shard
    (default: :func:`hip_analyse`); rank 0 returns all of them in target order, other ranks None.
    `chunk`: piece size of :func:`shard_plan` (default 8 192 targets)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    gpu = torch.cuda.is_available() and (not dist.is_initialized() or dist.get_backend() != "gloo")
    if gpu:
        torch.cuda.set_device(local_device())
    device = torch.device("cuda", local_device()) if gpu else torch.device("cpu")
    analyse = analyse or hip_analyse(db_path, params)
    load_records = load_records or host_records
    if rank == 0:
        keys, counts, k, canonical = load_records(db_path)
    else:
        keys = counts = None
        k = canonical = 0
    d_keys, d_cnts, n, k, canonical = broadcast_records(keys, counts, k, canonical, device)
    # every rank takes its pieces of the plan, in order, as ONE call of `analyse` (one table build);
    # whatever it raises travels to rank 0 with the gather instead of leaving the other ranks waiting
    # in it: the error of the earliest target wins there, as in a serial run
    plan = shard_plan(len(targets), world, chunk or CHUNK)
    mine = [t for lo, hi, r in plan if r == rank for t in targets[lo:hi]]
    try:
        payload = ("rows", analyse(d_keys, d_cnts, n, k, canonical, mine))
    except BaseException as exc:             # noqa: BLE001 — re-raised on rank 0
        if isinstance(exc, (KeyboardInterrupt, SystemExit)):
            raise
        payload = ("error", exc)
    if world == 1:
        if payload[0] == "error":
            raise payload[1]
        return payload[1]
    bucket = [None] * world if rank == 0 else None
    # whatever cannot be pickled (an exception holding a handle, say) is replaced by its text HERE, before
    # the collective: exactly one gather_object per rank, never a second one after a failure inside the first
    # (the ranks would then sit in different collectives)
    import pickle
    try:
        pickle.dumps(payload)
    except Exception:                          # noqa: BLE001
        what = payload[1]
        payload = ("error", RuntimeError("%s: %s" % (type(what).__name__, what) if isinstance(what, BaseException)
                                         else "rank %d: result of analyse() cannot be pickled (%s)" % (rank, type(what).__name__)))
    dist.gather_object(payload, bucket, dst=0)
    if rank != 0:
        return None
    out = [None] * len(targets)
    taken = [0] * world
    first_error = None
    for lo, hi, r in plan:
        kind, data = bucket[r]
        if kind == "error":
            if first_error is None:
                first_error = data
            break                                # nothing after the first failing piece is used
        out[lo:hi] = data[taken[r]:taken[r] + (hi - lo)]
        taken[r] += hi - lo
    if first_error is not None:
        raise first_error
    return out


def find_mutation_samples(db_paths, run_sample=None, targets=None, params=None):
    """Sample-sharded runs (BASELINE config 5; the shape of example/run_leucegene.sh:29-35):
    rank r opens and processes db_paths[r], db_paths[r + world], ... entirely on its own GPU
    (`run_sample(path)`; default :func:`hip_run_sample` over `targets`); there is no collective on
    this path, only the final gather of the results, in sample order, on rank 0."""
    if env_world()[2] > 1:
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
    else:                                   # a single process never imports torch (and with it a second HIP runtime)
        dist, rank, world = None, 0, 1
    if run_sample is None:
        if targets is None:
            raise ValueError("find_mutation_samples needs run_sample or targets")
        run_sample = hip_run_sample(targets, params)
    mine = [(i, run_sample(p)) for i, p in enumerate(db_paths) if i % world == rank]
    if world == 1:
        return [rows for _, rows in mine]
    bucket = [None] * world if rank == 0 else None
    dist.gather_object(mine, bucket, dst=0)
    if rank != 0:
        return None
    merged = sorted((item for part in bucket for item in part), key=lambda x: x[0])
    return [rows for _, rows in merged]


def sample_matrix(db_paths, target_files, out_dir, params=None, run_sample=None, read_target=None):
    """The Leucegene-scale driver: catalog x N samples, sample-sharded, written
    as one TSV stream per target — `out_dir/<target>.tsv` holds, sample after sample, exactly what
    `km find_mutation <target.fa> <sample.jf>` prints (the `#key:value` echo, the header, the rows;
    the `#Elapsed time` trailer carries 0) — i.e. the concatenation
    `km find_report -t <target.fa> -f table` consumes (km/tools/find_report.py:290-327), replacing
    the shell loop of example/run_leucegene.sh:29-35.  Returns the list of files (rank 0)."""
    from . import report
    from .cli import read_target as _read_target
    prm = dict(DEFAULT_PARAMS, **(params or {}))
    read_target = read_target or _read_target
    names = [os.path.splitext(os.path.basename(f))[0] for f in target_files]
    targets = [(nm, read_target(f)) for nm, f in zip(names, target_files)]
    per_sample = find_mutation_samples(db_paths, run_sample, targets, prm)
    if per_sample is None:
        return None
    os.makedirs(out_dir, exist_ok=True)
    files = []
    for ti, (nm, tf) in enumerate(zip(names, target_files)):
        path = os.path.join(out_dir, nm + ".tsv")
        with open(path, "w") as fh:
            for dbp, rows in zip(db_paths, per_sample):
                for key, val in (("count", prm["count"]), ("ratio", prm["ratio"]), ("steps", prm["steps"]),
                                 ("branchs", prm["branchs"]), ("nodes", prm["nodes"]), ("graphical", False),
                                 ("verbose", False), ("debug", False), ("target_fn", [tf]),
                                 ("jellyfish_fn", dbp)):
                    fh.write("#%s:%s\n" % (key, val))
                fh.write(report.HEADER + "\n")
                block = rows[ti]
                if isinstance(block, BaseException):
                    raise block
                for row in block:
                    fh.write(row + "\n")
                fh.write("#Elapsed time:0\n")
        files.append(path)
    return files
