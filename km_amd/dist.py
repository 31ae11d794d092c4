"""Multi-GPU plumbing: one process per GPU over torch.distributed (backend "nccl"
is RCCL on ROCm; "gloo" for CPU rehearsals).

The path shards embarrassingly by target: every (target x sample)
unit is independent.  The only exchange is ONE broadcast of the database —
the compact record arrays (12 B per k-mer), not the sparse 16 B-per-slot table: every
rank then builds its own HBM table with the insert kernel, which is far cheaper
than moving the sparse table over xGMI.  Results return to rank 0 in input order.
No collective sits on the data path.
"""

import os

import numpy as np


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), \
        int(os.environ.get("WORLD_SIZE", "1"))


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


def shard_range(n, rank, world):
    """Contiguous block [lo, hi) of `n` units for `rank` (sizes differ by at most 1)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_records(keys, counts, k, canonical, device, src=0):
    """Rank `src` passes numpy (keys uint64, counts uint32); every rank gets torch
    tensors (int64 / int32 views) on `device` plus (n, k, canonical).  One broadcast
    of the record arrays — the database's only trip over the links."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if rank == src:
        meta = torch.tensor([int(keys.size), int(k), int(bool(canonical))], dtype=torch.int64)
    else:
        meta = torch.zeros(3, dtype=torch.int64)
    meta = meta.to(device)
    if world > 1:
        dist.broadcast(meta, src)
    n, k, canonical = (int(x) for x in meta.tolist())
    if rank == src:
        d_keys = torch.from_numpy(np.ascontiguousarray(keys, dtype=np.uint64).view(np.int64)).to(device)
        d_cnts = torch.from_numpy(np.ascontiguousarray(counts, dtype=np.uint32).view(np.int32)).to(device)
    else:
        d_keys = torch.empty(n, dtype=torch.int64, device=device)
        d_cnts = torch.empty(n, dtype=torch.int32, device=device)
    if world > 1:
        dist.broadcast(d_keys, src)
        dist.broadcast(d_cnts, src)
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


def find_mutation_sharded(targets, db_path, analyse, load_records=None):
    """Target-sharded find_mutation.  `targets`: list of (name, seq) known to every
    rank; rank 0 reads the database (`load_records(db_path) -> keys, counts, k,
    canonical`), broadcasts it once, every rank calls
    `analyse(d_keys, d_cnts, n, k, canonical, my_targets) -> list[list[str]]`
    (rows per target) on its shard; rank 0 returns all rows in target order."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    device = torch.device("cuda", env_world()[1]) if torch.cuda.is_available() else torch.device("cpu")
    if rank == 0:
        keys, counts, k, canonical = load_records(db_path)
    else:
        keys = counts = None
        k = canonical = 0
    d_keys, d_cnts, n, k, canonical = broadcast_records(keys, counts, k, canonical, device)
    lo, hi = shard_range(len(targets), rank, world)
    rows = analyse(d_keys, d_cnts, n, k, canonical, targets[lo:hi])
    return gather_in_order(rows)


def find_mutation_samples(db_paths, run_sample):
    """Sample-sharded runs (BASELINE config 5; the shape of example/run_leucegene.sh:29-35):
    rank r opens and processes db_paths[r], db_paths[r + world], ... entirely on its own GPU
    (`run_sample(path) -> list[str]`); there is no collective on this path, only the final
    gather of the printed rows, in sample order, on rank 0."""
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    mine = [(i, run_sample(p)) for i, p in enumerate(db_paths) if i % world == rank]
    if world == 1:
        return [rows for _, rows in mine]
    bucket = [None] * world if rank == 0 else None
    dist.gather_object(mine, bucket, dst=0)
    if rank != 0:
        return None
    merged = sorted((item for part in bucket for item in part), key=lambda x: x[0])
    return [rows for _, rows in merged]
