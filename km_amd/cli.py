"""``km find_mutation`` / ``km min_cov`` drop-in command line.

Same flags, same ``#key:value`` echo, same TSV and ``#Elapsed time`` trailer as
km/tools/find_mutation.py:17-60 and km/argparser/find_mutation.py:4-58, so the
output pipes into ``km find_report`` unchanged.  The GPU is selected with the
environment variable KM_DEVICE (no extra flags: the reference's tests index
output lines); KM_DEVICES=0,1,.. (or a torchrun launch) runs one
rank per GPU: targets sharded for ``find_mutation``, samples sharded for ``samples``.
"""

import argparse
import io
import os
import sys
import time

from . import report
from .finder import BatchFinder, NodeLimitExceeded
from .jellyfish import Jellyfish


def add_find_mutation_args(p):
    p.add_argument("-c", "--count", action="store", nargs="?", default=5, type=int,
                   help="Minimum occurence needed for exploration of alternative (default: -c 5)")
    p.add_argument("-p", "--ratio", action="store", nargs="?", default=0.05, type=float,
                   help="Minimum occurence ratio needed for exploration of alternative (default: -p 0.05)")
    p.add_argument("-s", "--steps", action="store", nargs="?", default=500, type=int,
                   help="Maximum steps to discover a new branch on a target sequence (default: -s 500)")
    p.add_argument("-b", "--branchs", action="store", nargs="?", default=10, type=int,
                   help="Maximum branchs until getback to target sequence (default: -b 10)")
    p.add_argument("-n", "--nodes", action="store", nargs="?", default=10000, type=int,
                   help="Maximum nodes queried from jellyfish database (default: -n 5000)")
    p.add_argument("-g", "--graphical", action="store_true", help="Display coverage graph.")
    p.add_argument("-v", "--verbose", action="store_true", help="Get more information.")
    p.add_argument("-vv", "--debug", action="store_true", help="Get much more information.")
    p.add_argument("target_fn", nargs="*", help="Filename of the target sequence file or directory.")
    p.add_argument("jellyfish_fn", help="Filename of the jellyfish database.")


def list_target_files(args):
    """km/utils/common.py:7-17 (a single directory argument expands in listdir order)."""
    if len(args) == 1 and os.path.isdir(args[0]):
        return [os.path.join(args[0], f) for f in os.listdir(args[0])]
    return list(args)


def read_target(path):
    """All FASTA records of a file, concatenated and upper-cased
    (km/utils/common.py:25-45, km/tools/find_mutation.py:39-43)."""
    chunks, seen_header = [], False
    with open(path) as fh:
        for line in fh:
            if line.startswith(">"):
                seen_header = True
            elif seen_header:
                chunks.append(line.strip())
    return "".join(chunks).upper()


CHUNK = 8192          # targets per GPU batch; rows are flushed after every batch
STREAM_ABOVE = 2_000_000   # catalogs larger than this are printed batch by batch (see main_find_mut)


def _verbose_lines(name_seq, raw, t, k, err, glog=None):
    """The INFO lines of km/utils/MutationFinder.py:101,126,160-161,183-187 and km/utils/Graph.py:198,231, in the
    reference's order (format "VERBOSE: %(message)s", km/tools/find_mutation.py:20-24).  Node indices and the two
    edge counts follow our canonical node order (target k-mers first); the reference's depend on its hash seed
    (its `if last_cur` skips whichever node has index 0: its 'Removed' count is ours or ours - 1).
    `glog`: Batch.graph_log() of the run."""
    from . import kmer as km
    seq = name_seq[1]
    n_ref = int(raw["n_ref"][t])
    x0 = int(raw["extra_off"][t])
    n_nodes = n_ref + int(raw["extra_off"][t + 1]) - x0
    err.write("VERBOSE: Ref. set contains %d kmers.\n" % n_ref)
    if glog is not None:
        for node in glog[2].get(t, ()):                     # where the walk met a k-mer of its own stack
            mer = seq[node:node + k] if node < n_ref else km.unpack(int(raw["extra_kmer"][x0 + node - n_ref]), k)
            err.write("VERBOSE: Broke loop at kmer: %s\n" % mer)
    err.write("VERBOSE: k-mer graph contains %d nodes.\n" % (n_nodes + 2))
    err.write("VERBOSE: BigBang=%d, BigCrunch=%d\n" % (n_nodes, n_nodes + 1))
    err.write("VERBOSE: Start kmer %s %d\n" % (seq[:k], 0))
    err.write("VERBOSE: End   kmer %s %d\n" % (seq[n_ref - 1:n_ref - 1 + k], n_ref - 1))
    if glog is not None:
        err.write("VERBOSE: Removed %d ref edges.\n" % int(glog[0][t]))
        err.write("VERBOSE: %d edges in non-ref edge set.\n" % int(glog[1][t]))
    err.write("VERBOSE: %d path(s) from BigBang to BigCrunch.\n"
              % (int(raw["path_off"][t + 1]) - int(raw["path_off"][t])))


def _print_rows(blocks, out):
    for rows in blocks:
        if isinstance(rows, NodeLimitExceeded):
            out.flush()
            sys.exit(str(rows))
        if isinstance(rows, BaseException):        # what the reference raises while naming a variant
            raise rows
        for row in rows:
            out.write(row + "\n")
    out.flush()


def main_find_mut(args, out=None, err=None):
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    t0 = time.time()
    if args.graphical:
        sys.exit("ERROR: -g/--graphical (matplotlib coverage plots, km/utils/MutationFinder.py:591-611) "
                 "is not supported by km_amd")
    from . import dist as kd
    rank, _local_rank, world = kd.env_world()
    if world > 1:
        import torch
        dev = kd.local_device()
        torch.cuda.set_device(dev)
        kd.init(os.environ.get("KM_DIST_BACKEND", "nccl"), torch.device("cuda", dev))
    if rank == 0:
        for key, val in vars(args).items():
            out.write("#" + str(key) + ":" + str(val) + "\n")
    targets = []
    for f in list_target_files(args.target_fn):
        name = os.path.splitext(os.path.basename(f))[0]
        targets.append((name, read_target(f)))
    params = {"ratio": args.ratio, "count": args.count, "steps": args.steps, "branchs": args.branchs,
              "nodes": args.nodes}
    if world > 1:
        # one process per GPU (torchrun, or KM_DEVICES=0,1,.. which starts the ranks): targets are
        # sharded, the database records cross the links once, rank 0 prints in target order
        if (args.verbose or args.debug) and rank == 0:
            err.write("km_amd: -v / -d lines are printed by single-process runs only (the ranks return rows, not node lists)\n")
        blocks = kd.find_mutation_sharded(targets, args.jellyfish_fn, params=params)
        if rank == 0:
            out.write(report.HEADER + "\n")
            _print_rows(blocks, out)
            out.write("#Elapsed time:" + str(time.time() - t0) + "\n")
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
        return
    jf = Jellyfish(args.jellyfish_fn, cutoff=args.ratio, n_cutoff=args.count)
    out.write(report.HEADER + "\n")
    finder = BatchFinder(jf, args.steps, args.branchs, args.nodes)
    # The reference builds every RefSeq before the first MutationFinder (km/tools/find_mutation.py:37-45):
    # a target that is too short, has a non-ACGT base or repeats a k-mer stops the run BEFORE any row is
    # printed.  Those errors come out of the GPU batch here, so with more than one batch the rows are
    # held back until every batch has passed that check (up to STREAM_ABOVE targets; beyond that — some
    # GB of text — batches are printed as they finish and such an error may follow rows already out).
    hold = CHUNK < len(targets) <= STREAM_ABOVE
    held = []
    sink = out

    def release():
        for text in held:
            out.write(text)
        del held[:]

    for lo in range(0, len(targets), CHUNK):
        part = targets[lo:lo + CHUNK]
        if hold:
            sink = io.StringIO()
        try:
            finder.write_rows(part, sink)          # native reporting (km_report_rows), one write per batch
        except NodeLimitExceeded as e:
            if hold:
                held.append(sink.getvalue())
                release()                          # the rows of the earlier targets, then the reference's exit
            out.flush()
            sys.exit(str(e))
        except BaseException as e:
            if hold and not getattr(e, "km_input_error", False):
                held.append(sink.getvalue())
                release()                          # an exception while naming a variant: the earlier rows are out
                out.flush()
            raise
        if hold:
            held.append(sink.getvalue())
        if args.verbose or args.debug:
            glog = finder.graph_log(len(part))
            for t in range(len(part)):
                _verbose_lines(part[t], finder.last_raw, t, jf.k, err, glog)
        if not hold:
            out.flush()
    release()
    out.flush()
    out.write("#Elapsed time:" + str(time.time() - t0) + "\n")


def main_samples(args, out=None):
    """catalog x N samples, sample-sharded over the ranks (km_amd.dist.sample_matrix): one TSV
    stream per target for `km find_report -t <target.fa> -f table`, replacing the loop of
    example/run_leucegene.sh:29-35."""
    out = sys.stdout if out is None else out
    from . import dist as kd
    rank, _local_rank, world = kd.env_world()
    if world > 1:
        import torch
        dev = kd.local_device()
        torch.cuda.set_device(dev)
        kd.init(os.environ.get("KM_DIST_BACKEND", "nccl"), torch.device("cuda", dev))
    params = {"ratio": args.ratio, "count": args.count, "steps": args.steps, "branchs": args.branchs,
              "nodes": args.nodes}
    files = kd.sample_matrix(list(args.jellyfish_fn), list_target_files(args.targets), args.out_dir, params)
    if rank == 0:
        for f in files:
            out.write(f + "\n")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def main_min_cov(args, out=None):
    """km/tools/min_cov.py:10-25 over the batched probe kernel."""
    out = sys.stdout if out is None else out
    dbs = list_target_files(args.jellyfish_fn)
    seq = args.target_fn
    if os.path.isfile(seq):
        seq = read_target(seq)
    from . import common
    out.write("DB\tcount\tlength\tmin\tmax\tmean\tkmer_nb\tkmer_nb_0\n")
    for db in dbs:
        res = common.get_cov(db, seq)
        common.close(db)                 # one database in HBM at a time, as the reference holds one (min_cov.py:18-25)
        out.write("%s\t%d\t%d\t%d\t%d\t%.2f\t%d\t%d\n" % ((db,) + tuple(res)))


def main(argv=None):
    parser = argparse.ArgumentParser(prog="km")
    sub = parser.add_subparsers(dest="_cmd")
    fm = sub.add_parser("find_mutation")
    add_find_mutation_args(fm)
    mc = sub.add_parser("min_cov")
    mc.add_argument("target_fn")
    mc.add_argument("jellyfish_fn", nargs="*")
    sm = sub.add_parser("samples", help="catalog x N samples -> one TSV stream per target (find_report -f table)")
    for flag, dest, default, typ in (("-c", "--count", 5, int), ("-p", "--ratio", 0.05, float),
                                     ("-s", "--steps", 500, int), ("-b", "--branchs", 10, int),
                                     ("-n", "--nodes", 10000, int)):
        sm.add_argument(flag, dest, default=default, type=typ)
    sm.add_argument("-t", "--targets", nargs="+", required=True, help="target FASTA files or one directory")
    sm.add_argument("-o", "--out-dir", required=True)
    sm.add_argument("jellyfish_fn", nargs="+", help="one .jf per sample")
    args = parser.parse_args(argv)
    cmd = args._cmd
    del args._cmd
    # KM_DEVICES=0,1,..: one rank per listed GPU, started here (before anything touches a GPU)
    if cmd in ("find_mutation", "samples") and "RANK" not in os.environ:
        from . import dist as kd
        devs = kd.devices_from_env()
        if devs and len(devs) > 1:
            sys.exit(kd.launch_ranks(len(devs), sys.argv[1:] if argv is None else list(argv)))
    if cmd == "find_mutation":
        main_find_mut(args)
    elif cmd == "samples":
        main_samples(args)
    elif cmd == "min_cov":
        main_min_cov(args)
    else:
        parser.print_help(sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
