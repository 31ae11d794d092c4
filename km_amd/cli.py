"""``km find_mutation`` / ``km min_cov`` drop-in command line.

Same flags, same ``#key:value`` echo, same TSV and ``#Elapsed time`` trailer as
km/tools/find_mutation.py:17-60 and km/argparser/find_mutation.py:4-58, so the
output pipes into ``km find_report`` unchanged.  The GPU is selected with the
environment variable KM_DEVICE (no extra flags: the reference's tests index
output lines).
"""

import argparse
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


def main_find_mut(args, out=sys.stdout):
    t0 = time.time()
    for key, val in vars(args).items():
        out.write("#" + str(key) + ":" + str(val) + "\n")
    jf = Jellyfish(args.jellyfish_fn, cutoff=args.ratio, n_cutoff=args.count)
    out.write(report.HEADER + "\n")
    targets = []
    for f in list_target_files(args.target_fn):
        name = os.path.splitext(os.path.basename(f))[0]
        targets.append((name, read_target(f)))
    finder = BatchFinder(jf, args.steps, args.branchs, args.nodes)
    for rows in finder.rows(targets):              # native reporting (km_report_rows)
        if isinstance(rows, NodeLimitExceeded):
            out.flush()
            sys.exit(str(rows))
        if isinstance(rows, BaseException):        # what the reference raises while naming a variant
            raise rows
        for row in rows:
            out.write(row + "\n")
    out.write("#Elapsed time:" + str(time.time() - t0) + "\n")


def main_min_cov(args, out=sys.stdout):
    """km/tools/min_cov.py:10-25 over the batched probe kernel."""
    dbs = list_target_files(args.jellyfish_fn)
    seq = args.target_fn
    if os.path.isfile(seq):
        seq = read_target(seq)
    out.write("DB\tcount\tlength\tmin\tmax\tmean\tkmer_nb\tkmer_nb_0\n")
    for db in dbs:
        jf = Jellyfish(db)
        c = jf.query_seq(seq).astype("int64")
        mean = float(c.sum()) / len(c) if len(c) else 0
        out.write("%s\t%d\t%d\t%d\t%d\t%.2f\t%d\t%d\n" % (db, c.sum(), len(seq), c.min(), c.max(),
                                                          mean, len(c), int((c == 0).sum())))


def main(argv=None):
    parser = argparse.ArgumentParser(prog="km")
    sub = parser.add_subparsers(dest="_cmd")
    fm = sub.add_parser("find_mutation")
    add_find_mutation_args(fm)
    mc = sub.add_parser("min_cov")
    mc.add_argument("target_fn")
    mc.add_argument("jellyfish_fn", nargs="*")
    args = parser.parse_args(argv)
    cmd = args._cmd
    del args._cmd
    if cmd == "find_mutation":
        main_find_mut(args)
    elif cmd == "min_cov":
        main_min_cov(args)
    else:
        parser.print_help(sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
