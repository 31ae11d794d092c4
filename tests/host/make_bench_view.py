"""Dump a delivery view of bench-like shape (500-nt targets, k = 31, 30 % with a variant) for the CPU-only
timing driver tests/host/report_time.cpp.  The results come from the oracle (no GPU needed):
    python tests/host/make_bench_view.py /tmp/view.bin [n_targets]
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from km_amd import kmer as km, synth          # noqa: E402
from oracle import km_oracle as ko            # noqa: E402
import test_report_native as trn             # noqa: E402


def main():
    out = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    case = synth.make_case(n_targets=n, length=500, k=31, n_keys=200000, seed=77, variant_frac=float(os.environ.get("VFRAC", "0.3")),
                           variants_per_target=(1, 1), cov=(60, 500))
    db = ko.KmerDB(None, cutoff=0.05, n_cutoff=5,
                   records={"k": 31, "canonical": True, "keys": case["keys"], "counts": case["counts"]})
    results, seqs = [], []
    for row, name in zip(case["targets"], case["names"]):
        seq = km.decode(row)
        results.append(ko.analyse_target(seq, name, db))
        seqs.append(seq)
    trn._dump_view(out, trn._raw_from_oracle(results), seqs, 31)
    print("%d targets, %d with more than one path" % (n, sum(len(r["paths"]) > 1 for r in results)))


if __name__ == "__main__":
    main()
