#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --memory-copy-trace run of the PIPELINED bench (default: 4 batches in
flight on 4 streams): per stream the sequence k_pack .. k_out_pack, D2H; what the copy engine and the CUs
were doing; how long a stream sits idle between the end of its copy and its next k_pack.
usage: timeline_summary.py <trace dir> <out.json>"""
import collections
import csv
import glob
import json
import sys

root, out_path = sys.argv[1], sys.argv[2]


def rows(pattern):
    for path in sorted(glob.glob(root + "/**/" + pattern, recursive=True)):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                yield r


def col(r, *names):
    for n in names:
        if n in r and r[n] != "":
            return r[n]
    return None


kern = []
for r in rows("*kernel_trace.csv"):
    name = (col(r, "Kernel_Name") or "").split("(")[0].replace("void ", "").replace("kmd::", "")
    short = name.split("<")[0]
    kern.append((int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")), short,
                 col(r, "Stream_Id", "Queue_Id") or "?"))
copies = []
for r in rows("*memory_copy_trace.csv"):
    copies.append((int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")), col(r, "Direction") or "?",
                   col(r, "Stream_Id", "Queue_Id") or "?"))
kern.sort()
copies.sort()
ours = ("k_pack", "k_seed", "k_dfs", "k_graph_pure", "k_graph", "k_out_scan", "k_out_pack")
steps_by_stream = collections.defaultdict(list)        # stream -> list of dicts
cur = {}
for s, e, name, st in kern:
    if name not in ours:
        continue
    if name == "k_pack":
        cur[st] = {"stream": st, "start": s, "kernels": {}, "k_end": e}
        steps_by_stream[st].append(cur[st])
    step = cur.get(st)
    if step is None:
        # k_graph_pure runs on the batch's side stream: attach it to the step in flight that started last
        live = [v for v in cur.values() if v["start"] <= s]
        step = max(live, key=lambda v: v["start"]) if live else None
        if step is None:
            continue
    step["kernels"][name] = step["kernels"].get(name, 0) + (e - s)
    step["k_end"] = max(step["k_end"], e)
d2h = [c for c in copies if "DEVICE_TO_HOST" in c[2].upper() or "D2H" in c[2].upper()]
# a delivered step: the first D2H that starts after its k_out_pack ended, before the stream's next k_pack
all_steps = sorted((st for lst in steps_by_stream.values() for st in lst), key=lambda v: v["start"])
for lst in steps_by_stream.values():
    for i, stp in enumerate(lst):
        nxt = lst[i + 1]["start"] if i + 1 < len(lst) else None
        stp["next_start"] = nxt
        if "k_out_pack" not in stp["kernels"]:
            continue
        cand = [c for c in d2h if c[0] >= stp["k_end"] - 1000 and (nxt is None or c[0] < nxt) and c[1] - c[0] > 20000]
        if cand:
            stp["d2h"] = (cand[0][0], cand[0][1])
delivered = [v for v in all_steps if "d2h" in v and len(v["kernels"]) >= 6]
# the pipelined region: the longest stretch of delivered steps whose starts are < 2 ms apart, >= 3 streams alternating
best, run = [], []
for v in delivered:
    if run and v["start"] - run[-1]["start"] > 2_000_000:
        if len(run) > len(best):
            best = run
        run = []
    run.append(v)
if len(run) > len(best):
    best = run
res = {"kernel_rows": len(kern), "copy_rows": len(copies), "delivered_steps_found": len(delivered)}
if len(best) >= 8:
    t0, t1 = best[0]["start"], max(v["d2h"][1] for v in best)
    wall = t1 - t0
    n = len(best)

    def union(iv):
        iv = sorted(iv)
        tot, ce = 0, None
        cs = None
        for s, e in iv:
            if ce is None or s > ce:
                if ce is not None:
                    tot += ce - cs
                cs, ce = s, e
            else:
                ce = max(ce, e)
        if ce is not None:
            tot += ce - cs
        return tot

    kin = [(max(s, t0), min(e, t1)) for s, e, name, st in kern if name in ours and e > t0 and s < t1]
    cin = [(max(c[0], t0), min(c[1], t1)) for c in d2h if c[1] > t0 and c[0] < t1]
    gaps = [v["next_start"] - v["d2h"][1] for v in best if v["next_start"] is not None and v["next_start"] > v["d2h"][1]]
    waits = [v["d2h"][0] - v["k_end"] for v in best]
    per_k = collections.defaultdict(list)
    for v in best:
        for k_, d in v["kernels"].items():
            per_k[k_].append(d)
    spans = [v["d2h"][1] - v["start"] for v in best]
    res.update({
        "region": {"steps": n, "streams": len({v["stream"] for v in best}), "wall_us": wall / 1e3,
                   "us_per_step": wall / 1e3 / n},
        "some_kernel_running_frac": union(kin) / wall,
        "sum_of_kernel_durations_per_step_us": sum(e - s for s, e in kin) / 1e3 / n,
        "copy_engine_busy_frac": union(cin) / wall,
        "d2h_us": {"mean": sum(e - s for s, e in cin) / max(1, len(cin)) / 1e3, "count": len(cin)},
        "step_span_us_pack_to_copy_end": {"mean": sum(spans) / n / 1e3, "max": max(spans) / 1e3},
        "copy_start_after_last_kernel_us": {"mean": sum(waits) / n / 1e3, "max": max(waits) / 1e3},
        "stream_idle_copy_end_to_next_k_pack_us": {"mean": (sum(gaps) / len(gaps) / 1e3) if gaps else None,
                                                   "max": (max(gaps) / 1e3) if gaps else None, "count": len(gaps)},
        "kernel_us_inside_pipeline": {k_: sum(v) / len(v) / 1e3 for k_, v in per_k.items()},
        "note": "kernel durations inside the pipeline are stretched by sharing the CUs with the other batches' "
                "kernels (compare profiles/*kernel_stats.csv: each kernel alone)",
    })
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res))
