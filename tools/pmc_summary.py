"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (averages per dispatch)
and derive the HBM traffic of k_seed as MI355X_MICROARCH.md prescribes: read bytes from the
L2's memory-side request counters by request size (128-B requests are tallied at 64 B by
FETCH_SIZE on gfx950, so sizes are applied explicitly), write bytes from WRITE_SIZE."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(root + "/pmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kmd::", "")
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for kname, ctrs in agg.items():
    if not any(x in kname for x in ("k_seed", "k_dfs", "k_graph", "k_pack", "k_out")):
        continue
    out[kname] = {c: sum(v) / len(v) for c, v in ctrs.items()}
    out[kname]["dispatches"] = len(next(iter(ctrs.values())))
# k_seed comes in two instantiations (with / without the table-slot count, KM_RUN_COUNT_FETCHES): the one the timed
# steps launch is the one with the most dispatches; the other (the bench's warm-up) is kept but not summed
seeds = sorted((k_ for k_ in out if k_.startswith("k_seed")), key=lambda k_: -out[k_]["dispatches"])
for k_ in seeds[1:]:
    out["(warm-up) " + k_] = out.pop(k_)
res = {"per_kernel_avg_per_dispatch": out, "round": sys.argv[3] if len(sys.argv) > 3 else None,
       "collected_by": "tools/collect_evidence.sh: rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --steps 20 --warmup 4 "
                       "--no-cpu --only-step --check 0 --inflight 4 --one-at-a-time --serial (one pass per counter group; every launch "
                       "a kernel alone over one of four distinct target sets, never the set of the launch before it)"
                       if not (len(sys.argv) > 3 and sys.argv[3].endswith("_hard")) else
                       "tools/collect_evidence.sh: rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --steps 16 --warmup 4 "
                       "--no-cpu --profile-hard --inflight 2 (config4_hard, every kernel alone, rotating over its two target sets)"}
ks = next((v for k_, v in out.items() if k_.startswith("k_seed")), None)
if ks and "TCC_EA0_RDREQ_sum" in ks:
    r128 = ks.get("TCC_EA0_RDREQ_128B_sum", 0.0)
    r32 = ks.get("TCC_EA0_RDREQ_32B_sum", 0.0)
    r64 = ks["TCC_EA0_RDREQ_sum"] - r128 - r32
    rd = r128 * 128 + r64 * 64 + r32 * 32
    wr = ks.get("WRITE_SIZE", 0.0) * 1024
    res["k_seed_hbm_read_bytes_per_launch"] = rd
    res["k_seed_hbm_write_bytes_per_launch"] = wr
    res["k_seed_hbm_bytes_per_launch"] = rd + wr
    res["note"] = ("reads = RDREQ_128B*128 + RDREQ_64B*64 + RDREQ_32B*32 (FETCH_SIZE counts every "
                   "request as 64 B on gfx950 and under-reports by 2x here); writes = WRITE_SIZE KiB")


def hbm_bytes(c):
    r128 = c.get("TCC_EA0_RDREQ_128B_sum", 0.0)
    r32 = c.get("TCC_EA0_RDREQ_32B_sum", 0.0)
    r64 = c.get("TCC_EA0_RDREQ_sum", 0.0) - r128 - r32
    return r128 * 128 + r64 * 64 + r32 * 32 + c.get("WRITE_SIZE", 0.0) * 1024


# the walk stage = k_pack + k_seed + k_dfs (one launch each per step)
walk = [v for k_, v in out.items() if k_.startswith(("k_pack", "k_seed", "k_dfs")) and "TCC_EA0_RDREQ_sum" in v]
if walk:
    res["walk_stage_hbm_bytes_per_step"] = sum(hbm_bytes(v) for v in walk)
    res["per_kernel_hbm_bytes"] = {k_: hbm_bytes(v) for k_, v in out.items() if "TCC_EA0_RDREQ_sum" in v}
json.dump(res, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "per_kernel_avg_per_dispatch"}))
