"""traffic.json from the summary of tools/pmc_traffic.sh: HBM bytes per launch = (FETCH_SIZE + WRITE_SIZE) * 1024 per kernel.
usage: python tools/pmc_traffic_json.py gpurun_out/pmc_<tag>  > profiles/traffic_latest.json"""
import collections
import csv
import json
import os
import sys

root = sys.argv[1]
rows = list(csv.DictReader(open(os.path.join(root, "summary.csv"))))
by = collections.defaultdict(dict)
for r in rows:
    by[r["kernel"].strip()][r["counter"]] = (float(r["mean_per_dispatch"]), int(r["dispatches"]))
j = json.load(open(os.path.join(root, "g0", "bench.json")))
out = {"_note": "HBM bytes per launch = (FETCH_SIZE + WRITE_SIZE) * 1024, mean over the dispatches of "
                "'python bench.py --steps 2 --warmup 0 --cpu-contigs 0 --groups 1' (the default workload, ONE contig group: a launch = the whole job), separate rocprofv3 --pmc "
                "passes (tools/pmc_traffic.sh). Accesses of these kernels are 1-4 B per lane: the gfx950 half-counting of 16-B/lane "
                "streaming reads (MI355X_MICROARCH.md, HBM) is not applied; Infinity-Cache hits are counted.",
       "_config": j["config"]["config"], "_contigs": j["config"]["contigs"], "_aligned_bp": j["config"]["aligned_bp"],
       "_groups_per_gpu": j["config"]["groups_per_gpu"], "_groups": j["config"]["groups_per_gpu"], "_steps": j["steps"] + j["warmup"] + j.get("setup_steps", 0),
       "_commit": os.environ.get("HS_COMMIT"), "_dispatches": {}}
# the stat slot of bench.py is named after the kernel family: the four-positions-per-lane form of K2 reports under k_column_stats_tiled
ALIAS = {"k_column_stats_tiled_dw": "k_column_stats_tiled", "k_column_stats_tiled_dw_plain": "k_column_stats_tiled", "k_gather_tiles_direct": "k_gather_tiles"}
for k, v in sorted(by.items()):
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        name = k.split("<")[0]
        name = ALIAS.get(name, name)
        n_new = v["FETCH_SIZE"][1]
        val = (v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * 1024.0
        if name in out["_dispatches"]:      # two kernels of one family: mean over all their dispatches
            n_old = out["_dispatches"][name]
            val = (out[name] * n_old + val * n_new) / max(1, n_old + n_new)
            n_new += n_old
        out[name] = val
        out["_dispatches"][name] = n_new
print(json.dumps(out, indent=1))
