"""traffic.json from the summary of tools/pmc_traffic.sh: HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) * 1024 per kernel -- on gfx950
FETCH_SIZE reports half of the bytes of a streaming read whatever its width (1, 4 or 16 bytes per lane, aligned or not: calibrated on this path's own
access patterns with tools/probes/fetch_calib.hip, profiles/r06_fetch_calib.txt: 0.500-0.563 of the bytes; MI355X_MICROARCH.md states it for 16 B per lane),
WRITE_SIZE the bytes themselves (1.00-1.04).
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
out = {"_note": "HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) * 1024, mean over the dispatches of "
                "'python bench.py --steps 2 --warmup 0 --cpu-contigs 0 --groups 1' (the default workload, ONE contig group: a launch = the whole job), separate rocprofv3 --pmc "
                "passes (tools/pmc_traffic.sh). The factor 2: gfx950's FETCH_SIZE counts half of a streaming read's bytes at every access width of this path "
                "(tools/probes/fetch_calib.hip, profiles/r06_fetch_calib.txt; MI355X_MICROARCH.md, HBM); WRITE_SIZE is exact. Infinity-Cache hits are counted. "
                "_fetch / _write: the two parts per kernel, corrected.",
       "_config": j["config"]["config"], "_contigs": j["config"]["contigs"], "_aligned_bp": j["config"]["aligned_bp"],
       "_groups_per_gpu": j["config"]["groups_per_gpu"], "_groups": j["config"]["groups_per_gpu"], "_steps": j["steps"] + j["warmup"] + j.get("setup_steps", 0),
       "_commit": os.environ.get("HS_COMMIT") or None, "_dispatches": {}, "_fetch": {}, "_write": {}}
try:
    if not out["_commit"]:
        import subprocess
        out["_commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.decode().strip() or None
except Exception:
    pass
if not out["_commit"] and os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", ".commit")):
    out["_commit"] = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", ".commit")).read().strip() or None
# the stat slot of bench.py is named after the kernel family: the four-positions-per-lane form of K2 reports under k_column_stats_tiled
ALIAS = {"k_column_stats_tiled_dw": "k_column_stats_tiled", "k_gather_tiles_direct": "k_gather_tiles"}
for k, v in sorted(by.items()):
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        name = k.split("<")[0]
        name = ALIAS.get(name, name)
        n_new = v["FETCH_SIZE"][1]
        fe, wr = 2.0 * v["FETCH_SIZE"][0] * 1024.0, v["WRITE_SIZE"][0] * 1024.0
        val = fe + wr
        if name in out["_dispatches"]:      # two kernels of one family: mean over all their dispatches
            n_old = out["_dispatches"][name]
            val = (out[name] * n_old + val * n_new) / max(1, n_old + n_new)
            fe = (out["_fetch"][name] * n_old + fe * n_new) / max(1, n_old + n_new)
            wr = (out["_write"][name] * n_old + wr * n_new) / max(1, n_old + n_new)
            n_new += n_old
        out[name] = val; out["_fetch"][name] = fe; out["_write"][name] = wr
        out["_dispatches"][name] = n_new
print(json.dumps(out, indent=1))
