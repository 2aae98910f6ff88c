"""profiles/SUMMARY_<tag>.md: ONE table per round -- per kernel of the path its launches per step, its own time per step (every
kernel alone on the GPU: rocprofv3 --kernel-trace --stats of `bench.py --groups 1`), the algorithmic bytes per step (as bench.py
counts them: DESIGN.md section 4), the HBM bytes per step the PMC passes measured (FETCH_SIZE + WRITE_SIZE, same one-group command)
and what follows: PMC / algorithmic, and algorithmic bytes / own time as a fraction of the 8 TB/s HBM peak.
usage: make_summary.py <tag> <kernel_stats_groups1.csv> <bench_groups1.json> <bench_default.json> [traffic_groups1.json] > profiles/SUMMARY_<tag>.md"""
import csv
import json
import sys

tag, stats_csv, g1_json, def_json = sys.argv[1:5]
traffic = json.load(open(sys.argv[5])) if len(sys.argv) > 5 else {}
g1 = json.load(open(g1_json))
df = json.load(open(def_json))
steps_g1 = g1["steps"] + g1["warmup"] + g1.get("setup_steps", 0)
ALIAS = {"k_column_stats_tiled_dw": "k_column_stats_tiled", "k_gather_tiles_direct": "k_gather_tiles", "k_read_graph_rows<false>": "k_read_graph_rows", "k_read_graph_rows<true>": "k_read_graph_rows"}
own = {}
for r in csv.DictReader(open(stats_csv)):
    name = r["Name"].split("(")[0].replace("void ", "").replace("hsdev::", "").strip()
    base = ALIAS.get(name, name.split("<")[0])
    o = own.setdefault(base, {"calls": 0, "ns": 0.0, "names": []})
    o["calls"] += int(r["Calls"]); o["ns"] += float(r["TotalDurationNs"]); o["names"].append(name)
rows = []
for k, v in g1["kernels"].items():
    if k not in own or k in ("other",):
        continue
    o = own[k]
    launches = o["calls"] / steps_g1
    us = o["ns"] / steps_g1 / 1e3
    alg = v["algorithmic_bytes_per_launch"] * v["launches_per_step"]
    pmc = traffic.get(k)
    per_step = traffic.get("_dispatches", {}).get(k, 0) / max(1, traffic.get("_steps", 1))
    pmc_step = pmc * per_step if pmc else None
    fe = traffic.get("_fetch", {}).get(k); wr = traffic.get("_write", {}).get(k)
    rows.append((k, df["kernels"].get(k, {}).get("launches_per_step", 0), launches, us, alg, pmc_step, fe * per_step if fe is not None else None, wr * per_step if wr is not None else None))
rows.sort(key=lambda r: -r[3])
print("# Kernels of the path, round %s: own time, bytes, HBM fraction\n" % tag)
print("Workload: %s. Own time = rocprofv3 `--kernel-trace --stats` of `bench.py --groups 1` (one contig group: every kernel alone on the GPU), per step;" % df["config"]["workload"])
print("algorithmic bytes as `bench.py` counts them (DESIGN.md section 4); PMC = 2 x FETCH_SIZE + WRITE_SIZE of the same one-group command (tools/pmc_traffic.sh; gfx950's FETCH_SIZE")
print("counts half of a streaming read's bytes at every access width of this path: tools/probes/fetch_calib.hip, profiles/r06_fetch_calib.txt), read and written shown apart.")
print("Default run of the same commit: **%.2f ms per step** (%d contig groups, %s), %.0f CPU-ms per step, %.0f host waits per step.\n" % (
    df["ms_per_step"], df["config"]["groups_per_gpu"], df["config"]["pipeline"], df["host"]["process_cpu_ms_per_step"], df["host"]["waits_per_step"]))
print("| kernel | launches / step (default) | launches / step (one group) | own time / step (us) | algorithmic MB / step | PMC read MB / step | PMC written MB / step | PMC / algorithmic | fraction of 8 TB/s (algorithmic) | PMC bytes / own time (TB/s) |")
print("|---|---|---|---|---|---|---|---|---|---|")
tot = 0.0
for k, ld, l1, us, alg, pmc, fe, wr in rows:
    tot += us
    frac = alg / (us * 1e-6) / 8e12 if us > 0 and alg > 0 else None
    print("| `%s` | %.0f | %.1f | %.0f | %s | %s | %s | %s | %s | %s |" % (k, ld, l1, us, ("%.1f" % (alg / 1e6)) if alg else "-", ("%.1f" % (fe / 1e6)) if fe is not None else "-",
                                                                  ("%.1f" % (wr / 1e6)) if wr is not None else "-", ("%.2f" % (pmc / alg)) if pmc and alg else "-", ("%.3f" % frac) if frac else "-",
                                                                  ("%.2f" % (pmc / (us * 1e-6) / 1e12)) if pmc and us > 0 else "-"))
others = sorted(((n, o) for n, o in own.items() if n not in g1["kernels"]), key=lambda kv: -kv[1]["ns"])
for n, o in others[:12]:
    print("| `%s` (helper) | | %.1f | %.0f | | | | |" % (n, o["calls"] / steps_g1, o["ns"] / steps_g1 / 1e3))
    tot += o["ns"] / steps_g1 / 1e3
print("\nSum of the rows: %.2f ms of kernel time per step with one contig group." % (tot / 1e3))
wp = df["roofline"]["whole_path"]
print("Whole path by SURVEY.md 8(d)'s formula: %.2f GB per step; against the step time %.3f of 8 TB/s, against the summed kernel time of the default run %.3f."
      % (wp["bytes_per_step"] / 1e9, wp["frac_vs_step_time"], wp["frac_vs_kernel_time"] or 0))
