#!/usr/bin/env python3
"""Per-kernel own times from bench.py's JSON line(s): `python tools/kernel_ms.py file.json [kernel ...]` (no kernel names: all, by time)."""
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ks = j["kernels"]
names = sys.argv[2:] or sorted(ks, key=lambda k: -ks[k]["ms_per_step"])
print("ms/step %.2f  CPU-ms %.1f  parity %s" % (j["ms_per_step"], j["host"]["process_cpu_ms_per_step"], (j.get("parity") or {}).get("identical")))
for k in names:
    if k in ks: print("%-30s %8.1f us/step  %5.1f launches" % (k, ks[k]["ms_per_step"] * 1e3, ks[k]["launches_per_step"]))
print("sum %.2f ms" % sum(v["ms_per_step"] for v in ks.values()))
