#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against a known byte count, by access pattern (tools/probes/fetch_calib.hip): the factors tools/pmc_traffic_json.py applies.
# usage (through gpurun): bash tools/fetch_calib.sh <tag>   -> gpurun_out/fetch_calib_<tag>.txt
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
MB=${HS_CALIB_MB:-2048}
out=$R/gpurun_out/fetch_calib_${tag}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for g in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $out/$g -- $R/tools/probes/fetch_calib $MB > $out/$g.log 2> $out/$g.err
done
python3 - $out $MB > $R/gpurun_out/fetch_calib_${tag}.txt <<'P'
import csv, glob, os, sys, collections
root, mb = sys.argv[1], int(sys.argv[2])
n = mb << 20
acc = collections.defaultdict(float); cnt = collections.Counter()
for g in ("FETCH_SIZE", "WRITE_SIZE"):
    seen = collections.defaultdict(float)
    for f in glob.glob(os.path.join(root, g, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            seen[(row["Kernel_Name"].split("(")[0], row["Dispatch_Id"])] += float(row["Counter_Value"])
    for (k, d), v in seen.items():
        acc[(k, g)] += v; cnt[(k, g)] += 1
print("# bytes per launch %d (%d MiB); counter x 1024 / bytes, mean over the launches" % (n, mb))
print("%-10s %10s %10s %s" % ("kernel", "FETCH/bytes", "WRITE/bytes", "launches"))
for k in sorted(set(k for k, _ in acc)):
    f = acc[(k, "FETCH_SIZE")] * 1024 / max(1, cnt[(k, "FETCH_SIZE")]) / n
    w = acc[(k, "WRITE_SIZE")] * 1024 / max(1, cnt[(k, "WRITE_SIZE")]) / n
    print("%-10s %10.4f %10.4f %d" % (k, f, w, cnt[(k, "FETCH_SIZE")]))
for g in ("FETCH_SIZE",):
    print("# timings (", g, "run ):")
    print(open(os.path.join(root, g + ".log")).read())
P
find $out -name "*.csv" -size +1M -delete
cat $R/gpurun_out/fetch_calib_${tag}.txt
