#!/bin/bash
# usage: tools/gpu_k1_alone.sh "<EXTRA flags variant 1>" ...   K1's own duration (rocprofv3 kernel trace) on a batch with nothing behind it: for timing-only ablations
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  touch hairsplitter_amd/csrc/hs_capi.hip
  make -s -C hairsplitter_amd/csrc ARCH=gfx950 EXTRA="$v" 2>&1 | grep -E "error" -A3 | head
  rm -rf gpurun_out/k1alone; mkdir -p gpurun_out/k1alone
  timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k1alone -- python3 tools/gpu_k1_alone.py > gpurun_out/k1alone/out.txt 2> gpurun_out/k1alone/err.txt
  f=$(find gpurun_out/k1alone -name "*kernel_stats.csv" | head -1)
  python3 - "$v" "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[2]))) if sys.argv[2] else []
t = {r["Name"].split("(")[0].replace("hsdev::", ""): float(r["AverageNs"]) / 1e6 for r in rows}
print("[%s] k_pileup_runs avg %.4f ms | k_cigar_scan %.4f | k_run_task_ops %.4f" % (sys.argv[1], t.get("k_pileup_runs", 0), t.get("k_cigar_scan", 0), t.get("k_run_task_ops", 0)))
PY
  rm -rf gpurun_out/k1alone
done
