#!/bin/bash
# usage: tools/gpu_k4_variants.sh "<EXTRA flags>" ...   K4's own time in bench.py's one-group probe per build variant (ablations give wrong results: no tests)
for v in "$@"; do
  make -s -C hairsplitter_amd/csrc ARCH=gfx950 EXTRA="$v" 2>&1 | grep -E "error" -A3 | head
  HS_BENCH_NO_PARITY=1 timeout 300 python3 bench.py --cpu-contigs 0 --steps 6 > gpurun_out/k4v.json 2> gpurun_out/k4v.err
  python3 - "$v" <<'PY'
import json,sys
try:
    j=json.loads(open('gpurun_out/k4v.json').read().strip().splitlines()[-1])
    p=j['roofline']['probe_one_group']['kernels_ms_per_step']
    print('[%s] step %.2f ms | K4 lanes %.4f  K4 test %.4f'%(sys.argv[1], j['ms_per_step'], p.get('k_column_partition_lanes',0), p.get('k_column_partition_test',0)))
except Exception as e:
    print('[%s] failed: %s'%(sys.argv[1], e)); print(open('gpurun_out/k4v.err').read()[-400:])
PY
done
