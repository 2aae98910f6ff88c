#!/bin/bash
# usage: tools/gpu_k2_variants.sh "<EXTRA flags variant 1>" ...   (builds on the box; K2's unit tests, then its one-group probe time in bench.py)
for v in "$@"; do
  touch hairsplitter_amd/csrc/hs_capi.hip
  make -s -C hairsplitter_amd/csrc ARCH=gfx950 EXTRA="$v" 2>&1 | grep -E "error" -A3 | head
  ok=$(timeout 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "column or stats or tile" 2>&1 | tail -1)
  timeout 300 python3 bench.py --cpu-contigs 0 --steps 10 > gpurun_out/k2v.json 2> gpurun_out/k2v.err
  python3 - "$v" "$ok" <<'PY'
import json,sys
try:
    j=json.loads(open('gpurun_out/k2v.json').read().strip().splitlines()[-1])
    p=j['roofline']['probe_one_group']['kernels_ms_per_step']
    print('[%s] tests: %s | step %.2f ms | K1 %.4f K2 %.4f | parity %s'%(sys.argv[1], sys.argv[2], j['ms_per_step'], p.get('k_pileup_runs',0), p.get('k_column_stats_tiled',0), (j.get('parity') or {}).get('identical')))
except Exception as e:
    print('[%s] tests: %s | bench failed: %s'%(sys.argv[1], sys.argv[2], e)); print(open('gpurun_out/k2v.err').read()[-600:])
PY
done
