#!/bin/bash
# usage: tools/gpu_k1_variants.sh "<EXTRA flags variant 1>" "<variant 2>" ...   (builds on the box, prints K1's probe time per variant)
for v in "$@"; do
  touch hairsplitter_amd/csrc/hs_capi.hip
  make -s -C hairsplitter_amd/csrc ARCH=gfx950 EXTRA="$v" 2>&1 | grep -E "error" -A3 | head
  ok=$(timeout 300 python3 -m pytest tests/test_gpu_kernels.py -x -q -k pileup 2>&1 | tail -1)
  timeout 300 python3 bench.py --cpu-contigs 0 --steps 10 > gpurun_out/k1v.json 2> gpurun_out/k1v.err
  python3 - "$v" "$ok" <<'PY'
import json,sys
j=json.loads(open('gpurun_out/k1v.json').read().strip().splitlines()[-1])
p=j['roofline']['probe_one_group']['kernels_ms_per_step']
print('[%s] tests: %s | step %.2f ms | K1 %.4f K2 %.4f K0 %.4f'%(sys.argv[1], sys.argv[2], j['ms_per_step'], p.get('k_pileup_runs',0), p.get('k_column_stats_tiled',0), p.get('k_cigar_scan',0)))
PY
done
