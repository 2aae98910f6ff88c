#!/bin/bash
# usage: tools/gpu_myers_variants.sh "<EXTRA flags>" ...   A1 on 4096 reads of 10 kb against 12-kb windows (with paths / locations only), per build variant
for v in "$@"; do
  make -s -C hairsplitter_amd/csrc ARCH=gfx950 EXTRA="$v" 2>&1 | grep -E "error" -A3 | head
  ok=$(timeout 600 python3 -m pytest tests/test_gpu_realign.py tests/test_gpu_kernels.py -x -q -k "myers or edlib or realign or align" 2>&1 | tail -1)
  timeout 600 python3 tools/myers_bench.py ${MY_N:-4096} 12000 10000 > gpurun_out/myv.json 2> gpurun_out/myv.err
  python3 - "$v" "$ok" <<'PY'
import json,sys
try:
    j=json.loads(open('gpurun_out/myv.json').read().strip().splitlines()[-1])
    print('[%s] tests: %s | with paths %.0f reads/s, locations only %.0f reads/s'%(sys.argv[1], sys.argv[2], j['path']['pairs_per_s'], j['locations_only']['pairs_per_s']))
except Exception as e:
    print('[%s] tests: %s | failed %s'%(sys.argv[1], sys.argv[2], e))
PY
done
make -s -C hairsplitter_amd/csrc ARCH=gfx950 2>&1 | grep -E "error" | head -2
