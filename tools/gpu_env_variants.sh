#!/bin/bash
# usage: tools/gpu_env_variants.sh "<bench args>|<ENV=..> <ENV=..>" ...   one bench.py run per variant, ms per step and CPU-ms
mkdir -p gpurun_out
i=0
for v in "$@"; do
  args="${v%%|*}"; envs="${v#*|}"; [ "$envs" = "$v" ] && envs=""
  ( for kv in $envs; do export "$kv"; done; timeout 300 python3 bench.py --cpu-contigs 0 --steps 30 $args > gpurun_out/ev_$i.json 2> gpurun_out/ev_$i.err )
  python3 - "$v" $i <<'PY'
import json,sys
try:
    j=json.loads(open('gpurun_out/ev_%s.json'%sys.argv[2]).read().strip().splitlines()[-1])
    print('[%s] ms/step %.2f cpu %.0f'%(sys.argv[1], j['ms_per_step'], j['host']['process_cpu_ms_per_step']))
except Exception as e: print('[%s] failed %s'%(sys.argv[1], e))
PY
  i=$((i+1))
done
