#!/bin/bash
# experiment: the contig groups' shares (HS_GROUP_WEIGHTS) against the default taper; same box, alternating runs
mkdir -p gpurun_out
run() { # tag groups weights
  if [ -n "$3" ]; then export HS_GROUP_WEIGHTS="$3"; else unset HS_GROUP_WEIGHTS; fi
  timeout 300 python3 bench.py --cpu-contigs 0 --steps 30 --groups $2 > gpurun_out/gw_$1.json 2> gpurun_out/gw_$1.err
  python3 - "$1" "$3" <<'PY'
import json,sys
try:
    j=json.loads(open('gpurun_out/gw_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], sys.argv[2] or 'default', 'ms/step %.2f'%j['ms_per_step'], 'cpu %.0f'%j['host']['cpu_ms_per_step'] if 'cpu_ms_per_step' in j.get('host',{}) else j.get('host'))
except Exception as e: print(sys.argv[1], 'failed', e)
PY
}
for rep in 1 2; do
run d$rep 8 ""
run a$rep 8 "0.5,0.9,1,1,0.9,0.8,0.6,0.4"
run b$rep 8 "0.4,0.7,1,1,1,0.8,0.6,0.4"
run c$rep 10 "0.5,0.8,1,1,1,0.9,0.8,0.7,0.55,0.4"
run e$rep 8 "0.7,1,1,1,0.9,0.8,0.6,0.4"
run f$rep 8 "1,1,1,0.9,0.8,0.65,0.5,0.35"
done
