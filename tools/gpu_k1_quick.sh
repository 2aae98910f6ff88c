#!/bin/bash
# K1 after a change: its unit tests, then the one-group probe of bench.py (K1's own time per launch over the whole C4 job)
mkdir -p gpurun_out
timeout 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "pileup" 2>&1 | tail -3
for i in 1 2; do
timeout 300 python3 bench.py --cpu-contigs 0 --steps 10 > gpurun_out/k1q_$i.json 2> gpurun_out/k1q_$i.err
python3 - $i <<'PY'
import json,sys
j=json.loads(open('gpurun_out/k1q_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
r=j['roofline']
print('step %.2f ms  kernel %s  probe %s  frac_probe %s  parity %s'%(j['ms_per_step'], r.get('kernel'), r.get('probe_one_group',{}).get('avg_launch_ms'), r.get('frac_probe_one_group'), j.get('parity',{}).get('identical')))
PY
done
