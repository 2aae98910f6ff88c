#!/bin/bash
# K4 after a change: its kernel tests and the goldens, then its own time in bench.py's one-group probe
timeout 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "partition or k4 or column" 2>&1 | tail -1
timeout 900 python3 -m pytest tests/test_gpu_dropin.py -x -q -k "goldens or larger" 2>&1 | tail -1
timeout 300 python3 bench.py --cpu-contigs 0 --steps 10 > gpurun_out/k4q.json 2> gpurun_out/k4q.err
python3 - <<'PY'
import json
j=json.loads(open('gpurun_out/k4q.json').read().strip().splitlines()[-1])
p=j['roofline']['probe_one_group']['kernels_ms_per_step']
print('step %.2f ms | K4 lanes %.4f  K4 test %.4f  transpose %s | parity %s'%(j['ms_per_step'], p.get('k_column_partition_lanes',0), p.get('k_column_partition_test',0), p.get('k_partition_transpose'), (j.get('parity') or {}).get('identical')))
PY
