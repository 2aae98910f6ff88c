#!/bin/bash
# bench.py's file-to-file leg (drop-ins on the whole C4 job, no detach / detached) with and without the huge-page restart, alternating
mkdir -p gpurun_out
for rep in 1 2; do for mode in on off; do
  if [ $mode = off ]; then export HS_NO_REEXEC=1; else unset HS_NO_REEXEC; fi
  timeout 600 python3 bench.py --steps 3 --warmup 1 --no-f2f-reference-full > gpurun_out/f2fab_${mode}_$rep.json 2> gpurun_out/f2fab_${mode}_$rep.err
  python3 - $mode $rep <<'PY'
import json,sys
j=json.loads(open('gpurun_out/f2fab_%s_%s.json'%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
f=j['file_to_file']['job']
print('restart %s run %s: no detach cv %.3f sr %.3f total %.3f | detached cv %.3f sr %.3f total %.3f'%(sys.argv[1],sys.argv[2],f['dropin_s']['call_variants'],f['dropin_s']['separate_reads'],f['dropin_s']['total'],f['dropin_detached_s']['call_variants'],f['dropin_detached_s']['separate_reads'],f['dropin_detached_s']['total']))
PY
done; done
