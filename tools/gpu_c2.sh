cd /root/repo
run() { echo "== $1 | $2"; env $1 python bench.py --config C2 --contigs 256 --steps 20 --warmup 3 --no-f2f-job $2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=sorted(d['step_ms']); k=d['kernels']; print(round(d['value']/1e9,2), 'G bp/s', round(d['ms_per_step'],2), 'ms/step  median', s[len(s)//2], ' cpu', round(d['host']['process_cpu_ms_per_step']), ' cw', round(k.get('k_cw_seeded_lanes',{}).get('ms_per_step',0),2), d['pipeline_wall_ms_per_step'])"; }
run "HS_X=1" ""
run "HS_CW_NO_LANES=1" ""
run "HS_X=1" "--threads 64"
run "HS_X=1" "--threads 32"
