cd /root/repo
run() { echo "== $1"; env $1 python bench.py --config C2 --contigs 256 --steps 20 --warmup 3 --no-f2f-job 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=sorted(d['step_ms']); print(round(d['value']/1e9,2), 'G bp/s', round(d['ms_per_step'],2), 'ms/step  median', s[len(s)//2], ' cpu', round(d['host']['process_cpu_ms_per_step']))"; }
for e in "$@"; do run "$e"; done
