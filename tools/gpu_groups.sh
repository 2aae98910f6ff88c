# ms/step and CPU-ms/step of the default bench for several group counts
cd /root/repo
for g in "$@"; do echo "== groups $g"; python bench.py --steps 30 --warmup 2 --no-f2f-job --groups $g 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=sorted(d['step_ms']); print(round(d['value']/1e9,2), 'G bp/s', round(d['ms_per_step'],2), 'ms/step  median step', s[len(s)//2], ' cpu', round(d['host']['process_cpu_ms_per_step']), d['host']['cfs_throttled_during_timed_steps'])"; done
