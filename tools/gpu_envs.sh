# A/B of environment switches on the default bench: ms/step, CPU-ms/step (every run bounded: a hung run must not eat the box)
cd /root/repo
run() { echo "== $1"; env $1 timeout -s USR1 -k 15 200 python bench.py --steps 30 --warmup 2 --no-f2f-job 2>gpurun_out/envs_last.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=sorted(d['step_ms']); print(round(d['value']/1e9,2), 'G bp/s', round(d['ms_per_step'],2), 'ms/step  median step', s[len(s)//2], ' cpu', round(d['host']['process_cpu_ms_per_step']), d['host']['cfs_throttled_during_timed_steps'])" || tail -5 gpurun_out/envs_last.err; }
for e in "$@"; do run "$e"; done
