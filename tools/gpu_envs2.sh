# like gpu_envs.sh, plus the summed launch time of the kernels and the dominant one
cd /root/repo
run() { echo "== $1"; env $1 python bench.py --steps 30 --warmup 2 --no-f2f-job 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=sorted(d['step_ms']); r=d['roofline']; print(round(d['value']/1e9,2), 'G bp/s', round(d['ms_per_step'],2), 'ms/step  median', s[len(s)//2], ' cpu', round(d['host']['process_cpu_ms_per_step']), ' kernels ms/step', round(r['whole_path']['sum_of_kernel_ms_per_step'],1), ' dominant', r['kernel'], round(r['ms_per_step'],2), 'ms frac', round(r['frac'],4))"; }
for e in "$@"; do run "$e"; done
