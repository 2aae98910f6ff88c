# round-3 baseline: GPU suite, bench with a host CPU profile, bench with HS_TIMING wait counts
cd /root/repo
mkdir -p gpurun_out
TAG=${1:-r03a}
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1
tail -5 gpurun_out/${TAG}_pytest.log
HS_CPU_PROFILE=/root/repo/gpurun_out/${TAG}_cpu_prof.txt timeout 900 python bench.py --steps 20 --warmup 3 --cpu-contigs 0 > gpurun_out/${TAG}_bench_prof.json 2> gpurun_out/${TAG}_bench_prof.err
python tools/cpuprof_report.py gpurun_out/${TAG}_cpu_prof.txt 60 > gpurun_out/${TAG}_cpu_profile_top.txt 2>&1
python - <<P
import json
j=json.load(open('gpurun_out/${TAG}_bench_prof.json'))
print(j['value']/1e9, j['ms_per_step'], j['host'])
P
HS_TIMING=1 timeout 900 python bench.py --steps 3 --warmup 1 --cpu-contigs 0 > gpurun_out/${TAG}_bench_timing.json 2> gpurun_out/${TAG}_bench_timing.err
grep "host waits" gpurun_out/${TAG}_bench_timing.err | tail -3
