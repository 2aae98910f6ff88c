cd /root/repo
mkdir -p gpurun_out
TAG=${1:-r03b}
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1
tail -15 gpurun_out/${TAG}_pytest.log
HS_TIMING=1 timeout 900 python bench.py --steps 10 --warmup 2 --cpu-contigs 0 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
grep -E "cv range \[0|host waits" gpurun_out/${TAG}_bench.err | tail -4
python - <<P
import json
j=json.load(open('gpurun_out/${TAG}_bench.json'))
print(j['value']/1e9, j['ms_per_step'], j['host'])
for k,v in j['kernels'].items(): print(k, v['ms_per_step'], v['launches_per_step'], round(v['achieved_GBs'],1))
P
