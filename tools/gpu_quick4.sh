cd /root/repo
mkdir -p gpurun_out
TAG=${1:-r03k}
timeout 1200 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_full_configs.py -m gpu -x -q -k "not c5_uncut" > gpurun_out/${TAG}_pytest.log 2>&1
tail -4 gpurun_out/${TAG}_pytest.log
timeout 900 python bench.py --steps 30 --warmup 3 --cpu-contigs 0 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<P
import json
j=json.load(open('gpurun_out/${TAG}_bench.json'))
print(j['value']/1e9, j['ms_per_step'], j['host'])
for k,v in list(j['kernels'].items())[:26]: print(k, v['ms_per_step'], v['launches_per_step'], round(v['achieved_GBs'],1))
P
