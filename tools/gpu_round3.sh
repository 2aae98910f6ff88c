# full GPU suite + the default bench line + rocprofv3 kernel stats of the same command
cd /root/repo
TAG=${1:-r03x}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1
tail -4 gpurun_out/${TAG}_pytest.log
timeout 1500 python bench.py > gpurun_out/${TAG}_bench_c4.json 2> gpurun_out/${TAG}_bench.err
tail -c 400 gpurun_out/${TAG}_bench.err
python - <<P
import json
j=json.load(open('gpurun_out/${TAG}_bench_c4.json'))
print(j['value']/1e9, j['ms_per_step'], j['host'])
print(json.dumps(j['roofline'])[:600])
print(json.dumps(j.get('file_to_file'))[:1500]); print(json.dumps(j.get('cpu_baseline')))
print(j['host_fallbacks_per_step'], j['with_h2d_and_plans'])
for k,v in j['kernels'].items(): print(k, v['ms_per_step'], v['launches_per_step'], round(v['achieved_GBs'],1))
P
