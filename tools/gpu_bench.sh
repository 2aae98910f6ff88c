set -x
cd /root/repo
mkdir -p gpurun_out
TAG=${1:-r02c}
shift
timeout 1500 python bench.py --steps 10 --warmup 2 "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench.err
python - <<P
import json
j=json.load(open('gpurun_out/${TAG}_bench.json'))
print(j['value']/1e9, j['ms_per_step'], j['host'])
print(json.dumps(j['roofline'])[:1500])
for k,v in j['kernels'].items(): print(k, v)
print(json.dumps(j.get('file_to_file')), json.dumps(j.get('cpu_baseline')))
print(j['pipeline_wall_ms_per_step'], j['phase_ms_per_step'])
P
