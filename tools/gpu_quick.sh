# quick parity round: drop-in tests + kernel tests (no full configs), then a short bench line
cd /root/repo
mkdir -p gpurun_out
TAG=${1:-q}
timeout 1200 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_kernels.py -x -q > gpurun_out/${TAG}_pytest.log 2>&1
tail -5 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py --cpu-contigs 0 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 300 gpurun_out/${TAG}_bench.err
python - <<P
import json
j=json.load(open('gpurun_out/${TAG}_bench.json'))
print(round(j['value']/1e9,2), 'Gbp/s', round(j['ms_per_step'],2), 'ms', j['host']['process_cpu_ms_per_step'], 'CPU-ms', j['host']['waits_per_step'], 'waits')
print(j['pipeline_wall_ms_per_step'], j['phase_ms_per_step'])
for k,v in list(j['kernels'].items())[:30]: print(k, round(v['ms_per_step'],3), v['launches_per_step'], round(v['achieved_GBs'],1))
P
