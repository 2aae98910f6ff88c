# kernel tests + drop-in goldens + C2-C4 full sizes, then the bench at groups=1 (intrinsic kernel times) and at the default
cd /root/repo
mkdir -p gpurun_out
TAG=${1:-r03q}
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dropin.py tests/test_gpu_full_configs.py -m gpu -x -q -k "not c5_uncut" > gpurun_out/${TAG}_pytest.log 2>&1
tail -4 gpurun_out/${TAG}_pytest.log
for G in 1 0; do
timeout 900 python bench.py --steps 20 --warmup 3 --cpu-contigs 0 --groups $G > gpurun_out/${TAG}_bench_g$G.json 2> gpurun_out/${TAG}_bench_g$G.err
python - <<P
import json
j=json.load(open('gpurun_out/${TAG}_bench_g$G.json'))
print("groups", j['config']['groups_per_gpu'], j['value']/1e9, j['ms_per_step'], j['host']['process_cpu_ms_per_step'], j['host']['cfs_throttled_during_timed_steps'])
for k,v in list(j['kernels'].items())[:24]: print("  ", k, v['ms_per_step'], v['launches_per_step'], round(v['achieved_GBs'],1))
P
done
