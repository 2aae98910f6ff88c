# quick: kernel tests, then intrinsic kernel times (groups=1)
cd /root/repo
mkdir -p gpurun_out
if [ "$1" != "notest" ]; then timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -2; fi
timeout 600 python bench.py --steps 10 --warmup 2 --cpu-contigs 0 --groups 1 > gpurun_out/k_quick.json 2> gpurun_out/k_quick.err
python - <<P
import json
j=json.load(open('gpurun_out/k_quick.json'))
print(j['ms_per_step'])
for k,v in list(j['kernels'].items())[:14]: print("  ", k, v['ms_per_step'], v['launches_per_step'], round(v['achieved_GBs'],1))
P
