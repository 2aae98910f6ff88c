cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --steps 10 --warmup 2 --cpu-contigs 0 --groups 1 > gpurun_out/tail_check.json 2> gpurun_out/tail_check.err
python - <<P
import json
j=json.load(open('gpurun_out/tail_check.json'))
print(j['ms_per_step'])
for k,v in list(j['kernels'].items())[:12]: print("  ", k, v['ms_per_step'], v['launches_per_step'], round(v['achieved_GBs'],1))
P
