cd /root/repo
mkdir -p gpurun_out

HS_TIMING=1 timeout 600 python bench.py --steps 2 --warmup 1 --cpu-contigs 0 --groups 1 > gpurun_out/tail_diag.json 2> gpurun_out/tail_diag.err
grep "k_window_tail cycles" gpurun_out/tail_diag.err | tail -1
python - <<P
import json
j=json.load(open('gpurun_out/tail_diag.json'))
print(j['ms_per_step'], j['kernels']['k_window_tail'])
P
