# one contig group, kernel stats only (diagnostic): the kernels' own times in a short run
cd /root/repo
mkdir -p gpurun_out
HS_BENCH_NO_PROBE=1 timeout 300 python bench.py --cpu-contigs 0 --steps 6 --warmup 2 --groups 1 > gpurun_out/g1_bench.json 2> gpurun_out/g1_bench.err
python - <<P
import json
j=json.load(open('gpurun_out/g1_bench.json'))
print(round(j['ms_per_step'],2), 'ms/step at one group')
for k,v in list(j['kernels'].items())[:26]: print('%-28s %.3f ms x%.0f' % (k, v['ms_per_step'], v['launches_per_step']))
P
