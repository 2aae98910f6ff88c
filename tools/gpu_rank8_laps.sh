# HS_TIMING laps of a rank of 8 (its 1/8 shard, the threads and groups a rank gets): where the short chain spends its wall time
cd /root/repo
mkdir -p gpurun_out
HS_TIMING=abs HS_BENCH_NO_PROBE=1 timeout 600 python bench.py --as-rank-of 8 --steps 4 --warmup 2 --cpu-contigs 0 > gpurun_out/r8laps_bench.json 2> gpurun_out/r8laps.err
grep -E "laps \(ms\)|fused call|\[hs timing\]" gpurun_out/r8laps.err | tail -n 24 | cut -c1-420
python - <<P
import json
j=json.load(open('gpurun_out/r8laps_bench.json'))
print(round(j['ms_per_step'],2),'ms/step', j['host']['process_cpu_ms_per_step'],'CPU-ms', j['host']['waits_per_step'],'waits', j['config'].get('pipeline'), j['config'].get('groups_per_gpu'), j['config'].get('host_threads_per_rank'))
P
