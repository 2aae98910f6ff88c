# HS_TIMING laps of the contig groups over a few steps of the default job (diagnostic: where a group's chain spends its wall time)
cd /root/repo
mkdir -p gpurun_out
HS_TIMING=abs HS_BENCH_NO_PROBE=1 timeout 600 python bench.py --steps 3 --warmup 1 --cpu-contigs 0 > gpurun_out/laps_bench.json 2> gpurun_out/laps.err
grep -E "laps \(ms\)|fused call" gpurun_out/laps.err | tail -n 20 | cut -c1-400 > gpurun_out/laps_tail.txt
